// fused_mfma_kernel.h — the fused ARM block on the CDNA4 matrix cores (gfx950), fp32 end to end.
// Template + launcher; instantiated per padded embedding width (16 / 32 / 64 / 128) in fused_mfma_e*.hip.
//
// Measured facts this kernel is built around (tools/ubench/valu_rate.hip, profiles/):
//   * v_mfma_f32_16x16x4_f32 runs at the f32 VECTOR rate and its cycles ADD to the VALU cycles of
//     the same SIMD (no overlap) -> the budget per sample is 40 MFMAs * 32 cycles + every other instruction;
//   * one wave issues a VALU op at most every ~8 cycles -> >= 3-4 waves/SIMD (<= 128 VGPRs) are needed;
//   * the LDS pipe is otherwise idle -> cross-lane reductions and parameter fetches go through LDS.
//
// One WAVE owns a group of SPW samples at a time and never talks to another wave (wave-private
// LDS tile, no block barrier after the prologue).  Per group:
//
//   stage     coalesced 16-byte chunk loads (4-byte aligned: any nemb >= 4) of the F embedding rows of each sample
//             (adjacent lanes share a row), scaled by clamp(value), written to the wave's LDS tile whose rows
//             are zero-padded to E = 16/32/64/128 floats.  Rows of the NEXT group are already in flight
//             (registers) and the raw ids of the group after that are being fetched.
//   per 16-neuron pass nt:
//   MFMA #1   gates  G[(s,f), o] = X[(s,f), :] . q_fold[o, :]   (v_mfma_f32_16x16x4_f32, exact fp32).
//             Tile rows are ordered so that accumulator register r of tile t is "quarter-step"
//             q = 4t + r = s*NQ + j of ONE sample (NQ = ceil(F/4) rounded up to even): lane
//             (c = l&15, g = l>>4) holds, for neuron o = 16*nt + c, the gates of fields f = 4j + g.
//   sparse    entmax / softmax over the fields of each (sample, neuron) row, in registers, two
//   map       elements per instruction: t = clamp01(x - tau) is ONE v_pk_add_f32 with the neg and
//             clamp modifiers (x - tau <= 1 always holds because tau >= max - 1).  A row is spread
//             over the 4 lane groups g: partial sums meet through a 1 KiB LDS scratch.
//             alpha = 2: Michelot (= Newton from the left, finite), alpha = 1.5 / generic: Newton;
//             alpha > 2, n_iter < 24, ARMNET_F_FAITHFUL_BISECT: the reference's bisection, literally.
//   MFMA #2   Z^T[e, o] = sum_f X[f, e] * W[o, f]:  the C layout of MFMA #1 IS the B-operand layout
//             of MFMA #2 (k = lane group g <-> field 4j+g): the weights never move.
//   epilogue  1/sum(p) folded into the exponent scale, exp2, eval-BatchNorm affine, one 16-byte store per lane
//             and 16 embedding dims (4-byte aligned; the partial last chunk of a row as 8 bytes or elements).
//
// Shapes: nemb 4..128, nfield <= 48, nhead*nhid <= 256 per launch (neurons padded to 16 per pass).
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "armnet_common.h"

namespace armnet {

// 16 zero bytes in device memory: what a staging lane reads when its chunk is padding
static __device__ float kZeroRow[4] = {0.f, 0.f, 0.f, 0.f};   // not const: keeps the select with the table pointer in the global address space

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
// 4-byte aligned views for global memory: gfx950 runs in unaligned access mode, so these still compile to ONE
// global_load/store_dwordx4 / dwordx2 (rows of nemb floats are only 4-byte aligned when nemb is odd)
typedef f32x4 f32x4u __attribute__((aligned(4)));
typedef f32x2 f32x2u __attribute__((aligned(4)));

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }
constexpr float kLog2e = 1.44269502162933349609375f;
constexpr float kLn2 = 0.693147182464599609375f;

// t = clamp01(x - tau), two elements per instruction
__device__ __forceinline__ f32x2 pk_sub_clamp01(f32x2 x, f32x2 tau) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(x), "v"(tau));
    return r;
}
// the same with ONE threshold for both elements: op_sel_hi makes the high half read the low dword of `tau`, so the
// threshold needs no broadcast move (only tau[0] is read)
__device__ __forceinline__ f32x2 pk_sub_clamp01_lo(f32x2 x, f32x2 tau) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1] clamp" : "=v"(r) : "v"(x), "v"(tau));
    return r;
}
// clamp01(a * b), two elements per instruction (indicator of a > 0 when b is huge)
__device__ __forceinline__ f32x2 pk_mul_clamp01(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the same with the second operand in scalar registers (a constant pair that never costs a VALU move)
__device__ __forceinline__ f32x2 pk_mul_clamp01_s(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(r) : "v"(a), "s"(b));
    return r;
}
// a + b that the SLP vectoriser cannot pair up with a neighbour
__device__ __forceinline__ float vadd(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Read-once inputs (ids, values) and the write-once output with the non-temporal hint, so that they do not displace the
// embedding table — the only data this kernel reuses — from the L2 / Infinity Cache (developer switch ARMNET_NT_IO)
template <typename T>
__device__ __forceinline__ T stream_load(const T* p) {
#ifdef ARMNET_NT_IO
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
#ifdef ARMNET_NT_IO
#define ARMNET_STREAM_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define ARMNET_STREAM_STORE(ptr, val) (*(ptr) = (val))
#endif

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wave is serviced in issue order; this only stops the COMPILER from moving
    // a lane's reads across other lanes' writes.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// All-reduce of two per-lane partials over the 4 lane groups {l, l^16, l^32, l^48} through LDS:
// lane-linear 8-byte writes (conflict-free), then every lane reads the 4 partials of its column.
struct Red2 { f32x2 g0, g1, g2, g3; };
__device__ __forceinline__ void red_write(float* scratch, int slot, int lane, float a, float b) {
    *reinterpret_cast<f32x2*>(scratch + slot * 128 + lane * 2) = f32x2{a, b};
}
__device__ __forceinline__ Red2 red_read(const float* scratch, int slot, int c) {
    const float* p = scratch + slot * 128 + c * 2;
    Red2 r;
    r.g0 = *reinterpret_cast<const f32x2*>(p);
    r.g1 = *reinterpret_cast<const f32x2*>(p + 32);
    r.g2 = *reinterpret_cast<const f32x2*>(p + 64);
    r.g3 = *reinterpret_cast<const f32x2*>(p + 96);
    return r;
}

// max without the sNaN-quieting v_max x,x,x the compiler adds around fmaxf on MFMA results
__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// The asm forms above must NOT read MFMA accumulators: the compiler's hazard recognizer does not look into inline
// asm and an XDL write has to be >= 12 wait states old before a VALU read (seen as NaNs / wrong thresholds in the
// instantiations that happened to schedule a v_max3 right behind the last MFMA).  The first readers of the gate
// accumulators are therefore these compiler-visible maxima (MFMA results count as canonical: no extra v_max x,x);
// everything in asm that touches the gates afterwards depends on their result.
__device__ __forceinline__ float cmax3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float cmax2(float a, float b) { return __builtin_fmaxf(a, b); }


// E = nemb padded to 16/32/64; NQ = quarter-steps per sample (even); SPW samples per wave-group;
// SRC: 0 = int64 ids, 1 = int32 ids, 2 = pre-gathered rows; WPS = waves/SIMD the register budget targets;
// MODEL: MODEL_ARM, or one of the sibling models that share the staging, the tile layout and MFMA #2:
//   MODEL_GC_ARM  gates += their row sum before the sparse map (gc_arm.py:37-41); MFMA #2 runs on emb_bn(exp(x)) formed
//                 in registers from the tile (gc_arm.py:89); no outer exp (gc_arm.py:92-94);
//   MODEL_AFN     the tile holds log2(x); no MFMA #1 and no sparse map: the B operand of MFMA #2 is afn.weight with the
//                 emb_bn scale folded in, the emb_bn shift goes into the bias (afn.py:63-66).
//
// Waves per block (blockDim.x / 64: 4, 8, 12 or 16) is a LAUNCH parameter: the waves never talk to each other, they only
// share the block's parameter copies in LDS.  Few neurons: 4 (several blocks per CU).  Many neurons (>= 128): the
// parameter copies dominate the LDS, so ONE block of 12 waves per CU keeps 3 waves/SIMD where blocks of 4 would leave 2
// (or 1 at 256 neurons, which is why such blocks used to be cut into slices that each re-gather the rows).
// (round 3: 16 where the register budget allows 4+ waves per SIMD — 43 fields, 256 neurons: 636 -> 599 us, trained-like
// weights 859 -> 795; nothing changes where 12 waves already fit beside the parameter copies.)
constexpr int mfma_max_wpb(int wps) { return wps >= 4 ? 16 : wps >= 3 ? 12 : 8; }

// High-occupancy configurations (round 3).  The kernel overlaps two floors of similar height — the memory system's and
// the SIMDs' issue time — and how well it does so is a matter of how many waves a SIMD can switch between: for the ARM
// block the launcher therefore takes the LARGEST number of waves per SIMD whose register budget still holds the working
// set without scratch (hipcc -S of every instantiation, tools/kernel_resources.py), with one-sample groups where that
// halves the per-wave state.  {samples per wave-group, waves per SIMD}; {0, 0} = the two-sample / 3-4 wave defaults.
struct OccCfg { int spw, wps; };
constexpr OccCfg occ_config(int E, int NQ, int MODE) {
#ifdef ARMNET_OCC_VARIANT                       // developer A/B: one wave per SIMD fewer than the table
    constexpr int less = 1;
#else
    constexpr int less = 0;
#endif
    if (E == 16) {
        return NQ == 2 ? OccCfg{2, 7 - less}          // 64-70 registers
             : NQ == 4 ? OccCfg{1, 8 - less}          // 56-58
             : NQ == 6 ? OccCfg{1, 6 - less}          // 72-76 (two-sample groups: 94 + scratch at 5 waves)
             : NQ == 8 ? OccCfg{1, 6 - less}          // 78-80
             : NQ == 10 ? OccCfg{1, 5}                // 94-96 (six waves spill: 119 us)
             : NQ == 12 ? OccCfg{1, (MODE == SOLVE_BISECT ? 4 : 5) - less} : OccCfg{0, 0};
    }
    // nemb 17..32 was measured with the same rule (2-6 more waves per CU, one-sample groups for 17-32 fields) and is NOT in the
    // table: alpha = 2 loses 1-10 % (22 fields, 32 / 128 neurons: 125.7 / 410.8 us against 120.3 / 393.3), only trained-like
    // weights at alpha = 1.7 gain (up to 8 %): its 128-byte rows and 4-16 KiB of output per sample leave less to overlap.
    return OccCfg{0, 0};
}
// ... and whether such a configuration also keeps the last evaluation's clamped differences (alpha = 2 / 1.5)
constexpr bool occ_keeps(int E, int NQ, int SPW, int WPS) { return E == 16 && WPS >= 5 && occ_config(E, NQ, 0).spw == SPW; }

// F16 (round 6): both contractions on the 16-bit matrix pipe with fp16 x 2 operand splits (three products, fp32 accumulate).
// The fp32 MFMA issues at the vector unit's own rate (32 cycles per SIMD for 16x16x4) and its cycles add to the VALU's;
// v_mfma_f32_16x16x32_f16 takes ~17 for eight times the contraction depth.  Per group the wave scales its tile by a power of
// two (exact) so that the largest |x| sits near 2^10, splits x = hi + lo (fp16 each: 22 significant bits) ONCE into the A
// operands of both contractions and keeps them in registers over the passes; q_fold and `values` get block-wide scales in the
// prologue; the weights p * values are split per pass.  G = hi.qh + lo.qh + hi.ql (the lo.ql term, <= 2^-24 of the product,
// is dropped), descaled by an exact power of two in the multiply the Newton modes have anyway; the same for MFMA #2 with the
// descale folded into the exponent's factor.  One-sample groups, nemb <= 16, ARM-Net only.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
struct HiLo4 { f16x4 hi, lo; };
// 4 fp32 (already scaled) -> hi, lo with hi + lo = x to 2^-22 |x| (round to nearest both times)
__device__ __forceinline__ HiLo4 split4(f32x4 x) {
    const f32x2 a = {x[0], x[1]}, b = {x[2], x[3]};
    const f16x2v ha = __builtin_convertvector(a, f16x2v), hb = __builtin_convertvector(b, f16x2v);
    const f32x2 ra = a - __builtin_convertvector(ha, f32x2), rb = b - __builtin_convertvector(hb, f32x2);
    const f16x2v la = __builtin_convertvector(ra, f16x2v), lb = __builtin_convertvector(rb, f16x2v);
    HiLo4 r;
    r.hi = f16x4{ha[0], ha[1], hb[0], hb[1]};
    r.lo = f16x4{la[0], la[1], lb[0], lb[1]};
    return r;
}
__device__ __forceinline__ f16x8 cat8(f16x4 a, f16x4 b) { return f16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]}; }
// Every MFMA of the F16 path starts from C = 0 and the partial tiles are added on the VALU (16 v_pk_add_f32 per pass): there is no
// matrix-to-matrix dependency in it at all.  On gfx950 a v_mfma_f32_16x16x16_f16 that takes its source C from a
// v_mfma_f32_16x16x32_f16 issued fewer than 5 wait states earlier reads a partly stale C — wrong values in one or two registers of
// the tile, in place or not; the opposite order needs more — and hipcc (ROCm 7.2) inserts no wait states for such a pair
// (tools/ubench/mfma_dep_dst.hip; first seen here as errors in embedding columns 4g, 4g+1 of whole passes).  Chains that stay within one
// instruction pass that reproducer, but a build with them (2-4 % faster) returned values that differed from run to run in one
// instantiation (softmax, 29-32 fields); in-place chains written as asm behind hand-counted s_nops are correct and slower than the fp32
// form (tools/experiments/README.md).  The independent form is bit-equal run to run over every tile family, solver and id source
// (tools/scratch/r6_f16_determinism.py).
// 2^k as a float, k clamped to the normal range
__device__ __forceinline__ float pow2i(int k) {
    k = k < -126 ? -126 : k > 127 ? 127 : k;
    return __builtin_bit_cast(float, (uint32_t)(k + 127) << 23);
}
// the power-of-two scale that brings a non-negative maximum m to [2^9, 2^10): exponent k (inf / NaN: a tiny scale, the
// non-finite value then poisons what it touches as it would in fp32; zero: 0)
__device__ __forceinline__ int scale_exp(float m) {
    const int eb = (int)((__builtin_bit_cast(uint32_t, m) >> 23) & 0xffu);
    const int k = 136 - eb;
    return eb == 0 ? 0 : (k > 100 ? 100 : k);
}

// HOIST (round 6, one-sample groups of blocks with MANY 16-neuron passes): the pass-invariant A operands of both
// contractions — MFMA #1's tile rows (NTILE x EB f32x4) and MFMA #2's transposed scalars (NQ x EB) — are read from the LDS
// tile ONCE per group and kept in registers across the passes, instead of 3 + 10 LDS reads behind a fence in every pass.
template <int E, int NQ, int SPW, int MODE, int SRC, int WPS, int MODEL = MODEL_ARM, bool HOIST = false, bool F16 = false>
__global__ void __launch_bounds__(64 * mfma_max_wpb(WPS), WPS) fused_mfma_kernel(FusedArgs a) {
    static_assert(!F16 || (E == 16 && SPW == 1 && MODEL == MODEL_ARM && !HOIST && MODE != SOLVE_BISECT), "F16: one-sample groups, nemb <= 16");
    constexpr int NQT = SPW * NQ;             // quarter-steps per group
    constexpr int NTILE = (NQT + 3) / 4;      // 16-row MFMA tiles per group (last one may be half pad)
    constexpr int ES = E + 4;                 // LDS row stride (floats)
    constexpr int CF = 4;                     // floats per staging lane (16-byte chunks)
    constexpr int CH = E / CF;                // chunks per (padded) row
    constexpr int RPI = 64 / CH;              // rows per staging instruction
    constexpr int NI = NTILE * 16 / RPI;      // staging instructions per group
    constexpr int EB = E / 16;                // 16-wide blocks of the embedding dim
    constexpr int NP = NQ / 2;                // element pairs per row
    constexpr bool FROM_ROWS = (SRC == 2);
    constexpr int TILE_FLOATS = NTILE * 16 * ES;
    constexpr int WAVE_FLOATS = TILE_FLOATS + 256;       // + reduction scratch (2 slots x 128 floats)
    static_assert(E % 16 == 0 && (NTILE * 16) % RPI == 0 && NQ % 2 == 0 && SPW <= 2, "shape");
    using RowT = f32x4;
    using RowTU = f32x4u;                     // as read from global memory

    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform -> SGPR
    const int wpb = (int)(blockDim.x >> 6);                              // waves per block (launch parameter)
    const int nthreads = (int)blockDim.x;
    const int c = lane & 15, g = lane >> 4;
    const int F = a.F, O = a.O, Er = a.E;     // Er: real embedding width (<= E)
    const int O_out = a.O_out ? a.O_out : O;  // row stride of `out` in neurons (a slice of a wider block)
    const int NT = (O + 15) / 16;             // 16-neuron passes (neurons padded)
    float* xt = lds_all + wave * WAVE_FLOATS;
    float* red = xt + TILE_FLOATS;
    // block-shared, lane-ready parameters (zero-padded in e, f and o)
    float* p_bq = lds_all + wpb * WAVE_FLOATS;             // [NT][EB][64] f32x4
    float* p_vv = p_bq + NT * EB * 64 * 4;                 // [NT][NP][64] f32x2
    float* p_bn = p_vv + NT * NP * 64 * 2;                 // [NT][16] f32x2 {scale, shift}
    float* p_x = p_bn + NT * 32;                           // GC-ARM: [NQ][4] f32x2 emb_bn {scale, shift} of field 4j+g; AFN: [NT][16] bias * log2(e)

    // all group bookkeeping is 32-bit and wave-uniform (SALU): launcher guarantees B*F*8 < 2^32
    const int Bi = (int)a.B;
    const uint32_t BF = (uint32_t)Bi * (uint32_t)F;
    const int ngroups = (Bi + SPW - 1) / SPW;
    const int nwaves = (int)gridDim.x * wpb;
    int grp = (int)blockIdx.x * wpb + wave;

    // ---- group-invariant staging geometry: which (sample, field) each staging lane fetches -------
    const int chunk = lane % CH;
    // a chunk is full (inside the row), partial (the row ends inside it; it is loaded from the LAST 16
    // bytes of the row instead and rotated into place) or padding (reads zeros)
    const int rem = Er & 3;                              // floats of the partial chunk (0: there is none)
    const bool chunk_part = rem != 0 && chunk == (Er >> 2);
    const bool chunk_ok = (chunk + 1) * CF <= Er || chunk_part;
    uint32_t off4[NI];   // 4 * (s*F + f) of the row this lane stages in instruction n (0 for a pad row)
    bool pad[NI];        // pad row (field >= nfield, or a quarter-step past the group): zeroed after staging
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int row = n * RPI + lane / CH;
        const int t = row >> 4, i = row & 15;
        const int q = 4 * t + (i & 3);
        const int s = q / NQ, j = q - s * NQ;
        const int f = 4 * j + (i >> 2);
        pad[n] = !(q < NQT && f < F);
        off4[n] = (q < NQT && f < F) ? 4u * (uint32_t)(s * F + f) : 0u;
    }
    // this lane's last two quarter-steps may be pad fields: their gates become -inf
    const float padneg_a = ((4 * (NQ - 1) + g) >= F) ? -INFINITY : 0.f;
    const float padneg_b = (NQ >= 2 && (4 * (NQ - 2) + g) >= F) ? -INFINITY : 0.f;
    const bool two_pad = F <= 4 * (NQ - 1);              // the even rounding of NQ added a whole pad quarter-step
    const bool write_vals = (a.flags & ARMNET_F_WRITE_CLAMPED_VALS) != 0;
    const bool check_ids = a.id_status != nullptr;
    // ablation switches for profiling (tools/kbench.py against a `make EXTRA=-DARMNET_DEV_FLAGS` build)
#ifdef ARMNET_DEV_FLAGS
    const bool dbg_no_solve = (a.flags & 0x100u) != 0;   // skip the Newton iterations
    const bool dbg_hot_rows = (a.flags & 0x200u) != 0;   // fold ids into 1024 rows (cache-resident gather)
    const bool dbg_no_store = (a.flags & 0x400u) != 0;   // skip the output stores
    const bool dbg_no_mfma = (a.flags & 0x800u) != 0;    // replace the MFMAs by register copies
#else
    // product build: the switches are compile-time false (a runtime test in front of every MFMA of the unrolled
    // second contraction splits it into 20 basic blocks)
    constexpr bool dbg_no_solve = false, dbg_hot_rows = false, dbg_no_store = false, dbg_no_mfma = false;
#endif
    const uint32_t id_mask = dbg_hot_rows ? 1023u : 0xffffffffu;
    const uint32_t id_max = (uint32_t)a.nfeat - 1u;
    // lanes whose chunk lies in the zero padding of a row (nemb < E) read zeros: lane-constant base and stride
    const uint32_t row_bytes = chunk_ok ? (uint32_t)Er * 4u : 0u;
    const char* row_base = chunk_ok ? reinterpret_cast<const char*>(FROM_ROWS ? a.rows : a.table) +
                                          (chunk_part ? (Er - 4) * 4 : chunk * 16)
                                    : reinterpret_cast<const char*>(kZeroRow);
    // pad rows are staged like any other (their lanes re-read element 0 of the group) and then overwritten with
    // zeros, so that a non-finite embedding of one sample cannot leak into its group neighbour through 0 * NaN.
    // Up to 7 pad fields per sample (nfield rounded up to 4*NQ) + the quarter-steps past the group.
    constexpr int XQ = 4 * NTILE - NQT;                                  // quarter-steps past the group
    constexpr int NZ = ((SPW * 7 + 4 * XQ) * (E / 4) + 63) / 64;         // zeroing instructions (upper bound)
    const int npf = 4 * NQ - F;                                          // pad fields per sample
    int zoff[NZ];                                                        // LDS float offset or -1
#pragma unroll
    for (int m = 0; m < NZ; ++m) {
        const int idx = lane + 64 * m;
        const int cc = idx % (E / 4), kk = idx / (E / 4);
        int q, gg;
        if (kk < SPW * npf) {
            const int sz = kk / (npf > 0 ? npf : 1), f = F + kk - sz * npf;
            q = sz * NQ + (f >> 2);
            gg = f & 3;
        } else {
            const int x = kk - SPW * npf;
            q = NQT + (x >> 2);
            gg = x & 3;
        }
        zoff[m] = (q < 4 * NTILE) ? ((q >> 2) * 16 + 4 * gg + (q & 3)) * ES + 4 * cc : -1;
    }
    const bool any_pad = SPW * npf + 4 * XQ > 0;
    const bool full_rows = (Er == E);                    // 16-byte stores possible
    const uint32_t lane_out_off = (uint32_t)c * (uint32_t)Er + 4u * (uint32_t)g;   // floats; per-lane part of every store
    const uint32_t F4 = 4u * (uint32_t)F;

    // ---- software pipeline -------------------------------------------------------------------------
    // iteration k:  stage rows(k) -> LDS | first pass of compute(k) | range-check ids(k+1), issue row + value
    //               loads(k+1) | issue RAW id loads(k+2) | stores, further passes.  Nothing loaded in an iteration
    //               is looked at before the next one.  All per-group addressing is scalar base (SALU) + a per-lane
    //               constant offset; groups past the end re-read the last group, and in a short last
    //               group the lanes of the missing sample re-read sample 0 (results never stored).
    RowT rows_cur[NI];
    float val_cur[NI];
    uint32_t raw_lo[NI], raw_hi[NI];

    auto lane_off = [&](int n, bool short_grp) -> uint32_t {
        return (short_grp && off4[n] >= F4) ? off4[n] - F4 : off4[n];
    };
    // One-sample groups fetch the sample's ids and values with ONE coalesced load each (lane f holds field f) and hand
    // them to the four staging lanes of every row through ds_bpermute (cross-lane, no LDS memory): 3 + 2 vector-memory
    // instructions per group instead of 3 + 3 + 3, in a kernel whose waves queue at vector-memory issue.
#ifdef ARMNET_NO_COALESCED_IO
    constexpr bool CO = false;
#else
    constexpr bool CO = (SPW == 1 && !FROM_ROWS);
#endif
    const uint32_t co_off = 4u * (uint32_t)(lane < F ? lane : F - 1);
    auto fetch_raw = [&](int gidx) {
        if constexpr (!FROM_ROWS) {
            const int gc = gidx < ngroups ? gidx : ngroups - 1;
            const uint32_t e0 = (uint32_t)(gc * SPW) * (uint32_t)F;
            const bool short_grp = gc * SPW + SPW > Bi;              // wave-uniform, true at most once
            const char* ids_g = reinterpret_cast<const char*>(a.ids) + (size_t)e0 * (SRC == 0 ? 8 : 4);
            if constexpr (CO) {
                if constexpr (SRC == 0) {
                    const u32x2 w = stream_load(reinterpret_cast<const u32x2*>(ids_g + (co_off << 1)));
                    raw_lo[0] = w[0];
                    raw_hi[0] = w[1];
                } else {
                    raw_lo[0] = stream_load(reinterpret_cast<const uint32_t*>(ids_g + co_off));
                    raw_hi[0] = 0u;
                }
                return;
            }
            auto body = [&](auto is_short) {                         // two copies: the common one has no per-lane select
#pragma unroll
                for (int n = 0; n < NI; ++n) {
                    const uint32_t o = decltype(is_short)::value ? lane_off(n, true) : off4[n];
                    if constexpr (SRC == 0) {
                        const u32x2 w = stream_load(reinterpret_cast<const u32x2*>(ids_g + (o << 1)));
                        raw_lo[n] = w[0];
                        raw_hi[n] = w[1];
                    } else {
                        raw_lo[n] = stream_load(reinterpret_cast<const uint32_t*>(ids_g + o));
                        raw_hi[n] = 0u;
                    }
                }
            };
            if (__builtin_expect(short_grp, 0)) body(std::true_type{});
            else body(std::false_type{});
        }
    };
    auto issue_rows_vals = [&](int gidx) {
        const int gc = gidx < ngroups ? gidx : ngroups - 1;
        const uint32_t e0 = (uint32_t)(gc * SPW) * (uint32_t)F;
        const bool short_grp = gc * SPW + SPW > Bi;
        const char* vals_g = reinterpret_cast<const char*>(a.vals) + (size_t)e0 * 4;
        if constexpr (CO) {
            if (check_ids && (raw_hi[0] != 0u || raw_lo[0] > id_max)) flag_bad_id(a.id_status);   // lanes >= F repeat field F-1
            val_cur[0] = stream_load(reinterpret_cast<const float*>(vals_g + co_off));   // distributed at staging time
            const int idc = (int)(min(raw_lo[0], id_max) & id_mask);                   // memory-safe even when unchecked
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const uint32_t id = (uint32_t)__builtin_amdgcn_ds_bpermute((int)off4[n], idc);      // off4 = 4 * field = its lane's byte address
                rows_cur[n] = *reinterpret_cast<const RowTU*>(row_base + (size_t)id * row_bytes);
            }
            return;
        }
        if constexpr (!FROM_ROWS) {
            if (check_ids) {                                         // wave-uniform branch
                bool bad = false;
#pragma unroll
                for (int n = 0; n < NI; ++n) bad |= !pad[n] && (raw_hi[n] != 0u || raw_lo[n] > id_max);
                if (bad && chunk == 0) flag_bad_id(a.id_status);
            }
        }
        auto body = [&](auto is_short) {                             // two copies: the common one has no per-lane select
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const uint32_t o = decltype(is_short)::value ? lane_off(n, true) : off4[n];
                val_cur[n] = stream_load(reinterpret_cast<const float*>(vals_g + o));
                const char* src;
                if constexpr (FROM_ROWS) {
                    src = row_base + (size_t)(e0 + (o >> 2)) * row_bytes;
                } else {
                    const uint32_t id = min(raw_lo[n], id_max) & id_mask;   // memory-safe even when unchecked
                    src = row_base + (size_t)id * row_bytes;
                }
                rows_cur[n] = *reinterpret_cast<const RowTU*>(src);
            }
        };
        if (__builtin_expect(short_grp, 0)) body(std::true_type{});
        else body(std::false_type{});
    };

    if (grp < ngroups) fetch_raw(grp);        // first dependent load of the pipeline: issue before anything else
    [[maybe_unused]] int kq = 0, kw = 0;      // F16: exponents of the block's scales of q_fold and `values`
    if constexpr (F16) {
        float mq = 0.f, mv = 0.f;
        for (int i = threadIdx.x; i < O * Er; i += nthreads) mq = __builtin_fmaxf(mq, __builtin_fabsf(a.q_fold[i]));
        for (int i = threadIdx.x; i < O * F; i += nthreads) mv = __builtin_fmaxf(mv, __builtin_fabsf(a.values[i]));
        uint32_t* sc = reinterpret_cast<uint32_t*>(lds_all);       // wave 0's tile: nobody stages before the barrier below
        if (threadIdx.x < 2) sc[threadIdx.x] = 0u;
        __syncthreads();
        atomicMax(sc + 0, __builtin_bit_cast(uint32_t, mq));       // non-negative floats order like their bit patterns
        atomicMax(sc + 1, __builtin_bit_cast(uint32_t, mv));
        __syncthreads();
        kq = scale_exp(__builtin_bit_cast(float, sc[0]));
        kw = scale_exp(__builtin_bit_cast(float, sc[1]));
        __syncthreads();
    }
    for (int i = threadIdx.x; MODEL != MODEL_AFN && i < NT * EB * 64; i += nthreads) {
        const int l = i & 63, kb = (i >> 6) % EB, nt = (i >> 6) / EB;
        const int o = 16 * nt + (l & 15);
        const int e0 = 16 * kb + 4 * (l >> 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (o < O)
            for (int r = 0; r < 4; ++r)
                if (e0 + r < Er) v[r] = a.q_fold[(size_t)o * Er + e0 + r];
        if constexpr (F16) {                       // the same 16 bytes: {hi x 4, lo x 4} of 2^kq * q_fold
            const HiLo4 h = split4(v * pow2i(kq));
            v = __builtin_bit_cast(f32x4, cat8(h.hi, h.lo));
        }
        *reinterpret_cast<f32x4*>(p_bq + i * 4) = v;
    }
    for (int i = threadIdx.x; i < NT * NP * 64; i += nthreads) {
        const int l = i & 63, jp = (i >> 6) % NP, nt = (i >> 6) / NP;
        const int o = 16 * nt + (l & 15);
        const int f0 = 4 * (2 * jp) + (l >> 4), f1 = f0 + 4;
        f32x2 v;
        v[0] = (o < O && f0 < F) ? a.values[(size_t)o * F + f0] : 0.f;
        v[1] = (o < O && f1 < F) ? a.values[(size_t)o * F + f1] : 0.f;
        if constexpr (MODEL == MODEL_AFN) {
            // the tile holds log2(x): W[o,f] * (ln(x) * s_f + t_f) = (W[o,f] * s_f * ln2) * log2(x) + W[o,f] * t_f
            if (f0 < F) v[0] *= a.emb_scale[f0] * kLn2;
            if (f1 < F) v[1] *= a.emb_scale[f1] * kLn2;
        }
        if constexpr (F16) v *= pow2i(kw);         // the weights p * values come out scaled
        *reinterpret_cast<f32x2*>(p_vv + i * 2) = v;
    }
    if constexpr (MODEL == MODEL_GC_ARM) {
        for (int i = threadIdx.x; i < NQ * 4; i += nthreads)      // pad fields: 0 * exp(0) + 0 = 0
            *reinterpret_cast<f32x2*>(p_x + i * 2) = i < F ? f32x2{a.emb_scale[i], a.emb_shift[i]} : f32x2{0.f, 0.f};
    }
    if constexpr (MODEL == MODEL_AFN) {
        for (int o = threadIdx.x; o < NT * 16; o += nthreads) {
            float bsum = 0.f;
            if (o < O) {
                bsum = a.lin_bias[o];
                for (int f = 0; f < F; ++f) bsum = fmaf(a.values[(size_t)o * F + f], a.emb_shift[f], bsum);
            }
            p_x[o] = bsum * kLog2e;
        }
    }
    for (int i = threadIdx.x; i < NT * 16; i += nthreads)
        *reinterpret_cast<f32x2*>(p_bn + i * 2) = i < O ? f32x2{a.bn_scale[i], a.bn_shift[i]} : f32x2{0.f, 0.f};
    __syncthreads();
    if (grp >= ngroups) return;
    issue_rows_vals(grp);
    fetch_raw(grp + nwaves);
#ifdef ARMNET_PHASE_TIMING
    // developer build: per-phase s_memtime deltas, summed over all waves into id_status[0..7] (as uint32)
    // 0: what is left of the staging phase, 1: MFMA #1, 2: row statistics, 3: solver, 4: weights + MFMA #2, 5: epilogue + stores,
    // 6: wait for the group's rows, 7: staging (clamp, scale, LDS writes), 8: wait for the next group's ids,
    // 9: ISSUE of its value / row loads, 10: ISSUE of the id loads of the group after
    unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#define PHASE(i) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ph_acc[i] += _n - ph_t; ph_t = _n; } while (0)
#else
#define PHASE(i) do {} while (0)
#endif

    const float am1 = a.cfg.am1;
    const float rr = a.cfg.r, rm1 = a.cfg.r - 1.0f;
    const float invF = 1.0f / (float)F;
    const float tau_off = a.cfg.tau_hi_off;
    const float tau_tol = a.cfg.tau_tol;                  // generic alpha: step tolerance, scaled by alpha - 1 below 1.7
    const float lin_tol = a.cfg.lin_tol;                  // generic alpha / 1.5: a step below this needs no confirming evaluation
    const float L2E = kLog2e;

    for (; grp < ngroups; grp += nwaves) {
        const int b0 = grp * SPW;
        // ---- stage the current group's rows (scaled) into the wave's LDS tile -----------------------
        wave_lds_fence();
#ifdef ARMNET_PHASE_TIMING
#pragma unroll
        for (int n = 0; n < NI; ++n) asm volatile("" : "+v"(rows_cur[n]), "+v"(val_cur[n]));    // all of this group's loads have landed
        PHASE(6);
#endif
        bool changed = false;
        float vcl[NI];
        [[maybe_unused]] float xmax = 0.f;             // F16: largest |x| this lane staged
        if constexpr (CO) {
            const int vco = __builtin_bit_cast(int, val_cur[0]);
#pragma unroll
            for (int n = NI - 1; n >= 0; --n) val_cur[n] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((int)off4[n], vco));
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            // clamp (armnet_1h.py:81; NaN stays NaN), scale (layers.py:21)
            const float vraw = val_cur[n];
            float v = __builtin_amdgcn_fmed3f(vraw, 1e-3f, 1.0f);
            v = (vraw != vraw) ? vraw : v;
            vcl[n] = v;
            changed |= (v != vraw) && !pad[n];
            RowT r = rows_cur[n] * v;
            if constexpr (MODEL == MODEL_AFN) {
                // afn.py:63 log(x_emb) as log2 (ln2 is folded into the weights); x > 0 after embedding_clip, a
                // negative or NaN entry gives NaN as in the reference.  Lanes of the row padding keep zeros.
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = chunk_ok ? __builtin_amdgcn_logf(r[q]) : 0.f;
            }
            {
                if (rem != 0) {                                         // kernel-uniform
                    // partial chunk: the lane holds the row's last 4 floats; its own are the last `rem` of them
                    const f32x4 t = rem == 1 ? f32x4{r[3], 0.f, 0.f, 0.f}
                                  : rem == 2 ? f32x4{r[2], r[3], 0.f, 0.f} : f32x4{r[1], r[2], r[3], 0.f};
                    r = chunk_part ? t : r;
                }
            }
            const int row = n * RPI + lane / CH;
            *reinterpret_cast<RowT*>(xt + row * ES + chunk * CF) = r;
            if constexpr (F16) {
                if (!pad[n])                               // (a pad row is zeroed below; its lanes re-read a real row)
                    xmax = __builtin_fmaxf(__builtin_fmaxf(xmax, __builtin_fmaxf(__builtin_fabsf(r[0]), __builtin_fabsf(r[1]))),
                                           __builtin_fmaxf(__builtin_fabsf(r[2]), __builtin_fabsf(r[3])));
            }
        }
        if (any_pad) {
#pragma unroll
            for (int m = 0; m < NZ; ++m)
                if (zoff[m] >= 0) *reinterpret_cast<f32x4*>(xt + zoff[m]) = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // the reference's in-place clamp_: rare (a value outside [1e-3, 1]), so one wave-uniform test
        if (write_vals && __builtin_amdgcn_ballot_w64(changed)) {
            const uint32_t e0 = (uint32_t)b0 * (uint32_t)F;
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                const uint32_t idx = e0 + (off4[n] >> 2);
                if (vcl[n] != val_cur[n] && chunk == 0 && !pad[n] && idx < BF) a.vals[idx] = vcl[n];
            }
        }
#ifdef ARMNET_PHASE_TIMING
        wave_lds_fence();
        PHASE(7);
#endif
        // ---- keep the memory pipeline full: rows of the next group, raw ids of the one after -----------------------
        // (one-sample id-fed groups of up to 40 fields at nemb <= 16 do this behind their first pass instead: LATE below)
        auto prefetch = [&]() {
#ifdef ARMNET_PHASE_TIMING
            if constexpr (!FROM_ROWS) {
#pragma unroll
                for (int n = 0; n < NI; ++n) asm volatile("" : "+v"(raw_lo[n]), "+v"(raw_hi[n]));        // the ids have landed
            }
            PHASE(8);
#endif
            issue_rows_vals(grp + nwaves);
#ifdef ARMNET_PHASE_TIMING
            PHASE(9);
#endif
            fetch_raw(grp + 2 * nwaves);
#ifdef ARMNET_PHASE_TIMING
            PHASE(10);
#endif
        };
        constexpr bool LATE = (E == 16 && SPW == 1 && NQ <= 10 && !FROM_ROWS);
        if constexpr (!LATE) prefetch();
        wave_lds_fence();
        PHASE(0);

        // F16: the group's scale, then the A operands of both contractions as {hi x 4, lo x 4}, kept over the passes
        [[maybe_unused]] f16x8 a1f[F16 ? NTILE : 1];    // MFMA #1: tile row 16t + c, embedding columns 4g .. 4g+3
        [[maybe_unused]] f16x8 a2f[F16 ? NTILE : 1];    // MFMA #2: column c of the tile rows 16J + 4g .. + 3 (quarter-steps 4J .. 4J+3 of field group g)
        [[maybe_unused]] float dsc_g = 1.f, dsc_z = 1.f;
        if constexpr (F16) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) xmax = __builtin_fmaxf(xmax, __shfl_xor(xmax, off, 64));
            const int kx = __builtin_amdgcn_readfirstlane(scale_exp(xmax));
            const float sx = pow2i(kx);
            dsc_g = pow2i(-kx) * pow2i(-kq);
            dsc_z = pow2i(-kx) * pow2i(-kw);
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
                const HiLo4 h = split4(*reinterpret_cast<const f32x4*>(xt + (16 * t + c) * ES + 4 * g) * sx);
                a1f[t] = cat8(h.hi, h.lo);
                f32x4 col;
#pragma unroll
                for (int i = 0; i < 4; ++i) col[i] = xt[(16 * t + 4 * g + i) * ES + c];
                const HiLo4 k = split4(col * sx);
                a2f[t] = cat8(k.hi, k.lo);
            }
        }
        f32x4 avh[HOIST ? EB * NTILE : 1];
        float a2h[HOIST ? NQ * EB * SPW : 1];
        if constexpr (HOIST) {
#pragma unroll
            for (int kb = 0; kb < EB; ++kb)
#pragma unroll
                for (int t = 0; t < NTILE; ++t)
                    avh[kb * NTILE + t] = *reinterpret_cast<const f32x4*>(xt + (16 * t + c) * ES + 16 * kb + 4 * g);
#pragma unroll
            for (int j = 0; j < NQ; ++j)
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        const int q = s * NQ + j;
                        a2h[(j * EB + eb) * SPW + s] = xt[(16 * (q >> 2) + 4 * g + (q & 3)) * ES + 16 * eb + c];
                    }
        }
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 c1[NTILE];                  // gates, then the B operand of MFMA #2
            float kexp[SPW];                  // scale of the contraction's result
            // element j of sample s; pairs (2jp, 2jp+1) are register-pair aligned because NQ is even
#define XG(s, j) c1[((s) * NQ + (j)) >> 2][((s) * NQ + (j)) & 3]
#define XP_GET(s, jp) (f32x2{XG(s, 2 * (jp)), XG(s, 2 * (jp) + 1)})
#define XP_SET(s, jp, v)              \
    do {                              \
        const f32x2 _v = (v);         \
        XG(s, 2 * (jp)) = _v[0];      \
        XG(s, 2 * (jp) + 1) = _v[1];  \
    } while (0)
            const float* vv_base = p_vv + (nt * NP * 64 + lane) * 2;
#define VV(jp) (*reinterpret_cast<const f32x2*>(vv_base + (jp) * 128))
            if constexpr (MODEL == MODEL_AFN) {
                // afn.py:64: a plain Linear over the fields: the B operand is the (folded) weight itself
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) XG(s, j) = VV(j >> 1)[j & 1];
                    kexp[s] = L2E;
                }
            } else {
                // ---- MFMA #1: gates; the NTILE accumulator chains are interleaved (40-cycle dependent latency)
                if constexpr (F16) {
                    const f16x8 bq = __builtin_bit_cast(f16x8, *reinterpret_cast<const f32x4*>(p_bq + (nt * 64 + lane) * 4));
                    const f16x4 qh = {bq[0], bq[1], bq[2], bq[3]}, ql = {bq[4], bq[5], bq[6], bq[7]};
                    const f16x8 qhh = cat8(qh, qh);
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    f32x4 part[NTILE];
#pragma unroll
                    for (int t = 0; t < NTILE; ++t)         // (x_hi | x_lo) . (q_hi | q_hi)
                        c1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1f[t], qhh, zero, 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NTILE; ++t) {       // x_hi . q_lo
                        const f16x4 xh = {a1f[t][0], a1f[t][1], a1f[t][2], a1f[t][3]};
                        part[t] = __builtin_amdgcn_mfma_f32_16x16x16f16(xh, ql, zero, 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < NTILE; ++t) c1[t] += part[t];
                }
#pragma unroll
                for (int kb = 0; kb < (F16 ? 0 : EB); ++kb) {
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(p_bq + ((nt * EB + kb) * 64 + lane) * 4);
                    f32x4 av[NTILE];
#pragma unroll
                    for (int t = 0; t < NTILE; ++t) {
                        if constexpr (HOIST) av[t] = avh[kb * NTILE + t];
                        else av[t] = *reinterpret_cast<const f32x4*>(xt + (16 * t + c) * ES + 16 * kb + 4 * g);
                    }
                    if (dbg_no_mfma) {
#pragma unroll
                        for (int t = 0; t < NTILE; ++t) c1[t] = av[t] * bq;
                        continue;
                    }
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int t = 0; t < NTILE; ++t) {
                            if (kb == 0 && kk == 0)
                                c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], bq[kk], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            else
                                c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], bq[kk], c1[t], 0, 0, 0);
                        }
                }
                PHASE(1);

                // ---- sparse map over the fields ------------------------------------------------------------
                // On exit XG holds the UNNORMALISED weights p * values and kexp[s] = log2(e) / sum(p).
                // row sum (for the mean start / NaN detection) and row max, partials to LDS
                wave_lds_fence();
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    f32x2 sm2;
#pragma unroll
                    for (int jp = 0; jp < NP; ++jp) {
                        f32x2 x = XP_GET(s, jp);
                        if constexpr (MODE != SOLVE_MICHELOT && MODE != SOLVE_SOFTMAX) {
                            const float m = F16 ? am1 * dsc_g : am1;      // (a power of two: the same rounding as descale, then * am1)
                            x *= f32x2{m, m};                   // entmax.py:42
                            XP_SET(s, jp, x);
                        } else if constexpr (F16) {
                            x *= f32x2{dsc_g, dsc_g};
                            XP_SET(s, jp, x);
                        }
                        sm2 = jp == 0 ? x : sm2 + x;            // a pad field's gate is exactly 0
                    }
                    XG(s, NQ - 1) += padneg_a;                  // pad fields -> -inf (never in the support)
                    if (two_pad) XG(s, NQ - 2) += padneg_b;     // wave-uniform
                    float mx;
                    if constexpr (NQ == 2) {                    // compiler-visible: first readers of the accumulators
                        mx = cmax2(XG(s, 0), XG(s, 1));
                    } else {
                        mx = cmax3(XG(s, 0), XG(s, 1), XG(s, 2));
#pragma unroll
                        for (int j = 3; j + 1 < NQ; j += 2) mx = cmax3(mx, XG(s, j), XG(s, j + 1));
                        mx = cmax2(mx, XG(s, NQ - 1));
                    }
                    red_write(red, s & 1, lane, mx, sm2[0] + sm2[1]);
                }
                wave_lds_fence();
                float tau[SPW], Ssum[SPW];
                float tau_hi[MODE == SOLVE_BISECT ? SPW : 1];
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    const Red2 r = red_read(red, s & 1, c);
                    float mx = vmax2(vmax3(r.g0[0], r.g1[0], r.g2[0]), r.g3[0]);
                    float sm = (r.g0[1] + r.g1[1]) + (r.g2[1] + r.g3[1]);
                    if constexpr (MODEL == MODEL_GC_ARM) {
                        // gc_arm.py:37-41: the global context is the bilinear form of the field SUM = the sum of the
                        // gates (already scaled by alpha - 1 here); added to every gate of the row (pads stay -inf)
                        const float gcx = sm;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) XG(s, j) += gcx;
                        mx += gcx;
                        sm = fmaf((float)F, gcx, sm);
                    }
                    if constexpr (MODE == SOLVE_SOFTMAX) {
                        tau[s] = mx + (sm - sm);                // NaN / inf anywhere -> NaN row
                    } else if constexpr (MODE == SOLVE_BISECT) {
                        tau[s] = (mx - 1.0f) + (sm - sm);       // tau_lo (entmax.py:46); NaN / inf anywhere -> NaN row
                        tau_hi[s] = mx - tau_off;               // entmax.py:47
                    } else {
                        // tau0 = max(mx - 1, mean - d^-(alpha-1)) <= root; NaN/inf gates poison the row
                        tau[s] = vmax2(mx - 1.0f, fmaf(sm, invF, -tau_off)) + (sm - sm);
                    }
                    Ssum[s] = 1.0f;
                }
                PHASE(2);
                if constexpr (MODE == SOLVE_SOFTMAX) {
                    wave_lds_fence();
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        float S = 0.f;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            const float p = __builtin_amdgcn_exp2f((XG(s, j) - tau[s]) * L2E);
                            S += p;
                            XG(s, j) = p * VV(j >> 1)[j & 1];
                        }
                        red_write(red, s & 1, lane, S, 0.f);
                    }
                    wave_lds_fence();
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        const Red2 q = red_read(red, s & 1, c);
                        Ssum[s] = (q.g0[0] + q.g1[0]) + (q.g2[0] + q.g3[0]);
                    }
                } else if constexpr (MODE == SOLVE_BISECT) {
                    // The reference's bisection, statement for statement (utils/entmax.py:49-64), for alpha > 2, n_iter < 24
                    // and ARMNET_F_FAITHFUL_BISECT: f_lo at tau_lo, then n_iter halvings; p is the one of the LAST tau_m.
                    // t^(1/(alpha-1)) through the hardware log2/exp2 pair (2 transcendentals per element and step).
                    f32x2 pkeep[SPW * NP];
                    float f_lo[SPW], dm[SPW];
                    auto eval = [&](int s_, float t_at) -> float {          // p(t_at) into pkeep, partial row sum
                        const f32x2 tk = {t_at, t_at};
                        f32x2 S2;
#pragma unroll
                        for (int jp = 0; jp < NP; ++jp) {
                            const f32x2 t = pk_sub_clamp01(XP_GET(s_, jp), tk);     // tau >= max - 1  =>  x - tau <= 1
                            f32x2 pv;
                            pv[0] = __builtin_amdgcn_exp2f(rr * __builtin_amdgcn_logf(t[0]));   // log2(0) = -inf -> 0
                            pv[1] = __builtin_amdgcn_exp2f(rr * __builtin_amdgcn_logf(t[1]));
                            pkeep[s_ * NP + jp] = pv;
                            S2 = jp == 0 ? pv : S2 + pv;
                        }
                        return S2[0] + S2[1];
                    };
                    wave_lds_fence();
#pragma unroll
                    for (int s = 0; s < SPW; ++s) red_write(red, s & 1, lane, eval(s, tau[s]), 0.f);
                    wave_lds_fence();
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        const Red2 r = red_read(red, s & 1, c);
                        Ssum[s] = (r.g0[0] + r.g1[0]) + (r.g2[0] + r.g3[0]);
                        f_lo[s] = Ssum[s] - 1.0f;
                        dm[s] = tau_hi[s] - tau[s];
                    }
                    // a sample whose rows have all settled (tau_lo + dm == tau_lo: every later step would evaluate the same
                    // tau_m again) leaves the loop on its own; its p and S are those of its last evaluation
#ifdef ARMNET_SAMPLE_EXIT
                    constexpr bool kBisectSampleExit = true;
#else
                    constexpr bool kBisectSampleExit = false;
#endif
                    unsigned long long live[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) live[s] = ~0ull;
                    for (int it = 0; it < a.cfg.n_iter; ++it) {
                        float tm[SPW];
                        wave_lds_fence();
#pragma unroll
                        for (int s = 0; s < SPW; ++s) {
                            if (kBisectSampleExit && SPW > 1 && !live[s]) continue;                  // scalar branch
                            dm[s] *= 0.5f;
                            tm[s] = tau[s] + dm[s];
                            red_write(red, s & 1, lane, eval(s, tm[s]), 0.f);
                        }
                        wave_lds_fence();
                        unsigned long long any_live = 0;
#pragma unroll
                        for (int s = 0; s < SPW; ++s) {
                            if (kBisectSampleExit && SPW > 1 && !live[s]) continue;
                            const Red2 r = red_read(red, s & 1, c);
                            Ssum[s] = (r.g0[0] + r.g1[0]) + (r.g2[0] + r.g3[0]);
                            const float f_m = Ssum[s] - 1.0f;
                            // NaN rows never settle: all n_iter steps
                            live[s] = __builtin_amdgcn_ballot_w64(!(tm[s] == tau[s]));
                            tau[s] = (f_m * f_lo[s] >= 0.f) ? tm[s] : tau[s];
                            any_live |= live[s];
                        }
                        // once tau_lo + dm rounds to tau_lo in every row of the wave (dm below half an ulp: ~25 steps),
                        // every later step would evaluate the same tau_m again: stopping here is bit-identical
                        if (!any_live) break;
                    }
                    PHASE(3);
#pragma unroll
                    for (int s = 0; s < SPW; ++s)
#pragma unroll
                        for (int jp = 0; jp < NP; ++jp) XP_SET(s, jp, pkeep[s * NP + jp] * VV(jp));
                } else {
                    // Newton from the left on f(tau) = sum p(tau) - 1; wave-uniform loop, rows go passive as
                    // they converge.  When the loop ends every row's S was evaluated at its final threshold.
                    // generic alpha: p = t^r of the LAST evaluation is kept (the loop always ends on an evaluation
                    // at the final threshold), which saves the two transcendentals per element of a final pass
                    // alpha = 2 with a 168-register budget (<= 3 waves/SIMD), or one of the high-occupancy configurations of
                    // occ_config (few pairs per lane): the clamped differences of the last evaluation are kept as well,
                    // which saves their recomputation in the weight pass
                    constexpr bool KEEP = (MODE == SOLVE_NEWTON) || ((MODE == SOLVE_MICHELOT || MODE == SOLVE_NEWTON15) && (WPS <= 3 || occ_keeps(E, NQ, SPW, WPS)));
                    f32x2 pkeep[KEEP ? SPW * NP : 1];
                    // Round 5 — the last evaluation of a row only CONFIRMS a step that was already tiny.  Newton converges
                    // quadratically from the left (trained-like weights, alpha = 1.7: f = 0.8, 0.13, 9e-3, 1.3e-4, 1.6e-7),
                    // so once the step f / D is below `lin_tol` the threshold it leads to is final to O(step^2) and p there
                    // is known to first order from what this evaluation already holds: p - r t^(r-1) step (clamped at 0;
                    // alpha = 1.5: (t - step)^2 recomputed exactly).  Such a row leaves the loop with the step taken and the
                    // correction PENDING; if other rows keep the wave going it is simply evaluated again (exactly) and the
                    // correction is dropped; if the loop ends, one pass over the kept values applies it and sum(p) is the
                    // Newton model's 1.  tools/solver_sim_r5.py: 5.23 -> 4.5 evaluations per pass at alpha = 1.7, 4.57 -> 3.8
                    // at alpha = 1.5 (trained-like weights); random-init weights take one evaluation either way.
#ifdef ARMNET_NO_LIN
                    constexpr bool LIN = false;
#else
                    constexpr bool LIN = (MODE == SOLVE_NEWTON || MODE == SOLVE_NEWTON15);
#endif
                    constexpr bool LIN_GEN = LIN && MODE == SOLVE_NEWTON;
                    f32x2 ukeep[LIN_GEN ? SPW * NP : 1];
                    float dlin[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) dlin[s] = 0.f;
                    // A sample whose 16 rows have all converged leaves the loop on its own (wave-uniform masks on the scalar
                    // unit): with sparse supports the two samples of a group rarely finish in the same step.  Its tau, S and
                    // kept clamped differences are those of ITS last evaluation, which was at its final thresholds.
#ifdef ARMNET_SAMPLE_EXIT
                    constexpr bool kSampleExit = true;
#else
                    constexpr bool kSampleExit = false;
#endif
                    unsigned long long live[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) live[s] = ~0ull;
                    const f32x2 huge2 = {0x1p120f, 0x1p120f};
                    f32x2 tau2[SPW];
#pragma unroll
                    for (int s = 0; s < SPW; ++s) tau2[s] = f32x2{tau[s], tau[s]};
                    for (int it = 0; it < kNewtonMaxIter; ++it) {
                        wave_lds_fence();
#pragma unroll
                        for (int s = 0; s < SPW; ++s) {
                            if (kSampleExit && SPW > 1 && !live[s]) continue;          // scalar branch
                            tau2[s][0] = tau[s];                        // only the low half is read (op_sel_hi)
                            f32x2 S2, D2;
#pragma unroll
                            for (int jp = 0; jp < NP; ++jp) {
                                const f32x2 t = pk_sub_clamp01_lo(XP_GET(s, jp), tau2[s]);
                                f32x2 sv, dv;
                                if constexpr (MODE == SOLVE_MICHELOT) {
                                    sv = t;
                                    dv = pk_mul_clamp01_s(t, huge2);
                                    if constexpr (KEEP) pkeep[s * NP + jp] = t;
                                } else if constexpr (MODE == SOLVE_NEWTON15) {
                                    sv = t * t;
                                    dv = t;
                                    if constexpr (KEEP) pkeep[s * NP + jp] = sv;
                                } else {
                                    f32x2 u;   // t^(r-1); log2(0) = -inf -> exp2(-inf) = 0 (r > 1)
                                    u[0] = __builtin_amdgcn_exp2f(rm1 * __builtin_amdgcn_logf(t[0]));
                                    u[1] = __builtin_amdgcn_exp2f(rm1 * __builtin_amdgcn_logf(t[1]));
                                    sv = u * t;
                                    dv = u;
                                    pkeep[s * NP + jp] = sv;
                                    if constexpr (LIN_GEN) ukeep[s * NP + jp] = u;
                                }
                                S2 = jp == 0 ? sv : S2 + sv;
                                D2 = jp == 0 ? dv : D2 + dv;
                            }
                            // horizontal adds as plain v_add_f32: written as C++ the vectoriser packs {S2.x + S2.y, D2.x + D2.y}
                            // into one v_pk_add_f32 behind six register shuffles
                            // (NP == 1: the operands would be raw v_exp_f32 results, and inline asm is invisible to the
                            // compiler's trans-use hazard handling: seen as NaNs / run-to-run differences at nfield <= 8)
                            if constexpr (NP > 1) red_write(red, s & 1, lane, vadd(S2[0], S2[1]), vadd(D2[0], D2[1]));
                            else red_write(red, s & 1, lane, S2[0] + S2[1], D2[0] + D2[1]);
                        }
                        wave_lds_fence();
                        unsigned long long any_live = 0;
#pragma unroll
                        for (int s = 0; s < SPW; ++s) {
                            if (kSampleExit && SPW > 1 && !live[s]) continue;
                            const Red2 r = red_read(red, s & 1, c);
                            const f32x2 sd = (r.g0 + r.g1) + (r.g2 + r.g3);     // {S, Dv} in one register pair
                            float Dv = sd[1];
                            if constexpr (MODE == SOLVE_NEWTON15) Dv *= 2.0f;
                            if constexpr (MODE == SOLVE_NEWTON) Dv *= rr;
                            Ssum[s] = sd[0];
                            const float f = sd[0] - 1.0f;
                            const float step = f * __builtin_amdgcn_rcpf(Dv);             // Newton self-corrects: 1-ulp rcp
                            const float tn = tau[s] + step;
                            // A row is done when the residual OR — generic alpha — the Newton step f / D is small.  The step
                            // bounds the error of every p_i (|dp_i| <= r t_i^(r-1) |dtau| <= r |dtau|: 3e-7 at most); a DENSE
                            // row's residual, on the other hand, carries the rounding noise of its many terms (39 x
                            // exp2(r log2 t): ~2.5e-6) while its slope is large, and the residual-only test sent such waves
                            // into a second evaluation that moved tau by 1e-7 (bench.py's random-init weights at alpha = 1.7:
                            // two evaluations per pass, 117 us; now one, 100 us; parity margin unchanged at 0.45 of the bar).
                            // alpha = 2 / 1.5 have no transcendental noise, take one evaluation on dense rows anyway, and lose
                            // parity margin to the looser test (0.42 -> 0.54 / 0.22 -> 0.29): residual only.
                            float thr = kNewtonTol;
                            if constexpr (MODE == SOLVE_NEWTON) thr = __builtin_fmaxf(thr, tau_tol * Dv);
                            const bool c_f = f > thr, c_t = tn > tau[s];
                            const bool act = c_f && c_t && !dbg_no_solve;
                            tau[s] = act ? tn : tau[s];
                            // two compare masks and a scalar AND (the ballot of the combined bool costs two more VALU ops)
                            live[s] = dbg_no_solve ? 0ull : (__builtin_amdgcn_ballot_w64(c_f) & __builtin_amdgcn_ballot_w64(c_t));
                            if constexpr (LIN) {
                                const bool c_l = step < lin_tol;                       // the step was taken; what it leads to is known
                                dlin[s] = (act && c_l) ? step : 0.f;
                                live[s] &= ~__builtin_amdgcn_ballot_w64(c_l);
                            }
                            any_live |= live[s];
                        }
                        if (!any_live) break;
                    }
                    if constexpr (LIN) {
#pragma unroll
                        for (int s = 0; s < SPW; ++s) {
                            if (__builtin_amdgcn_ballot_w64(dlin[s] != 0.f)) {          // wave-uniform: some row ended on a pending step
                                if constexpr (LIN_GEN) {
                                    const float k = -rr * dlin[s];                      // rows without a pending step: 0
                                    const f32x2 k2 = {k, k};
#pragma unroll
                                    for (int jp = 0; jp < NP; ++jp) {
                                        f32x2 pc;                                       // clamp: p in [0, 1]; the tangent of a vanishing element goes below 0
                                        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(pc) : "v"(ukeep[s * NP + jp]), "v"(k2), "v"(pkeep[s * NP + jp]));
                                        pkeep[s * NP + jp] = pc;
                                    }
                                } else if constexpr (KEEP) {                            // alpha = 1.5: p = (t - step)^2, exactly
                                    tau2[s][0] = tau[s];
#pragma unroll
                                    for (int jp = 0; jp < NP; ++jp) {
                                        const f32x2 t = pk_sub_clamp01_lo(XP_GET(s, jp), tau2[s]);
                                        pkeep[s * NP + jp] = t * t;
                                    }
                                }                                                       // (not KEEP: the weight pass recomputes p from tau)
                                Ssum[s] = dlin[s] != 0.f ? 1.0f : Ssum[s];              // the Newton model's sum at the new threshold
                            }
                        }
                    }
                    PHASE(3);
                    // unnormalised weights p * values (armnet_1h.py:34)
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        const f32x2 tk = {tau[s], tau[s]};
#pragma unroll
                        for (int jp = 0; jp < NP; ++jp) {
                            f32x2 p;
                            if constexpr (KEEP) {
                                p = pkeep[s * NP + jp];
                            } else {
                                const f32x2 t = pk_sub_clamp01(XP_GET(s, jp), tk);
                                if constexpr (MODE == SOLVE_MICHELOT) p = t;
                                else p = t * t;
                            }
                            XP_SET(s, jp, p * VV(jp));
                        }
                    }
                }
                // normaliser (entmax.py:63-64) folded into the exponent scale: 1/S by rcp + one Newton step
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    const float S = Ssum[s] + (tau[s] - tau[s]);        // NaN threshold -> NaN row
                    float r = __builtin_amdgcn_rcpf(S);
                    r = fmaf(fmaf(-S, r, 1.0f), r, r);
                    kexp[s] = MODEL == MODEL_GC_ARM ? r : L2E * r;      // GC-ARM: no outer exp (gc_arm.py:92-94)
                    if constexpr (F16) kexp[s] *= dsc_z;               // the accumulators of MFMA #2 carry 2^(kx + kw)
                }
            }

            // ---- MFMA #2: Z^T[e, o] = sum_f X[f, e] * W[o, f]; sample chains interleaved -------------
            const f32x2 bn = *reinterpret_cast<const f32x2*>(p_bn + (nt * 16 + c) * 2);
            const float xb = MODEL == MODEL_AFN ? p_x[nt * 16 + c] : 0.f;
            f32x4 c2[SPW][EB];
            if constexpr (F16) {
                // k-tile J = quarter-steps 4J .. 4J+3: the lane's four accumulator elements of MFMA #1's tile J ARE its four
                // contraction elements (field 4(4J+i) + g), so the weights go from the C layout straight into the B operand
                f16x4 wh[NTILE], wl[NTILE];
#pragma unroll
                for (int J = 0; J < NTILE; ++J) {
                    f32x4 w = c1[J];
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (4 * J + i >= NQ) w[i] = 0.f;                 // quarter-steps past the sample (compile-time)
                    const HiLo4 h = split4(w);
                    wh[J] = h.hi;
                    wl[J] = h.lo;
                }
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 pa[NTILE], pb[NTILE];
#pragma unroll
                for (int J = 0; J < NTILE; ++J)             // (x_hi | x_lo) . (w_hi | w_hi)
                    pa[J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2f[J], cat8(wh[J], wh[J]), zero, 0, 0, 0);
#pragma unroll
                for (int J = 0; J < NTILE; ++J) {           // x_hi . w_lo
                    const f16x4 xh = {a2f[J][0], a2f[J][1], a2f[J][2], a2f[J][3]};
                    pb[J] = __builtin_amdgcn_mfma_f32_16x16x16f16(xh, wl[J], zero, 0, 0, 0);
                }
                f32x4 acc = pa[0] + pb[0];
#pragma unroll
                for (int J = 1; J < NTILE; ++J) acc += pa[J] + pb[J];
                c2[0][0] = acc;
            }
#pragma unroll
            for (int j = 0; j < (F16 ? 0 : NQ); ++j) {
                f32x2 es = {0.f, 0.f};
                if constexpr (MODEL == MODEL_GC_ARM) es = *reinterpret_cast<const f32x2*>(p_x + (4 * j + g) * 2);
#pragma unroll
                for (int eb = 0; eb < EB; ++eb)
#pragma unroll
                    for (int s = 0; s < SPW; ++s) {
                        const int q = s * NQ + j;
                        const int row = 16 * (q >> 2) + 4 * g + (q & 3);
                        float a2;
                        if constexpr (HOIST) a2 = a2h[(j * EB + eb) * SPW + s];
                        else a2 = xt[row * ES + 16 * eb + c];
                        // gc_arm.py:89: the interaction runs on emb_bn(exp(x)), field 4j+g of this lane group
                        if constexpr (MODEL == MODEL_GC_ARM) a2 = fmaf(__builtin_amdgcn_exp2f(a2 * L2E), es[0], es[1]);
                        if (dbg_no_mfma) {
                            const float w = a2 * XG(s, j);
                            c2[s][eb] = j == 0 ? f32x4{w, w, w, w} : c2[s][eb] + w;
                            continue;
                        }
                        if (j == 0)
                            c2[s][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, XG(s, j), f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            c2[s][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, XG(s, j), c2[s][eb], 0, 0, 0);
                    }
            }
            PHASE(4);
            // LATE: the next group's loads are issued HERE — behind the first pass's arithmetic, in front of its stores — and
            // not right after staging: the address computation needs the next group's ids, hipcc cannot count the
            // (conditional) stores that were issued after those ids and guards their first use with vmcnt(0), i.e. with a
            // drain of the PREVIOUS group's output stores.  Right after staging that drain was 12 % of a wave's time; one
            // pass later the stores have long landed (random-init / trained-like weights -0.4 / -0.9 %, alpha = 1.7 -1.1 %).
            // Only where the longer live ranges cost no scratch (two-sample groups, nemb 32 and 41+ fields spill with it).
            if constexpr (LATE) {
                if (nt == 0) prefetch();
            }
            // ---- epilogue: exp(z / S) = exp2(z * log2e / S) (rel. error <= ~|z| * 1.3e-7), BN affine, store
            const bool fast_store = full_rows && 16 * nt + 16 <= O;      // wave-uniform: whole 16-byte stores
            const bool o_ok = 16 * nt + c < O;
            if (!dbg_no_store) {
                const f32x2 bn0 = {bn[0], bn[0]}, bn1 = {bn[1], bn[1]};
#pragma unroll
                for (int s = 0; s < SPW; ++s) {
                    if (SPW == 1 || b0 + s < Bi) {      // (a one-sample group inside the loop is always a real sample)
                        // wave-uniform part of the address on the scalar unit, the lane's part (c, g) a constant offset:
                        // mixed in one expression the 64-bit multiply ran on the VALU (two quarter-rate v_mul_lo_u32 and
                        // a v_mad_u64_u32 per sample and pass)
                        float* dst = a.out + ((size_t)(b0 + s) * O_out + 16 * nt) * (size_t)Er + lane_out_off;
                        const f32x2 ke = {kexp[s], kexp[s]};
#pragma unroll
                        for (int eb = 0; eb < EB; ++eb) {
                            f32x2 zlo = f32x2{c2[s][eb][0], c2[s][eb][1]} * ke;
                            f32x2 zhi = f32x2{c2[s][eb][2], c2[s][eb][3]} * ke;
                            if constexpr (MODEL == MODEL_AFN) {                 // + (bias + sum_f W t_f) * log2(e)
                                zlo += f32x2{xb, xb};
                                zhi += f32x2{xb, xb};
                            }
                            f32x2 elo = zlo, ehi = zhi;
                            if constexpr (MODEL != MODEL_GC_ARM) {
                                elo = f32x2{__builtin_amdgcn_exp2f(zlo[0]), __builtin_amdgcn_exp2f(zlo[1])};
                                ehi = f32x2{__builtin_amdgcn_exp2f(zhi[0]), __builtin_amdgcn_exp2f(zhi[1])};
                            }
                            const f32x2 vlo = __builtin_elementwise_fma(elo, bn0, bn1);
                            const f32x2 vhi = __builtin_elementwise_fma(ehi, bn0, bn1);
                            const int e = 16 * eb + 4 * g;
                            if (fast_store) {
                                ARMNET_STREAM_STORE(reinterpret_cast<f32x4u*>(dst + 16 * eb), (f32x4{vlo[0], vlo[1], vhi[0], vhi[1]}));
                            } else if (o_ok) {
                                if (e + 4 <= Er) {
                                    *reinterpret_cast<f32x4u*>(dst + 16 * eb) = f32x4{vlo[0], vlo[1], vhi[0], vhi[1]};
                                } else if (e + 2 == Er) {
                                    *reinterpret_cast<f32x2u*>(dst + 16 * eb) = vlo;
                                } else {
                                    if (e + 0 < Er) dst[16 * eb + 0] = vlo[0];
                                    if (e + 1 < Er) dst[16 * eb + 1] = vlo[1];
                                    if (e + 2 < Er) dst[16 * eb + 2] = vhi[0];
                                }
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int s = 0; s < SPW; ++s)
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb) asm volatile("" ::"v"(c2[s][eb]));
            }
#undef XG
#undef XP_GET
#undef XP_SET
#undef VV
            PHASE(5);
        }
    }
#ifdef ARMNET_PHASE_TIMING
    if (lane == 0 && a.id_status)
        for (int i = 0; i < 12; ++i) atomicAdd(reinterpret_cast<unsigned int*>(a.id_status) + i, (unsigned int)(ph_acc[i] >> 4));
#endif
}

#ifndef ARMNET_WPS
#define ARMNET_WPS 4
#endif

// Waves per block for a block whose waves need `wave_bytes` of LDS each and share `param_bytes`: the choice (4, 8, 12, 16;
// at most mfma_max_wpb(wps)) that puts the most waves on a CU — blocks per CU bounded by the LDS and by wps waves per
// SIMD —, the smaller block on a tie.  Returns 0 when not even 4 waves fit; *blocks_per_cu for the persistent grid.
static inline int mfma_pick_wpb(size_t wave_bytes, size_t param_bytes, int wps, int* blocks_per_cu) {
    int best = 0, best_waves = 0;
    for (int w = 4; w <= mfma_max_wpb(wps); w += 4) {
        const size_t lds = (size_t)w * wave_bytes + param_bytes;
        if (lds > 160 * 1024) break;
        int b = (int)(160 * 1024 / lds);
        if (b > wps * 4 / w) b = wps * 4 / w;
        if (b < 1) b = 1;
        if (b * w > best_waves) { best = w; best_waves = b * w; if (blocks_per_cu) *blocks_per_cu = b; }
    }
    return best;
}

// One configuration (samples per wave-group, waves per SIMD) of a shape: sizes the block, launches the persistent grid.
template <int E, int NQ, int SPW, int MODE, int SRC, int WPS, int MODEL, bool HOIST = false, bool F16 = false>
static int launch_cfg(const FusedArgs& a, hipStream_t st) {
    constexpr int NTILE = (SPW * NQ + 3) / 4;
    const int NT = (a.O + 15) / 16;
    const size_t wave_bytes = (size_t)(NTILE * 16 * (E + 4) + 256) * sizeof(float);
    const size_t param_bytes = ((size_t)NT * (E / 16) * 256 + (size_t)NT * (NQ / 2) * 128 + (size_t)NT * 32 +
                                (MODEL == MODEL_GC_ARM ? (size_t)NQ * 8 : MODEL == MODEL_AFN ? (size_t)NT * 16 : 0)) * sizeof(float);
    int per_cu = 0;
    int wpb = mfma_pick_wpb(wave_bytes, param_bytes, WPS, &per_cu);
    if (wpb == 0) return ARMNET_ERR_UNSUPPORTED;
#ifdef ARMNET_DEV_FLAGS
    if (const char* fw = getenv("ARMNET_FORCE_WPB")) {          // developer knob: waves per block
        const int w = atoi(fw);
        if (w >= 4 && w <= mfma_max_wpb(WPS) && (size_t)w * wave_bytes + param_bytes <= 160 * 1024) {
            wpb = w;
            per_cu = (int)(160 * 1024 / ((size_t)w * wave_bytes + param_bytes));
            if (per_cu > WPS * 4 / w) per_cu = WPS * 4 / w;
            if (per_cu < 1) per_cu = 1;
        }
    }
#endif
    size_t lds = (size_t)wpb * wave_bytes + param_bytes;
#ifdef ARMNET_DEV_FLAGS
    if (const char* lp = getenv("ARMNET_LDS_PAD")) {           // developer knob: unused LDS, to lower the occupancy
        lds += (size_t)atoi(lp);
        per_cu = (int)(160 * 1024 / lds);
        if (per_cu > WPS * 4 / wpb) per_cu = WPS * 4 / wpb;
        if (per_cu < 1) per_cu = 1;
    }
#endif
    const int64_t ngroups = (a.B + SPW - 1) / SPW;
    const int64_t blocks = (ngroups + wpb - 1) / wpb;
    const int64_t resident = (int64_t)device_cu_count() * per_cu;   // blocks the chip holds at once
    // persistent grid-stride waves: the software pipeline's prologue is paid once per wave
    int64_t want = blocks < resident ? blocks : resident;
#ifdef ARMNET_DEV_FLAGS
    if (const char* gm = getenv("ARMNET_GRID_MULT")) {          // developer knob: oversubscribe the grid
        want = (int64_t)(resident * atof(gm));
        if (want > blocks) want = blocks;
        if (want < 1) want = 1;
    }
#endif
    auto kern = fused_mfma_kernel<E, NQ, SPW, MODE, SRC, WPS, MODEL, HOIST, F16>;
    ARMNET_ALLOW_BIG_LDS(kern, lds);
    kern<<<(int)want, 64 * wpb, lds, st>>>(a);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// SPW: two samples per wave-group when NQ % 4 != 0 (their 2*NQ quarter-steps fill whole tiles), one otherwise;
// nemb = 64 always one sample (LDS / register budget; a half-pad last tile when NQ % 4 != 0)
template <int E, int NQ, int MODE, int SRC, int MODEL = MODEL_ARM>
static int launch_one(const FusedArgs& a, hipStream_t st) {
#ifdef ARMNET_FORCE_SPW1                    // developer probe: one sample per wave-group everywhere (half-pad last tile)
    constexpr int SPW = 1;
#else
    constexpr int SPW = (E >= 64 || NQ % 4 == 0) ? 1 : 2;
#endif
    // waves/SIMD the register allocator targets: 4 (128 VGPRs) where the working set fits without scratch
    // traffic in the solver loop, fewer for the wide shapes (nemb=64 is LDS-limited to 2 blocks/CU anyway;
    // generic-alpha Newton keeps two transcendental temporaries per pair alive)
    // nemb 65..128 (round 4): 8 accumulator tiles of MFMA #2 and up to 24 staging registers x 4 per wave
    constexpr int WPS = (E >= 128) ? (NQ >= 6 ? 2 : 3)
                        : (E >= 64) ? (NQ >= 10 ? 2 : 3)
                        : (E >= 32 || MODE == SOLVE_NEWTON || MODE == SOLVE_BISECT) ? 3      // measured: nemb=32 is faster at 3
                        : (MODE == SOLVE_SOFTMAX && SPW * NQ >= 20) ? 3
                        // alpha = 2, many fields: 3 waves/SIMD run as fast as 4 (measured with padded LDS: 92.7 vs 93.6 us)
                        // and the 168-register budget holds the last evaluation's clamped differences, so the weight pass
                        // does not recompute them: 88.1 -> 85.5 us.  Few fields (nfield = 10, 256 neurons): 4 is 7 % faster.
                        : ((MODE == SOLVE_MICHELOT || MODE == SOLVE_NEWTON15) && SPW * NQ >= 16) ? 3
                        : ARMNET_WPS;
#ifndef ARMNET_NO_OCC_TABLE
    // The ARM block at nemb <= 16 (every data set of the reference's run.sh: nemb = 10): the high-occupancy table above.
    // Measured against the 3-4 wave configurations (kbench, 1 x MI355X, 32 neurons, random-init / trained-like weights at
    // alpha = 2 / trained-like at alpha = 1.7): 3 fields -13 / -8 / -16 %, 10 fields -12 / -12 / -16 %, 22 fields -2 / -3 / -3 %,
    // 30 fields -3 / -8 / -8 %, 43 fields -3 / -5 / -6 %; 128 neurons: -7..-15 %, -10..-15 %, 0..-3 %, -3..-8 %, -2..-3 %.
    // Criteo-shaped blocks (33-40 fields; bench.py): random-init weights 87.4-87.6 us against 87.5-88.0, trained-like weights
    // 116.8-117.0 against 120.4-120.5, alpha = 1.7 96.4-96.8 / 178.9-179.5 against 98.1-98.2 / 184.2-184.5 — the solver loop
    // is wave-uniform and 16 rows need fewer evaluations than 32, twenty small waves per CU overlap better than twelve large
    // ones, and that pays for the half-padded third tile of MFMA #1; at 64 / 128 neurons +1.3 % on random-init weights and
    // -0.4 / -2.3 % on trained-like ones: those stay on two-sample groups.
    if constexpr (MODEL == MODEL_ARM && occ_config(E, NQ, MODE).spw != 0) {
        constexpr OccCfg oc = occ_config(E, NQ, MODE);
        if constexpr (NQ != 10) {
            return launch_cfg<E, NQ, oc.spw, MODE, SRC, oc.wps, MODEL>(a, st);
        } else {
            if (a.O <= 32) return launch_cfg<E, NQ, oc.spw, MODE, SRC, oc.wps, MODEL>(a, st);
#ifdef ARMNET_HOIST_WIDE                        // A/B switch of the round-6 experiment (see HOIST above)
            if (a.O >= 64) return launch_cfg<E, NQ, 1, MODE, SRC, 4, MODEL, true>(a, st);
#endif
            return launch_cfg<E, NQ, SPW, MODE, SRC, WPS, MODEL>(a, st);
        }
    } else {
        return launch_cfg<E, NQ, SPW, MODE, SRC, WPS, MODEL>(a, st);
    }
#else
    return launch_cfg<E, NQ, SPW, MODE, SRC, WPS, MODEL>(a, st);
#endif
}

template <int E, int NQ, int SRC, int MODEL = MODEL_ARM>
static int launch_mode(const FusedArgs& a, hipStream_t st) {
    switch (a.cfg.mode) {
        case SOLVE_SOFTMAX: return launch_one<E, NQ, SOLVE_SOFTMAX, SRC, MODEL>(a, st);
        case SOLVE_MICHELOT: return launch_one<E, NQ, SOLVE_MICHELOT, SRC, MODEL>(a, st);
        case SOLVE_NEWTON15: return launch_one<E, NQ, SOLVE_NEWTON15, SRC, MODEL>(a, st);
        case SOLVE_NEWTON: return launch_one<E, NQ, SOLVE_NEWTON, SRC, MODEL>(a, st);
        case SOLVE_BISECT: return launch_one<E, NQ, SOLVE_BISECT, SRC, MODEL>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

// sibling models (ids only): GC-ARM with every sparse map, AFN has none (one instantiation per shape)
template <int E, int NQ, int MODEL>
static int launch_sibling(const FusedArgs& a, hipStream_t st) {
    if (a.rows != nullptr) return ARMNET_ERR_UNSUPPORTED;
    if constexpr (MODEL == MODEL_AFN) {
        return a.id_type == ARMNET_ID_I64 ? launch_one<E, NQ, SOLVE_SOFTMAX, 0, MODEL>(a, st)
                                          : launch_one<E, NQ, SOLVE_SOFTMAX, 1, MODEL>(a, st);
    } else {
        return a.id_type == ARMNET_ID_I64 ? launch_mode<E, NQ, 0, MODEL>(a, st) : launch_mode<E, NQ, 1, MODEL>(a, st);
    }
}

// ids (int64 / int32) for every shape; the pre-gathered-rows source only where WITH_ROWS
template <int E, int NQ, bool WITH_ROWS>
static int launch_src(const FusedArgs& a, hipStream_t st) {
    if (a.rows != nullptr) {
        if constexpr (WITH_ROWS) return launch_mode<E, NQ, 2>(a, st);
        else return ARMNET_ERR_UNSUPPORTED;
    }
    return a.id_type == ARMNET_ID_I64 ? launch_mode<E, NQ, 0>(a, st) : launch_mode<E, NQ, 1>(a, st);
}

// F16 (round 6): the ARM block at nemb <= 16 with both contractions as fp16 x 2 splits — one-sample groups, four waves per SIMD
// (116-128 registers, no scratch but 12 bytes in the generic-alpha solver); every sparse map but the literal bisection
#ifndef ARMNET_F16_MIN_O
#define ARMNET_F16_MIN_O 64        // neurons of a launch from which the split form wins at 29-32 fields (fused_mfma.hip: 33 for 33+ fields, 256 for 17-28)
#endif
#ifndef ARMNET_F16_WPS
#define ARMNET_F16_WPS 4
#endif
template <int NQ, int SRC>
static int launch_f16_mode(const FusedArgs& a, hipStream_t st) {
    switch (a.cfg.mode) {
        case SOLVE_SOFTMAX: return launch_cfg<16, NQ, 1, SOLVE_SOFTMAX, SRC, ARMNET_F16_WPS, MODEL_ARM, false, true>(a, st);
        case SOLVE_MICHELOT: return launch_cfg<16, NQ, 1, SOLVE_MICHELOT, SRC, ARMNET_F16_WPS, MODEL_ARM, false, true>(a, st);
        case SOLVE_NEWTON15: return launch_cfg<16, NQ, 1, SOLVE_NEWTON15, SRC, ARMNET_F16_WPS, MODEL_ARM, false, true>(a, st);
        case SOLVE_NEWTON: return launch_cfg<16, NQ, 1, SOLVE_NEWTON, SRC, ARMNET_F16_WPS, MODEL_ARM, false, true>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}
template <int NQ>
static int launch_f16(const FusedArgs& a, hipStream_t st) {
    if (a.rows != nullptr) return launch_f16_mode<NQ, 2>(a, st);
    return a.id_type == ARMNET_ID_I64 ? launch_f16_mode<NQ, 0>(a, st) : launch_f16_mode<NQ, 1>(a, st);
}

// one translation unit per family keeps the build parallel
int launch_mfma_e16(const FusedArgs& a, int nq, hipStream_t st);
int launch_mfma_e16_f16(const FusedArgs& a, int nq, hipStream_t st);   // the F16 form (fused_mfma_e16h.hip); UNSUPPORTED: use the above
int launch_mfma_e32(const FusedArgs& a, int nq, hipStream_t st);
int launch_mfma_e64(const FusedArgs& a, int nq, hipStream_t st);
int launch_mfma_e128a(const FusedArgs& a, int nq, hipStream_t st);    // nq 2..6
int launch_mfma_e128b(const FusedArgs& a, int nq, hipStream_t st);    // nq 8..12
int launch_gc_e16(const FusedArgs& a, int nq, hipStream_t st);
int launch_gc_e32(const FusedArgs& a, int nq, hipStream_t st);
int launch_gc_e64(const FusedArgs& a, int nq, hipStream_t st);
int launch_gc_e128(const FusedArgs& a, int nq, hipStream_t st);      // nemb 65..128 (round 6)
int launch_afn(const FusedArgs& a, int ep, int nq, hipStream_t st);

}  // namespace armnet
