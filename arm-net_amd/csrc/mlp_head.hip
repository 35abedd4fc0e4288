// mlp_head.hip — the eval-mode prediction head (models/layers.py:68-88: (Linear, BatchNorm1d, ReLU, Dropout) x n, then
// Linear(., 1)) as ONE kernel on the CDNA4 16-bit matrix cores with fp32-equivalent numerics.  gfx950 only.
//
// Why not fp32 MFMA: v_mfma_f32_*_f32 runs at the fp32 VECTOR rate (157 TFLOP/s); hipBLASLt already sits there
// (2 x 110 us for the two GEMMs of the 2x256 head at B = 65 536, more than the fused ARM block in front of them).
// The 16-bit v_mfma_f32_32x32x16_{f16,bf16} are 16x faster per product.  Two operand splits, one kernel body:
//
//  * fp16 x 2 (round 6, the default): 3 products.  x' = c x, w' = 2^s_n w (exact power-of-two scales; s_n per weight
//      row puts the row's largest weight in [2^13, 2^14), c = 16 for the first layer, a per-SAMPLE power of two that puts
//      the sample's largest hidden activation in [2^14, 2^15) for the second);  x' = xh + xl, w' = wh + wl with
//      round-to-nearest fp16 parts: |x' - xh - xl| <= 2^-24 |x'| while xl is a normal fp16 number, <= 2^-25 absolute
//      (in x' units) once it is subnormal.
//          x' w' ~= xl*wh + xh*wl + xh*wh                                  (dropped: xl*wl, relative size 2^-24)
//      Every product of two fp16 numbers is exact in fp32; the matrix core accumulates in fp32; the accumulator is
//      rescaled by 2^-s_n / c (exact) in the bias epilogue.  HALF the matrix-core cycles of the bf16 split and 4 instead of
//      6 bytes of packed planes per weight through the LDS ring, 3 instead of 5.5 vector-ALU operations per activation.
//      Range: fp16 ends at 65 504.  The reference has no clamp behind exp (armnet_1h.py:86), so a first-layer input may
//      be anything: every wave tracks max |c x| of what it splits, the block votes once after layer 1 and a block that
//      met |c x| > 65 000 (|x| > 4 062) or inf — or a wave of it nothing but |x| < 1e-3 (every low part subnormal) — REDOES its
//      samples with the bf16 split below — same kernel, same launch,
//      no host involvement, nothing stored before the vote.  (A sample's last bits therefore depend on whether a
//      neighbour of its 128/256-sample block overflowed.)  Second-layer inputs cannot overflow: they are scaled per sample.
//  * bf16 x 3 (rounds 2-5; the fallback, `flags & ARMNET_MLP_F_BF16X3`, and armnet_linear_bf16x3_f32): 6 products.
//      x = xh + xm + xl exactly (three 8-bit slices of the 24-bit significand, by truncation), w likewise (round to
//      nearest);  x*w ~= xh*wh + xh*wm + xm*wh + xm*wm + xh*wl + xl*wh   (dropped: 2 terms of relative size 2^-24).
//      bf16 has fp32's exponent range: no scaling, no range test.
//
// Formulation: transposed, C'[n, m] = sum_k W[n, k] * X[m, k] — weights are the A operand (rows = hidden units),
// activations the B operand (columns = samples).  The C layout of v_mfma_f32_32x32x16 gives lane (m = l & 31,
// half = l >> 5) the hidden units n = 32 t + (r & 3) + 8 (r >> 2) + 4 half of sample m; the B operand of the NEXT
// layer wants, per lane (m, half), 8 values of the contraction index per k-step.  The contraction order is ours to
// choose, so k-step 2t+u of the next layer takes registers 8u..8u+7 of tile t and the packed weights are permuted to
// match: hidden activations never leave the registers between layers, no transpose, no LDS round trip.
//
// One wave owns 32 samples through all layers; a 512-thread block = 8 waves = 256 samples, one block per CU.  The packed
// weights are one linear stream of 1-KiB lane-ready blocks in consumption order; the block's waves pull it stage by stage
// (one k-step of all its tiles) through an LDS ring with global_load_lds_dwordx4, one barrier per stage; each wave's
// activations come through its own LDS ring the same way.  A request is issued D stages ahead of its use (rings of
// D + 1 slots); the counted vmcnt waits follow from the issue order (vm_allowed below).
#include <stdlib.h>

#include "armnet_common.h"

namespace armnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2m __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// operand split = bit planes per weight
constexpr int MLP_F16X2 = 2, MLP_BF16X3 = 3;
constexpr float kXScale = 16.0f;        // first-layer activations are multiplied by this before the fp16 split
constexpr float kXLimit = 65000.0f;     // a block that splits a scaled activation above this redoes its samples in bf16x3
constexpr float kXTiny = 0.015625f;     // ... and so does a block whose LARGEST scaled activation is below 2^-6 (|x| < 1e-3 everywhere:
                                        // the low fp16 part of every element would be subnormal — fp16's fixed 2^-24 spacing
                                        // instead of a relative 2^-11 —; the bf16 split has fp32's exponent range)
#ifndef ARMNET_MLP_DEPTH_F16
#define ARMNET_MLP_DEPTH_F16 2
#endif
// prefetch distance in stages (rings of depth + 1 slots)
__host__ __device__ constexpr int mlp_depth(int P) { return P == MLP_F16X2 ? ARMNET_MLP_DEPTH_F16 : 2; }

// layer >= 2 runs in groups of TG output tiles (the previous layer's 16*NT accumulator registers stay live as its B
// operands, so only TG*16 more can be spent on accumulators); a stage of its weight stream covers KPS k-steps
__host__ __device__ constexpr int mlp_tg(int NT) { return NT >= 8 ? 2 : (NT < 4 ? NT : 4); }
__host__ __device__ constexpr int mlp_kps(int NT) { return NT >= 8 ? 4 : 1; }

// The packed-parameter blob of one launch:
//   [fp16 stream: layer 1 (KS1 stages of NT tiles x 2 planes x 1 KiB) | layer 2 (NG * NS2 stages of KPS*TG tiles x 2 planes)]
//   [bf16 stream: the same with 3 planes]
//   [fp32 tables: bias1 | bias2 | wlast | rs1 | rs2 (NT*32 floats each, C-layout order) | blast (4 floats)]
//   [int32 row exponents s_n of layer 1 | layer 2 (NT*32 each, natural order; pack-time only)]
struct MlpLayout {
    int NT, TG, NG, KPS, KS1, NS2;    // NS2: stages per group of layer 2
    int64_t l1_off[4], l2_off[4];     // [planes]: start of the layer-1 / layer-2 stream of the split with that many planes
    int64_t tab_off;                  // start of the fp32 tables
    int64_t exp_off;                  // start of the row exponents
    int64_t total;
};

__host__ __device__ inline MlpLayout mlp_layout(int K0, int NT, int n_hidden) {
    MlpLayout L;
    L.NT = NT;
    L.TG = mlp_tg(NT);
    L.NG = NT / L.TG;
    L.KPS = mlp_kps(NT);
    L.NS2 = 2 * NT / L.KPS;
    L.KS1 = (K0 + 15) / 16;
    int64_t off = 0;
    for (int P = MLP_F16X2; P <= MLP_BF16X3; ++P) {
        L.l1_off[P] = off;
        off += (int64_t)L.KS1 * NT * P * 1024;
        L.l2_off[P] = off;
        if (n_hidden >= 2) off += (int64_t)L.NG * L.NS2 * L.KPS * L.TG * P * 1024;
    }
    L.tab_off = off;
    L.exp_off = off + ((int64_t)5 * NT * 32 + 4) * sizeof(float);
    L.total = L.exp_off + (int64_t)2 * NT * 32 * sizeof(int32_t);
    return L;
}

static inline int mlp_nt_for(int nhid) { return nhid <= 32 ? 1 : nhid <= 64 ? 2 : nhid <= 128 ? 4 : nhid <= 256 ? 8 : 0; }

// hidden unit held in accumulator register r of tile t by lane half `half` (C layout of the 32x32 MFMA)
__host__ __device__ inline int c_layout_unit(int t, int r, int half) { return 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half; }

// ---------------------------------------------------------------------------------------------------------------
// parameter-only precompute
__device__ inline uint32_t bf16_rn_bits(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

struct PackArgs {
    int K0, NT, n_hidden, slot, N, Kin;
    const float *W, *b, *bn_w, *bn_b, *bn_m, *bn_v;
    float eps;
    uint8_t* packed;
};

__device__ inline float pack_bn_scale(const PackArgs& p, int n) { return p.bn_w ? p.bn_w[n] / sqrtf(p.bn_v[n] + p.eps) : 1.0f; }

// s_n of the fp16 split: the BatchNorm-folded row n times 2^s_n has its largest magnitude in [2^13, 2^14).  One wave per row.
__global__ void mlp_rowexp_kernel(PackArgs p) {
    const MlpLayout L = mlp_layout(p.K0, p.NT, p.n_hidden);
    int32_t* rowexp = reinterpret_cast<int32_t*>(p.packed + L.exp_off) + p.slot * p.NT * 32;
    const int n = blockIdx.x, lane = threadIdx.x;
    float m = 0.f;
    if (n < p.N) {
        const float s = pack_bn_scale(p, n);
        for (int k = lane; k < p.Kin; k += 64) {
            const float w = fabsf(p.W[(size_t)n * p.Kin + k] * s);
            if (w < INFINITY) m = fmaxf(m, w);            // a non-finite weight poisons its products anyway
        }
    }
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) {
        int e = 0;
        if (m > 0.f) { (void)frexpf(m, &e); e = 14 - e; }
        rowexp[n] = e < -100 ? -100 : (e > 100 ? 100 : e);
    }
}

__global__ void mlp_pack_kernel(PackArgs p) {
    const MlpLayout L = mlp_layout(p.K0, p.NT, p.n_hidden);
    float* tabs = reinterpret_cast<float*>(p.packed + L.tab_off);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p.slot == 2) {                                   // final Linear(N, 1): weights in C-layout order + its bias
        if (gid < p.NT * 32) {
            const int half = (int)gid / (p.NT * 16), t = ((int)gid / 16) % p.NT, r = (int)gid & 15;
            const int n = c_layout_unit(t, r, half);
            tabs[2 * p.NT * 32 + gid] = n < p.N ? p.W[n] : 0.f;
        }
        if (gid == 0) tabs[5 * p.NT * 32] = p.b ? p.b[0] : 0.f;
        return;
    }
    const int32_t* rowexp = reinterpret_cast<const int32_t*>(p.packed + L.exp_off) + p.slot * p.NT * 32;
    // bias table of this hidden layer (BatchNorm folded): b' = b * s + (beta - mean * s); accumulator scale of the fp16 split
    if (gid < p.NT * 32) {
        const int half = (int)gid / (p.NT * 16), t = ((int)gid / 16) % p.NT, r = (int)gid & 15;
        const int n = c_layout_unit(t, r, half);
        float v = 0.f;
        if (n < p.N) {
            const float s = pack_bn_scale(p, n);
            const float bb = p.b ? p.b[n] : 0.f;
            v = p.bn_w ? bb * s + (p.bn_b[n] - p.bn_m[n] * s) : bb;
        }
        tabs[p.slot * p.NT * 32 + gid] = v;
        tabs[(3 + p.slot) * p.NT * 32 + gid] = ldexpf(p.slot == 0 ? 1.0f / kXScale : 1.0f, -rowexp[n]);
    }
    // weight stream: one thread per (stage, tile, lane, j) -> two fp16 planes and three bf16 planes
    // slot 0: KS1 stages of NT tiles; slot 1: NG * 2NT k-steps of TG tiles (KPS consecutive k-steps form a stage)
    const int tiles = p.slot == 0 ? p.NT : L.TG;
    const int64_t nstage = p.slot == 0 ? L.KS1 : (int64_t)L.NG * 2 * p.NT;
    const int64_t total = nstage * tiles * 512;
    if (gid >= total) return;
    const int j = (int)(gid & 7), lane = (int)((gid >> 3) & 63);
    const int64_t blk = gid >> 9;
    const int tt = (int)(blk % tiles);
    const int64_t stage = blk / tiles;
    int n, k;
    if (p.slot == 0) {
        n = 32 * tt + (lane & 31);
        k = (int)stage * 16 + 8 * (lane >> 5) + j;
    } else {
        const int g = (int)(stage / (2 * p.NT)), s2 = (int)(stage % (2 * p.NT));
        n = 32 * (g * L.TG + tt) + (lane & 31);
        k = c_layout_unit(s2 >> 1, 8 * (s2 & 1) + j, lane >> 5);   // which unit of the previous layer sits at (k-step, half, j)
    }
    float w = 0.f;
    if (n < p.N && k < p.Kin) w = p.W[(size_t)n * p.Kin + k] * pack_bn_scale(p, n);
    {   // bf16 x 3
        const uint32_t hb = bf16_rn_bits(w);
        const float r1 = w - __uint_as_float(hb << 16);
        const uint32_t mb = bf16_rn_bits(r1);
        const float r2 = r1 - __uint_as_float(mb << 16);
        const uint32_t lb = bf16_rn_bits(r2);
        uint16_t* dst = reinterpret_cast<uint16_t*>(p.packed + (p.slot == 0 ? L.l1_off[MLP_BF16X3] : L.l2_off[MLP_BF16X3])) +
                        ((size_t)stage * tiles + tt) * 3 * 512 + lane * 8 + j;
        dst[0] = (uint16_t)hb;
        dst[512] = (uint16_t)mb;
        dst[1024] = (uint16_t)lb;
    }
    {   // fp16 x 2 of the row-scaled weight
        const float ws = ldexpf(w, rowexp[n]);
        const _Float16 h = (_Float16)ws;
        const _Float16 l = (_Float16)(ws - (float)h);
        _Float16* dst = reinterpret_cast<_Float16*>(p.packed + (p.slot == 0 ? L.l1_off[MLP_F16X2] : L.l2_off[MLP_F16X2])) +
                        ((size_t)stage * tiles + tt) * 2 * 512 + lane * 8 + j;
        dst[0] = h;
        dst[512] = l;
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct MlpArgs {
    int64_t B;
    int K0, n_hidden, has_final, N;
    int64_t ldx, ldo;      // row strides (floats) of x and of the hidden-activation output
    const float* x;
    const uint8_t* packed;
    float* out;
    int bf16x3;   // 1: the bf16 split from the start (ARMNET_MLP_F_BF16X3, armnet_linear_bf16x3_f32)
    int linear;   // 1: a plain Linear — hidden activations are written WITHOUT the ReLU (armnet_linear_bf16x3_f32: the
                  // training head's GEMMs, where BatchNorm needs the batch's pre-activation values)
};

// B operand planes of one k-step: [0] = high part ... [P-1] = low part
template <int P>
struct BPlanes { u32x4 p[P]; };

// 8 fp32 -> three packed bf16x8 planes; h + m + l == x exactly (truncating 8-bit slices of the significand)
__device__ __forceinline__ void split_planes(const float (&x)[8], float, BPlanes<3>& b, float&) {
    uint32_t hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hb[i] = __float_as_uint(x[i]) & 0xffff0000u;
        const float r1 = x[i] - __uint_as_float(hb[i]);
        mb[i] = __float_as_uint(r1) & 0xffff0000u;
        lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));       // <= 8 significant bits: its top half is exact
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // dword i = {element 2i (low half), element 2i+1 (high half)}: bytes {hi[3], hi[2], lo[3], lo[2]}
        b.p[0][i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
        b.p[1][i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
        b.p[2][i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
    }
}

// 8 fp32 -> two packed fp16x8 planes of scale * x (round to nearest: v_cvt_pk_f16_f32); mx: running max |scale * x|
__device__ __forceinline__ void split_planes(const float (&x)[8], float scale, BPlanes<2>& b, float& mx) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2m v = {x[2 * i] * scale, x[2 * i + 1] * scale};
        mx = fmaxf(mx, fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2m r = v - __builtin_convertvector(h, f32x2m);
        const f16x2 l = __builtin_convertvector(r, f16x2);
        b.p[0][i] = __builtin_bit_cast(uint32_t, h);
        b.p[1][i] = __builtin_bit_cast(uint32_t, l);
    }
}

// nn.ReLU keeps a NaN (torch.relu(nan) = nan; layers.py:76); v_max_f32 would drop it
__device__ __forceinline__ float relu_keep_nan(float v) { return v <= 0.f ? 0.f : v; }

template <int P>
__device__ __forceinline__ f32x16 mfma16(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (P == MLP_F16X2)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A operands (P planes, [0] = high part) of a pair of output tiles
template <int P, int PAIR>
struct APlanes { u32x4 p[P][PAIR]; };

// ---- LDS reads by hand ------------------------------------------------------------------------------------------
// While an LDS-DMA (global_load_lds) is in flight hipcc's wait-count pass treats the LGKM counter as unordered ("pending
// flat": the DMA carries an LDS memory operand) and turns EVERY wait for a ds_read result into lgkmcnt(0) — a full drain
// that also waits for the reads issued one instruction earlier (measured: four exposed LDS round trips per k-step,
// a lone wave at 45 % of the matrix-core rate).  LDS returns in order, so the weight planes and activation tiles are
// read with inline-asm ds_read_b128 and waited for with exact counts; each wait names the registers it makes valid
// ("+v"), which orders their consumers behind it.
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int N, int PAIR>
__device__ __forceinline__ void lds_wait(u32x4 (&r)[PAIR]) {     // at most N younger LDS reads still outstanding
    if constexpr (PAIR == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r[0]) : "n"(N) : "memory");
}

// planes are read lowest part first: the order in which unit() consumes them
template <int P, int PAIR>
__device__ __forceinline__ void load_planes(APlanes<P, PAIR>& A, uint32_t blk, int t0) {   // blk: LDS byte address + lane*16
#pragma unroll
    for (int j = P - 1; j >= 0; --j)
#pragma unroll
        for (int u = 0; u < PAIR; ++u) A.p[j][u] = lds_read16(blk + ((t0 + u) * P + j) * 1024);
}

// One (k-step, tile pair) unit: the significant cross products for PAIR tiles, smallest terms first
//      bf16 x 3:   l*bh | m*bm  m*bh | h*bl  h*bm  h*bh          fp16 x 2:   l*bh | h*bl  h*bh
// with ONE set of A registers that rolls over to the next unit: a plane's registers are reloaded from LDS (`nxt`: LDS
// byte address + lane*16 of the next unit's k-step block; t0n its first tile) as soon as its last product has been
// issued.  At every wait exactly the (P-1)*PAIR reads issued after the awaited plane may still be in flight (fewer at
// the end of a stage).  The PAIR accumulators alternate so that consecutive MFMAs never wait on each other.
// sched_barrier(VALU) keeps MFMAs and LDS traffic in this order and lets the VALU work (the split of the next k-step)
// float into the MFMA shadows.
template <int P, int PAIR, bool NEXT>
__device__ __forceinline__ void unit(f32x16* acc, APlanes<P, PAIR>& A, const BPlanes<P>& b, uint32_t nxt, int t0n) {
    constexpr int W = (P - 1) * PAIR;
    auto reload = [&](int j) {
        if (NEXT) {
#pragma unroll
            for (int u = 0; u < PAIR; ++u) A.p[j][u] = lds_read16(nxt + ((t0n + u) * P + j) * 1024);
        }
    };
    auto prod = [&](int ja, int jb) {
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[u] = mfma16<P>(A.p[ja][u], b.p[jb], acc[u]);
    };
    if constexpr (P == MLP_BF16X3) {
        lds_wait<W, PAIR>(A.p[2]);
        prod(2, 0);
        __builtin_amdgcn_sched_barrier(0x2);
        reload(2);
        lds_wait<NEXT ? W : PAIR, PAIR>(A.p[1]);
        __builtin_amdgcn_sched_barrier(0x2);
        prod(1, 1);
        prod(1, 0);
        __builtin_amdgcn_sched_barrier(0x2);
        reload(1);
        lds_wait<NEXT ? W : 0, PAIR>(A.p[0]);
        __builtin_amdgcn_sched_barrier(0x2);
        prod(0, 2);
        prod(0, 1);
        prod(0, 0);
        __builtin_amdgcn_sched_barrier(0x2);
        reload(0);
        __builtin_amdgcn_sched_barrier(0x2);
    } else {
        lds_wait<W, PAIR>(A.p[1]);
        prod(1, 0);
        __builtin_amdgcn_sched_barrier(0x2);
        reload(1);
        lds_wait<NEXT ? W : 0, PAIR>(A.p[0]);
        __builtin_amdgcn_sched_barrier(0x2);
        prod(0, 1);
        prod(0, 0);
        __builtin_amdgcn_sched_barrier(0x2);
        reload(0);
        __builtin_amdgcn_sched_barrier(0x2);
    }
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// at most n (wave-uniform, not a compile-time constant) vector-memory operations outstanding; waiting for more is safe
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n) {
#define ARMNET_VM_CASE(k) case k: wait_vm<k>(); break;
        ARMNET_VM_CASE(1) ARMNET_VM_CASE(2) ARMNET_VM_CASE(3) ARMNET_VM_CASE(4) ARMNET_VM_CASE(5) ARMNET_VM_CASE(6)
        ARMNET_VM_CASE(7) ARMNET_VM_CASE(8) ARMNET_VM_CASE(9) ARMNET_VM_CASE(10) ARMNET_VM_CASE(11) ARMNET_VM_CASE(12)
        ARMNET_VM_CASE(13) ARMNET_VM_CASE(14) ARMNET_VM_CASE(15) ARMNET_VM_CASE(16) ARMNET_VM_CASE(17) ARMNET_VM_CASE(18)
        ARMNET_VM_CASE(19) ARMNET_VM_CASE(20)
#undef ARMNET_VM_CASE
        default: if (n > 20) wait_vm<20>(); else wait_vm<0>(); break;
    }
}
// raw barrier: __syncthreads() would drain every LDS-DMA in flight (it fences with vmcnt(0))
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// LDS of a block: [weight ring: (D+1) slots of NT*P KiB] [fp32 tables | vote word] [kWaves x (D+1) activation tiles of 2 KiB]
template <int NT, int P>
__host__ __device__ constexpr int mlp_tab_bytes() { return ((5 * NT * 32 + 4 + 4) * 4 + 15) & ~15; }
template <int NT, int kWaves, int P>
__host__ __device__ constexpr int mlp_lds_bytes() {
    return (mlp_depth(P) + 1) * NT * P * 1024 + mlp_tab_bytes<NT, P>() + kWaves * (mlp_depth(P) + 1) * 2048;
}

#ifdef ARMNET_DEV_FLAGS
// developer build: per-phase s_memtime sums over all waves (layer-1 stages): wait | barrier | first reads | units
__device__ unsigned long long g_mlp_phase[8];
#define MLP_PHASE(i) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ph[i] += _n - pt; pt = _n; } while (0)
#else
#define MLP_PHASE(i) do {} while (0)
#endif

// The whole head for this block's samples with the P-plane split.  Returns true (block-uniform, P == MLP_F16X2 only) when
// a first-layer activation left the fp16 range: nothing has been stored, every request has landed, the caller runs the
// bf16 body.
template <int NT, int kWaves, int P>
__device__ __forceinline__ bool mlp_body(const MlpArgs& a) {
    constexpr int TG = mlp_tg(NT), NG = NT / TG, KPS = mlp_kps(NT), NS2 = 2 * NT / KPS;
    constexpr int ST1 = NT * P * 1024, ST2 = KPS * TG * P * 1024;
    static_assert(ST2 <= ST1 && KPS * TG == NT, "both layers' stages hold NT*P blocks and fit a ring slot");
    constexpr int PAIR = NT >= 2 ? 2 : 1;
    constexpr int NP1 = NT / PAIR;          // tile pairs per k-step, layer 1
    constexpr int NP2 = TG / PAIR;          // tile pairs per k-step, layer 2 (one group)
    constexpr int D = mlp_depth(P), kWRing = D + 1, kXRing = D + 1;
    constexpr int TAB_BYTES = mlp_tab_bytes<NT, P>();
    // LDS-DMA instructions per wave and stage (1 KiB each); every wave issues the same number so that the wait counts
    // below are uniform (surplus instructions re-fetch an earlier block of the same stage: same bytes, same place)
    constexpr int NW = (NT * P + kWaves - 1) / kWaves;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* tabs = reinterpret_cast<float*>(lds + kWRing * ST1);         // bias1 | bias2 | wlast | rs1 | rs2 | blast
    volatile uint32_t* vote = reinterpret_cast<volatile uint32_t*>(tabs + 5 * NT * 32 + 4);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, hf = lane >> 5;
    uint8_t* xring = lds + kWRing * ST1 + TAB_BYTES + wave * (kXRing * 2048);
    // 32-bit LDS byte addresses for the hand-written ds_reads
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)lds;
    const uint32_t ring_a = lds0 + lane * 16;                                   // + slot * ST1: lane-ready plane blocks
    const uint32_t xring_a = lds0 + kWRing * ST1 + TAB_BYTES + wave * (kXRing * 2048);
    const MlpLayout L = mlp_layout(a.K0, NT, a.n_hidden);
    const int KS1 = L.KS1;
    const int Q = KS1 + (a.n_hidden >= 2 ? NG * NS2 : 0);              // stages of the whole stream
    const uint8_t* w1 = a.packed + L.l1_off[P];
    const uint8_t* w2 = a.packed + L.l2_off[P];

    // stage q of the weight stream -> ring slot q % kWRing: NT*P lane-linear 1-KiB blocks, NW per wave
    auto issue_w = [&](int q) {
#ifdef ARMNET_MLP_NOW              // developer ablation (compile-time, results are garbage): no weight LDS-DMA
        return;
#endif
        const uint8_t* src = q < KS1 ? w1 + (int64_t)q * ST1 : w2 + (int64_t)(q - KS1) * ST2;
        uint8_t* dst = lds + (q % kWRing) * ST1;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int blk = (wave + i * kWaves) % (NT * P);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + blk * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + blk * 1024), 16, 0, 0);
        }
    };
    // Activation tile of k-step s: this wave's 32 rows x 64 bytes, fetched straight into LDS by two 1-KiB LDS-DMA
    // instructions whose lanes cover whole 64-byte row segments (4 lanes per row: coalesced, unlike loading the MFMA
    // B fragment — one row per lane — directly).  The DMA's LDS image is lane-linear, so the bank-conflict-free
    // layout is made on the SOURCE side: LDS slot c' of row r holds 16-byte piece c' ^ ((r >> 2) & 3).
    const int64_t row0 = (int64_t)blockIdx.x * (32 * kWaves) + wave * 32;
    const float* xsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 16 * j + (lane >> 2);
        const int64_t rg = row0 + r < a.B ? row0 + r : a.B - 1;
        xsrc[j] = a.x + rg * a.ldx + 4 * ((lane & 3) ^ ((r >> 2) & 3));
    }
    auto issue_x = [&](int s) {
#ifdef ARMNET_MLP_NOX              // developer ablation (compile-time, results are garbage): no activation LDS-DMA
        return;
#endif
        uint8_t* dst = xring + (s % kXRing) * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xsrc[j] + 16 * s),
                                             (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    };
    // B fragment of lane (m, hf) for k-step s: pieces 2hf, 2hf+1 of row m
    const uint32_t xrd = (4 * m + ((2 * hf) ^ ((m >> 2) & 3))) * 16;
    auto read_x = [&](int s, u32x4 (&v)[2]) {             // two hand-written ds_reads (see lds_read16)
        const uint32_t t = xring_a + (s % kXRing) * 2048;
        v[0] = lds_read16(t + xrd);
        v[1] = lds_read16(t + (xrd ^ 16));
    };
    auto raw_floats = [&](const u32x4 (&v)[2], float (&x)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { x[i] = __uint_as_float(v[0][i]); x[4 + i] = __uint_as_float(v[1][i]); }
    };

    // ---- the wave's LDS-DMA instructions, in issue order, and the counted waits that follow from it ----------------
    //   prologue:            W(0) .. W(D-1) [NW each]   X(0) .. X(D) [2 each]
    //   during stage j:      W(j+D) if j+D < Q          X(j+D+1) if j+D+1 < KS1
    // Stage s needs W(s) and — to split it one k-step ahead — X(s+1); the counter retires in order, so everything
    // issued behind the later of the two may stay in flight.
    const int PW = NW * (D < Q ? D : Q);
    const int PX = PW + 2 * (D + 1 < KS1 ? D + 1 : KS1);
    auto issued_before = [&](int j) {                      // instructions issued before stage j begins
        int cw = j < Q - D ? j : Q - D;
        int cx = j < KS1 - D - 1 ? j : KS1 - D - 1;
        return PX + NW * (cw > 0 ? cw : 0) + 2 * (cx > 0 ? cx : 0);
    };
    auto w_end = [&](int q) { return q < D ? NW * (q + 1) : issued_before(q - D) + NW; };
    auto x_end = [&](int t) { return t <= D ? PW + 2 * (t + 1) : issued_before(t - D - 1) + NW + 2; };
    auto vm_allowed = [&](int s) {
        const int we = w_end(s), xe = s + 1 < KS1 ? x_end(s + 1) : 0;
        return issued_before(s) - (we > xe ? we : xe);
    };
    constexpr int kSteady1 = (D - 1) * (NW + 2), kSteady2 = (D - 1) * NW;
    auto stage_wait = [&](int s) {
        const int n = vm_allowed(s);
        if (n == kSteady1) wait_vm<kSteady1>();
        else if (n == kSteady2) wait_vm<kSteady2>();
        else wait_vm_n(n);
    };

    // ---- prologue -------------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < D; ++q) if (q < Q) issue_w(q);
#pragma unroll
    for (int s = 0; s <= D; ++s) if (s < KS1) issue_x(s);
    {   // tables -> LDS (plain loads; ordered before everything that reads them by the first barrier)
        const float* src = reinterpret_cast<const float*>(a.packed + L.tab_off);
        for (int i = threadIdx.x; i < 5 * NT * 32 + 4; i += 64 * kWaves) tabs[i] = src[i];
        if (threadIdx.x == 0) *vote = 0;
    }
    const int64_t row = row0 + m;

    // ---- layer 1: K0 -> NT*32 hidden units --------------------------------------------------------------------
    // Per k-step s (= one stage of the weight stream): wait until W(s) and X(s+1) have landed, barrier; then the
    // products with, in their shadows, the rolling LDS reads of the next tile pair's planes, the split of X(s+1),
    // and the requests for W(s+D) and X(s+D+1) (an LDS-DMA instruction holds the issuing wave for ~100 cycles:
    // measured 840 cycles per stage when 8 of them sat in front of the MFMAs).
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 rawv[2];
    float raw[8];
    BPlanes<P> bc, bn;
    float mx = 0.f;
    wait_vm_n(PX - x_end(0));                           // X(0) (and every W of the prologue) landed
    read_x(0, rawv);                                    // wave-private tile: no barrier needed
    lds_wait<0, 2>(rawv);
    raw_floats(rawv, raw);
    split_planes(raw, kXScale, bc, mx);
#ifdef ARMNET_DEV_FLAGS
    unsigned long long ph[4] = {0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < KS1; ++s) {
        stage_wait(s);
        MLP_PHASE(0);
        block_barrier();
        MLP_PHASE(1);
        const uint32_t st = ring_a + (s % kWRing) * ST1;
        if (s + 1 < KS1) read_x(s + 1, rawv);           // older than the plane reads: valid once the first plane is
        APlanes<P, PAIR> A;
        load_planes<P, PAIR>(A, st, 0);
        __builtin_amdgcn_sched_barrier(0);
        MLP_PHASE(2);
        lds_wait<P * PAIR, 2>(rawv);
        raw_floats(rawv, raw);
        split_planes(raw, kXScale, bn, mx);
#pragma unroll
        for (int p = 0; p < NP1; ++p) {
            if (p + 1 < NP1) unit<P, PAIR, true>(acc + p * PAIR, A, bc, st, (p + 1) * PAIR);
            else unit<P, PAIR, false>(acc + p * PAIR, A, bc, st, 0);
            if (p == 0) {                               // requests ride behind the first unit's MFMAs
                if (s + D < Q) issue_w(s + D);
                if (s + D + 1 < KS1) issue_x(s + D + 1);
                __builtin_amdgcn_sched_barrier(0x2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        MLP_PHASE(3);
        bc = bn;
    }
#ifdef ARMNET_DEV_FLAGS
    if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_mlp_phase[i], ph[i]);
#endif
    if constexpr (P == MLP_F16X2) {
        // the block's vote on the fp16 range of what its waves split (inf included; a NaN poisons its own sample only,
        // as it does in the reference)
        // (the all-tiny test is per WAVE: 32 samples x K0 inputs all below 1e-3 — dead or badly scaled inputs, not one quiet sample)
        const bool out_of_range = __builtin_amdgcn_ballot_w64(!(mx <= kXLimit)) != 0 || __builtin_amdgcn_ballot_w64(mx >= kXTiny) == 0;
        if (out_of_range && lane == 0) *vote = 1;
        block_barrier();
        if (*vote != 0) {
            wait_vm<0>();                               // the layer-2 prefetch must not land in the redo's ring
            block_barrier();
            return true;
        }
    }
    // bias (BatchNorm folded) + ReLU, in place: acc becomes H1 in C layout.  fp16 split: the accumulator carries c * 2^s_n
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (hf * NT + t) * 16 + r;
            const float v = P == MLP_F16X2 ? fmaf(acc[t][r], tabs[3 * NT * 32 + i], tabs[i]) : acc[t][r] + tabs[i];
            acc[t][r] = a.linear ? v : relu_keep_nan(v);
        }

    float part = 0.f;
    const float* wl = tabs + 2 * NT * 32;
    auto store_hidden = [&](const f32x16& h, int t) {
        if (row < a.B) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = c_layout_unit(t, r, hf);
                if (n < a.N) a.out[row * a.ldo + n] = h[r];
            }
        }
    };
    if (a.n_hidden >= 2) {
        // ---- layer 2: the previous layer's accumulators ARE this layer's B operands ----------------------------
        // fp16 split: one power of two per sample puts its largest hidden activation in [2^14, 2^15)
        float dn = 1.f, up = 1.f;
        if constexpr (P == MLP_F16X2) {
            float hm = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) hm = fmaxf(hm, acc[t][r]);
            hm = fmaxf(hm, __shfl_xor(hm, 32));
            int e = __builtin_amdgcn_frexp_expf(hm);    // hm = f * 2^e, f in [0.5, 1); 0 for hm = 0 / inf / nan
            e = hm > 0.f ? (e < -100 ? -100 : e) : 15;
            dn = __builtin_amdgcn_ldexpf(1.0f, 15 - e);
            up = __builtin_amdgcn_ldexpf(1.0f, e - 15);
        }
        // k-step s2 takes registers 8u..8u+7 (u = s2 & 1) of tile s2 >> 1; its planes are split one k-step ahead
        float unused = 0.f;
        auto split_step = [&](int s2, BPlanes<P>& b) {
            float xc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xc[j] = acc[s2 >> 1][8 * (s2 & 1) + j];
            split_planes(xc, dn, b, unused);
        };
        split_step(0, bc);
        int q = KS1;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x16 acc2[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
            for (int sg = 0; sg < NS2; ++sg) {
                stage_wait(q);
                block_barrier();
                const uint32_t st = ring_a + (q % kWRing) * ST1;
                APlanes<P, PAIR> A;
                load_planes<P, PAIR>(A, st, 0);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int U = KPS * NP2;                             // (k-step, tile pair) units of this stage
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    const int s2 = sg * KPS + kk;
                    split_step((s2 + 1) % (2 * NT), bn);                 // next k-step (wraps into the next group)
#pragma unroll
                    for (int p = 0; p < NP2; ++p) {
                        const int u = kk * NP2 + p;
                        const uint32_t nxt = st + ((u + 1) / NP2) * TG * P * 1024;
                        const int t0n = ((u + 1) % NP2) * PAIR;
                        if (u + 1 < U) unit<P, PAIR, true>(acc2 + p * PAIR, A, bc, nxt, t0n);
                        else unit<P, PAIR, false>(acc2 + p * PAIR, A, bc, nxt, t0n);
                        if (u == 0) {
                            if (q + D < Q) issue_w(q + D);
                            __builtin_amdgcn_sched_barrier(0x2);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bc = bn;
                }
                ++q;
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int tg = g * TG + t;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (hf * NT + tg) * 16 + r;
                    const float v = P == MLP_F16X2 ? fmaf(acc2[t][r], tabs[4 * NT * 32 + i] * up, tabs[NT * 32 + i])
                                                   : acc2[t][r] + tabs[NT * 32 + i];
                    const float h = relu_keep_nan(v);
                    if (a.has_final) part = fmaf(h, wl[i], part);
                    else acc2[t][r] = h;
                }
                if (!a.has_final) store_hidden(acc2[t], tg);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (a.has_final) {
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(acc[t][r], wl[(hf * NT + t) * 16 + r], part);
            } else {
                store_hidden(acc[t], t);
            }
        }
    }
    if (a.has_final) {
        part += __shfl_xor(part, 32);
        // has_final == 2: this launch holds a SLICE of a wider last hidden layer: add its share of the final Linear
        if (hf == 0 && row < a.B) {
            const float v = part + tabs[5 * NT * 32];
            a.out[row] = a.has_final == 2 ? a.out[row] + v : v;
        }
    }
    return false;
}

// Waves per block (kWaves, template parameter): 32 * kWaves samples share one pass of the weight stream.  8 at large
// batches (256 blocks of 256 samples fill the chip at B = 65 536), 4 when the batch would otherwise leave CUs idle
// (launch_mlp).
template <int NT, int kWaves>
__global__ void __launch_bounds__(64 * kWaves, kWaves >= 8 ? 2 : 1) mlp_head_kernel(MlpArgs a) {
    if (!a.bf16x3) {
        if (!mlp_body<NT, kWaves, MLP_F16X2>(a)) return;
    }
    mlp_body<NT, kWaves, MLP_BF16X3>(a);
}

template <int NT, int kWaves>
static int launch_mlp_kw(const MlpArgs& a, hipStream_t st) {
    constexpr size_t lds2 = mlp_lds_bytes<NT, kWaves, MLP_F16X2>(), lds3 = mlp_lds_bytes<NT, kWaves, MLP_BF16X3>();
    constexpr size_t lds = lds2 > lds3 ? lds2 : lds3;
    static_assert(lds <= 160 * 1024, "the block's rings exceed the LDS");
    auto kern = mlp_head_kernel<NT, kWaves>;
    ARMNET_ALLOW_BIG_LDS(kern, lds);
    const int64_t blocks = (a.B + 32 * kWaves - 1) / (32 * kWaves);
    kern<<<(int)blocks, 64 * kWaves, lds, st>>>(a);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// Waves per block for a batch.  A block runs for about the same time whatever the batch (it streams every weight), so
// 4-wave blocks (128 samples, one wave per SIMD, no spills at 512 registers) win as long as ALL of them are resident
// at once — one per CU: B <= 128 * CUs = 32 768 on MI355X (65-80 us against 94-107 us for 8-wave blocks) —; beyond
// that the 8-wave blocks' halved weight traffic per sample wins (B = 65 536: 147 against 171 us).
template <int NT>
static int launch_mlp(const MlpArgs& a, hipStream_t st) {
    int kw = (a.B + 127) / 128 <= device_cu_count() ? 4 : 8;
#if defined(ARMNET_DEV_FLAGS) || defined(ARMNET_MLP_KW_ENV)
    if (const char* e = getenv("ARMNET_MLP_KW")) kw = atoi(e);          // developer knob
#endif
    return kw == 4 ? launch_mlp_kw<NT, 4>(a, st) : launch_mlp_kw<NT, 8>(a, st);
}

static int launch_mlp_nt(const MlpArgs& a, int nhid, hipStream_t st) {
    switch (mlp_nt_for(nhid)) {
        case 1: return launch_mlp<1>(a, st);
        case 2: return launch_mlp<2>(a, st);
        case 4: return launch_mlp<4>(a, st);
        case 8: return launch_mlp<8>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet

using namespace armnet;

extern "C" {

int armnet_mlp_head_supported(int K0, int nhid, int n_hidden) {
    return (K0 >= 1 && nhid >= 1 && nhid <= 256 && (n_hidden == 1 || n_hidden == 2)) ? 1 : 0;
}

int64_t armnet_mlp_packed_bytes(int K0, int nhid, int n_hidden) {
    if (!armnet_mlp_head_supported(K0, nhid, n_hidden)) return -1;
    return mlp_layout(K0, mlp_nt_for(nhid), n_hidden).total;
}

int armnet_mlp_pack_layer_f32(int K0, int nhid, int n_hidden, int slot, const float* W, int Kin, const float* b,
                              const float* bn_weight, const float* bn_bias, const float* bn_running_mean,
                              const float* bn_running_var, float bn_eps, void* packed, void* stream) {
    if (!armnet_mlp_head_supported(K0, nhid, n_hidden) || !W || !packed) return ARMNET_ERR_BAD_ARG;
    if (slot < 0 || slot > 2 || (slot == 1 && n_hidden < 2)) return ARMNET_ERR_BAD_ARG;
    if ((slot == 0 && Kin != K0) || (slot >= 1 && Kin != nhid)) return ARMNET_ERR_BAD_ARG;
    if (bn_weight && (!bn_bias || !bn_running_mean || !bn_running_var)) return ARMNET_ERR_BAD_ARG;
    PackArgs p{};
    p.K0 = K0; p.NT = mlp_nt_for(nhid); p.n_hidden = n_hidden; p.slot = slot; p.N = nhid; p.Kin = Kin;
    p.W = W; p.b = b; p.bn_w = bn_weight; p.bn_b = bn_bias; p.bn_m = bn_running_mean; p.bn_v = bn_running_var;
    p.eps = bn_eps; p.packed = static_cast<uint8_t*>(packed);
    const MlpLayout L = mlp_layout(K0, p.NT, n_hidden);
    int64_t work = p.NT * 32;
    if (slot == 0) work = L.KS1 * (int64_t)p.NT * 512;
    if (slot == 1) work = (int64_t)L.NG * 2 * p.NT * L.TG * 512;   // k-steps x tiles x (64 lanes x 8)
    if (work < p.NT * 32) work = p.NT * 32;
    if (slot < 2) {
        mlp_rowexp_kernel<<<p.NT * 32, 64, 0, (hipStream_t)stream>>>(p);
        ARMNET_LAUNCH_CHECK();
    }
    mlp_pack_kernel<<<(int)((work + 255) / 256), 256, 0, (hipStream_t)stream>>>(p);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

int armnet_mlp_head_ex_f32(int64_t B, int K0, int nhid, int n_hidden, int has_final, const float* x, int64_t ldx,
                           const void* packed, float* out, int64_t ldo, uint32_t flags, void* stream) {
    // rows of x are read in whole 16-float k-steps: the row stride must cover the rounded-up width (the columns past
    // K0 meet zero weights; they only have to be readable and finite)
    if (B < 0 || !armnet_mlp_head_supported(K0, nhid, n_hidden) || ldx < (int64_t)((K0 + 15) / 16) * 16) return ARMNET_ERR_BAD_ARG;
    if (!has_final && ldo < nhid) return ARMNET_ERR_BAD_ARG;
    if (flags & ~(uint32_t)ARMNET_MLP_F_BF16X3) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!x || !packed || !out) return ARMNET_ERR_BAD_ARG;
    MlpArgs a{};
    if (has_final < 0 || has_final > 2) return ARMNET_ERR_BAD_ARG;
    a.B = B; a.K0 = K0; a.n_hidden = n_hidden; a.has_final = has_final; a.N = nhid; a.ldx = ldx; a.ldo = ldo;
    a.x = x; a.packed = static_cast<const uint8_t*>(packed); a.out = out;
    a.bf16x3 = (flags & ARMNET_MLP_F_BF16X3) ? 1 : 0;
#if defined(ARMNET_DEV_FLAGS) || defined(ARMNET_MLP_KW_ENV)
    if (const char* e = getenv("ARMNET_MLP_BF16X3")) a.bf16x3 = atoi(e);   // developer knob
#endif
    return launch_mlp_nt(a, nhid, (hipStream_t)stream);
}

int armnet_mlp_head_f32(int64_t B, int K0, int nhid, int n_hidden, int has_final, const float* x, int64_t ldx,
                        const void* packed, float* out, int64_t ldo, void* stream) {
    return armnet_mlp_head_ex_f32(B, K0, nhid, n_hidden, has_final, x, ldx, packed, out, ldo, 0u, stream);
}

// A plain Linear on the same kernel (round 5): out[b, n] = bias[n] + sum_k x[b, k] W[n, k], N <= 256 outputs, no
// BatchNorm fold, no ReLU, bf16x3 split.  `packed` = armnet_mlp_pack_layer_f32(K, N, 1, slot 0, W [N, K], K, bias | NULL, no BatchNorm).
int armnet_linear_bf16x3_f32(int64_t B, int K, int N, const float* x, int64_t ldx, const void* packed, float* out,
                             int64_t ldo, void* stream) {
    if (B < 0 || !armnet_mlp_head_supported(K, N, 1) || ldx < (int64_t)((K + 15) / 16) * 16 || ldo < N) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!x || !packed || !out) return ARMNET_ERR_BAD_ARG;
    MlpArgs a{};
    a.B = B; a.K0 = K; a.n_hidden = 1; a.has_final = 0; a.N = N; a.ldx = ldx; a.ldo = ldo;
    a.x = x; a.packed = static_cast<const uint8_t*>(packed); a.out = out; a.linear = 1; a.bf16x3 = 1;
    return launch_mlp_nt(a, N, (hipStream_t)stream);
}

#ifdef ARMNET_DEV_FLAGS
// developer build only: read and reset the per-phase cycle sums of mlp_head_kernel
void armnet_dev_mlp_phases(unsigned long long* out8) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mlp_phase), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_phase), z, sizeof(z));
}
#endif

}  // extern "C"
