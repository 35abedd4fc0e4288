// mlp_head.hip — the eval-mode prediction head (models/layers.py:68-88: (Linear, BatchNorm1d, ReLU, Dropout) x n, then
// Linear(., 1)) as ONE kernel on the CDNA4 bf16 matrix cores with fp32-equivalent numerics.  gfx950 only.
//
// Why not fp32 MFMA: v_mfma_f32_*_f32 runs at the fp32 VECTOR rate (157 TFLOP/s); hipBLASLt already sits there
// (2 x 110 us for the two GEMMs of the 2x256 head at B = 65 536, more than the fused ARM block in front of them).
// v_mfma_f32_32x32x16_bf16 is 16x faster per product, so a 3-way bf16 split with the 6 significant cross products
//      x = xh + xm + xl (exactly: three 8-bit slices of the 24-bit significand, by truncation)
//      w = wh + wm + wl (round-to-nearest, parameter-only precompute: armnet_mlp_pack_layer_f32)
//      x*w ~= xh*wh + xh*wm + xm*wh + xm*wm + xh*wl + xl*wh          (dropped: 2 terms of relative size 2^-24)
// accumulated in fp32 inside the matrix core costs 6/16 of the fp32-MFMA time at fp32-class error: every product of
// two bf16 numbers is exact in fp32, the dropped terms are at the level of one fp32 rounding of the product.
// The split of the ACTIVATIONS costs ~5.5 VALU ops per element and each element feeds 8 x 6 MFMAs; the split of the
// weights is free (done once, like q_fold).
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
// (one k-step of all its tiles) through a 3-deep LDS ring with global_load_lds_dwordx4, one barrier per stage; each
// wave's activations come through its own 3-deep LDS ring the same way.
#include <stdlib.h>

#include "armnet_common.h"

namespace armnet {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4m __attribute__((ext_vector_type(4)));
typedef f32x4m f32x4mu __attribute__((aligned(4)));   // gfx950: unaligned-mode dwordx4

// layer >= 2 runs in groups of TG output tiles (the previous layer's 16*NT accumulator registers stay live as its B
// operands, so only TG*16 more can be spent on accumulators); a stage of its weight stream covers KPS k-steps
__host__ __device__ constexpr int mlp_tg(int NT) { return NT >= 8 ? 2 : (NT < 4 ? NT : 4); }
__host__ __device__ constexpr int mlp_kps(int NT) { return NT >= 8 ? 4 : 1; }

struct MlpLayout {
    int NT, TG, NG, KPS, KS1, NS2;    // NS2: stages per group of layer 2
    int64_t st1_bytes, st2_bytes;     // bytes per stage of layer 1 / layer 2
    int64_t l2_off;                   // start of the layer-2 stream
    int64_t tab_off;                  // start of the fp32 tables: bias1, bias2, wlast (NT*32 floats each), blast (4 floats)
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
    L.st1_bytes = (int64_t)NT * 3 * 1024;
    L.st2_bytes = (int64_t)L.KPS * L.TG * 3 * 1024;
    L.l2_off = (int64_t)L.KS1 * L.st1_bytes;
    const int64_t l2_bytes = n_hidden >= 2 ? (int64_t)L.NG * L.NS2 * L.st2_bytes : 0;
    L.tab_off = L.l2_off + l2_bytes;
    L.total = L.tab_off + ((int64_t)3 * NT * 32 + 4) * sizeof(float);
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

__global__ void mlp_pack_kernel(PackArgs p) {
    const MlpLayout L = mlp_layout(p.K0, p.NT, p.n_hidden);
    float* tabs = reinterpret_cast<float*>(p.packed + L.tab_off);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto scale_of = [&](int n) -> float { return p.bn_w ? p.bn_w[n] / sqrtf(p.bn_v[n] + p.eps) : 1.0f; };
    if (p.slot == 2) {                                   // final Linear(N, 1): weights in C-layout order + its bias
        if (gid < p.NT * 32) {
            const int half = (int)gid / (p.NT * 16), t = ((int)gid / 16) % p.NT, r = (int)gid & 15;
            const int n = c_layout_unit(t, r, half);
            tabs[2 * p.NT * 32 + gid] = n < p.N ? p.W[n] : 0.f;
        }
        if (gid == 0) tabs[3 * p.NT * 32] = p.b ? p.b[0] : 0.f;
        return;
    }
    // bias table of this hidden layer (BatchNorm folded): b' = b * s + (beta - mean * s)
    if (gid < p.NT * 32) {
        const int half = (int)gid / (p.NT * 16), t = ((int)gid / 16) % p.NT, r = (int)gid & 15;
        const int n = c_layout_unit(t, r, half);
        float v = 0.f;
        if (n < p.N) {
            const float s = scale_of(n);
            const float bb = p.b ? p.b[n] : 0.f;
            v = p.bn_w ? bb * s + (p.bn_b[n] - p.bn_m[n] * s) : bb;
        }
        tabs[p.slot * p.NT * 32 + gid] = v;
    }
    // weight stream: one thread per (stage, tile, lane, j) -> three bf16 planes
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
    if (n < p.N && k < p.Kin) w = p.W[(size_t)n * p.Kin + k] * scale_of(n);
    const uint32_t hb = bf16_rn_bits(w);
    const float r1 = w - __uint_as_float(hb << 16);
    const uint32_t mb = bf16_rn_bits(r1);
    const float r2 = r1 - __uint_as_float(mb << 16);
    const uint32_t lb = bf16_rn_bits(r2);
    uint16_t* dst = reinterpret_cast<uint16_t*>(p.packed + (p.slot == 0 ? 0 : L.l2_off)) +
                    ((size_t)stage * tiles + tt) * 3 * 512 + lane * 8 + j;
    dst[0] = (uint16_t)hb;
    dst[512] = (uint16_t)mb;
    dst[1024] = (uint16_t)lb;
}

// ---------------------------------------------------------------------------------------------------------------
struct MlpArgs {
    int64_t B;
    int K0, n_hidden, has_final, N;
    int64_t ldx, ldo;      // row strides (floats) of x and of the hidden-activation output
    const float* x;
    const uint8_t* packed;
    float* out;
    int dbg;      // developer ablation switches (ARMNET_DEV_FLAGS builds only; 0 in product builds)
    int linear;   // 1: a plain Linear — hidden activations are written WITHOUT the ReLU (armnet_linear_bf16x3_f32: the
                  // training head's GEMMs, where BatchNorm needs the batch's pre-activation values)
};

// 8 fp32 -> three packed bf16x8 planes; h + m + l == x exactly (truncating 8-bit slices of the significand)
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& ph, u32x4& pm, u32x4& pl) {
#ifdef ARMNET_MLP_NOSPLIT          // developer ablation (compile-time, results are garbage): no bf16 split
    for (int i = 0; i < 4; ++i) { ph[i] = __float_as_uint(x[i]); pm[i] = __float_as_uint(x[4 + i]); pl[i] = ph[i]; }
    return;
#endif
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
        ph[i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
        pm[i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
        pl[i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// A operands (three bf16 planes) of a pair of output tiles
template <int PAIR>
struct APlanes { u32x4 h[PAIR], m[PAIR], l[PAIR]; };

// ---- LDS reads by hand ------------------------------------------------------------------------------------------
// While an LDS-DMA (global_load_lds) is in flight hipcc's wait-count pass treats the LGKM counter as unordered ("pending
// flat": the DMA carries an LDS memory operand) and turns EVERY wait for a ds_read result into lgkmcnt(0) — a full drain
// that also waits for the reads issued one instruction earlier (measured: four exposed LDS round trips per k-step,
// a lone wave at 45 % of the matrix-core rate).  LDS returns in order, so the weight planes and activation tiles are
// read with inline-asm ds_read_b128 and waited for with exact counts; each wait names the registers it makes valid
// ("+v"), which orders their consumers behind it.
__device__ __forceinline__ u32x4 lds_read16(uint32_t addr) {
    u32x4 v;
#ifdef ARMNET_MLP_NOREAD           // developer ablation (compile-time, results are garbage): no LDS plane / tile reads
    asm volatile("v_mov_b32 %0, %1" : "=v"(v[0]) : "v"(addr));
    v[1] = v[2] = v[3] = addr;
    return v;
#endif
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int N, int PAIR>
__device__ __forceinline__ void lds_wait(u32x4 (&r)[PAIR]) {     // at most N younger LDS reads still outstanding
    if constexpr (PAIR == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r[0]), "+v"(r[1]) : "n"(N) : "memory");
    else asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r[0]) : "n"(N) : "memory");
}

template <int PAIR>
__device__ __forceinline__ void load_planes(APlanes<PAIR>& A, uint32_t blk, int t0) {   // blk: LDS byte address + lane*16
#pragma unroll
    for (int u = 0; u < PAIR; ++u) A.l[u] = lds_read16(blk + ((t0 + u) * 3 + 2) * 1024);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) A.m[u] = lds_read16(blk + ((t0 + u) * 3 + 1) * 1024);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) A.h[u] = lds_read16(blk + ((t0 + u) * 3 + 0) * 1024);
}

// One (k-step, tile pair) unit: the six significant cross products for PAIR tiles, smallest terms first
//      l*bh | m*bm  m*bh | h*bl  h*bm  h*bh
// with ONE set of A registers that rolls over to the next unit: a plane's registers are reloaded from LDS (`nxt`: LDS
// byte address + lane*16 of the next unit's k-step block; t0n its first tile) as soon as its last product has been
// issued, so the next unit's l / m / h planes have 5 / 4 / 3 MFMA slots (64 cycles each) to arrive.  At every wait
// exactly the 2*PAIR reads issued after the awaited plane may still be in flight (fewer at the end of a stage).  The
// PAIR accumulators alternate so that consecutive MFMAs never wait on each other.  sched_barrier(VALU) keeps MFMAs and
// LDS traffic in this order and lets the VALU work (the bf16 split of the next k-step) float into the MFMA shadows.
template <int PAIR, bool NEXT>
__device__ __forceinline__ void unit(f32x16* acc, APlanes<PAIR>& A, u32x4 bh, u32x4 bm, u32x4 bl, uint32_t nxt, int t0n) {
    constexpr int W = 2 * PAIR;
    lds_wait<W, PAIR>(A.l);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) acc[u] = mfma_bf16(A.l[u], bh, acc[u]);
    __builtin_amdgcn_sched_barrier(0x2);
    if (NEXT) {
#pragma unroll
        for (int u = 0; u < PAIR; ++u) A.l[u] = lds_read16(nxt + ((t0n + u) * 3 + 2) * 1024);
    }
    lds_wait<NEXT ? W : PAIR, PAIR>(A.m);
    __builtin_amdgcn_sched_barrier(0x2);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) acc[u] = mfma_bf16(A.m[u], bm, acc[u]);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) acc[u] = mfma_bf16(A.m[u], bh, acc[u]);
    __builtin_amdgcn_sched_barrier(0x2);
    if (NEXT) {
#pragma unroll
        for (int u = 0; u < PAIR; ++u) A.m[u] = lds_read16(nxt + ((t0n + u) * 3 + 1) * 1024);
    }
    lds_wait<NEXT ? W : 0, PAIR>(A.h);
    __builtin_amdgcn_sched_barrier(0x2);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) acc[u] = mfma_bf16(A.h[u], bl, acc[u]);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) acc[u] = mfma_bf16(A.h[u], bm, acc[u]);
#pragma unroll
    for (int u = 0; u < PAIR; ++u) acc[u] = mfma_bf16(A.h[u], bh, acc[u]);
    __builtin_amdgcn_sched_barrier(0x2);
    if (NEXT) {
#pragma unroll
        for (int u = 0; u < PAIR; ++u) A.h[u] = lds_read16(nxt + ((t0n + u) * 3 + 0) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0x2);
}

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// raw barrier: __syncthreads() would drain every LDS-DMA in flight (it fences with vmcnt(0))
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// Waves per block (kWaves, template parameter): 32 * kWaves samples share one pass of the weight stream.  8 at large
// batches (256 blocks of 256 samples fill the chip at B = 65 536), 4 when the batch would otherwise leave CUs idle
// (launch_mlp).
constexpr int kWRing = 3;                 // weight stages resident in LDS (consumed | landing | requested)
constexpr int kXRing = 3;                 // activation tiles per wave in LDS (read | landing | requested)

#ifdef ARMNET_DEV_FLAGS
// developer build: per-phase s_memtime sums over all waves (layer-1 stages): wait | barrier | first reads | units
__device__ unsigned long long g_mlp_phase[8];
#define MLP_PHASE(i) do { const unsigned long long _n = __builtin_amdgcn_s_memtime(); ph[i] += _n - pt; pt = _n; } while (0)
#else
#define MLP_PHASE(i) do {} while (0)
#endif

template <int NT, int kWaves>
__global__ void __launch_bounds__(64 * kWaves, kWaves >= 8 ? 2 : 1) mlp_head_kernel(MlpArgs a) {
    constexpr int TG = mlp_tg(NT), NG = NT / TG, KPS = mlp_kps(NT), NS2 = 2 * NT / KPS;
    constexpr int ST1 = NT * 3 * 1024, ST2 = KPS * TG * 3 * 1024;
    static_assert(ST2 <= ST1, "a layer-2 stage must fit a ring slot");
    constexpr int PAIR = NT >= 2 ? 2 : 1;
    constexpr int NP1 = NT / PAIR;          // tile pairs per k-step, layer 1
    constexpr int NP2 = TG / PAIR;          // tile pairs per k-step, layer 2 (one group)
    constexpr int TAB_BYTES = ((3 * NT * 32 + 4) * 4 + 15) & ~15;
    // LDS-DMA instructions per wave and stage (1 KiB each); every wave issues the same number so that the wait counts
    // below are uniform (surplus instructions re-fetch an earlier block of the same stage: same bytes, same place)
    constexpr int NW = (NT * 3 + kWaves - 1) / kWaves;
    static_assert(KPS * TG == NT, "both layers' stages hold NT*3 blocks");
    // LDS: [kWRing][ST1] weight ring | fp32 tables | [kWaves][kXRing][2 KiB] activation tiles
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    float* tabs = reinterpret_cast<float*>(lds + kWRing * ST1);         // bias1 | bias2 | wlast | blast
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
    // developer ablations: runtime (ARMNET_DEV_FLAGS + ARMNET_MLP_DBG; the instrumentation itself costs ~50 %) or
    // compile-time (-DARMNET_MLP_NODMA / NOSYNC / NOREAD / NOSPLIT: clean timings, garbage results)
#if defined(ARMNET_DEV_FLAGS)
    const bool dbg_nosync = a.dbg & 1, dbg_nox = a.dbg & 2, dbg_noglds = a.dbg & 4;
#elif defined(ARMNET_MLP_NODMA) && defined(ARMNET_MLP_NOSYNC)
    constexpr bool dbg_nosync = true, dbg_nox = true, dbg_noglds = true;
#elif defined(ARMNET_MLP_NODMA)
    constexpr bool dbg_nosync = false, dbg_nox = true, dbg_noglds = true;
#elif defined(ARMNET_MLP_NOSYNC)
    constexpr bool dbg_nosync = true, dbg_nox = false, dbg_noglds = false;
#else
    constexpr bool dbg_nosync = false, dbg_nox = false, dbg_noglds = false;
#endif

    // stage q of the weight stream -> ring slot q % kWRing: NT*3 lane-linear 1-KiB blocks, NW per wave
    auto issue_w = [&](int q) {
        if (dbg_noglds) return;
        const uint8_t* src = a.packed + (q < KS1 ? (int64_t)q * ST1 : L.l2_off + (int64_t)(q - KS1) * ST2);
        uint8_t* dst = lds + (q % kWRing) * ST1;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int blk = (wave + i * kWaves) % (NT * 3);
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
        if (dbg_nox) return;
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

#ifdef ARMNET_MLP_SETPRIO
    if (wave >= kWaves / 2) __builtin_amdgcn_s_setprio(1);   // static priority for the younger wave of every SIMD
#endif
    // ---- prologue -------------------------------------------------------------------------------------------
    // in order: W(0) W(1) X(0) X(1) X(2)
    issue_w(0);
    if (Q > 1) issue_w(1);
    issue_x(0);
    if (KS1 > 1) issue_x(1);
    if (KS1 > 2) issue_x(2);
    {   // tables -> LDS (plain loads; ordered before everything that reads them by the first barrier)
        const float* src = reinterpret_cast<const float*>(a.packed + L.tab_off);
        for (int i = threadIdx.x; i < 3 * NT * 32 + 4; i += 64 * kWaves) tabs[i] = src[i];
    }
    const int64_t row = row0 + m;

    // ---- layer 1: K0 -> NT*32 hidden units --------------------------------------------------------------------
    // Per k-step s (= one stage of the weight stream): wait until W(s) and X(s+1) — requested two stages ago — have
    // landed, barrier; then 6*NT MFMAs with, in their shadows, the rolling LDS reads of the next tile pair's planes,
    // the bf16 split of X(s+1), and the requests for W(s+2) and X(s+3) (an LDS-DMA instruction holds the issuing wave
    // for ~100 cycles: measured 840 cycles per stage when 8 of them sat in front of the MFMAs).
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    u32x4 rawv[2];
    float raw[8];
    u32x4 bh, bm, bl, nh, nm, nl;
    // W(0), W(1), X(0) landed; X(1), X(2) may still be in flight
    if (KS1 > 2) wait_vm<4>(); else if (KS1 > 1) wait_vm<2>(); else wait_vm<0>();
    read_x(0, rawv);                                    // wave-private tile: no barrier needed
    lds_wait<0, 2>(rawv);
    raw_floats(rawv, raw);
    split3(raw, bh, bm, bl);
#ifdef ARMNET_DEV_FLAGS
    unsigned long long ph[4] = {0, 0, 0, 0}, pt = __builtin_amdgcn_s_memtime();
#endif
    for (int s = 0; s < KS1; ++s) {
        if (!dbg_nosync) {
            // in flight and allowed to stay: what stage s-1 requested, W(s+1) and X(s+2) (at s = 0 the prologue's X(2))
            const bool w_out = s >= 1 && s + 1 < Q, x_out = s + 2 < KS1;
            if (w_out && x_out) wait_vm<NW + 2>();
            else if (w_out) wait_vm<NW>();
            else if (x_out) wait_vm<2>();
            else wait_vm<0>();
            MLP_PHASE(0);
            block_barrier();
            MLP_PHASE(1);
        }
        const uint32_t st = ring_a + (s % kWRing) * ST1;
        if (s + 1 < KS1) read_x(s + 1, rawv);           // older than the plane reads: valid once the first plane is
        APlanes<PAIR> A;
        load_planes<PAIR>(A, st, 0);
        __builtin_amdgcn_sched_barrier(0);
        MLP_PHASE(2);
        lds_wait<3 * PAIR, 2>(rawv);
        raw_floats(rawv, raw);
        split3(raw, nh, nm, nl);
#pragma unroll
        for (int p = 0; p < NP1; ++p) {
            if (p + 1 < NP1) unit<PAIR, true>(acc + p * PAIR, A, bh, bm, bl, st, (p + 1) * PAIR);
            else unit<PAIR, false>(acc + p * PAIR, A, bh, bm, bl, st, 0);
            if (p == 0) {                               // requests ride behind the first unit's MFMAs
                if (s + 2 < Q) issue_w(s + 2);
                if (s + 3 < KS1) issue_x(s + 3);
                __builtin_amdgcn_sched_barrier(0x2);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        MLP_PHASE(3);
        bh = nh; bm = nm; bl = nl;
    }
#ifdef ARMNET_DEV_FLAGS
    if (lane == 0) for (int i = 0; i < 4; ++i) atomicAdd(&g_mlp_phase[i], ph[i]);
#endif
    // bias (BatchNorm folded) + ReLU, in place: acc becomes H1 in C layout
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[t][r] + tabs[(hf * NT + t) * 16 + r];
            acc[t][r] = a.linear ? v : fmaxf(v, 0.f);
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
        // k-step s2 takes registers 8u..8u+7 (u = s2 & 1) of tile s2 >> 1; its planes are split one k-step ahead
        auto split_step = [&](int s2, u32x4& ph_, u32x4& pm_, u32x4& pl_) {
            float xc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) xc[j] = acc[s2 >> 1][8 * (s2 & 1) + j];
            split3(xc, ph_, pm_, pl_);
        };
        split_step(0, bh, bm, bl);
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
                if (!dbg_nosync) {
                    if (q >= 1 && q + 1 < Q) wait_vm<NW>(); else wait_vm<0>();   // W(q+1) may stay in flight
                    block_barrier();
                }
                const uint32_t st = ring_a + (q % kWRing) * ST1;
                APlanes<PAIR> A;
                load_planes<PAIR>(A, st, 0);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int U = KPS * NP2;                             // (k-step, tile pair) units of this stage
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                    const int s2 = sg * KPS + kk;
                    split_step((s2 + 1) % (2 * NT), nh, nm, nl);         // next k-step (wraps into the next group)
#pragma unroll
                    for (int p = 0; p < NP2; ++p) {
                        const int u = kk * NP2 + p;
                        const uint32_t nxt = st + ((u + 1) / NP2) * TG * 3 * 1024;
                        const int t0n = ((u + 1) % NP2) * PAIR;
                        if (u + 1 < U) unit<PAIR, true>(acc2 + p * PAIR, A, bh, bm, bl, nxt, t0n);
                        else unit<PAIR, false>(acc2 + p * PAIR, A, bh, bm, bl, nxt, t0n);
                        if (u == 0) {
                            if (q + 2 < Q) issue_w(q + 2);
                            __builtin_amdgcn_sched_barrier(0x2);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    bh = nh; bm = nm; bl = nl;
                }
                ++q;
            }
#pragma unroll
            for (int t = 0; t < TG; ++t) {
                const int tg = g * TG + t;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h = fmaxf(acc2[t][r] + tabs[NT * 32 + (hf * NT + tg) * 16 + r], 0.f);
                    if (a.has_final) part = fmaf(h, wl[(hf * NT + tg) * 16 + r], part);
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
            const float v = part + tabs[3 * NT * 32];
            a.out[row] = a.has_final == 2 ? a.out[row] + v : v;
        }
    }
}

template <int NT, int kWaves>
static int launch_mlp_kw(const MlpArgs& a, hipStream_t st) {
    static_assert(mlp_kps(NT) * mlp_tg(NT) <= NT, "a layer-2 stage must fit a ring slot");
    const size_t lds = (size_t)kWRing * NT * 3 * 1024 + ((((size_t)3 * NT * 32 + 4) * sizeof(float) + 15) & ~(size_t)15) +
                       (size_t)kWaves * kXRing * 2048;
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
    mlp_pack_kernel<<<(int)((work + 255) / 256), 256, 0, (hipStream_t)stream>>>(p);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

int armnet_mlp_head_f32(int64_t B, int K0, int nhid, int n_hidden, int has_final, const float* x, int64_t ldx,
                        const void* packed, float* out, int64_t ldo, void* stream) {
    // rows of x are read in whole 16-float k-steps: the row stride must cover the rounded-up width (the columns past
    // K0 meet zero weights; they only have to be readable and finite)
    if (B < 0 || !armnet_mlp_head_supported(K0, nhid, n_hidden) || ldx < (int64_t)((K0 + 15) / 16) * 16) return ARMNET_ERR_BAD_ARG;
    if (!has_final && ldo < nhid) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!x || !packed || !out) return ARMNET_ERR_BAD_ARG;
    MlpArgs a{};
    if (has_final < 0 || has_final > 2) return ARMNET_ERR_BAD_ARG;
    a.B = B; a.K0 = K0; a.n_hidden = n_hidden; a.has_final = has_final; a.N = nhid; a.ldx = ldx; a.ldo = ldo;
    a.x = x; a.packed = static_cast<const uint8_t*>(packed); a.out = out;
#ifdef ARMNET_DEV_FLAGS
    if (const char* e = getenv("ARMNET_MLP_DBG")) a.dbg = atoi(e);
#endif
    switch (mlp_nt_for(nhid)) {
        case 1: return launch_mlp<1>(a, (hipStream_t)stream);
        case 2: return launch_mlp<2>(a, (hipStream_t)stream);
        case 4: return launch_mlp<4>(a, (hipStream_t)stream);
        case 8: return launch_mlp<8>(a, (hipStream_t)stream);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

// A plain Linear on the same kernel (round 5): out[b, n] = bias[n] + sum_k x[b, k] W[n, k], N <= 256 outputs, no
// BatchNorm fold, no ReLU.  `packed` = armnet_mlp_pack_layer_f32(K, N, 1, slot 0, W [N, K], K, bias | NULL, no BatchNorm).
int armnet_linear_bf16x3_f32(int64_t B, int K, int N, const float* x, int64_t ldx, const void* packed, float* out,
                             int64_t ldo, void* stream) {
    if (B < 0 || !armnet_mlp_head_supported(K, N, 1) || ldx < (int64_t)((K + 15) / 16) * 16 || ldo < N) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!x || !packed || !out) return ARMNET_ERR_BAD_ARG;
    MlpArgs a{};
    a.B = B; a.K0 = K; a.n_hidden = 1; a.has_final = 0; a.N = N; a.ldx = ldx; a.ldo = ldo;
    a.x = x; a.packed = static_cast<const uint8_t*>(packed); a.out = out; a.linear = 1;
    switch (mlp_nt_for(N)) {
        case 1: return launch_mlp<1>(a, (hipStream_t)stream);
        case 2: return launch_mlp<2>(a, (hipStream_t)stream);
        case 4: return launch_mlp<4>(a, (hipStream_t)stream);
        case 8: return launch_mlp<8>(a, (hipStream_t)stream);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
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
