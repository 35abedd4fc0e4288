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
// One wave owns 32 samples through all layers; a 256-thread block = 4 waves = 128 samples.  The packed weights are one
// linear stream of 1-KiB lane-ready blocks in consumption order; the block's waves pull it stage by stage
// (one k-step of all its tiles) through a 2-deep LDS ring with global_load_lds_dwordx4, one barrier per stage.
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
__host__ __device__ constexpr int mlp_kps(int NT) { return NT >= 8 ? 2 : 1; }

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
    int64_t ldx;
    const float* x;
    const uint8_t* packed;
    float* out;
};

// 8 fp32 -> three packed bf16x8 planes; h + m + l == x exactly (truncating 8-bit slices of the significand)
__device__ __forceinline__ void split3(const float (&x)[8], u32x4& ph, u32x4& pm, u32x4& pl) {
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

// the six significant cross products of one k-step for NTL tiles; tiles interleaved so that consecutive MFMAs never
// wait on the same accumulator
template <int NTL>
__device__ __forceinline__ void step_mfma(f32x16* acc, const u32x4* st, int lane, u32x4 bh, u32x4 bm, u32x4 bl) {
    constexpr int PAIR = NTL >= 2 ? 2 : 1;
#pragma unroll
    for (int t = 0; t < NTL; t += PAIR) {
        u32x4 ah[PAIR], am[PAIR], al[PAIR];
#pragma unroll
        for (int u = 0; u < PAIR; ++u) {
            ah[u] = st[((t + u) * 3 + 0) * 64 + lane];
            am[u] = st[((t + u) * 3 + 1) * 64 + lane];
            al[u] = st[((t + u) * 3 + 2) * 64 + lane];
        }
        // small terms first, the leading product last
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[t + u] = mfma_bf16(al[u], bh, acc[t + u]);
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[t + u] = mfma_bf16(ah[u], bl, acc[t + u]);
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[t + u] = mfma_bf16(am[u], bm, acc[t + u]);
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[t + u] = mfma_bf16(am[u], bh, acc[t + u]);
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[t + u] = mfma_bf16(ah[u], bm, acc[t + u]);
#pragma unroll
        for (int u = 0; u < PAIR; ++u) acc[t + u] = mfma_bf16(ah[u], bh, acc[t + u]);
    }
}

template <int NT>
__global__ void __launch_bounds__(256, 2) mlp_head_kernel(MlpArgs a) {
    constexpr int TG = mlp_tg(NT), NG = NT / TG, KPS = mlp_kps(NT), NS2 = 2 * NT / KPS;
    constexpr int ST1 = NT * 3 * 1024, ST2 = KPS * TG * 3 * 1024;
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];      // [2][ST1] ring | tables
    float* tabs = reinterpret_cast<float*>(lds + 2 * ST1);              // bias1 | bias2 | wlast | blast
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, hf = lane >> 5;
    const MlpLayout L = mlp_layout(a.K0, NT, a.n_hidden);
    const int KS1 = L.KS1;
    const int Q = KS1 + (a.n_hidden >= 2 ? NG * NS2 : 0);              // stages of the whole stream

    // stage q of the stream -> ring slot q & 1 (1-KiB lane-linear blocks, round-robin over the 4 waves)
    auto issue = [&](int q) {
        const bool l1 = q < KS1;
        const uint8_t* src = a.packed + (l1 ? (int64_t)q * ST1 : L.l2_off + (int64_t)(q - KS1) * ST2);
        const int nblk = l1 ? NT * 3 : KPS * TG * 3;
        uint8_t* dst = lds + (q & 1) * ST1;
        for (int blk = wave; blk < nblk; blk += 4)
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src + blk * 1024 + lane * 16),
                (__attribute__((address_space(3))) void*)(dst + blk * 1024), 16, 0, 0);
    };
    int q = 0;
    // make stage q consumable (its loads were issued one stage ago), then start stage q+1 into the slot every wave
    // has finished reading
    auto advance = [&]() -> const u32x4* {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (q + 1 < Q) issue(q + 1);
        const u32x4* st = reinterpret_cast<const u32x4*>(lds + (q & 1) * ST1);
        ++q;
        return st;
    };

    issue(0);
    {   // tables -> LDS
        const float* src = reinterpret_cast<const float*>(a.packed + L.tab_off);
        for (int i = threadIdx.x; i < 3 * NT * 32 + 4; i += 256) tabs[i] = src[i];
    }
    const int64_t row = (int64_t)blockIdx.x * 128 + wave * 32 + m;
    const int64_t rowc = row < a.B ? row : a.B - 1;
    const float* xr = a.x + rowc * a.ldx + 8 * hf;
    const bool k_tail = (a.K0 & 15) != 0;

    auto load_x = [&](int s, float (&v)[8]) {
        if (!k_tail || s + 1 < KS1) {
            const f32x4m lo = *reinterpret_cast<const f32x4mu*>(xr + 16 * s);
            const f32x4m hi = *reinterpret_cast<const f32x4mu*>(xr + 16 * s + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = lo[i]; v[4 + i] = hi[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (16 * s + 8 * hf + i) < a.K0 ? xr[16 * s + i] : 0.f;
        }
    };

    // ---- layer 1: K0 -> NT*32 hidden units --------------------------------------------------------------------
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float xn[8];
    load_x(0, xn);
    for (int s = 0; s < KS1; ++s) {
        const u32x4* st = advance();
        float xc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) xc[i] = xn[i];
        if (s + 1 < KS1) load_x(s + 1, xn);
        u32x4 bh, bm, bl;
        split3(xc, bh, bm, bl);
        step_mfma<NT>(acc, st, lane, bh, bm, bl);
    }
    // bias (BatchNorm folded) + ReLU, in place: acc becomes H1 in C layout
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = fmaxf(acc[t][r] + tabs[(hf * NT + t) * 16 + r], 0.f);

    float part = 0.f;
    const float* wl = tabs + 2 * NT * 32;
    auto store_hidden = [&](const f32x16& h, int t) {
        if (row < a.B) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = c_layout_unit(t, r, hf);
                if (n < a.N) a.out[row * (int64_t)a.N + n] = h[r];
            }
        }
    };
    if (a.n_hidden >= 2) {
        // ---- layer 2: the previous layer's accumulators ARE this layer's B operands ----------------------------
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x16 acc2[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
#pragma unroll
            for (int sg = 0; sg < NS2; ++sg) {
                const u32x4* st = advance();
#pragma unroll
                for (int kk = 0; kk < KPS; ++kk) {
                                    const int s2 = sg * KPS + kk;                        // k-step: registers 8u..8u+7 of tile s2 >> 1
                    float xc[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) xc[j] = acc[s2 >> 1][8 * (s2 & 1) + j];
                    u32x4 bh, bm, bl;
                    split3(xc, bh, bm, bl);
                    step_mfma<TG>(acc2, st + kk * TG * 3 * 64, lane, bh, bm, bl);
                }
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
        if (hf == 0 && row < a.B) a.out[row] = part + tabs[3 * NT * 32];
    }
}

template <int NT>
static int launch_mlp(const MlpArgs& a, hipStream_t st) {
    static_assert(mlp_kps(NT) * mlp_tg(NT) <= NT, "a layer-2 stage must fit a ring slot");
    const size_t lds = (size_t)2 * NT * 3 * 1024 + ((size_t)3 * NT * 32 + 4) * sizeof(float);
    auto kern = mlp_head_kernel<NT>;
    if (lds > 64 * 1024)
        ARMNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t blocks = (a.B + 127) / 128;
    kern<<<(int)blocks, 256, lds, st>>>(a);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
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
                        const void* packed, float* out, void* stream) {
    if (B < 0 || !armnet_mlp_head_supported(K0, nhid, n_hidden) || ldx < K0) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!x || !packed || !out) return ARMNET_ERR_BAD_ARG;
    MlpArgs a{};
    a.B = B; a.K0 = K0; a.n_hidden = n_hidden; a.has_final = has_final ? 1 : 0; a.N = nhid; a.ldx = ldx;
    a.x = x; a.packed = static_cast<const uint8_t*>(packed); a.out = out;
    switch (mlp_nt_for(nhid)) {
        case 1: return launch_mlp<1>(a, (hipStream_t)stream);
        case 2: return launch_mlp<2>(a, (hipStream_t)stream);
        case 4: return launch_mlp<4>(a, (hipStream_t)stream);
        case 8: return launch_mlp<8>(a, (hipStream_t)stream);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // extern "C"
