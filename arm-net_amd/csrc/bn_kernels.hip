// bn_kernels.hip — training-mode BatchNorm1d (forward statistics / apply, backward reductions / apply), gfx950.
//
// The reference normalises the exponential neurons with nn.BatchNorm1d(nhid) on a [B, nhid, nemb] tensor
// (models/armnet_1h.py:65,85; models/armnet.py:67,88-89) and every MLP layer with nn.BatchNorm1d(nhid) on
// [B, nhid] (models/layers.py:77).  In training these are batch-statistics layers: two HBM-bound reductions
// and two HBM-bound elementwise passes per layer and step.  All four are one pass over x viewed as
// [N, C, L] (L = 1 for 2-D input) with the same skeleton:
//
//   a thread owns ONE inner position i = c * L + l (coalesced across the wave), so its channel parameters are
//   loaded once; a block owns 256 positions and a contiguous range of the N rows, two batches of 8 rows in flight
//   per thread (the passes are latency-bound otherwise); reductions go thread -> LDS (one atomic per thread) ->
//   one global atomic per channel and block.
//
// Variance uses sums shifted by k[c] = x[0, c, 0] (the exponential neurons have |mean| >> std: the plain
// E[x^2] - E[x]^2 would cancel).  Optional fused ReLU (the MLP's BatchNorm1d -> ReLU, layers.py:77-78):
// forward clamps, backward masks dy with (x * relu_scale[c] + relu_shift[c] > 0), recomputed from x.
#include "armnet_common.h"

namespace armnet {

constexpr int BN_TPB = 256;
constexpr int BN_UNROLL = 8;

enum BnOp { BN_STATS = 0, BN_APPLY = 1, BN_BWD_REDUCE = 2, BN_BWD_APPLY = 3 };

struct BnArgs {
    int64_t N;
    int C, L;
    const float* x;
    const float* dy;       // backward ops
    const float* p0;       // apply: scale | bwd_reduce: mean | bwd_apply: coefA
    const float* p1;       // apply: shift | bwd_reduce: rstd | bwd_apply: coefB
    const float* p2;       //                                 bwd_apply: coefC
    const float* rs;       // relu_scale (backward ops; null = no ReLU)
    const float* rt;       // relu_shift
    float* out;            // apply: y | bwd_apply: dx
    float* sums;           // stats / bwd_reduce: [2C] accumulators (+=)
    int relu;              // apply: clamp at 0
    int rows_per_block;
};

template <int OP>
__global__ void __launch_bounds__(BN_TPB) bn_pass_kernel(BnArgs a) {
    extern __shared__ float acc[];                       // [2C] for the reducing ops
    const int CL = a.C * a.L;
    constexpr bool REDUCE = (OP == BN_STATS || OP == BN_BWD_REDUCE);
    constexpr bool HAS_DY = (OP == BN_BWD_REDUCE || OP == BN_BWD_APPLY);
    constexpr bool WRITES = (OP == BN_APPLY || OP == BN_BWD_APPLY);
    if constexpr (REDUCE) {
        for (int i = threadIdx.x; i < 2 * a.C; i += BN_TPB) acc[i] = 0.f;
        __syncthreads();
    }
    // blockIdx.y: which 256 inner positions (a thread owns ONE position i = c * L + l); blockIdx.x: which rows
    const int i = blockIdx.y * BN_TPB + threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t n1 = n0 + a.rows_per_block;
    if (n1 > a.N) n1 = a.N;
    if (i < CL) {
        const int c = i / a.L;
        float q0 = 0.f, q1 = 0.f, q2 = 0.f, r0 = 0.f, r1 = 0.f;
        bool mask = false;
        if constexpr (OP == BN_STATS) q0 = a.x[(size_t)c * a.L];                     // shift k[c] = x[0, c, 0]
        if constexpr (OP == BN_APPLY) { q0 = a.p0[c]; q1 = a.p1[c]; }
        if constexpr (OP == BN_BWD_REDUCE) { q0 = a.p0[c]; q1 = a.p1[c]; }
        if constexpr (OP == BN_BWD_APPLY) { q0 = a.p0[c]; q1 = a.p1[c]; q2 = a.p2[c]; }
        if constexpr (HAS_DY) {
            mask = a.rs != nullptr;
            if (mask) { r0 = a.rs[c]; r1 = a.rt[c]; }
        }
        float s1 = 0.f, s2 = 0.f;
        const float* xp = a.x + (size_t)n0 * CL + i;
        const float* gp = HAS_DY ? a.dy + (size_t)n0 * CL + i : nullptr;
        float* op = WRITES ? a.out + (size_t)n0 * CL + i : nullptr;
        const bool relu = a.relu != 0;

        auto one = [&](float xv, float gv, float* dst) {
            if constexpr (OP == BN_STATS) {
                const float d = xv - q0;
                s1 += d;
                s2 = fmaf(d, d, s2);
            } else if constexpr (OP == BN_APPLY) {
                float y = fmaf(xv, q0, q1);
                if (relu) y = y > 0.f ? y : (y != y ? y : 0.f);                     // NaN stays NaN like torch.relu
                *dst = y;
            } else {
                float g = gv;
                if (mask) g = fmaf(xv, r0, r1) > 0.f ? g : 0.f;
                if constexpr (OP == BN_BWD_REDUCE) {
                    s1 += g;
                    s2 = fmaf(g, (xv - q0) * q1, s2);                               // dy * xhat
                } else {
                    *dst = fmaf(q0, g, fmaf(q2, xv, q1));
                }
            }
        };
        auto load = [&](float (&xv)[BN_UNROLL], float (&gv)[BN_UNROLL], int64_t row) {
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                xv[u] = xp[(size_t)(row + u) * CL];
                if constexpr (HAS_DY) gv[u] = gp[(size_t)(row + u) * CL];
            }
        };
        auto consume = [&](const float (&xv)[BN_UNROLL], const float (&gv)[BN_UNROLL], int64_t row) {
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) one(xv[u], HAS_DY ? gv[u] : 0.f, WRITES ? op + (size_t)(row + u) * CL : nullptr);
        };
        // two batches of BN_UNROLL rows in flight (ping-pong registers): the pass is latency-bound otherwise
        const int64_t rows = n1 - n0;
        const int64_t nb = rows / BN_UNROLL;
        float xa[BN_UNROLL], ga[BN_UNROLL], xb[BN_UNROLL], gb[BN_UNROLL];
        if (nb > 0) load(xa, ga, 0);
        int64_t b = 0;
        for (; b + 2 <= nb; b += 2) {
            load(xb, gb, (b + 1) * BN_UNROLL);
            consume(xa, ga, b * BN_UNROLL);
            if (b + 2 < nb) load(xa, ga, (b + 2) * BN_UNROLL);
            consume(xb, gb, (b + 1) * BN_UNROLL);
        }
        if (b < nb) consume(xa, ga, b * BN_UNROLL);
        for (int64_t r = nb * BN_UNROLL; r < rows; ++r)
            one(xp[(size_t)r * CL], HAS_DY ? gp[(size_t)r * CL] : 0.f, WRITES ? op + (size_t)r * CL : nullptr);
        if constexpr (REDUCE) {
            atomicAdd(&acc[c], s1);
            atomicAdd(&acc[a.C + c], s2);
        }
    }
    if constexpr (REDUCE) {
        __syncthreads();
        // this block saw only the channels of its 256 positions
        const int c_lo = (blockIdx.y * BN_TPB) / a.L;
        int c_hi = (blockIdx.y * BN_TPB + BN_TPB - 1) / a.L;
        if (c_hi >= a.C) c_hi = a.C - 1;
        for (int c = c_lo + threadIdx.x; c <= c_hi; c += BN_TPB) {
            unsafeAtomicAdd(a.sums + c, acc[c]);
            unsafeAtomicAdd(a.sums + a.C + c, acc[a.C + c]);
        }
    }
}

template <int OP>
static int launch_bn_pass(BnArgs a, hipStream_t st) {
    if (a.N == 0) return ARMNET_OK;
    const int64_t CL = (int64_t)a.C * a.L;
    const int ny = (int)((CL + BN_TPB - 1) / BN_TPB);
    if (ny > 65535) return ARMNET_ERR_UNSUPPORTED;
    // ~8 blocks per CU in total; at least 2 * BN_UNROLL rows per block so the pipelined loop is the common path
    int64_t nx = 2048 / ny;
    if (nx < 1) nx = 1;
    int64_t rpb = (a.N + nx - 1) / nx;
    if (rpb < 2 * BN_UNROLL) rpb = 2 * BN_UNROLL;
    a.rows_per_block = (int)rpb;
    nx = (a.N + rpb - 1) / rpb;
    const size_t lds = (OP == BN_STATS || OP == BN_BWD_REDUCE) ? (size_t)2 * a.C * sizeof(float) : 0;
    if (lds > 64 * 1024) return ARMNET_ERR_UNSUPPORTED;
    bn_pass_kernel<OP><<<dim3((unsigned)nx, (unsigned)ny), BN_TPB, lds, st>>>(a);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// per-channel epilogue of the forward statistics (torch.nn.functional.batch_norm, training = True)
__global__ void bn_finalize_kernel(int C, float count, const float* stats, const float* x, int L,
                                   const float* weight, const float* bias, float eps, float momentum,
                                   float* running_mean, float* running_var, float* mean, float* rstd,
                                   float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float k = x[(size_t)c * L];
    const float m1 = stats[c] / count;
    float var = stats[C + c] / count - m1 * m1;          // biased (what normalises the batch)
    var = var > 0.f ? var : (var != var ? var : 0.f);
    const float mu = k + m1;
    const float r = 1.0f / sqrtf(var + eps);
    const float w = weight ? weight[c] : 1.0f, b = bias ? bias[c] : 0.f;
    mean[c] = mu;
    rstd[c] = r;
    scale[c] = w * r;
    shift[c] = b - mu * (w * r);
    if (running_mean) running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * mu;
    if (running_var) {
        const float unbiased = count > 1.0f ? var * (count / (count - 1.0f)) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * unbiased;
    }
}

// per-channel epilogue of the backward reductions:  dx = A * dy + Cc * x + Bc
//   dbeta = sum dy, dgamma = sum dy * xhat;  dx = gamma * rstd * (dy - mean(dy) - xhat * mean(dy * xhat))
__global__ void bn_bwd_coef_kernel(int C, float count, const float* sums, const float* weight, const float* mean,
                                   const float* rstd, float* d_weight, float* d_bias, float* coefA, float* coefB,
                                   float* coefC) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float db = sums[c], dg = sums[C + c];
    if (d_bias) d_bias[c] = db;
    if (d_weight) d_weight[c] = dg;
    const float w = weight ? weight[c] : 1.0f;
    const float A = w * rstd[c];
    const float Cc = -A * rstd[c] * (dg / count);
    coefA[c] = A;
    coefC[c] = Cc;
    coefB[c] = -A * (db / count) - Cc * mean[c];
}

}  // namespace armnet

using namespace armnet;

static bool bn_shape_ok(int64_t N, int C, int L) { return N >= 0 && C > 0 && L > 0 && (int64_t)C * L < ((int64_t)1 << 30); }

extern "C" int armnet_bn_stats_f32(int64_t N, int C, int L, const float* x, float* stats, void* stream) {
    if (!bn_shape_ok(N, C, L) || !x || !stats) return ARMNET_ERR_BAD_ARG;
    BnArgs a{};
    a.N = N; a.C = C; a.L = L; a.x = x; a.sums = stats;
    return launch_bn_pass<BN_STATS>(a, (hipStream_t)stream);
}

extern "C" int armnet_bn_finalize_f32(int C, int64_t count, const float* stats, const float* x, int L,
                                      const float* weight, const float* bias, float eps, float momentum,
                                      float* running_mean, float* running_var, float* mean, float* rstd,
                                      float* scale, float* shift, void* stream) {
    if (C <= 0 || count <= 0 || L <= 0 || !stats || !x || !mean || !rstd || !scale || !shift) return ARMNET_ERR_BAD_ARG;
    bn_finalize_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, (float)count, stats, x, L, weight, bias, eps,
                                                                         momentum, running_mean, running_var, mean,
                                                                         rstd, scale, shift);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

extern "C" int armnet_bn_apply_f32(int64_t N, int C, int L, const float* x, const float* scale, const float* shift,
                                   int relu, float* y, void* stream) {
    if (!bn_shape_ok(N, C, L) || !x || !scale || !shift || !y) return ARMNET_ERR_BAD_ARG;
    BnArgs a{};
    a.N = N; a.C = C; a.L = L; a.x = x; a.p0 = scale; a.p1 = shift; a.relu = relu; a.out = y;
    return launch_bn_pass<BN_APPLY>(a, (hipStream_t)stream);
}

extern "C" int armnet_bn_bwd_reduce_f32(int64_t N, int C, int L, const float* x, const float* dy, const float* mean,
                                        const float* rstd, const float* relu_scale, const float* relu_shift,
                                        float* sums, void* stream) {
    if (!bn_shape_ok(N, C, L) || !x || !dy || !mean || !rstd || !sums) return ARMNET_ERR_BAD_ARG;
    if ((relu_scale == nullptr) != (relu_shift == nullptr)) return ARMNET_ERR_BAD_ARG;
    BnArgs a{};
    a.N = N; a.C = C; a.L = L; a.x = x; a.dy = dy; a.p0 = mean; a.p1 = rstd;
    a.rs = relu_scale; a.rt = relu_shift; a.sums = sums;
    return launch_bn_pass<BN_BWD_REDUCE>(a, (hipStream_t)stream);
}

extern "C" int armnet_bn_bwd_coef_f32(int C, int64_t count, const float* sums, const float* weight, const float* mean,
                                      const float* rstd, float* d_weight, float* d_bias, float* coefA, float* coefB,
                                      float* coefC, void* stream) {
    if (C <= 0 || count <= 0 || !sums || !mean || !rstd || !coefA || !coefB || !coefC) return ARMNET_ERR_BAD_ARG;
    bn_bwd_coef_kernel<<<(C + 255) / 256, 256, 0, (hipStream_t)stream>>>(C, (float)count, sums, weight, mean, rstd,
                                                                         d_weight, d_bias, coefA, coefB, coefC);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

extern "C" int armnet_bn_bwd_apply_f32(int64_t N, int C, int L, const float* x, const float* dy, const float* coefA,
                                       const float* coefB, const float* coefC, const float* relu_scale,
                                       const float* relu_shift, float* dx, void* stream) {
    if (!bn_shape_ok(N, C, L) || !x || !dy || !coefA || !coefB || !coefC || !dx) return ARMNET_ERR_BAD_ARG;
    if ((relu_scale == nullptr) != (relu_shift == nullptr)) return ARMNET_ERR_BAD_ARG;
    BnArgs a{};
    a.N = N; a.C = C; a.L = L; a.x = x; a.dy = dy; a.p0 = coefA; a.p1 = coefB; a.p2 = coefC;
    a.rs = relu_scale; a.rt = relu_shift; a.out = dx;
    return launch_bn_pass<BN_BWD_APPLY>(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------------------------------
// The tail of the sibling models' training backward (gc_arm.py:89 / afn.py:63 under autograd) in ONE pass: the embedding
// BatchNorm's dx = coefA[f] * dy + coefC[f] * t + coefB[f] on t = exp(x) (GC-ARM) or t = log(x) (AFN), the derivative of that
// map (exp(x) = t, or 1 / x = exp(-t)), the value scale of the lookup and the scatter-add into the table gradient
// (layers.py:20-21) — instead of armnet_bn_bwd_apply_f32 + an elementwise op + armnet_scatter_add_f32 (three passes over
// [B,F,E] and two temporaries).  One lane per element; rows are (sample, field) pairs, channel = row mod C.
namespace armnet {
template <typename IdT, int MAP>
__global__ void bn_bwd_scatter_kernel(int64_t n_rows, int C, int E, const IdT* __restrict__ ids,
                                      const float* __restrict__ vals, const float* __restrict__ t,
                                      const float* __restrict__ dy, const float* __restrict__ coefA,
                                      const float* __restrict__ coefB, const float* __restrict__ coefC, int64_t nfeat,
                                      float* __restrict__ d_table) {
    const int64_t total = n_rows * E, stride = (int64_t)gridDim.x * blockDim.x;
    constexpr int U = 4;                                 // elements per thread in flight (loads of U trips before the atomics)
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * U) {
        uint32_t id[U];
        float tv[U], gv[U], vv[U];
        int f[U], e[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i = i0 + u * stride;
            ok[u] = i < total;
            const int64_t ic = ok[u] ? i : total - 1;
            const int64_t r = ic / E;
            e[u] = (int)(ic - r * E);
            f[u] = (int)(r % C);
            bool bad;
            id[u] = load_id_checked(ids + r, nfeat, bad);
            ok[u] = ok[u] && !bad;                       // the forward already raised on a bad id
            tv[u] = t[ic];
            gv[u] = dy[ic];
            vv[u] = vals[r];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float dt = fmaf(coefA[f[u]], gv[u], fmaf(coefC[f[u]], tv[u], coefB[f[u]]));
            const float dx = MAP == 0 ? dt * tv[u] : dt * __expf(-tv[u]);
            if (ok[u]) unsafeAtomicAdd(d_table + (size_t)id[u] * E + e[u], dx * vv[u]);
        }
    }
}
}  // namespace armnet

extern "C" int armnet_bn_bwd_scatter_f32(int64_t n_rows, int C, int E, const void* ids, int id_type, const float* vals,
                                         const float* t, const float* dy, const float* coefA, const float* coefB,
                                         const float* coefC, int map, int64_t nfeat, float* d_table, void* stream) {
    if (n_rows < 0 || C <= 0 || E <= 0 || nfeat <= 0 || (map != 0 && map != 1)) return ARMNET_ERR_BAD_ARG;
    if (n_rows == 0) return ARMNET_OK;
    if (!ids || !vals || !t || !dy || !coefA || !coefB || !coefC || !d_table) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    int64_t grid = (n_rows * E + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    hipStream_t s = (hipStream_t)stream;
#define ARMNET_BBS(IdT, MAP) \
    bn_bwd_scatter_kernel<IdT, MAP><<<(int)grid, 256, 0, s>>>(n_rows, C, E, (const IdT*)ids, vals, t, dy, coefA, coefB, coefC, nfeat, d_table)
    if (id_type == ARMNET_ID_I64) { if (map == 0) ARMNET_BBS(int64_t, 0); else ARMNET_BBS(int64_t, 1); }
    else { if (map == 0) ARMNET_BBS(int32_t, 0); else ARMNET_BBS(int32_t, 1); }
#undef ARMNET_BBS
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The head of the sibling models' training forward in ONE pass (gc_arm.py:87-89 / afn.py:61-63): lookup * value, the
// map in front of the embedding BatchNorm (exp for GC-ARM, log for AFN) written to out [B,F,E], and that BatchNorm's
// shifted batch sums per field (armnet_bn_stats_f32's: shift k[f] = out[0, f, 0]) — instead of armnet_gather_scale_f32 +
// an elementwise op + armnet_bn_stats_f32 (three passes over [B,F,E]).  Skeleton of bn_pass_kernel: a thread owns one
// inner position (field, e) and a range of samples; 16 lanes share a table row (64-byte runs).
namespace armnet {
template <typename IdT, int MAP>
__global__ void __launch_bounds__(BN_TPB)
gather_map_stats_kernel(int64_t B, int F, int E, const IdT* __restrict__ ids, const float* __restrict__ vals,
                        const float* __restrict__ table, int64_t nfeat, float* __restrict__ out, float* __restrict__ sums,
                        int32_t* id_status, int rows_per_block) {
    extern __shared__ float acc[];                       // [2F]
    for (int i = threadIdx.x; i < 2 * F; i += BN_TPB) acc[i] = 0.f;
    __syncthreads();
    const int CL = F * E;
    const int i = blockIdx.y * BN_TPB + threadIdx.x;
    const int64_t n0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t n1 = n0 + rows_per_block;
    if (n1 > B) n1 = B;
    auto map = [](float x) -> float { return MAP == 0 ? expf(x) : logf(x); };
    if (i < CL) {
        const int f = i / E, e = i - f * E;
        bool bad0;
        const uint32_t id0 = load_id_checked(ids + f, nfeat, bad0);
        const float pivot = map(table[(size_t)id0 * E] * vals[f]);                   // = out[0, f, 0]
        float s1 = 0.f, s2 = 0.f;
        bool any_bad = false;
        constexpr int U = 8;
        for (int64_t n = n0; n < n1; n += U) {
            uint32_t id[U];
            float v[U], t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t r = (n + u < n1 ? n + u : n1 - 1) * F + f;
                bool bad;
                id[u] = load_id_checked(ids + r, nfeat, bad);
                any_bad |= bad;
                v[u] = vals[r];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) t[u] = table[(size_t)id[u] * E + e];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (n + u < n1) {
                    const float xv = map(t[u] * v[u]);
                    out[(size_t)(n + u) * CL + i] = xv;
                    const float d = xv - pivot;
                    s1 += d;
                    s2 = fmaf(d, d, s2);
                }
            }
        }
        if (any_bad && id_status) flag_bad_id(id_status);
        atomicAdd(&acc[f], s1);
        atomicAdd(&acc[F + f], s2);
    }
    __syncthreads();
    const int c_lo = (blockIdx.y * BN_TPB) / E;
    int c_hi = (blockIdx.y * BN_TPB + BN_TPB - 1) / E;
    if (c_hi >= F) c_hi = F - 1;
    for (int c = c_lo + threadIdx.x; c <= c_hi; c += BN_TPB) {
        unsafeAtomicAdd(sums + c, acc[c]);
        unsafeAtomicAdd(sums + F + c, acc[F + c]);
    }
}
}  // namespace armnet

extern "C" int armnet_gather_map_stats_f32(int64_t B, int F, int E, const void* ids, int id_type, const float* vals,
                                           const float* table, int64_t nfeat, int map, float* out, float* stats,
                                           int32_t* id_status, void* stream) {
    if (B < 0 || F <= 0 || E <= 0 || nfeat <= 0 || (map != 0 && map != 1)) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!ids || !vals || !table || !out || !stats) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31) || (int64_t)F * E >= ((int64_t)1 << 24)) return ARMNET_ERR_UNSUPPORTED;
    const int CL = F * E, ny = (CL + BN_TPB - 1) / BN_TPB;
    int64_t nx = 2048 / ny;
    if (nx < 1) nx = 1;
    int64_t rpb = (B + nx - 1) / nx;
    if (rpb < 16) rpb = 16;
    nx = (B + rpb - 1) / rpb;
    const size_t lds = (size_t)2 * F * sizeof(float);
    if (lds > 64 * 1024) return ARMNET_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)nx, (unsigned)ny);
#define ARMNET_GMS(IdT, MAP) \
    gather_map_stats_kernel<IdT, MAP><<<grid, BN_TPB, lds, s>>>(B, F, E, (const IdT*)ids, vals, table, nfeat, out, stats, id_status, (int)rpb)
    if (id_type == ARMNET_ID_I64) { if (map == 0) ARMNET_GMS(int64_t, 0); else ARMNET_GMS(int64_t, 1); }
    else { if (map == 0) ARMNET_GMS(int32_t, 0); else ARMNET_GMS(int32_t, 1); }
#undef ARMNET_GMS
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}
