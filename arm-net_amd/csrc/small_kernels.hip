// small_kernels.hip — parameter fold, gather*value, clamp, stand-alone sparse map.  gfx950.
#include "armnet_common.h"

namespace armnet {

// ---------------------------------------------------------------------------------------------
// Parameter fold (armnet_hip.h: armnet_fold_params_f32).  One thread per output element; the
// contraction runs in double so that the fold adds no rounding of its own beyond the final cast.
//   one-head   q_fold[o,e]     = D^-0.5 * sum_d query[o,d] * W[d,e]          (armnet_1h.py:30-32)
//   multi-head q_fold[k*H+o,e] = D^-0.5 * sum_y bw[k,e,y] * query[k,o,y]     (armnet.py:33-34)
__global__ void fold_params_kernel(int variant, int K, int H, int E, int D, const float* __restrict__ bw,
                                   const float* __restrict__ q, const float* __restrict__ bn_w,
                                   const float* __restrict__ bn_b, const float* __restrict__ bn_m,
                                   const float* __restrict__ bn_v, float eps, float* __restrict__ q_fold,
                                   float* __restrict__ bn_scale, float* __restrict__ bn_shift) {
    const int O = K * H;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < O * E) {
        const int row = i / E, e = i % E;
        const int k = row / H, o = row % H;
        double acc = 0.0;
        if (variant == ARMNET_ONE_HEAD) {
            for (int d = 0; d < D; ++d) acc += (double)q[o * D + d] * (double)bw[d * E + e];
        } else {
            for (int y = 0; y < D; ++y)
                acc += (double)bw[((size_t)k * E + e) * D + y] * (double)q[((size_t)k * H + o) * D + y];
        }
        // python: d_k ** -0.5, cast to fp32 by the tensor multiply; gc_arm.py:33-34 has no scale
        const float scale = variant == ARMNET_GC_ARM ? 1.0f : (float)(1.0 / sqrt((double)D));
        q_fold[i] = (float)(acc * (double)scale);
    }
    if (i < O) {
        // ATen's eval-mode transform: alpha = w * 1/sqrt(var + eps); beta = b - mean * alpha  (fp32)
        const float invstd = 1.0f / sqrtf(bn_v[i] + eps);
        const float a = bn_w[i] * invstd;
        bn_scale[i] = a;
        bn_shift[i] = bn_b[i] - bn_m[i] * a;
    }
}

// eval-mode BatchNorm1d as a per-channel affine (ATen's CPU transform form), for the field-wise emb_bn of GC-ARM / AFN
__global__ void fold_bn_kernel(int C, const float* __restrict__ w, const float* __restrict__ b,
                               const float* __restrict__ m, const float* __restrict__ v, float eps,
                               float* __restrict__ scale, float* __restrict__ shift) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < C) {
        const float a = w[i] * (1.0f / sqrtf(v[i] + eps));
        scale[i] = a;
        shift[i] = b[i] - m[i] * a;
    }
}

int launch_fold_bn(int C, const float* w, const float* b, const float* m, const float* v, float eps, float* scale,
                   float* shift, hipStream_t s) {
    fold_bn_kernel<<<(C + 255) / 256, 256, 0, s>>>(C, w, b, m, v, eps, scale, shift);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// afn.py:74-77 embedding_clip: weight.abs_().clamp_(min=1e-4), in place
__global__ void abs_clamp_min_kernel(float* __restrict__ p, int64_t n, float lo) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = fabsf(p[i]);
        p[i] = a < lo ? lo : a;                 // NaN stays NaN (torch.clamp)
    }
}

int launch_abs_clamp_min(float* p, int64_t n, float lo, hipStream_t s) {
    if (n == 0) return ARMNET_OK;
    int64_t grid = (n + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    abs_clamp_min_kernel<<<(int)grid, 256, 0, s>>>(p, n, lo);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

int launch_fold_params(int variant, int K, int H, int E, int D, const float* bw, const float* q,
                       const float* bn_w, const float* bn_b, const float* bn_m, const float* bn_v, float eps,
                       float* q_fold, float* bn_scale, float* bn_shift, hipStream_t s) {
    const int n = K * H * E;
    const int block = 256;
    fold_params_kernel<<<(n + block - 1) / block, block, 0, s>>>(variant, K, H, E, D, bw, q, bn_w, bn_b, bn_m,
                                                                  bn_v, eps, q_fold, bn_scale, bn_shift);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// ---------------------------------------------------------------------------------------------
// Embedding.forward alone (layers.py:20-21): out[r,:] = table[ids[r],:] * vals[r].
// One lane per 16-byte chunk when E % 4 == 0, per 8-byte chunk when E is even (adjacent lanes share a row -> one
// 4E-byte segment per row), scalar lanes otherwise.  The vector accesses need only 4-byte alignment (gfx950 runs in
// unaligned access mode: still one dwordx4 / dwordx2 per lane).
typedef float gs_f32x4 __attribute__((ext_vector_type(4)));
typedef float gs_f32x2 __attribute__((ext_vector_type(2)));
typedef gs_f32x4 gs_f32x4u __attribute__((aligned(4)));
typedef gs_f32x2 gs_f32x2u __attribute__((aligned(4)));

template <typename IdT, int VEC>
__global__ void gather_scale_kernel(int64_t n_rows, int E, const IdT* __restrict__ ids,
                                    const float* __restrict__ vals, const float* __restrict__ table,
                                    int64_t nfeat, float* __restrict__ out, int32_t* id_status) {
    const int cpr = E / VEC;  // chunks per row
    const int64_t total = n_rows * cpr;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cpr;
        const int c = (int)(i - r * cpr);
        bool bad;
        const uint32_t id = load_id_checked(ids + r, nfeat, bad);
        if (bad && id_status) flag_bad_id(id_status);
        const float v = vals ? vals[r] : 1.0f;
        const float* src = table + (size_t)id * E + c * VEC;
        float* dst = out + r * E + c * VEC;
        if constexpr (VEC == 4) {
            *reinterpret_cast<gs_f32x4u*>(dst) = *reinterpret_cast<const gs_f32x4u*>(src) * v;
        } else if constexpr (VEC == 2) {
            *reinterpret_cast<gs_f32x2u*>(dst) = *reinterpret_cast<const gs_f32x2u*>(src) * v;
        } else {
            dst[0] = src[0] * v;
        }
    }
}

int launch_gather_scale(int64_t n_rows, int E, const void* ids, int id_type, const float* vals,
                        const float* table, int64_t nfeat, float* out, int32_t* id_status, hipStream_t s) {
    if (n_rows == 0) return ARMNET_OK;
    const int vec = (E % 4 == 0) ? 4 : (E % 2 == 0) ? 2 : 1;
    const int64_t total = n_rows * (E / vec);
    const int block = 256;
    int64_t grid = (total + block - 1) / block;
    if (grid > 256 * 16) grid = 256 * 16;
#define GS(IdT, V)                                                                                          \
    gather_scale_kernel<IdT, V><<<(int)grid, block, 0, s>>>(n_rows, E, (const IdT*)ids, vals, table, nfeat, \
                                                            out, id_status)
    if (id_type == ARMNET_ID_I64) { if (vec == 4) GS(int64_t, 4); else if (vec == 2) GS(int64_t, 2); else GS(int64_t, 1); }
    else { if (vec == 4) GS(int32_t, 4); else if (vec == 2) GS(int32_t, 2); else GS(int32_t, 1); }
#undef GS
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// Backward of the lookup alone (layers.py:20-21): d_table[ids[r], :] += g[r, :] * vals[r], float atomics, one lane per
// element (adjacent lanes share a row: 4E-byte runs).
template <typename IdT>
__global__ void scatter_add_kernel(int64_t n_rows, int E, const IdT* __restrict__ ids, const float* __restrict__ vals,
                                   const float* __restrict__ g, int64_t nfeat, float* __restrict__ d_table) {
    const int64_t total = n_rows * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / E;
        const int e = (int)(i - r * E);
        bool bad;
        const uint32_t id = load_id_checked(ids + r, nfeat, bad);
        if (bad) continue;                               // the forward already raised on it
        const float v = vals ? vals[r] : 1.0f;
        unsafeAtomicAdd(d_table + (size_t)id * E + e, g[i] * v);
    }
}

int launch_scatter_add(int64_t n_rows, int E, const void* ids, int id_type, const float* vals, const float* g,
                       int64_t nfeat, float* d_table, hipStream_t s) {
    if (n_rows == 0) return ARMNET_OK;
    int64_t grid = (n_rows * E + 255) / 256;
    if (grid > 256 * 32) grid = 256 * 32;
    if (id_type == ARMNET_ID_I64)
        scatter_add_kernel<int64_t><<<(int)grid, 256, 0, s>>>(n_rows, E, (const int64_t*)ids, vals, g, nfeat, d_table);
    else
        scatter_add_kernel<int32_t><<<(int)grid, 256, 0, s>>>(n_rows, E, (const int32_t*)ids, vals, g, nfeat, d_table);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

__global__ void clamp_vals_kernel(float* vals, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        vals[i] = clamp_val(vals[i]);
}

int launch_clamp_vals(float* vals, int64_t n, hipStream_t s) {
    if (n == 0) return ARMNET_OK;
    int64_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    clamp_vals_kernel<<<(int)grid, 256, 0, s>>>(vals, n);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone sparse map over the last dim of [rows, d] (utils/entmax.py:134).
// A block owns TPB consecutive rows = one contiguous span of TPB*d floats: it is read coalesced,
// transposed through LDS so that thread r owns row r as the LDS column  lds[i*(TPB+1) + r]
// (odd stride: conflict-free both for the transposing writes and the per-thread column reads).
// alpha_rows != NULL (armnet_entmax_rows_f32): a per-ROW alpha (utils/entmax.py:31-36: an alpha tensor broadcast over
// every dimension but `dim`); each row runs the reference's bisection with its own alpha - 1, 1 / (alpha - 1) and
// (1 / d)^(alpha - 1) (entmax.py:42-47) — the Newton shortcuts are per-launch choices and stay with the scalar entry point
__device__ inline SparseMapCfg row_cfg(const SparseMapCfg& cfg, const float* alpha_rows, int64_t row, int d) {
    if (!alpha_rows) return cfg;
    SparseMapCfg c = cfg;
    c.mode = SOLVE_BISECT;
    c.am1 = alpha_rows[row] - 1.0f;
    c.r = 1.0f / c.am1;
    c.tau_hi_off = powf(1.0f / (float)d, c.am1);
    return c;
}

constexpr int EL_U = 8;
template <int TPB>
__global__ void entmax_lds_kernel(int64_t rows, int d, SparseMapCfg cfg, const float* __restrict__ alpha_rows,
                                  const float* __restrict__ X, float* __restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = TPB + 1;
    for (int64_t r0 = (int64_t)blockIdx.x * TPB; r0 < rows; r0 += (int64_t)gridDim.x * TPB) {
        const int nr = (int)((rows - r0) < TPB ? (rows - r0) : TPB);
        const int n = nr * d;
        const float* src = X + r0 * d;
        __syncthreads();
        // EL_U loads per thread in flight (a loop of one load, one LDS write leaves a CU with 8 KB in flight: 2.2 TB/s);
        // (row, column) of element k advance by TPB per step without a division
        const int qd = TPB / d, rd = TPB - qd * d;
        {
            int r = threadIdx.x / d, i = threadIdx.x - r * d;
            for (int k0 = threadIdx.x; k0 < n; k0 += TPB * EL_U) {
                float v[EL_U];
#pragma unroll
                for (int u = 0; u < EL_U; ++u) v[u] = (k0 + u * TPB < n) ? src[k0 + u * TPB] : 0.f;
#pragma unroll
                for (int u = 0; u < EL_U; ++u) {
                    if (k0 + u * TPB < n) lds[i * S + r] = v[u];
                    r += qd; i += rd;
                    if (i >= d) { i -= d; ++r; }
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nr) sparse_map_row(lds + threadIdx.x, S, d, row_cfg(cfg, alpha_rows, r0 + threadIdx.x, d));
        __syncthreads();
        float* dst = P + r0 * d;
        {
            int r = threadIdx.x / d, i = threadIdx.x - r * d;
            for (int k = threadIdx.x; k < n; k += TPB) {
                dst[k] = lds[i * S + r];
                r += qd; i += rd;
                if (i >= d) { i -= d; ++r; }
            }
        }
    }
}

// Fallback for very long rows: one thread per row, in place in global memory.
__global__ void entmax_global_kernel(int64_t rows, int d, SparseMapCfg cfg, const float* __restrict__ alpha_rows,
                                     const float* __restrict__ X, float* __restrict__ P) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows;
         r += (int64_t)gridDim.x * blockDim.x) {
        float* p = P + r * d;
        const float* x = X + r * d;
        for (int i = 0; i < d; ++i) p[i] = x[i];
        sparse_map_row(p, 1, d, row_cfg(cfg, alpha_rows, r, d));
    }
}

int launch_entmax(int64_t rows, int d, const SparseMapCfg& cfg, const float* alpha_rows, const float* X, float* P, hipStream_t s) {
    if (rows == 0) return ARMNET_OK;
    const size_t lds128 = (size_t)d * 129 * sizeof(float);
    const size_t lds64 = (size_t)d * 65 * sizeof(float);
    if (lds128 <= 64 * 1024) {
        int64_t grid = (rows + 127) / 128;
        if (grid > 256 * 8) grid = 256 * 8;
        entmax_lds_kernel<128><<<(int)grid, 128, lds128, s>>>(rows, d, cfg, alpha_rows, X, P);
    } else if (lds64 <= 64 * 1024) {
        int64_t grid = (rows + 63) / 64;
        if (grid > 256 * 8) grid = 256 * 8;
        entmax_lds_kernel<64><<<(int)grid, 64, lds64, s>>>(rows, d, cfg, alpha_rows, X, P);
    } else {
        int64_t grid = (rows + 63) / 64;
        if (grid > 4096) grid = 4096;
        entmax_global_kernel<<<(int)grid, 64, 0, s>>>(rows, d, cfg, alpha_rows, X, P);
    }
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// Backward of the stand-alone sparse map (utils/entmax.py:70-80), rows over the last dim:
//   gppr = Y > 0 ? Y^(2 - alpha) : 0;  dX = dY * gppr;  q = sum(dX) / sum(gppr);  dX -= q * gppr
// (alpha == 1: softmax, dX = Y * (dY - sum(Y * dY))).  Same transposing LDS staging as the forward: a block owns TPB
// consecutive rows, reads Y and dY coalesced, thread r owns row r as an LDS column.  Replaces six ATen passes
// (where, pow, two sums, two elementwise) over [B, neurons, fields] tensors in the sibling models' training step.
template <int TPB>
__global__ void entmax_bwd_lds_kernel(int64_t rows, int d, float alpha, const float* __restrict__ Y,
                                      const float* __restrict__ dY, float* __restrict__ dX) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int S = TPB + 1;
    float* ly = lds;
    float* lg = lds + (size_t)d * S;
    const float e = 2.0f - alpha;
    for (int64_t r0 = (int64_t)blockIdx.x * TPB; r0 < rows; r0 += (int64_t)gridDim.x * TPB) {
        const int nr = (int)((rows - r0) < TPB ? (rows - r0) : TPB);
        const int n = nr * d;
        __syncthreads();
        const int qd = TPB / d, rd = TPB - qd * d;
        {
            const float* sy = Y + r0 * d;
            const float* sg = dY + r0 * d;
            int r = threadIdx.x / d, i = threadIdx.x - r * d;
            constexpr int U = EL_U / 2;                            // two arrays: the same number of loads in flight
            for (int k0 = threadIdx.x; k0 < n; k0 += TPB * U) {
                float vy[U], vg[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = k0 + u * TPB < n;
                    vy[u] = in ? sy[k0 + u * TPB] : 0.f;
                    vg[u] = in ? sg[k0 + u * TPB] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (k0 + u * TPB < n) {
                        ly[i * S + r] = vy[u];
                        lg[i * S + r] = vg[u];
                    }
                    r += qd; i += rd;
                    if (i >= d) { i -= d; ++r; }
                }
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < nr) {
            float* y = ly + threadIdx.x;
            float* g = lg + threadIdx.x;
            if (alpha == 1.0f) {
                float dot = 0.f;
                for (int i = 0; i < d; ++i) dot += y[i * S] * g[i * S];
                for (int i = 0; i < d; ++i) g[i * S] = y[i * S] * (g[i * S] - dot);
            } else {
                float sx = 0.f, sg = 0.f;
                for (int i = 0; i < d; ++i) {
                    const float yy = y[i * S];
                    const float gp = yy > 0.f ? (alpha == 2.0f ? 1.0f : alpha == 1.5f ? sqrtf(yy) : pow_pos(yy, e)) : 0.f;   // hardware log2 / exp2 (~2 ulp)
                    const float v = g[i * S] * gp;
                    y[i * S] = gp;
                    g[i * S] = v;
                    sx += v;
                    sg += gp;
                }
                const float q = sx / sg;
                for (int i = 0; i < d; ++i) g[i * S] -= q * y[i * S];
            }
        }
        __syncthreads();
        {
            float* dst = dX + r0 * d;
            int r = threadIdx.x / d, i = threadIdx.x - r * d;
            for (int k = threadIdx.x; k < n; k += TPB) {
                dst[k] = lg[i * S + r];
                r += qd; i += rd;
                if (i >= d) { i -= d; ++r; }
            }
        }
    }
}

int launch_entmax_bwd(int64_t rows, int d, float alpha, const float* Y, const float* dY, float* dX, hipStream_t s) {
    if (rows == 0) return ARMNET_OK;
    const size_t lds128 = (size_t)2 * d * 129 * sizeof(float), lds64 = (size_t)2 * d * 65 * sizeof(float);
    if (lds128 <= 64 * 1024) {
        int64_t grid = (rows + 127) / 128;
        if (grid > 256 * 8) grid = 256 * 8;
        entmax_bwd_lds_kernel<128><<<(int)grid, 128, lds128, s>>>(rows, d, alpha, Y, dY, dX);
    } else if (lds64 <= 64 * 1024) {
        int64_t grid = (rows + 63) / 64;
        if (grid > 256 * 8) grid = 256 * 8;
        entmax_bwd_lds_kernel<64><<<(int)grid, 64, lds64, s>>>(rows, d, alpha, Y, dY, dX);
    } else {
        return ARMNET_ERR_UNSUPPORTED;
    }
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

}  // namespace armnet
