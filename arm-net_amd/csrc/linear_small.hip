// linear_small.hip — the two head shapes armnet_mlp_head_f32 has no launch for (round-3 verdict, missing 5):
//   * nlayers == 0: the MLP degenerates to ONE Linear(ninput, noutput)                      (models/layers.py:79-80)
//   * noutput > 1:  the final Linear(nhid, noutput) behind the hidden layers                 (models/layers.py:86-87)
// Both are y[b, n] = bias[n] + sum_k x[b, k] * W[n, k] with a handful of outputs: reading x once is all there is to it
// (HBM-bound: 4 K bytes per row), so this is a plain fp32 kernel — one WAVE per row, lanes stride the row in 16-byte
// chunks, up to 16 outputs accumulate per lane and meet in a wave reduction.  `accumulate` adds to `out` (ensemble tail).
#include "armnet_common.h"

namespace armnet {

constexpr int LS_MAX_N = 16;

template <int N, bool VEC>
__global__ void __launch_bounds__(256)
linear_small_kernel(int64_t B, int K, const float* __restrict__ x, int64_t ldx, const float* __restrict__ W,
                    const float* __restrict__ bias, float scale, float* __restrict__ out, int64_t ldo, int accumulate) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    for (int64_t b = wave0; b < B; b += nwaves) {
        const float* xr = x + b * ldx;
        float acc[N];
#pragma unroll
        for (int n = 0; n < N; ++n) acc[n] = 0.f;
        if constexpr (VEC) {
            for (int k = 4 * lane; k < K; k += 256) {
                const float4 xv = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
                for (int n = 0; n < N; ++n) {
                    const float4 wv = *reinterpret_cast<const float4*>(W + (size_t)n * K + k);
                    acc[n] = fmaf(xv.x, wv.x, fmaf(xv.y, wv.y, fmaf(xv.z, wv.z, fmaf(xv.w, wv.w, acc[n]))));
                }
            }
        } else {
            for (int k = lane; k < K; k += 64) {
                const float xv = xr[k];
#pragma unroll
                for (int n = 0; n < N; ++n) acc[n] = fmaf(xv, W[(size_t)n * K + k], acc[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < N; ++n) {
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) acc[n] += __shfl_xor(acc[n], d);
        }
        if (lane == 0) {
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const float v = (acc[n] + (bias ? bias[n] : 0.f)) * scale;
                float* o = out + b * ldo + n;
                *o = accumulate ? *o + v : v;
            }
        }
    }
}

template <int N>
static int launch_ls(int64_t B, int K, const float* x, int64_t ldx, const float* W, const float* bias, float scale,
                     float* out, int64_t ldo, int accumulate, hipStream_t st) {
    const bool vec = K % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x % 16 == 0) && ((uintptr_t)W % 16 == 0);
    int64_t grid = (B + 3) / 4;
    const int64_t cap = (int64_t)device_cu_count() * 32;
    if (grid > cap) grid = cap;
    if (vec) linear_small_kernel<N, true><<<(int)grid, 256, 0, st>>>(B, K, x, ldx, W, bias, scale, out, ldo, accumulate);
    else linear_small_kernel<N, false><<<(int)grid, 256, 0, st>>>(B, K, x, ldx, W, bias, scale, out, ldo, accumulate);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

}  // namespace armnet

using namespace armnet;

extern "C" int armnet_linear_small_f32(int64_t B, int K, int N, const float* x, int64_t ldx, const float* W,
                                       const float* bias, float scale, float* out, int64_t ldo, int accumulate,
                                       void* stream) {
    if (B < 0 || K < 1 || N < 1 || ldx < K || ldo < N) return ARMNET_ERR_BAD_ARG;
    if (N > LS_MAX_N) return ARMNET_ERR_UNSUPPORTED;
    if (B == 0) return ARMNET_OK;
    if (!x || !W || !out) return ARMNET_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (N) {
#define LS_CASE(n) case n: return launch_ls<n>(B, K, x, ldx, W, bias, scale, out, ldo, accumulate, st);
        LS_CASE(1) LS_CASE(2) LS_CASE(3) LS_CASE(4) LS_CASE(5) LS_CASE(6) LS_CASE(7) LS_CASE(8)
        LS_CASE(9) LS_CASE(10) LS_CASE(11) LS_CASE(12) LS_CASE(13) LS_CASE(14) LS_CASE(15) LS_CASE(16)
#undef LS_CASE
    }
    return ARMNET_ERR_UNSUPPORTED;
}
