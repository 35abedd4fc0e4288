// mfma_bf16_chain.hip — issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 as a function of (i) how many independent
// accumulators alternate (dependent-accumulator latency), (ii) VALU fillers per MFMA, (iii) waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_chain.hip -o mfma_bf16_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int NVALU>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.f, v3 = 3.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8 / NACC; ++rep)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
                if (NVALU >= 1) v0 = fmaf(v0, 1.0001f, 0.5f);
                if (NVALU >= 2) v1 = fmaf(v1, 1.0001f, 0.5f);
                if (NVALU >= 3) v2 = fmaf(v2, 1.0001f, 0.5f);
                if (NVALU >= 4) v3 = fmaf(v3, 1.0001f, 0.5f);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int NVALU>
void run(int waves_per_simd, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    // 256 CUs x waves_per_simd blocks of 256 threads (one wave per SIMD per block)
    k<NACC, NVALU><<<256 * waves_per_simd, 256>>>(out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("accumulators %d, VALU/MFMA %d, waves/SIMD %d: %.1f cycles per MFMA per wave (s_memtime ticks at 100 MHz-independent shader clock)\n",
           NACC, NVALU, waves_per_simd, (double)c / (iters * 8.0));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 8);
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>(w, out, cyc); run<2, 0>(w, out, cyc); run<4, 0>(w, out, cyc); run<8, 0>(w, out, cyc);
        run<2, 2>(w, out, cyc); run<2, 4>(w, out, cyc); run<4, 2>(w, out, cyc); run<4, 4>(w, out, cyc);
    }
    return 0;
}
