// mfma_bf16_chain.hip — issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 as a function of (i) how many independent
// accumulators alternate (dependent-accumulator latency), (ii) VALU fillers per MFMA, (iii) LDS reads per MFMA,
// (iv) one or two waves per SIMD (256- / 512-thread blocks, one block per CU: the placement is then certain).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_chain.hip -o mfma_bf16_chain ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU, int LDSREAD, int TPB>
__global__ void __launch_bounds__(TPB) k(float* out, unsigned long long* cyc, int iters) {
    __shared__ u32x4 buf[1024];
    for (int i = threadIdx.x; i < 1024; i += TPB) buf[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    __syncthreads();
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    float v0 = threadIdx.x, v1 = 1.0f, v2 = 2.f, v3 = 3.f;
    u32x4 l = {0, 0, 0, 0};
    const int lane = threadIdx.x & 63;
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
                if (LDSREAD && ((rep * NACC + i) % LDSREAD) == 0) {
                    const u32x4 t = buf[(lane + 64 * ((it + i) & 15)) & 1023];
                    l[0] ^= t[0]; l[1] ^= t[3];
                }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = v0 + v1 + v2 + v3 + (float)(l[0] + l[1]);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int NVALU, int LDSREAD, int TPB>
void run(float* out, unsigned long long* cyc) {
    const int iters = 20000;
    k<NACC, NVALU, LDSREAD, TPB><<<256, TPB>>>(out, cyc, 200);          // warm-up; one block per CU
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<NACC, NVALU, LDSREAD, TPB><<<256, TPB>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const int wps = TPB / 256;
    const double n_mfma = (double)iters * 8.0;                           // per wave
    const double tflops = n_mfma * wps * 1024.0 * (2.0 * 32 * 32 * 16) / (ms * 1e-3) / 1e12;
    printf("acc %d  VALU/MFMA %d  LDS b128 read every %d MFMAs  waves/SIMD %d: %6.1f s_memtime ticks per MFMA per wave; wall %.3f ms = %6.0f TFLOP/s = %5.1f ns per MFMA per SIMD\n",
           NACC, NVALU, LDSREAD, wps, (double)c / n_mfma, ms, tflops, ms * 1e6 / (n_mfma * wps));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 8);
    run<1, 0, 0, 256>(out, cyc); run<2, 0, 0, 256>(out, cyc); run<2, 2, 0, 256>(out, cyc); run<2, 4, 0, 256>(out, cyc);
    run<2, 0, 2, 256>(out, cyc); run<2, 2, 2, 256>(out, cyc);
    run<1, 0, 0, 512>(out, cyc); run<2, 0, 0, 512>(out, cyc); run<2, 1, 0, 512>(out, cyc); run<2, 2, 0, 512>(out, cyc);
    run<2, 3, 0, 512>(out, cyc); run<2, 4, 0, 512>(out, cyc); run<2, 0, 2, 512>(out, cyc); run<2, 2, 2, 512>(out, cyc);
    run<2, 3, 2, 512>(out, cyc); run<2, 2, 1, 512>(out, cyc);
    return 0;
}
