// mfma4x4.hip — layout and issue rate of v_mfma_f32_4x4x1_16b_f32 on gfx950 (developer microbenchmark).
//   hipcc --offload-arch=gfx950 -O3 -o mfma4x4 mfma4x4.hip && ./mfma4x4
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    const int l = threadIdx.x;
    // A = 100 * lane, B = lane: D[i][j] (block b) = A[b][i] * B[b][j]
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(100 * l), (float)l, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void rate(float* out, int iters, long long* cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
__global__ void rate16(float* out, int iters, long long* cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x * 0.002f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[threadIdx.x + blockIdx.x * blockDim.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float* d; long long* c;
    hipMalloc(&d, 1 << 20); hipMalloc(&c, 8);
    layout<<<1, 64>>>(d);
    float h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("layout of v_mfma_f32_4x4x1_16b_f32 (A = 100*lane, B = lane): lane: reg0..3\n");
    for (int l = 0; l < 64; l += 1)
        if (l < 8 || l >= 60) printf("  lane %2d: %8.0f %8.0f %8.0f %8.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    const int iters = 2000;
    long long hc;
#define RUN(K, N, W)                                                                                  \
    K<N><<<1, 64 * W>>>(d, iters, c); hipDeviceSynchronize();                                         \
    K<N><<<1, 64 * W>>>(d, iters, c); hipDeviceSynchronize();                                         \
    hipMemcpy(&hc, c, 8, hipMemcpyDeviceToHost);                                                      \
    printf("%-8s NACC=%2d waves/block=%2d: %.2f s_memtime ticks per MFMA per wave\n", #K, N, W, (double)hc / (iters * N));
    RUN(rate, 1, 1) RUN(rate, 2, 1) RUN(rate, 4, 1) RUN(rate, 8, 1) RUN(rate, 8, 4) RUN(rate, 8, 8) RUN(rate, 8, 16)
    RUN(rate, 4, 16) RUN(rate, 2, 16) RUN(rate, 1, 16)
    RUN(rate16, 1, 1) RUN(rate16, 4, 1) RUN(rate16, 4, 4) RUN(rate16, 4, 8) RUN(rate16, 4, 16)
    // chip-level throughput: every CU busy, 4 waves/SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int big_iters = 20000, blocks = 256 * 4;
    float ms;
    rate<8><<<blocks, 256>>>(d, 100, c); hipDeviceSynchronize();
    hipEventRecord(e0); rate<8><<<blocks, 256>>>(d, big_iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("4x4x1   chip: %.1f TMAC/s (%.1f TFLOP/s)\n", (double)blocks * 4 * big_iters * 8 * 256 / (ms * 1e-3) / 1e12, 2.0 * blocks * 4 * big_iters * 8 * 256 / (ms * 1e-3) / 1e12);
    rate16<4><<<blocks, 256>>>(d, 100, c); hipDeviceSynchronize();
    hipEventRecord(e0); rate16<4><<<blocks, 256>>>(d, big_iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("16x16x4 chip: %.1f TMAC/s (%.1f TFLOP/s)\n", (double)blocks * 4 * big_iters * 4 * 1024 / (ms * 1e-3) / 1e12, 2.0 * blocks * 4 * big_iters * 4 * 1024 / (ms * 1e-3) / 1e12);
    return 0;
}
