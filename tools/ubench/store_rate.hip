// store_rate.hip — per-CU rate of the vector-memory STORE path on MI355X (round 6): 128 MB written per launch as
// global_store_dword / dwordx2 / dwordx4, plain and non-temporal, on 32 and on 256 CUs (hipExtStreamCreateWithCUMask).
// gather_stream's `cus` mode says the fused block's 2 KiB per sample of output costs ~68 clocks of a CU per 1 KiB store
// instruction; this asks whether another store form is cheaper.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// every wave writes contiguous 1 KiB pieces (like the block's epilogue: a sample's 2 KiB = 2 pieces)
template <int W, int NT>
__global__ void __launch_bounds__(256) k(float* out, size_t n_floats, float v) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
    for (size_t p = wave * 256; p + 256 <= n_floats; p += nw * 256) {
        float* base = out + p;
        if constexpr (W == 4) {
            f32x4 x = {v, v + 1, v + 2, v + 3};
            if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x4*>(base) + lane);
            else reinterpret_cast<f32x4*>(base)[lane] = x;
        } else if constexpr (W == 2) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x2 x = {v + q, v + 1};
                if constexpr (NT) __builtin_nontemporal_store(x, reinterpret_cast<f32x2*>(base) + q * 64 + lane);
                else reinterpret_cast<f32x2*>(base)[q * 64 + lane] = x;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (NT) __builtin_nontemporal_store(v + q, base + q * 64 + lane);
                else base[q * 64 + lane] = v + q;
            }
        }
    }
}

int main() {
    const size_t n = (size_t)32 << 20;       // 128 MB
    float* out[4];
    for (auto& o : out) hipMalloc(&o, n * 4);
    for (int cus : {32, 256}) {
        uint32_t mask[8] = {0};
        for (int b = 0; b < cus; ++b) mask[b / 32] |= 1u << (b % 32);
        hipStream_t s;
        if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("no CU mask\n"); return 1; }
        auto run = [&](auto kern, const char* name) {
            for (int i = 0; i < 4; ++i) kern<<<1024, 256, 0, s>>>(out[i % 4], n, 1.f);
            hipStreamSynchronize(s);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, s);
            for (int i = 0; i < 20; ++i) kern<<<1024, 256, 0, s>>>(out[i % 4], n, (float)i);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
            const double clk_per_kib = ms * 1e-3 * 2.4e9 / ((double)n * 4 / 1024 / cus);
            printf("CUs %3d %-22s: %7.1f us  %6.0f GB/s  %5.1f clocks of a CU per KiB (2.4 GHz)\n", cus, name, ms * 1e3,
                   (double)n * 4 / ms / 1e6, clk_per_kib);
        };
        run(k<4, 0>, "dwordx4");
        run(k<4, 1>, "dwordx4 nontemporal");
        run(k<2, 0>, "dwordx2");
        run(k<2, 1>, "dwordx2 nontemporal");
        run(k<1, 0>, "dword");
        run(k<1, 1>, "dword nontemporal");
    }
    return 0;
}
