// gather_delay.hip — does the fused block's PIPELINE SHAPE (12 waves per CU, each with ONE group of row loads in flight
// while it computes for several microseconds) explain why it needs 87 us for memory traffic that gather_stream moves
// in 70 us?  Same traffic as gather_stream (39 ids + 39 values + 39 random 64-byte rows read, 2 KiB written per sample),
// software-pipelined like the fused kernel (rows of group k+1 .. k+DEPTH in flight, ids one group further), plus a
// dependent VALU chain of `delay` steps per group standing in for the block's compute, at a chosen occupancy.
//   gather_delay <waves_per_cu: 4|8|12|16> <delay steps per group> [rotate]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void __launch_bounds__(256) k(const long long* ids, const float* vals, const float* table, float* out,
                                         int B, int F, int delay) {
    extern __shared__ float pad_lds[];                       // occupancy control only
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = gridDim.x * 4;
    const int chunk = lane & 3, r = lane >> 2;
    const int ngroups = B / 2;
    f32x4 rows[DEPTH][5];
    float vv[DEPTH][5];
    long long idn[5];
    auto load_ids = [&](int g) {
        const int gg = g < ngroups ? g : ngroups - 1;
        const size_t e0 = (size_t)gg * 2 * F;
#pragma unroll
        for (int n = 0; n < 5; ++n) { int row = n * 16 + r; if (row >= 2 * F) row = 2 * F - 1; idn[n] = ids[e0 + row]; }
    };
    auto load_rows = [&](int g, int d) {
        const int gg = g < ngroups ? g : ngroups - 1;
        const size_t e0 = (size_t)gg * 2 * F;
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            int row = n * 16 + r; if (row >= 2 * F) row = 2 * F - 1;
            vv[d][n] = vals[e0 + row];
            rows[d][n] = *reinterpret_cast<const f32x4*>(table + (size_t)idn[n] * 16 + chunk * 4);
        }
    };
    int g = blockIdx.x * 4 + wave;
    if (g >= ngroups) return;
    load_ids(g);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) { load_rows(g + d * nw, d); load_ids(g + (d + 1) * nw); }
    f32x4 acc = {0, 0, 0, 0};
    for (; g < ngroups; g += DEPTH * nw) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int gg = g + d * nw;
            f32x4 s = {0, 0, 0, 0};
#pragma unroll
            for (int n = 0; n < 5; ++n) s += rows[d][n] * vv[d][n];
            load_rows(gg + DEPTH * nw, d);                   // refill the buffer that has just been consumed
            load_ids(gg + (DEPTH + 1) * nw);
            float x = s[0];
            for (int i = 0; i < delay; ++i) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);      // the block's compute
            s[0] = x;
            acc += s;
            if (gg < ngroups) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4*>(out + (size_t)gg * 1024 + q * 256 + lane * 4) = s + (float)q;
            }
        }
    }
    if (acc[0] == 12345.f) out[0] = acc[1] + pad_lds[0];
}

int main(int argc, char** argv) {
    const int wpc = argc > 1 ? atoi(argv[1]) : 12;
    const int delay = argc > 2 ? atoi(argv[2]) : 0;
    const int ROT = argc > 3 ? atoi(argv[3]) : 4;
    const int B = 65536, F = 39, NF = 1000000;
    std::vector<long long*> ids(ROT); std::vector<float*> vals(ROT), out(ROT);
    float* table;
    (void)hipMalloc(&table, (size_t)NF * 64); (void)hipMemset(table, 0, (size_t)NF * 64);
    std::vector<long long> h_ids((size_t)B * F);
    srand(1);
    for (int r = 0; r < ROT; ++r) {
        for (auto& x : h_ids) x = ((long long)rand() * 32768 + rand()) % NF;
        (void)hipMalloc(&ids[r], h_ids.size() * 8); (void)hipMalloc(&vals[r], h_ids.size() * 4); (void)hipMalloc(&out[r], (size_t)B * 2048);
        (void)hipMemcpy(ids[r], h_ids.data(), h_ids.size() * 8, hipMemcpyHostToDevice);
        (void)hipMemset(vals[r], 0, h_ids.size() * 4);
    }
    const int bpc = wpc / 4;                                  // 4-wave blocks per CU
    const size_t lds = (size_t)(160 * 1024 / bpc) - 1024;     // LDS request that admits exactly bpc blocks per CU
    const int blocks = 256 * bpc;
    auto run = [&](auto kern, const char* name) {
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        for (int i = 0; i < 8; ++i) kern<<<blocks, 256, lds>>>(ids[i % ROT], vals[i % ROT], table, out[i % ROT], B, F, delay);
        (void)hipDeviceSynchronize();
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        for (int i = 0; i < 40; ++i) kern<<<blocks, 256, lds>>>(ids[i % ROT], vals[i % ROT], table, out[i % ROT], B, F, delay);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 40;
        printf("waves/CU=%2d delay=%5d %-8s: %7.1f us\n", wpc, delay, name, ms * 1e3);
    };
    run(k<1>, "depth1");
    run(k<2>, "depth2");
    run(k<3>, "depth3");
    return 0;
}
