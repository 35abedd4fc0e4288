// gather_calib.hip — what ONE random row request costs at the fabric, by the counters (round-2 verdict, weak 10).
//   gather_calib <row_bytes: 64|128|256> <nfeat> <rows_per_launch> [launches]
// A gather-ONLY kernel: row ids come from an in-kernel hash (no id / value stream, no stores), every lane group of
// row_bytes/16 lanes reads one whole random row with 16-byte loads — the fused kernel's staging access.  Run under
//   rocprofv3 --pmc FETCH_SIZE                       (and TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum, TCC_REQ_sum / TCC_MISS_sum)
// and divide by rows_per_launch.  `stream` as row_bytes reads the same number of bytes as a coalesced stream (the
// guide's calibration case: FETCH_SIZE tallies it at half).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// the same 16-byte load with a cache-policy modifier (POL: 0 default, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc0 nt,
// 7 sc1 nt): does any of them make the L2 ask the fabric for LESS than a whole 128-byte line per 64-byte row?
template <int POL>
__device__ __forceinline__ f32x4 load16(const float* p) {
    f32x4 v;
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int POL>
__global__ void __launch_bounds__(256) gather_only_pol(const float* __restrict__ table, float* sink, unsigned nfeat,
                                                       unsigned rows, unsigned salt) {
    constexpr int LPR = 4;                                   // 64-byte rows
    const unsigned tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
    const unsigned chunk = tid % LPR;
    f32x4 acc = {0, 0, 0, 0};
    for (unsigned r = tid / LPR; r < rows; r += 4 * (nthreads / LPR)) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned rr = r + u * (nthreads / LPR);
            if (rr >= rows) rr = rows - 1;
            const unsigned id = hash32(rr * 2654435761u + salt) % nfeat;
            v[u] = load16<POL>(table + (size_t)id * (LPR * 4) + chunk * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] == 12345.f) sink[0] = acc[1] + acc[2] + acc[3];
}

template <int LPR>   // lanes per row (row_bytes / 16)
__global__ void __launch_bounds__(256) gather_only(const float* __restrict__ table, float* sink, unsigned nfeat,
                                                   unsigned rows, unsigned salt) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
    const unsigned chunk = tid % LPR;
    f32x4 acc = {0, 0, 0, 0};
    // 4 independent requests in flight per lane
    for (unsigned r = tid / LPR; r < rows; r += 4 * (nthreads / LPR)) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned rr = r + u * (nthreads / LPR);
            const unsigned id = hash32(rr * 2654435761u + salt) % nfeat;
            v[u] = rr < rows ? *reinterpret_cast<const f32x4*>(table + (size_t)id * (LPR * 4) + chunk * 4) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] == 12345.f) sink[0] = acc[1] + acc[2] + acc[3];
}

__global__ void __launch_bounds__(256) stream_only(const float* __restrict__ src, float* sink, size_t n16) {
    f32x4 acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        acc += reinterpret_cast<const f32x4*>(src)[i];
    if (acc[0] == 12345.f) sink[0] = acc[1] + acc[2] + acc[3];
}

int main(int argc, char** argv) {
    const bool stream = argc > 1 && !strcmp(argv[1], "stream");
    const int row_bytes = stream ? 64 : (argc > 1 ? atoi(argv[1]) : 64);
    const unsigned nfeat = argc > 2 ? (unsigned)atoll(argv[2]) : 1000000u;
    const unsigned rows = argc > 3 ? (unsigned)atoll(argv[3]) : 2555904u;    // 65536 x 39
    const int launches = argc > 4 ? atoi(argv[4]) : 20;
    const int pol = argc > 5 ? atoi(argv[5]) : -1;           // >= 0: 64-byte rows through load16<pol>
    float *table, *sink;
    const size_t tbytes = (size_t)nfeat * row_bytes;
    hipMalloc(&table, tbytes); hipMemset(table, 0, tbytes); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int i) {
        if (pol >= 0) {
            switch (pol) {
                case 0: gather_only_pol<0><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                case 1: gather_only_pol<1><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                case 2: gather_only_pol<2><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                case 3: gather_only_pol<3><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                case 4: gather_only_pol<4><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                case 5: gather_only_pol<5><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                case 6: gather_only_pol<6><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
                default: gather_only_pol<7><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i); break;
            }
        } else if (stream) stream_only<<<4096, 256>>>(table, sink, (size_t)rows * 64 / 16 < tbytes / 16 ? (size_t)rows * 4 : tbytes / 16);
        else if (row_bytes == 64) gather_only<4><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i);
        else if (row_bytes == 128) gather_only<8><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i);
        else gather_only<16><<<2048, 256>>>(table, sink, nfeat, rows, 977u * i);
    };
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) launch(100 + i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= launches;
    if (pol >= 0) printf("POLICY %d (0 default, 1 nt, 2 sc0, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc0 nt, 7 sc1 nt): ", pol);
    printf("%s row_bytes=%d nfeat=%u (table %.0f MB) rows/launch=%u: %.1f us per launch, %.2f G rows/s, %.0f GB/s of row payload\n",
           stream ? "STREAM" : "GATHER", row_bytes, nfeat, tbytes / 1e6, rows, ms * 1e3, rows / ms / 1e6,
           (double)rows * row_bytes / ms / 1e6);
    return 0;
}
