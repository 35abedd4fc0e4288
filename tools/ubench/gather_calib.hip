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
    float *table, *sink;
    const size_t tbytes = (size_t)nfeat * row_bytes;
    hipMalloc(&table, tbytes); hipMemset(table, 0, tbytes); hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int i) {
        if (stream) stream_only<<<4096, 256>>>(table, sink, (size_t)rows * 64 / 16 < tbytes / 16 ? (size_t)rows * 4 : tbytes / 16);
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
    printf("%s row_bytes=%d nfeat=%u (table %.0f MB) rows/launch=%u: %.1f us per launch, %.2f G rows/s, %.0f GB/s of row payload\n",
           stream ? "STREAM" : "GATHER", row_bytes, nfeat, tbytes / 1e6, rows, ms * 1e3, rows / ms / 1e6,
           (double)rows * row_bytes / ms / 1e6);
    return 0;
}
