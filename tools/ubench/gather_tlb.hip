// gather_tlb.hip — does address translation bound a random row gather once the table outgrows the TLB reach?
// (round-3 verdict, weak 5: config 4 at full size — nfeat 100 M x 64 floats = 25.6 GB — runs 2.25x slower than the same
// kernel on a 2.56 GB table with identical request counts.)
//   gather_tlb <row_bytes 64|256> <table_GB> <mode> [alloc] [rows_per_launch] [launches]
// mode  0 uniform   : every lane group reads a uniformly random row of the whole table (the fused block's pattern)
//       1 xcd-slab  : block b only reads rows of slab (b mod 8) of the table (blocks are dealt round-robin to the 8
//                     XCDs, each with its own L2 / UTCL2: one XCD then touches 1/8 of the pages)
//       2 sorted    : rows are visited in ascending address order (what a page-sorted lookup list would give); the
//                     lanes in flight at any time cover a window of table_bytes * (lanes in flight) / rows
//       3 window    : uniformly random inside a window of <window_MB> that moves with the block's progress
// alloc 0 hipMalloc, 1 hipExtMallocWithFlags(hipDeviceMallocContiguous)
// Run plain for the time, and under rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum ... for the why.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int LPR, int MODE>
__global__ void __launch_bounds__(256) gather_rows(const float* __restrict__ table, float* sink, unsigned long long nfeat,
                                                   unsigned rows, unsigned salt, unsigned long long window_rows) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x, nthreads = gridDim.x * 256;
    const unsigned chunk = tid % LPR;
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long slab = nfeat / 8;
    for (unsigned r = tid / LPR; r < rows; r += 4 * (nthreads / LPR)) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned rr = r + u * (nthreads / LPR);
            const unsigned long long h = ((unsigned long long)hash32(rr * 2654435761u + salt) << 20) ^ hash32(rr + 0x9e3779b9u * salt);
            unsigned long long id;
            if constexpr (MODE == 0) id = h % nfeat;
            else if constexpr (MODE == 1) id = (blockIdx.x % 8) * slab + h % slab;
            else if constexpr (MODE == 2) id = (unsigned long long)((double)rr / rows * (double)(nfeat - 1));
            else {
                const unsigned long long base = (unsigned long long)((double)r / rows * (double)(nfeat - window_rows));
                id = base + h % window_rows;
            }
            v[u] = rr < rows ? *reinterpret_cast<const f32x4*>(table + id * (LPR * 4) + chunk * 4) : f32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u];
    }
    if (acc[0] == 12345.f) sink[0] = acc[1] + acc[2] + acc[3];
}

__global__ void __launch_bounds__(256) touch(float* p, size_t n16) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
        reinterpret_cast<f32x4*>(p)[i] = f32x4{0, 0, 0, 0};
}

int main(int argc, char** argv) {
    const int row_bytes = argc > 1 ? atoi(argv[1]) : 256;
    const double gb = argc > 2 ? atof(argv[2]) : 25.6;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int alloc = argc > 4 ? atoi(argv[4]) : 0;
    const unsigned rows = argc > 5 ? (unsigned)atoll(argv[5]) : 2555904u;    // 65536 x 39
    const int launches = argc > 6 ? atoi(argv[6]) : 10;
    const double window_mb = argc > 7 ? atof(argv[7]) : 2048.0;
    const unsigned long long nfeat = (unsigned long long)(gb * 1e9 / row_bytes);
    const size_t tbytes = (size_t)nfeat * row_bytes;
    float *table = nullptr, *sink;
    hipError_t e = alloc == 1 ? hipExtMallocWithFlags((void**)&table, tbytes, hipDeviceMallocContiguous) : hipMalloc(&table, tbytes);
    if (e != hipSuccess) { printf("alloc(%d) of %.1f GB failed: %s\n", alloc, gb, hipGetErrorString(e)); return 1; }
    touch<<<4096, 256>>>(table, tbytes / 16);
    hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    unsigned long long wr = (unsigned long long)(window_mb * 1e6 / row_bytes);
    if (wr > nfeat) wr = nfeat;
    auto launch = [&](int i) {
        const unsigned salt = 977u * i + 1;
#define GO(L, M) gather_rows<L, M><<<2048, 256>>>(table, sink, nfeat, rows, salt, wr)
        if (row_bytes == 64) { if (mode == 0) GO(4, 0); else if (mode == 1) GO(4, 1); else if (mode == 2) GO(4, 2); else GO(4, 3); }
        else { if (mode == 0) GO(16, 0); else if (mode == 1) GO(16, 1); else if (mode == 2) GO(16, 2); else GO(16, 3); }
    };
    for (int i = 0; i < 3; ++i) launch(i);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipEventRecord(e0);
    for (int i = 0; i < launches; ++i) launch(100 + i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= launches;
    const char* mn[] = {"uniform", "xcd-slab", "sorted", "window"};
    printf("GATHER_TLB row_bytes=%d table=%.2f GB mode=%s%s alloc=%s rows/launch=%u: %.1f us per launch, %.2f G rows/s, %.0f GB/s payload\n",
           row_bytes, tbytes / 1e9, mn[mode], mode == 3 ? (window_mb >= 1000 ? " (GB-window)" : " (MB-window)") : "",
           alloc ? "contiguous" : "hipMalloc", rows, ms * 1e3, rows / ms / 1e6, (double)rows * row_bytes / ms / 1e6);
    if (mode == 3) printf("    window = %.0f MB\n", window_mb);
    return 0;
}
