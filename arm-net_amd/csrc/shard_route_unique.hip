// shard_route_unique.hip — routing WITH per-rank de-duplication for the row-sharded lookup (SURVEY.md §8e:
// "exchange volume cut by per-rank id de-dup").  Same contract as armnet_shard_route_ids, except that every
// distinct id of the batch is sent (and its row received) once: send_local has n_unique = sum(counts)
// entries and perm[i] is the slot of id i's row among them.  With 2.5 M uniform lookups into 1 M rows a
// rank requests ~0.92 M rows instead of 2.5 M; skewed (Zipf) ids shrink further.
//
// Direct-address marking, no sort, no hash: position of id = (id % R) * L + id / R  (owner-major, L = rows
// per owner), mark[pos] = 1  ->  exclusive scan  ->  slot[pos]; the compacted positions are already grouped
// by owner and sorted by local row index (owner-side gathers walk their shard monotonically).
#include <hipcub/hipcub.hpp>

#include "armnet_common.h"

namespace armnet {

template <typename IdT>
__global__ void uniq_mark_kernel(int64_t n, const IdT* __restrict__ ids, int R, int64_t nfeat, int64_t L,
                                 int* __restrict__ mark, int32_t* id_status) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t v = (uint64_t)(int64_t)ids[i];
        const bool bad = v >= (uint64_t)nfeat;
        if (bad && id_status) flag_bad_id(id_status);
        const uint32_t id = bad ? 0u : (uint32_t)v;
        mark[(int64_t)(id % (uint32_t)R) * L + id / (uint32_t)R] = 1;
    }
}

__global__ void uniq_compact_kernel(int64_t P, int64_t L, const int* __restrict__ mark,
                                    const int* __restrict__ slot, int32_t* __restrict__ send_local) {
    for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (int64_t)gridDim.x * blockDim.x)
        if (mark[p]) send_local[slot[p]] = (int32_t)(p % L);
}

__global__ void uniq_counts_kernel(int R, int64_t L, const int* __restrict__ mark, const int* __restrict__ slot,
                                   int32_t* __restrict__ counts, int32_t* __restrict__ n_unique) {
    const int r = threadIdx.x;
    const int64_t P = (int64_t)R * L;
    if (r < R) {
        const int lo = slot[(int64_t)r * L];
        const int hi = (r + 1 < R) ? slot[(int64_t)(r + 1) * L] : slot[P - 1] + mark[P - 1];
        counts[r] = hi - lo;
    }
    if (r == 0 && n_unique) *n_unique = slot[P - 1] + mark[P - 1];
}

template <typename IdT>
__global__ void uniq_perm_kernel(int64_t n, const IdT* __restrict__ ids, int R, int64_t nfeat, int64_t L,
                                 const int* __restrict__ slot, int32_t* __restrict__ perm) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t v = (uint64_t)(int64_t)ids[i];
        const uint32_t id = v >= (uint64_t)nfeat ? 0u : (uint32_t)v;
        perm[i] = slot[(int64_t)(id % (uint32_t)R) * L + id / (uint32_t)R];
    }
}

static size_t scan_temp_bytes(int64_t P) {
    size_t bytes = 0;
    // size query only (null temp storage): cannot fail for a valid item count
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, (const int*)nullptr, (int*)nullptr, (int)P);
    return (bytes + 255) & ~(size_t)255;
}

size_t shard_route_unique_ws_bytes(int R, int64_t nfeat) {
    const int64_t L = (nfeat + R - 1) / R, P = (int64_t)R * L;
    return 2 * (((size_t)P * sizeof(int) + 255) & ~(size_t)255) + scan_temp_bytes(P);
}

int launch_shard_route_unique(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* counts,
                              int32_t* send_local, int32_t* perm, int32_t* n_unique, void* ws, size_t ws_bytes,
                              int32_t* id_status, hipStream_t s) {
    if (R < 1 || R > 1024) return ARMNET_ERR_UNSUPPORTED;
    const int64_t L = (nfeat + R - 1) / R, P = (int64_t)R * L;
    if (P >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < shard_route_unique_ws_bytes(R, nfeat)) return ARMNET_ERR_BAD_ARG;
    const size_t seg = ((size_t)P * sizeof(int) + 255) & ~(size_t)255;
    int* mark = reinterpret_cast<int*>(ws);
    int* slot = reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + seg);
    void* tmp = reinterpret_cast<char*>(ws) + 2 * seg;
    size_t tmp_bytes = scan_temp_bytes(P);
    ARMNET_HIP_TRY(hipMemsetAsync(mark, 0, (size_t)P * sizeof(int), s));
    const int tpb = 256;
    const int gn = (int)((n + tpb - 1) / tpb < 4096 ? (n + tpb - 1) / tpb : 4096);
    const int gp = (int)((P + tpb - 1) / tpb < 8192 ? (P + tpb - 1) / tpb : 8192);
    if (n > 0) {
        if (id_type == ARMNET_ID_I64) uniq_mark_kernel<int64_t><<<gn, tpb, 0, s>>>(n, (const int64_t*)ids, R, nfeat, L, mark, id_status);
        else uniq_mark_kernel<int32_t><<<gn, tpb, 0, s>>>(n, (const int32_t*)ids, R, nfeat, L, mark, id_status);
        ARMNET_LAUNCH_CHECK();
    }
    ARMNET_HIP_TRY(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, mark, slot, (int)P, s));
    uniq_compact_kernel<<<gp, tpb, 0, s>>>(P, L, mark, slot, send_local);
    ARMNET_LAUNCH_CHECK();
    uniq_counts_kernel<<<1, 1024, 0, s>>>(R, L, mark, slot, counts, n_unique);
    ARMNET_LAUNCH_CHECK();
    if (n > 0) {
        if (id_type == ARMNET_ID_I64) uniq_perm_kernel<int64_t><<<gn, tpb, 0, s>>>(n, (const int64_t*)ids, R, nfeat, L, slot, perm);
        else uniq_perm_kernel<int32_t><<<gn, tpb, 0, s>>>(n, (const int32_t*)ids, R, nfeat, L, slot, perm);
        ARMNET_LAUNCH_CHECK();
    }
    return ARMNET_OK;
}

}  // namespace armnet

using namespace armnet;

extern "C" int64_t armnet_shard_route_unique_ws_bytes(int R, int64_t nfeat) {
    if (R < 1 || nfeat <= 0) return -1;
    return (int64_t)shard_route_unique_ws_bytes(R, nfeat);
}

extern "C" int armnet_shard_route_unique_ids(int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                                             int32_t* counts, int32_t* send_local, int32_t* perm,
                                             int32_t* n_unique, void* workspace, int64_t ws_bytes,
                                             int32_t* id_status, void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || !counts || (n > 0 && (!ids || !send_local || !perm))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_shard_route_unique(n, ids, id_type, R, nfeat, counts, send_local, perm, n_unique, workspace,
                                     (size_t)ws_bytes, id_status, (hipStream_t)stream);
}
