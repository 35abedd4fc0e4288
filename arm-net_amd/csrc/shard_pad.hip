// shard_pad.hip — fixed-capacity layout of a routed lookup (row-sharded table; no reference counterpart).
//
// armnet_shard_route_ids / _unique_ids leave the local row indices grouped by owner back to back (counts[r] of them
// for owner r).  Exchanging that needs the counts on the HOST (all_to_all_single split sizes): one device->host
// synchronisation in the middle of every step.  This kernel re-lays the routed lookup into R equal slots of `cap`
// indices, so that both exchanges are equal-split all-to-alls whose sizes the host knows without looking at the data:
//     send_pad[o*cap + s] = send_local[start[o] + s]   (s < counts[o]; the rest of the slot holds index 0: a valid row,
//                                                       fetched and never looked at)
//     perm_pad[i]         = o*cap + s                  for the compact position perm[i] = start[o] + s
// If an owner's count exceeds `cap`, *overflow is OR-ed with 1 (the surplus lookups read slot 0 of that owner): the
// caller checks the flag when it next synchronises anyway and repeats the step with the exact protocol.
#include "armnet_common.h"

namespace armnet {

constexpr int PAD_MAX_R = 64;

__global__ void __launch_bounds__(256)
shard_pad_kernel(int64_t n, int R, int64_t cap, const int32_t* __restrict__ counts,
                 const int32_t* __restrict__ send_local, const int32_t* __restrict__ perm,
                 int32_t* __restrict__ send_pad, int32_t* __restrict__ perm_pad, int32_t* overflow) {
    __shared__ int64_t start[PAD_MAX_R + 1];
    if (threadIdx.x == 0) {
        int64_t acc = 0;
        bool over = false;
        for (int r = 0; r < R; ++r) {
            start[r] = acc;
            acc += counts[r];
            over |= counts[r] > cap;
        }
        start[R] = acc;
        if (over && blockIdx.x == 0) atomicOr(overflow, 1);
    }
    __syncthreads();
    const int64_t total = (int64_t)R * cap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total || i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i < total) {
            const int o = (int)(i / cap);
            const int64_t s = i - (int64_t)o * cap;
            send_pad[i] = s < counts[o] ? send_local[start[o] + s] : 0;
        }
        if (i < n) {
            const int64_t p = perm[i];
            int o = 0;
            while (o + 1 < R && p >= start[o + 1]) ++o;       // R <= 64: a short linear search
            const int64_t s = p - start[o];
            perm_pad[i] = (int32_t)((int64_t)o * cap + (s < cap ? s : 0));
        }
    }
}

int launch_shard_pad(int64_t n, int R, int64_t cap, const int32_t* counts, const int32_t* send_local,
                     const int32_t* perm, int32_t* send_pad, int32_t* perm_pad, int32_t* overflow, hipStream_t st) {
    if (R < 1 || R > PAD_MAX_R) return ARMNET_ERR_UNSUPPORTED;
    if ((int64_t)R * cap >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    const int64_t work = (int64_t)R * cap > n ? (int64_t)R * cap : n;
    if (work == 0) return ARMNET_OK;
    int64_t grid = (work + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    shard_pad_kernel<<<(int)grid, 256, 0, st>>>(n, R, cap, counts, send_local, perm, send_pad, perm_pad, overflow);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

// Whole-shard exchange (the batch asks for most of every shard: the slot of the de-duplicated request list would be
// the shard itself): the owners ship their shards as they are (all-gather, L = ceil(nfeat / R) rows each, the last
// shards padded by one row), no request list, no owner-side gather; row `id` then sits at the direct address
// (id % R) * L + id / R of the gathered buffer.  Out-of-range ids are flagged and read row 0.
template <typename IdT>
__global__ void __launch_bounds__(256)
shard_direct_perm_kernel(int64_t n, const IdT* __restrict__ ids, int R, int64_t nfeat, int64_t L,
                         int32_t* __restrict__ perm, int32_t* id_status) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t v = (uint64_t)(int64_t)ids[i];
        const bool bad = v >= (uint64_t)nfeat;
        if (bad && id_status) flag_bad_id(id_status);
        const uint32_t id = bad ? 0u : (uint32_t)v;
        perm[i] = (int32_t)((int64_t)(id % (uint32_t)R) * L + id / (uint32_t)R);
    }
}

int launch_shard_direct_perm(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm,
                             int32_t* id_status, hipStream_t st) {
    const int64_t L = (nfeat + R - 1) / R;
    if ((int64_t)R * L >= ((int64_t)1 << 31) || nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (n == 0) return ARMNET_OK;
    int64_t grid = (n + 255) / 256;
    if (grid > 256 * 16) grid = 256 * 16;
    if (id_type == ARMNET_ID_I64)
        shard_direct_perm_kernel<int64_t><<<(int)grid, 256, 0, st>>>(n, (const int64_t*)ids, R, nfeat, L, perm, id_status);
    else
        shard_direct_perm_kernel<int32_t><<<(int)grid, 256, 0, st>>>(n, (const int32_t*)ids, R, nfeat, L, perm, id_status);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

}  // namespace armnet

using namespace armnet;

extern "C" int armnet_shard_direct_perm(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm,
                                        int32_t* id_status, void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || (n > 0 && (!ids || !perm))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_shard_direct_perm(n, ids, id_type, R, nfeat, perm, id_status, (hipStream_t)stream);
}

extern "C" int armnet_shard_pad_route(int64_t n, int R, int64_t cap, const int32_t* counts, const int32_t* send_local,
                                      const int32_t* perm, int32_t* send_pad, int32_t* perm_pad, int32_t* overflow,
                                      void* stream) {
    if (n < 0 || R < 1 || cap < 1 || !counts || !send_pad || !overflow || (n > 0 && (!send_local || !perm || !perm_pad)))
        return ARMNET_ERR_BAD_ARG;
    return launch_shard_pad(n, R, cap, counts, send_local, perm, send_pad, perm_pad, overflow, (hipStream_t)stream);
}
