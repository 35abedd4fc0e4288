// shard_route.hip — routing step of the row-sharded embedding lookup (no reference counterpart:
// SURVEY.md §8e).  Rank r of R owns table rows {i : i % R == r} at local index i / R.  For the N = B*F ids
// of a rank's batch this produces, without global atomics and deterministically:
//   counts[r]      how many ids go to owner r
//   send_local[p]  local row index (id / R) of the id placed at send position p (grouped by owner)
//   perm[i]        send position p of id i  (the received rows come back in send order, so perm is the
//                  "id" array the fused kernel uses to index the received row buffer)
// Three launches: per-block LDS histogram -> tiny scan -> per-block scatter.
#include "armnet_common.h"

namespace armnet {

constexpr int ROUTE_TPB = 256;
constexpr int ROUTE_CHUNK = 4096;   // ids per block
constexpr int ROUTE_MAX_R = 64;

template <typename IdT>
__device__ __forceinline__ void owner_of(const IdT* ids, int64_t i, int R, int64_t nfeat, int& owner,
                                         int& local, bool& bad) {
    const uint64_t v = (uint64_t)(int64_t)ids[i];
    bad = v >= (uint64_t)nfeat;
    const uint32_t id = bad ? 0u : (uint32_t)v;
    owner = (int)(id % (uint32_t)R);
    local = (int)(id / (uint32_t)R);
}

template <typename IdT>
__global__ void __launch_bounds__(ROUTE_TPB)
route_count_kernel(int64_t n, const IdT* __restrict__ ids, int R, int64_t nfeat, int* __restrict__ block_counts,
                   int32_t* id_status) {
    __shared__ int hist[ROUTE_MAX_R];
    if (threadIdx.x < R) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * ROUTE_CHUNK;
    for (int k = threadIdx.x; k < ROUTE_CHUNK; k += ROUTE_TPB) {
        const int64_t i = base + k;
        if (i < n) {
            int owner, local; bool bad;
            owner_of(ids, i, R, nfeat, owner, local, bad);
            if (bad && id_status) flag_bad_id(id_status);
            atomicAdd(&hist[owner], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < R) block_counts[(size_t)blockIdx.x * R + threadIdx.x] = hist[threadIdx.x];
}

// one block: counts[r] = sum over blocks; block_offsets[b][r] = start of owner r's segment + sum_{b'<b} counts[b'][r]
__global__ void route_scan_kernel(int nblk, int R, const int* __restrict__ block_counts,
                                  int* __restrict__ block_offsets, int* __restrict__ counts) {
    __shared__ int tot[ROUTE_MAX_R];
    const int r = threadIdx.x;
    if (r < R) {
        int acc = 0;
        for (int b = 0; b < nblk; ++b) {
            block_offsets[(size_t)b * R + r] = acc;
            acc += block_counts[(size_t)b * R + r];
        }
        tot[r] = acc;
        counts[r] = acc;
    }
    __syncthreads();
    if (r < R) {
        int start = 0;
        for (int q = 0; q < r; ++q) start += tot[q];
        for (int b = 0; b < nblk; ++b) block_offsets[(size_t)b * R + r] += start;
    }
}

template <typename IdT>
__global__ void __launch_bounds__(ROUTE_TPB)
route_scatter_kernel(int64_t n, const IdT* __restrict__ ids, int R, int64_t nfeat,
                     const int* __restrict__ block_offsets, int32_t* __restrict__ send_local,
                     int32_t* __restrict__ perm) {
    __shared__ int cursor[ROUTE_MAX_R];
    if (threadIdx.x < R) cursor[threadIdx.x] = block_offsets[(size_t)blockIdx.x * R + threadIdx.x];
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * ROUTE_CHUNK;
    // sequential over the chunk's 16 strips so that positions are deterministic: within a strip, ranks
    // come from a wave ballot per owner plus a per-wave LDS reservation taken in wave order
    for (int k0 = 0; k0 < ROUTE_CHUNK; k0 += ROUTE_TPB) {
        const int64_t i = base + k0 + threadIdx.x;
        int owner = -1, local = 0; bool bad = false;
        if (i < n) owner_of(ids, i, R, nfeat, owner, local, bad);
        int pos = -1;
        for (int w = 0; w < ROUTE_TPB / 64; ++w) {          // waves take turns: deterministic order
            if ((int)(threadIdx.x >> 6) == w) {
                for (int r = 0; r < R; ++r) {
                    const unsigned long long m = __ballot(owner == r);
                    if (m == 0ull) continue;
                    const int lane = threadIdx.x & 63;
                    const int rank = __popcll(m & ((1ull << lane) - 1ull));
                    int b = 0;
                    if (owner == r && rank == 0) { b = cursor[r]; cursor[r] = b + __popcll(m); }
                    b = __shfl(b, __ffsll((long long)m) - 1);
                    if (owner == r) pos = b + rank;
                }
            }
            __syncthreads();
        }
        if (i < n) {
            send_local[pos] = local;
            perm[i] = pos;
        }
    }
}

size_t shard_route_ws_bytes(int64_t n, int R) {
    const int64_t nblk = (n + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
    return (size_t)(2 * nblk * R) * sizeof(int);
}

int launch_shard_route(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* counts,
                       int32_t* send_local, int32_t* perm, void* ws, size_t ws_bytes, int32_t* id_status,
                       hipStream_t s) {
    if (R < 1 || R > ROUTE_MAX_R) return ARMNET_ERR_UNSUPPORTED;
    if (n >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    const int64_t nblk = (n + ROUTE_CHUNK - 1) / ROUTE_CHUNK;
    if (n == 0) {
        ARMNET_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * R, s));
        return ARMNET_OK;
    }
    if (ws_bytes < shard_route_ws_bytes(n, R) || !ws) return ARMNET_ERR_BAD_ARG;
    int* block_counts = reinterpret_cast<int*>(ws);
    int* block_offsets = block_counts + nblk * R;
    if (id_type == ARMNET_ID_I64)
        route_count_kernel<int64_t><<<(int)nblk, ROUTE_TPB, 0, s>>>(n, (const int64_t*)ids, R, nfeat, block_counts, id_status);
    else
        route_count_kernel<int32_t><<<(int)nblk, ROUTE_TPB, 0, s>>>(n, (const int32_t*)ids, R, nfeat, block_counts, id_status);
    ARMNET_LAUNCH_CHECK();
    route_scan_kernel<<<1, ROUTE_MAX_R, 0, s>>>((int)nblk, R, block_counts, block_offsets, counts);
    ARMNET_LAUNCH_CHECK();
    if (id_type == ARMNET_ID_I64)
        route_scatter_kernel<int64_t><<<(int)nblk, ROUTE_TPB, 0, s>>>(n, (const int64_t*)ids, R, nfeat, block_offsets, send_local, perm);
    else
        route_scatter_kernel<int32_t><<<(int)nblk, ROUTE_TPB, 0, s>>>(n, (const int32_t*)ids, R, nfeat, block_offsets, send_local, perm);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

}  // namespace armnet
