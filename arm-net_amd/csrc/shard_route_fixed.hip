// shard_route_fixed.hip — routing of the row-sharded lookup's FIXED-capacity protocol in one entry point (SURVEY.md §8e;
// no reference counterpart).  Round 3 ran the request-list step as route (counting sort, or direct-address mark + hipCUB
// scan + compact + perm) followed by a separate pad pass: 91 us per 2.56 M lookups on one MI355X, more than the fused
// block it feeds.  The slots of the fixed protocol make most of that unnecessary: an owner's slot is `cap` entries wide
// whatever the data looks like, so a lookup only needs a UNIQUE position inside its owner's slot — not a sorted one — and
// the padded layout can be written directly.
//
//   armnet_shard_route_fixed(n, ids, R, nfeat, cap, dedup, ...) ->
//       send_pad[o*cap + s]  local row index (id / R) of the s-th request to owner o (= id % R); unused entries 0
//       perm_pad[i]          o*cap + s of lookup i (index of its row in the buffer the row exchange returns)
//       counts[o]            requests to owner o (may exceed cap: *overflow |= 1, surplus lookups point at slot 0)
//
// dedup = 0 — every lookup is a request.  ONE kernel: a block takes 2048 lookups, ranks them per owner with LDS
//   atomics, reserves its share of each owner's slot with one global atomicAdd per owner, writes both arrays.
//   Positions inside a slot depend on the order in which blocks reserve (run to run), the rows fetched do not.
// dedup = 1 — every DISTINCT id is a request (2.56 M uniform lookups of 1 M rows ask for 0.92 M rows).  Direct-address
//   BYTE map over (owner, local) — measured (tools/ubench/route_mark.hip): 2.56 M byte stores into a 1 MB map 14 us,
//   4-byte stores 32 us, atomicOr into a bitmap 98-124 us — then chunk sums, a one-block scan with a restart at every
//   owner, and an emit pass that writes the slots and a COMPACT position table of the map's shape BESIDE it (the map
//   itself keeps the epoch of the last step that marked a position, see below): a marked id's byte of the `rank` array is
//   its rank inside its 256-id group (one wave of the emit pass) and every group gets a 4-byte base, so
//   perm_pad[i] = o * cap + base[p >> 8] + rank[p] is one 1-byte gather from a table of nfeat bytes (L2-resident; the
//   4-byte table it replaces was 4 MB per million rows, written and gathered at 4x the traffic) plus a 4-byte one from
//   a table 1/64 of that.  Requests inside a slot come out sorted by local row index (the owner-side gather walks its
//   shard monotonically).
//
// Hot rows (round 5, SURVEY.md §8e's third lever): ids below `hot.rows` — the head of a frequency-ordered id space —
// are replicated on every rank and never cross the links: such a lookup gets perm_pad[i] = hot.base + id (the caller
// appends its replicated hot rows to the received buffer at row hot.base) and takes part in no slot, mark or count.
#include "armnet_common.h"

namespace armnet {

constexpr int RF_TPB = 256;
constexpr int RF_PER = 8;                     // lookups per thread (dedup = 0)
constexpr int RF_MAX_R = 64;
constexpr int RF_CHUNK = 1024;                // bytes of the mark map per block (dedup = 1): 4 per thread (one 32-bit load)

// id -> (owner, local) with a shift when R is a power of two (the usual 2, 4, 8 ranks)
struct OwnerMap {
    uint32_t R;
    int shift;                                // log2(R), or -1
    __device__ __forceinline__ void split(uint32_t id, uint32_t& owner, uint32_t& local) const {
        if (shift >= 0) {
            owner = id & (R - 1u);
            local = id >> shift;
        } else {
            local = id / R;
            owner = id - local * R;
        }
    }
};
static OwnerMap make_owner_map(int R) {
    OwnerMap m;
    m.R = (uint32_t)R;
    m.shift = -1;
    for (int s = 0; s < 31; ++s)
        if ((1 << s) == R) m.shift = s;
    return m;
}

// the replicated head of the id space: ids < rows are served from row base + id of the consumer's buffer, never routed
struct HotSet {
    int64_t rows, base;
    __device__ __forceinline__ bool has(uint32_t id) const { return (int64_t)id < rows; }
};

template <typename IdT>
__device__ __forceinline__ uint32_t checked_id(const IdT* ids, int64_t i, int64_t nfeat, int32_t* id_status) {
    const uint64_t v = (uint64_t)(int64_t)ids[i];
    const bool bad = v >= (uint64_t)nfeat;
    if (bad && id_status) flag_bad_id(id_status);
    return bad ? 0u : (uint32_t)v;
}

// ---- dedup = 0 ------------------------------------------------------------------------------------------------------------
template <typename IdT>
__global__ void __launch_bounds__(RF_TPB)
route_slots_kernel(int64_t n, const IdT* __restrict__ ids, OwnerMap om, int64_t nfeat, int64_t cap, HotSet hot,
                   int32_t* __restrict__ send_pad, int32_t* __restrict__ perm_pad, int32_t* __restrict__ counts,
                   int32_t* overflow, int32_t* id_status) {
    __shared__ int hist[RF_MAX_R];
    __shared__ int base[RF_MAX_R];
    const int R = (int)om.R;
    if ((int)threadIdx.x < R) hist[threadIdx.x] = 0;
    __syncthreads();
    const int64_t i0 = (int64_t)blockIdx.x * (RF_TPB * RF_PER) + threadIdx.x;
    uint32_t owner[RF_PER], local[RF_PER];
    int rank[RF_PER];                                              // -1: a hot id (local[k] holds the id itself)
#pragma unroll
    for (int k = 0; k < RF_PER; ++k) {
        const int64_t i = i0 + (int64_t)k * RF_TPB;
        if (i < n) {
            const uint32_t id = checked_id(ids, i, nfeat, id_status);
            if (hot.has(id)) {
                local[k] = id;
                owner[k] = 0;
                rank[k] = -1;
            } else {
                om.split(id, owner[k], local[k]);
                rank[k] = atomicAdd(&hist[owner[k]], 1);           // LDS: unique rank of this lookup among the block's requests to that owner
            }
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < R) base[threadIdx.x] = hist[threadIdx.x] ? atomicAdd(&counts[threadIdx.x], hist[threadIdx.x]) : 0;
    __syncthreads();
    bool over = false;
#pragma unroll
    for (int k = 0; k < RF_PER; ++k) {
        const int64_t i = i0 + (int64_t)k * RF_TPB;
        if (i < n) {
            if (rank[k] < 0) {
                perm_pad[i] = (int32_t)(hot.base + (int64_t)local[k]);
                continue;
            }
            const int64_t s = (int64_t)base[owner[k]] + rank[k];
            const int64_t slot0 = (int64_t)owner[k] * cap;
            if (s < cap) {
                send_pad[slot0 + s] = (int32_t)local[k];
                perm_pad[i] = (int32_t)(slot0 + s);
            } else {
                perm_pad[i] = (int32_t)slot0;                      // a valid row, the wrong one: the step is repeated exactly
                over = true;
            }
        }
    }
    if (over) atomicOr(overflow, 1);
}

// ---- dedup = 1 ------------------------------------------------------------------------------------------------------------
// position of an id in the (owner, local) order, owners padded to Lp (a multiple of RF_CHUNK) entries
template <typename IdT>
__global__ void __launch_bounds__(RF_TPB)
uniq_mark_bytes_kernel(int64_t n, const IdT* __restrict__ ids, OwnerMap om, int64_t nfeat, int64_t Lp, HotSet hot,
                       unsigned char* __restrict__ mark, unsigned char epoch, int32_t* id_status) {
    for (int64_t i = (int64_t)blockIdx.x * RF_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * RF_TPB) {
        uint32_t o, l;
        const uint32_t id = checked_id(ids, i, nfeat, id_status);
        if (hot.has(id)) continue;                                  // replicated: asks nobody
        om.split(id, o, l);
        mark[(int64_t)o * Lp + l] = epoch;
    }
}

// The map holds the EPOCH of the last step that marked a position (1..255; 0 = never): a step's marks are the bytes equal
// to its epoch, so the map is zeroed once per 255 steps instead of once per step (a 1 MB fill is a launch like any other).
__device__ __forceinline__ uint32_t eq_bytes(uint32_t w, uint32_t e4) {
    // 0x01 in every byte of w that equals the epoch (e4 = epoch replicated into 4 bytes)
    const uint32_t x = w ^ e4;                                      // zero bytes <=> equal
    const uint32_t nz = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;      // bit 7 of a byte set <=> byte non-zero
    return (~nz >> 7) & 0x01010101u;
}

__global__ void __launch_bounds__(RF_TPB)
uniq_chunk_sums_kernel(const unsigned char* __restrict__ mark, uint32_t e4, int* __restrict__ sums) {
    __shared__ int wsum[RF_TPB / 64];
    const uint32_t v = reinterpret_cast<const uint32_t*>(mark + (size_t)blockIdx.x * RF_CHUNK)[threadIdx.x];
    int c = __popc(eq_bytes(v, e4));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one block: exclusive scan of the chunk sums that restarts at every owner (cpo chunks per owner); counts, overflow
__global__ void __launch_bounds__(1024)
uniq_scan_kernel(int nchunk, int cpo, int R, int64_t cap, const int* __restrict__ sums, int* __restrict__ base,
                 int32_t* __restrict__ counts, int32_t* overflow) {
    __shared__ int part[1024];
    // thread t scans a contiguous run of chunks; runs never straddle an owner when cpo % per == 0 is not guaranteed, so the
    // scan carries (value, owner of the run's first chunk) explicitly: simple two-level scheme per owner instead
    for (int o = 0; o < R; ++o) {
        const int c0 = o * cpo;
        const int per = (cpo + 1023) / 1024;
        const int lo = c0 + (int)threadIdx.x * per;
        const int hi = min(lo + per, c0 + cpo);
        int acc = 0;
        for (int c = lo; c < hi; ++c) acc += sums[c];
        part[threadIdx.x] = acc;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                        // Hillis-Steele inclusive scan over the 1024 partials
            const int v = (int)threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        int run = part[threadIdx.x] - acc;                          // exclusive prefix of this thread's run
        for (int c = lo; c < hi; ++c) {
            base[c] = run;
            run += sums[c];
        }
        if (threadIdx.x == 1023) {
            counts[o] = part[1023];
            if (part[1023] > cap) atomicOr(overflow, 1);
        }
        __syncthreads();
    }
    (void)nchunk;
}

// slots + position table; the blocks of an owner share the zero fill of its slot's unused tail.
// SCAN_HERE: the block sums the chunk sums of its owner itself (the ones in front of it: its base; all of them: the
// owner's count) instead of reading a scanned array — one launch and 10 R block-wide barriers of a single block less;
// cpo loads per block, so only while an owner has few chunks (launcher: cpo <= 8192).
template <bool SCAN_HERE>
__global__ void __launch_bounds__(RF_TPB)
uniq_emit_kernel(int cpo, int64_t Lp, int64_t cap, const unsigned char* __restrict__ mark, uint32_t e4,
                 unsigned char* __restrict__ rank, const int* __restrict__ base, int32_t* __restrict__ counts,
                 int32_t* __restrict__ send_pad, int32_t* __restrict__ grp_base, int64_t* cap_out, int32_t* overflow) {
    __shared__ int wsum[RF_TPB / 64];
    __shared__ int bsum[2][RF_TPB / 64];
    const int chunk = blockIdx.x, o = chunk / cpo, j = chunk - o * cpo;
    int my_base, my_count;
    if constexpr (SCAN_HERE) {
        const int* sums = base;                                     // un-scanned chunk sums
        int before = 0, all = 0;
        for (int c = threadIdx.x; c < cpo; c += RF_TPB) {
            const int v = sums[o * cpo + c];
            all += v;
            before += c < j ? v : 0;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            before += __shfl_xor(before, d);
            all += __shfl_xor(all, d);
        }
        if ((threadIdx.x & 63) == 0) {
            bsum[0][threadIdx.x >> 6] = before;
            bsum[1][threadIdx.x >> 6] = all;
        }
        __syncthreads();
        my_base = bsum[0][0] + bsum[0][1] + bsum[0][2] + bsum[0][3];
        my_count = bsum[1][0] + bsum[1][1] + bsum[1][2] + bsum[1][3];
        if (j == 0 && threadIdx.x == 0) {
            counts[o] = my_count;
            if (my_count > cap) atomicOr(overflow, 1);
        }
    } else {
        my_base = base[chunk];
        my_count = counts[o];
    }
    // a thread owns 4 consecutive positions (one 32-bit word of the map), a wave one 256-id group: the four ranks inside
    // the group (<= 255) go to the rank table as one word, the group's base to grp_base; the slot entries of consecutive
    // set bytes are consecutive
    if (blockIdx.x == 0 && threadIdx.x == 0) *cap_out = cap;       // (the position gather may run as a call of its own)
    const int64_t p0 = (int64_t)chunk * RF_CHUNK + (int64_t)threadIdx.x * 4;
    uint32_t* wp = reinterpret_cast<uint32_t*>(rank + (size_t)chunk * RF_CHUNK) + threadIdx.x;
    const uint32_t w = eq_bytes(reinterpret_cast<const uint32_t*>(mark + (size_t)chunk * RF_CHUNK)[threadIdx.x], e4);
    const int mine = __popc(w);
    int incl = mine;                                                // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += t;
    }
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    int gbase = my_base;                                            // requests of this owner in front of the wave's group
    for (int q = 0; q < (int)(threadIdx.x >> 6); ++q) gbase += wsum[q];
    if ((threadIdx.x & 63) == 0) grp_base[(int64_t)chunk * (RF_CHUNK / 256) + (threadIdx.x >> 6)] = gbase;
    int rk = incl - mine;                                           // rank inside the group of this thread's first byte
    int s = gbase + rk;
    const int64_t slot0 = (int64_t)o * cap;
    const int64_t l0 = p0 - (int64_t)o * Lp;                        // local row index of this thread's first byte
    uint32_t rw = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const bool set = (w >> (8 * b)) & 1u;
        if (set && s < cap) send_pad[slot0 + s] = (int32_t)(l0 + b);   // past cap: overflow (flagged with the count)
        rw |= (uint32_t)rk << (8 * b);                              // (unset positions: never read)
        s += set ? 1 : 0;
        rk += set ? 1 : 0;
    }
    *wp = rw;
    // unused tail of the owner's slot: a valid row index (0), fetched and never looked at
    const int64_t used = my_count < cap ? my_count : cap;
    const int64_t tail = cap - used, per = (tail + cpo - 1) / cpo;
    const int64_t t0 = used + (int64_t)j * per, t1 = t0 + per < cap ? t0 + per : cap;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += RF_TPB) send_pad[slot0 + t] = 0;
}

template <typename IdT>
__global__ void __launch_bounds__(RF_TPB)
uniq_perm_pad_kernel(int64_t n, const IdT* __restrict__ ids, OwnerMap om, int64_t nfeat, int64_t Lp, HotSet hot,
                     const unsigned char* __restrict__ rank, const int32_t* __restrict__ grp_base,
                     const int64_t* __restrict__ cap_in, int32_t* __restrict__ perm_pad) {
    const int64_t cap = *cap_in;
    for (int64_t i = (int64_t)blockIdx.x * RF_TPB + threadIdx.x; i < n; i += (int64_t)gridDim.x * RF_TPB) {
        const uint64_t v = (uint64_t)(int64_t)ids[i];
        const uint32_t id = v >= (uint64_t)nfeat ? 0u : (uint32_t)v;
        if (hot.has(id)) {
            perm_pad[i] = (int32_t)(hot.base + (int64_t)id);
            continue;
        }
        uint32_t o, l;
        om.split(id, o, l);
        const int64_t p = (int64_t)o * Lp + l;
        const int64_t sl = (int64_t)grp_base[p >> 8] + rank[p];
        perm_pad[i] = (int32_t)((int64_t)o * cap + (sl < cap ? sl : 0));   // overflow: a valid row, the wrong one (step repeated)
    }
}

// The owner-side gather of the request list and the position gather of the de-duplicating route as ONE launch: the two
// only depend on the emit pass, are needed only by the fused block, and bound differently (the row gather by bytes, the
// position gather by the address unit: one cache line per lane), so their blocks — interleaved in the grid — overlap
// instead of queueing (one rank, headline step: 23 + 18 us in a row -> see profiles/r04_routing_fixed_protocol.txt).
//   rows:  out[j, :] = table[idx[j], :]   in W-float chunks (W = 4 / 2 / 1 by the divisibility of nemb)
//   perm:  uniq_perm_pad_kernel
template <typename IdT, int W>
__global__ void __launch_bounds__(RF_TPB)
gather_rows_perm_kernel(int64_t n_rows, int EW, const int32_t* __restrict__ idx, const float* __restrict__ table,
                        int64_t table_rows, float* __restrict__ out, int gather_blocks, int perm_blocks, int64_t n,
                        const IdT* __restrict__ ids, OwnerMap om, int64_t nfeat, int64_t Lp, HotSet hot,
                        const unsigned char* __restrict__ rank, const int32_t* __restrict__ grp_base,
                        const int64_t* __restrict__ cap_in, int32_t* __restrict__ perm_pad) {
    // block roles interleaved while both kinds last: even -> rows, odd -> positions
    const int b = blockIdx.x, paired = 2 * (gather_blocks < perm_blocks ? gather_blocks : perm_blocks);
    bool rows_role;
    int role_idx;
    if (b < paired) {
        rows_role = (b & 1) == 0;
        role_idx = b >> 1;
    } else {
        rows_role = gather_blocks > perm_blocks;
        role_idx = paired / 2 + (b - paired);
    }
    if (rows_role) {
        typedef float vecW __attribute__((ext_vector_type(W)));
        const int64_t total = n_rows * EW;
        const vecW* __restrict__ src = reinterpret_cast<const vecW*>(table);
        vecW* __restrict__ dst = reinterpret_cast<vecW*>(out);
        for (int64_t i = (int64_t)role_idx * RF_TPB + threadIdx.x; i < total; i += (int64_t)gather_blocks * RF_TPB) {
            const int64_t r = i / EW;
            const int c = (int)(i - r * EW);
            const uint32_t row = (uint32_t)idx[r];
            if constexpr (W == 1) dst[i] = src[(size_t)(row < (uint64_t)table_rows ? row : 0u) * EW + c];
            else dst[i] = src[(size_t)(row < (uint64_t)table_rows ? row : 0u) * EW + c];
        }
    } else {
        const int64_t cap = *cap_in;
        for (int64_t i = (int64_t)role_idx * RF_TPB + threadIdx.x; i < n; i += (int64_t)perm_blocks * RF_TPB) {
            const uint64_t v = (uint64_t)(int64_t)ids[i];
            const uint32_t id = v >= (uint64_t)nfeat ? 0u : (uint32_t)v;
            if (hot.has(id)) {
                perm_pad[i] = (int32_t)(hot.base + (int64_t)id);
                continue;
            }
            uint32_t o, l;
            om.split(id, o, l);
            const int64_t p = (int64_t)o * Lp + l;
            const int64_t sl = (int64_t)grp_base[p >> 8] + rank[p];
            perm_pad[i] = (int32_t)((int64_t)o * cap + (sl < cap ? sl : 0));
        }
    }
}

static int64_t rf_Lp(int R, int64_t nfeat) {
    const int64_t L = (nfeat + R - 1) / R;
    return (L + RF_CHUNK - 1) / RF_CHUNK * RF_CHUNK;
}

size_t shard_route_fixed_ws_bytes(int R, int64_t nfeat, int dedup) {
    if (!dedup) return 16;
    const int64_t P = (int64_t)R * rf_Lp(R, nfeat), nchunk = P / RF_CHUNK;
    return (size_t)P /* mark epochs */ + (size_t)P /* ranks */ + (size_t)(P / 256) * 4 /* group bases */ + (size_t)nchunk * 8 /* sums, base */ + 256 /* cap */;
}

// the position gather of the de-duplicating route on its own: perm_pad[i] = pos[p(id_i)] from the workspace a preceding
// armnet_shard_route_fixed(dedup = 1, perm_pad = NULL) left behind.  The request list (send_pad) does not depend on it, so
// the caller may run it on a side stream beside the index exchange / owner-side gather (armnet_hip/sharded.py).
int launch_shard_route_fixed_perm(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm_pad,
                                  const void* ws, size_t ws_bytes, HotSet hot, hipStream_t st) {
    if (R < 1 || R > RF_MAX_R) return ARMNET_ERR_UNSUPPORTED;
    if (hot.rows < 0 || hot.rows > nfeat || hot.base < 0) return ARMNET_ERR_BAD_ARG;
    if (hot.base + hot.rows >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (n == 0) return ARMNET_OK;
    const int64_t Lp = rf_Lp(R, nfeat), P = (int64_t)R * Lp;
    if (!ws || ws_bytes < shard_route_fixed_ws_bytes(R, nfeat, 1)) return ARMNET_ERR_BAD_ARG;
    const OwnerMap om = make_owner_map(R);
    const unsigned char* rank = reinterpret_cast<const unsigned char*>(ws) + P;
    const int32_t* grp_base = reinterpret_cast<const int32_t*>(rank + P);
    const int64_t* cap_in = reinterpret_cast<const int64_t*>(grp_base + P / 256 + 2 * (P / RF_CHUNK));
    const int gn = (int)((n + RF_TPB - 1) / RF_TPB < 4096 ? (n + RF_TPB - 1) / RF_TPB : 4096);
    if (id_type == ARMNET_ID_I64) uniq_perm_pad_kernel<int64_t><<<gn, RF_TPB, 0, st>>>(n, (const int64_t*)ids, om, nfeat, Lp, hot, rank, grp_base, cap_in, perm_pad);
    else uniq_perm_pad_kernel<int32_t><<<gn, RF_TPB, 0, st>>>(n, (const int32_t*)ids, om, nfeat, Lp, hot, rank, grp_base, cap_in, perm_pad);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

int launch_shard_gather_perm(int64_t n_rows, int E, const int32_t* idx, const float* table, int64_t table_rows, float* out,
                             int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm_pad, const void* ws,
                             size_t ws_bytes, HotSet hot, hipStream_t st) {
    if (R < 1 || R > RF_MAX_R) return ARMNET_ERR_UNSUPPORTED;
    if (hot.rows < 0 || hot.rows > nfeat || hot.base < 0) return ARMNET_ERR_BAD_ARG;
    if (hot.base + hot.rows >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (n_rows == 0 && n == 0) return ARMNET_OK;
    const int64_t Lp = rf_Lp(R, nfeat), P = (int64_t)R * Lp;
    if (!ws || ws_bytes < shard_route_fixed_ws_bytes(R, nfeat, 1)) return ARMNET_ERR_BAD_ARG;
    const OwnerMap om = make_owner_map(R);
    const unsigned char* rank = reinterpret_cast<const unsigned char*>(ws) + P;
    const int32_t* grp_base = reinterpret_cast<const int32_t*>(rank + P);
    const int64_t* cap_in = reinterpret_cast<const int64_t*>(grp_base + P / 256 + 2 * (P / RF_CHUNK));
    const bool a16 = ((uintptr_t)table % 16 == 0) && ((uintptr_t)out % 16 == 0), a8 = ((uintptr_t)table % 8 == 0) && ((uintptr_t)out % 8 == 0);
    const int W = (E % 4 == 0 && a16) ? 4 : (E % 2 == 0 && a8) ? 2 : 1;
    const int EW = E / W;
    const int64_t chunks = n_rows * EW;
    // ~4 chunks per thread for the rows, one lookup per thread and trip for the positions (as the stand-alone kernels)
    int gb = (int)((chunks + RF_TPB * 4 - 1) / (RF_TPB * 4) < 4096 ? (chunks + RF_TPB * 4 - 1) / (RF_TPB * 4) : 4096);
    int pb = (int)((n + RF_TPB - 1) / RF_TPB < 4096 ? (n + RF_TPB - 1) / RF_TPB : 4096);
    if (n_rows > 0 && gb < 1) gb = 1;
    if (n > 0 && pb < 1) pb = 1;
#define ARMNET_GRP(IdT, W_) \
    gather_rows_perm_kernel<IdT, W_><<<gb + pb, RF_TPB, 0, st>>>(n_rows, EW, idx, table, table_rows, out, gb, pb, n, (const IdT*)ids, \
                                                              om, nfeat, Lp, hot, rank, grp_base, cap_in, perm_pad)
    if (id_type == ARMNET_ID_I64) { if (W == 4) ARMNET_GRP(int64_t, 4); else if (W == 2) ARMNET_GRP(int64_t, 2); else ARMNET_GRP(int64_t, 1); }
    else { if (W == 4) ARMNET_GRP(int32_t, 4); else if (W == 2) ARMNET_GRP(int32_t, 2); else ARMNET_GRP(int32_t, 1); }
#undef ARMNET_GRP
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

int launch_shard_route_fixed(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap, int dedup,
                             int32_t* send_pad, int32_t* perm_pad, int32_t* counts, int32_t* overflow, int32_t* id_status,
                             void* ws, size_t ws_bytes, int epoch, HotSet hot, hipStream_t st) {
    if (R < 1 || R > RF_MAX_R) return ARMNET_ERR_UNSUPPORTED;
    if (hot.rows < 0 || hot.rows > nfeat || hot.base < 0) return ARMNET_ERR_BAD_ARG;
    if (hot.base + hot.rows >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if ((int64_t)R * cap >= ((int64_t)1 << 31) || n >= ((int64_t)1 << 31) || nfeat >= ((int64_t)1 << 32)) return ARMNET_ERR_UNSUPPORTED;
    const OwnerMap om = make_owner_map(R);
    if (!dedup) {
        // unused slot entries: row 0; the reservation counters start at 0.  One fill when the caller put counts right behind
        // send_pad (armnet_hip/sharded.py does): a fill of 32 bytes costs a launch like one of 13 MB
        if (counts == send_pad + (size_t)R * cap) {
            ARMNET_HIP_TRY(hipMemsetAsync(send_pad, 0, sizeof(int32_t) * ((size_t)R * cap + R), st));
        } else {
            ARMNET_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * R, st));
            ARMNET_HIP_TRY(hipMemsetAsync(send_pad, 0, sizeof(int32_t) * (size_t)R * cap, st));
        }
        if (n == 0) return ARMNET_OK;
        const int64_t grid = (n + RF_TPB * RF_PER - 1) / (RF_TPB * RF_PER);
        if (id_type == ARMNET_ID_I64)
            route_slots_kernel<int64_t><<<(int)grid, RF_TPB, 0, st>>>(n, (const int64_t*)ids, om, nfeat, cap, hot, send_pad, perm_pad, counts, overflow, id_status);
        else
            route_slots_kernel<int32_t><<<(int)grid, RF_TPB, 0, st>>>(n, (const int32_t*)ids, om, nfeat, cap, hot, send_pad, perm_pad, counts, overflow, id_status);
        ARMNET_LAUNCH_CHECK();
        return ARMNET_OK;
    }
    const int64_t Lp = rf_Lp(R, nfeat), P = (int64_t)R * Lp, nchunk = P / RF_CHUNK;
    if (P >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (!ws || ws_bytes < shard_route_fixed_ws_bytes(R, nfeat, 1)) return ARMNET_ERR_BAD_ARG;
    if (n == 0) ARMNET_HIP_TRY(hipMemsetAsync(counts, 0, sizeof(int32_t) * R, st));   // (otherwise the emit pass writes every count)
    unsigned char* mark = reinterpret_cast<unsigned char*>(ws);
    unsigned char* rank = mark + P;
    int32_t* grp_base = reinterpret_cast<int32_t*>(rank + P);
    int* sums = reinterpret_cast<int*>(grp_base + P / 256);
    int* base = sums + nchunk;
    int64_t* cap_out = reinterpret_cast<int64_t*>(base + nchunk);
    // epoch 0: a map in an unknown state — zero it and mark with 1; 1..255: the caller vouches that the map last saw a fill or
    // a smaller epoch (armnet_hip/sharded.py cycles 0, 2, 3, .., 255, 0, ..)
    if (epoch == 0) ARMNET_HIP_TRY(hipMemsetAsync(mark, 0, (size_t)P, st));
    const unsigned char ep = (unsigned char)(epoch == 0 ? 1 : epoch);
    const uint32_t e4 = 0x01010101u * ep;
    const int gn = (int)((n + RF_TPB - 1) / RF_TPB < 4096 ? (n + RF_TPB - 1) / RF_TPB : 4096);
    if (n > 0) {
        if (id_type == ARMNET_ID_I64) uniq_mark_bytes_kernel<int64_t><<<gn, RF_TPB, 0, st>>>(n, (const int64_t*)ids, om, nfeat, Lp, hot, mark, ep, id_status);
        else uniq_mark_bytes_kernel<int32_t><<<gn, RF_TPB, 0, st>>>(n, (const int32_t*)ids, om, nfeat, Lp, hot, mark, ep, id_status);
        ARMNET_LAUNCH_CHECK();
    }
    uniq_chunk_sums_kernel<<<(int)nchunk, RF_TPB, 0, st>>>(mark, e4, sums);
    ARMNET_LAUNCH_CHECK();
    const int cpo = (int)(Lp / RF_CHUNK);
    if (cpo <= 8192) {
        uniq_emit_kernel<true><<<(int)nchunk, RF_TPB, 0, st>>>(cpo, Lp, cap, mark, e4, rank, sums, counts, send_pad, grp_base, cap_out, overflow);
    } else {
        uniq_scan_kernel<<<1, 1024, 0, st>>>((int)nchunk, cpo, R, cap, sums, base, counts, overflow);
        ARMNET_LAUNCH_CHECK();
        uniq_emit_kernel<false><<<(int)nchunk, RF_TPB, 0, st>>>(cpo, Lp, cap, mark, e4, rank, base, counts, send_pad, grp_base, cap_out, overflow);
    }
    ARMNET_LAUNCH_CHECK();
    if (perm_pad) return launch_shard_route_fixed_perm(n, ids, id_type, R, nfeat, perm_pad, ws, ws_bytes, hot, st);
    return ARMNET_OK;                         // perm_pad == NULL: the caller runs armnet_shard_route_fixed_perm itself
}

}  // namespace armnet

using namespace armnet;

extern "C" int64_t armnet_shard_route_fixed_ws_bytes(int R, int64_t nfeat, int dedup) {
    if (R < 1 || nfeat <= 0) return -1;
    return (int64_t)shard_route_fixed_ws_bytes(R, nfeat, dedup);
}

extern "C" int armnet_shard_route_fixed(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap,
                                        int dedup, int32_t* send_pad, int32_t* perm_pad, int32_t* counts,
                                        int32_t* overflow, int32_t* id_status, void* workspace, int64_t ws_bytes,
                                        void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || cap < 1 || !send_pad || !counts || !overflow || (n > 0 && (!ids || (!perm_pad && !dedup))))
        return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_shard_route_fixed(n, ids, id_type, R, nfeat, cap, dedup, send_pad, perm_pad, counts, overflow, id_status,
                                    workspace, (size_t)ws_bytes, 0, HotSet{0, 0}, (hipStream_t)stream);
}

// The same with a caller-managed MARK EPOCH for the de-duplicating route (dedup != 0): epoch 0 = armnet_shard_route_fixed
// (the map is zeroed, marks are 1); epoch e in 2..255 = no fill, marks are e — valid when the previous call on this workspace
// used epoch 0 or a smaller e (a caller cycles 0, 2, 3, .., 255, 0, ..): the 1-byte-per-row map is then zeroed once per 255 steps.
extern "C" int armnet_shard_route_fixed_epoch(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap,
                                              int dedup, int32_t* send_pad, int32_t* perm_pad, int32_t* counts,
                                              int32_t* overflow, int32_t* id_status, void* workspace, int64_t ws_bytes,
                                              int epoch, void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || cap < 1 || !send_pad || !counts || !overflow || (n > 0 && (!ids || (!perm_pad && !dedup))))
        return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (epoch < 0 || epoch == 1 || epoch > 255) return ARMNET_ERR_BAD_ARG;
    return launch_shard_route_fixed(n, ids, id_type, R, nfeat, cap, dedup, send_pad, perm_pad, counts, overflow, id_status,
                                    workspace, (size_t)ws_bytes, epoch, HotSet{0, 0}, (hipStream_t)stream);
}

// Round 5 — hot-row replication (SURVEY.md §8e's third lever): ids < hot_rows (the head of a frequency-ordered id space) are
// held by every rank and never routed: perm_pad[i] = hot_base + id for them (the caller places its replicated hot rows at row
// hot_base of the buffer the fused block reads — sharded.py: right behind the R * cap received rows), they take no slot, no
// mark, no count.  hot_rows = 0 is armnet_shard_route_fixed_epoch.  The position gather that follows a dedup route with
// perm_pad = NULL must be given the same (hot_rows, hot_base): armnet_shard_route_fixed_perm_hot / armnet_shard_gather_perm_hot_f32.
extern "C" int armnet_shard_route_fixed_hot(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap,
                                            int dedup, int32_t* send_pad, int32_t* perm_pad, int32_t* counts,
                                            int32_t* overflow, int32_t* id_status, void* workspace, int64_t ws_bytes,
                                            int epoch, int64_t hot_rows, int64_t hot_base, void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || cap < 1 || !send_pad || !counts || !overflow || (n > 0 && (!ids || (!perm_pad && !dedup))))
        return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (epoch < 0 || epoch == 1 || epoch > 255) return ARMNET_ERR_BAD_ARG;
    return launch_shard_route_fixed(n, ids, id_type, R, nfeat, cap, dedup, send_pad, perm_pad, counts, overflow, id_status,
                                    workspace, (size_t)ws_bytes, epoch, HotSet{hot_rows, hot_base}, (hipStream_t)stream);
}

extern "C" int armnet_shard_route_fixed_perm_hot(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm_pad,
                                                 const void* workspace, int64_t ws_bytes, int64_t hot_rows, int64_t hot_base,
                                                 void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || (n > 0 && (!ids || !perm_pad))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_shard_route_fixed_perm(n, ids, id_type, R, nfeat, perm_pad, workspace, (size_t)ws_bytes,
                                         HotSet{hot_rows, hot_base}, (hipStream_t)stream);
}

extern "C" int armnet_shard_gather_perm_hot_f32(int64_t n_rows, int E, const int32_t* idx, const float* table, int64_t table_rows,
                                                float* out, int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                                                int32_t* perm_pad, const void* workspace, int64_t ws_bytes, int64_t hot_rows,
                                                int64_t hot_base, void* stream) {
    if (n_rows < 0 || n < 0 || E <= 0 || R < 1 || nfeat <= 0 || table_rows <= 0) return ARMNET_ERR_BAD_ARG;
    if ((n_rows > 0 && (!idx || !table || !out)) || (n > 0 && (!ids || !perm_pad))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (n_rows * (int64_t)E >= ((int64_t)1 << 40)) return ARMNET_ERR_UNSUPPORTED;
    return launch_shard_gather_perm(n_rows, E, idx, table, table_rows, out, n, ids, id_type, R, nfeat, perm_pad, workspace,
                                    (size_t)ws_bytes, HotSet{hot_rows, hot_base}, (hipStream_t)stream);
}

extern "C" int armnet_shard_route_fixed_perm(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm_pad,
                                             const void* workspace, int64_t ws_bytes, void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || (n > 0 && (!ids || !perm_pad))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_shard_route_fixed_perm(n, ids, id_type, R, nfeat, perm_pad, workspace, (size_t)ws_bytes, HotSet{0, 0},
                                         (hipStream_t)stream);
}

// the owner-side gather out[j, :] = table[idx[j], :] (idx: int32 local row indices of a request list; out-of-range -> row 0)
// and the position gather of armnet_shard_route_fixed_perm in one launch (see gather_rows_perm_kernel)
extern "C" int armnet_shard_gather_perm_f32(int64_t n_rows, int E, const int32_t* idx, const float* table, int64_t table_rows,
                                            float* out, int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                                            int32_t* perm_pad, const void* workspace, int64_t ws_bytes, void* stream) {
    if (n_rows < 0 || n < 0 || E <= 0 || R < 1 || nfeat <= 0 || table_rows <= 0) return ARMNET_ERR_BAD_ARG;
    if ((n_rows > 0 && (!idx || !table || !out)) || (n > 0 && (!ids || !perm_pad))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (n_rows * (int64_t)E >= ((int64_t)1 << 40)) return ARMNET_ERR_UNSUPPORTED;
    return launch_shard_gather_perm(n_rows, E, idx, table, table_rows, out, n, ids, id_type, R, nfeat, perm_pad, workspace,
                                    (size_t)ws_bytes, HotSet{0, 0}, (hipStream_t)stream);
}
