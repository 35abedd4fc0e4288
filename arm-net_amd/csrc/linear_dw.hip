// linear_dw.hip — the WEIGHT gradient of nn.Linear in the training head (models/layers.py:68-88 under train.py:108-114):
//      dW[n, k] = sum_b dY[b, n] * X[b, k]        (and db[n] = sum_b dY[b, n])
// on the CDNA4 bf16 matrix cores with fp32-equivalent numerics.  gfx950 only.  Round 6 (round-5 verdict, next 4b).
//
// Why a kernel of its own: the contraction runs over the SAMPLES (65 536 of them at bench.py's batch) and the result is a
// parameter-sized [256, 512] matrix — 128 output tiles for 256 CUs.  hipBLASLt's fp32 GEMM runs it as a split-K batched GEMM +
// partial sums (232 + 110 + 88 us per step for the head's three Linears, profiles/r05_train_step_times.txt) at the fp32
// vector rate; the eval head's operand split (mlp_head.hip) does not carry over because there the contraction index is
// contiguous in memory for both operands, here it is the ROW index of both.
//
// What makes it cheap here: the MFMA operand fragment of v_mfma_f32_32x32x16 is, per lane (i = l & 31, half = l >> 5), eight
// CONSECUTIVE contraction elements of row / column i.  With the contraction over samples that is eight consecutive ROWS of
// one COLUMN of a row-major matrix: a lane reads them with eight global_load_dword, each of which is coalesced ACROSS the
// wave (32 lanes = 32 adjacent columns = 128 contiguous bytes, two such runs per instruction).  No transpose, no LDS, no
// barrier: every wave is on its own.  A wave owns a 128 x 128 tile of dW (4 x 4 MFMA tiles, 256 accumulator registers, one
// wave per SIMD) over one contiguous RANGE of samples and adds its partial sums to dW with float atomics (coalesced: a
// row of the C layout is 32 adjacent floats); tiles x ranges fill the chip's 1 024 SIMDs.
// Operand split: bf16 x 3, six cross products (gradients are tiny — fp16 would need a range pass): x = xh + xm + xl exactly,
// smallest products first, fp32 accumulate in the matrix core.
#include "armnet_common.h"

namespace armnet {

typedef float dw_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 dw_bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t dw_u32x4 __attribute__((ext_vector_type(4)));

struct DwArgs {
    int64_t B, ldy, ldx, range;   // range: samples per wave (a multiple of 16)
    int N, K, tiles_n, tiles_k, S;
    const float *dY, *X;
    float *dW, *db;
};

struct DwPlanes { dw_u32x4 h, m, l; };

// 8 fp32 -> three packed bf16x8 planes; h + m + l == x exactly (truncating 8-bit slices of the significand)
__device__ __forceinline__ void dw_split3(const float (&x)[8], DwPlanes& p) {
    uint32_t hb[8], mb[8], lb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hb[i] = __float_as_uint(x[i]) & 0xffff0000u;
        const float r1 = x[i] - __uint_as_float(hb[i]);
        mb[i] = __float_as_uint(r1) & 0xffff0000u;
        lb[i] = __float_as_uint(r1 - __uint_as_float(mb[i]));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p.h[i] = __builtin_amdgcn_perm(hb[2 * i + 1], hb[2 * i], 0x07060302u);
        p.m[i] = __builtin_amdgcn_perm(mb[2 * i + 1], mb[2 * i], 0x07060302u);
        p.l[i] = __builtin_amdgcn_perm(lb[2 * i + 1], lb[2 * i], 0x07060302u);
    }
}

__device__ __forceinline__ dw_f32x16 dw_mfma(dw_u32x4 a, dw_u32x4 b, dw_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dw_bf16x8, a), __builtin_bit_cast(dw_bf16x8, b), c, 0, 0, 0);
}

template <int TN, int TK>
__global__ void __launch_bounds__(256, 1) linear_dw_kernel(DwArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, hf = lane >> 5;
    const int WT = a.tiles_n * a.tiles_k;
    const int w = (int)blockIdx.x * 4 + wave;
    const int tile = w % WT, rng = w / WT;
    if (rng >= a.S) return;
    const int tn = tile / a.tiles_k, tk = tile - tn * a.tiles_k;
    const int n0 = tn * 32 * TN, k0 = tk * 32 * TK;
    const int64_t b_lo = (int64_t)rng * a.range;
    const int64_t b_hi = b_lo + a.range < a.B ? b_lo + a.range : a.B;
    if (b_lo >= b_hi) return;

    // this lane's column of every operand tile (columns past the matrix read column 0 and are masked)
    const float* pa[TN];
    const float* pb[TK];
    bool oka[TN], okb[TK];
#pragma unroll
    for (int t = 0; t < TN; ++t) {
        const int col = n0 + 32 * t + m;
        oka[t] = col < a.N;
        pa[t] = a.dY + (oka[t] ? col : 0);
    }
#pragma unroll
    for (int t = 0; t < TK; ++t) {
        const int col = k0 + 32 * t + m;
        okb[t] = col < a.K;
        pb[t] = a.X + (okb[t] ? col : 0);
    }
    dw_f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float dbs[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) dbs[t] = 0.f;

    float ra[TN][8], rb[TK][8];
    // rows b + 8 hf + j of the lane's columns (rows past the range re-read its first row and are masked)
    auto load = [&](int64_t b) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = b + 8 * hf + j;
            const bool ok = row < b_hi;
            const int64_t r = ok ? row : b_lo;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const float v = pa[t][r * a.ldy];
                ra[t][j] = (ok && oka[t]) ? v : 0.f;
            }
#pragma unroll
            for (int t = 0; t < TK; ++t) {
                const float v = pb[t][r * a.ldx];
                rb[t][j] = (ok && okb[t]) ? v : 0.f;
            }
        }
    };
    load(b_lo);
    const bool want_db = a.db != nullptr && tk == 0;
    for (int64_t b = b_lo; b < b_hi; b += 16) {
        DwPlanes A[TN], Bp[TK];
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            if (want_db) {
#pragma unroll
                for (int j = 0; j < 8; ++j) dbs[t] += ra[t][j];
            }
            dw_split3(ra[t], A[t]);
        }
#pragma unroll
        for (int t = 0; t < TK; ++t) dw_split3(rb[t], Bp[t]);
        if (b + 16 < b_hi) load(b + 16);             // the next k-step's 64 loads fly under this one's 96 MFMAs
        // six cross products per tile, smallest terms first; 16 independent accumulator chains
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = dw_mfma(A[i].l, Bp[j].h, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = dw_mfma(A[i].m, Bp[j].m, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = dw_mfma(A[i].m, Bp[j].h, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = dw_mfma(A[i].h, Bp[j].l, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = dw_mfma(A[i].h, Bp[j].m, acc[i][j]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int j = 0; j < TK; ++j) acc[i][j] = dw_mfma(A[i].h, Bp[j].h, acc[i][j]);
    }
    // C layout of the 32x32 MFMA: lane (column m, half) holds rows (r & 3) + 8 (r >> 2) + 4 half of the tile
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int col = k0 + 32 * j + m;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = n0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (row < a.N && col < a.K) unsafeAtomicAdd(a.dW + (size_t)row * a.K + col, acc[i][j][r]);
            }
        }
    if (want_db) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const float s = dbs[t] + __shfl_xor(dbs[t], 32);
            if (hf == 0 && oka[t]) unsafeAtomicAdd(a.db + n0 + 32 * t + m, s);
        }
    }
}

}  // namespace armnet

using namespace armnet;

extern "C" int armnet_linear_dw_f32(int64_t B, int N, int K, const float* dY, int64_t ldy, const float* X, int64_t ldx,
                                    float* dW, float* db, void* stream) {
    if (B < 0 || N < 1 || K < 1 || ldy < N || ldx < K) return ARMNET_ERR_BAD_ARG;
    if (!dW) return ARMNET_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    ARMNET_HIP_TRY(hipMemsetAsync(dW, 0, (size_t)N * K * sizeof(float), st));
    if (db) ARMNET_HIP_TRY(hipMemsetAsync(db, 0, (size_t)N * sizeof(float), st));
    if (B == 0) return ARMNET_OK;
    if (!dY || !X) return ARMNET_ERR_BAD_ARG;
    DwArgs a{};
    a.B = B; a.N = N; a.K = K; a.ldy = ldy; a.ldx = ldx; a.dY = dY; a.X = X; a.dW = dW; a.db = db;
    a.tiles_n = (N + 127) / 128;
    a.tiles_k = (K + 127) / 128;
    const int64_t WT = (int64_t)a.tiles_n * a.tiles_k;
    const int64_t slots = (int64_t)device_cu_count() * 4;             // one wave per SIMD
    int64_t S = slots / WT;
    if (S < 1) S = 1;
    int64_t range = ((B + S - 1) / S + 15) / 16 * 16;
    if (range < 64) range = 64;                                      // a wave's 256 atomics per lane need some work in front of them
    S = (B + range - 1) / range;
    a.range = range;
    a.S = (int)S;
    const int64_t waves = WT * S;
    linear_dw_kernel<4, 4><<<(int)((waves + 3) / 4), 256, 0, st>>>(a);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}
