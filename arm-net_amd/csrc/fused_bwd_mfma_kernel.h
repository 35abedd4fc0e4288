// fused_bwd_mfma_kernel.h — backward of the fused ARM block on the CDNA4 matrix cores (gfx950), fp32.
// Template + launcher; instantiated per family in fused_bwd_mfma_*.hip.  Math as in fused_bwd.hip (the
// shape-agnostic kernel this one replaces where it applies); layouts as in fused_mfma_kernel.h.
//
// One WAVE owns one sample at a time (wave-private LDS, no block barrier between prologue and flush).
// Per sample the scaled rows X[f, :] are staged once; then per pass of 16 neurons:
//
//   MFMA #1   gates  G[f, o] = X[f, :] . q_fold[o, :]              (as the forward)
//   solve     p[o, :] = entmax / softmax of the row, normalised      (as the forward, one sample per wave)
//   MFMA #3   dW[f, o] = X[f, :] . ds[o, :],  ds = dz * z            same shape as #1, B operand from HBM:
//             a lane's 16-byte dz / z loads are exactly its B-operand k-steps
//   VALU      d_values += p * dW (per-wave register accumulators over all its samples);  dp = values * dW;
//             dg = J_entmax(p)^T dp (entmax.py:70-80: one 4-lane-group reduction);  w = p * values
//   MFMA #4   dq_fold^T[e, o] += sum_f X[f, e] dg[o, f]              same shape as the forward's #2 (dg in the
//             C layout IS its B operand); the accumulator tile lives over all samples of the wave
//   MFMA #5   dx[f, e] += sum_o w[o, f] ds[o, e] + dg[o, f] q_fold[o, e]     contraction over the NEURONS, which
//             the C layouts keep in the low lane bits: w and dg go through a transposing LDS buffer
//             ([o][tile row], 16-byte writes, 4-byte operand reads); ds is read back from LDS in k = o form.
//             The dx tiles stay in accumulators over all passes.
//   scatter   d_table[id[f], :] += dx[f, :] * val[f]                global float atomics (64-byte runs)
//
// MFMAs per sample and pass (nemb = 16, nfield = 39): 12 + 12 + 10 + 24 = 58 (forward: 20).
#pragma once
#include "fused_mfma_kernel.h"

namespace armnet {

// Passes (16 neurons each) per launch: the d_values / d_qfold accumulators of a launch's neuron slice stay in
// registers over all samples of a wave (LDS float atomics from 4 waves on the same addresses cost more than the rest
// of the kernel: measured 888 -> 368 us), so the slice is what the register file holds beside the dx tiles:
// 4 passes for nemb <= 16, 2 above.  Wider blocks run as several launches, each of which
// re-stages the rows and scatters its part of dx.
#ifndef ARMNET_BWD_BLOCKS_PER_CU
#define ARMNET_BWD_BLOCKS_PER_CU 2      // = waves per SIMD the register allocator targets
#endif
#ifndef ARMNET_BWD_E16_PASSES
#define ARMNET_BWD_E16_PASSES 4
#endif
#ifndef ARMNET_BWD_E64_PASSES
#define ARMNET_BWD_E64_PASSES 2    // measured: 2 passes with ~128 B of scratch beat 1 pass without (1.42 vs 1.74 ms)
#endif
// nemb 65..128 (round 4): 8 accumulator tiles per 16 neurons and per 16 tile rows — one pass per launch, one wave per SIMD
// (the 512-register budget holds the dx tiles of up to 32 fields beside the staged rows and dz / z)
constexpr int bwd_passes(int E) { return E >= 128 ? 1 : E >= 64 ? ARMNET_BWD_E64_PASSES : E > 16 ? 2 : ARMNET_BWD_E16_PASSES; }
constexpr int bwd_blocks_per_cu(int E) { return E >= 128 ? 1 : ARMNET_BWD_BLOCKS_PER_CU; }
// the siblings at nemb 33..64: one pass per launch — their second dx accumulator set takes the registers of the second pass
constexpr int bwd_passes_model(int E, int model) { return (model != MODEL_ARM && E >= 64) ? 1 : bwd_passes(E); }

// MODEL_GC_ARM (models/gc_arm.py:82-95, round 4): the same kernel for GC-ARM's block
//     arm[b,o,:] = sum_f w[b,o,f] y[b,f,:],   y = emb_bn(exp(x)) = emb_scale[f] * exp(x[b,f,:]) + emb_shift[f]   (no outer exp)
//     w = entmax(g + sum_f g) * values        (the global context shifts every gate of a row by the same amount: the sparse
//                                              map is shift-invariant, its Jacobian's rows sum to zero, and so does the
//                                              context's gradient — analytically 0, rounding noise in the reference)
//   ds = dz (not dz * z);  dW = y . ds with y formed from the staged rows as the forward does;  the two halves of MFMA #5
//   go to different places: dg . q_fold is the gradient of x and is scattered into the table gradient here, w . ds is the
//   gradient of y and is WRITTEN to d_y [B,F,E]: emb_bn runs on batch statistics in training, its backward needs the sums
//   of d_y and d_y * y_hat over the whole batch before any of it can reach the table (the caller's BatchNorm passes).
//
// MODEL_AFN (models/afn.py:56-69): z[b,o,:] = exp(sum_f W[o,f] l[b,f,:] + bias[o]),  l = emb_bn(log(x)) = emb_scale[f] * log(x) +
// emb_shift[f], W = afn.weight (BwdArgs.values).  No gates, no sparse map, no q_fold: per pass MFMA #3 (dW[f,o] = l . ds, summed
// over the wave's samples like d_values), the bias gradient (row sums of ds) and the first half of MFMA #5 (dl = W^T ds, written
// to d_y like GC-ARM's: emb_bn's backward needs whole-batch sums).  Nothing is scattered here (BwdArgs.d_table unused).
struct BwdExtra {
    const float* emb_scale;   // [F]
    const float* emb_shift;   // [F]
    float* d_y;               // [B,F,E]
    int accumulate;           // 0: d_y = ..., 1: d_y += ... (a later neuron slice of the same step)
    float* d_bias;            // AFN: [O] +=
};

template <int E, int NQ, int MODE, int SRC, int MODEL = MODEL_ARM>
__global__ void __launch_bounds__(256, bwd_blocks_per_cu(E)) fused_bwd_mfma_kernel(BwdArgs a, BwdExtra gx) {
    constexpr bool GC = (MODEL == MODEL_GC_ARM);
    constexpr bool AFN = (MODEL == MODEL_AFN);
    constexpr bool DY = GC || AFN;            // the gradient of the embedding BatchNorm's output goes to gx.d_y
    constexpr int NTILE = (NQ + 3) / 4;       // 16-row tiles per sample (last one may be half pad)
    constexpr int ROWS = NTILE * 16;
    constexpr int ES = E + 4;                 // LDS row stride of X, ds, q_fold (floats)
    constexpr int CF = 4, CH = E / CF, RPI = 64 / CH, NI = ROWS / RPI;     // 16-byte staging chunks
    constexpr int EB = E / 16, NP = NQ / 2;
    constexpr int RS = ROWS + 4;              // row stride of the transposing buffers
    constexpr int FP = 4 * NQ;                // nfield padded
    constexpr int XT = ROWS * ES, RED = 128, TW = 16 * RS, DSL = 16 * ES, IDV = 2 * ROWS;
    constexpr int WAVE_FLOATS = XT + RED + TW + DSL + IDV;
    static_assert(E % 16 == 0 && ROWS % RPI == 0 && NQ % 2 == 0, "shape");
    using RowT = f32x4;
    using RowTU = f32x4u;                     // as read from global memory

    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int F = a.F, O = a.O, Er = a.E;
    const int O_all = a.O_all ? a.O_all : O;
    const int NT = (O + 15) / 16, OP = NT * 16;
    float* xt = lds_all + wave * WAVE_FLOATS;
    float* red = xt + XT;
    float* tw = red + RED;                    // [16 o][RS]  transposing buffer: w = p * values, then dg
    float* dsl = tw + TW;                     // [16 o][ES]  ds = dz * z of this pass
    uint32_t* idl = reinterpret_cast<uint32_t*>(dsl + DSL);   // [ROWS] table row of each tile row (~0u: pad row)
    float* vll = dsl + DSL + ROWS;            // [ROWS] value of each tile row
    // block-shared
    float* p_bq = lds_all + 4 * WAVE_FLOATS;  // [NT][EB][64] f32x4   q_fold as the B operand of MFMA #1
    float* p_vv = p_bq + NT * EB * 64 * 4;    // [NT][NP][64] f32x2   values in the C layout
    float* qfl = p_vv + NT * NP * 64 * 2;     // [OP][ES]            q_fold, plain (B operand of MFMA #5)
    float* p_cf = qfl + OP * ES;              // [OP] f32x4 {A, B, C, -}: dz = A * dz_in + C * z + B (BatchNorm backward)
    [[maybe_unused]] float* p_x = p_cf + OP * 4;   // GC-ARM / AFN: [ROWS] f32x2 {emb_scale, emb_shift} per field (0 past nfield)
    // block accumulators of the final flush: they ALIAS the wave-private regions (used only after every wave is done)
    float* acc_dv = lds_all;                  // [OP][FP]
    float* acc_dq = acc_dv + OP * FP;         // [OP][E]
    [[maybe_unused]] float* acc_db = acc_dq + OP * E;   // AFN: [OP]

    const int Bi = (int)a.B;
    const int nwaves = (int)gridDim.x * 4;

    // ---- staging geometry (one sample per wave): fused_mfma_kernel.h with SPW = 1 -------------------
    const int chunk = lane % CH;
    // full / partial (read from the row's last 16 bytes, rotated into place) / padding chunks: fused_mfma_kernel.h
    const int rem = Er & 3;
    const bool chunk_part = rem != 0 && chunk == (Er >> 2);
    const bool chunk_ok = (chunk + 1) * CF <= Er || chunk_part;
    int fld[NI];         // field of the row this lane stages in instruction n (0 for a pad row)
    bool pad[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int row = n * RPI + lane / CH;
        const int q = 4 * (row >> 4) + (row & 3);
        const int f = 4 * q + ((row & 15) >> 2);
        pad[n] = !(q < NQ && f < F);
        fld[n] = pad[n] ? 0 : f;
    }
    const float padneg_a = ((4 * (NQ - 1) + g) >= F) ? -INFINITY : 0.f;
    const float padneg_b = (NQ >= 2 && (4 * (NQ - 2) + g) >= F) ? -INFINITY : 0.f;
    const bool two_pad = F <= 4 * (NQ - 1);
    const uint32_t id_max = (uint32_t)a.nfeat - 1u;
    const uint32_t row_bytes = chunk_ok ? (uint32_t)Er * 4u : 0u;
    const char* row_base = chunk_ok ? reinterpret_cast<const char*>(a.table) + (chunk_part ? (Er - 4) * 4 : chunk * 16)
                                    : reinterpret_cast<const char*>(kZeroRow);
    constexpr int XQ = 4 * NTILE - NQ;
    constexpr int NZ = ((7 + 4 * XQ) * (E / 4) + 63) / 64;
    const int npf = 4 * NQ - F;
    int zoff[NZ];
#pragma unroll
    for (int m = 0; m < NZ; ++m) {
        const int idx = lane + 64 * m;
        const int cc = idx % (E / 4), kk = idx / (E / 4);
        int q, gg;
        if (kk < npf) {
            const int f = F + kk;
            q = f >> 2;
            gg = f & 3;
        } else {
            const int x = kk - npf;
            q = NQ + (x >> 2);
            gg = x & 3;
        }
        zoff[m] = (q < 4 * NTILE) ? ((q >> 2) * 16 + 4 * gg + (q & 3)) * ES + 4 * cc : -1;
    }
    const bool full_rows = (Er == E);
    // ablation switches for profiling (tools/bwd_bench.py --flags against a `make EXTRA=-DARMNET_DEV_FLAGS` build)
#ifdef ARMNET_DEV_FLAGS
    const bool dbg_no_scatter = (a.flags & 0x400u) != 0;   // skip the d_table atomics
    const bool dbg_one_pass = (a.flags & 0x2000u) != 0;    // only the first 16-neuron pass
    const bool dbg_hot_rows = (a.flags & 0x200u) != 0;     // fold ids into 1024 rows
    const bool dbg_no_flush = (a.flags & 0x4000u) != 0;    // skip the block's d_values / d_qfold atomics
#else
    constexpr bool dbg_no_scatter = false, dbg_one_pass = false, dbg_hot_rows = false, dbg_no_flush = false;
#endif

    // ---- block prologue: parameters of this neuron slice, zeroed accumulators -------------------------
    for (int i = threadIdx.x; !AFN && i < NT * EB * 64; i += 256) {
        const int l = i & 63, kb = (i >> 6) % EB, nt = (i >> 6) / EB;
        const int o = 16 * nt + (l & 15);
        const int e0 = 16 * kb + 4 * (l >> 4);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (o < O)
            for (int r = 0; r < 4; ++r)
                if (e0 + r < Er) v[r] = a.q_fold[(size_t)o * Er + e0 + r];
        *reinterpret_cast<f32x4*>(p_bq + i * 4) = v;
    }
    for (int i = threadIdx.x; i < NT * NP * 64; i += 256) {
        const int l = i & 63, jp = (i >> 6) % NP, nt = (i >> 6) / NP;
        const int o = 16 * nt + (l & 15);
        const int f0 = 4 * (2 * jp) + (l >> 4), f1 = f0 + 4;
        f32x2 v;
        v[0] = (o < O && f0 < F) ? a.values[(size_t)o * F + f0] : 0.f;
        v[1] = (o < O && f1 < F) ? a.values[(size_t)o * F + f1] : 0.f;
        *reinterpret_cast<f32x2*>(p_vv + i * 2) = v;
    }
    for (int i = threadIdx.x; !AFN && i < OP * ES; i += 256) {
        const int o = i / ES, e = i - o * ES;
        qfl[i] = (o < O && e < Er) ? a.q_fold[(size_t)o * Er + e] : 0.f;
    }
    if constexpr (GC) {
        for (int i = threadIdx.x; i < ROWS; i += 256)
            *reinterpret_cast<f32x2*>(p_x + 2 * i) = i < F ? f32x2{gx.emb_scale[i], gx.emb_shift[i]} : f32x2{0.f, 0.f};
    }
    if constexpr (AFN) {                      // log(x) = log2(x) * ln 2
        for (int i = threadIdx.x; i < ROWS; i += 256)
            *reinterpret_cast<f32x2*>(p_x + 2 * i) =
                i < F ? f32x2{gx.emb_scale[i] * 0.693147182464599609375f, gx.emb_shift[i]} : f32x2{0.f, 0.f};
    }
    for (int i = threadIdx.x; i < OP; i += 256) {
        const bool on = a.bn_a != nullptr && i < O;
        *reinterpret_cast<f32x4*>(p_cf + 4 * i) = f32x4{on ? a.bn_a[i] : 1.0f, on ? a.bn_b[i] : 0.f, on ? a.bn_c[i] : 0.f, 0.f};
    }
    __syncthreads();

    const float am1 = a.cfg.am1;
    const float rr = a.cfg.r, rm1 = a.cfg.r - 1.0f;
    const float invF = 1.0f / (float)F;
    const float tau_off = a.cfg.tau_hi_off;
    const float L2E = 1.44269502162933349609375f;
    const float two_m_alpha = 2.0f - a.alpha;

    // per-wave accumulators over all its samples: d_values in the C layout, d_qfold^T as MFMA #4's accumulator
    constexpr int NTS = bwd_passes_model(E, MODEL);
    f32x2 dvacc[NTS][NP];                     // pairs (2jp, 2jp+1) like the gate registers
    f32x4 dqacc[NTS][EB];
    [[maybe_unused]] float dbacc[NTS];        // AFN: this lane's part of d_bias[16 nt + c]
#pragma unroll
    for (int n = 0; n < NTS; ++n) {
        dbacc[n] = 0.f;
#pragma unroll
        for (int jp = 0; jp < NP; ++jp) dvacc[n][jp] = f32x2{0.f, 0.f};
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) dqacc[n][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- software pipeline over the wave's samples: rows of the next sample and ids / values of the one after are
    //      in flight while the current one computes; dz / z of the next pass (or of the next sample's first pass)
    //      are fetched one pass ahead.  Samples past the end re-read the last one (never used).
    uint32_t idC[NI], idN[NI];
    float vC[NI], vN[NI];
    RowT rwC[NI];
    f32x4 zN[EB], dN[EB];
    auto fetch_ids = [&](int bb, uint32_t (&idr)[NI], float (&vr)[NI]) {
        const size_t e0 = (size_t)(bb < Bi ? bb : Bi - 1) * F;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            uint32_t lo, hi = 0u;
            if constexpr (SRC == 0) {
                const uint2 w = reinterpret_cast<const uint2*>(a.ids)[e0 + fld[n]];
                lo = w.x;
                hi = w.y;
            } else {
                lo = reinterpret_cast<const uint32_t*>(a.ids)[e0 + fld[n]];
            }
            idr[n] = (hi != 0u || lo > id_max) ? 0u : lo;          // the forward already raised on bad ids
            if (dbg_hot_rows) idr[n] &= 1023u;
            vr[n] = a.vals[e0 + fld[n]];
        }
    };
    auto fetch_rows = [&]() {
#pragma unroll
        for (int n = 0; n < NI; ++n) rwC[n] = *reinterpret_cast<const RowTU*>(row_base + (size_t)idC[n] * row_bytes);
    };
    auto fetch_zd = [&](int bb, int nt) {
        const int o = 16 * nt + c;
        const size_t zo = ((size_t)(bb < Bi ? bb : Bi - 1) * O_all + o) * (size_t)Er + 4 * g;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb) {
            zN[eb] = f32x4{0.f, 0.f, 0.f, 0.f};
            dN[eb] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (o < O) {
                const int e0 = 16 * eb + 4 * g;
                if (full_rows || e0 + 4 <= Er) {
                    zN[eb] = *reinterpret_cast<const f32x4u*>(a.z + zo + 16 * eb);
                    dN[eb] = *reinterpret_cast<const f32x4u*>(a.dz + zo + 16 * eb);
                } else if (e0 < Er) {
                    // the row ends inside this lane's 4 floats: read the row's LAST 4 (in bounds) and keep the tail;
                    // rotated into place when consumed (zd_rotate)
                    zN[eb] = *reinterpret_cast<const f32x4u*>(a.z + zo - 4 * g + (Er - 4));
                    dN[eb] = *reinterpret_cast<const f32x4u*>(a.dz + zo - 4 * g + (Er - 4));
                }
            }
        }
    };
    const int b_first = (int)blockIdx.x * 4 + wave;
    const int npass = dbg_one_pass ? 1 : NT;
    // prefetch depth by register budget: 2 = rows of the next sample + ids of the one after + dz/z one pass ahead
    // (nemb <= 16), 1 = ids of the next sample + dz/z one pass ahead (nemb <= 32), 0 = nothing carried (nemb = 64)
    constexpr int PF = (E <= 16 && NQ <= 6) ? 2 : (E <= 16 || (E <= 32 && NQ <= 8)) ? 1 : 0;
    // PF == 1, LATE (round 4): the next sample's rows are requested BEHIND the passes, in front of the scatter — their registers are
    // live only while the dx atomics go out, not across the passes (where depth 2 does not fit), and the gather no longer waits
    // at the top of every sample
#ifdef ARMNET_BWD_NO_LATE_ROWS
    constexpr bool LATE = false;
#else
    constexpr bool LATE = (PF == 1);
#endif
    if (b_first < Bi) {
        if constexpr (PF == 2 || LATE) {
            fetch_ids(b_first, idC, vC);
            fetch_zd(b_first, 0);
            fetch_rows();
            fetch_ids(b_first + nwaves, idN, vN);
        } else if constexpr (PF == 1) {
            fetch_ids(b_first, idN, vN);
            fetch_zd(b_first, 0);
        }
    }

    for (int b = b_first; b < Bi; b += nwaves) {
        // ---- stage the sample's rows (scaled) into the wave's tile; remember id / value per tile row ----
        wave_lds_fence();
        if constexpr (PF == 1 && !LATE) {
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                idC[n] = idN[n];
                vC[n] = vN[n];
            }
            fetch_rows();
            fetch_ids(b + nwaves, idN, vN);
        } else if constexpr (PF == 0) {
            fetch_ids(b, idC, vC);
            fetch_rows();
        }
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int row = n * RPI + lane / CH;
            RowT r = rwC[n] * vC[n];
            {
                if (rem != 0) {                                         // kernel-uniform
                    const f32x4 t = rem == 1 ? f32x4{r[3], 0.f, 0.f, 0.f}
                                  : rem == 2 ? f32x4{r[2], r[3], 0.f, 0.f} : f32x4{r[1], r[2], r[3], 0.f};
                    r = chunk_part ? t : r;
                }
            }
            *reinterpret_cast<RowT*>(xt + row * ES + chunk * CF) = r;
            if (chunk == 0) {
                idl[row] = pad[n] ? 0xffffffffu : idC[n];
                vll[row] = vC[n];
            }
        }
#pragma unroll
        for (int m = 0; m < NZ; ++m)
            if (zoff[m] >= 0) *reinterpret_cast<f32x4*>(xt + zoff[m]) = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (PF == 2) {
            // next sample's rows (its ids arrived during the previous sample), ids / values of the one after
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                idC[n] = idN[n];
                vC[n] = vN[n];
            }
            fetch_rows();
            fetch_ids(b + 2 * nwaves, idN, vN);
        }
        wave_lds_fence();

        f32x4 cdx[NTILE][EB];
        [[maybe_unused]] f32x4 cdy[DY ? NTILE : 1][EB];              // GC-ARM / AFN: the gradient of emb_bn's output
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int eb = 0; eb < EB; ++eb) {
                cdx[t][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
                if constexpr (DY) cdy[t][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

#pragma unroll
        for (int nt = 0; nt < NTS; ++nt) {
            if (nt >= npass) break;                                      // wave-uniform
            // ---- ds = dz * z of this lane's neuron: the B operand of MFMA #3 (fetched one pass ahead) ----------
            f32x4 ds4[EB];
            if constexpr (PF == 0) fetch_zd(b, nt);
            {
                const f32x4 cf = *reinterpret_cast<const f32x4*>(p_cf + 4 * (16 * nt + c));
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    if (rem != 0) {                                     // kernel-uniform: partial last chunk of z / dz
                        const bool part = (16 * eb + 4 * g < Er) && (16 * eb + 4 * g + 4 > Er);
                        const f32x4 zz = zN[eb], dd = dN[eb];
                        const f32x4 zt = rem == 1 ? f32x4{zz[3], 0.f, 0.f, 0.f}
                                       : rem == 2 ? f32x4{zz[2], zz[3], 0.f, 0.f} : f32x4{zz[1], zz[2], zz[3], 0.f};
                        const f32x4 dt = rem == 1 ? f32x4{dd[3], 0.f, 0.f, 0.f}
                                       : rem == 2 ? f32x4{dd[2], dd[3], 0.f, 0.f} : f32x4{dd[1], dd[2], dd[3], 0.f};
                        zN[eb] = part ? zt : zz;
                        dN[eb] = part ? dt : dd;
                    }
                    f32x4 dzv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) dzv[r] = fmaf(cf[0], dN[eb][r], fmaf(cf[2], zN[eb][r], cf[1]));
                    if constexpr (GC) {
                        // no outer exp (gc_arm.py:92-94): ds = dz.  Columns past nemb must not carry the BatchNorm shift
                        // into the contraction over e (padding NEURONS hold {1, 0, 0} coefficients and dz = 0)
#pragma unroll
                        for (int r = 0; r < 4; ++r) dzv[r] = (16 * eb + 4 * g + r < Er) ? dzv[r] : 0.f;
                        ds4[eb] = dzv;
                    } else {
                        ds4[eb] = dzv * zN[eb]; // padding lanes (neuron >= O, e >= nemb) hold z = 0: ds = 0 whatever the shift
                    }
                }
            }
            if constexpr (PF >= 1) {
                if (nt + 1 < npass) fetch_zd(b, nt + 1);
                else fetch_zd(b + nwaves, 0);
            }
            if constexpr (AFN) {
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) dbacc[nt] += (ds4[eb][0] + ds4[eb][1]) + (ds4[eb][2] + ds4[eb][3]);
            }
            // ---- MFMA #1: gates -------------------------------------------------------------------------
            f32x4 c1[NTILE];
#pragma unroll
            for (int kb = 0; !AFN && kb < EB; ++kb) {
                const f32x4 bq = *reinterpret_cast<const f32x4*>(p_bq + ((nt * EB + kb) * 64 + lane) * 4);
                f32x4 av[NTILE];
#pragma unroll
                for (int t = 0; t < NTILE; ++t)
                    av[t] = *reinterpret_cast<const f32x4*>(xt + (16 * t + c) * ES + 16 * kb + 4 * g);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = 0; t < NTILE; ++t) {
                        if (kb == 0 && kk == 0)
                            c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], bq[kk], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            c1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], bq[kk], c1[t], 0, 0, 0);
                    }
            }
#define XG(j) c1[(j) >> 2][(j) & 3]
#define XP_GET(jp) (f32x2{XG(2 * (jp)), XG(2 * (jp) + 1)})
#define XP_SET(jp, v)            \
    do {                         \
        const f32x2 _v = (v);    \
        XG(2 * (jp)) = _v[0];    \
        XG(2 * (jp) + 1) = _v[1]; \
    } while (0)
            const float* vv_base = p_vv + (nt * NP * 64 + lane) * 2;
#define VV(jp) (*reinterpret_cast<const f32x2*>(vv_base + (jp) * 128))

            if constexpr (AFN) {                     // the [neurons, fields] weights are afn.weight itself
#pragma unroll
                for (int t = 0; t < NTILE; ++t) c1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NQ; ++j) XG(j) = VV(j >> 1)[j & 1];
            }
            // ---- sparse map: p (normalised) left in the gate registers -------------------------------------
            wave_lds_fence();
            float tau = 0.f, Ssum = 1.0f;
            [[maybe_unused]] float tau_hi = 0.f;
            if constexpr (!AFN) {
                f32x2 sm2;
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) {
                    f32x2 x = XP_GET(jp);
                    if constexpr (MODE != SOLVE_MICHELOT && MODE != SOLVE_SOFTMAX) {
                        x *= f32x2{am1, am1};
                        XP_SET(jp, x);
                    }
                    sm2 = jp == 0 ? x : sm2 + x;
                }
                XG(NQ - 1) += padneg_a;
                if (two_pad) XG(NQ - 2) += padneg_b;
                float mx = cmax2(XG(0), XG(1));             // compiler-visible first readers of the accumulators
#pragma unroll
                for (int j = 2; j + 1 < NQ; j += 2) mx = cmax3(mx, XG(j), XG(j + 1));
                red_write(red, 0, lane, mx, sm2[0] + sm2[1]);
                wave_lds_fence();
                const Red2 r = red_read(red, 0, c);
                mx = vmax2(vmax3(r.g0[0], r.g1[0], r.g2[0]), r.g3[0]);
                float sm = (r.g0[1] + r.g1[1]) + (r.g2[1] + r.g3[1]);
                if constexpr (GC) {                         // global context: fused_mfma_kernel.h, gc_arm.py:37-41
                    const float gcx = sm;
#pragma unroll
                    for (int j = 0; j < NQ; ++j) XG(j) += gcx;
                    mx += gcx;
                    sm = fmaf((float)F, gcx, sm);
                }
                if constexpr (MODE == SOLVE_SOFTMAX) tau = mx + (sm - sm);
                else if constexpr (MODE == SOLVE_BISECT) { tau = (mx - 1.0f) + (sm - sm); tau_hi = mx - tau_off; }
                else tau = vmax2(mx - 1.0f, fmaf(sm, invF, -tau_off)) + (sm - sm);
            }
            if constexpr (!AFN) {
            if constexpr (MODE == SOLVE_SOFTMAX) {
                wave_lds_fence();
                float S = 0.f;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const float p = __builtin_amdgcn_exp2f((XG(j) - tau) * L2E);
                    S += p;
                    XG(j) = p;
                }
                red_write(red, 0, lane, S, 0.f);
                wave_lds_fence();
                const Red2 q = red_read(red, 0, c);
                Ssum = (q.g0[0] + q.g1[0]) + (q.g2[0] + q.g3[0]);
            } else if constexpr (MODE == SOLVE_BISECT) {
                // the reference's bisection, statement for statement (fused_mfma_kernel.h; utils/entmax.py:49-64)
                f32x2 pkeep[NP];
                auto eval = [&](float t_at) -> float {
                    const f32x2 tk = {t_at, t_at};
                    f32x2 S2;
#pragma unroll
                    for (int jp = 0; jp < NP; ++jp) {
                        const f32x2 t = pk_sub_clamp01(XP_GET(jp), tk);
                        f32x2 pv;
                        pv[0] = __builtin_amdgcn_exp2f(rr * __builtin_amdgcn_logf(t[0]));
                        pv[1] = __builtin_amdgcn_exp2f(rr * __builtin_amdgcn_logf(t[1]));
                        pkeep[jp] = pv;
                        S2 = jp == 0 ? pv : S2 + pv;
                    }
                    return S2[0] + S2[1];
                };
                wave_lds_fence();
                red_write(red, 0, lane, eval(tau), 0.f);
                wave_lds_fence();
                {
                    const Red2 r = red_read(red, 0, c);
                    Ssum = (r.g0[0] + r.g1[0]) + (r.g2[0] + r.g3[0]);
                }
                const float f_lo = Ssum - 1.0f;
                float dm = tau_hi - tau;
                for (int it = 0; it < a.cfg.n_iter; ++it) {
                    wave_lds_fence();
                    dm *= 0.5f;
                    const float tm = tau + dm;
                    const bool moving = !(tm == tau);
                    red_write(red, 0, lane, eval(tm), 0.f);
                    wave_lds_fence();
                    const Red2 r = red_read(red, 0, c);
                    Ssum = (r.g0[0] + r.g1[0]) + (r.g2[0] + r.g3[0]);
                    tau = ((Ssum - 1.0f) * f_lo >= 0.f) ? tm : tau;
                    if (!__builtin_amdgcn_ballot_w64(moving)) break;        // settled everywhere: later steps repeat this one
                }
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) XP_SET(jp, pkeep[jp]);
            } else {
                f32x2 pkeep[MODE == SOLVE_NEWTON ? NP : 1];
                for (int it = 0; it < kNewtonMaxIter; ++it) {
                    wave_lds_fence();
                    const f32x2 tk = {tau, tau};
                    f32x2 S2, D2;
#pragma unroll
                    for (int jp = 0; jp < NP; ++jp) {
                        const f32x2 t = pk_sub_clamp01(XP_GET(jp), tk);
                        f32x2 sv, dv;
                        if constexpr (MODE == SOLVE_MICHELOT) {
                            sv = t;
                            dv = pk_mul_clamp01(t, f32x2{0x1p120f, 0x1p120f});
                        } else if constexpr (MODE == SOLVE_NEWTON15) {
                            sv = t * t;
                            dv = t;
                        } else {
                            f32x2 u;
                            u[0] = __builtin_amdgcn_exp2f(rm1 * __builtin_amdgcn_logf(t[0]));
                            u[1] = __builtin_amdgcn_exp2f(rm1 * __builtin_amdgcn_logf(t[1]));
                            sv = u * t;
                            dv = u;
                            pkeep[jp] = sv;
                        }
                        S2 = jp == 0 ? sv : S2 + sv;
                        D2 = jp == 0 ? dv : D2 + dv;
                    }
                    red_write(red, 0, lane, S2[0] + S2[1], D2[0] + D2[1]);
                    wave_lds_fence();
                    const Red2 r = red_read(red, 0, c);
                    const f32x2 sd = (r.g0 + r.g1) + (r.g2 + r.g3);
                    float Dv = sd[1];
                    if constexpr (MODE == SOLVE_NEWTON15) Dv *= 2.0f;
                    if constexpr (MODE == SOLVE_NEWTON) Dv *= rr;
                    Ssum = sd[0];
                    const float f = sd[0] - 1.0f;
                    const float tn = fmaf(f, __builtin_amdgcn_rcpf(Dv), tau);
                    const bool act = (f > kNewtonTol) && (tn > tau);
                    tau = act ? tn : tau;
                    if (!__builtin_amdgcn_ballot_w64(act)) break;
                }
                const f32x2 tk = {tau, tau};
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) {
                    f32x2 p;
                    if constexpr (MODE == SOLVE_NEWTON) {
                        p = pkeep[jp];
                    } else {
                        const f32x2 t = pk_sub_clamp01(XP_GET(jp), tk);
                        if constexpr (MODE == SOLVE_MICHELOT) p = t;
                        else p = t * t;
                    }
                    XP_SET(jp, p);
                }
            }
            {
                const float S = Ssum + (tau - tau);
                float r = __builtin_amdgcn_rcpf(S);
                r = fmaf(fmaf(-S, r, 1.0f), r, r);
                const f32x2 r2 = {r, r};
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) XP_SET(jp, XP_GET(jp) * r2);
            }
            }   // !AFN

            // ---- MFMA #3: dW[f, o] = X[f, :] . ds[o, :] (layout of the gates) -------------------------------
            f32x4 c3[NTILE];
#pragma unroll
            for (int kb = 0; kb < EB; ++kb) {
                f32x4 av[NTILE];
#pragma unroll
                for (int t = 0; t < NTILE; ++t) {
                    av[t] = *reinterpret_cast<const f32x4*>(xt + (16 * t + c) * ES + 16 * kb + 4 * g);
                    if constexpr (GC) {
                        // gc_arm.py:89-92: the interaction runs on y = emb_bn(exp(x)); tile row 16t + c is field
                        // 16t + 4(c & 3) + (c >> 2) (pad rows: scale = shift = 0)
                        const f32x2 es = *reinterpret_cast<const f32x2*>(p_x + 2 * (16 * t + 4 * (c & 3) + (c >> 2)));
#pragma unroll
                        for (int r = 0; r < 4; ++r) av[t][r] = fmaf(__builtin_amdgcn_exp2f(av[t][r] * L2E), es[0], es[1]);
                    }
                    if constexpr (AFN) {
                        // afn.py:63: l = emb_bn(log(x)); pad rows and columns past nemb hold 0 (log2 = -inf): selected away
                        const int fr = 16 * t + 4 * (c & 3) + (c >> 2);
                        const f32x2 es = *reinterpret_cast<const f32x2*>(p_x + 2 * fr);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float lg = fmaf(__builtin_amdgcn_logf(av[t][r]), es[0], es[1]);
                            av[t][r] = (fr < F && 16 * kb + 4 * g + r < Er) ? lg : 0.f;
                        }
                    }
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int t = 0; t < NTILE; ++t) {
                        if (kb == 0 && kk == 0)
                            c3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], ds4[kb][kk], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        else
                            c3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][kk], ds4[kb][kk], c3[t], 0, 0, 0);
                    }
            }
#define DG(j) c3[(j) >> 2][(j) & 3]
            // ds in LDS for the k = neuron contraction of MFMA #5 (rows of padding neurons are zero)
#pragma unroll
            for (int eb = 0; eb < EB; ++eb) *reinterpret_cast<f32x4*>(dsl + c * ES + 16 * eb + 4 * g) = ds4[eb];

            // ---- d_values, dp, entmax / softmax Jacobian-vector product: two elements per instruction -----------------
#define DP_GET(jp) (f32x2{DG(2 * (jp)), DG(2 * (jp) + 1)})
#define DP_SET(jp, v)            \
    do {                         \
        const f32x2 _v = (v);    \
        DG(2 * (jp)) = _v[0];    \
        DG(2 * (jp) + 1) = _v[1]; \
    } while (0)
            auto gppr = [&](f32x2 p2) -> f32x2 {                       // p^(2 - alpha) on the support, 0 off it
                if constexpr (MODE == SOLVE_MICHELOT) {
                    return pk_mul_clamp01(p2, f32x2{0x1p120f, 0x1p120f});
                } else if constexpr (MODE == SOLVE_NEWTON15) {
                    return f32x2{__builtin_sqrtf(p2[0]), __builtin_sqrtf(p2[1])};
                } else if constexpr (MODE == SOLVE_BISECT) {           // any alpha (2 - alpha may be <= 0): select on p > 0
                    return f32x2{p2[0] > 0.f ? __builtin_amdgcn_exp2f(two_m_alpha * __builtin_amdgcn_logf(p2[0])) : 0.f,
                                 p2[1] > 0.f ? __builtin_amdgcn_exp2f(two_m_alpha * __builtin_amdgcn_logf(p2[1])) : 0.f};
                } else {                                               // log2(0) = -inf -> exp2(-inf) = 0 (alpha < 2)
                    return f32x2{__builtin_amdgcn_exp2f(two_m_alpha * __builtin_amdgcn_logf(p2[0])),
                                 __builtin_amdgcn_exp2f(two_m_alpha * __builtin_amdgcn_logf(p2[1]))};
                }
            };
            f32x2 s12 = {0.f, 0.f}, s22 = {0.f, 0.f};
            if constexpr (AFN) {
#pragma unroll
                for (int jp = 0; jp < NP; ++jp) dvacc[nt][jp] += DP_GET(jp);      // d afn.weight[o, f]
            }
#pragma unroll
            for (int jp = 0; !AFN && jp < NP; ++jp) {
                const f32x2 p2 = XP_GET(jp), dW2 = DP_GET(jp), v2 = VV(jp);
                dvacc[nt][jp] = __builtin_elementwise_fma(p2, dW2, dvacc[nt][jp]);
                const f32x2 dp2 = v2 * dW2;
                if constexpr (MODE == SOLVE_SOFTMAX) {
                    DP_SET(jp, dp2);
                    s12 = __builtin_elementwise_fma(p2, dp2, s12);
                } else {
                    const f32x2 gp2 = gppr(p2);
                    const f32x2 dxp2 = dp2 * gp2;
                    DP_SET(jp, dxp2);
                    s12 += dxp2;
                    s22 += gp2;
                }
            }
            wave_lds_fence();
            if constexpr (!AFN) red_write(red, 0, lane, s12[0] + s12[1], s22[0] + s22[1]);
            wave_lds_fence();
            float qv = 0.f;
            if constexpr (!AFN) {
                const Red2 r = red_read(red, 0, c);
                const f32x2 sd = (r.g0 + r.g1) + (r.g2 + r.g3);
                qv = (MODE == SOLVE_SOFTMAX) ? sd[0] : sd[0] / sd[1];
            }
            const f32x2 nq2 = {-qv, -qv};
#pragma unroll
            for (int jp = 0; !AFN && jp < NP; ++jp) {
                const f32x2 p2 = XP_GET(jp);
                if constexpr (MODE == SOLVE_SOFTMAX) DP_SET(jp, p2 * (DP_GET(jp) + nq2));
                else DP_SET(jp, __builtin_elementwise_fma(nq2, gppr(p2), DP_GET(jp)));
                XP_SET(jp, p2 * VV(jp));                                 // w = p * values
            }
#undef DP_GET
#undef DP_SET
            // ---- w transposed ([o][tile row]) for the contraction over neurons (dg follows through the same buffer) ----
#pragma unroll
            for (int t = 0; t < NTILE; ++t) *reinterpret_cast<f32x4*>(tw + c * RS + 16 * t + 4 * g) = c1[t];
            // ---- MFMA #4: dq_fold^T[e, o] += sum_f X[f, e] dg[o, f], accumulated over the wave's samples ------
#pragma unroll
            for (int j = 0; !AFN && j < NQ; ++j)
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) {
                    const int row = 16 * (j >> 2) + 4 * g + (j & 3);
                    const float a2 = xt[row * ES + 16 * eb + c];
                    dqacc[nt][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, DG(j), dqacc[nt][eb], 0, 0, 0);
                }
            // ---- MFMA #5: dx[f, e] += sum_o w[o, f] ds[o, e] + dg[o, f] q_fold[o, e]: two halves through one buffer ------
            wave_lds_fence();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int ol = 4 * kk + g;                               // k index of this lane: neuron within the pass
                float b1[EB];
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) b1[eb] = dsl[ol * ES + 16 * eb + c];
#pragma unroll
                for (int t = 0; t < NTILE; ++t) {
                    const float a1 = tw[ol * RS + 16 * t + c];
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb) {
                        if constexpr (DY) cdy[t][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[eb], cdy[t][eb], 0, 0, 0);
                        else cdx[t][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1[eb], cdx[t][eb], 0, 0, 0);
                    }
                }
            }
            wave_lds_fence();
#pragma unroll
            for (int t = 0; !AFN && t < NTILE; ++t) *reinterpret_cast<f32x4*>(tw + c * RS + 16 * t + 4 * g) = c3[t];
            wave_lds_fence();
#pragma unroll
            for (int kk = 0; !AFN && kk < 4; ++kk) {
                const int ol = 4 * kk + g;
                float b2[EB];
#pragma unroll
                for (int eb = 0; eb < EB; ++eb) b2[eb] = qfl[(16 * nt + ol) * ES + 16 * eb + c];
#pragma unroll
                for (int t = 0; t < NTILE; ++t) {
                    const float a2 = tw[ol * RS + 16 * t + c];
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb)
                        cdx[t][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2[eb], cdx[t][eb], 0, 0, 0);
                }
            }
            wave_lds_fence();
            // keep the unrolled passes apart: without this the scheduler hoists the next passes' operand loads over
            // the current one and the four passes' live ranges no longer fit the register file
            __builtin_amdgcn_sched_barrier(0);
#undef XG
#undef XP_GET
#undef XP_SET
#undef VV
#undef DG
        }
        if constexpr (LATE) {
            // next sample's rows (its ids arrived during this sample), ids / values of the one after
#pragma unroll
            for (int n = 0; n < NI; ++n) {
                idC[n] = idN[n];
                vC[n] = vN[n];
            }
            fetch_rows();
            fetch_ids(b + 2 * nwaves, idN, vN);
        }
        // ---- scatter: d_table[id[f], e] += dx[f, e] * val[f]   (x = table[id] * val, layers.py:20-21) ----------
#pragma unroll
        for (int t = 0; t < NTILE; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * g + r;
                const uint32_t id = idl[row];
                const float v = vll[row];
                if (!AFN && id != 0xffffffffu && !dbg_no_scatter) {
                    float* dst = a.d_table + (size_t)id * Er + c;
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb)
                        if (16 * eb + c < Er) unsafeAtomicAdd(dst + 16 * eb, cdx[t][eb][r] * v);
                }
                if constexpr (DY) {
                    // the gradient of y: every (sample, field, e) belongs to one lane of one wave — plain stores
                    if (id != 0xffffffffu) {
                        float* dy = gx.d_y + ((size_t)b * F + (16 * t + 4 * r + g)) * Er + c;
#pragma unroll
                        for (int eb = 0; eb < EB; ++eb)
                            if (16 * eb + c < Er) {
                                float v2 = cdy[t][eb][r];
                                if (gx.accumulate) v2 += dy[16 * eb];
                                dy[16 * eb] = v2;
                            }
                    }
                }
            }
    }
    // ---- wave accumulators -> block accumulators (LDS atomics, once per wave) -> global ------------------------
    __syncthreads();                                                             // every wave is done with its region
    for (int i = threadIdx.x; i < OP * (FP + E) + (AFN ? OP : 0); i += 256) acc_dv[i] = 0.f;     // acc_dv, acc_dq (, acc_db) are contiguous
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTS; ++nt) {
        if (nt >= NT) break;
        float* dv_row = acc_dv + (16 * nt + c) * FP + g;
#pragma unroll
        for (int j = 0; j < NQ; ++j) atomicAdd(dv_row + 4 * j, dvacc[nt][j >> 1][j & 1]);   // pad slots receive zeros
        if constexpr (AFN) {
            atomicAdd(acc_db + 16 * nt + c, dbacc[nt]);
            continue;
        }
        float* dq_row = acc_dq + (16 * nt + c) * E + 4 * g;
#pragma unroll
        for (int eb = 0; eb < EB; ++eb)
#pragma unroll
            for (int r = 0; r < 4; ++r) atomicAdd(dq_row + 16 * eb + r, dqacc[nt][eb][r]);
    }
    __syncthreads();
    if (dbg_no_flush) return;
    for (int i = threadIdx.x; i < O * F; i += 256) {
        const int o = i / F, f = i - o * F;
        unsafeAtomicAdd(a.d_values + i, acc_dv[o * FP + f]);
    }
    if constexpr (AFN) {
        for (int i = threadIdx.x; i < O; i += 256) unsafeAtomicAdd(gx.d_bias + i, acc_db[i]);
        return;
    }
    for (int i = threadIdx.x; i < O * Er; i += 256) {
        const int o = i / Er, e = i - o * Er;
        unsafeAtomicAdd(a.d_qfold + i, acc_dq[o * E + e]);
    }
}

template <int E, int NQ, int MODE, int SRC, int MODEL = MODEL_ARM>
static int launch_bwd_one(const BwdArgs& a, const BwdExtra& gx, hipStream_t st) {
    constexpr int NTILE = (NQ + 3) / 4, ROWS = NTILE * 16;
    constexpr int WAVE_FLOATS = ROWS * (E + 4) + 128 + 16 * (ROWS + 4) + 16 * (E + 4) + 2 * ROWS;
    const int NT = (a.O + 15) / 16, OP = NT * 16;
    size_t lds = ((size_t)4 * WAVE_FLOATS + (size_t)NT * (E / 16) * 256 + (size_t)NT * (NQ / 2) * 128 +
                  (size_t)OP * (E + 4) + (size_t)OP * 4 + (MODEL != MODEL_ARM ? (size_t)2 * ROWS : 0)) * sizeof(float);
    if ((size_t)OP * (4 * NQ + E + 1) > (size_t)4 * WAVE_FLOATS) return ARMNET_ERR_UNSUPPORTED;    // the aliased accumulators
    if (lds > 160 * 1024) return ARMNET_ERR_UNSUPPORTED;
#ifdef ARMNET_DEV_FLAGS
    if (const char* pad = getenv("ARMNET_BWD_LDS_PAD")) lds += (size_t)atoi(pad);     // developer knob: lower the occupancy
#endif
    int per_cu = (int)(160 * 1024 / lds);
    if (per_cu > bwd_blocks_per_cu(E)) per_cu = bwd_blocks_per_cu(E);
    const int64_t blocks = (a.B + 3) / 4;
    const int64_t resident = (int64_t)device_cu_count() * per_cu;
    const int64_t want = blocks < resident ? blocks : resident;
    auto kern = fused_bwd_mfma_kernel<E, NQ, MODE, SRC, MODEL>;
    ARMNET_ALLOW_BIG_LDS(kern, lds);
    kern<<<(int)want, 256, lds, st>>>(a, gx);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

template <int E, int NQ, int MODEL = MODEL_ARM>
static int launch_bwd_src(const BwdArgs& a, hipStream_t st, const BwdExtra& gx = BwdExtra{}) {
#define ARMNET_BWD_MODE(SRC)                                                                         \
    switch (a.cfg.mode) {                                                                            \
        case SOLVE_SOFTMAX: return launch_bwd_one<E, NQ, SOLVE_SOFTMAX, SRC, MODEL>(a, gx, st);             \
        case SOLVE_MICHELOT: return launch_bwd_one<E, NQ, SOLVE_MICHELOT, SRC, MODEL>(a, gx, st);           \
        case SOLVE_NEWTON15: return launch_bwd_one<E, NQ, SOLVE_NEWTON15, SRC, MODEL>(a, gx, st);           \
        case SOLVE_NEWTON: return launch_bwd_one<E, NQ, SOLVE_NEWTON, SRC, MODEL>(a, gx, st);               \
        case SOLVE_BISECT: return launch_bwd_one<E, NQ, SOLVE_BISECT, SRC, MODEL>(a, gx, st);               \
        default: return ARMNET_ERR_UNSUPPORTED;                                                      \
    }
    if (a.id_type == ARMNET_ID_I64) { ARMNET_BWD_MODE(0) }
    ARMNET_BWD_MODE(1)
#undef ARMNET_BWD_MODE
}

int launch_bwd_mfma_e16(const BwdArgs& a, int nq, hipStream_t st);
int launch_bwd_mfma_e32(const BwdArgs& a, int nq, hipStream_t st);
int launch_bwd_mfma_e64(const BwdArgs& a, int nq, hipStream_t st);
int launch_bwd_mfma_e128(const BwdArgs& a, int nq, hipStream_t st);    // nq 2..8 (wider samples do not fit the LDS)
// GC-ARM (fused_bwd_gc_*.hip)
int launch_bwd_gc_e16(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st);
int launch_bwd_gc_e32(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st);
int launch_bwd_gc_e64(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st);
int launch_bwd_gc_e128(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st);   // nq 2..8, like launch_bwd_mfma_e128

// AFN (fused_bwd_afn.hip): no sparse map, so one solver-mode instantiation per shape
template <int E, int NQ>
static int launch_bwd_afn_t(const BwdArgs& a, const BwdExtra& gx, hipStream_t st) {
    if (a.id_type == ARMNET_ID_I64) return launch_bwd_one<E, NQ, SOLVE_SOFTMAX, 0, MODEL_AFN>(a, gx, st);
    return launch_bwd_one<E, NQ, SOLVE_SOFTMAX, 1, MODEL_AFN>(a, gx, st);
}

}  // namespace armnet
