// fused_mfma.hip — the fused ARM block on the CDNA4 matrix cores (gfx950), fp32 end to end.
//
// One WAVE owns a group of SPW samples at a time and never talks to another wave (wave-private
// LDS tile, no block barrier).  Per group:
//
//   stage     coalesced 16-B chunk loads of the F embedding rows of each sample (adjacent lanes
//             share a row), scaled by clamp(value), written to the wave's LDS tile.  Rows for the
//             NEXT group are already in flight (registers) and ids/values for the group after that
//             are being fetched, so the id -> row dependent latency chain is off the critical path.
//   MFMA #1   gates  G[(s,f), o] = X[(s,f), :] . q_fold[o, :]   v_mfma_f32_16x16x4_f32, exact fp32.
//             The tile rows are ordered so that accumulator register r of tile t is "quarter-step"
//             q = 4t + r = s*NQ + j of ONE sample: lane (c = l&15, g = l>>4) then holds, for neuron
//             o = 16*nt + c, the gates of fields f = 4j + g, j = 0..NQ-1  -> NQ values per row.
//   sparse    entmax / softmax over the F fields of every (sample, neuron) row, in registers.  A row
//   map       is spread over the 4 lane groups g: reductions are 2 v_permlane{32,16}_swap + 2 adds.
//             alpha = 2: Michelot (= Newton from the left, finite), alpha = 1.5 / generic: Newton.
//   MFMA #2   Z^T[e, o] = sum_f X[f, e] * W[o, f]:  the C layout of MFMA #1 IS the B-operand layout
//             of MFMA #2 (k = lane group g <-> field 4j+g), so the weights never move; the
//             contraction index is just visited in the permuted order both operands agree on.
//   epilogue  exp, eval-BatchNorm affine, one 16-byte store per lane (lane holds 4 consecutive e).
//
// LDS tile: NTILE*16 rows x (E+4) floats (row stride padded by one 16-B slot: conflict-free
// ds_read_b32 column reads for MFMA #2, <= 2-way on the ds_read_b128 row reads of MFMA #1).
#include "armnet_common.h"

namespace armnet {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int cgcd(int a, int b) { return b == 0 ? a : cgcd(b, a % b); }

__device__ __forceinline__ float sum_over_groups(float v) {
    // all-reduce over the 4 lane groups {l, l^16, l^32, l^48}: two swaps + two adds
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__device__ __forceinline__ float max_over_groups(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wave is serviced in issue order; this only stops the COMPILER from moving
    // a lane's reads across other lanes' writes.
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <int E, int NQ, int NTC, int MODE, int SRC>   // SRC: 0 = int64 ids, 1 = int32 ids, 2 = pre-gathered rows
__global__ void __launch_bounds__(256, (E >= 64 ? 1 : 2)) fused_mfma_kernel(FusedArgs a) {
    constexpr int SPW = 4 / cgcd(NQ, 4);      // samples per wave-group
    constexpr int NTILE = SPW * NQ / 4;       // 16-row MFMA tiles per group
    constexpr int ES = E + 4;                 // LDS row stride (floats)
    constexpr int CH = E / 4;                 // 16-byte chunks per row
    constexpr int RPI = 64 / CH;              // rows per staging instruction
    constexpr int NI = NTILE * 16 / RPI;      // staging instructions per group
    constexpr int EB = E / 16;                // 16-wide blocks of the embedding dim
    constexpr int NR = SPW * NTC;             // (sample, neuron) rows per lane
    constexpr bool FROM_ROWS = (SRC == 2);
    static_assert(E % 16 == 0 && (NTILE * 16) % RPI == 0, "shape");

    extern __shared__ __attribute__((aligned(16))) float lds_all[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float* xt = lds_all + wave * (NTILE * 16 * ES);
    const int c = lane & 15, g = lane >> 4;
    const int F = a.F, O = a.O;
    const int64_t ngroups = (a.B + SPW - 1) / SPW;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    int64_t grp = (int64_t)blockIdx.x * 4 + wave;
    if (grp >= ngroups) return;

    // ---- group-invariant staging geometry: which (sample, field) each staging lane fetches -------
    const int chunk = lane % CH;
    int sf_off[NI];      // s*F + f of the row this lane stages in instruction n, or -1 for a pad row
    int s_of[NI];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
        const int row = n * RPI + lane / CH;
        const int t = row >> 4, i = row & 15;
        const int q = 4 * t + (i & 3);
        const int s = q / NQ, j = q - s * NQ;
        const int f = 4 * j + (i >> 2);
        s_of[n] = s;
        sf_off[n] = (f < F) ? s * F + f : -1;
    }
    const bool pad_last = (4 * (NQ - 1) + g) >= F;   // this lane's last quarter-step is a pad field
    const bool write_vals = (a.flags & ARMNET_F_WRITE_CLAMPED_VALS) != 0;
    // ablation switches for profiling (tools/kbench.py); never set by the product path
    const bool dbg_no_solve = (a.flags & 0x100u) != 0;   // skip the Newton iterations
    const bool dbg_hot_rows = (a.flags & 0x200u) != 0;   // fold ids into 1024 rows (cache-resident gather)
    const bool dbg_no_store = (a.flags & 0x400u) != 0;   // skip the output stores

    // ---- per-lane parameters of the neuron chunk (hoisted when O == 16*NTC) ---------------------
    const int n_chunks = O / (16 * NTC);
    f32x4 bq[NTC][EB];     // q_fold[o][16kb + 4g .. +3]          (B operand of MFMA #1)
    float vv[NTC][NQ];     // values[o][4j + g]
    float sc[NTC], sh[NTC];
    auto load_chunk_params = [&](int o0) {
#pragma unroll
        for (int nt = 0; nt < NTC; ++nt) {
            const int o = o0 + 16 * nt + c;
#pragma unroll
            for (int kb = 0; kb < EB; ++kb)
                bq[nt][kb] = *reinterpret_cast<const f32x4*>(a.q_fold + (size_t)o * E + 16 * kb + 4 * g);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int f = 4 * j + g;
                vv[nt][j] = (f < F) ? a.values[(size_t)o * F + f] : 0.f;
            }
            sc[nt] = a.bn_scale[o];
            sh[nt] = a.bn_shift[o];
        }
    };
    if (n_chunks == 1) load_chunk_params(0);

    // ---- software pipeline -------------------------------------------------------------------------
    // iteration k:  stage rows(k) -> LDS | finish ids/vals(k+1) (clamp, range check) | issue row loads(k+1)
    //               | issue RAW id/val loads(k+2) | compute(k).  Nothing loaded in an iteration is looked
    //               at before the next one, so both legs of the id -> row latency chain overlap compute.
    f32x4 rows_cur[NI];            // raw rows of the CURRENT group (loads issued one iteration ago)
    float val_cur[NI];             // clamped values of the current group
    uint32_t raw_lo[NI], raw_hi[NI];   // untouched id words of the NEXT group (hi only for int64 ids)
    float raw_val[NI];                 // untouched values of the next group
    const int Bi = (int)a.B;       // launcher guarantees B < 2^31

    auto lane_valid = [&](int n, int64_t gidx) -> bool {
        const int64_t b0 = gidx * SPW;
        return sf_off[n] >= 0 && gidx < ngroups && (int)b0 + s_of[n] < Bi;
    };
    auto fetch_raw = [&](int64_t gidx) {
        const int64_t b0 = gidx * SPW;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int64_t gi = lane_valid(n, gidx) ? b0 * F + sf_off[n] : 0;   // element 0 is always readable
            raw_val[n] = a.vals[gi];
            if constexpr (SRC == 0) {
                const uint2 w = reinterpret_cast<const uint2*>(a.ids)[gi];
                raw_lo[n] = w.x;
                raw_hi[n] = w.y;
            } else if constexpr (SRC == 1) {
                raw_lo[n] = reinterpret_cast<const uint32_t*>(a.ids)[gi];
                raw_hi[n] = 0u;
            }
        }
    };
    // clamp (armnet_1h.py:81), optional write-back of the clamp, id range check; then the row loads
    auto finish_and_issue = [&](int64_t gidx, float* vals_out) {
        const int64_t b0 = gidx * SPW;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const bool valid = lane_valid(n, gidx);
            const float vraw = raw_val[n];
            const float v = clamp_val(vraw);
            if (write_vals && valid && chunk == 0 && v != vraw) a.vals[b0 * F + sf_off[n]] = v;
            vals_out[n] = valid ? v : 0.f;       // pad rows / tail samples stage zeros
            const float* src;
            if constexpr (FROM_ROWS) {
                src = a.rows + (valid ? (b0 * F + sf_off[n]) * (int64_t)E : 0) + chunk * 4;
            } else {
                uint32_t id = raw_lo[n];
                const bool bad = valid && (raw_hi[n] != 0u || id >= (uint32_t)a.nfeat);
                if (bad && a.id_status && chunk == 0) atomicOr(a.id_status, 1);
                if (bad || !valid) id = 0u;
                if (dbg_hot_rows) id &= 1023u;
                src = a.table + (size_t)id * E + chunk * 4;
            }
            rows_cur[n] = *reinterpret_cast<const f32x4*>(src);
        }
    };

    fetch_raw(grp);
    finish_and_issue(grp, val_cur);
    fetch_raw(grp + nwaves);

    for (; grp < ngroups; grp += nwaves) {
        const int64_t b0 = grp * SPW;
        // ---- stage the current group's rows (scaled) into the wave's LDS tile -----------------------
        wave_lds_fence();
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            f32x4 r = rows_cur[n] * val_cur[n];
            const int row = n * RPI + lane / CH;
            *reinterpret_cast<f32x4*>(xt + row * ES + chunk * 4) = r;
        }
        // ---- keep the memory pipeline full: rows of the next group, raw ids of the one after -----
        finish_and_issue(grp + nwaves, val_cur);
        fetch_raw(grp + 2 * nwaves);
        wave_lds_fence();

        for (int ch = 0; ch < n_chunks; ++ch) {
            const int o0 = ch * 16 * NTC;
            if (n_chunks > 1) load_chunk_params(o0);

            // ---- MFMA #1: gates ---------------------------------------------------------------------
            f32x4 c1[NTILE][NTC];
#pragma unroll
            for (int t = 0; t < NTILE; ++t) {
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt) c1[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < EB; ++kb) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(xt + (16 * t + c) * ES + 16 * kb + 4 * g);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int nt = 0; nt < NTC; ++nt)
                            c1[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kk], bq[nt][kb][kk], c1[t][nt], 0, 0, 0);
                }
            }
#define XG(s, nt, j) c1[((s) * NQ + (j)) >> 2][nt][((s) * NQ + (j)) & 3]

            // ---- sparse map over the fields, rows spread over the 4 lane groups ---------------------
            float tau[NR], inv[NR];
            if constexpr (MODE == SOLVE_SOFTMAX) {
#pragma unroll
                for (int s = 0; s < SPW; ++s)
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt) {
                        float mx = -INFINITY, sm = 0.f;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            float x = XG(s, nt, j);
                            sm += x;
                            if (j == NQ - 1 && pad_last) x = -INFINITY;
                            XG(s, nt, j) = x;
                            mx = fmaxf(mx, x);
                        }
                        mx = max_over_groups(mx);
                        sm = sum_over_groups(sm);
                        float S = 0.f;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            const float p = expf(XG(s, nt, j) - mx);
                            XG(s, nt, j) = p;
                            S += p;
                        }
                        S = sum_over_groups(S);
                        if (sm != sm) S = NAN;
                        inv[s * NTC + nt] = 1.0f / S;
                    }
            } else {
                const float am1 = a.cfg.am1;
                const float rr = a.cfg.r, rm1 = a.cfg.r - 1.0f;
                const float invF = 1.0f / (float)F;
                // scale (entmax.py:42), starting threshold
#pragma unroll
                for (int s = 0; s < SPW; ++s)
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt) {
                        float mx = -INFINITY, sm = 0.f;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            float x = XG(s, nt, j);
                            if constexpr (MODE != SOLVE_MICHELOT) x *= am1;
                            sm += x;                                    // a pad field's gate is exactly 0
                            if (j == NQ - 1 && pad_last) x = -INFINITY;
                            XG(s, nt, j) = x;
                            mx = fmaxf(mx, x);
                        }
                        mx = max_over_groups(mx);
                        sm = sum_over_groups(sm);
                        float t0 = fmaxf(mx - 1.0f, sm * invF - a.cfg.tau_hi_off);
                        if (!(mx < INFINITY) || sm != sm) t0 = NAN;     // +inf / NaN gate -> NaN row
                        tau[s * NTC + nt] = t0;
                    }
                // Newton from the left; wave-uniform loop, rows drop out as they converge
                for (int it = 0; it < (dbg_no_solve ? 0 : kNewtonMaxIter); ++it) {
                    bool any_active = false;
#pragma unroll
                    for (int s = 0; s < SPW; ++s)
#pragma unroll
                        for (int nt = 0; nt < NTC; ++nt) {
                            const float tk = tau[s * NTC + nt];
                            float S = 0.f, Dv = 0.f;
#pragma unroll
                            for (int j = 0; j < NQ; ++j) {
                                const float t = fmaxf(XG(s, nt, j) - tk, 0.f);
                                if constexpr (MODE == SOLVE_MICHELOT) {
                                    S += t;
                                    Dv += (t > 0.f) ? 1.f : 0.f;
                                } else if constexpr (MODE == SOLVE_NEWTON15) {
                                    S = fmaf(t, t, S);
                                    Dv += t;
                                } else {
                                    const float u = t > 0.f ? pow_pos(t, rm1) : 0.f;
                                    S = fmaf(u, t, S);
                                    Dv += u;
                                }
                            }
                            S = sum_over_groups(S);
                            Dv = sum_over_groups(Dv);
                            if constexpr (MODE == SOLVE_NEWTON15) Dv *= 2.0f;
                            if constexpr (MODE == SOLVE_NEWTON) Dv *= rr;
                            const float f = S - 1.0f;
                            const float tn = fmaf(f, __builtin_amdgcn_rcpf(Dv), tk);   // Newton self-corrects: 1-ulp rcp is enough
                            const bool act = (f > kNewtonTol) && (tn > tk);
                            tau[s * NTC + nt] = act ? tn : tk;
                            any_active |= act;
                        }
                    if (!__builtin_amdgcn_ballot_w64(any_active)) break;
                }
                // p at the converged threshold, row sum for the normalisation (entmax.py:63-64)
#pragma unroll
                for (int s = 0; s < SPW; ++s)
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt) {
                        const float tk = tau[s * NTC + nt];
                        float S = 0.f;
#pragma unroll
                        for (int j = 0; j < NQ; ++j) {
                            const float t = fmaxf(XG(s, nt, j) - tk, 0.f);
                            float p;
                            if constexpr (MODE == SOLVE_MICHELOT) p = t;
                            else if constexpr (MODE == SOLVE_NEWTON15) p = t * t;
                            else p = t > 0.f ? pow_pos(t, rr) : 0.f;
                            XG(s, nt, j) = p;
                            S += p;
                        }
                        S = sum_over_groups(S);
                        if (tk != tk) S = NAN;
                        inv[s * NTC + nt] = 1.0f / S;
                    }
            }
            // ---- value weighting (armnet_1h.py:34): W = p/sum * values, in place -------------------
#pragma unroll
            for (int s = 0; s < SPW; ++s)
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                    for (int j = 0; j < NQ; ++j)
                        XG(s, nt, j) = (XG(s, nt, j) * inv[s * NTC + nt]) * vv[nt][j];

            // ---- MFMA #2: Z^T[e, o] = sum_f X[f, e] * W[o, f];  epilogue; store ---------------------
#pragma unroll
            for (int s = 0; s < SPW; ++s) {
                f32x4 c2[NTC][EB];
#pragma unroll
                for (int nt = 0; nt < NTC; ++nt)
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb) c2[nt][eb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    const int q = s * NQ + j;
                    const int row = 16 * (q >> 2) + 4 * g + (q & 3);
#pragma unroll
                    for (int eb = 0; eb < EB; ++eb) {
                        const float a2 = xt[row * ES + 16 * eb + c];
#pragma unroll
                        for (int nt = 0; nt < NTC; ++nt)
                            c2[nt][eb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, XG(s, nt, j), c2[nt][eb], 0, 0, 0);
                    }
                }
                if (b0 + s < a.B && !dbg_no_store) {
#pragma unroll
                    for (int nt = 0; nt < NTC; ++nt) {
                        float* dst = a.out + ((b0 + s) * O + o0 + 16 * nt + c) * (int64_t)E + 4 * g;
#pragma unroll
                        for (int eb = 0; eb < EB; ++eb) {
                            f32x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaf(exp_accurate(c2[nt][eb][r]), sc[nt], sh[nt]);
                            *reinterpret_cast<f32x4*>(dst + 16 * eb) = v;
                        }
                    }
                }
            }
#undef XG
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
struct MfmaShape { int E, NQ; };
static constexpr MfmaShape kShapes[] = {{16, 10}, {64, 10}, {32, 6}};

bool fused_mfma_supports(int F, int E, int O) {
    if (O % 32 != 0) return false;
    const int nq = (F + 3) / 4;
    for (const auto& s : kShapes)
        if (s.E == E && s.NQ == nq) return true;
    return false;
}

template <int E, int NQ, int MODE, int SRC>
static int launch_one(const FusedArgs& a, hipStream_t st) {
    constexpr int NTC = 2;
    constexpr int SPW = 4 / cgcd(NQ, 4);
    constexpr int NTILE = SPW * NQ / 4;
    constexpr size_t lds = (size_t)4 * NTILE * 16 * (E + 4) * sizeof(float);
    const int64_t ngroups = (a.B + SPW - 1) / SPW;
    int64_t blocks = (ngroups + 3) / 4;
    const int64_t resident = 256 * (lds > 40 * 1024 ? 1 : 2);    // blocks the chip holds at once
    // persistent grid-stride waves: the software pipeline's prologue is paid once per wave
    const int64_t want = blocks < resident ? blocks : resident;
    auto kern = fused_mfma_kernel<E, NQ, NTC, MODE, SRC>;
    if (lds > 64 * 1024)
        ARMNET_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    kern<<<(int)want, 256, lds, st>>>(a);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

template <int E, int NQ, int SRC>
static int launch_mode(const FusedArgs& a, hipStream_t st) {
    switch (a.cfg.mode) {
        case SOLVE_SOFTMAX: return launch_one<E, NQ, SOLVE_SOFTMAX, SRC>(a, st);
        case SOLVE_MICHELOT: return launch_one<E, NQ, SOLVE_MICHELOT, SRC>(a, st);
        case SOLVE_NEWTON15: return launch_one<E, NQ, SOLVE_NEWTON15, SRC>(a, st);
        case SOLVE_NEWTON: return launch_one<E, NQ, SOLVE_NEWTON, SRC>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

int launch_fused_mfma(const FusedArgs& a, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (a.B >= ((int64_t)1 << 31) / a.F) return ARMNET_ERR_UNSUPPORTED;   // 32-bit sample*field indices
    if (!fused_mfma_supports(a.F, a.E, a.O)) return ARMNET_ERR_UNSUPPORTED;
    if (((uintptr_t)a.out | (uintptr_t)a.q_fold | (uintptr_t)(a.rows ? a.rows : a.table)) % 16) return ARMNET_ERR_UNSUPPORTED;
    const int nq = (a.F + 3) / 4;
    const int src = a.rows != nullptr ? 2 : (a.id_type == ARMNET_ID_I64 ? 0 : 1);
#define DISPATCH(E_, NQ_)                                                                        \
    if (a.E == E_ && nq == NQ_)                                                                  \
        return src == 2 ? launch_mode<E_, NQ_, 2>(a, st)                                         \
                        : (src == 0 ? launch_mode<E_, NQ_, 0>(a, st) : launch_mode<E_, NQ_, 1>(a, st));
    DISPATCH(16, 10)
    DISPATCH(64, 10)
    DISPATCH(32, 6)
#undef DISPATCH
    return ARMNET_ERR_UNSUPPORTED;
}

}  // namespace armnet
