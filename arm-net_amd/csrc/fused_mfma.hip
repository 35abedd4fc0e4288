// fused_mfma.hip — shape dispatch of the fused MFMA kernel (fused_mfma_kernel.h).
#include "fused_mfma_kernel.h"

namespace armnet {

// nemb 4..128 (any, odd too: 16-byte staging chunks at the rows' natural 4-byte alignment; 65..128 since round 4 — the
// reference's own best-AUC command is --nemb 100, README.md:32-42); nfield <= 48; neurons <= 1024 (slices of <= 256 per launch)
static int padded_nemb(int E) { return E <= 16 ? 16 : E <= 32 ? 32 : E <= 64 ? 64 : 128; }

// Waves a CU holds for a slice of `o_slice` neurons (same LDS formula and block-size choice as launch_one; the sibling
// models add a table of < 1 KiB: a shape at the very edge is refused by launch_one and runs on the generic kernel).
// The high-occupancy table is the ARM block's only (launch_one: MODEL_ARM): GC-ARM / AFN slices are sized from the
// configuration they are launched with (round-3 advisor finding).
static int mfma_cu_waves(int F, int E, int o_slice, int model) {
    const int nq = (((F + 3) / 4) + 1) & ~1;
    const int ep = padded_nemb(E);
    int spw = (ep >= 64 || nq % 4 == 0) ? 1 : 2;
    int wps = ep >= 128 ? (nq >= 6 ? 2 : 3) : ep >= 64 ? (nq >= 10 ? 2 : 3) : (ep >= 32 || spw * nq >= 16) ? 3 : 4;   // launch_one's register budget (alpha = 2)
    const OccCfg oc = occ_config(ep, nq, SOLVE_MICHELOT);                              // ... or its high-occupancy table
    if (model == MODEL_ARM && oc.spw != 0 && (nq != 10 || o_slice <= 32)) { spw = oc.spw; wps = oc.wps; }
    const int ntile = (spw * nq + 3) / 4, nt = (o_slice + 15) / 16;
    const size_t wave_bytes = (size_t)(ntile * 16 * (ep + 4) + 256) * sizeof(float);
    const size_t param_bytes = ((size_t)nt * (ep / 16) * 256 + (size_t)nt * (nq / 2) * 128 + (size_t)nt * 32) * sizeof(float);
    int b = 0;
    const int w = mfma_pick_wpb(wave_bytes, param_bytes, wps, &b);
    return w * b;
}

// neurons per launch: as many as possible (each slice re-gathers the rows) while the CU still holds 3 waves per SIMD —
// or, where no slice reaches that (nemb >= 64), the slice that holds the most
static int mfma_slice(int F, int E, int O, int model) {
    int best = 64, best_waves = -1;
    for (int slice = 256; slice >= 64; slice /= 2) {
        if (slice / 2 >= O && slice > 64) continue;                      // a smaller slice already covers O
        int w = mfma_cu_waves(F, E, slice < O ? slice : O, model);
        if (w > 12) w = 12;
        if (w > best_waves) { best = slice; best_waves = w; }
    }
    // nemb > 64: the tile of a 40-row sample is 25 KiB, so only short slices leave room for a block at all
    for (int slice = 32; slice >= 16 && best_waves <= 0; slice /= 2) {
        const int w = mfma_cu_waves(F, E, slice < O ? slice : O, model);
        if (w > best_waves) { best = slice; best_waves = w; }
    }
    return best;
}

static bool mfma_supports_model(int F, int E, int O, int model) {
    if (E < 4 || E > 128 || O < 1 || O > 1024 || F < 1 || F > 48) return false;
    const int slice = mfma_slice(F, E, O, model);
    return mfma_cu_waves(F, E, O < slice ? O : slice, model) > 0;
}

bool fused_mfma_supports(int F, int E, int O) { return mfma_supports_model(F, E, O, MODEL_ARM); }
bool fused_mfma_supports_model(int F, int E, int O, int model) { return mfma_supports_model(F, E, O, model); }

int launch_fused_mfma(const FusedArgs& a, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (a.B * a.F >= ((int64_t)1 << 29)) return ARMNET_ERR_UNSUPPORTED;   // 32-bit byte offsets into ids/vals
    if (!mfma_supports_model(a.F, a.E, a.O, a.model)) return ARMNET_ERR_UNSUPPORTED;
    const int nq = (((a.F + 3) / 4) + 1) & ~1;            // quarter-steps per sample, rounded up to even
    // many neurons: the lane-ready parameter copies crowd the tiles out of LDS -> slices (each re-gathers the rows;
    // the in-place clamp is idempotent)
    const int slice = mfma_slice(a.F, a.E, a.O, a.model);
    for (int o0 = 0; o0 < a.O; o0 += slice) {
        FusedArgs s = a;
        s.O = a.O - o0 < slice ? a.O - o0 : slice;
        s.O_out = a.O;
        s.q_fold = a.q_fold + (size_t)o0 * a.E;
        s.values = a.values + (size_t)o0 * a.F;
        s.bn_scale = a.bn_scale + o0;
        s.bn_shift = a.bn_shift + o0;
        s.out = a.out + (size_t)o0 * a.E;
        if (a.model == MODEL_AFN) s.lin_bias = a.lin_bias + o0;
        int rc;
        if (a.model == MODEL_AFN) rc = launch_afn(s, padded_nemb(a.E), nq, st);
        else if (a.model == MODEL_GC_ARM)
            rc = a.E <= 16 ? launch_gc_e16(s, nq, st) : a.E <= 32 ? launch_gc_e32(s, nq, st)
               : a.E <= 64 ? launch_gc_e64(s, nq, st) : launch_gc_e128(s, nq, st);
        else if (a.E <= 16) {
            // wide blocks: both contractions as fp16 x 2 splits on the 16-bit matrix pipe (F16, fused_mfma_kernel.h)
            rc = ARMNET_ERR_UNSUPPORTED;
            // (measured, kbench, B = 65 536: 33+ fields 0..-5 % at 40-48 neurons, -5..-8 % at 56-64, -11..-20 % from 128 up; 29-32 fields
            // +4 % at 48, -7 % at 128; 17-28 fields +-2 % at 128, -13 % at 512; 16 fields or fewer lose 0..24 % — their fp32 form runs
            // at 6-8 waves per SIMD and has little matrix work; 32 neurons or fewer: -8 %, five waves per SIMD beat four)
            const int f16_min_o = nq >= 10 ? 33 : nq >= 8 ? ARMNET_F16_MIN_O : 4 * ARMNET_F16_MIN_O;
            if (nq >= 6 && s.O >= f16_min_o && s.cfg.mode != SOLVE_BISECT && !(s.flags & ARMNET_F_FP32_CONTRACTIONS))
                rc = launch_mfma_e16_f16(s, nq, st);
            if (rc == ARMNET_ERR_UNSUPPORTED) rc = launch_mfma_e16(s, nq, st);
        }
        else if (a.E <= 32) rc = launch_mfma_e32(s, nq, st);
        else if (a.E <= 64) rc = launch_mfma_e64(s, nq, st);
        else rc = nq <= 6 ? launch_mfma_e128a(s, nq, st) : launch_mfma_e128b(s, nq, st);
        if (rc != ARMNET_OK) return rc;      // a refusal can only happen on the first slice (same shape after it)
    }
    return ARMNET_OK;
}

}  // namespace armnet
