// fused_mfma.hip — shape dispatch of the fused MFMA kernel (fused_mfma_kernel.h).
#include "fused_mfma_kernel.h"

namespace armnet {

// nemb 4..64 (any, odd too: 16-byte staging chunks at the rows' natural 4-byte alignment); nfield <= 48; neurons <= 1024 (slices of <= 256 per launch)
// LDS of one block for a slice of `o_slice` neurons (same formula as launch_one): 4 wave tiles + the lane-ready
// parameters of the slice (the sibling models add a table of < 1 KiB: a shape at the very edge is refused by
// launch_one and runs on the generic kernel)
static size_t mfma_lds_bytes(int F, int E, int o_slice) {
    const int nq = (((F + 3) / 4) + 1) & ~1;
    const int ep = E <= 16 ? 16 : E <= 32 ? 32 : 64;
    const int spw = (ep >= 64 || nq % 4 == 0) ? 1 : 2;
    const int ntile = (spw * nq + 3) / 4, nt = (o_slice + 15) / 16;
    return ((size_t)4 * (ntile * 16 * (ep + 4) + 256) + (size_t)nt * (ep / 16) * 256 + (size_t)nt * (nq / 2) * 128 +
            (size_t)nt * 32) * sizeof(float);
}

// neurons per launch: as many as possible (each slice re-gathers the rows) while two blocks still fit a CU's LDS —
// at 256 neurons the parameter copies of a 39-field block leave room for one block (1 wave/SIMD): measured 853 us
// against 2 x 373 us for two launches of 128
static int mfma_slice(int F, int E, int O) {
    int slice = 256;
    while (slice > 64 && slice / 2 >= 16 && (O > slice / 2) && mfma_lds_bytes(F, E, slice < O ? slice : O) > 80 * 1024)
        slice /= 2;
    return slice;
}

bool fused_mfma_supports(int F, int E, int O) {
    if (E < 4 || E > 64 || O < 1 || O > 1024 || F < 1 || F > 48) return false;
    const int slice = mfma_slice(F, E, O);
    return mfma_lds_bytes(F, E, O < slice ? O : slice) <= 160 * 1024;
}

int launch_fused_mfma(const FusedArgs& a, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (a.B * a.F >= ((int64_t)1 << 29)) return ARMNET_ERR_UNSUPPORTED;   // 32-bit byte offsets into ids/vals
    if (!fused_mfma_supports(a.F, a.E, a.O)) return ARMNET_ERR_UNSUPPORTED;
    const int nq = (((a.F + 3) / 4) + 1) & ~1;            // quarter-steps per sample, rounded up to even
    // many neurons: the lane-ready parameter copies crowd the tiles out of LDS -> slices (each re-gathers the rows;
    // the in-place clamp is idempotent)
    const int slice = mfma_slice(a.F, a.E, a.O);
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
        if (a.model == MODEL_AFN) rc = launch_afn(s, a.E <= 16 ? 16 : a.E <= 32 ? 32 : 64, nq, st);
        else if (a.model == MODEL_GC_ARM)
            rc = a.E <= 16 ? launch_gc_e16(s, nq, st) : a.E <= 32 ? launch_gc_e32(s, nq, st) : launch_gc_e64(s, nq, st);
        else if (a.E <= 16) rc = launch_mfma_e16(s, nq, st);
        else if (a.E <= 32) rc = launch_mfma_e32(s, nq, st);
        else rc = launch_mfma_e64(s, nq, st);
        if (rc != ARMNET_OK) return rc;      // a refusal can only happen on the first slice (same shape after it)
    }
    return ARMNET_OK;
}

}  // namespace armnet
