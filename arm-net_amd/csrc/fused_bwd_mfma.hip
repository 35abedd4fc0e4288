// fused_bwd_mfma.hip — shape dispatch of the matrix-core backward kernel (fused_bwd_mfma_kernel.h).
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

// nemb 4..128 (any, odd too; above 64: nfield <= 32), nfield <= 48; any neuron count (slices)
bool fused_bwd_mfma_supports(int F, int E, int O) {
    return !(E < 4 || E > 128 || O < 1 || F < 1 || F > (E > 64 ? 32 : 48));
}

int launch_fused_bwd_mfma(const BwdArgs& a, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (!fused_bwd_mfma_supports(a.F, a.E, a.O)) return ARMNET_ERR_UNSUPPORTED;
    if (a.B * a.F >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    const int nq = (((a.F + 3) / 4) + 1) & ~1;
    // slices of 32 (nemb = 64: 16) neurons: each re-stages the rows and adds its part of dx to the table gradient
    const int slice = 16 * bwd_passes(a.E <= 16 ? 16 : a.E <= 32 ? 32 : a.E <= 64 ? 64 : 128);
    for (int o0 = 0; o0 < a.O; o0 += slice) {
        BwdArgs s = a;
        s.O = a.O - o0 < slice ? a.O - o0 : slice;
        s.O_all = a.O_all ? a.O_all : a.O;
        s.q_fold = a.q_fold + (size_t)o0 * a.E;
        s.values = a.values + (size_t)o0 * a.F;
        s.z = a.z + (size_t)o0 * a.E;
        s.dz = a.dz + (size_t)o0 * a.E;
        if (a.bn_a) { s.bn_a = a.bn_a + o0; s.bn_b = a.bn_b + o0; s.bn_c = a.bn_c + o0; }
        s.d_values = a.d_values + (size_t)o0 * a.F;
        s.d_qfold = a.d_qfold + (size_t)o0 * a.E;
        int rc;
        if (a.E <= 16) rc = launch_bwd_mfma_e16(s, nq, st);
        else if (a.E <= 32) rc = launch_bwd_mfma_e32(s, nq, st);
        else if (a.E <= 64) rc = launch_bwd_mfma_e64(s, nq, st);
        else rc = launch_bwd_mfma_e128(s, nq, st);
        if (rc != ARMNET_OK) return rc;
    }
    return ARMNET_OK;
}

}  // namespace armnet
