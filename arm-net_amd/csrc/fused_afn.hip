// fused_afn.hip — AFN mode of the fused MFMA kernel (models/afn.py): the logarithmic transformation layer on the
// block's staging, tile layout and second contraction; no gates, no sparse map.
#include "fused_mfma_kernel.h"

namespace armnet {

template <int E>
static int launch_afn_e(const FusedArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_sibling<E, 2, MODEL_AFN>(a, st);
        case 4: return launch_sibling<E, 4, MODEL_AFN>(a, st);
        case 6: return launch_sibling<E, 6, MODEL_AFN>(a, st);
        case 8: return launch_sibling<E, 8, MODEL_AFN>(a, st);
        case 10: return launch_sibling<E, 10, MODEL_AFN>(a, st);
        case 12: return launch_sibling<E, 12, MODEL_AFN>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

int launch_afn(const FusedArgs& a, int ep, int nq, hipStream_t st) {
    return ep == 16 ? launch_afn_e<16>(a, nq, st) : ep == 32 ? launch_afn_e<32>(a, nq, st)
         : ep == 64 ? launch_afn_e<64>(a, nq, st) : launch_afn_e<128>(a, nq, st);     // 128: nemb 65..128 (round 6)
}

}  // namespace armnet
