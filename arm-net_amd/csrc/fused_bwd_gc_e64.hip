// fused_bwd_gc_e64.hip — GC-ARM instantiations of the matrix-core backward kernel for nemb padded to 64.
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

int launch_bwd_gc_e64(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_bwd_src<64, 2, MODEL_GC_ARM>(a, st, gx);
        case 4: return launch_bwd_src<64, 4, MODEL_GC_ARM>(a, st, gx);
        case 6: return launch_bwd_src<64, 6, MODEL_GC_ARM>(a, st, gx);
        case 8: return launch_bwd_src<64, 8, MODEL_GC_ARM>(a, st, gx);
        case 10: return launch_bwd_src<64, 10, MODEL_GC_ARM>(a, st, gx);
        case 12: return launch_bwd_src<64, 12, MODEL_GC_ARM>(a, st, gx);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
