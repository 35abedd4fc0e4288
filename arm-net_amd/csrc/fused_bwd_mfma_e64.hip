// fused_bwd_mfma_e64.hip — instantiations of the matrix-core backward kernel for nemb padded to 64.
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

int launch_bwd_mfma_e64(const BwdArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_bwd_src<64, 2>(a, st);
        case 4: return launch_bwd_src<64, 4>(a, st);
        case 6: return launch_bwd_src<64, 6>(a, st);
        case 8: return launch_bwd_src<64, 8>(a, st);
        case 10: return launch_bwd_src<64, 10>(a, st);
        case 12: return launch_bwd_src<64, 12>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
