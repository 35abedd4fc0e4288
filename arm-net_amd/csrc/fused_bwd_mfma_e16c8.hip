// fused_bwd_mfma_e16c8.hip — instantiations of the matrix-core backward kernel for nemb padded to 16, 8-byte staging chunks.
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

int launch_bwd_mfma_e16_c8(const BwdArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_bwd_src<16, 2, 8>(a, st);
        case 4: return launch_bwd_src<16, 4, 8>(a, st);
        case 6: return launch_bwd_src<16, 6, 8>(a, st);
        case 8: return launch_bwd_src<16, 8, 8>(a, st);
        case 10: return launch_bwd_src<16, 10, 8>(a, st);
        case 12: return launch_bwd_src<16, 12, 8>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
