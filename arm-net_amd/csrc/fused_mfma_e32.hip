// fused_mfma_e32.hip — instantiations of the fused MFMA kernel for nemb padded to 32.
#include "fused_mfma_kernel.h"

namespace armnet {

int launch_mfma_e32(const FusedArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_src<32, 2, true>(a, st);
        case 4: return launch_src<32, 4, true>(a, st);
        case 6: return launch_src<32, 6, true>(a, st);
        case 8: return launch_src<32, 8, true>(a, st);
        case 10: return launch_src<32, 10, true>(a, st);
        case 12: return launch_src<32, 12, true>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
