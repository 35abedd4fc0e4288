// fused_mfma_e128b.hip — instantiations of the fused MFMA kernel for nemb padded to 128 (nemb 65..128), nfield 25..48.
#include "fused_mfma_kernel.h"

namespace armnet {

int launch_mfma_e128b(const FusedArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 8: return launch_src<128, 8, true>(a, st);
        case 10: return launch_src<128, 10, true>(a, st);
        case 12: return launch_src<128, 12, true>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
