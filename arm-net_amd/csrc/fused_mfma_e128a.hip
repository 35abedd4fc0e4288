// fused_mfma_e128a.hip — instantiations of the fused MFMA kernel for nemb padded to 128 (nemb 65..128), nfield <= 24.
#include "fused_mfma_kernel.h"

namespace armnet {

int launch_mfma_e128a(const FusedArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_src<128, 2, true>(a, st);
        case 4: return launch_src<128, 4, true>(a, st);
        case 6: return launch_src<128, 6, true>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
