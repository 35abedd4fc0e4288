// fused_mfma_e16h.hip — instantiations of the fused MFMA kernel for nemb padded to 16 with both contractions on the 16-bit
// matrix pipe (fp16 x 2 operand splits; F16 in fused_mfma_kernel.h): wide ARM blocks of 17+ fields (fewer fields: the fp32 form wins).
#include "fused_mfma_kernel.h"

namespace armnet {

int launch_mfma_e16_f16(const FusedArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 6: return launch_f16<6>(a, st);
        case 8: return launch_f16<8>(a, st);
        case 10: return launch_f16<10>(a, st);
        case 12: return launch_f16<12>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
