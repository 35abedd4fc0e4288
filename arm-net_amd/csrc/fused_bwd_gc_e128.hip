// fused_bwd_gc_e128.hip — GC-ARM instantiations of the matrix-core backward kernel for nemb padded to 128 (nemb 65..128).
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

int launch_bwd_gc_e128(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_bwd_src<128, 2, MODEL_GC_ARM>(a, st, gx);
        case 4: return launch_bwd_src<128, 4, MODEL_GC_ARM>(a, st, gx);
        case 6: return launch_bwd_src<128, 6, MODEL_GC_ARM>(a, st, gx);
        case 8: return launch_bwd_src<128, 8, MODEL_GC_ARM>(a, st, gx);
        default: return ARMNET_ERR_UNSUPPORTED;    // 33+ fields x 128 floats: the wave tiles do not fit the LDS
    }
}

}  // namespace armnet
