// fused_gc_e128.hip — GC-ARM mode of the fused MFMA kernel, padded embedding width 128 (models/gc_arm.py; nemb 65..128, round 6).
#include "fused_mfma_kernel.h"

namespace armnet {

int launch_gc_e128(const FusedArgs& a, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_sibling<128, 2, MODEL_GC_ARM>(a, st);
        case 4: return launch_sibling<128, 4, MODEL_GC_ARM>(a, st);
        case 6: return launch_sibling<128, 6, MODEL_GC_ARM>(a, st);
        case 8: return launch_sibling<128, 8, MODEL_GC_ARM>(a, st);
        case 10: return launch_sibling<128, 10, MODEL_GC_ARM>(a, st);
        case 12: return launch_sibling<128, 12, MODEL_GC_ARM>(a, st);
        default: return ARMNET_ERR_UNSUPPORTED;
    }
}

}  // namespace armnet
