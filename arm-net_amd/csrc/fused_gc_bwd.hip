// fused_gc_bwd.hip — backward of GC-ARM's block (models/gc_arm.py:82-95 under train.py:108-114) on the matrix cores:
// launcher over neuron slices + the C ABI.  The kernel is fused_bwd_mfma_kernel<..., MODEL_GC_ARM>.
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

// nemb 4..128 (above 64: nfield <= 32, like ARM-Net's own backward), nfield <= 48, any neuron count (slices); wider shapes keep
// the composed device ops (siblings.py)
static bool gc_bwd_supports(int F, int E, int O) { return !(E < 4 || E > 128 || O < 1 || F < 1 || F > (E > 64 ? 32 : 48)); }

static int launch_gc_bwd(const BwdArgs& a, const BwdExtra& gx0, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (!gc_bwd_supports(a.F, a.E, a.O)) return ARMNET_ERR_UNSUPPORTED;
    if (a.B * a.F >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    const int nq = (((a.F + 3) / 4) + 1) & ~1;
    const int slice = 16 * bwd_passes_model(a.E <= 16 ? 16 : a.E <= 32 ? 32 : a.E <= 64 ? 64 : 128, MODEL_GC_ARM);
    for (int o0 = 0; o0 < a.O; o0 += slice) {
        BwdArgs s = a;
        BwdExtra gx = gx0;
        gx.accumulate = o0 > 0;                        // d_y: the first slice writes, the later ones add (same stream)
        s.O = a.O - o0 < slice ? a.O - o0 : slice;
        s.O_all = a.O;
        s.q_fold = a.q_fold + (size_t)o0 * a.E;
        s.values = a.values + (size_t)o0 * a.F;
        s.z = a.z + (size_t)o0 * a.E;
        s.dz = a.dz + (size_t)o0 * a.E;
        if (a.bn_a) { s.bn_a = a.bn_a + o0; s.bn_b = a.bn_b + o0; s.bn_c = a.bn_c + o0; }
        s.d_values = a.d_values + (size_t)o0 * a.F;
        s.d_qfold = a.d_qfold + (size_t)o0 * a.E;
        const int rc = a.E <= 16 ? launch_bwd_gc_e16(s, gx, nq, st) : a.E <= 32 ? launch_bwd_gc_e32(s, gx, nq, st)
                     : a.E <= 64 ? launch_bwd_gc_e64(s, gx, nq, st) : launch_bwd_gc_e128(s, gx, nq, st);
        if (rc != ARMNET_OK) return rc;
    }
    return ARMNET_OK;
}

}  // namespace armnet

using namespace armnet;

extern "C" int armnet_gc_fused_bwd_supported(int F, int E, int O) { return gc_bwd_supports(F, E, O) ? 1 : 0; }

extern "C" int armnet_gc_fused_bwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                                       const void* ids, int id_type, const float* vals, const float* table,
                                       int64_t nfeat, const float* q_fold, const float* values, const float* emb_scale,
                                       const float* emb_shift, const float* z, const float* dy, const float* coefA,
                                       const float* coefB, const float* coefC, float* d_table, float* d_values,
                                       float* d_qfold, float* d_y, void* stream) {
    if (B < 0 || F <= 0 || E <= 0 || O <= 0 || n_iter < 0 || nfeat <= 0) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!ids || !vals || !table || !q_fold || !values || !emb_scale || !emb_shift || !z || !dy || !d_table || !d_values ||
        !d_qfold || !d_y)
        return ARMNET_ERR_BAD_ARG;
    if ((coefA || coefB || coefC) && !(coefA && coefB && coefC)) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (!(alpha >= 1.0f)) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    BwdArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = ids; a.id_type = id_type; a.vals = vals; a.table = table; a.nfeat = nfeat;
    a.q_fold = q_fold; a.values = values; a.z = z; a.dz = dy;
    a.bn_a = coefA; a.bn_b = coefB; a.bn_c = coefC;
    a.d_table = d_table; a.d_values = d_values; a.d_qfold = d_qfold;
    a.cfg = make_sparse_cfg(alpha, n_iter, F, 1, flags);
    a.alpha = alpha;
    a.flags = flags;
    BwdExtra gx{emb_scale, emb_shift, d_y, 0, nullptr};
    return launch_gc_bwd(a, gx, (hipStream_t)stream);
}
