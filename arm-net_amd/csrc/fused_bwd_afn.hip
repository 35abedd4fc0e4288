// fused_bwd_afn.hip — backward of AFN's logarithmic-transformation block (models/afn.py:56-69 under train.py:108-114)
// on the matrix cores: fused_bwd_mfma_kernel<..., MODEL_AFN> instantiations, the launcher over neuron slices, the C ABI.
#include "fused_bwd_mfma_kernel.h"

namespace armnet {

// nemb 4..128 (above 64: nfield <= 32), nfield <= 48, any afn_hid (slices); wider shapes keep the composed device ops (siblings.py)
static bool afn_bwd_supports(int F, int E, int O) { return !(E < 4 || E > 128 || O < 1 || F < 1 || F > (E > 64 ? 32 : 48)); }

template <int E>
static int launch_afn_nq(const BwdArgs& a, const BwdExtra& gx, int nq, hipStream_t st) {
    switch (nq) {
        case 2: return launch_bwd_afn_t<E, 2>(a, gx, st);
        case 4: return launch_bwd_afn_t<E, 4>(a, gx, st);
        case 6: return launch_bwd_afn_t<E, 6>(a, gx, st);
        case 8: return launch_bwd_afn_t<E, 8>(a, gx, st);
        default: break;
    }
    if constexpr (E <= 64) {                           // 33+ fields x 128 floats: the wave tiles do not fit the LDS
        if (nq == 10) return launch_bwd_afn_t<E, 10>(a, gx, st);
        if (nq == 12) return launch_bwd_afn_t<E, 12>(a, gx, st);
    }
    return ARMNET_ERR_UNSUPPORTED;
}

static int launch_afn_bwd(const BwdArgs& a, const BwdExtra& gx0, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (!afn_bwd_supports(a.F, a.E, a.O)) return ARMNET_ERR_UNSUPPORTED;
    if (a.B * a.F >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    const int nq = (((a.F + 3) / 4) + 1) & ~1;
    const int slice = 16 * bwd_passes_model(a.E <= 16 ? 16 : a.E <= 32 ? 32 : a.E <= 64 ? 64 : 128, MODEL_AFN);
    for (int o0 = 0; o0 < a.O; o0 += slice) {
        BwdArgs s = a;
        BwdExtra gx = gx0;
        gx.accumulate = o0 > 0;
        gx.d_bias = gx0.d_bias + o0;
        s.O = a.O - o0 < slice ? a.O - o0 : slice;
        s.O_all = a.O;
        s.values = a.values + (size_t)o0 * a.F;
        s.z = a.z + (size_t)o0 * a.E;
        s.dz = a.dz + (size_t)o0 * a.E;
        if (a.bn_a) { s.bn_a = a.bn_a + o0; s.bn_b = a.bn_b + o0; s.bn_c = a.bn_c + o0; }
        s.d_values = a.d_values + (size_t)o0 * a.F;
        const int rc = a.E <= 16 ? launch_afn_nq<16>(s, gx, nq, st) : a.E <= 32 ? launch_afn_nq<32>(s, gx, nq, st)
                     : a.E <= 64 ? launch_afn_nq<64>(s, gx, nq, st) : launch_afn_nq<128>(s, gx, nq, st);
        if (rc != ARMNET_OK) return rc;
    }
    return ARMNET_OK;
}

}  // namespace armnet

using namespace armnet;

extern "C" int armnet_afn_fused_bwd_supported(int F, int E, int O) { return afn_bwd_supports(F, E, O) ? 1 : 0; }

extern "C" int armnet_afn_fused_bwd_f32(int64_t B, int F, int E, int O, uint32_t flags, const void* ids, int id_type,
                                        const float* vals, const float* table, int64_t nfeat, const float* weight,
                                        const float* emb_scale, const float* emb_shift, const float* z, const float* dy,
                                        const float* coefA, const float* coefB, const float* coefC, float* d_weight,
                                        float* d_bias, float* d_y, void* stream) {
    if (B < 0 || F <= 0 || E <= 0 || O <= 0 || nfeat <= 0) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!ids || !vals || !table || !weight || !emb_scale || !emb_shift || !z || !dy || !d_weight || !d_bias || !d_y)
        return ARMNET_ERR_BAD_ARG;
    if ((coefA || coefB || coefC) && !(coefA && coefB && coefC)) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    BwdArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = ids; a.id_type = id_type; a.vals = vals; a.table = table; a.nfeat = nfeat;
    a.values = weight; a.z = z; a.dz = dy;
    a.bn_a = coefA; a.bn_b = coefB; a.bn_c = coefC;
    a.d_values = d_weight;
    a.cfg = make_sparse_cfg(1.0f, 0, F, 1, flags);
    a.alpha = 1.0f;
    a.flags = flags;
    BwdExtra gx{emb_scale, emb_shift, d_y, 0, d_bias};
    return launch_afn_bwd(a, gx, (hipStream_t)stream);
}
