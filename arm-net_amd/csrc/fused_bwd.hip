// fused_bwd.hip — backward of the fused ARM block, shape-agnostic, fp32.  gfx950.
//
// Given dZ = dLoss/d(neurons) with neurons z[b,o,:] = exp(sum_f w[b,o,f] x[b,f,:]) (pre-BatchNorm, the
// training-mode output of armnet_fused_fwd_f32 with an identity affine), per sample:
//     ds[o,e]  = dZ[o,e] * z[o,e]
//     dW[o,f]  = sum_e ds[o,e] x[f,e]                 dvalues[o,f] += p[o,f] * dW[o,f]     (armnet_1h.py:34)
//     dp[o,f]  = values[o,f] * dW[o,f]
//     dg[o,:]  = J_entmax(p[o,:])^T dp[o,:]           gppr = p^(2-alpha) on the support; dg = dp*gppr -
//                                                     gppr * sum(dp*gppr)/sum(gppr)        (entmax.py:70-80)
//                                                     alpha == 1: dg = p * (dp - sum(p*dp)) (softmax)
//     dq_fold[o,e] += sum_f dg[o,f] x[f,e]            (chain rule through the fold is done by the caller)
//     dx[f,e]  = sum_o w[o,f] ds[o,e] + dg[o,f] q_fold[o,e]
//     dtable[ids[f], :] += dx[f,:] * vals[f]          (x = table[id] * val, layers.py:20-21)
// p and w are recomputed from (ids, vals, table, q_fold) with the same solver as the forward.
// One 128/256-thread block owns S samples per iteration; thread (s,o) owns one neuron row; d_values and
// d_qfold are accumulated per block in LDS and flushed once with global atomics.
#include "armnet_common.h"

namespace armnet {


template <typename IdT, int TPB>
__global__ void __launch_bounds__(TPB) fused_bwd_kernel(BwdArgs a, int S) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int F = a.F, E = a.E, O = a.O;
    const int O_all = a.O_all ? a.O_all : O;          // neurons per sample in z / dz (this launch may be a slice)
    constexpr int CS = TPB + 1;                       // column stride (odd: conflict-free)
    float* xs = lds;                                  // [S][F][E]
    float* dss = xs + (((size_t)S * F * E + 3) & ~(size_t)3);      // [S][O][E]
    float* wcol = dss + (((size_t)S * O * E + 3) & ~(size_t)3);    // [F][CS]   p (then w = p*values)
    float* gcol = wcol + (size_t)F * CS;              // [F][CS]   dp (then dg)
    float* acc_dv = gcol + (size_t)F * CS;            // [O][F]
    float* acc_dq = acc_dv + (size_t)O * F;           // [O][E]
    float* vl = acc_dq + (size_t)O * E;               // [S][F] values
    uint32_t* idl = reinterpret_cast<uint32_t*>(vl + (size_t)S * F);   // [S][F] row ids
    const IdT* ids = reinterpret_cast<const IdT*>(a.ids);
    const int tid = threadIdx.x;
    const float alpha = a.alpha;

    for (int i = tid; i < O * F + O * E; i += TPB) acc_dv[i] = 0.f;   // acc_dv and acc_dq are contiguous

    for (int64_t b0 = (int64_t)blockIdx.x * S; b0 < a.B; b0 += (int64_t)gridDim.x * S) {
        const int ns = (int)((a.B - b0) < S ? (a.B - b0) : S);
        __syncthreads();
        for (int k = tid; k < ns * F; k += TPB) {
            const int64_t gi = b0 * F + k;
            bool bad;
            idl[k] = load_id_checked(ids + gi, a.nfeat, bad);
            vl[k] = a.vals[gi];
        }
        __syncthreads();
        for (int k = tid; k < ns * F * E; k += TPB) {
            const int sf = k / E, e = k - sf * E;
            xs[k] = a.table[(size_t)idl[sf] * E + e] * vl[sf];
        }
        __syncthreads();
        // ---- phase 1: thread (s,o) -----------------------------------------------------------------
        if (tid < ns * O) {
            const int s = tid / O, o = tid - s * O;
            const float* x = xs + (size_t)s * F * E;
            const float* qf = a.q_fold + (size_t)o * E;
            float* pw = wcol + tid;
            float* gg = gcol + tid;
            float* dsr = dss + ((size_t)s * O + o) * E;
            for (int f = 0; f < F; ++f) {
                float acc = 0.f;
                for (int e = 0; e < E; ++e) acc = fmaf(x[f * E + e], qf[e], acc);
                pw[f * CS] = acc;
            }
            sparse_map_row(pw, CS, F, a.cfg);                         // p (same solver as the forward)
            const size_t zo = ((size_t)(b0 + s) * O_all + o) * E;
            const float cA = a.bn_a ? a.bn_a[o] : 1.0f, cB = a.bn_a ? a.bn_b[o] : 0.f, cC = a.bn_a ? a.bn_c[o] : 0.f;
            for (int e = 0; e < E; ++e) {
                const float zv = a.z[zo + e];
                dsr[e] = fmaf(cA, a.dz[zo + e], fmaf(cC, zv, cB)) * zv;
            }
            // dW, dvalues, dp; entmax / softmax Jacobian-vector product
            float s1 = 0.f, s2 = 0.f;
            for (int f = 0; f < F; ++f) {
                float dW = 0.f;
                for (int e = 0; e < E; ++e) dW = fmaf(dsr[e], x[f * E + e], dW);
                const float p = pw[f * CS];
                const float val = a.values[(size_t)o * F + f];
                atomicAdd(&acc_dv[o * F + f], p * dW);
                const float dp = val * dW;
                if (alpha == 1.0f) {
                    gg[f * CS] = dp;
                    s1 += p * dp;
                } else {
                    const float gppr = p > 0.f ? (alpha == 2.0f ? 1.0f : (alpha == 1.5f ? sqrtf(p) : pow_pos(p, 2.0f - alpha))) : 0.f;
                    const float dxp = dp * gppr;
                    gg[f * CS] = dxp;
                    s1 += dxp;
                    s2 += gppr;
                }
            }
            // second sweep: finish dg (needs the row sums), then turn the p column into w = p * values
            const float q = alpha == 1.0f ? s1 : s1 / s2;
            for (int f = 0; f < F; ++f) {
                const float p = pw[f * CS];
                if (alpha == 1.0f) {
                    gg[f * CS] = p * (gg[f * CS] - q);
                } else {
                    const float gppr = p > 0.f ? (alpha == 2.0f ? 1.0f : (alpha == 1.5f ? sqrtf(p) : pow_pos(p, 2.0f - alpha))) : 0.f;
                    gg[f * CS] = gg[f * CS] - q * gppr;
                }
                pw[f * CS] = p * a.values[(size_t)o * F + f];
            }
        }
        __syncthreads();
        // ---- phase 2a: dx -> scatter into the table gradient -------------------------------------------
        for (int k = tid; k < ns * F * E; k += TPB) {
            const int sf = k / E, e = k - sf * E;
            const int s = sf / F, f = sf - s * F;
            float acc = 0.f;
            for (int o = 0; o < O; ++o) {
                const int col = s * O + o;
                acc = fmaf(wcol[f * CS + col], dss[((size_t)s * O + o) * E + e], acc);
                acc = fmaf(gcol[f * CS + col], a.q_fold[(size_t)o * E + e], acc);
            }
            atomicAdd(a.d_table + (size_t)idl[sf] * E + e, acc * vl[sf]);
        }
        // ---- phase 2b: d q_fold (block-local) -----------------------------------------------------------
        for (int k = tid; k < O * E; k += TPB) {
            const int o = k / E, e = k - o * E;
            float acc = 0.f;
            for (int s = 0; s < ns; ++s)
                for (int f = 0; f < F; ++f) acc = fmaf(gcol[f * CS + s * O + o], xs[((size_t)s * F + f) * E + e], acc);
            acc_dq[k] += acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < O * F; i += TPB) atomicAdd(a.d_values + i, acc_dv[i]);
    for (int i = tid; i < O * E; i += TPB) atomicAdd(a.d_qfold + i, acc_dq[i]);
}

template <typename IdT, int TPB>
static int launch_bwd_t(const BwdArgs& a, hipStream_t st) {
    int S = TPB / a.O;
    if (S < 1) return ARMNET_ERR_UNSUPPORTED;
    auto need = [&](int s_) {
        return (((size_t)s_ * a.F * a.E + 3) & ~(size_t)3) + (((size_t)s_ * a.O * a.E + 3) & ~(size_t)3) +
               2 * (size_t)a.F * (TPB + 1) + (size_t)a.O * a.F + (size_t)a.O * a.E + 2 * (size_t)s_ * a.F;
    };
    while (S > 1 && need(S) * sizeof(float) > 150 * 1024) --S;
    const size_t bytes = need(S) * sizeof(float);
    if (bytes > 150 * 1024) return ARMNET_ERR_UNSUPPORTED;
    auto kern = fused_bwd_kernel<IdT, TPB>;
    ARMNET_ALLOW_BIG_LDS(kern, bytes);
    int64_t grid = (a.B + S - 1) / S;
    if (grid > 1024) grid = 1024;
    kern<<<(int)grid, TPB, bytes, st>>>(a, S);
    ARMNET_LAUNCH_CHECK();
    return ARMNET_OK;
}

int launch_fused_bwd(const BwdArgs& a, hipStream_t st) {
    if (a.B == 0) return ARMNET_OK;
    if (!(a.flags & ARMNET_F_FORCE_GENERIC) && fused_bwd_mfma_supports(a.F, a.E, a.O)) {
        const int rc = launch_fused_bwd_mfma(a, st);
        if (rc != ARMNET_ERR_UNSUPPORTED) return rc;
    }
    // slices of <= 256 neurons (a thread per neuron row): the table gradient is additive over the neurons, the
    // parameter gradients and z / dz are addressed per slice — the forward accepts any neuron count, so must this.
    // Wide rows (nfield > 48) need the per-thread gate columns of 256 threads to fit LDS: else slices of 128.
    auto lds_need = [&](int tpb, int o) {
        return ((((size_t)a.F * a.E + 3) & ~(size_t)3) + (((size_t)o * a.E + 3) & ~(size_t)3) +
                2 * (size_t)a.F * (tpb + 1) + (size_t)o * a.F + (size_t)o * a.E + 2 * (size_t)a.F) * sizeof(float);
    };
    int slice = (a.O > 128 && lds_need(256, a.O < 256 ? a.O : 256) <= 150 * 1024) ? 256 : 128;
    // wide rows x wide embeddings (nemb > 64 with 40+ fields, round 4): shorter neuron slices until one sample's tiles fit
    while (slice > 16 && lds_need(128, a.O < slice ? a.O : slice) > 150 * 1024) slice /= 2;
    for (int o0 = 0; o0 < a.O; o0 += slice) {
        BwdArgs s = a;
        s.O = a.O - o0 < slice ? a.O - o0 : slice;
        s.O_all = a.O_all ? a.O_all : a.O;
        s.q_fold = a.q_fold + (size_t)o0 * a.E;
        s.values = a.values + (size_t)o0 * a.F;
        s.z = a.z + (size_t)o0 * a.E;
        s.dz = a.dz + (size_t)o0 * a.E;
        if (a.bn_a) { s.bn_a = a.bn_a + o0; s.bn_b = a.bn_b + o0; s.bn_c = a.bn_c + o0; }
        s.d_values = a.d_values + (size_t)o0 * a.F;
        s.d_qfold = a.d_qfold + (size_t)o0 * a.E;
        const int rc = s.O <= 128
            ? (a.id_type == ARMNET_ID_I64 ? launch_bwd_t<int64_t, 128>(s, st) : launch_bwd_t<int32_t, 128>(s, st))
            : (a.id_type == ARMNET_ID_I64 ? launch_bwd_t<int64_t, 256>(s, st) : launch_bwd_t<int32_t, 256>(s, st));
        if (rc != ARMNET_OK) return rc;
    }
    return ARMNET_OK;
}

}  // namespace armnet

using namespace armnet;

static int fused_bwd_impl(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags, const void* ids,
                          int id_type, const float* vals, const float* table, int64_t nfeat, const float* q_fold,
                          const float* values, const float* z, const float* dz, const float* bn_a, const float* bn_b,
                          const float* bn_c, float* d_table, float* d_values, float* d_qfold, void* stream) {
    if (B < 0 || F <= 0 || E <= 0 || O <= 0 || n_iter < 0 || nfeat <= 0) return ARMNET_ERR_BAD_ARG;
    if (B == 0) return ARMNET_OK;
    if (!ids || !vals || !table || !q_fold || !values || !z || !dz || !d_table || !d_values || !d_qfold)
        return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (!(alpha >= 1.0f)) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    BwdArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = ids; a.id_type = id_type; a.vals = vals; a.table = table; a.nfeat = nfeat;
    a.q_fold = q_fold; a.values = values; a.z = z; a.dz = dz;
    a.bn_a = bn_a; a.bn_b = bn_b; a.bn_c = bn_c;
    a.d_table = d_table; a.d_values = d_values; a.d_qfold = d_qfold;
    a.cfg = make_sparse_cfg(alpha, n_iter, F, 1, flags);
    a.alpha = alpha;
    a.flags = flags;
    return launch_fused_bwd(a, (hipStream_t)stream);
}

extern "C" int armnet_fused_bwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                                    const void* ids, int id_type, const float* vals, const float* table,
                                    int64_t nfeat, const float* q_fold, const float* values, const float* z,
                                    const float* dz, float* d_table, float* d_values, float* d_qfold,
                                    void* stream) {
    return fused_bwd_impl(B, F, E, O, alpha, n_iter, flags, ids, id_type, vals, table, nfeat, q_fold, values, z, dz,
                          nullptr, nullptr, nullptr, d_table, d_values, d_qfold, stream);
}

extern "C" int armnet_fused_bwd_bn_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                                       const void* ids, int id_type, const float* vals, const float* table,
                                       int64_t nfeat, const float* q_fold, const float* values, const float* z,
                                       const float* dy, const float* coefA, const float* coefB, const float* coefC,
                                       float* d_table, float* d_values, float* d_qfold, void* stream) {
    if (!coefA || !coefB || !coefC) return ARMNET_ERR_BAD_ARG;
    return fused_bwd_impl(B, F, E, O, alpha, n_iter, flags, ids, id_type, vals, table, nfeat, q_fold, values, z, dy,
                          coefA, coefB, coefC, d_table, d_values, d_qfold, stream);
}
