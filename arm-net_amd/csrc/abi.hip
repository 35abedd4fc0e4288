// abi.hip — the extern "C" boundary declared in include/armnet_hip.h.  Argument validation,
// solver selection and kernel dispatch only; no allocation, no synchronisation, no global state.
#include <stdio.h>
#include <string.h>

#include "armnet_common.h"

namespace armnet {

static thread_local char g_hip_err[256] = "";

void set_hip_error(hipError_t e, const char* where) {
    snprintf(g_hip_err, sizeof(g_hip_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
}

int device_cu_count() {
    static thread_local int cached_dev = -1, cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        cached_cus = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cached_dev = dev;
    }
    return cached_cus;
}

}  // namespace armnet

using namespace armnet;

extern "C" {

int armnet_abi_version(void) { return ARMNET_ABI_VERSION; }

const char* armnet_strerror(int status) {
    switch (status) {
        case ARMNET_OK: return "ok";
        case ARMNET_ERR_BAD_ARG: return "bad argument (null pointer or non-positive size)";
        case ARMNET_ERR_UNSUPPORTED: return "unsupported shape (per-block LDS tile would exceed 64 KiB, or nfeat >= 2^31)";
        case ARMNET_ERR_ID_RANGE: return "index out of range in self";
        case ARMNET_ERR_HIP: return "HIP runtime error (see armnet_last_hip_error)";
        default: return "unknown status";
    }
}

const char* armnet_last_hip_error(void) { return g_hip_err; }

int armnet_fold_params_f32(int variant, int K, int H, int E, int D, const float* bilinear_w,
                           const float* query, const float* bn_weight, const float* bn_bias,
                           const float* bn_running_mean, const float* bn_running_var, float bn_eps,
                           float* q_fold, float* bn_scale, float* bn_shift, void* stream) {
    if (K <= 0 || H <= 0 || E <= 0 || D <= 0) return ARMNET_ERR_BAD_ARG;
    if (variant != ARMNET_ONE_HEAD && variant != ARMNET_MULTI_HEAD && variant != ARMNET_GC_ARM) return ARMNET_ERR_BAD_ARG;
    if (variant == ARMNET_ONE_HEAD && K != 1) return ARMNET_ERR_BAD_ARG;
    if (!bilinear_w || !query || !bn_weight || !bn_bias || !bn_running_mean || !bn_running_var || !q_fold ||
        !bn_scale || !bn_shift)
        return ARMNET_ERR_BAD_ARG;
    return launch_fold_params(variant, K, H, E, D, bilinear_w, query, bn_weight, bn_bias, bn_running_mean,
                              bn_running_var, bn_eps, q_fold, bn_scale, bn_shift, (hipStream_t)stream);
}

int armnet_fused_kernel_kind(int F, int E, int O, float alpha, int n_iter, uint32_t flags) {
    if (F <= 0 || E <= 0 || O <= 0 || n_iter < 0 || !(alpha >= 1.0f)) return ARMNET_ERR_BAD_ARG;
    const SparseMapCfg cfg = make_sparse_cfg(alpha, n_iter, F, 1, flags);
    (void)cfg;
    return (!(flags & ARMNET_F_FORCE_GENERIC) && fused_mfma_supports(F, E, O)) ? 1 : 0;
}

int armnet_sibling_kernel_kind(int afn, int F, int E, int O) {
    if (F <= 0 || E <= 0 || O <= 0) return ARMNET_ERR_BAD_ARG;
    return fused_mfma_supports_model(F, E, O, afn ? MODEL_AFN : MODEL_GC_ARM) ? 1 : 0;
}

static int fused_common(FusedArgs& a, float alpha, int n_iter, void* stream) {
    if (a.B < 0 || a.F <= 0 || a.E <= 0 || a.O <= 0 || n_iter < 0) return ARMNET_ERR_BAD_ARG;
    if (a.B == 0) return ARMNET_OK;                       // empty batch: pointers may be null
    if (!a.vals || !a.q_fold || !a.values || !a.bn_scale || !a.bn_shift || !a.out) return ARMNET_ERR_BAD_ARG;
    if (!(alpha >= 1.0f)) return ARMNET_ERR_BAD_ARG;
    a.cfg = make_sparse_cfg(alpha, n_iter, a.F, 1, a.flags);
    if (!(a.flags & ARMNET_F_FORCE_GENERIC) && fused_mfma_supports(a.F, a.E, a.O)) {
        const int rc = launch_fused_mfma(a, (hipStream_t)stream);
        if (rc != ARMNET_ERR_UNSUPPORTED) return rc;
    }
    return launch_fused_generic(a, (hipStream_t)stream);
}

int armnet_fused_fwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                         const void* ids, int id_type, float* vals, const float* table, int64_t nfeat,
                         const float* q_fold, const float* values, const float* bn_scale,
                         const float* bn_shift, float* out, int32_t* id_status, void* stream) {
    if (B == 0 && F > 0 && E > 0 && O > 0) return ARMNET_OK;
    if (!ids || !table || nfeat <= 0) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    FusedArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = ids; a.id_type = id_type; a.rows = nullptr; a.vals = vals;
    a.table = table; a.nfeat = nfeat;
    a.q_fold = q_fold; a.values = values; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.out = out; a.id_status = id_status; a.flags = flags;
    return fused_common(a, alpha, n_iter, stream);
}

int armnet_gc_fused_fwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                            const void* ids, int id_type, float* vals, const float* table, int64_t nfeat,
                            const float* q_fold, const float* values, const float* emb_scale, const float* emb_shift,
                            const float* bn_scale, const float* bn_shift, float* out, int32_t* id_status, void* stream) {
    if (B == 0 && F > 0 && E > 0 && O > 0) return ARMNET_OK;
    if (!ids || !table || nfeat <= 0 || !emb_scale || !emb_shift) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    FusedArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = ids; a.id_type = id_type; a.rows = nullptr; a.vals = vals;
    a.table = table; a.nfeat = nfeat;
    a.q_fold = q_fold; a.values = values; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.out = out; a.id_status = id_status; a.flags = flags;
    a.model = MODEL_GC_ARM; a.emb_scale = emb_scale; a.emb_shift = emb_shift;
    return fused_common(a, alpha, n_iter, stream);
}

int armnet_afn_fused_fwd_f32(int64_t B, int F, int E, int O, uint32_t flags, const void* ids, int id_type, float* vals,
                             const float* table, int64_t nfeat, const float* weight, const float* bias,
                             const float* emb_scale, const float* emb_shift, const float* bn_scale,
                             const float* bn_shift, float* out, int32_t* id_status, void* stream) {
    if (B == 0 && F > 0 && E > 0 && O > 0) return ARMNET_OK;
    if (!ids || !table || nfeat <= 0 || !emb_scale || !emb_shift || !weight || !bias) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    FusedArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = ids; a.id_type = id_type; a.rows = nullptr; a.vals = vals;
    a.table = table; a.nfeat = nfeat;
    a.q_fold = weight;              // unused by the AFN path; non-null for the common argument check
    a.values = weight; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.out = out; a.id_status = id_status; a.flags = flags;
    a.model = MODEL_AFN; a.emb_scale = emb_scale; a.emb_shift = emb_shift; a.lin_bias = bias;
    return fused_common(a, 1.0f, 0, stream);
}

int armnet_fold_bn_f32(int C, const float* weight, const float* bias, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, void* stream) {
    if (C <= 0 || !weight || !bias || !running_mean || !running_var || !scale || !shift) return ARMNET_ERR_BAD_ARG;
    return launch_fold_bn(C, weight, bias, running_mean, running_var, eps, scale, shift, (hipStream_t)stream);
}

int armnet_abs_clamp_min_f32(float* p, int64_t n, float lo, void* stream) {
    if (n < 0 || (!p && n > 0)) return ARMNET_ERR_BAD_ARG;
    return launch_abs_clamp_min(p, n, lo, (hipStream_t)stream);
}

int armnet_fused_fwd_from_rows_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                                   const float* rows, float* vals, const float* q_fold,
                                   const float* values, const float* bn_scale, const float* bn_shift,
                                   float* out, void* stream) {
    if (B == 0 && F > 0 && E > 0 && O > 0) return ARMNET_OK;
    if (!rows) return ARMNET_ERR_BAD_ARG;
    FusedArgs a{};
    a.B = B; a.F = F; a.E = E; a.O = O;
    a.ids = nullptr; a.id_type = ARMNET_ID_I64; a.rows = rows; a.vals = vals;
    a.table = nullptr; a.nfeat = 0;
    a.q_fold = q_fold; a.values = values; a.bn_scale = bn_scale; a.bn_shift = bn_shift;
    a.out = out; a.id_status = nullptr; a.flags = flags;
    return fused_common(a, alpha, n_iter, stream);
}

int armnet_gather_scale_f32(int64_t n_rows, int E, const void* ids, int id_type, const float* vals,
                            const float* table, int64_t nfeat, float* out, int32_t* id_status,
                            void* stream) {
    if (n_rows < 0 || E <= 0 || nfeat <= 0 || !ids || !table || !out) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_gather_scale(n_rows, E, ids, id_type, vals, table, nfeat, out, id_status,
                               (hipStream_t)stream);
}

int armnet_scatter_add_f32(int64_t n_rows, int E, const void* ids, int id_type, const float* vals, const float* grad,
                           int64_t nfeat, float* d_table, void* stream) {
    if (n_rows < 0 || E <= 0 || !ids || !grad || !d_table || nfeat <= 0) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    return launch_scatter_add(n_rows, E, ids, id_type, vals, grad, nfeat, d_table, (hipStream_t)stream);
}

int armnet_clamp_vals_f32(float* vals, int64_t n, void* stream) {
    if (n < 0 || (!vals && n > 0)) return ARMNET_ERR_BAD_ARG;
    return launch_clamp_vals(vals, n, (hipStream_t)stream);
}

int armnet_entmax_f32(int64_t rows, int d, float alpha, int n_iter, int ensure_sum_one, uint32_t flags,
                      const float* X, float* P, void* stream) {
    if (rows < 0 || d <= 0 || n_iter < 0 || !X || !P) return ARMNET_ERR_BAD_ARG;
    if (!(alpha >= 1.0f)) return ARMNET_ERR_BAD_ARG;
    const SparseMapCfg cfg = make_sparse_cfg(alpha, n_iter, d, ensure_sum_one, flags);
    return launch_entmax(rows, d, cfg, nullptr, X, P, (hipStream_t)stream);
}

int armnet_entmax_rows_f32(int64_t rows, int d, const float* alpha_rows, int n_iter, int ensure_sum_one, const float* X,
                           float* P, void* stream) {
    if (rows < 0 || d <= 0 || n_iter < 0) return ARMNET_ERR_BAD_ARG;
    if (rows == 0) return ARMNET_OK;
    if (!alpha_rows || !X || !P) return ARMNET_ERR_BAD_ARG;
    // the per-row fields of the configuration are filled in by the kernel (row_cfg); alpha = 3 only selects the bisection
    const SparseMapCfg cfg = make_sparse_cfg(3.0f, n_iter, d, ensure_sum_one, ARMNET_F_FAITHFUL_BISECT);
    return launch_entmax(rows, d, cfg, alpha_rows, X, P, (hipStream_t)stream);
}

int armnet_entmax_bwd_f32(int64_t rows, int d, float alpha, const float* Y, const float* dY, float* dX, void* stream) {
    if (rows < 0 || d <= 0 || !(alpha >= 1.0f)) return ARMNET_ERR_BAD_ARG;
    if (rows == 0) return ARMNET_OK;
    if (!Y || !dY || !dX) return ARMNET_ERR_BAD_ARG;
    return launch_entmax_bwd(rows, d, alpha, Y, dY, dX, (hipStream_t)stream);
}

int64_t armnet_shard_route_ws_bytes(int64_t n, int R) {
    if (n < 0 || R < 1) return -1;
    return (int64_t)shard_route_ws_bytes(n, R);
}

int armnet_shard_route_ids(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* counts,
                           int32_t* send_local, int32_t* perm, void* workspace, int64_t ws_bytes,
                           int32_t* id_status, void* stream) {
    if (n < 0 || R < 1 || nfeat <= 0 || !counts || (n > 0 && (!ids || !send_local || !perm))) return ARMNET_ERR_BAD_ARG;
    if (id_type != ARMNET_ID_I64 && id_type != ARMNET_ID_I32) return ARMNET_ERR_BAD_ARG;
    if (nfeat >= ((int64_t)1 << 31)) return ARMNET_ERR_UNSUPPORTED;
    return launch_shard_route(n, ids, id_type, R, nfeat, counts, send_local, perm, workspace, (size_t)ws_bytes,
                              id_status, (hipStream_t)stream);
}

}  // extern "C"
