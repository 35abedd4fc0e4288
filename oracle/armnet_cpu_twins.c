/*
 * armnet_cpu_twins.c — the `_cpu` twins of the C ABI (include/armnet_hip.h): the same entry points with the same
 * arguments minus the stream, computed on the host by the reference's algorithm (SURVEY.md §8b lists them as part
 * of the boundary: "the CPU restatement").
 *
 * TEST INFRASTRUCTURE ONLY, like armnet_oracle.c whose stage functions they compose: a parity test hands ONE set of
 * arguments to armnet_fused_fwd_f32 (HIP) and to armnet_fused_fwd_f32_cpu and compares the outputs.  The product
 * path never loads this library.
 *
 * Differences from the stage-by-stage oracle: the twins take the FOLDED parameters of the ABI (q_fold, bn_scale,
 * bn_shift), so gates are x . q_fold and the BatchNorm is the affine; the sparse map is always the reference's
 * n_iter-step bisection (utils/entmax.py:29-68; softmax for alpha == 1), whatever solver the HIP side picks.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TW_OK 0
#define TW_BAD_ARG (-1)
#define TW_UNSUPPORTED (-2)
#define TW_F_WRITE_CLAMPED_VALS 0x1u

/* armnet_oracle.c */
void oracle_clamp_vals(float* vals, int64_t n);
void oracle_entmax_bisect(const float* X, int64_t rows, int d, float alpha, int n_iter, int ensure_sum_one, float* P);
void oracle_softmax(const float* X, int64_t rows, int d, float* P);
void oracle_interact_exp(const float* x, const float* p, const float* values, int64_t B, int F, int E, int O,
                         float* arm_weight, float* neurons);

static int64_t id_at(const void* ids, int id_type, int64_t i) {
    return id_type == 0 ? ((const int64_t*)ids)[i] : (int64_t)((const int32_t*)ids)[i];
}

/* include/armnet_hip.h: armnet_fold_params_f32 (contractions in double, one rounding — exact folds in real
 * arithmetic of armnet_1h.py:30-32 / armnet.py:33-34 and of eval BatchNorm1d) */
int armnet_fold_params_f32_cpu(int variant, int K, int H, int E, int D, const float* bilinear_w, const float* query,
                               const float* bn_weight, const float* bn_bias, const float* bn_running_mean,
                               const float* bn_running_var, float bn_eps, float* q_fold, float* bn_scale,
                               float* bn_shift) {
    if (K <= 0 || H <= 0 || E <= 0 || D <= 0 || (variant != 0 && variant != 1) || (variant == 0 && K != 1))
        return TW_BAD_ARG;
    if (!bilinear_w || !query || !bn_weight || !bn_bias || !bn_running_mean || !bn_running_var || !q_fold || !bn_scale ||
        !bn_shift)
        return TW_BAD_ARG;
    const double scale = pow((double)D, -0.5);
    for (int k = 0; k < K; ++k)
        for (int o = 0; o < H; ++o)
            for (int e = 0; e < E; ++e) {
                double acc = 0.0;
                for (int y = 0; y < D; ++y) {
                    /* one-head: nn.Linear weight [D,E]; multi-head: bilinear_w [K,E,D] */
                    const double w = variant == 0 ? bilinear_w[(size_t)y * E + e] : bilinear_w[((size_t)k * E + e) * D + y];
                    acc += w * (double)query[((size_t)k * H + o) * D + y];
                }
                q_fold[((size_t)k * H + o) * E + e] = (float)(scale * acc);
            }
    for (int c = 0; c < K * H; ++c) {
        const double s = (double)bn_weight[c] / sqrt((double)bn_running_var[c] + (double)bn_eps);
        bn_scale[c] = (float)s;
        bn_shift[c] = (float)((double)bn_bias[c] - (double)bn_running_mean[c] * s);
    }
    return TW_OK;
}

int armnet_clamp_vals_f32_cpu(float* vals, int64_t n) {
    if (n < 0 || (!vals && n > 0)) return TW_BAD_ARG;
    oracle_clamp_vals(vals, n);
    return TW_OK;
}

int armnet_gather_scale_f32_cpu(int64_t n_rows, int E, const void* ids, int id_type, const float* vals,
                                const float* table, int64_t nfeat, float* out, int32_t* id_status) {
    if (n_rows < 0 || E <= 0 || nfeat <= 0 || !ids || !table || !out || (id_type != 0 && id_type != 1)) return TW_BAD_ARG;
    for (int64_t r = 0; r < n_rows; ++r) {
        int64_t id = id_at(ids, id_type, r);
        if (id < 0 || id >= nfeat) {
            if (id_status) *id_status |= 1;
            id = 0;
        }
        const float v = vals ? vals[r] : 1.0f;
        for (int e = 0; e < E; ++e) out[r * E + e] = table[id * E + e] * v;
    }
    return TW_OK;
}

int armnet_entmax_f32_cpu(int64_t rows, int d, float alpha, int n_iter, int ensure_sum_one, uint32_t flags,
                          const float* X, float* P) {
    (void)flags;
    if (rows < 0 || d <= 0 || n_iter < 0 || !X || !P || !(alpha >= 1.0f)) return TW_BAD_ARG;
    if (alpha == 1.0f) oracle_softmax(X, rows, d, P);
    else oracle_entmax_bisect(X, rows, d, alpha, n_iter, ensure_sum_one, P);
    return TW_OK;
}

/* x [B,F,E] (already scaled) -> out [B,O,E] */
static int block_from_x(int64_t B, int F, int E, int O, float alpha, int n_iter, const float* x, const float* q_fold,
                        const float* values, const float* bn_scale, const float* bn_shift, float* out) {
    float* gates = (float*)malloc(sizeof(float) * (size_t)B * O * F);
    float* p = (float*)malloc(sizeof(float) * (size_t)B * O * F);
    float* w = (float*)malloc(sizeof(float) * (size_t)B * O * F);
    if (!gates || !p || !w) { free(gates); free(p); free(w); return TW_UNSUPPORTED; }
    for (int64_t b = 0; b < B; ++b)
        for (int o = 0; o < O; ++o)
            for (int f = 0; f < F; ++f) {
                float acc = 0.f;
                for (int e = 0; e < E; ++e) acc = fmaf(x[(b * F + f) * E + e], q_fold[(size_t)o * E + e], acc);
                gates[(b * O + o) * F + f] = acc;
            }
    if (alpha == 1.0f) oracle_softmax(gates, B * O, F, p);
    else oracle_entmax_bisect(gates, B * O, F, alpha, n_iter, 1, p);
    oracle_interact_exp(x, p, values, B, F, E, O, w, out);
    for (int64_t b = 0; b < B; ++b)
        for (int o = 0; o < O; ++o)
            for (int e = 0; e < E; ++e) {
                float* z = out + (b * O + o) * E + e;
                *z = *z * bn_scale[o] + bn_shift[o];
            }
    free(gates); free(p); free(w);
    return TW_OK;
}

static int fused_common(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags, const void* ids,
                        int id_type, const float* rows, float* vals, const float* table, int64_t nfeat,
                        const float* q_fold, const float* values, const float* bn_scale, const float* bn_shift,
                        float* out, int32_t* id_status) {
    if (B < 0 || F <= 0 || E <= 0 || O <= 0 || n_iter < 0 || !(alpha >= 1.0f)) return TW_BAD_ARG;
    if (B == 0) return TW_OK;
    if (!vals || !q_fold || !values || !bn_scale || !bn_shift || !out) return TW_BAD_ARG;
    float* v = vals;
    if (!(flags & TW_F_WRITE_CLAMPED_VALS)) {              /* clamp a private copy */
        v = (float*)malloc(sizeof(float) * (size_t)B * F);
        if (!v) return TW_UNSUPPORTED;
        memcpy(v, vals, sizeof(float) * (size_t)B * F);
    }
    oracle_clamp_vals(v, B * F);
    float* x = (float*)malloc(sizeof(float) * (size_t)B * F * E);
    if (!x) { if (v != vals) free(v); return TW_UNSUPPORTED; }
    int rc = TW_OK;
    if (rows) {
        for (int64_t i = 0; i < B * F; ++i)
            for (int e = 0; e < E; ++e) x[i * E + e] = rows[i * E + e] * v[i];
    } else {
        rc = armnet_gather_scale_f32_cpu(B * F, E, ids, id_type, v, table, nfeat, x, id_status);
    }
    if (rc == TW_OK) rc = block_from_x(B, F, E, O, alpha, n_iter, x, q_fold, values, bn_scale, bn_shift, out);
    free(x);
    if (v != vals) free(v);
    return rc;
}

int armnet_fused_fwd_f32_cpu(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags, const void* ids,
                             int id_type, float* vals, const float* table, int64_t nfeat, const float* q_fold,
                             const float* values, const float* bn_scale, const float* bn_shift, float* out,
                             int32_t* id_status) {
    if (B > 0 && (!ids || !table || nfeat <= 0 || (id_type != 0 && id_type != 1))) return TW_BAD_ARG;
    return fused_common(B, F, E, O, alpha, n_iter, flags, ids, id_type, NULL, vals, table, nfeat, q_fold, values,
                        bn_scale, bn_shift, out, id_status);
}

int armnet_fused_fwd_from_rows_f32_cpu(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                                       const float* rows, float* vals, const float* q_fold, const float* values,
                                       const float* bn_scale, const float* bn_shift, float* out) {
    if (B > 0 && !rows) return TW_BAD_ARG;
    return fused_common(B, F, E, O, alpha, n_iter, flags, NULL, 0, rows, vals, NULL, 0, q_fold, values, bn_scale,
                        bn_shift, out, NULL);
}
