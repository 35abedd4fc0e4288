/*
 * armnet_oracle.c — CPU restatement of the ARM-Net forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP
 * kernels in arm-net_amd/csrc and the `cpu_baseline` leg of bench.py.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it;
 * the product path (arm-net_amd/) never links, imports or calls it.
 *
 * It restates, stage by stage and WITHOUT the algebraic folds the HIP path
 * uses, what the reference computes with ATen ops (file:line cite the
 * read-only reference tree):
 *
 *   clamp        x['value'].clamp_(1e-3, 1.)         models/armnet_1h.py:81, models/armnet.py:82
 *   embed        emb(ids) * vals.unsqueeze(2)         models/layers.py:20-21
 *   gates (1h)   Linear(E->D, no bias) then einsum    models/armnet_1h.py:30-32
 *   gates (mh)   einsum('bfx,kxy,koy->bkof') * scale  models/armnet.py:33-34
 *   entmax       50-step bisection on tau             utils/entmax.py:29-68
 *   softmax      nn.Softmax(dim=-1) when alpha == 1   models/armnet_1h.py:12, models/armnet.py:12
 *   weighting    einsum('bof,of->bof', p, values)     models/armnet_1h.py:34, models/armnet.py:36
 *   interaction  exp(einsum('bfe,bof->boe'))          models/armnet_1h.py:85-86, models/armnet.py:86-87
 *   arm_bn       BatchNorm1d(channels=O), eval/train  models/armnet_1h.py:65,85, models/armnet.py:67,88-89
 *   MLP          (Linear,BN1d,ReLU,Dropout)*n,Linear  models/layers.py:68-88
 *
 * Parity pin: validated against the golden vectors (tests/golden, .npz files) that
 * tests/golden/make_golden.py captured from the real reference running on
 * CPU (torch 2.10) — see tests/test_oracle_golden.py.  The reference has no
 * tests or known-answer vectors of its own.
 *
 * Arithmetic: all fp32, like the reference's tensors.  Contractions are
 * sequential fmaf chains (the order inside ATen's BLAS calls is unspecified;
 * measured difference to the golden vectors is <= 3e-7 abs).  pow is libm
 * powf where the reference calls aten::pow (Sleef): <= 1 ulp apart.
 *
 * Build: see oracle/Makefile (gcc -O2 -mfma -mavx2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

#define ORACLE_OK 0
#define ORACLE_ERR_ID_RANGE (-3)
#define ORACLE_ERR_BAD_ARG (-1)

int armnet_oracle_version(void) { return 1; }

int armnet_oracle_max_threads(void) {
#if defined(_OPENMP)
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void armnet_oracle_set_threads(int n) {
#if defined(_OPENMP)
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* models/armnet_1h.py:81 — in place, on the caller's buffer. */
void oracle_clamp_vals(float* vals, int64_t n) {
    for (int64_t i = 0; i < n; ++i) {
        float v = vals[i];
        /* torch.clamp propagates NaN; fminf/fmaxf would not */
        if (v < 1e-3f) v = 1e-3f;
        if (v > 1.0f) v = 1.0f;
        vals[i] = v;
    }
}

/* models/layers.py:20-21 — x[b,f,:] = table[ids[b,f],:] * vals[b,f].
 * Out-of-range ids: the reference raises IndexError (aten::embedding). */
int oracle_embed(const int64_t* ids, const float* vals, const float* table, int64_t nfeat,
                 int64_t B, int F, int E, float* x) {
    for (int64_t i = 0; i < B * F; ++i) {
        int64_t id = ids[i];
        if (id < 0 || id >= nfeat) return ORACLE_ERR_ID_RANGE;
        const float* row = table + id * E;
        float v = vals[i];
        float* o = x + i * E;
        for (int e = 0; e < E; ++e) o[e] = row[e] * v;
    }
    return ORACLE_OK;
}

/* models/armnet_1h.py:30-32 — keys = x @ W^T (nn.Linear, W:[D,E]);
 * gates[b,o,f] = (sum_d keys[b,f,d] * q[o,d]) * D^-0.5. */
void oracle_gates_1h(const float* x, const float* W, const float* q, int64_t B, int F, int E, int D,
                     int H, float* gates) {
    const float scale = (float)pow((double)D, -0.5); /* python: d_k ** -0.5 (double), then * fp32 tensor */
    float* keys = (float*)malloc(sizeof(float) * (size_t)F * D);
    for (int64_t b = 0; b < B; ++b) {
        const float* xb = x + b * F * E;
        for (int f = 0; f < F; ++f)
            for (int d = 0; d < D; ++d) {
                float acc = 0.f;
                for (int e = 0; e < E; ++e) acc = fmaf(xb[f * E + e], W[d * E + e], acc);
                keys[f * D + d] = acc;
            }
        float* gb = gates + b * H * F;
        for (int o = 0; o < H; ++o)
            for (int f = 0; f < F; ++f) {
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc = fmaf(keys[f * D + d], q[o * D + d], acc);
                gb[o * F + f] = acc * scale;
            }
    }
    free(keys);
}

/* models/armnet.py:33-34 — einsum('bfx,kxy,koy->bkof') * scale, contracted left to right
 * (no opt_einsum in the reference's environment): t[b,f,k,y] = sum_x x[b,f,x] w[k,x,y];
 * gates[b,k,o,f] = (sum_y t[b,f,k,y] q[k,o,y]) * D^-0.5. */
void oracle_gates_mh(const float* x, const float* bw, const float* q, int64_t B, int F, int E, int D,
                     int K, int H, float* gates) {
    const float scale = (float)pow((double)D, -0.5);
    float* t = (float*)malloc(sizeof(float) * (size_t)F * K * D);
    for (int64_t b = 0; b < B; ++b) {
        const float* xb = x + b * F * E;
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < K; ++k)
                for (int y = 0; y < D; ++y) {
                    float acc = 0.f;
                    for (int e = 0; e < E; ++e) acc = fmaf(xb[f * E + e], bw[(k * E + e) * D + y], acc);
                    t[(f * K + k) * D + y] = acc;
                }
        float* gb = gates + b * K * H * F;
        for (int k = 0; k < K; ++k)
            for (int o = 0; o < H; ++o)
                for (int f = 0; f < F; ++f) {
                    float acc = 0.f;
                    for (int y = 0; y < D; ++y)
                        acc = fmaf(t[(f * K + k) * D + y], q[(k * H + o) * D + y], acc);
                    gb[(k * H + o) * F + f] = acc * scale;
                }
    }
    free(t);
}

/* utils/entmax.py:24-26 — _p(X, alpha) = clamp(X, min=0) ** (1/(alpha-1)) */
static inline float entmax_p(float x, float inv_am1) {
    float c = x > 0.f ? x : (x != x ? x : 0.f); /* clamp(min=0) keeps NaN */
    return powf(c, inv_am1);
}

/* utils/entmax.py:29-68 — EntmaxBisectFunction.forward over the last dim, one row at a time.
 * Statement-for-statement:  :42 X*(a-1)  :44 max  :46 tau_lo  :47 tau_hi  :49 f_lo  :51 dm
 * :53-61 loop (dm/=2; tau_m; p_m; f_m; mask = f_m*f_lo >= 0; tau_lo = where(mask,tau_m,tau_lo))
 * :63-64 p_m /= p_m.sum()   (p_m of the LAST tau_m, not of tau_lo). */
void oracle_entmax_bisect(const float* X, int64_t rows, int d, float alpha, int n_iter,
                          int ensure_sum_one, float* P) {
    const float am1 = alpha - 1.0f;          /* fp32 tensor arithmetic, entmax.py:31-36,42 */
    const float inv_am1 = 1.0f / am1;        /* entmax.py:22 */
    const float gp_one = powf(1.0f, am1);    /* entmax.py:46: _gp(1, alpha) == 1 */
    const float gp_invd = powf((float)(1.0 / (double)d), am1); /* entmax.py:47: (1/d) ** (alpha-1) */
#pragma omp parallel
    {
        float* xs = (float*)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
        for (int64_t r = 0; r < rows; ++r) {
            const float* x = X + r * d;
            float* p = P + r * d;
            float mx = -INFINITY;
            int has_nan = 0;
            for (int i = 0; i < d; ++i) {
                xs[i] = x[i] * am1;
                if (xs[i] != xs[i]) has_nan = 1;
                if (xs[i] > mx) mx = xs[i];
            }
            if (has_nan) mx = NAN; /* torch.max propagates NaN */
            float tau_lo = mx - gp_one;
            const float tau_hi = mx - gp_invd;
            float f_lo = 0.f;
            for (int i = 0; i < d; ++i) f_lo += entmax_p(xs[i] - tau_lo, inv_am1);
            f_lo -= 1.0f;
            float dm = tau_hi - tau_lo;
            for (int i = 0; i < d; ++i) p[i] = 0.f;
            for (int it = 0; it < n_iter; ++it) {
                dm *= 0.5f;
                const float tau_m = tau_lo + dm;
                float s = 0.f;
                for (int i = 0; i < d; ++i) {
                    p[i] = entmax_p(xs[i] - tau_m, inv_am1);
                    s += p[i];
                }
                const float f_m = s - 1.0f;
                if (f_m * f_lo >= 0.f) tau_lo = tau_m;
            }
            if (ensure_sum_one) {
                float s = 0.f;
                for (int i = 0; i < d; ++i) s += p[i];
                for (int i = 0; i < d; ++i) p[i] = p[i] / s;
            }
        }
        free(xs);
    }
}

/* utils/entmax.py:29-68 with a TENSOR alpha (entmax.py:31-36: alpha broadcast over every dimension but `dim`, i.e. one alpha per
 * row; round 6): the same statements with per-row  alpha - 1 (:42),  1 / (alpha - 1) (:22),  (1 / d) ** (alpha - 1) (:47). */
void oracle_entmax_bisect_rows(const float* X, const float* alpha_rows, int64_t rows, int d, int n_iter, int ensure_sum_one,
                               float* P) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float am1 = alpha_rows[r] - 1.0f;
        const float inv_am1 = 1.0f / am1;
        const float gp_invd = powf((float)(1.0 / (double)d), am1);
        const float* x = X + r * d;
        float* p = P + r * d;
        float mx = -INFINITY;
        int has_nan = 0;
        for (int i = 0; i < d; ++i) {
            const float v = x[i] * am1;
            if (v != v) has_nan = 1;
            if (v > mx) mx = v;
        }
        if (has_nan) mx = NAN;
        float tau_lo = mx - 1.0f;
        const float tau_hi = mx - gp_invd;
        float f_lo = 0.f;
        for (int i = 0; i < d; ++i) f_lo += entmax_p(x[i] * am1 - tau_lo, inv_am1);
        f_lo -= 1.0f;
        float dm = tau_hi - tau_lo;
        for (int i = 0; i < d; ++i) p[i] = 0.f;
        for (int it = 0; it < n_iter; ++it) {
            dm *= 0.5f;
            const float tau_m = tau_lo + dm;
            float s = 0.f;
            for (int i = 0; i < d; ++i) {
                p[i] = entmax_p(x[i] * am1 - tau_m, inv_am1);
                s += p[i];
            }
            const float f_m = s - 1.0f;
            if (f_m * f_lo >= 0.f) tau_lo = tau_m;
        }
        if (ensure_sum_one) {
            float s = 0.f;
            for (int i = 0; i < d; ++i) s += p[i];
            for (int i = 0; i < d; ++i) p[i] = p[i] / s;
        }
    }
}

/* nn.Softmax(dim=-1) — the alpha == 1 branch (models/armnet_1h.py:12). */
void oracle_softmax(const float* X, int64_t rows, int d, float* P) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const float* x = X + r * d;
        float* p = P + r * d;
        float mx = -INFINITY;
        for (int i = 0; i < d; ++i)
            if (x[i] > mx) mx = x[i];
        float s = 0.f;
        for (int i = 0; i < d; ++i) {
            p[i] = expf(x[i] - mx);
            s += p[i];
        }
        for (int i = 0; i < d; ++i) p[i] = p[i] / s;
    }
}

/* models/armnet_1h.py:34 + :85-86 (armnet.py:36 + :86-87), O = nhid (1h) or nhead*nhid (mh):
 * w[b,o,f] = p[b,o,f] * values[o,f];  neurons[b,o,e] = exp(sum_f w[b,o,f] * x[b,f,e]). */
void oracle_interact_exp(const float* x, const float* p, const float* values, int64_t B, int F, int E,
                         int O, float* arm_weight, float* neurons) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* xb = x + b * F * E;
        for (int o = 0; o < O; ++o) {
            const float* pr = p + (b * O + o) * F;
            float* wr = arm_weight + (b * O + o) * F;
            for (int f = 0; f < F; ++f) wr[f] = pr[f] * values[o * F + f];
            for (int e = 0; e < E; ++e) {
                float acc = 0.f;
                for (int f = 0; f < F; ++f) acc = fmaf(wr[f], xb[f * E + e], acc);
                neurons[(b * O + o) * E + e] = expf(acc);
            }
        }
    }
}

/* nn.BatchNorm1d in eval mode on [B,C,L] (L=1 for the MLP's [B,C]):
 * y = x * (w / sqrt(var+eps)) + (b - mean * w / sqrt(var+eps))  — ATen's CPU transform form. */
void oracle_bn_eval(const float* x, const float* w, const float* b, const float* mean, const float* var,
                    float eps, int64_t B, int C, int L, float* y) {
    for (int c = 0; c < C; ++c) {
        const float invstd = 1.0f / sqrtf(var[c] + eps);
        const float a = w[c] * invstd;
        const float s = b[c] - mean[c] * a;
        for (int64_t n = 0; n < B; ++n)
            for (int l = 0; l < L; ++l) {
                const int64_t i = (n * C + c) * L + l;
                y[i] = x[i] * a + s;
            }
    }
}

/* nn.BatchNorm1d in training mode: batch statistics over (B,L) per channel (biased variance for the
 * normalisation, unbiased for the running update, momentum 0.1).  Statistics accumulate in double. */
void oracle_bn_train(const float* x, const float* w, const float* b, float* run_mean, float* run_var,
                     float eps, float momentum, int64_t B, int C, int L, float* y) {
    const double n = (double)B * L;
    for (int c = 0; c < C; ++c) {
        double s = 0.0, ss = 0.0;
        for (int64_t i = 0; i < B; ++i)
            for (int l = 0; l < L; ++l) s += x[(i * C + c) * L + l];
        const double mean = s / n;
        for (int64_t i = 0; i < B; ++i)
            for (int l = 0; l < L; ++l) {
                double dlt = x[(i * C + c) * L + l] - mean;
                ss += dlt * dlt;
            }
        const double var = ss / n;
        const float invstd = (float)(1.0 / sqrt(var + eps));
        for (int64_t i = 0; i < B; ++i)
            for (int l = 0; l < L; ++l) {
                const int64_t k = (i * C + c) * L + l;
                y[k] = (x[k] - (float)mean) * invstd * w[c] + b[c];
            }
        if (run_mean) {
            run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
            run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(ss / (n - 1.0));
        }
    }
}

/* nn.Linear: y[b,o] = sum_i x[b,i] W[o,i] + bias[o]   (models/layers.py:75,81) */
void oracle_linear(const float* x, const float* W, const float* bias, int64_t B, int I, int O, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b)
        for (int o = 0; o < O; ++o) {
            float acc = 0.f;
            for (int i = 0; i < I; ++i) acc = fmaf(x[b * I + i], W[(int64_t)o * I + i], acc);
            y[b * O + o] = acc + (bias ? bias[o] : 0.f);
        }
}

void oracle_relu(float* x, int64_t n) {
    for (int64_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f;
}

/* The fused block a2..a9 of SURVEY.md §8(a) in one call, eval mode, batch-parallel with OpenMP.
 * variant 0 = one-head (models/armnet_1h.py:76-87), 1 = multi-head (models/armnet.py:77-90).
 * vals is clamped IN PLACE (the reference's side effect).  out: [B, K*H, E] post-BN.
 * Used by tests as the end-to-end checker and by bench.py as the CPU baseline ("port"). */
int oracle_arm_block(int variant, int64_t B, int F, int E, int D, int K, int H, float alpha, int n_iter,
                     const int64_t* ids, float* vals, const float* table, int64_t nfeat,
                     const float* bilinear_w, const float* query, const float* values,
                     const float* bn_w, const float* bn_b, const float* bn_mean, const float* bn_var,
                     float bn_eps, float* out) {
    if (B < 0 || F <= 0 || E <= 0 || D <= 0 || K <= 0 || H <= 0) return ORACLE_ERR_BAD_ARG;
    const int O = K * H;
    for (int64_t i = 0; i < B * F; ++i)
        if (ids[i] < 0 || ids[i] >= nfeat) return ORACLE_ERR_ID_RANGE;
    oracle_clamp_vals(vals, B * F);
    const int64_t chunk = 256;
    const int64_t nchunk = (B + chunk - 1) / chunk;
    int err = 0;
#pragma omp parallel
    {
        float* x = (float*)malloc(sizeof(float) * (size_t)chunk * F * E);
        float* g = (float*)malloc(sizeof(float) * (size_t)chunk * O * F);
        float* p = (float*)malloc(sizeof(float) * (size_t)chunk * O * F);
        float* w = (float*)malloc(sizeof(float) * (size_t)chunk * O * F);
        float* z = (float*)malloc(sizeof(float) * (size_t)chunk * O * E);
#pragma omp for schedule(dynamic, 1)
        for (int64_t c = 0; c < nchunk; ++c) {
            const int64_t b0 = c * chunk;
            const int64_t nb = (B - b0) < chunk ? (B - b0) : chunk;
            if (oracle_embed(ids + b0 * F, vals + b0 * F, table, nfeat, nb, F, E, x) != 0) err = 1;
            if (variant == 0) oracle_gates_1h(x, bilinear_w, query, nb, F, E, D, H, g);
            else oracle_gates_mh(x, bilinear_w, query, nb, F, E, D, K, H, g);
            /* nested omp-for inside these helpers binds to a team of one: serial per chunk */
            if (alpha == 1.0f) oracle_softmax(g, nb * O, F, p);
            else oracle_entmax_bisect(g, nb * O, F, alpha, n_iter, 1, p);
            oracle_interact_exp(x, p, values, nb, F, E, O, w, z);
            oracle_bn_eval(z, bn_w, bn_b, bn_mean, bn_var, bn_eps, nb, O, E, out + b0 * O * E);
        }
        free(x); free(g); free(p); free(w); free(z);
    }
    return err ? ORACLE_ERR_ID_RANGE : ORACLE_OK;
}

/* ---- sibling models: GC-ARM (models/gc_arm.py) and AFN (models/afn.py) ---------------------- */

/* torch.exp / torch.log, element-wise (gc_arm.py:89, afn.py:63) */
void oracle_exp(const float* x, int64_t n, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = expf(x[i]);
}
void oracle_log(const float* x, int64_t n, float* y) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) y[i] = logf(x[i]);
}

/* gc_arm.py:30-41 — bilinear = einsum('bfx,kxy,koy->bkof', keys, bilinear, Q)   (no scale)
 *                   context = sum_f x[b,f,:];  gc = einsum('bx,kxy,koy->bko', context, bilinear, Q)
 *                   attn_gates = bilinear + gc.unsqueeze(-1)
 * both einsums contracted left to right like oracle_gates_mh. */
void oracle_gates_gc(const float* x, const float* bil, const float* Q, int64_t B, int F, int E, int K, int H,
                     float* gates) {
    float* t = (float*)malloc(sizeof(float) * (size_t)(F + 1) * K * E);
    float* ctx = (float*)malloc(sizeof(float) * (size_t)E);
    for (int64_t b = 0; b < B; ++b) {
        const float* xb = x + b * F * E;
        for (int e = 0; e < E; ++e) {
            float acc = 0.f;
            for (int f = 0; f < F; ++f) acc += xb[f * E + e];
            ctx[e] = acc;
        }
        for (int f = 0; f <= F; ++f) {                    /* row F: the context vector */
            const float* row = f < F ? xb + f * E : ctx;
            for (int k = 0; k < K; ++k)
                for (int y = 0; y < E; ++y) {
                    float acc = 0.f;
                    for (int e = 0; e < E; ++e) acc = fmaf(row[e], bil[(k * E + e) * E + y], acc);
                    t[(f * K + k) * E + y] = acc;
                }
        }
        float* gb = gates + b * K * H * F;
        for (int k = 0; k < K; ++k)
            for (int o = 0; o < H; ++o) {
                float gc = 0.f;
                for (int y = 0; y < E; ++y) gc = fmaf(t[(F * K + k) * E + y], Q[(k * H + o) * E + y], gc);
                for (int f = 0; f < F; ++f) {
                    float acc = 0.f;
                    for (int y = 0; y < E; ++y) acc = fmaf(t[(f * K + k) * E + y], Q[(k * H + o) * E + y], acc);
                    gb[(k * H + o) * F + f] = acc + gc;
                }
            }
    }
    free(t);
    free(ctx);
}

/* gc_arm.py:46 + :92 — w = p * values;  arm[b,o,e] = sum_f x_exp[b,f,e] * w[b,o,f]   (no exp) */
void oracle_interact_sum(const float* xexp, const float* p, const float* values, int64_t B, int F, int E, int O,
                         float* arm_weight, float* arm) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* xb = xexp + b * F * E;
        for (int o = 0; o < O; ++o) {
            const float* pr = p + (b * O + o) * F;
            float* wr = arm_weight + (b * O + o) * F;
            for (int f = 0; f < F; ++f) wr[f] = pr[f] * values[o * F + f];
            for (int e = 0; e < E; ++e) {
                float acc = 0.f;
                for (int f = 0; f < F; ++f) acc = fmaf(xb[f * E + e], wr[f], acc);
                arm[(b * O + o) * E + e] = acc;
            }
        }
    }
}

/* afn.py:64-66 — afn[b,o,e] = exp(sum_f xlog[b,f,e] * W[o,f] + bias[o])   (Linear over the field dim of the
 * transposed tensor, then transposed back) */
void oracle_afn_linear_exp(const float* xlog, const float* W, const float* bias, int64_t B, int F, int E, int O,
                           float* out) {
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
        const float* xb = xlog + b * F * E;
        for (int o = 0; o < O; ++o)
            for (int e = 0; e < E; ++e) {
                float acc = 0.f;
                for (int f = 0; f < F; ++f) acc = fmaf(xb[f * E + e], W[o * F + f], acc);
                out[(b * O + o) * E + e] = expf(acc + bias[o]);
            }
    }
}
