/*
 * armnet_hip.h — C ABI of the MI355X-native ARM-Net forward hot path.
 *
 * The reference (nusdbsystem/ARM-Net) has no FFI layer: its hot path is a chain
 * of ATen calls made from two nn.Modules.  The entry points below are what a
 * binding of that path replaces; each cites the reference lines it stands for.
 * INTEGRATION.md shows the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no torch / HIP types in signatures; `stream` is a hipStream_t
 *     passed as void* (NULL = the null stream).
 *   - every pointer is a DEVICE pointer unless stated otherwise; the caller
 *     owns all memory; nothing is retained after return.
 *   - calls enqueue work on `stream` and return without synchronising.
 *   - return value: 0 = ARMNET_OK, negative = armnet_status; never throws, never exits.
 *   - re-entrant; no global mutable state (the last HIP error string is thread-local).
 *   - all floating point is IEEE fp32; tensors are dense row-major.
 *
 * Shapes: B batch, F nfield, E nemb, O = nhead * nhid exponential neurons
 * (one-head model: nhead = 1), D = d_k.
 */
#ifndef ARMNET_HIP_H
#define ARMNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* History (every version only ADDS entry points; nothing that was exported ever changed its signature):
 *   1  round 1: fold, fused forward / backward, lookup, entmax, BatchNorm passes, exact-protocol routing
 *   2  round 2: prediction head, GC-ARM / AFN forward entry points, fixed-capacity and whole-shard routing helpers
 *   3  round 3: siblings' training helpers (entmax backward, linear_small), wide heads
 *   4  round 4: armnet_shard_route_fixed(_perm, _epoch), armnet_shard_gather_perm_f32, GC-ARM / AFN fused backward,
 *               armnet_gather_map_stats_f32, armnet_bn_bwd_scatter_f32
 *   5  round 5: hot-row replication of the row-sharded lookup (armnet_shard_route_fixed_hot, _perm_hot,
 *               armnet_shard_gather_perm_hot_f32); armnet_linear_bf16x3_f32 (the training head's GEMMs) */
#define ARMNET_ABI_VERSION 7

typedef enum armnet_status {
    ARMNET_OK = 0,
    ARMNET_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, inconsistent arguments */
    ARMNET_ERR_UNSUPPORTED = -2,  /* shape outside what the kernels cover (see armnet_strerror) */
    ARMNET_ERR_ID_RANGE = -3,     /* reserved for host-side checked wrappers (IndexError) */
    ARMNET_ERR_HIP = -4           /* a HIP runtime call failed: see armnet_last_hip_error() */
} armnet_status;

/* ids may be int64 (torch.LongTensor, data_loader.py:20) or int32 */
typedef enum armnet_id_type { ARMNET_ID_I64 = 0, ARMNET_ID_I32 = 1 } armnet_id_type;

/* model variant for parameter folding */
typedef enum armnet_variant {
    ARMNET_ONE_HEAD = 0,  /* models/armnet_1h.py: bilinear_w is nn.Linear weight [D,E], query [H,D] */
    ARMNET_MULTI_HEAD = 1,/* models/armnet.py:    bilinear_w [K,E,D], query [K,H,D] */
    ARMNET_GC_ARM = 2     /* models/gc_arm.py:    bilinear [K,E,E], Q [K,H,E]; no d_k^-0.5 scale (gc_arm.py:33-34) */
} armnet_variant;

/* flags for the fused forward */
#define ARMNET_F_WRITE_CLAMPED_VALS 0x1u /* reproduce x['value'].clamp_() on the caller's buffer */
#define ARMNET_F_FAITHFUL_BISECT    0x2u /* the reference's n_iter-step bisection, statement for statement (entmax.py:44-64:
                                            * bracket, halvings, the sign test on f_m * f_lo, p of the LAST tau_m, renormalise)
                                            * instead of the Newton / Michelot solve of the same root.  In every kernel the
                                            * power t^(1/(alpha-1)) is exp2(r * log2 t) on the hardware transcendental units
                                            * (about 2 ulp; the reference's ATen pow is libm-class): a sign decision that falls
                                            * within those ulps of zero can differ — the iteration is the reference's, the
                                            * arithmetic is fp32 on this device.  Held to the reference's vectors at <= 2e-6
                                            * (alpha <= 2) / 2e-5 (alpha > 2) by tests/test_hip_parity.py */
#define ARMNET_F_FORCE_GENERIC      0x4u /* bypass the MFMA-specialised kernel (testing) */
#define ARMNET_F_NO_LIN_FINISH      0x8u /* 1 < alpha < 2 on the matrix-core kernel: every Newton step is confirmed by an
                                            evaluation (no first-order finish of a tiny last step; rounds 1-4 behaviour) —
                                            a run-time switch for bisecting a regression without a rebuild */
#define ARMNET_F_FP32_CONTRACTIONS  0x10u /* wide ARM blocks (nemb <= 16; 33+ fields from 33 neurons per launch, 29-32 fields from
                                            64, 17-28 from 256) run both contractions as fp16 x 2 operand splits on the 16-bit matrix pipe
                                            (three products, fp32 accumulate, exact power-of-two scales per sample / per
                                            parameter slice: 22 significant bits per operand, results within 6e-7 of the
                                            fp32 form); this switch keeps them on the fp32 MFMAs (A/B and parity tests) */

int armnet_abi_version(void);
const char* armnet_strerror(int status);
const char* armnet_last_hip_error(void);

/*
 * Parameter-only precompute (re-run when weights change).  Replaces nothing the
 * reference executes per batch; it folds, exactly in real arithmetic,
 *   - the key projection into the query (models/armnet_1h.py:30-32, models/armnet.py:33-34):
 *       q_fold[k*H+o, e] = D^-0.5 * sum_y W[k][e,y] * query[k][o,y]
 *   - eval-mode BatchNorm1d into an affine (models/armnet_1h.py:65,85, models/armnet.py:67,89):
 *       bn_scale[c] = weight[c] / sqrt(running_var[c] + eps);  bn_shift[c] = bias[c] - running_mean[c] * bn_scale[c]
 * Outputs: q_fold [K*H, E], bn_scale [K*H], bn_shift [K*H].
 */
int armnet_fold_params_f32(int variant, int K, int H, int E, int D,
                           const float* bilinear_w, const float* query,
                           const float* bn_weight, const float* bn_bias,
                           const float* bn_running_mean, const float* bn_running_var, float bn_eps,
                           float* q_fold, float* bn_scale, float* bn_shift, void* stream);

/*
 * The fused block, rows a2..a9 of SURVEY.md §8(a), eval mode:
 *   vals <- clamp(vals, 1e-3, 1)                         armnet_1h.py:81   / armnet.py:82
 *   x[b,f,:] = table[ids[b,f],:] * vals[b,f]             layers.py:20-21
 *   g[b,o,f] = sum_e x[b,f,e] * q_fold[o,e]              armnet_1h.py:30-32 / armnet.py:33-34 (folded)
 *   p[b,o,:] = entmax_alpha(g[b,o,:])  (softmax if alpha == 1)   entmax.py:29-68, armnet_1h.py:12,33
 *   w[b,o,f] = p[b,o,f] * values[o,f]                    armnet_1h.py:34   / armnet.py:36
 *   z[b,o,:] = exp(sum_f w[b,o,f] * x[b,f,:])            armnet_1h.py:85-86 / armnet.py:86-87
 *   out[b,o,:] = z[b,o,:] * bn_scale[o] + bn_shift[o]    armnet_1h.py:85   / armnet.py:88-89
 * out is [B, O, E]; viewed as [B, O*E] it is the MLP head's input (armnet_1h.py:87-89).
 *
 * ids: [B,F] of `id_type`; vals: [B,F] (read; written only with ARMNET_F_WRITE_CLAMPED_VALS);
 * table: [nfeat,E]; values: [O,F].  n_iter is the reference's bisection count (default 50): with
 * n_iter >= 24 and alpha <= 2 a converged Newton/Michelot solve of the same root is used unless
 * ARMNET_F_FAITHFUL_BISECT is set.
 * id_status: optional int32 the DEVICE can write — device memory, or (round 6) pinned host memory mapped into the
 * device: 1 is stored (relaxed, system scope; never cleared by the library) when any id is outside [0,nfeat); such ids
 * read row 0 instead of faulting.  The host wrapper turns it into IndexError — by default at its NEXT call or at its
 * poll(), not with a synchronisation behind every launch: the reference's nn.Embedding on a GPU (layers.py:20) reports a
 * bad id by an asynchronous device-side assert, and train.py:117-121 never synchronises for it.  The same holds for every
 * id_status argument below.
 */
int armnet_fused_fwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                         const void* ids, int id_type, float* vals,
                         const float* table, int64_t nfeat,
                         const float* q_fold, const float* values,
                         const float* bn_scale, const float* bn_shift,
                         float* out, int32_t* id_status, void* stream);

/*
 * Which kernel the fused calls above run for a shape: 1 = the matrix-core (MFMA) kernel, 0 = the generic
 * thread-per-row kernel (nemb < 4 or > 64, nfield > 48, > 1024 neurons, ARMNET_F_FORCE_GENERIC), negative =
 * armnet_status.  Host-only, no GPU work.  (The reference's literal bisection — alpha > 2, n_iter < 24,
 * ARMNET_F_FAITHFUL_BISECT — is a solver mode of the matrix-core kernel.)
 * Buffers need only their natural alignment (4 bytes for floats / int32, 8 for int64 ids).
 */
int armnet_fused_kernel_kind(int F, int E, int O, float alpha, int n_iter, uint32_t flags);

/*
 * Same block fed with pre-gathered, UNSCALED rows [B,F,E] (the receiver side of the row-sharded
 * lookup: rows arrive by all-to-all, DESIGN.md "multi-GPU").  vals as above.
 */
int armnet_fused_fwd_from_rows_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                                   const float* rows, float* vals,
                                   const float* q_fold, const float* values,
                                   const float* bn_scale, const float* bn_shift,
                                   float* out, void* stream);

/*
 * models/layers.py:15-21 alone — Embedding.forward: out[b,f,:] = table[ids[b,f],:] * vals[b,f].
 * Used for the ensemble branch's second table (armnet_1h.py:91, armnet.py:94).  vals may be NULL
 * (plain gather, the owner side of the sharded lookup).  No clamp is applied here.
 */
int armnet_gather_scale_f32(int64_t n_rows, int E, const void* ids, int id_type, const float* vals,
                            const float* table, int64_t nfeat, float* out, int32_t* id_status,
                            void* stream);

/*
 * Backward of armnet_gather_scale_f32 with respect to the table (what autograd derives for nn.Embedding followed by
 * the multiply, layers.py:20-21): d_table[ids[r], :] += grad[r, :] * vals[r]  (float atomics; vals may be NULL).
 * Used by the ensemble branch's second table in training.
 */
int armnet_scatter_add_f32(int64_t n_rows, int E, const void* ids, int id_type, const float* vals,
                           const float* grad, int64_t nfeat, float* d_table, void* stream);

/* armnet_1h.py:81 alone: vals <- clamp(vals, 1e-3, 1) in place. */
int armnet_clamp_vals_f32(float* vals, int64_t n, void* stream);

/*
 * utils/entmax.py:134 entmax_bisect(X, alpha, dim=-1, n_iter, ensure_sum_one) over the last
 * dimension of a [rows, d] matrix (alpha == 1 -> softmax).  Same solver selection as the fused call.
 */
int armnet_entmax_f32(int64_t rows, int d, float alpha, int n_iter, int ensure_sum_one, uint32_t flags,
                      const float* X, float* P, void* stream);

/*
 * The same with one alpha PER ROW (round 6) — utils/entmax.py:31-36: `alpha` may be a tensor that broadcasts over every
 * dimension of X but `dim`; the host wrapper expands it to alpha_rows [rows] (all > 1, as the reference requires).  Every row
 * runs the reference's n_iter-step bisection (entmax.py:44-64) with its own alpha - 1, 1 / (alpha - 1), (1 / d)^(alpha - 1).
 * The gradients (entmax.py:70-98, with respect to X and to alpha) are tensor operations on P in the host wrapper (armnet_hip/block.py).
 */
int armnet_entmax_rows_f32(int64_t rows, int d, const float* alpha_rows, int n_iter, int ensure_sum_one, const float* X,
                           float* P, void* stream);

/*
 * Backward of the sparse map w.r.t. its input (utils/entmax.py:70-80; softmax's Jacobian for alpha == 1), rows over the
 * last dim: Y = the forward's output [rows, d], dY its gradient, dX [rows, d] (may alias dY).  The gradient w.r.t. alpha
 * (entmax.py:81-98) is not part of this entry point: with a float alpha there is none to give; a tensor alpha goes through
 * armnet_entmax_rows_f32 and the host wrapper's tensor operations (round 6).
 */
int armnet_entmax_bwd_f32(int64_t rows, int d, float alpha, const float* Y, const float* dY, float* dX, void* stream);

/*
 * Backward of the fused block (training).  `z` is the forward's output computed with an
 * identity BatchNorm affine (bn_scale = 1, bn_shift = 0: the pre-BN neurons of armnet_1h.py:85-86; the
 * training-mode BatchNorm1d and the MLP stay with torch autograd), `dz` its gradient.  vals must already be
 * clamped (the forward did it).  Accumulates (+=, caller zero-initialises):
 *   d_table [nfeat,E]  dense gradient of the embedding table (x = table[id] * val, layers.py:20-21),
 *   d_values [O,F]     gradient of attn_layer.values (armnet_1h.py:34),
 *   d_qfold [O,E]      gradient of the folded query; the caller applies the chain rule of the fold to
 *                      get d(query) and d(bilinear_w).
 * The entmax Jacobian-vector product follows utils/entmax.py:70-80 (softmax when alpha == 1).
 */
int armnet_fused_bwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                         const void* ids, int id_type, const float* vals, const float* table, int64_t nfeat,
                         const float* q_fold, const float* values, const float* z, const float* dz,
                         float* d_table, float* d_values, float* d_qfold, void* stream);

/*
 * The same backward with the training-mode BatchNorm1d that follows the block (armnet_1h.py:85 / armnet.py:88-89)
 * folded in: `dy` is the gradient of the BatchNorm OUTPUT and the kernel forms
 *     dz = coefA[o] * dy + coefC[o] * z + coefB[o]
 * on the fly (coefficients from armnet_bn_bwd_coef_f32), so the gradient of the pre-BN neurons is never
 * written to memory.
 */
int armnet_fused_bwd_bn_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                            const void* ids, int id_type, const float* vals, const float* table, int64_t nfeat,
                            const float* q_fold, const float* values, const float* z, const float* dy,
                            const float* coefA, const float* coefB, const float* coefC,
                            float* d_table, float* d_values, float* d_qfold, void* stream);

/*
 * Training-mode torch.nn.BatchNorm1d (armnet_1h.py:65,85, armnet.py:67,88-89 on [B, nhid, nemb]; layers.py:77
 * on [B, nhid]) as HBM-bound passes over x viewed as [N, C, L] (L = 1 for 2-D input).  Forward:
 *   armnet_bn_stats_f32     stats[c] += sum (x - k_c), stats[C + c] += sum (x - k_c)^2 with k_c = x[0, c, 0]
 *                           (caller zero-initialises stats[2C]; shifted sums: no cancellation when |mean| >> std)
 *   armnet_bn_finalize_f32  per channel: mean, rstd = 1/sqrt(biased var + eps), scale = weight * rstd,
 *                           shift = bias - mean * scale; running_mean / running_var (unbiased) updated in place
 *                           with `momentum` when non-NULL (count = N * L; weight / bias may be NULL)
 *   armnet_bn_apply_f32     y = x * scale[c] + shift[c], optionally followed by ReLU (layers.py:78)
 * Backward (dy = gradient of y; with relu_scale/relu_shift = the forward's scale/shift the ReLU mask
 * x * scale + shift > 0 is recomputed and applied to dy):
 *   armnet_bn_bwd_reduce_f32  sums[c] += sum dy, sums[C + c] += sum dy * xhat   (caller zero-initialises)
 *   armnet_bn_bwd_coef_f32    d_bias = sums[:C], d_weight = sums[C:], and the coefficients of
 *                             dx = coefA * dy + coefC * x + coefB
 *   armnet_bn_bwd_apply_f32   dx elementwise (for the block's BatchNorm use armnet_fused_bwd_bn_f32 instead)
 */
/*
 * armnet_gather_map_stats_f32 — the head of the sibling models' training forward in one pass (round 4):
 *   out[b,f,:] = map(table[ids[b,f], :] * vals[b,f]),  map = exp (0, gc_arm.py:87-89) or log (1, afn.py:61-63; table clipped);
 *   stats[f] += sum (out - k_f), stats[F + f] += sum (out - k_f)^2 with k_f = out[0, f, 0]  — exactly what
 *   armnet_bn_stats_f32 would add for out viewed as [B, F, E] (caller zero-initialises stats[2F], then armnet_bn_finalize_f32).
 *   vals must already be clamped (armnet_clamp_vals_f32); *id_status |= 1 for an id outside [0, nfeat) (it reads row 0).
 */
int armnet_gather_map_stats_f32(int64_t B, int F, int E, const void* ids, int id_type, const float* vals,
                                const float* table, int64_t nfeat, int map, float* out, float* stats,
                                int32_t* id_status, void* stream);
/*
 * armnet_bn_bwd_scatter_f32 — the tail of the sibling models' training backward in one pass (round 4): rows are the
 * (sample, field) pairs of the lookup, channel = row mod C; t = exp(x) (map 0, gc_arm.py:89) or log(x) (map 1, afn.py:63);
 *   d_table[ids[r], e] += (coefA[f] * dy[r,e] + coefC[f] * t[r,e] + coefB[f]) * (map == 0 ? t : exp(-t)) * vals[r]
 * = armnet_bn_bwd_apply_f32, the derivative of exp / log, and armnet_scatter_add_f32 without the two temporaries.
 */
int armnet_bn_bwd_scatter_f32(int64_t n_rows, int C, int E, const void* ids, int id_type, const float* vals,
                              const float* t, const float* dy, const float* coefA, const float* coefB,
                              const float* coefC, int map, int64_t nfeat, float* d_table, void* stream);
int armnet_bn_stats_f32(int64_t N, int C, int L, const float* x, float* stats, void* stream);
int armnet_bn_finalize_f32(int C, int64_t count, const float* stats, const float* x, int L,
                           const float* weight, const float* bias, float eps, float momentum,
                           float* running_mean, float* running_var,
                           float* mean, float* rstd, float* scale, float* shift, void* stream);
int armnet_bn_apply_f32(int64_t N, int C, int L, const float* x, const float* scale, const float* shift,
                        int relu, float* y, void* stream);
int armnet_bn_bwd_reduce_f32(int64_t N, int C, int L, const float* x, const float* dy, const float* mean,
                             const float* rstd, const float* relu_scale, const float* relu_shift,
                             float* sums, void* stream);
int armnet_bn_bwd_coef_f32(int C, int64_t count, const float* sums, const float* weight, const float* mean,
                           const float* rstd, float* d_weight, float* d_bias,
                           float* coefA, float* coefB, float* coefC, void* stream);
int armnet_bn_bwd_apply_f32(int64_t N, int C, int L, const float* x, const float* dy, const float* coefA,
                            const float* coefB, const float* coefC, const float* relu_scale,
                            const float* relu_shift, float* dx, void* stream);

/*
 * Routing step of the row-sharded embedding lookup (multi-GPU; no reference counterpart — the
 * reference is single-device).  Rank r of R owns table rows {i : i % R == r},
 * stored at local index i / R.  For n ids: counts[r] = ids owned by r; send_local[p] = local row
 * index of the id at send position p (positions grouped by owner, deterministic); perm[i] = send
 * position of id i.  The rows come back from the all-to-all in send order, so `perm` is exactly the
 * int32 "ids" array with which armnet_fused_fwd_f32 indexes the received row buffer as its table.
 * workspace: armnet_shard_route_ws_bytes(n, R) bytes of device memory (contents undefined).
 */
int64_t armnet_shard_route_ws_bytes(int64_t n, int R);
int armnet_shard_route_ids(int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                           int32_t* counts, int32_t* send_local, int32_t* perm,
                           void* workspace, int64_t ws_bytes, int32_t* id_status, void* stream);

/*
 * The same routing with per-rank de-duplication: every distinct id of the batch is sent (and its row
 * received) once.  send_local then has n_unique = sum(counts) entries (capacity n), perm[i] is the slot of
 * id i's row among them, *n_unique (optional device int32) receives the total.  Direct-address marking +
 * one exclusive scan over R * ceil(nfeat / R) flags: pays when the batch is not much smaller than the table.
 */
int64_t armnet_shard_route_unique_ws_bytes(int R, int64_t nfeat);
int armnet_shard_route_unique_ids(int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                                  int32_t* counts, int32_t* send_local, int32_t* perm, int32_t* n_unique,
                                  void* workspace, int64_t ws_bytes, int32_t* id_status, void* stream);

/*
 * Fixed-capacity layout of a routed lookup: R equal slots of `cap` indices instead of back-to-back owner segments, so
 * that both exchanges of the row-sharded lookup are EQUAL-SPLIT all-to-alls and no count has to reach the host
 * (csrc/shard_pad.hip).  counts / send_local / perm as produced by armnet_shard_route_ids or _unique_ids;
 * send_pad [R*cap], perm_pad [n]; *overflow |= 1 when an owner's count exceeds cap (the caller repeats that step with
 * the exact, host-synchronised protocol).  Unused slot entries hold local index 0.
 */
int armnet_shard_pad_route(int64_t n, int R, int64_t cap, const int32_t* counts, const int32_t* send_local,
                           const int32_t* perm, int32_t* send_pad, int32_t* perm_pad, int32_t* overflow, void* stream);

/*
 * Routing of the fixed-capacity protocol in ONE call (csrc/shard_route_fixed.hip; round 4): what armnet_shard_route_ids /
 * _unique_ids followed by armnet_shard_pad_route produce, without the intermediate back-to-back layout — a lookup only
 * needs a unique position inside its owner's slot.  send_pad [R*cap]: local row indices (id / R) requested from owner
 * o = id % R in entries [o*cap, o*cap + min(counts[o], cap)), index 0 in the rest; perm_pad [n]: o*cap + s of lookup i;
 * counts [R]; *overflow |= 1 when counts[o] > cap (the surplus lookups point at entry o*cap: the caller repeats the step
 * with the exact protocol).  dedup != 0: every DISTINCT id is requested once (byte map over (owner, local) + chunk scan;
 * requests sorted by local index inside a slot; workspace armnet_shard_route_fixed_ws_bytes(R, nfeat, 1) bytes).
 * dedup == 0: every lookup is a request; slot positions are reserved with atomics, so send_pad / perm_pad may differ
 * from run to run — send_pad[perm_pad[i]] == id_i / R always holds; no workspace.  *id_status |= 1 for an id outside
 * [0, nfeat) (it reads row 0).  No reference counterpart (the reference is single-device).
 * With dedup != 0, perm_pad may be NULL: the position gather perm_pad[i] = pos[(id % R, id / R)] is then left to
 * armnet_shard_route_fixed_perm on the same workspace — the request list does not depend on it, so a caller can run it on a
 * side stream beside the index exchange and the owner-side gather (the workspace must stay untouched until it is done).
 */
/*
 * armnet_shard_gather_perm_f32 — the owner-side gather of a request list, out[j, :] = table[idx[j], :] (int32 local row
 * indices; an index outside [0, table_rows) reads row 0), AND the position gather of armnet_shard_route_fixed_perm as one
 * launch: both depend only on armnet_shard_route_fixed(dedup != 0, perm_pad = NULL) — on one rank idx is its send_pad, on
 * several the received request list —, both are needed only by the consumer of the rows, and they are bound by different
 * parts of the memory system, so their blocks overlap instead of queueing.
 */
int armnet_shard_gather_perm_f32(int64_t n_rows, int E, const int32_t* idx, const float* table, int64_t table_rows,
                                 float* out, int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                                 int32_t* perm_pad, const void* workspace, int64_t ws_bytes, void* stream);
int64_t armnet_shard_route_fixed_ws_bytes(int R, int64_t nfeat, int dedup);
int armnet_shard_route_fixed_perm(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm_pad,
                                  const void* workspace, int64_t ws_bytes, void* stream);
/* armnet_shard_route_fixed with a caller-managed MARK EPOCH for dedup != 0: the byte map holds the epoch of the last step that
 * marked a position, a step's marks are the bytes equal to its epoch.  epoch 0 = armnet_shard_route_fixed (the call zeroes the
 * map, marks are 1); epoch e in 2..255 = no fill — valid when the previous call on this workspace (same R, nfeat) used epoch 0
 * or a smaller e.  A caller cycling 0, 2, 3, .., 255, 0, .. zeroes the map once per 255 steps.  Not for a step captured in a
 * hipGraph (the replay would repeat the epoch). */
int armnet_shard_route_fixed_epoch(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap, int dedup,
                                   int32_t* send_pad, int32_t* perm_pad, int32_t* counts, int32_t* overflow,
                                   int32_t* id_status, void* workspace, int64_t ws_bytes, int epoch, void* stream);
int armnet_shard_route_fixed(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap, int dedup,
                             int32_t* send_pad, int32_t* perm_pad, int32_t* counts, int32_t* overflow,
                             int32_t* id_status, void* workspace, int64_t ws_bytes, void* stream);
/*
 * Hot-row replication (round 5; SURVEY.md §8e names it as the third lever for the 8-GPU target, the reference is
 * single-device: no counterpart).  Real click logs are skewed: the head of a frequency-ordered id space carries most
 * lookups.  Every rank keeps rows [0, hot_rows) replicated; the routing then treats an id < hot_rows as already
 * answered: perm_pad[i] = hot_base + id, no slot entry, no mark, no count, nothing on the links.  The caller places
 * its hot rows at row hot_base of the buffer the fused block reads (armnet_hip/sharded.py: right behind the R * cap
 * received rows, hot_base = R * cap).  hot_rows = 0 is armnet_shard_route_fixed_epoch.  Everything else — arguments,
 * overflow / id_status flags, epochs, NULL perm_pad with dedup — as armnet_shard_route_fixed_epoch; the position gather
 * left pending by perm_pad = NULL must be given the same (hot_rows, hot_base).
 */
int armnet_shard_route_fixed_hot(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int64_t cap, int dedup,
                                 int32_t* send_pad, int32_t* perm_pad, int32_t* counts, int32_t* overflow,
                                 int32_t* id_status, void* workspace, int64_t ws_bytes, int epoch, int64_t hot_rows,
                                 int64_t hot_base, void* stream);
int armnet_shard_route_fixed_perm_hot(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm_pad,
                                      const void* workspace, int64_t ws_bytes, int64_t hot_rows, int64_t hot_base,
                                      void* stream);
int armnet_shard_gather_perm_hot_f32(int64_t n_rows, int E, const int32_t* idx, const float* table, int64_t table_rows,
                                     float* out, int64_t n, const void* ids, int id_type, int R, int64_t nfeat,
                                     int32_t* perm_pad, const void* workspace, int64_t ws_bytes, int64_t hot_rows,
                                     int64_t hot_base, void* stream);

/*
 * Whole-shard exchange of the row-sharded lookup (csrc/shard_pad.hip): when a batch asks for most of every shard the
 * owners all-gather their shards (L = ceil(nfeat / R) rows each) instead of answering request lists; perm[i] is then
 * the direct address (id % R) * L + id / R of id i's row in the gathered [R * L, E] buffer.  ids int64 / int32;
 * *id_status |= 1 for an id outside [0, nfeat) (such an id reads row 0).
 */
int armnet_shard_direct_perm(int64_t n, const void* ids, int id_type, int R, int64_t nfeat, int32_t* perm,
                             int32_t* id_status, void* stream);

/* Which kernel the sibling models' fused forward gets for a block shape (round 6): 1 = the matrix-core kernel (nemb 4..128 —
 * 65..128 since round 6 —, nfield <= 48, <= 1024 neurons), 0 = the shape-agnostic thread-per-row kernel.  afn: 0 = GC-ARM
 * (models/gc_arm.py), 1 = AFN (models/afn.py).  (armnet_fused_kernel_kind answers the same for ARM-Net itself.) */
int armnet_sibling_kernel_kind(int afn, int F, int E, int O);

/*
 * Sibling models on the same kernels, eval mode.
 *
 * armnet_gc_fused_fwd_f32 — models/gc_arm.py:82-95 (GC_ARMModel.forward up to arm_bn):
 *   vals <- clamp(vals, 1e-3, 1);  x = table[ids] * vals                              gc_arm.py:86-87
 *   g[b,o,f] = sum_e x[b,f,e] q_fold[o,e]     (fold with ARMNET_GC_ARM: no scale)     gc_arm.py:33-34
 *   g[b,o,:] += sum_f g[b,o,f]                (global context: the gates of sum_f x)  gc_arm.py:37-41
 *   w = entmax_alpha(g) * values                                                      gc_arm.py:43-46
 *   arm[b,o,e] = sum_f w[b,o,f] * (exp(x[b,f,e]) * emb_scale[f] + emb_shift[f])       gc_arm.py:89,92  (no outer exp)
 *   out = arm * bn_scale[o] + bn_shift[o]                                             gc_arm.py:94
 * armnet_afn_fused_fwd_f32 — models/afn.py:56-69 (AFNModel.forward up to afn_bn; the table must already be clipped,
 * armnet_abs_clamp_min_f32):
 *   x as above;  l = log(x) * emb_scale[f] + emb_shift[f]                             afn.py:63
 *   out[b,o,e] = exp(sum_f weight[o,f] * l[b,f,e] + bias[o]) * bn_scale[o] + bn_shift[o]    afn.py:64-66
 * armnet_fold_bn_f32 — eval-mode BatchNorm1d as an affine: scale = w / sqrt(var + eps), shift = b - mean * scale.
 * armnet_abs_clamp_min_f32 — afn.py:74-77 embedding_clip, in place: p <- max(|p|, lo).
 * Both fused entry points write vals back (ARMNET_F_WRITE_CLAMPED_VALS) and flag out-of-range ids like
 * armnet_fused_fwd_f32.
 */
int armnet_gc_fused_fwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags,
                            const void* ids, int id_type, float* vals, const float* table, int64_t nfeat,
                            const float* q_fold, const float* values, const float* emb_scale, const float* emb_shift,
                            const float* bn_scale, const float* bn_shift, float* out, int32_t* id_status, void* stream);
int armnet_afn_fused_fwd_f32(int64_t B, int F, int E, int O, uint32_t flags, const void* ids, int id_type, float* vals,
                             const float* table, int64_t nfeat, const float* weight, const float* bias,
                             const float* emb_scale, const float* emb_shift, const float* bn_scale,
                             const float* bn_shift, float* out, int32_t* id_status, void* stream);
int armnet_fold_bn_f32(int C, const float* weight, const float* bias, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, void* stream);
int armnet_abs_clamp_min_f32(float* p, int64_t n, float lo, void* stream);

/*
 * GC-ARM's block backward on the matrix cores (round 4) — the training step of `train.py:108-114 --model gc_arm`
 * (models/gc_arm.py:82-95 under autograd), one kernel per slice of 64 (nemb <= 16) / 32 (nemb <= 32) / 16 (nemb <= 128) neurons:
 *   inputs as armnet_gc_fused_fwd_f32; emb_scale / emb_shift = the affine emb_bn applies to exp(x) in THIS step
 *   (training mode: from the batch statistics, armnet_bn_finalize_f32);  z = the pre-arm_bn block output of the forward
 *   run with an identity bn_scale / bn_shift;  dy = the gradient of z, or — with coefA/B/C (armnet_bn_bwd_coef_f32, all
 *   three or none) — the gradient of arm_bn's output, dz = coefA[o] * dy + coefC[o] * z + coefB[o] formed in the kernel.
 * Outputs:
 *   d_table  += the part of the table gradient that flows through the gates (x = table[id] * val -> g = x . q_fold)
 *   d_values += , d_qfold +=     as armnet_fused_bwd_f32 (caller zero-initialises all three)
 *   d_y [B,F,E] = the gradient of y = emb_bn(exp(x)) (overwritten).  emb_bn normalises with batch statistics, so
 *                 its backward needs sums over the whole batch: the caller runs armnet_bn_bwd_reduce/coef/apply_f32 on
 *                 (exp(x), d_y), multiplies by exp(x) and adds the result to d_table with armnet_scatter_add_f32.
 * The global context (gc_arm.py:37-41) adds the same number to every gate of a row; the sparse map is invariant to
 * that, its Jacobian's rows sum to zero, and the context's gradient is analytically zero (the reference's autograd
 * produces rounding noise there) — it is not formed.
 * armnet_gc_fused_bwd_supported: 1 when the shape has a kernel (nemb 4..64 with nfield <= 48; since round 6 nemb 65..128 with
 * nfield <= 32, the range of armnet_fused_bwd_f32's own matrix-core kernel — the same for armnet_afn_fused_bwd_supported); otherwise
 * ARMNET_ERR_UNSUPPORTED and the caller keeps its composed device ops.
 */
/*
 * AFN's block backward (afn.py:56-69 under train.py:108-114), same kernel family:
 *   l = emb_scale[f] * log(x) + emb_shift[f]  (this step's emb_bn affine of log(x));  z = exp(weight . l + bias) = the pre-afn_bn
 *   output of armnet_afn_fused_fwd_f32 run with an identity bn_scale / bn_shift;  dy / coefA,B,C as above.
 *   d_weight [O,F] += , d_bias [O] +=  (caller zero-initialises);  d_y [B,F,E] = the gradient of l (overwritten): the caller
 *   runs emb_bn's backward passes on (log(x), d_y), divides by x and adds to the table gradient with armnet_scatter_add_f32.
 */
int armnet_afn_fused_bwd_supported(int F, int E, int O);
int armnet_afn_fused_bwd_f32(int64_t B, int F, int E, int O, uint32_t flags, const void* ids, int id_type,
                             const float* vals, const float* table, int64_t nfeat, const float* weight,
                             const float* emb_scale, const float* emb_shift, const float* z, const float* dy,
                             const float* coefA, const float* coefB, const float* coefC, float* d_weight,
                             float* d_bias, float* d_y, void* stream);
int armnet_gc_fused_bwd_supported(int F, int E, int O);
int armnet_gc_fused_bwd_f32(int64_t B, int F, int E, int O, float alpha, int n_iter, uint32_t flags, const void* ids,
                            int id_type, const float* vals, const float* table, int64_t nfeat, const float* q_fold,
                            const float* values, const float* emb_scale, const float* emb_shift, const float* z,
                            const float* dy, const float* coefA, const float* coefB, const float* coefC,
                            float* d_table, float* d_values, float* d_qfold, float* d_y, void* stream);

/*
 * The eval-mode prediction head — models/layers.py:68-88 `MLP`: n x (Linear, BatchNorm1d, ReLU, Dropout) then
 * Linear(., 1) — as built by models/armnet_1h.py:67 / models/armnet.py:69 (and the ensemble's deep_mlp,
 * armnet.py:96-97), on the bf16 matrix cores with fp32-equivalent numerics: activations and BatchNorm-folded
 * weights are split into three bf16 slices each and the six significant cross products are accumulated in fp32
 * (csrc/mlp_head.hip).  One launch fuses up to two hidden layers and, optionally, the final Linear; deeper heads
 * chain launches through the hidden-activation output.  Eval mode only (running statistics, Dropout = identity).
 *
 *   armnet_mlp_head_supported   1 when (K0 = input width, nhid = hidden width <= 256, n_hidden = 1|2) has a kernel
 *   armnet_mlp_packed_bytes     size of the packed-parameter blob of one launch (device memory, caller-allocated)
 *   armnet_mlp_pack_layer_f32   parameter-only precompute (re-run when weights change), one call per layer:
 *       slot 0: first hidden layer  W [nhid, K0],  slot 1: second hidden layer W [nhid, nhid],
 *       slot 2: final Linear W [1, nhid], b [1].   bn_* = the BatchNorm1d behind the Linear (NULL: none):
 *       W' = W * s, b' = b * s + (beta - mean * s), s = gamma / sqrt(var + eps); W' is then split into fp16 hi/lo of
 *       the row-scaled weight AND bf16 hi/mid/lo (round-to-nearest) in the kernel's operand order.
 *   armnet_mlp_head_f32         x [B, K0] (row stride ldx floats) -> has_final ? out [B] (the logits, layers.py:88)
 *                                                                              : out [B, nhid] (post-ReLU activations,
 *                                                                                row stride ldo floats)
 *       has_final = 2: out [B] += this launch's share of the final Linear (no overwrite).  Hidden layers wider than
 *       256 (run.sh:18-19,44-45 ask for 500) run as SLICES of <= 256 units, one hidden layer per launch: a slice is a
 *       layer of its own whose W / b / BatchNorm rows are the slice's (pointer offsets), whose activations land in
 *       columns n0.. of a [B, ldo] buffer (out + n0), and whose share of the final Linear is written by the first
 *       slice (has_final = 1, with the bias) and added by the others (has_final = 2, packed with b = NULL).
 *       x is read in whole 16-float k-steps: ldx >= 16 * ceil(K0 / 16), and the columns K0 .. of every row must be
 *       readable and finite (they meet zero weights).  A contiguous [B, K0] tensor qualifies when K0 % 16 == 0.
 */
/*
 * The head shapes without a matrix-core launch (csrc/linear_small.hip, round 4): a Linear with a handful of outputs,
 * out[b, n] (+)= (bias[n] + sum_k x[b, k] W[n, k]) * scale, N <= 16 — the whole MLP when nlayers == 0
 * (models/layers.py:79-80) and the final Linear(nhid, noutput) of a head with noutput > 1 (models/layers.py:86-87).
 * x [B, K] row stride ldx, W [N, K] dense, bias [N] or NULL, out [B, N] row stride ldo; accumulate != 0 adds to out.
 */
int armnet_linear_small_f32(int64_t B, int K, int N, const float* x, int64_t ldx, const float* W, const float* bias,
                            float scale, float* out, int64_t ldo, int accumulate, void* stream);

/*
 * A plain Linear on the head's matrix-core path (round 5): out[b, n] = bias[n] + sum_k x[b, k] W[n, k] for N <= 256 outputs,
 * bf16x3-split operands, six cross products, fp32 accumulate — the GEMMs of the TRAINING head (models/layers.py:68-88 under
 * train.py:108-114: nn.Linear forward, and its input gradient dX = dY W as a Linear with the transposed weight), where
 * BatchNorm1d needs the batch's pre-activation values, so nothing is folded and no ReLU is applied.
 * packed = armnet_mlp_pack_layer_f32(K, N, 1, 0, W [N, K], K, bias | NULL, NULL, NULL, NULL, NULL, 0, packed) of
 * armnet_mlp_packed_bytes(K, N, 1) bytes — a per-step precompute here, the weights change with every optimizer step.
 * x [B, K] row stride ldx >= 16 * ceil(K / 16) (columns past K readable and finite), out [B, N] row stride ldo >= N.
 */
int armnet_linear_bf16x3_f32(int64_t B, int K, int N, const float* x, int64_t ldx, const void* packed, float* out,
                             int64_t ldo, void* stream);

int armnet_mlp_head_supported(int K0, int nhid, int n_hidden);
int64_t armnet_mlp_packed_bytes(int K0, int nhid, int n_hidden);
int armnet_mlp_pack_layer_f32(int K0, int nhid, int n_hidden, int slot, const float* W, int Kin, const float* b,
                              const float* bn_weight, const float* bn_bias, const float* bn_running_mean,
                              const float* bn_running_var, float bn_eps, void* packed, void* stream);
int armnet_mlp_head_f32(int64_t B, int K0, int nhid, int n_hidden, int has_final, const float* x, int64_t ldx,
                        const void* packed, float* out, int64_t ldo, void* stream);
/*
 * Round 6: the head's operand split.  Default (armnet_mlp_head_f32 = flags 0): fp16 x 2 — activations and row-scaled
 * weights as hi + lo fp16 parts (round to nearest), THREE cross products on v_mfma_f32_32x32x16_f16, fp32 accumulate,
 * exact power-of-two rescale — with an in-kernel range vote: a block whose first-layer inputs leave the fp16 range
 * (|x| > 4 062, inf; or a wave of it nothing but |x| < 1e-3) redoes its samples with the bf16 x 3 split (six products) inside the same launch.
 * ARMNET_MLP_F_BF16X3 forces the bf16 x 3 split for every block (the rounds 2-5 kernel; A/B and bisecting).
 * Either way the result matches models/layers.py:68-88 evaluated in fp32 to fp32-GEMM class error.
 */
#define ARMNET_MLP_F_BF16X3 0x1u
int armnet_mlp_head_ex_f32(int64_t B, int K0, int nhid, int n_hidden, int has_final, const float* x, int64_t ldx,
                           const void* packed, float* out, int64_t ldo, uint32_t flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ARMNET_HIP_H */
