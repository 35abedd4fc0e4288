"""ctypes front-end of oracle/armnet_oracle.c (TEST INFRASTRUCTURE ONLY).

The C file restates the reference's ATen op chain stage by stage (citations
there).  This module only marshals numpy arrays and strings the stages
together the way ARMNetModel.forward does (models/armnet_1h.py:76-98,
models/armnet.py:77-101), driven by a reference-format ``state_dict``.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libarmnet_oracle.so")
_SRC = os.path.join(_HERE, "armnet_oracle.c")
_lib = None

c_f = ctypes.POINTER(ctypes.c_float)
c_i64 = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [_SRC, os.path.join(_HERE, "armnet_cpu_twins.c")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(p) for p in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libarmnet_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_arm_block.restype = ctypes.c_int
        _lib.oracle_embed.restype = ctypes.c_int
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(c_f)


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_f)


def set_threads(n):
    lib().armnet_oracle_set_threads(int(n))


def max_threads():
    return int(lib().armnet_oracle_max_threads())


def effective_cpus():
    """CPUs this process may actually use: min(affinity, cgroup v2 quota, OpenMP default)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, max_threads()))


def clamp_vals(vals):
    assert vals.dtype == np.float32 and vals.flags["C_CONTIGUOUS"]
    lib().oracle_clamp_vals(_fp(vals), ctypes.c_int64(vals.size))
    return vals


def embed(ids, vals, table):
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    B, F = ids.shape
    nfeat, E = table.shape
    vals, pv = _f(vals)
    table, pt = _f(table)
    x = np.empty((B, F, E), np.float32)
    rc = lib().oracle_embed(ids.ctypes.data_as(c_i64), pv, pt, ctypes.c_int64(nfeat),
                            ctypes.c_int64(B), F, E, _fp(x))
    if rc != 0:
        raise IndexError("index out of range in self")
    return x


def gates_1h(x, W, q):
    B, F, E = x.shape
    D = W.shape[0]
    H = q.shape[0]
    x, px = _f(x); W, pW = _f(W); q, pq = _f(q)
    g = np.empty((B, H, F), np.float32)
    lib().oracle_gates_1h(px, pW, pq, ctypes.c_int64(B), F, E, D, H, _fp(g))
    return g


def gates_mh(x, bw, q):
    B, F, E = x.shape
    K, _, D = bw.shape
    H = q.shape[1]
    x, px = _f(x); bw, pb = _f(bw); q, pq = _f(q)
    g = np.empty((B, K, H, F), np.float32)
    lib().oracle_gates_mh(px, pb, pq, ctypes.c_int64(B), F, E, D, K, H, _fp(g))
    return g


def entmax_bisect(X, alpha=1.5, n_iter=50, ensure_sum_one=True):
    """utils/entmax.py:134 entmax_bisect(X, alpha, dim=-1, n_iter, ensure_sum_one)."""
    X, pX = _f(X)
    d = X.shape[-1]
    rows = X.size // d
    P = np.empty_like(X)
    lib().oracle_entmax_bisect(pX, ctypes.c_int64(rows), d, ctypes.c_float(alpha), int(n_iter),
                               int(bool(ensure_sum_one)), _fp(P))
    return P


def entmax_bisect_rows(X, alpha_rows, n_iter=50, ensure_sum_one=True):
    """utils/entmax.py:31-36,134 with one alpha per row of the last dimension: X [..., d], alpha_rows broadcastable to X[..., :1]"""
    X, pX = _f(X)
    d = X.shape[-1]
    rows = X.size // d
    A = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_rows, np.float32), X.shape[:-1] + (1,)).reshape(rows))
    P = np.empty_like(X)
    lib().oracle_entmax_bisect_rows(pX, _fp(A), ctypes.c_int64(rows), d, int(n_iter), int(bool(ensure_sum_one)), _fp(P))
    return P


def softmax(X):
    X, pX = _f(X)
    d = X.shape[-1]
    P = np.empty_like(X)
    lib().oracle_softmax(pX, ctypes.c_int64(X.size // d), d, _fp(P))
    return P


def sparse_map(X, alpha, n_iter=50):
    return softmax(X) if float(alpha) == 1.0 else entmax_bisect(X, alpha, n_iter)


def interact_exp(x, p, values):
    """p: [B,O,F], values: [O,F] -> (arm_weight [B,O,F], neurons [B,O,E])."""
    B, F, E = x.shape
    O = values.shape[0]
    x, px = _f(x); p, pp = _f(p); values, pv = _f(values)
    w = np.empty((B, O, F), np.float32)
    z = np.empty((B, O, E), np.float32)
    lib().oracle_interact_exp(px, pp, pv, ctypes.c_int64(B), F, E, O, _fp(w), _fp(z))
    return w, z


def bn_eval(x, w, b, mean, var, eps=1e-5):
    x, px = _f(x)
    B, C = x.shape[0], x.shape[1]
    L = x.size // (B * C) if x.size else 1
    w, pw = _f(w); b, pb = _f(b); mean, pm = _f(mean); var, pvv = _f(var)
    y = np.empty_like(x)
    lib().oracle_bn_eval(px, pw, pb, pm, pvv, ctypes.c_float(eps), ctypes.c_int64(B), C, L, _fp(y))
    return y


def bn_train(x, w, b, run_mean, run_var, eps=1e-5, momentum=0.1):
    """Returns (y, new_running_mean, new_running_var)."""
    x, px = _f(x)
    B, C = x.shape[0], x.shape[1]
    L = x.size // (B * C)
    w, pw = _f(w); b, pb = _f(b)
    rm = np.array(run_mean, dtype=np.float32, copy=True)
    rv = np.array(run_var, dtype=np.float32, copy=True)
    y = np.empty_like(x)
    lib().oracle_bn_train(px, pw, pb, _fp(rm), _fp(rv), ctypes.c_float(eps), ctypes.c_float(momentum),
                          ctypes.c_int64(B), C, L, _fp(y))
    return y, rm, rv


def linear(x, W, b=None):
    x, px = _f(x); W, pW = _f(W)
    B, I = x.shape
    O = W.shape[0]
    y = np.empty((B, O), np.float32)
    if b is not None:
        b, pb = _f(b)
    else:
        pb = None
    lib().oracle_linear(px, pW, pb, ctypes.c_int64(B), I, O, _fp(y))
    return y


def mlp(x, sd, prefix, train=False):
    """models/layers.py:68-88 — nn.Sequential of (Linear, BN1d, ReLU, Dropout)*n + Linear.
    Layer indices in the state_dict: Linear at 4i, BatchNorm1d at 4i+1; last Linear at 4n."""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
    lin = [i for i in idx if (prefix + f"{i}.running_mean") not in sd]
    h = x
    for i in lin[:-1]:
        h = linear(h, sd[prefix + f"{i}.weight"], sd[prefix + f"{i}.bias"])
        j = i + 1
        if train:
            h, _, _ = bn_train(h, sd[prefix + f"{j}.weight"], sd[prefix + f"{j}.bias"],
                               sd[prefix + f"{j}.running_mean"], sd[prefix + f"{j}.running_var"])
        else:
            h = bn_eval(h, sd[prefix + f"{j}.weight"], sd[prefix + f"{j}.bias"],
                        sd[prefix + f"{j}.running_mean"], sd[prefix + f"{j}.running_var"])
        h = np.maximum(h, 0.0)          # ReLU; Dropout is identity in eval (and p=0 in the fixtures)
    i = lin[-1]
    return linear(h, sd[prefix + f"{i}.weight"], sd[prefix + f"{i}.bias"])


def forward(variant, ctor, sd, ids, vals, train=False, n_iter=50):
    """Whole ARMNetModel.forward from a reference-format state_dict.

    variant '1h' -> models/armnet_1h.py:76-98, 'mh' -> models/armnet.py:77-101.
    Returns every intermediate the golden fixtures hold.  ``vals`` is copied; the clamped copy is
    returned as ``vals_clamped`` (the reference clamps the caller's tensor in place)."""
    out = {}
    vals = np.array(vals, dtype=np.float32, copy=True)
    clamp_vals(vals)
    out["vals_clamped"] = vals
    x = embed(ids, vals, sd["embedding.embedding.weight"])
    out["x_emb"] = x
    alpha = float(ctor["alpha"])
    if variant == "1h":
        g = gates_1h(x, sd["attn_layer.bilinear_w.weight"], sd["attn_layer.query"])
        values = sd["attn_layer.values"]
    else:
        g = gates_mh(x, sd["attn_layer.bilinear_w"], sd["attn_layer.query"])
        values = sd["attn_layer.values"]
    out["gates"] = g
    p = sparse_map(g, alpha, n_iter)
    out["p"] = p
    B = x.shape[0]
    F = x.shape[1]
    O = int(np.prod(values.shape[:-1]))
    w, z = interact_exp(x, p.reshape(B, O, F), values.reshape(O, F))
    out["arm_weight"] = w.reshape(g.shape)
    out["neurons"] = z if variant == "1h" else z.reshape(B, O, -1)
    if train:
        xa, rm, rv = bn_train(z, sd["arm_bn.weight"], sd["arm_bn.bias"], sd["arm_bn.running_mean"],
                              sd["arm_bn.running_var"])
        out["after/arm_bn.running_mean"], out["after/arm_bn.running_var"] = rm, rv
    else:
        xa = bn_eval(z, sd["arm_bn.weight"], sd["arm_bn.bias"], sd["arm_bn.running_mean"],
                     sd["arm_bn.running_var"])
    out["x_arm"] = xa
    y = mlp(xa.reshape(B, -1), sd, "mlp.mlp.", train=train)
    if "ensemble_layer.weight" in sd:
        xd = embed(ids, vals, sd["deep_embedding.embedding.weight"]).reshape(B, -1)
        yd = mlp(xd, sd, "deep_mlp.mlp.", train=train)
        y = linear(np.concatenate([y, yd], axis=1), sd["ensemble_layer.weight"], sd["ensemble_layer.bias"])
    out["logits"] = np.squeeze(y)
    return out


def _tail(block, sd, ids, vals, train=False):
    """MLP head + optional DNN ensemble of the sibling models (gc_arm.py:96-105, afn.py:68-72): y.squeeze(1)"""
    B = block.shape[0]
    y = mlp(block.reshape(B, -1), sd, "mlp.mlp.", train=train)
    if "ensemble_layer.weight" in sd:
        xd = embed(ids, vals, sd["deep_embedding.embedding.weight"]).reshape(B, -1)
        yd = mlp(xd, sd, "deep_mlp.mlp.", train=train)
        y = linear(np.concatenate([y, yd], axis=1), sd["ensemble_layer.weight"], sd["ensemble_layer.bias"])
    return y[:, 0]


def forward_gc_arm(ctor, sd, ids, vals, n_iter=50):
    """GC_ARMModel.forward in eval mode (models/gc_arm.py:82-105), stage by stage."""
    out = {}
    vals = np.array(vals, dtype=np.float32, copy=True)
    clamp_vals(vals)
    out["vals_clamped"] = vals
    x = embed(ids, vals, sd["embedding.embedding.weight"])                     # gc_arm.py:87
    B, F, E = x.shape
    xe = np.empty_like(x)
    lib().oracle_exp(_fp(x), ctypes.c_int64(x.size), _fp(xe))                   # gc_arm.py:89
    x_exp = bn_eval(xe, sd["emb_bn.weight"], sd["emb_bn.bias"], sd["emb_bn.running_mean"], sd["emb_bn.running_var"])
    Q, pQ = _f(sd["attn_layers.Q"]); bil, pbil = _f(sd["attn_layers.bilinear"])
    K, H = Q.shape[0], Q.shape[1]
    g = np.empty((B, K, H, F), np.float32)
    lib().oracle_gates_gc(_fp(x), pbil, pQ, ctypes.c_int64(B), F, E, K, H, _fp(g))   # gc_arm.py:30-41
    out["gates"] = g
    p = sparse_map(g, float(ctor["alpha"]), n_iter)                            # gc_arm.py:43
    out["p"] = p
    O = K * H
    values, pv = _f(np.asarray(sd["attn_layers.values"]).reshape(O, F))
    pp, ppp = _f(p.reshape(B, O, F)); xx, pxx = _f(x_exp)
    w = np.empty((B, O, F), np.float32); arm = np.empty((B, O, E), np.float32)
    lib().oracle_interact_sum(pxx, ppp, pv, ctypes.c_int64(B), F, E, O, _fp(w), _fp(arm))   # gc_arm.py:45-46,92
    xa = bn_eval(arm, sd["arm_bn.weight"], sd["arm_bn.bias"], sd["arm_bn.running_mean"], sd["arm_bn.running_var"])
    out["x_arm"] = xa
    out["logits"] = _tail(xa, sd, ids, vals)
    return out


def forward_afn(ctor, sd, ids, vals):
    """AFNModel.forward in eval mode (models/afn.py:49-77), stage by stage.  Returns the clipped table too."""
    out = {}
    vals = np.array(vals, dtype=np.float32, copy=True)
    clamp_vals(vals)
    out["vals_clamped"] = vals
    table = np.maximum(np.abs(np.asarray(sd["embedding.embedding.weight"], dtype=np.float32)), np.float32(1e-4))
    out["table_after"] = table                                                 # afn.py:74-77
    sd = dict(sd); sd["embedding.embedding.weight"] = table
    x = embed(ids, vals, table)                                                # afn.py:61
    B, F, E = x.shape
    xl = np.empty_like(x)
    lib().oracle_log(_fp(x), ctypes.c_int64(x.size), _fp(xl))                   # afn.py:63
    x_log = bn_eval(xl, sd["emb_bn.weight"], sd["emb_bn.bias"], sd["emb_bn.running_mean"], sd["emb_bn.running_var"])
    W, pW = _f(sd["afn.weight"]); b, pb = _f(sd["afn.bias"])
    O = W.shape[0]
    xx, pxx = _f(x_log)
    a = np.empty((B, O, E), np.float32)
    lib().oracle_afn_linear_exp(pxx, pW, pb, ctypes.c_int64(B), F, E, O, _fp(a))   # afn.py:64-65
    xa = bn_eval(a, sd["afn_bn.weight"], sd["afn_bn.bias"], sd["afn_bn.running_mean"], sd["afn_bn.running_var"])
    out["x_arm"] = xa                                                          # afn.py:66 (same slot as the ARM block)
    out["logits"] = _tail(xa, sd, ids, vals)
    return out


def arm_block(variant, ids, vals, sd, alpha, n_iter=50, threads=None):
    """Fused block a2..a9 (eval) in one OpenMP call; vals clamped IN PLACE.  Returns [B, O, E]."""
    if threads:
        set_threads(threads)
    ids = np.ascontiguousarray(ids, dtype=np.int64)
    assert vals.dtype == np.float32 and vals.flags["C_CONTIGUOUS"]
    B, F = ids.shape
    table, pt = _f(sd["embedding.embedding.weight"])
    nfeat, E = table.shape
    if variant == "1h":
        bw, pb = _f(sd["attn_layer.bilinear_w.weight"]); D = bw.shape[0]; K = 1
        q, pq = _f(sd["attn_layer.query"]); H = q.shape[0]
    else:
        bw, pb = _f(sd["attn_layer.bilinear_w"]); K, _, D = bw.shape
        q, pq = _f(sd["attn_layer.query"]); H = q.shape[1]
    v, pv = _f(sd["attn_layer.values"])
    w, pw = _f(sd["arm_bn.weight"]); b, pbb = _f(sd["arm_bn.bias"])
    m, pm = _f(sd["arm_bn.running_mean"]); var, pvar = _f(sd["arm_bn.running_var"])
    out = np.empty((B, K * H, E), np.float32)
    rc = lib().oracle_arm_block(0 if variant == "1h" else 1, ctypes.c_int64(B), F, E, D, K, H,
                                ctypes.c_float(alpha), int(n_iter), ids.ctypes.data_as(c_i64), _fp(vals),
                                pt, ctypes.c_int64(nfeat), pb, pq, pv, pw, pbb, pm, pvar,
                                ctypes.c_float(1e-5), _fp(out))
    if rc == -3:
        raise IndexError("index out of range in self")
    if rc != 0:
        raise RuntimeError(f"oracle_arm_block failed: {rc}")
    return out


# ---- `_cpu` twins of the C ABI (oracle/armnet_cpu_twins.c): same arguments as include/armnet_hip.h minus the stream ----
def twin_fold_params(variant, K, H, E, D, bilinear_w, query, bn_weight, bn_bias, bn_mean, bn_var, eps=1e-5):
    """armnet_fold_params_f32_cpu -> (q_fold [K*H,E], bn_scale [K*H], bn_shift [K*H])"""
    bw, pbw = _f(bilinear_w); q, pq = _f(query)
    w, pw = _f(bn_weight); b, pb = _f(bn_bias); m, pm = _f(bn_mean); v, pv = _f(bn_var)
    qf = np.empty((K * H, E), np.float32); sc = np.empty(K * H, np.float32); sh = np.empty(K * H, np.float32)
    rc = lib().armnet_fold_params_f32_cpu(int(variant), K, H, E, D, pbw, pq, pw, pb, pm, pv, ctypes.c_float(eps),
                                          _fp(qf), _fp(sc), _fp(sh))
    if rc:
        raise RuntimeError(f"armnet_fold_params_f32_cpu failed: {rc}")
    return qf, sc, sh


def twin_fused_fwd(ids, vals, table, q_fold, values, bn_scale, bn_shift, alpha, n_iter=50, flags=0, rows=None):
    """armnet_fused_fwd_f32_cpu / armnet_fused_fwd_from_rows_f32_cpu -> (out [B,O,E], id_status); vals clamped in
    place with flags & 1, like the HIP entry point"""
    assert vals.dtype == np.float32 and vals.flags["C_CONTIGUOUS"]
    B, F = vals.shape
    qf, pqf = _f(q_fold); O, E = qf.shape
    v2, pv2 = _f(np.asarray(values).reshape(O, F)); sc, psc = _f(bn_scale); sh, psh = _f(bn_shift)
    out = np.empty((B, O, E), np.float32)
    status = ctypes.c_int32(0)
    if rows is not None:
        r, pr = _f(rows)
        rc = lib().armnet_fused_fwd_from_rows_f32_cpu(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                                      ctypes.c_uint32(flags), pr, _fp(vals), pqf, pv2, psc, psh, _fp(out))
    else:
        ids = np.ascontiguousarray(ids)
        assert ids.dtype in (np.int64, np.int32)
        t, pt = _f(table)
        rc = lib().armnet_fused_fwd_f32_cpu(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                            ctypes.c_uint32(flags), ids.ctypes.data_as(ctypes.c_void_p),
                                            0 if ids.dtype == np.int64 else 1, _fp(vals), pt, ctypes.c_int64(t.shape[0]),
                                            pqf, pv2, psc, psh, _fp(out), ctypes.byref(status))
    if rc:
        raise RuntimeError(f"armnet_fused_fwd_f32_cpu failed: {rc}")
    return out, status.value
