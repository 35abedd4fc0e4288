#!/usr/bin/env python3
"""round 6 (CPU, no GPU): what would the block's two contractions cost in parity if their operands were fp16 x 2 splits (three products,
the dropped lo*lo term) instead of fp32?  Every eval fixture of ARM-Net: the block in float64 from the folded parameters with
 (a) exact operands, (b) operands replaced by hi + lo (fp16 pairs, power-of-two scales to ~2^10) and the lo*lo product dropped,
 (c) operands AND accumulation in fp32 (what the fp32 MFMA does, in numpy's summation order),
each against the reference's golden neurons in units of the tests' bar 1e-5 * max(1, |ref|)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import armnet_oracle as orc  # noqa: E402
from golden_util import load, model_cases  # noqa: E402
from tol_util import elem_excess  # noqa: E402


def split(x, target=10):
    """x (float64, exactly fp32 values) -> (hi, lo) as float64 holding fp16-representable numbers of x * 2^k, and 2^-k"""
    m = float(np.max(np.abs(x[np.isfinite(x)]))) if np.isfinite(x).any() else 1.0
    k = target - int(np.ceil(np.log2(m))) if m > 0 else 0
    xs = x * 2.0 ** k
    hi = xs.astype(np.float16).astype(np.float64)
    lo = (xs - hi).astype(np.float32).astype(np.float16).astype(np.float64)
    return hi, lo, 2.0 ** -k


def entmax64(g, alpha, n_iter=100):
    if alpha == 1.0:
        e = np.exp(g - g.max(-1, keepdims=True))
        return e / e.sum(-1, keepdims=True)
    am1 = alpha - 1.0
    x = g * am1
    mx = x.max(-1, keepdims=True)
    lo, hi = mx - 1.0, mx - (1.0 / g.shape[-1]) ** am1
    for _ in range(n_iter):
        mid = (lo + hi) / 2
        s = (np.clip(x - mid, 0, None) ** (1.0 / am1)).sum(-1, keepdims=True)
        lo = np.where(s >= 1, mid, lo)
        hi = np.where(s >= 1, hi, mid)
    p = np.clip(x - lo, 0, None) ** (1.0 / am1)
    return p / p.sum(-1, keepdims=True)


def block(x, qf, values, sc, sh, alpha, mode):
    """x [B,F,E], qf [O,E], values [O,F] float64"""
    if mode == "f32":
        g = np.einsum("bfe,oe->bof", x.astype(np.float32), qf.astype(np.float32)).astype(np.float64)
    elif mode == "split":
        xh, xl, sx = split(x)
        qh, ql, sq = split(qf)
        g = (np.einsum("bfe,oe->bof", xh, qh) + np.einsum("bfe,oe->bof", xh, ql) + np.einsum("bfe,oe->bof", xl, qh)) * (sx * sq)
    else:
        g = np.einsum("bfe,oe->bof", x, qf)
    p = entmax64(g, alpha)
    w = p * values[None]
    if mode == "f32":
        z = np.einsum("bof,bfe->boe", w.astype(np.float32), x.astype(np.float32)).astype(np.float64)
    elif mode == "split":
        wh, wl, sw = split(w)
        z = (np.einsum("bof,bfe->boe", wh, xh) + np.einsum("bof,bfe->boe", wh, xl) + np.einsum("bof,bfe->boe", wl, xh)) * (sw * sx)
    else:
        z = np.einsum("bof,bfe->boe", w, x)
    return np.exp(z) * sc[None, :, None] + sh[None, :, None]


worst = {}
for name in [n for n in model_cases() if "train" not in n]:
    meta, sd, ids, vals, ref = load(name)
    c = meta["ctor"]
    mh = meta["variant"] == "mh"
    K = c["nhead"] if mh else 1
    H, E, D = c["nhid"], c["nemb"], c["d_k"]
    bw = sd["attn_layer.bilinear_w"] if mh else sd["attn_layer.bilinear_w.weight"]
    qf, sc, sh = orc.twin_fold_params(1 if mh else 0, K, H, E, D, bw, sd["attn_layer.query"], sd["arm_bn.weight"], sd["arm_bn.bias"],
                                      sd["arm_bn.running_mean"], sd["arm_bn.running_var"])
    v = np.clip(vals, 1e-3, 1.0).astype(np.float32)
    x = (sd["embedding.embedding.weight"][ids] * v[..., None]).astype(np.float32).astype(np.float64)   # layers.py:21 in fp32
    values = np.asarray(sd["attn_layer.values"], np.float64).reshape(K * H, -1)
    want = ref["x_arm"].reshape(ids.shape[0], K * H, E)
    a = float(c["alpha"])
    row = []
    for mode in ("exact", "split", "f32"):
        got = block(x, qf.astype(np.float64), values, sc.astype(np.float64), sh.astype(np.float64), a, mode)
        row.append(elem_excess(got, want, 1e-5))
    w = worst.setdefault(a, [0.0, 0.0, 0.0, ""])
    if row[1] > w[1]:
        w[3] = name
    for i in range(3):
        w[i] = max(w[i], row[i])
    if row[1] > 0.5 or "wide" in name or "stress" in name:
        print(f"{name:52s} alpha {a}: exact {row[0]:.3f}  fp16x2 split {row[1]:.3f}  fp32 {row[2]:.3f}   (x the 1e-5 bar)", flush=True)
print()
for a in sorted(worst):
    w = worst[a]
    print(f"alpha {a}: worst element, in units of the bar: exact operands {w[0]:.3f} | fp16x2 split {w[1]:.3f} ({w[3]}) | fp32 operands and sums {w[2]:.3f}")
