"""The module surface on HOST tensors: the reference's ATen op chain, for a model that was never moved to the GPU.

The reference's model runs wherever its tensors are (`models/model_utils.py:86` calls `.cuda()` only if a device is
available; `train.py:103-106`).  This file is the branch `ARMNetModel.arm_block`, `models.layers.Embedding` and
`utils.entmax.entmax_bisect` take when BOTH the parameters and the batch are CPU tensors: plain torch ops in the
reference's order — in-place clamp, embedding x value, key projection, gates,
the 50-step bisection of `utils/entmax.py:29-68` with tensor-tensor `pow`, value weighting, einsum + exp — differentiable
by autograd (the sparse map through `block._EntmaxFn`, whose backward is `entmax.py:70-80`).

It is a DEVICE DISPATCH, not a fallback: device tensors never come here (a missing / unloadable HIP library, or a model
and a batch on different devices, still raise), nothing here touches `oracle/`, and `bench.py` times it as the
`cpu_baseline.aten_chain` leg next to the GPU number."""
import torch
import torch.nn.functional as Fn


def on_host(*tensors):
    """True when every tensor lives in host memory (the reference's CPU path); mixed placements are the caller's error"""
    return all(not t.is_cuda for t in tensors)


def note_host_branch(owner):
    """A model that was never moved to the GPU runs the reference's ATen chain on the host — the reference's own device
    semantics (model_utils.py:86), but orders of magnitude slower than the HIP path.  On a box that HAS a GPU that is most
    likely a forgotten `.cuda()`: say so once per model (round-5 advisor finding; `model.allow_host = True` or
    ARMNET_ALLOW_HOST=1 silences it).  On a box without a GPU there is nothing to warn about."""
    if getattr(owner, "_host_noted", False) or getattr(owner, "allow_host", False):
        return
    owner._host_noted = True
    import os
    if torch.cuda.is_available() and not os.environ.get("ARMNET_ALLOW_HOST"):
        import warnings
        warnings.warn(f"{type(owner).__name__}: model and batch are in host memory — running the reference's ATen op chain on "
                      "the CPU, not the HIP kernels (this box has a GPU: call model.cuda() and move the batch, or set "
                      "model.allow_host = True to silence this)", RuntimeWarning, stacklevel=3)


def clamp_vals_(vals):
    """armnet_1h.py:81 / armnet.py:82: x['value'].clamp_(0.001, 1.) in place"""
    return vals.clamp_(0.001, 1.0)


def embedding(ids, vals, table):
    """layers.py:15-21: table[ids] * vals.unsqueeze(2); out-of-range ids raise torch's own IndexError"""
    x = Fn.embedding(ids, table)
    return x if vals is None else x * vals.unsqueeze(-1)


def entmax_bisect(X, alpha, dim=-1, n_iter=50, ensure_sum_one=True):
    """utils/entmax.py:29-68 statement for statement on ATen ops (softmax when alpha == 1, as armnet_1h.py:12 builds it):
    alpha as a broadcast TENSOR (the reference's `pow` is tensor-tensor), tau_lo = max - 1, tau_hi = max - (1/d)^(alpha-1),
    n_iter halvings keeping the side on which f has the sign of f_lo, p of the LAST tau_m, renormalised."""
    moved = dim % X.dim() != X.dim() - 1
    if torch.is_tensor(alpha) and alpha.numel() != 1:
        # entmax.py:31-36: a tensor alpha, already expanded to X's shape with extent 1 along `dim`
        Xm = X.movedim(dim, -1) if moved else X
        al = (alpha.movedim(dim, -1) if moved else alpha).to(Xm.dtype)
    else:
        if float(alpha) == 1.0:
            return torch.softmax(X, dim=dim)
        Xm = X.movedim(dim, -1) if moved else X
        al = torch.full((1,) * Xm.dim(), float(alpha), dtype=Xm.dtype, device=Xm.device).expand(*Xm.shape[:-1], 1)
    d = Xm.shape[-1]
    am1 = al - 1
    inv = 1 / am1
    Xs = Xm * am1
    mx = Xs.max(dim=-1, keepdim=True).values
    tau_lo = mx - 1.0
    tau_hi = mx - (1.0 / d) ** am1
    f_lo = torch.clamp(Xs - tau_lo, min=0).pow(inv).sum(-1, keepdim=True) - 1
    dm = tau_hi - tau_lo
    p = None
    for _ in range(n_iter):
        dm = dm / 2
        tau_m = tau_lo + dm
        p = torch.clamp(Xs - tau_m, min=0).pow(inv)
        f_m = p.sum(-1, keepdim=True) - 1
        tau_lo = torch.where((f_m * f_lo) >= 0, tau_m, tau_lo)
    if p is None:                                               # n_iter == 0: the reference fails on the unbound p_m too
        raise ValueError("entmax_bisect needs n_iter >= 1")
    if ensure_sum_one:
        p = p / p.sum(-1, keepdim=True)
    return p.movedim(-1, dim) if dim % X.dim() != X.dim() - 1 else p


def arm_block(variant_one_head, ids, vals, table, bilinear_w, query, values, alpha, n_iter=50):
    """rows a2..a8 of SURVEY.md §8a on host tensors -> the pre-BatchNorm exponential neurons [B, K*H, E].
    `vals` is clamped in place.  one head: bilinear_w [D,E] (nn.Linear weight), query [H,D], values [H,F];
    multi-head: bilinear_w [K,E,D], query [K,H,D], values [K,H,F]."""
    from .block import entmax_forward
    clamp_vals_(vals)                                                           # armnet_1h.py:81 / armnet.py:82
    x = embedding(ids, vals, table)                                             # layers.py:20-21
    scale = query.shape[-1] ** -0.5
    if variant_one_head:                                                        # armnet_1h.py:30-34, 85-86
        keys = Fn.linear(x, bilinear_w)
        gates = torch.einsum("bfe,oe->bof", keys, query) * scale
        p = entmax_forward(gates, alpha, dim=-1, n_iter=n_iter)
        w = torch.einsum("bof,of->bof", p, values)
        return torch.exp(torch.einsum("bfe,bof->boe", x, w))
    gates = torch.einsum("bfx,kxy,koy->bkof", x, bilinear_w, query) * scale    # armnet.py:33-36, 86-88
    p = entmax_forward(gates, alpha, dim=-1, n_iter=n_iter)
    w = torch.einsum("bkof,kof->bkof", p, values)
    z = torch.exp(torch.einsum("bfe,bkof->bkoe", x, w))
    return z.reshape(z.shape[0], -1, z.shape[-1])                               # 'b k o e -> b (k o) e'
