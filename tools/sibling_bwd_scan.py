#!/usr/bin/env python3
"""Shape scan of the siblings' matrix-core backward kernels (developer tool, GPU box): armnet_gc_fused_bwd_f32 and
armnet_afn_fused_bwd_f32 against the same backward written out in float64 torch ops from the math of
include/armnet_hip.h (forward probabilities by a float64 bisection), over nfield x nemb x neurons x alpha.
    python tools/sibling_bwd_scan.py [--quick | --wide]      # --wide: the nemb 65..128 family (round 6, nfield <= 32) only"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip import native  # noqa: E402

DEV = "cuda:0"
D = torch.float64


def entmax64(g, alpha, n_iter=60):
    """p = entmax_alpha(g) over the last dim in float64 (utils/entmax.py:29-68 run to convergence; alpha = 1: softmax)"""
    if alpha == 1.0:
        return torch.softmax(g, -1)
    am1 = alpha - 1.0
    x = g * am1
    mx = x.max(-1, keepdim=True).values
    lo, hi = mx - 1.0, mx - (1.0 / g.shape[-1]) ** am1
    for _ in range(n_iter):
        mid = (lo + hi) / 2
        s = torch.clamp(x - mid, min=0).pow(1.0 / am1).sum(-1, keepdim=True)
        lo = torch.where(s >= 1, mid, lo)
        hi = torch.where(s >= 1, hi, mid)
    p = torch.clamp(x - lo, min=0).pow(1.0 / am1)
    entmax64.margin = float((x - lo).abs().min())       # distance of the closest gate to its row's threshold (diagnostic)
    return p / p.sum(-1, keepdim=True)


entmax64.margin = float("inf")


def jvp_T(p, dp, alpha):
    """dg = J^T dp of the sparse map at p (utils/entmax.py:70-80; softmax for alpha = 1)"""
    if alpha == 1.0:
        return p * (dp - (p * dp).sum(-1, keepdim=True))
    gppr = torch.where(p > 0, p.clamp(min=1e-300).pow(2.0 - alpha), torch.zeros_like(p))
    dxp = dp * gppr
    return dxp - dxp.sum(-1, keepdim=True) / gppr.sum(-1, keepdim=True) * gppr


def case(kind, F, E, O, alpha, B, g):
    nfeat = 61
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
    if kind == "gc":
        table = (torch.rand(nfeat, E, generator=g) * 1.6 - 0.8).to(DEV)
    else:
        table = (torch.rand(nfeat, E, generator=g) * 0.9 + 0.05).to(DEV)
    qf = (torch.randn(O, E, generator=g) * 0.8).to(DEV)
    values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
    es, et = (torch.rand(F, generator=g) + 0.5).to(DEV), (torch.randn(F, generator=g) * 0.3).to(DEV)
    dz = torch.randn(B, O, E, generator=g).to(DEV)
    cA, cB, cC = (torch.rand(O, generator=g) + 0.5).to(DEV), (torch.randn(O, generator=g) * 0.1).to(DEV), (torch.randn(O, generator=g) * 0.1).to(DEV)
    x = (table[ids] * vals[..., None]).to(D)
    y = (torch.exp(x) if kind == "gc" else torch.log(x)) * es.to(D)[None, :, None] + et.to(D)[None, :, None]
    if kind == "gc":
        gat = torch.einsum("bfe,oe->bof", x, qf.to(D))
        gat = gat + gat.sum(-1, keepdim=True)
        p = entmax64(gat, alpha)
        w = p * values.to(D)[None]
        z = torch.einsum("bof,bfe->boe", w, y)
    else:
        bias = (torch.randn(O, generator=g) * 0.2).to(DEV)
        w = values.to(D)[None].expand(B, O, F)
        z = torch.exp(torch.einsum("bof,bfe->boe", w, y) + bias.to(D)[None, :, None])
    z32 = z.float().contiguous()
    ds = cA.to(D)[None, :, None] * dz.to(D) + cC.to(D)[None, :, None] * z32.to(D) + cB.to(D)[None, :, None]
    if kind == "afn":
        ds = ds * z32.to(D)
    dW = torch.einsum("boe,bfe->bof", ds, y)
    d_y_ref = torch.einsum("bof,boe->bfe", w, ds)
    d_table = torch.zeros(nfeat, E, device=DEV)
    d_y = torch.full((B, F, E), float("nan"), device=DEV)
    out = {}
    slack, mass = {}, {}
    if kind == "gc":
        d_values_ref = (p * dW).sum(0)
        mass = {"d_values": float((p * dW).abs().sum(0).max())}   # signed sums over the batch: judged against the terms' mass too

        def through_gates(pp):
            dg = jvp_T(pp, values.to(D)[None] * dW, alpha)
            dq = torch.einsum("bof,bfe->oe", dg, x)
            dx = torch.einsum("bof,oe->bfe", dg, qf.to(D)) * vals.to(D)[..., None]
            return dq, torch.zeros(nfeat, E, device=DEV, dtype=D).index_add_(0, ids.reshape(-1), dx.reshape(-1, E))
        d_qf_ref, d_table_ref = through_gates(p)
        margin = entmax64.margin
        # conditioning: the Jacobian of the sparse map jumps when an element enters the support (alpha = 2: by a finite amount,
        # alpha > 2: p^(2-alpha) is unbounded at p -> 0+), so a gate within fp32 rounding of its row's threshold makes the
        # float64 answer itself ambiguous: the same reference from gates moved by fp32-sized noise bounds what can be asked
        noise = torch.randn(gat.shape, generator=torch.Generator(device=DEV).manual_seed(1), device=DEV, dtype=D)
        slack = {"d_qfold": 0.0, "d_table": 0.0}
        for sign in (1.0, -1.0):                        # (a gate on the edge flips for one of the two signs)
            dq2, dt2 = through_gates(entmax64(gat + sign * 2e-6 * gat.abs().max() * noise, alpha))
            slack["d_qfold"] = max(slack["d_qfold"], float((dq2 - d_qf_ref).abs().max()) / max(float(d_qf_ref.abs().max()), 1e-12))
            slack["d_table"] = max(slack["d_table"], float((dt2 - d_table_ref).abs().max()) / max(float(d_table_ref.abs().max()), 1e-12))
        d_values, d_qf = torch.zeros(O, F, device=DEV), torch.zeros(O, E, device=DEV)
        native.gc_fused_bwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, es, et, z32, dz, cA, cB, cC, d_table,
                            d_values, d_qf, d_y)
        out = {"d_table": (d_table, d_table_ref), "d_values": (d_values, d_values_ref), "d_qfold": (d_qf, d_qf_ref)}
    else:
        d_w, d_b = torch.zeros(O, F, device=DEV), torch.zeros(O, device=DEV)
        native.afn_fused_bwd(B, F, E, O, 0, ids, vals, table, values, es, et, z32, dz, cA, cB, cC, d_w, d_b, d_y)
        out = {"d_weight": (d_w, dW.sum(0)), "d_bias": (d_b, ds.sum((0, 2)))}
        mass = {"d_bias": float(ds.abs().sum((0, 2)).max())}       # a signed sum of B*E terms: judged against the terms' mass
    out["d_y"] = (d_y, d_y_ref)
    worst = 0.0
    for k, (a, b) in out.items():
        scale = max(float(b.abs().max()), 1e-12)
        if k in mass:
            scale = max(scale, 0.05 * mass[k])
        err = float((a.to(D) - b).abs().max()) / scale
        if not (err <= (2e-3 if alpha > 2 else 3e-5) + 4.0 * slack.get(k, 0.0)):
            print(f"{kind} F={F} E={E} O={O} alpha={alpha} {k}: rel err {err:.2e} (conditioning {slack.get(k, 0.0):.1e}, closest gate to a threshold {margin if kind == 'gc' else 0:.1e} of {float(gat.abs().max()) if kind == 'gc' else 0:.1f})", flush=True)
            worst = max(worst, err)
    return worst


def main():
    quick = "--quick" in sys.argv
    wide = "--wide" in sys.argv
    g = torch.Generator().manual_seed(0)
    n = bad = 0
    for F in range(1, 33 if wide else 49):
        for E in (tuple(range(65, 129)) if wide else (4, 10, 16, 27, 32, 48, 64) if quick else tuple(range(4, 65))):
            if not quick and (F * 7 + E) % (5 if wide else 3):
                continue
            for O in (1, 20, 70):
                for kind, alphas in (("gc", (1.0, 1.5, 1.7, 2.0) + ((2.5,) if (F + E) % 5 == 0 else ())), ("afn", (1.0,))):
                    for alpha in alphas:
                        if quick and (F + E + O + int(alpha * 10)) % 4:
                            continue
                        n += 1
                        bad += case(kind, F, E, O, alpha, 37, g) > 0
    print(f"{n} cases scanned, {bad} with disagreements")


if __name__ == "__main__":
    main()
