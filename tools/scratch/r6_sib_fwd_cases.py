#!/usr/bin/env python3
"""The alpha = 2.5 cases where tools/sibling_fwd_scan.py's two fp32 kernels differ by more than 1e-5: each against float64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip import native  # noqa: E402
from sibling_bwd_scan import entmax64  # noqa: E402

DEV, D = "cuda:0", torch.float64
B, nfeat = 37, 53
for F, E, O, alpha in ((29, 16, 70, 2.5), (43, 33, 24, 2.5), (23, 72, 70, 2.5), (42, 66, 7, 2.5), (42, 128, 70, 2.5), (39, 16, 32, 2.5)):
    g = torch.Generator().manual_seed(F * 1000 + E * 10 + O)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
    es, et = (torch.rand(F, generator=g) + 0.5).to(DEV), (torch.randn(F, generator=g) * 0.3).to(DEV)
    sc, sh = (torch.rand(O, generator=g) + 0.5).to(DEV), torch.randn(O, generator=g).to(DEV)
    table = (torch.rand(nfeat, E, generator=g) * 1.6 - 0.8).to(DEV)
    qf = (torch.randn(O, E, generator=g) * (0.8 / max(1.0, (E / 16) ** 0.5))).to(DEV)
    values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
    z, zg = torch.empty(B, O, E, device=DEV), torch.empty(B, O, E, device=DEV)
    for out, fl in ((z, 0), (zg, native.F_FORCE_GENERIC)):
        native.gc_fused_fwd(B, F, E, O, alpha, 50, fl, ids, vals.clone(), table, qf, values, es, et, sc, sh, out)
    x = (table[ids] * vals[..., None]).to(D)
    y = torch.exp(x) * es.to(D)[None, :, None] + et.to(D)[None, :, None]
    gat = torch.einsum("bfe,oe->bof", x, qf.to(D))
    gat = gat + gat.sum(-1, keepdim=True)
    p = entmax64(gat, alpha, n_iter=80)
    ref = torch.einsum("bof,bfe->boe", p * values.to(D)[None], y) * sc.to(D)[None, :, None] + sh.to(D)[None, :, None]
    # the reference's own fp32 bisection stops after 50 halvings of an interval of width ~1: it is exact to fp32 rounding of tau
    den = max(1.0, float(ref.abs().max()))
    print(f"F={F} E={E} O={O} alpha={alpha}: matrix-core vs float64 {float((z.to(D) - ref).abs().max()) / den:.2e}   shape-agnostic vs float64 "
          f"{float((zg.to(D) - ref).abs().max()) / den:.2e}   between them {float((z - zg).abs().max()) / den:.2e}   largest |gate| {float(gat.abs().max()):.1f}"
          f"   closest gate to its threshold {entmax64.margin:.1e}", flush=True)
