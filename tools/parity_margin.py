#!/usr/bin/env python3
"""Developer tool (GPU box): worst elementwise error of the fused block over every eval fixture, in units of the 1e-5 bar
(|got - ref| / (1e-5 * max(1, |ref|))) — the parity margin a solver tolerance or epilogue change has to live within."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from golden_util import load, model_cases
from model_util import build_model
from tol_util import elem_excess

worst = {}
for name in [n for n in model_cases() if "train" not in n]:
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd, "cuda:0")
    with torch.no_grad():
        got = m.arm_block(torch.from_numpy(ids).cuda(), torch.from_numpy(vals.copy()).cuda()).cpu().numpy()
    a = float(meta["ctor"]["alpha"])
    worst[a] = max(worst.get(a, 0.0), elem_excess(got, ref["x_arm"].reshape(got.shape), 1e-5))
for a in sorted(worst):
    print(f"alpha {a}: worst element at {worst[a]:.4f} x the 1e-5 bar ({worst[a] * 1e-5:.2e})")
