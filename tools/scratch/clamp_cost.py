#!/usr/bin/env python3
"""What the in-place clamp write-back costs (developer tool, GPU box): the fused kernel on a batch whose values were
already clamped by an earlier call (bench.py's steady state) against a fresh batch every call."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native

B, F, E, O, nfeat = 65536, 39, 16, 32, 1_000_000
g = torch.Generator().manual_seed(1)
table = ((torch.rand(nfeat, E, generator=g) * 2 - 1) * 0.002).cuda()
qf = (torch.randn(O, E, generator=g) * 0.3).cuda(); values = (torch.randn(O, F, generator=g) * 0.3).cuda()
sc, sh = torch.ones(O).cuda(), torch.zeros(O).cuda()
ids = torch.randint(0, nfeat, (B, F), generator=g).cuda()
src = torch.rand(B, F, generator=g).cuda()
vals = src.clone(); out = torch.empty(B, O, E).cuda()
fl = native.F_WRITE_CLAMPED_VALS
for fresh in (False, True):
    tot = 0.0
    for it in range(60):
        if fresh or it == 0:
            vals.copy_(src)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        native.fused_fwd(B, F, E, O, 2.0, 50, fl, ids, vals, table, qf, values, sc, sh, out)
        e1.record(); torch.cuda.synchronize()
        if it >= 10:
            tot += e0.elapsed_time(e1)
    print(f"fresh values every call: {fresh}:  {tot / 50 * 1e3:.1f} us per launch  ({int((src < 1e-3).sum())} of {B * F} values need the clamp)")
