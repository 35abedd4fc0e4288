#!/usr/bin/env python3
"""armnet_gather_scale_f32 / armnet_scatter_add_f32 timing (developer tool, GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native


def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for B, F, E, nfeat in ((65536, 39, 16, 1_000_000), (131072, 22, 32, 2_000_000), (65536, 39, 10, 1_000_000), (65536, 39, 64, 1_000_000)):
    table = torch.randn(nfeat, E, device="cuda")
    ids = torch.randint(0, nfeat, (B, F), device="cuda")
    vals = torch.rand(B, F, device="cuda")
    out = torch.empty(B, F, E, device="cuda")
    us = t(lambda: native.gather_scale(B * F, E, ids, vals, table, out))
    mb = B * F * (12 + 8 * E) / 1e6
    g = torch.randn(B, F, E, device="cuda")
    dt = torch.zeros_like(table)
    us2 = t(lambda: native.scatter_add(ids, vals, g, dt))
    print(f"B={B} F={F} E={E}: gather_scale {us:7.1f} us ({mb / us:5.2f} TB/s alg)   scatter_add {us2:7.1f} us ({mb / us2:5.2f} TB/s alg)")
