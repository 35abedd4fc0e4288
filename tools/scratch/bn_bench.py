#!/usr/bin/env python3
"""BatchNorm pass timing (developer tool, GPU box): HipBatchNorm1d training forward / backward vs torch's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip.modules import HipBatchNorm1d


def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for shape in ((65536, 32, 16), (65536, 256), (4096, 32, 16), (4096, 256), (65536, 128, 10)):
    x = torch.randn(*shape, device="cuda") + 1.0
    dy = torch.randn(*shape, device="cuda")
    mb = x.numel() * 4 / 1e6
    for cls in (torch.nn.BatchNorm1d, HipBatchNorm1d):
        bn = cls(shape[1]).cuda().train()
        xr = x.clone().requires_grad_(True)
        fwd = t(lambda: bn(xr))
        y = bn(xr)
        bwd = t(lambda: torch.autograd.grad(y, (xr, bn.weight, bn.bias), dy, retain_graph=True))
        print(f"{cls.__name__:16s} {str(shape):18s} {mb:6.1f} MB  fwd {fwd:7.1f} us ({3 * mb / fwd:5.2f} TB/s eff)  bwd {bwd:7.1f} us ({5 * mb / bwd:5.2f} TB/s eff)")
