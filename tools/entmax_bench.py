#!/usr/bin/env python3
"""Stand-alone sparse map (utils.entmax.entmax_bisect -> armnet_entmax_f32 / armnet_entmax_bwd_f32) on [B*O, F] rows:
time per call, forward and backward (developer tool, GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip import native  # noqa: E402


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for rows, d in ((65536 * 32, 39), (65536 * 64, 39), (65536 * 32, 10), (65536 * 8, 64)):
    for scale in (0.05, 2.0):
        X = (torch.randn(rows, d, device="cuda") * scale).contiguous()
        P, dY, dX = torch.empty_like(X), torch.randn_like(X), torch.empty_like(X)
        for alpha in (1.0, 1.5, 1.7, 2.0, 2.5):
            fw = t(lambda: native.entmax(rows, d, alpha, 50, True, 0, X, P))
            bw = t(lambda: native.entmax_bwd(rows, d, alpha, P, dY, dX))
            gb = rows * d * 4 / 1e9
            print(f"rows={rows:8d} d={d:2d} gate scale {scale:4.2f} alpha={alpha}: forward {fw:9.1f} us ({2 * gb / fw * 1e6 / 1e3:6.2f} TB/s)   "
                  f"backward {bw:9.1f} us ({3 * gb / bw * 1e6 / 1e3:6.2f} TB/s)", flush=True)
