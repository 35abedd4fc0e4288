#!/usr/bin/env python3
"""Prediction-head timing (GPU box): armnet_mlp_head_f32 against the folded hipBLASLt path, B = 65 536."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip.modules import _MLP

DEV = "cuda:0"
B = int(os.environ.get("B", 65536))
CFGS = ((512, 2, 256), (2048, 2, 256), (704, 2, 256), (512, 2, 128), (512, 1, 256), (320, 2, 32))
if os.environ.get("ONLY"):
    CFGS = CFGS[:int(os.environ["ONLY"])]
MODES = (("hip", True), ("hip_bf16x3", True)) if os.environ.get("HIP_ONLY") else (("hip", True), ("hip_bf16x3", True), ("blas", False))
for K0, nlayers, nhid in CFGS:
    m = _MLP(K0, nlayers, nhid, 0.0).eval().to(DEV)
    xs = [torch.randn(B, K0, device=DEV) for _ in range(3)]
    res = {}
    for name, flag in MODES:
        m.hip_head = flag
        m.bf16x3 = name == "hip_bf16x3"
        with torch.no_grad():
            import time
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < float(os.environ.get('SETTLE', 1.0)):           # let the device clocks settle (cold: ~20 % slower)
                for i in range(6):
                    m(xs[i % 3])
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(30):
                m(xs[i % 3])
            e1.record()
            torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 30 * 1e3
    fl = 2.0 * B * (K0 * nhid + (nlayers - 1) * nhid * nhid + nhid)
    line = f"K0={K0} nlayers={nlayers} nhid={nhid} B={B}: HIP head {res['hip']:.1f} us ({fl / res['hip'] / 1e6:.0f} fp32-equivalent TFLOP/s)"
    line += f", bf16x3 split {res['hip_bf16x3']:.1f} us"
    if "blas" in res:
        line += f", hipBLASLt {res['blas']:.1f} us ({fl / res['blas'] / 1e6:.0f} TFLOP/s)"
    print(line)
