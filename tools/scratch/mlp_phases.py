#!/usr/bin/env python3
"""Developer tool (dev build of the library, GPU box): per-phase s_memtime sums of mlp_head_kernel's layer-1 stages."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native
from armnet_hip.modules import _MLP
lib = native.load()
B, K0, nl, nh = int(os.environ.get("B", 65536)), 512, int(os.environ.get("NL", 2)), 256
m = _MLP(K0, nl, nh, 0.0).eval().to("cuda:0")
x = torch.randn(B, K0, device="cuda:0")
buf = (ctypes.c_ulonglong * 8)()
with torch.no_grad():
    for _ in range(3):
        m(x)
    lib.armnet_dev_mlp_phases(buf)
    n = 10
    for _ in range(n):
        m(x)
    lib.armnet_dev_mlp_phases(buf)
waves, stages = (B + 31) // 32, K0 // 16
names = ["wait vmcnt", "barrier", "issue + first reads", "units (MFMA)"]
tot = sum(buf[i] for i in range(4))
for i in range(4):
    print(f"{names[i]:22s} {buf[i] / n / waves / stages:9.1f} ticks per stage per wave  ({100.0 * buf[i] / tot:.1f} %)")
print("sum", tot / n / waves / stages, "ticks per stage (48 MFMAs = 1536 shader cycles if s_memtime ticks at the shader clock)")
