#!/usr/bin/env python3
"""Per-phase s_memtime breakdown of the fused kernel (needs lib/exp/libarmnet_phase.so, built with
-DARMNET_PHASE_TIMING; run with ARMNET_HIP_LIB pointing at it)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native

def main():
    alpha = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    regime = sys.argv[2] if len(sys.argv) > 2 else "fresh"
    flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
    O = int(sys.argv[4]) if len(sys.argv) > 4 else 32
    B, F, E, nfeat = 65536, 39, 16, 1_000_000
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1)
    bound = (6.0 / (nfeat + E)) ** 0.5
    table = ((torch.rand(nfeat, E, generator=g) * 2 - 1) * (bound if regime == "fresh" else 0.9)).to(dev)
    qf = (torch.randn(O, E, generator=g) * (0.3 if regime == "fresh" else 1.5)).to(dev)
    values = (torch.randn(O, F, generator=g) * 0.3).to(dev)
    sc = (torch.rand(O, generator=g) + 0.5).to(dev); sh = torch.randn(O, generator=g).to(dev)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(dev); vals = torch.rand(B, F, generator=g).to(dev)
    out = torch.empty(B, O, E, device=dev)
    st = torch.zeros(16, device=dev, dtype=torch.int32)
    for _ in range(3):
        native.fused_fwd(B, F, E, O, alpha, 50, flags, ids, vals, table, qf, values, sc, sh, out, None)
    torch.cuda.synchronize()
    native.fused_fwd(B, F, E, O, alpha, 50, flags, ids, vals, table, qf, values, sc, sh, out, st)
    torch.cuda.synchronize()
    v = st.cpu().numpy().astype("uint32").astype(float)[:11]
    names = ["rest of the staging phase", "MFMA#1 (+A reads)", "setup (sum,max,LDS reduce)", "newton loop", "final w + MFMA#2", "epilogue+store",
             "wait for the group's rows", "staging (clamp, scale, LDS writes)", "wait for the next group's ids",
             "issue value + row loads of the next group", "issue id loads of the group after"]
    tot = v.sum()
    for n, x in zip(names, v):
        print(f"{n:28s} {100 * x / tot:5.1f} %")
    print("alpha", alpha, regime, "flags", hex(flags), "neurons", O)

main()
