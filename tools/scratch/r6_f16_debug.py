import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native
dev = "cuda:0"
B, F, E, O, nfeat = 8, 39, 16, 32, 1000
for alpha in (1.0, 2.0, 1.5, 1.7):
    g = torch.Generator().manual_seed(1)
    table = ((torch.rand(nfeat, E, generator=g) * 2 - 1) * 0.9).to(dev)
    qf = (torch.randn(O, E, generator=g) * 1.5).to(dev)
    values = (torch.randn(O, F, generator=g) * 0.3).to(dev)
    sc = (torch.rand(O, generator=g) + 0.5).to(dev); sh = torch.randn(O, generator=g).to(dev)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(dev)
    vals = torch.rand(B, F, generator=g).to(dev)
    o1, o2 = torch.empty(B, O, E, device=dev), torch.empty(B, O, E, device=dev)
    native.fused_fwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, sc, sh, o1)
    native.fused_fwd(B, F, E, O, alpha, 50, 0x10, ids, vals, table, qf, values, sc, sh, o2)
    d = ((o1 - o2).abs() / o2.abs().clamp(min=1.0))
    bad = (d > 1e-5)
    print(f"alpha {alpha}: max rel err {float(d.max()):.3e}; bad elements {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        print("  bad per e :", bad.sum((0, 1)).tolist())
        print("  bad per o :", bad.sum((0, 2)).tolist())
        print("  bad per b :", bad.sum((1, 2)).tolist())
