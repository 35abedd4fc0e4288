import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native
DEV = "cuda:0"
B, F, E, O, alpha = 20011, 3, 10, 24, 1.7
nfeat = 5003
g = torch.Generator().manual_seed(F * 1000 + E * 10 + O)
table = (torch.rand(nfeat, E, generator=g) * 1.6 - 0.8).to(DEV)
qf = (torch.randn(O, E, generator=g) * 0.8).to(DEV)
values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
vals0 = (torch.rand(B, F, generator=g) * 1.2 - 0.1).to(DEV)
sc, sh = (torch.rand(O, generator=g) + 0.5).to(DEV), torch.randn(O, generator=g).to(DEV)
outs = {}
for rep in range(2):
    for name, idt in (("i64", ids), ("i32", ids.to(torch.int32))):
        v = vals0.clone(); z = torch.empty(B, O, E, device=DEV)
        native.fused_fwd(B, F, E, O, alpha, 50, native.F_WRITE_CLAMPED_VALS, idt, v, table, qf, values, sc, sh, z)
        outs[(name, rep)] = z
    rows = table[ids].contiguous(); v = vals0.clone(); z = torch.empty(B, O, E, device=DEV)
    native.fused_fwd_from_rows(B, F, E, O, alpha, 50, native.F_WRITE_CLAMPED_VALS, rows, v, qf, values, sc, sh, z)
    outs[("rows", rep)] = z
ref = outs[("i64", 0)]
for k, z in outs.items():
    d = (z - ref).abs()
    nz = (d > 0).nonzero()
    print(k, "differing elements", nz.shape[0], "max", float(d.max()), "rel", float((d / ref.abs().clamp_min(1e-6)).max()))
    if nz.shape[0]:
        print("   first:", nz[:5].tolist(), "samples parity", (nz[:, 0] % 2).float().mean().item(), "neurons", sorted(set(nz[:, 1].tolist()))[:30])
