#!/usr/bin/env python3
"""round 6: soak of the fp16 x 2 form of the wide blocks — many batches, every solver instantiation, several shapes: each launch against the
fp32 form (5e-6 of max(1, |ref|)) and bit-equal to a second launch on the same inputs (run-to-run determinism under different wave timing)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native
dev = "cuda:0"
nfeat = 200_000
worst, n, t0 = 0.0, 0, time.time()
g = torch.Generator().manual_seed(0)
for it in range(60):
    for F, E, O in ((39, 16, 128), (43, 10, 256), (30, 16, 64), (22, 16, 512), (39, 16, 300)):
        for alpha in (1.0, 1.5, 1.7, 2.0):
            B = (65536, 20011, 4096)[it % 3]
            table = ((torch.rand(nfeat, E, generator=g) * 2 - 1) * (0.9 if it % 2 else 0.003)).to(dev)
            qf = (torch.randn(O, E, generator=g) * (1.0 if it % 2 else 0.3)).to(dev)
            values = (torch.randn(O, F, generator=g) * 0.3).to(dev)
            sc, sh = (torch.rand(O, generator=g) + 0.5).to(dev), torch.randn(O, generator=g).to(dev)
            ids = torch.randint(0, nfeat, (B, F), generator=g).to(dev)
            vals = torch.rand(B, F, generator=g).to(dev)
            outs = []
            for fl in (0, 0, native.F_FP32_CONTRACTIONS):
                z = torch.empty(B, O, E, device=dev)
                native.fused_fwd(B, F, E, O, alpha, 50, fl, ids, vals.clone(), table, qf, values, sc, sh, z)
                outs.append(z)
            assert torch.equal(outs[0], outs[1]), (it, F, E, O, alpha, "run-to-run difference")
            rel = float(((outs[0] - outs[2]).abs() / outs[2].abs().clamp(min=1.0)).max())
            assert rel <= 5e-6 and bool(torch.isfinite(outs[0]).all()), (it, F, E, O, alpha, rel)
            worst = max(worst, rel)
            n += 1
    if time.time() - t0 > 400:
        break
print(f"{n} launches x 2 of the fp16 x 2 form: bit-equal run to run, worst difference to the fp32 form {worst:.2e} of max(1, |ref|)")
