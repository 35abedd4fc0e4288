#!/usr/bin/env python3
"""round 6: the siblings at nemb 65..128 — block forward on the matrix cores vs the shape-agnostic kernel, training step through
the fused matrix-core backward vs the composed device ops (developer tool, GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip import native  # noqa: E402
from models.afn import AFNModel  # noqa: E402
from models.gc_arm import GC_ARMModel  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


lossf = torch.nn.BCEWithLogitsLoss()
for F, E, nfeat, B in ((10, 100, 5382, 4096), (10, 100, 5382, 65536), (22, 100, 1_000_000, 16384), (32, 128, 1_000_000, 16384)):
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = torch.rand(B, F, generator=g).to(DEV)
    y = (torch.rand(B, generator=g) > 0.5).float().to(DEV)
    for name, build, blk in (("gc_arm K=2 H=16 a=1.7", lambda: GC_ARMModel(F, nfeat, E, 2, 1.7, 16, 2, 256, 0.0, False, 2, 256), "arm_block"),
                             ("afn H=32", lambda: AFNModel(F, nfeat, E, 32, 2, 256, 0.0, False, 2, 256), "afn_block")):
        torch.manual_seed(0)
        m = build().to(DEV)
        m.check_ids = False
        m.eval()
        res = {}
        with torch.no_grad():
            for label, flags in (("mfma", 0), ("generic", native.F_FORCE_GENERIC)):
                m.kernel_flags = flags
                res[label] = timeit(lambda: getattr(m, blk)(ids, vals))
            m.kernel_flags = 0
            res["eval forward"] = timeit(lambda: m({"id": ids, "value": vals}))
        m.train()
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)

        def step():
            loss = lossf(m({"id": ids, "value": vals}), y)
            opt.zero_grad()
            loss.backward()
            opt.step()

        for fused in (True, False):
            m.fused_training = fused
            res["train fused" if fused else "train composed"] = timeit(step, n=10)
        print(f"F={F:2d} E={E:3d} B={B:6d} {name:22s} " + "  ".join(f"{k} {v:9.1f} us" for k, v in res.items()), flush=True)
        del m, opt
