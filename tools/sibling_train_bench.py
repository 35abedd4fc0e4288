#!/usr/bin/env python3
"""Training-step timing (forward + backward + Adam) of the GC-ARM / AFN modules on synthetic Criteo-shaped data
(developer tool, GPU box): armnet_hip/siblings.py (fused block backward where the shape has a kernel), eager and as one hipGraph."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from models.afn import AFNModel  # noqa: E402
from models.gc_arm import GC_ARMModel  # noqa: E402


def main():
    F, E, nfeat = 39, 16, 1_000_000
    lossf = torch.nn.BCEWithLogitsLoss()
    for B in (4096, 65536):
        for name, build in (("gc_arm K=2 H=32 alpha 1.7", lambda: GC_ARMModel(F, nfeat, E, 2, 1.7, 32, 2, 256, 0.0, False, 2, 256)),
                            ("afn H=64", lambda: AFNModel(F, nfeat, E, 64, 2, 256, 0.0, False, 2, 256))):
            torch.manual_seed(0)
            m = build().cuda().train()
            m.check_ids = False
            if os.environ.get("ARMNET_HEAD_GEMM", "mfma") == "hipblaslt":   # A/B: the head's Linear forward / dX on hipBLASLt fp32
                m.mlp.mfma_train = False
            opt = torch.optim.Adam(m.parameters(), lr=1e-3)
            ids = torch.randint(0, nfeat, (B, F)).cuda()
            vals = torch.rand(B, F).cuda()
            y = (torch.rand(B) > 0.5).float().cuda()

            def step():
                loss = lossf(m({"id": ids, "value": vals}), y)
                opt.zero_grad()
                loss.backward()
                opt.step()

            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 0.3:
                step()
                torch.cuda.synchronize()
            n = 10
            t0 = time.perf_counter()
            for _ in range(n):
                step()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            m.eval()
            with torch.no_grad():
                for _ in range(3):
                    m({"id": ids, "value": vals})
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    m({"id": ids, "value": vals})
                torch.cuda.synchronize()
                di = (time.perf_counter() - t0) / n
            # the same step replayed as one hipGraph (armnet_hip.modules.GraphedTrainStep): what the host costs at small batches
            from armnet_hip.modules import GraphedTrainStep
            m.train()
            opt2 = torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True)
            gstep = GraphedTrainStep(m, opt2, lossf, ids, vals, y)
            for _ in range(3):
                gstep(ids, vals, y)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                gstep(ids, vals, y)
            torch.cuda.synchronize()
            dg = (time.perf_counter() - t0) / n
            print(f"{name:28s} B={B:6d}: train step {dt * 1e3:8.2f} ms ({B / dt / 1e6:6.2f} M samples/s), as one hipGraph {dg * 1e3:6.2f} ms "
                  f"({B / dg / 1e6:6.2f} M)   eval forward {di * 1e3:7.3f} ms ({B / di / 1e6:7.1f} M samples/s)")
            del m, opt


main()
