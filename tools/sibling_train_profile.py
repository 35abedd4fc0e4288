#!/usr/bin/env python3
"""torch.profiler kernel table of the GC-ARM / AFN training step at B = 65 536 (developer tool, GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
from models.afn import AFNModel  # noqa: E402
from models.gc_arm import GC_ARMModel  # noqa: E402

F, E, nfeat, B = 39, 16, 1_000_000, 65536
lossf = torch.nn.BCEWithLogitsLoss()
for name, build in (("gc_arm K=2 H=32 alpha 1.7", lambda: GC_ARMModel(F, nfeat, E, 2, 1.7, 32, 2, 256, 0.0, False, 2, 256)),
                    ("afn H=64", lambda: AFNModel(F, nfeat, E, 64, 2, 256, 0.0, False, 2, 256))):
    torch.manual_seed(0)
    m = build().cuda().train()
    m.check_ids = False
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ids = torch.randint(0, nfeat, (B, F)).cuda()
    vals = torch.rand(B, F).cuda()
    y = (torch.rand(B) > 0.5).float().cuda()

    def step():
        loss = lossf(m({"id": ids, "value": vals}), y)
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    print("=====", name)
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=90))
