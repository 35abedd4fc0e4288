#!/usr/bin/env python3
"""Where a training step's GPU time goes (developer tool, GPU box): torch.profiler kernel table.
    python tools/train_profile.py [B] [alpha] [ens]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from torch.profiler import profile, ProfilerActivity
from models.armnet_1h import ARMNetModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
F, E, H, nfeat = 39, int(os.environ.get("NEMB", 16)), int(os.environ.get("NHID", 32)), 1_000_000
torch.manual_seed(0)
ens = len(sys.argv) > 3 and sys.argv[3] == "ens"
m = ARMNetModel(F, nfeat, E, alpha, H, E, 2, 256, 0.0, ens, 2, 256).cuda().train()
m.check_ids = False
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
ids = torch.randint(0, nfeat, (B, F)).cuda(); vals = torch.rand(B, F).cuda(); y = (torch.rand(B) > 0.5).float().cuda()
lossf = torch.nn.BCEWithLogitsLoss()
def step():
    loss = lossf(m({"id": ids, "value": vals}), y)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
