"""torch.profiler view of a few whole forwards (kernels, copies, host ops)."""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "arm-net_amd")); sys.path.insert(0, ROOT)
import torch
from torch.profiler import profile, ProfilerActivity
from models.armnet_1h import ARMNetModel
dev = "cuda:0"
m = ARMNetModel(39, 1000000, 16, 2.0, 32, 16, 2, 256, 0.0, False, 2, 256).eval().to(dev)
m.check_ids = False
B = 65536
ids = torch.randint(0, 1000000, (B, 39), device=dev); vals = torch.rand(B, 39, device=dev)
with torch.no_grad():
    for _ in range(5): m({"id": ids, "value": vals})
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(5): m({"id": ids, "value": vals})
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
for e in prof.events():
    if "emcpy" in e.name or "copy" in e.name.lower():
        print(e.name, e.device_type, [str(s) for s in (e.stack or [])][:6]); break
