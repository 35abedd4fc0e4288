"""Host time per arm_block call (ctypes + guards) and a cProfile of it."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "arm-net_amd")); sys.path.insert(0, ROOT)
import torch
from models.armnet_1h import ARMNetModel
dev = "cuda:0"
m = ARMNetModel(39, 1000000, 16, 2.0, 32, 16, 2, 256, 0.0, False, 2, 256).eval().to(dev)
m.check_ids = False
B = 64
ids = torch.randint(0, 1000000, (B, 39), device=dev); vals = torch.rand(B, 39, device=dev)
out = torch.empty(B, 32, 16, device=dev)
with torch.no_grad():
    for _ in range(50): m.arm_block(ids, vals, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 2000
    for _ in range(N): m.arm_block(ids, vals, out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"host time per arm_block call: {(t1 - t0) / N * 1e6:.1f} us")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): m.arm_block(ids, vals, out=out)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
