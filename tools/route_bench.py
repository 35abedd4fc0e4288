"""Routing of the fixed-capacity protocol, round-3 kernels (route + pad_route) against armnet_shard_route_fixed (round 4):
time per call for n lookups, with / without de-duplication, R owners (HIP events, clocks warmed by the loop itself)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "arm-net_amd")]
from armnet_hip.sharded import HipShardOps  # noqa: E402

dev = "cuda:0"
ops = HipShardOps()


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for nfeat, n, dedup in ((1_000_000, 39 * 65536, True), (1_000_000, 39 * 65536, False), (100_000_000, 39 * 65536, False),
                        (10_000_000, 39 * 65536, True), (1_000_000, 39 * 8192, True)):
    ids = torch.randint(0, nfeat, (n,), device=dev)
    for R in (1, 8):
        cap = int(1.25 * n / R) + 4 * int((n / R) ** 0.5) + 16
        if dedup:
            cap = min(cap, (nfeat + R - 1) // R)
        over = torch.zeros(1, device=dev, dtype=torch.int32)

        def old():
            c, s, p = ops.route(ids, R, nfeat, dedup=dedup)
            return ops.pad_route(c, s, p, R, cap, over)

        def new():
            return ops.route_fixed(ids, R, nfeat, cap, dedup, over)

        t_old, t_new = timeit(old), timeit(new)
        print(f"nfeat={nfeat:>11,} n={n:>9,} R={R} dedup={int(dedup)} cap={cap:>9,}: route+pad {t_old:7.1f} us   "
              f"route_fixed {t_new:7.1f} us   overflow={int(over.item())}", flush=True)
