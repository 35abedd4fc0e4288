#!/usr/bin/env python3
"""Where the whole forward's time goes (developer tool, GPU box; round-2 verdict item 5): the fused block and the
prediction head alone and chained, with the block's output either re-used from ONE buffer (it then sits in the 256 MiB
Infinity Cache when the head reads it) or rotated over four buffers like bench.py (536 MB of activations: the head reads
them from HBM).  If `chained` ~= `block` + `head` and `head (one buffer)` ~= `head (four buffers)`, neither the launch
gap nor the activation round trip through HBM is what a block->head fusion would win back."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402


def timeit(fn, steps=100):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        for _ in range(16):
            fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def main():
    sys.argv = [sys.argv[0]] + sys.argv[1:]
    a = bench.parse()
    dev = torch.device("cuda", 0)
    a.shard = "replicate"
    m = bench.build_model(a, dev, regime="fresh")
    NB = 4
    batches = [bench.make_batch(a, 0, dev, k) for k in range(NB)]
    O = a.nhead * a.nhid
    outs = [torch.empty(a.batch, O, a.nemb, device=dev) for _ in range(NB)]
    t = [0]

    def block_rot():
        k = t[0] % NB; t[0] += 1
        with torch.no_grad():
            m.arm_block(batches[k][0], batches[k][1], out=outs[k])

    def head_one():
        with torch.no_grad():
            m.mlp(outs[0].view(a.batch, -1))

    def head_rot():
        k = t[0] % NB; t[0] += 1
        with torch.no_grad():
            m.mlp(outs[k].view(a.batch, -1))

    def chained_rot():
        k = t[0] % NB; t[0] += 1
        with torch.no_grad():
            m.arm_block(batches[k][0], batches[k][1], out=outs[k])
            m.mlp(outs[k].view(a.batch, -1))

    def chained_one():
        with torch.no_grad():
            m.arm_block(batches[0][0], batches[0][1], out=outs[0])
            m.mlp(outs[0].view(a.batch, -1))

    for k in range(NB):
        block_rot()
    r = {n: timeit(f) for n, f in (("block (4 rotating batches)", block_rot), ("head, ONE activation buffer (MALL-resident)", head_one),
                                   ("head, four activation buffers (536 MB: from HBM)", head_rot),
                                   ("block -> head chained, four buffers", chained_rot), ("block -> head chained, one buffer", chained_one))}
    for n, v in r.items():
        print(f"{n:52s} {v:8.1f} us")
    print(f"sum of the parts (rotating) {r['block (4 rotating batches)'] + r['head, four activation buffers (536 MB: from HBM)']:.1f} us "
          f"against chained {r['block -> head chained, four buffers']:.1f} us")


main()
