"""Fused block with 1 / 2 / 3 batches in flight on alternating streams (rotating batches, clocks settled)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "arm-net_amd"))
import torch
import bench

sys.argv = ["bench.py", "--regime", "fresh"]
a = bench.parse()
dev = torch.device("cuda", 0)
model = bench.build_model(a, dev, regime="fresh")
batches = [bench.make_batch(a, 0, dev, k) for k in range(4)]
outs = [torch.empty(a.batch, a.nhid, a.nemb, device=dev) for _ in range(4)]
for nfl in (1, 2, 3, 1, 2, 3):
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())

    def step(k):
        with torch.no_grad(), torch.cuda.stream(streams[k % nfl]):
            return model.arm_block(batches[k % 4][0], batches[k % 4][1], out=outs[k % 4])
    bench.settle_clocks(lambda: step(0), 150)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 200
    for k in range(N):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {nfl}: {a.batch * N / dt / 1e6:7.1f} M samples/s  ({dt / N * 1e6:6.1f} us per batch)", flush=True)
