"""Whole forward with 1 / 2 / 3 batches in flight on alternating streams (rotating batches), samples/s."""
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
for nfl in (1, 2, 3, 4, 1, 2, 3, 4):
    streams = [torch.cuda.Stream() for _ in range(nfl)]
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())

    def step(k):
        with torch.no_grad(), torch.cuda.stream(streams[k % nfl]):
            return model({"id": batches[k % 4][0], "value": batches[k % 4][1]})
    for k in range(8):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 60
    for k in range(N):
        step(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"in flight {nfl}: {a.batch * N / dt / 1e6:7.1f} M samples/s  ({dt / N * 1e6:6.1f} us per forward)", flush=True)
