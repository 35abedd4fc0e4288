"""Block time against the neuron count for the ARM block and the AFN mode (no gates, no sparse map) on rotating batches:
separates the memory side of the fused kernel from its instruction mix."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "arm-net_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from models.armnet import ARMNetModel
from models.afn import AFNModel

dev = "cuda:0"
B, F, E, nfeat, R = 65536, 39, 16, 1_000_000, 4
g = torch.Generator().manual_seed(0)
batches = [(torch.randint(0, nfeat, (B, F), generator=g).to(dev), torch.rand(B, F, generator=g).to(dev)) for _ in range(R)]


def timeit(fn, n=40):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:                   # let the device clocks settle (cold: ~20 % slower)
        for k in range(8):
            fn(k)
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for k in range(n):
        fn(k)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for O in (16, 32, 64, 128):
    outs = [torch.empty(B, O, E, device=dev) for _ in range(R)]
    arm = ARMNetModel(F, nfeat, E, 1, 2.0, O, 3, 512, 0.0, False, 1, 8).eval().to(dev)
    afn = AFNModel(F, nfeat, E, O, 3, 512, 0.0, False, 1, 8).eval().to(dev)
    arm.check_ids = afn.check_ids = False                 # no host sync per call
    with torch.no_grad():
        afn.embedding_clip()
        t_arm = timeit(lambda k: arm.arm_block(batches[k % R][0], batches[k % R][1], out=outs[k % R]))
        t_afn = timeit(lambda k: afn.afn_block(batches[k % R][0], batches[k % R][1]))
    mb = (B * F * (8 + 4 + 64) + B * O * E * 4) / 1e6
    print(f"O={O:4d}  bytes {mb:6.0f} MB   arm {t_arm:7.1f} us ({mb / t_arm:5.2f} TB/s)   afn {t_afn:7.1f} us ({mb / t_afn:5.2f} TB/s)", flush=True)
