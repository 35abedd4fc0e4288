"""Time the GC-ARM / AFN block launches (matrix-core mode vs the shape-agnostic kernel) on the Criteo shape."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "arm-net_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from armnet_hip import native
from models.gc_arm import GC_ARMModel
from models.afn import AFNModel

dev = "cuda:0"
B, F, E, nfeat = 65536, 39, 16, 1_000_000
g = torch.Generator().manual_seed(0)
ids = torch.randint(0, nfeat, (B, F), generator=g).to(dev)
vals = torch.rand(B, F, generator=g).to(dev)


def timeit(fn, n=20):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:                   # let the device clocks settle (cold: ~20 % slower)
        for _ in range(8):
            fn()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, m, blk in (("gc_arm a=2.0 K=1 O=64", GC_ARMModel(F, nfeat, E, 1, 2.0, 64, 3, 512, 0.0, False, 1, 8), "arm_block"),
                     ("gc_arm a=1.7 K=8 O=64", GC_ARMModel(F, nfeat, E, 8, 1.7, 8, 3, 512, 0.0, False, 1, 8), "arm_block"),
                     ("afn O=64", AFNModel(F, nfeat, E, 64, 3, 512, 0.0, False, 1, 8), "afn_block")):
    m = m.eval().to(dev)
    m.check_ids = False                                   # no host sync per call
    res = {}
    with torch.no_grad():
        for label, flags in (("matrix-core", 0), ("generic", native.F_FORCE_GENERIC)):
            m.kernel_flags = flags
            res[label] = timeit(lambda: getattr(m, blk)(ids, vals))
    print(f"{name:28s} " + "  ".join(f"{k} {v:8.1f} us" for k, v in res.items()), flush=True)
