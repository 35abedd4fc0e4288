#!/usr/bin/env python3
"""Kernel-level micro benchmark / ablation of the fused block (developer tool, GPU box only).

    python tools/kbench.py [--alpha 2.0] [--regime fresh|stress] [--flags 0x100 ...]
Times armnet_fused_fwd_f32 alone (HIP events around back-to-back launches)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alpha", type=float, default=2.0)
    ap.add_argument("--regime", default="fresh")
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--F", type=int, default=39)
    ap.add_argument("--E", type=int, default=16)
    ap.add_argument("--O", type=int, default=32)
    ap.add_argument("--nfeat", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--flags", type=lambda s: int(s, 0), nargs="*", default=[0])
    ap.add_argument("--cus", type=int, default=0, help="launch on a stream restricted to the first n bits of the CU mask "
                                                         "(bit i -> XCD i %% 8): is the kernel bound by its CUs or by the fabric?")
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator().manual_seed(1)
    bound = (6.0 / (a.nfeat + a.E)) ** 0.5
    table = ((torch.rand(a.nfeat, a.E, generator=g) * 2 - 1) * (bound if a.regime == "fresh" else 0.9)).to(dev)
    qf = (torch.randn(a.O, a.E, generator=g) * (0.3 if a.regime == "fresh" else 1.5)).to(dev)
    values = (torch.randn(a.O, a.F, generator=g) * 0.3).to(dev)
    sc = (torch.rand(a.O, generator=g) + 0.5).to(dev)
    sh = torch.randn(a.O, generator=g).to(dev)
    ids = torch.randint(0, a.nfeat, (a.B, a.F), generator=g).to(dev)
    vals = torch.rand(a.B, a.F, generator=g).to(dev)
    out = torch.empty(a.B, a.O, a.E, device=dev)
    bytes_alg = a.B * (a.F * (12 + 4 * a.E) + 4 * a.O * a.E)
    if a.cus:
        import ctypes
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        words = (ctypes.c_uint32 * 8)()
        for b in range(a.cus):
            words[b // 32] |= 1 << (b % 32)
        sp = ctypes.c_void_p()
        assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, words) == 0
        torch.cuda.set_stream(torch.cuda.ExternalStream(sp.value, device=dev))
    for fl in a.flags:
        def run():
            native.fused_fwd(a.B, a.F, a.E, a.O, a.alpha, 50, fl, ids, vals, table, qf, values, sc, sh, out)
        import time
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.15:               # let the device clocks settle (cold: ~20 % slower)
            for _ in range(16):
                run()
            torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.steps
        print((f"CUs {a.cus:3d} " if a.cus else "") + f"flags={fl:#06x} alpha={a.alpha} {a.regime:6s} B={a.B} F={a.F} E={a.E} O={a.O}: {ms * 1e3:8.1f} us  "
              f"{a.B / ms / 1e3:8.1f} Msamp/s  {bytes_alg / ms / 1e6:7.0f} GB/s alg", flush=True)


if __name__ == "__main__":
    main()
