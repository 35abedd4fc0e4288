#!/usr/bin/env python3
"""Backward kernel timing (developer tool, GPU box): armnet_fused_bwd_f32, matrix-core vs generic kernel.
    python tools/bwd_bench.py [--F 39 --E 16 --O 32 --B 65536 --alpha 2.0 --generic]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from armnet_hip import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--alpha", type=float, default=2.0)
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--F", type=int, default=39)
    ap.add_argument("--E", type=int, default=16)
    ap.add_argument("--O", type=int, default=32)
    ap.add_argument("--nfeat", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--generic", action="store_true")
    ap.add_argument("--flags", type=lambda s: int(s, 0), nargs="*", default=[0])
    ap.add_argument("--cus", type=int, default=0, help="launch on a stream restricted to the first n bits of the CU mask")
    a = ap.parse_args()
    dev = "cuda:0"
    if a.cus:
        import ctypes
        hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
        words = (ctypes.c_uint32 * 8)()
        for b in range(a.cus):
            words[b // 32] |= 1 << (b % 32)
        sp = ctypes.c_void_p()
        assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(sp), 8, words) == 0
        torch.cuda.set_stream(torch.cuda.ExternalStream(sp.value, device=dev))
    g = torch.Generator().manual_seed(1)
    table = ((torch.rand(a.nfeat, a.E, generator=g) * 2 - 1) * 0.5).to(dev)
    qf = (torch.randn(a.O, a.E, generator=g) * 0.5).to(dev)
    values = (torch.randn(a.O, a.F, generator=g) * 0.3).to(dev)
    ids = torch.randint(0, a.nfeat, (a.B, a.F), generator=g).to(dev)
    vals = (torch.rand(a.B, a.F, generator=g) * 0.999 + 1e-3).to(dev)
    one, zero = torch.ones(a.O, device=dev), torch.zeros(a.O, device=dev)
    z = torch.empty(a.B, a.O, a.E, device=dev)
    native.fused_fwd(a.B, a.F, a.E, a.O, a.alpha, 50, 0, ids, vals, table, qf, values, one, zero, z)
    dz = torch.randn(a.B, a.O, a.E, device=dev)
    dt, dv, dq = torch.zeros_like(table), torch.zeros_like(values), torch.zeros_like(qf)
    for flags, name in tuple((f, f"mfma {f:#06x}") for f in a.flags) + (((native.F_FORCE_GENERIC, "generic"),) if a.generic else ()):
        run = lambda: native.fused_bwd(a.B, a.F, a.E, a.O, a.alpha, 50, flags, ids, vals, table, qf, values, z, dz, dt, dv, dq)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.steps):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.steps
        print((f"CUs {a.cus:3d} " if a.cus else "") + f"bwd {name:12s} alpha={a.alpha} B={a.B} F={a.F} E={a.E} O={a.O}: {us:9.1f} us  {a.B / us:8.1f} Msamp/s")


if __name__ == "__main__":
    main()
