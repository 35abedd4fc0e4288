#!/usr/bin/env python3
"""round 6 experiment: the block (memory-request bound) and the head (matrix-pipe / LDS bound) of consecutive batches on DISJOINT
sets of CUs (hipExtStreamCreateWithCUMask), pipelined: does the whole forward approach max(block, head) instead of their sum?
    python tools/scratch/r6_cu_partition.py"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
from models.armnet_1h import ARMNetModel  # noqa: E402

DEV = "cuda:0"
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))


def masked_stream(bits):
    """bits: iterable of CU indices (0..255) that the stream may use"""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=DEV)


def timeit(fn, n=20, settle=0.4):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < settle:
        fn()
        torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e6)
    return best


B, F, E, nfeat = 65536, 39, 16, 1_000_000
torch.manual_seed(0)
m = ARMNetModel(F, nfeat, E, 2.0, 32, E, 2, 256, 0.0, False, 2, 256).to(DEV).eval()
m.check_ids = False
g = torch.Generator().manual_seed(0)
NB = 4
ids = [torch.randint(0, nfeat, (B, F), generator=g).to(DEV) for _ in range(NB)]
vals = [torch.rand(B, F, generator=g).to(DEV) for _ in range(NB)]
zbuf = [torch.empty(B, 32, E, device=DEV) for _ in range(NB)]
with torch.no_grad():
    want = m({"id": ids[0], "value": vals[0]}).clone()
    k = [0]

    def block_on(s):
        def f():
            i = k[0] % NB
            k[0] += 1
            with torch.cuda.stream(s):
                m.arm_block(ids[i], vals[i], out=zbuf[i])
        return f

    def head_on(s):
        def f():
            with torch.cuda.stream(s):
                m.mlp(zbuf[0].view(B, -1))
        return f

    def whole_on(s):
        def f():
            i = k[0] % NB
            k[0] += 1
            with torch.cuda.stream(s):
                m({"id": ids[i], "value": vals[i]})
        return f

    full = masked_stream(range(256))
    print(f"all 256 CUs: block {timeit(block_on(full)):7.1f} us  head {timeit(head_on(full)):7.1f} us  whole forward {timeit(whole_on(full)):7.1f} us", flush=True)
    for layout in ("interleaved", "contiguous"):
        for nblk in (64, 96, 128, 160, 192):
            if layout == "interleaved":         # bit i -> XCD i % 8 (if the driver spreads the mask round-robin): the same share of every XCD
                a_bits, b_bits = range(0, nblk), range(nblk, 256)
            else:                               # the other reading: whole XCDs
                a_bits = [i for i in range(256) if (i % 8) < nblk // 32 or ((i % 8) == nblk // 32 and i // 8 < nblk % 32)]
                b_bits = [i for i in range(256) if i not in set(a_bits)]
            sa, sb = masked_stream(a_bits), masked_stream(b_bits)
            tb, th = timeit(block_on(sa)), timeit(head_on(sb))
            ev_z = [torch.cuda.Event() for _ in range(NB)]
            ev_h = [torch.cuda.Event() for _ in range(NB)]
            started = [False]

            def piped():
                i = k[0] % NB
                k[0] += 1
                with torch.cuda.stream(sa):
                    if started[0]:
                        sa.wait_event(ev_h[i])            # the head is done with zbuf[i] (NB batches ago)
                    m.arm_block(ids[i], vals[i], out=zbuf[i])
                    ev_z[i].record(sa)
                with torch.cuda.stream(sb):
                    sb.wait_event(ev_z[i])
                    y = m.mlp(zbuf[i].view(B, -1))
                    ev_h[i].record(sb)
                if i == NB - 1:
                    started[0] = True
                return y

            tp = timeit(piped)
            k[0] = 0
            started[0] = False
            torch.cuda.synchronize()
            y0 = piped().squeeze().clone()
            torch.cuda.synchronize()
            ok = torch.equal(y0, want)
            print(f"{layout:11s} block on {nblk:3d} CUs {tb:7.1f} us | head on {256 - nblk:3d} CUs {th:7.1f} us | pipelined whole forward "
                  f"{tp:7.1f} us/batch = {B / tp:6.1f} M samples/s  bit-equal {ok}", flush=True)
    # the same pipeline without masks (two plain streams): what bench.py's batches_in_flight does
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    ev_z = [torch.cuda.Event() for _ in range(NB)]
    ev_h = [torch.cuda.Event() for _ in range(NB)]
    started = [False]

    def piped2():
        i = k[0] % NB
        k[0] += 1
        with torch.cuda.stream(sa):
            if started[0]:
                sa.wait_event(ev_h[i])
            m.arm_block(ids[i], vals[i], out=zbuf[i])
            ev_z[i].record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(ev_z[i])
            m.mlp(zbuf[i].view(B, -1))
            ev_h[i].record(sb)
        if i == NB - 1:
            started[0] = True

    tp = timeit(piped2)
    print(f"no masks, two streams: pipelined whole forward {tp:7.1f} us/batch = {B / tp:6.1f} M samples/s")
