#!/usr/bin/env python3
"""round 6: run-to-run determinism of the fp16 x 2 form over every tile family x solver instantiation x id source (a wait-state hazard between
matrix instructions shows up as values that depend on how fast a wave issues)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native
from test_hip_f16_contractions import _case, _run, DEV
n = bad = 0
for (F, O) in ((17, 256), (22, 512), (28, 256), (29, 64), (30, 96), (32, 128), (33, 40), (36, 64), (39, 128), (40, 256), (41, 64), (43, 128), (48, 96)):
    for E in (16, 10):
        for alpha in (1.0, 1.5, 1.7, 2.0):
            for B in (20011, 65536):
                table, qf, values, sc, sh, ids, vals = _case(F, E, O, 7 * F + O + int(alpha * 10), B=B, nfeat=50021)
                zf, _ = _run(B, F, E, O, alpha, native.F_FP32_CONTRACTIONS, ids, vals, table, qf, values, sc, sh)
                den = zf.abs().clamp(min=1.0)
                ref = None
                for rep in range(4):
                    for src in ("i64", "i32", "rows"):
                        if src == "rows":
                            rows = table[ids].contiguous(); v = vals.clone(); z = torch.empty(B, O, E, device=DEV)
                            native.fused_fwd_from_rows(B, F, E, O, alpha, 50, native.F_WRITE_CLAMPED_VALS, rows, v, qf, values, sc, sh, z)
                        else:
                            z, _ = _run(B, F, E, O, alpha, 0, ids if src == "i64" else ids.to(torch.int32), vals, table, qf, values, sc, sh)
                        n += 1
                        if ref is None:
                            ref = z
                            err = float(((z - zf).abs() / den).max())
                            if not err <= 5e-6:
                                bad += 1; print("vs fp32 form", F, E, O, alpha, B, err, flush=True)
                        elif not torch.equal(z, ref):
                            bad += 1
                            print("NOT bit-equal", F, E, O, alpha, B, src, rep, float(((z - ref).abs() / den).max()), int((z != ref).sum()), flush=True)
print(f"{n} launches, {bad} problems")
