#!/usr/bin/env python3
"""Exhaustive shape scan (developer tool, GPU box): the matrix-core forward and backward kernels against the
shape-agnostic ones over nfield x nemb x neurons x alpha, small batch.  Prints every disagreement."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native

DEV = "cuda:0"
bwd = "--bwd" in sys.argv
B, nfeat = 37, 53
bad = n = 0
for F in range(1, 49):
    wide = "--wide" in sys.argv            # round 4: the nemb 65..128 family only
    for E in ((65, 66, 67, 72, 80, 96, 97, 100, 120, 127, 128) if wide else
              (4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 20, 31, 32, 33, 48, 63, 64)):
        for O in (1, 7, 16, 24, 32, 40, 70):
            for alpha in (1.0, 1.5, 1.7, 2.0) + ((2.5,) if (F + E + O) % 5 == 0 else ()):
                if native.fused_kernel_kind(F, E, O, alpha) != 1:
                    continue
                g = torch.Generator().manual_seed(F * 1000 + E * 10 + O)
                table = (torch.rand(nfeat, E, generator=g) * 1.6 - 0.8).to(DEV)
                qf = (torch.randn(O, E, generator=g) * 0.8).to(DEV)
                values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
                ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
                vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
                sc, sh = (torch.rand(O, generator=g) + 0.5).to(DEV), torch.randn(O, generator=g).to(DEV)
                z, zg = torch.empty(B, O, E, device=DEV), torch.empty(B, O, E, device=DEV)
                native.fused_fwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, sc, sh, z)
                native.fused_fwd(B, F, E, O, alpha, 50, native.F_FORCE_GENERIC, ids, vals, table, qf, values, sc, sh, zg)
                err = float((z - zg).abs().max()) / max(1.0, float(zg.abs().max()))
                n += 1
                if not (err <= 1e-5):
                    bad += 1
                    print(f"FWD mismatch F={F} E={E} O={O} alpha={alpha}: {err}")
                if bwd and (F + E + O) % 3 == 0:
                    dz = torch.randn(B, O, E, generator=g).to(DEV)
                    outs = []
                    for flags in (native.F_FORCE_GENERIC, 0):
                        dt, dv, dq = torch.zeros_like(table), torch.zeros_like(values), torch.zeros_like(qf)
                        try:
                            native.fused_bwd(B, F, E, O, alpha, 50, flags, ids, vals, table, qf, values, zg, dz, dt, dv, dq)
                        except native.ArmnetNativeError as e:
                            bad += 1
                            print(f"BWD refused F={F} E={E} O={O} alpha={alpha} flags={flags}: {e}")
                        outs.append((dt, dv, dq))
                    for nm, a, b in zip(("d_table", "d_values", "d_qfold"), outs[1], outs[0]):
                        e2 = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
                        # alpha > 2: gppr = p^(2 - alpha) grows without bound as p -> 0+, so the ~2e-7 difference between
                        # the kernels' pow forms (libm powf vs exp2/log2) is amplified; the reference-gradient fixtures pin it
                        if not (e2 <= (5e-3 if alpha > 2.0 else 5e-5)):
                            bad += 1
                            print(f"BWD mismatch F={F} E={E} O={O} alpha={alpha} {nm}: {e2}")
print(f"{n} shapes scanned, {bad} disagreements")
