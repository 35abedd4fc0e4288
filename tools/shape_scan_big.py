#!/usr/bin/env python3
"""Large-batch scan (developer tool, GPU box): persistent-loop / pipeline paths of the matrix-core kernels (several
groups per wave, short last group, int32 ids, pre-gathered rows, value write-back) against the shape-agnostic kernels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native

DEV = "cuda:0"
nfeat = 5003
bad = n = 0
wide = "--wide" in sys.argv                 # round 4: the nemb 65..128 family (smaller batches: the generic kernel is slow there)
for B in ((4099, 20011) if wide else (20011, 65537)):
    for F in (1, 3, 7, 10, 13, 16, 22, 24, 31, 39, 43, 48):
        for E in ((65, 100, 128) if wide else (5, 10, 16, 20, 32, 64)):
            for O, alpha in ((7, 2.0), (32, 1.5), (40, 1.0), (24, 1.7)):
                if native.fused_kernel_kind(F, E, O, alpha) != 1:
                    continue
                g = torch.Generator().manual_seed(F * 1000 + E * 10 + O)
                table = (torch.rand(nfeat, E, generator=g) * 1.6 - 0.8).to(DEV)
                qf = (torch.randn(O, E, generator=g) * 0.8).to(DEV)
                values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
                ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
                vals0 = (torch.rand(B, F, generator=g) * 1.2 - 0.1).to(DEV)          # some outside [1e-3, 1]
                sc, sh = (torch.rand(O, generator=g) + 0.5).to(DEV), torch.randn(O, generator=g).to(DEV)
                outs = {}
                for name, flags, idt in (("gen", native.F_FORCE_GENERIC | native.F_WRITE_CLAMPED_VALS, ids),
                                         ("mfma", native.F_WRITE_CLAMPED_VALS, ids),
                                         ("mfma32", native.F_WRITE_CLAMPED_VALS, ids.to(torch.int32))):
                    v = vals0.clone()
                    z = torch.empty(B, O, E, device=DEV)
                    native.fused_fwd(B, F, E, O, alpha, 50, flags, idt, v, table, qf, values, sc, sh, z)
                    outs[name] = (z, v)
                rows = table[ids].contiguous()
                v = vals0.clone(); z = torch.empty(B, O, E, device=DEV)
                native.fused_fwd_from_rows(B, F, E, O, alpha, 50, native.F_WRITE_CLAMPED_VALS, rows, v, qf, values, sc, sh, z)
                outs["rows"] = (z, v)
                zg, vg = outs["gen"]
                n += 1
                for k in ("mfma", "mfma32", "rows"):
                    z, v = outs[k]
                    err = float((z - zg).abs().max()) / max(1.0, float(zg.abs().max()))
                    if not (err <= 1e-5) or not torch.equal(v, vg):
                        bad += 1
                        print(f"{k} mismatch B={B} F={F} E={E} O={O} alpha={alpha}: {err} vals_equal={torch.equal(v, vg)}")
print(f"{n} cases, {bad} disagreements")

# backward, large batch: pipelined loads past the end, several samples per wave, BatchNorm coefficients folded in
bad = n = 0
for B in (20011,):
    for F, E, O, alpha in (((10, 100, 10, 1.7), (22, 72, 40, 2.0), (32, 128, 16, 1.5), (3, 65, 33, 1.0)) if wide else
                           ((39, 16, 32, 2.0), (39, 10, 128, 1.5), (22, 32, 40, 2.0), (43, 64, 24, 1.7), (10, 10, 70, 1.0),
                            (3, 5, 7, 2.0), (48, 20, 33, 2.0), (13, 48, 16, 1.5))):
        g = torch.Generator().manual_seed(F * 1000 + E * 10 + O)
        table = (torch.rand(nfeat, E, generator=g) * 1.6 - 0.8).to(DEV)
        qf = (torch.randn(O, E, generator=g) * 0.5).to(DEV)
        values = (torch.randn(O, F, generator=g) * 0.3).to(DEV)
        ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
        vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
        one, zero = torch.ones(O, device=DEV), torch.zeros(O, device=DEV)
        z = torch.empty(B, O, E, device=DEV)
        native.fused_fwd(B, F, E, O, alpha, 50, native.F_FORCE_GENERIC, ids, vals, table, qf, values, one, zero, z)
        dy = torch.randn(B, O, E, generator=g).to(DEV)
        cA, cB, cC = (torch.rand(O, generator=g) + 0.5).to(DEV), (torch.randn(O, generator=g) * 0.1).to(DEV), (torch.randn(O, generator=g) * 0.1).to(DEV)
        outs = []
        for flags in (native.F_FORCE_GENERIC, 0):
            dt, dv, dq = torch.zeros_like(table), torch.zeros_like(values), torch.zeros_like(qf)
            native.fused_bwd_bn(B, F, E, O, alpha, 50, flags, ids, vals, table, qf, values, z, dy, cA, cB, cC, dt, dv, dq)
            outs.append((dt, dv, dq))
        n += 1
        for nm, a, b in zip(("d_table", "d_values", "d_qfold"), outs[1], outs[0]):
            e2 = float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)
            if not (e2 <= 1e-4):
                bad += 1
                print(f"BWD mismatch B={B} F={F} E={E} O={O} alpha={alpha} {nm}: {e2}")
print(f"backward: {n} cases, {bad} disagreements")
