import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "arm-net_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from armnet_hip import native
from test_hip_f16_contractions import _case, _run, DEV
for (F, E, O, alpha) in ((30, 16, 96, 1.0), (39, 16, 128, 2.0), (43, 10, 64, 1.7), (30, 16, 96, 2.0), (30, 16, 128, 1.0), (29, 16, 96, 1.5)):
    table, qf, values, sc, sh, ids, vals = _case(F, E, O, 7 * F + O, B=20011)
    B = ids.shape[0]
    z64, _ = _run(B, F, E, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
    z64b, _ = _run(B, F, E, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
    z32, _ = _run(B, F, E, O, alpha, 0, ids.to(torch.int32), vals, table, qf, values, sc, sh)
    zf, _ = _run(B, F, E, O, alpha, native.F_FP32_CONTRACTIONS, ids, vals, table, qf, values, sc, sh)
    rows = table[ids].contiguous(); v = vals.clone(); zr = torch.empty(B, O, E, device=DEV)
    native.fused_fwd_from_rows(B, F, E, O, alpha, 50, native.F_WRITE_CLAMPED_VALS, rows, v, qf, values, sc, sh, zr)
    den = zf.abs().clamp(min=1.0)
    print(F, E, O, alpha, "run-to-run equal", torch.equal(z64, z64b), "| i64 vs i32", float(((z64 - z32).abs() / den).max()), int((z64 != z32).sum()),
          "| i64 vs rows", float(((z64 - zr).abs() / den).max()), int((z64 != zr).sum()), "| vs fp32 form: i64", float(((z64 - zf).abs() / den).max()),
          "i32", float(((z32 - zf).abs() / den).max()), "rows", float(((zr - zf).abs() / den).max()))
