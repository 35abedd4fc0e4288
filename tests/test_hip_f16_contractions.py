"""Wide ARM blocks (nemb <= 16; 33+ fields from 33 neurons per launch, 29-32 fields from 64, 17-28 from 256) run both contractions of the block
(models/armnet_1h.py:33-36, models/armnet.py:33-36,86-89) as fp16 x 2 operand splits on the 16-bit matrix pipe (F16 in
csrc/fused_mfma_kernel.h): three products per tile, fp32 accumulate, exact power-of-two scales per sample (embeddings) and per
parameter slice (q_fold, values).  Held here to the CPU oracle at the tests' bar, to the fp32-MFMA form of the same kernel
(ARMNET_F_FP32_CONTRACTIONS) at 5e-6 (half the parity bar), bit-equal across the id sources, and to the fp32 form's behaviour on extreme magnitudes and
non-finite embeddings."""
import numpy as np
import pytest
import torch

from oracle import armnet_oracle as orc
from tol_util import TOL, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

# (nfield, nemb, neurons, alpha): every F16 tile family (17-28 / 29-32 / 33-40 / 41-48 fields), odd nemb, several neuron slices,
# every sparse map it serves (softmax, alpha = 1.5, 2, generic)
SHAPES = [(39, 16, 128, 2.0), (39, 16, 64, 1.7), (39, 10, 256, 1.5), (39, 16, 70, 1.0), (33, 7, 96, 2.0), (40, 16, 128, 1.3),
          (43, 16, 128, 2.0), (48, 12, 64, 1.7), (30, 16, 128, 2.0), (29, 5, 64, 1.5), (22, 16, 512, 2.0), (17, 10, 256, 1.7),
          (39, 16, 300, 2.0), (39, 4, 64, 1.9), (39, 16, 40, 2.0), (43, 10, 48, 1.7), (36, 16, 33, 1.5)]


def _case(F, E, O, seed, B=777, nfeat=5003, table_scale=0.9, q_scale=1.5):
    g = torch.Generator().manual_seed(seed)
    table = ((torch.rand(nfeat, E, generator=g) * 2 - 1) * table_scale).to(DEV)
    qf = (torch.randn(O, E, generator=g) * q_scale).to(DEV)
    values = (torch.randn(O, F, generator=g) * 0.3).to(DEV)
    sc, sh = (torch.rand(O, generator=g) + 0.5).to(DEV), torch.randn(O, generator=g).to(DEV)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 1.2 - 0.1).to(DEV)              # some outside [1e-3, 1]: the clamp is live
    return table, qf, values, sc, sh, ids, vals


def _run(B, F, E, O, alpha, flags, ids, vals, table, qf, values, sc, sh):
    from armnet_hip import native
    v = vals.clone()
    z = torch.empty(B, O, E, device=DEV)
    native.fused_fwd(B, F, E, O, alpha, 50, flags | native.F_WRITE_CLAMPED_VALS, ids, v, table, qf, values, sc, sh, z)
    return z, v


@pytest.mark.parametrize("F,E,O,alpha", SHAPES)
def test_split_contractions_match_the_oracle_and_the_fp32_form(F, E, O, alpha):
    from armnet_hip import native
    table, qf, values, sc, sh, ids, vals = _case(F, E, O, F * 1000 + E * 10 + O)
    B = ids.shape[0]
    z16, v16 = _run(B, F, E, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
    z32, v32 = _run(B, F, E, O, alpha, native.F_FP32_CONTRACTIONS, ids, vals, table, qf, values, sc, sh)
    vv = vals.cpu().numpy().copy()
    want, status = orc.twin_fused_fwd(ids.cpu().numpy(), vv, table.cpu().numpy(), qf.cpu().numpy(), values.cpu().numpy(),
                                      sc.cpu().numpy(), sh.cpu().numpy(), alpha, flags=1)
    assert status == 0
    assert_close(z16.cpu().numpy(), want, TOL, "fp16 x 2 contractions vs oracle")
    assert_close(z32.cpu().numpy(), want, TOL, "fp32 contractions vs oracle")
    np.testing.assert_array_equal(v16.cpu().numpy(), vv)
    assert torch.equal(v16, v32)
    rel = float(((z16 - z32).abs() / z32.abs().clamp(min=1.0)).max())
    assert rel <= 5e-6, rel
    assert not torch.equal(z16, z32), "the flag selects a different kernel: identical bits mean it did not"


@pytest.mark.parametrize("F,E,O,alpha", [(39, 16, 128, 2.0), (43, 10, 64, 1.7), (30, 16, 96, 1.0)])
def test_split_contractions_are_bit_equal_across_id_sources(F, E, O, alpha):
    """int64 ids, int32 ids and pre-gathered rows stage the same tile: the same bits out"""
    from armnet_hip import native
    table, qf, values, sc, sh, ids, vals = _case(F, E, O, 7 * F + O, B=20011)
    B = ids.shape[0]
    z64, _ = _run(B, F, E, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
    z32, _ = _run(B, F, E, O, alpha, 0, ids.to(torch.int32), vals, table, qf, values, sc, sh)
    rows = table[ids].contiguous()
    v = vals.clone()
    zr = torch.empty(B, O, E, device=DEV)
    native.fused_fwd_from_rows(B, F, E, O, alpha, 50, native.F_WRITE_CLAMPED_VALS, rows, v, qf, values, sc, sh, zr)
    assert torch.equal(z64, z32) and torch.equal(z64, zr)


def _block_f64_sparsemax(ids, vals, table, qf, values, sc, sh):
    """the block at alpha = 2 in float64 (exact sparsemax by sorting: entmax.py's bisection converges to it)"""
    D = torch.float64
    x = table.to(D)[ids] * vals.clamp(1e-3, 1.0).to(D)[..., None]
    g = torch.einsum("bfe,oe->bof", x, qf.to(D))
    srt = torch.sort(g, dim=-1, descending=True).values
    k = torch.arange(1, g.shape[-1] + 1, device=g.device, dtype=D)
    cs = srt.cumsum(-1) - 1.0
    supp = (srt * k > cs).sum(-1, keepdim=True)
    tau = cs.gather(-1, supp - 1) / supp.to(D)
    p = (g - tau).clamp(min=0)
    z = torch.einsum("bof,bfe->boe", p * values.to(D)[None], x)
    return torch.exp(z) * sc.to(D)[None, :, None] + sh.to(D)[None, :, None]


@pytest.mark.parametrize("F,O", [(22, 512), (30, 96), (32, 128), (36, 40), (39, 128), (43, 64), (48, 96)])
def test_split_contractions_are_run_to_run_deterministic(F, O):
    """a missing wait state between matrix instructions shows up as values that depend on how fast a wave issues (round 6: a build with
    accumulator chains failed exactly this at 29-32 fields with the softmax): every solver instantiation, three launches each"""
    for alpha in (1.0, 1.5, 1.7, 2.0):
        table, qf, values, sc, sh, ids, vals = _case(F, 16, O, 5 * F + O + int(alpha * 10), B=20011)
        B = ids.shape[0]
        z0, _ = _run(B, F, 16, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
        for _rep in range(2):
            z, _ = _run(B, F, 16, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
            assert torch.equal(z, z0), (F, O, alpha)


@pytest.mark.parametrize("table_scale,q_scale", [(1e-12, 1.5), (2.0, 1e-3), (0.9, 1e-20), (1e-30, 1e6), (1.5, 6.0), (1.5, 40.0)])
def test_scales_follow_the_magnitudes(table_scale, q_scale):
    """the per-sample / per-slice power-of-two scales keep the operands inside fp16's range whatever the parameters' magnitude:
    against the block in float64 the split form is as close as the fp32 form (which needs no scale) — large gates cost both the
    same conditioning"""
    from armnet_hip import native
    F, E, O, alpha = 39, 16, 128, 2.0
    table, qf, values, sc, sh, ids, vals = _case(F, E, O, 11, table_scale=table_scale, q_scale=q_scale)
    B = ids.shape[0]
    z16, _ = _run(B, F, E, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
    z32, _ = _run(B, F, E, O, alpha, native.F_FP32_CONTRACTIONS, ids, vals, table, qf, values, sc, sh)
    ref = _block_f64_sparsemax(ids, vals, table, qf, values, sc, sh)
    assert torch.isfinite(z16).all() and torch.isfinite(z32).all()
    den = ref.abs().clamp(min=1.0)
    e16, e32 = float(((z16.double() - ref).abs() / den).max()), float(((z32.double() - ref).abs() / den).max())
    print(f"table {table_scale:g} q {q_scale:g}: split form {e16:.2e}, fp32 form {e32:.2e} of max(1, |ref|)")
    assert e16 <= max(1e-5, 2.0 * e32), (e16, e32)
    assert e16 <= 2.0 * e32 + 1e-6, (e16, e32)


def test_a_non_finite_embedding_poisons_its_sample_only():
    from armnet_hip import native
    F, E, O, alpha = 39, 16, 128, 2.0
    table, qf, values, sc, sh, ids, vals = _case(F, E, O, 3, B=300)
    B = ids.shape[0]
    table[ids[5, 7], 3] = float("nan")
    table[ids[9, 0], 0] = float("inf")
    hit = (ids == ids[5, 7]).any(1) | (ids == ids[9, 0]).any(1)          # every sample that looks either row up
    z16, _ = _run(B, F, E, O, alpha, 0, ids, vals, table, qf, values, sc, sh)
    z32, _ = _run(B, F, E, O, alpha, native.F_FP32_CONTRACTIONS, ids, vals, table, qf, values, sc, sh)
    assert not torch.isfinite(z16[hit]).all(dim=(1, 2)).any(), "a poisoned sample came out finite"
    assert torch.isfinite(z16[~hit]).all() and torch.isfinite(z32[~hit]).all()
    rel = float(((z16 - z32)[~hit].abs() / z32[~hit].abs().clamp(min=1.0)).max())
    assert rel <= 5e-6, rel


def test_module_wide_block_runs_the_split_form_and_matches_the_reference_fixture():
    """configs[2]'s fixture (4 heads x 32 neurons, 39 fields): the module's default path is the split form"""
    from golden_util import load
    from model_util import build_model
    meta, sd, ids, vals, ref = load("g3_criteo_mh4_a2.0_stress")
    m = build_model(meta, sd, DEV)
    with torch.no_grad():
        i = torch.from_numpy(ids).to(DEV)
        got = m.arm_block(i, torch.from_numpy(vals.copy()).to(DEV))
        from armnet_hip import native
        m.kernel_flags = native.F_FP32_CONTRACTIONS
        got32 = m.arm_block(i, torch.from_numpy(vals.copy()).to(DEV))
    want = ref["x_arm"].reshape(got.shape)
    assert_close(got.cpu().numpy(), want, TOL, "split form vs the reference")
    assert_close(got32.cpu().numpy(), want, TOL, "fp32 form vs the reference")
    assert not torch.equal(got, got32)
