"""The CPU oracle (oracle/armnet_oracle.c) against the golden vectors captured from the real reference.

This is the pin SURVEY.md §8(c) asks for: the reference has no tests of its own, so every stage of the
restatement is checked against tensors hooked out of the reference's own forward."""
import numpy as np
import pytest

from golden_util import load, load_entmax, model_cases
from oracle import armnet_oracle as orc
from tol_util import elem_excess

STAGES = ["vals_clamped", "x_emb", "gates", "p", "arm_weight", "neurons", "x_arm", "logits"]
# measured worst case over all fixtures is ~3e-7 (fp32 summation order inside ATen BLAS); bars below
TOL = dict(vals_clamped=0.0, x_emb=0.0, gates=2e-6, p=2e-6, arm_weight=2e-6, neurons=2e-6, x_arm=5e-6,
           logits=1e-5)


@pytest.mark.parametrize("name", model_cases())
def test_forward_matches_reference(name):
    meta, sd, ids, vals, ref = load(name)
    got = orc.forward(meta["variant"], meta["ctor"], sd, ids, vals, train=meta["train"])
    for k in STAGES:
        if k not in ref:                       # the B = 64 fixtures keep the block's outputs only
            continue
        assert got[k].shape == ref[k].shape, (k, got[k].shape, ref[k].shape)
        err = float(np.max(np.abs(got[k].astype(np.float64) - ref[k]))) if ref[k].size else 0.0
        bar = TOL[k] * max(1.0, float(np.max(np.abs(ref[k])))) if ref[k].size else 0.0
        assert err <= bar, f"{name}: stage {k} max abs err {err:.3e} > {bar:.3e}"
        if k != "logits":                      # every element within 1e-5 of its own magnitude (wide-range tensors)
            assert elem_excess(got[k], ref[k], 1e-5) <= 1.0, f"{name}: stage {k} elementwise"
    if meta["train"]:
        for k in ("arm_bn.running_mean", "arm_bn.running_var"):
            np.testing.assert_allclose(got["after/" + k], ref["after/" + k], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", [n for n in model_cases() if "train" not in n])
def test_fused_block_entry_matches_reference(name):
    """oracle_arm_block (the bench's cpu_baseline entry) == staged path == reference x_arm."""
    meta, sd, ids, vals, ref = load(name)
    v = np.array(vals, dtype=np.float32, copy=True)
    out = orc.arm_block(meta["variant"], ids, v, sd, float(meta["ctor"]["alpha"]))
    np.testing.assert_array_equal(v, ref["vals_clamped"])        # in-place side effect
    want = ref["x_arm"].reshape(out.shape)
    assert np.max(np.abs(out - want)) <= TOL["x_arm"] * max(1.0, float(np.max(np.abs(want))))
    assert elem_excess(out, want, 1e-5) <= 1.0


def test_entmax_alone_matches_reference():
    worst = 0.0
    for m, X, P in load_entmax():
        got = orc.entmax_bisect(X, m["alpha"], m["n_iter"], m.get("ensure_sum_one", True))
        err = float(np.max(np.abs(got - P)))
        # alpha > 2 is ill-conditioned: 1-ulp pow differences move the bisection
        bar = 2e-6 if m["alpha"] <= 2.0 else 2e-5
        assert err <= bar, (m, err)
        worst = max(worst, err)
    assert worst < 2e-5


def test_entmax_with_one_alpha_per_row_matches_reference():
    """round 6: utils/entmax.py:31-36 (tensor alpha) — the oracle's per-row restatement against the reference's own vectors"""
    from golden_util import load_entmax_row_alpha
    for m, X, A, P in load_entmax_row_alpha():
        nd = X.ndim
        dim = m["dim"] % nd
        Xl = np.moveaxis(X, dim, -1)
        Al = np.moveaxis(np.broadcast_to(A, X.shape[:dim] + (1,) + X.shape[dim + 1:]) if A.ndim == nd else
                         np.broadcast_to(A, np.broadcast_shapes(A.shape, tuple(1 if i == dim else n for i, n in enumerate(X.shape)))), dim, -1)
        got = np.moveaxis(orc.entmax_bisect_rows(np.ascontiguousarray(Xl), np.ascontiguousarray(Al), m["n_iter"], m["ensure_sum_one"]), -1, dim)
        hi = np.broadcast_to(np.moveaxis(Al, -1, dim) > 2.0, P.shape)
        err = np.abs(got - P)
        assert float(err[~hi].max(initial=0.0)) <= 2e-6 and float(err[hi].max(initial=0.0)) <= 2e-5, (m, float(err.max()))


def test_entmax_edge_rows():
    one_hot = orc.entmax_bisect(np.array([[5.0, 0.0, -1.0, 0.5]], np.float32), 1.5)
    np.testing.assert_array_equal(one_hot, [[1, 0, 0, 0]])
    uni = orc.entmax_bisect(np.zeros((1, 8), np.float32), 1.7)
    np.testing.assert_allclose(uni, np.full((1, 8), 0.125, np.float32), rtol=3e-7)
    nan = orc.entmax_bisect(np.array([[np.inf, 0.0, 1.0]], np.float32), 1.5)
    assert np.isnan(nan).all()
    d1 = orc.entmax_bisect(np.array([[3.0]], np.float32), 2.0)
    np.testing.assert_array_equal(d1, [[1.0]])


def test_out_of_range_id_raises_indexerror():
    meta, sd, ids, vals, _ = load("g7_odd_1h_f13_e12_h7_a1.5")
    bad = ids.copy()
    bad[0, 0] = sd["embedding.embedding.weight"].shape[0]
    with pytest.raises(IndexError):
        orc.forward("1h", meta["ctor"], sd, bad, vals)


@pytest.mark.parametrize("name", [n for n in model_cases() if "train" not in n])
def test_cpu_twins_of_the_abi_match_the_reference(name):
    """oracle/armnet_cpu_twins.c: armnet_fold_params_f32_cpu + armnet_fused_fwd_f32_cpu (the ABI's arguments minus the
    stream, folded parameters, the reference's bisection) against the golden post-BatchNorm neurons"""
    meta, sd, ids, vals, ref = load(name)
    c = meta["ctor"]
    K = c["nhead"] if meta["variant"] == "mh" else 1
    H, E, D = c["nhid"], c["nemb"], c["d_k"]
    bw = sd["attn_layer.bilinear_w"] if meta["variant"] == "mh" else sd["attn_layer.bilinear_w.weight"]
    qf, sc, sh = orc.twin_fold_params(1 if meta["variant"] == "mh" else 0, K, H, E, D, bw, sd["attn_layer.query"],
                                      sd["arm_bn.weight"], sd["arm_bn.bias"], sd["arm_bn.running_mean"],
                                      sd["arm_bn.running_var"])
    v = vals.copy()
    out, status = orc.twin_fused_fwd(ids, v, sd["embedding.embedding.weight"], qf, sd["attn_layer.values"], sc, sh,
                                     float(c["alpha"]), flags=1)
    assert status == 0
    want = ref["x_arm"].reshape(out.shape)
    assert float(np.max(np.abs(out - want))) <= 1e-5 * max(1.0, float(np.max(np.abs(want))))
    assert elem_excess(out, want, 1e-5) <= 1.0
    np.testing.assert_array_equal(v, ref["vals_clamped"])
