"""The module surface on HOST tensors (round-4 verdict, missing 3 / next 6).

The reference's model runs wherever its tensors are (`models/model_utils.py:86`).  A model of this build that was never
moved to the GPU, called with CPU tensors, runs the reference's ATen op chain (`armnet_hip/host_ops.py`) — a dispatch on
the tensors' device, never a fallback for device tensors and never the oracle.  Held here to the fixtures the real
reference produced (`tests/golden/make_golden.py`): the same ops on the same torch build, so the eval fixtures are held to
1e-6 (measured: bit-equal blocks), the training steps to the gradient bars of the GPU suite."""
import glob
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, grad_cases, load, load_entmax, model_cases
from model_util import build_model

EVAL_CASES = [n for n in model_cases() if "train" not in n]
SIB_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "s[12]_*.npz")))
SIB_GRAD = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "s3_grad_*.npz")))


def _x(ids, vals):
    return {"id": torch.from_numpy(ids), "value": torch.from_numpy(vals.copy())}


def _close(got, want, rtol):
    want = np.asarray(want, dtype=np.float64)
    got = np.asarray(got, dtype=np.float64).reshape(want.shape)
    return float(np.max(np.abs(got - want))) <= rtol * max(1.0, float(np.abs(want).max()))


@pytest.mark.parametrize("name", EVAL_CASES)
def test_host_model_matches_the_reference_fixture(name):
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd)
    x = _x(ids, vals)
    with torch.no_grad():
        block = m.arm_block(x["id"], x["value"].clone())
        y = m(x)
    assert _close(block.numpy(), ref["x_arm"], 1e-6), "block"
    assert tuple(y.shape) == tuple(ref["logits"].shape)                       # 0-dim at B = 1 (armnet.py:101)
    # logits: 1e-6 of the head's input scale (the wide fixtures sum 4e3-sized terms into an O(1) logit)
    assert float(np.max(np.abs(y.numpy() - ref["logits"]))) <= 1e-6 * max(1.0, float(np.abs(ref["x_arm"]).max()))
    np.testing.assert_array_equal(x["value"].numpy(), ref["vals_clamped"])    # the in-place clamp_ side effect


def test_host_train_mode_uses_batch_statistics_and_updates_running_stats():
    meta, sd, ids, vals, ref = load("g8_train_1h_a1.7_stress")
    m = build_model(meta, sd).train()
    with torch.no_grad():
        y = m(_x(ids, vals))
    assert _close(y.numpy(), ref["logits"], 1e-5)
    np.testing.assert_allclose(m.arm_bn.running_mean.numpy(), ref["after/arm_bn.running_mean"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(m.arm_bn.running_var.numpy(), ref["after/arm_bn.running_var"], rtol=1e-6, atol=1e-7)


def _training_step(m, meta, ids, vals, ref, zero_bar):
    m.train(meta["train"])
    x = _x(ids, vals)
    logits = m(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, torch.from_numpy(ref["target"]))
    loss.backward()
    assert _close(logits.detach().numpy(), ref["logits"], 2e-5), "logits"
    assert abs(float(loss.detach()) - float(ref["loss"])) <= 2e-6
    np.testing.assert_array_equal(x["value"].numpy(), ref["vals_clamped"])
    gmax = max(float(np.abs(ref["grad/" + k]).max()) for k, _ in m.named_parameters())
    for k, p in m.named_parameters():
        g = ref["grad/" + k]
        if float(np.abs(g).max()) < zero_bar * gmax:
            continue          # analytically zero (a bias in front of a train-mode BatchNorm): rounding noise on both sides
        assert p.grad is not None, k
        err = float(np.max(np.abs(p.grad.numpy().astype(np.float64) - g))) / max(float(np.abs(g).max()), 1e-12)
        assert err <= 5e-5, f"grad of {k}: rel err {err:.2e}"


@pytest.mark.parametrize("name", grad_cases())
def test_host_training_step_matches_reference_gradients(name):
    """train.py:108-113 on host tensors: autograd through the op chain, the sparse map's backward = entmax.py:70-80"""
    meta, sd, ids, vals, ref = load(name)
    _training_step(build_model(meta, sd), meta, ids, vals, ref, 1e-6)


def _build_sibling(meta, sd):
    import test_siblings
    return test_siblings._build(meta, sd)


@pytest.mark.parametrize("name", SIB_CASES)
def test_host_siblings_match_the_reference_fixture(name):
    """GC-ARM / AFN (gc_arm.py:82-105, afn.py:49-77) from the modules' own sub-modules on host tensors"""
    meta, sd, ids, vals, ref = load(name)
    m = _build_sibling(meta, sd)
    x = _x(ids, vals)
    with torch.no_grad():
        y = m(x)
    assert tuple(y.shape) == tuple(ref["logits"].shape)
    assert float(np.max(np.abs(y.numpy() - ref["logits"]))) <= 1e-6 * max(1.0, float(np.abs(ref["x_arm"]).max()))
    np.testing.assert_array_equal(x["value"].numpy(), ref["vals_clamped"])
    if "table_after" in ref:                                                  # afn.py:74-77 embedding_clip, in place
        np.testing.assert_array_equal(m.embedding.embedding.weight.detach().numpy(), ref["table_after"])


@pytest.mark.parametrize("name", SIB_GRAD)
def test_host_sibling_training_step_matches_reference_gradients(name):
    meta, sd, ids, vals, ref = load(name)
    _training_step(_build_sibling(meta, sd), meta, ids, vals, ref, 1e-5)


def test_host_entmax_is_the_reference_bisection():
    from utils.entmax import EntmaxBisect, entmax_bisect
    for m, X, P in load_entmax():
        got = entmax_bisect(torch.from_numpy(X), alpha=m["alpha"], dim=-1, n_iter=m["n_iter"],
                            ensure_sum_one=m.get("ensure_sum_one", True)).numpy()
        assert float(np.max(np.abs(got - P))) <= 1e-7, m
    z = np.load(os.path.join(GOLDEN, "g6_entmax_grad.npz"))
    keys = sorted(k[2:] for k in z.files if k.startswith("X/"))
    assert len(keys) == 9
    for k in keys:
        X = torch.from_numpy(z["X/" + k]).requires_grad_(True)
        if k == "dim1":
            Y = entmax_bisect(X, alpha=1.5, dim=1)
        else:
            Y = EntmaxBisect(alpha=float(k.split("_")[0][1:]), dim=-1)(X)
        Y.backward(torch.from_numpy(z["dY/" + k]))
        assert float((Y.detach() - torch.from_numpy(z["Y/" + k])).abs().max()) <= 1e-7, k
        ref = torch.from_numpy(z["dX/" + k])
        assert float((X.grad - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), k


def test_host_submodule_surfaces():
    """models.layers.Embedding / MLP and the attention sub-module on host tensors (layers.py:15-21,68-88)"""
    from models.layers import MLP, Embedding
    torch.manual_seed(0)
    emb = Embedding(50, 6)
    ids = torch.randint(0, 50, (4, 3))
    vals = torch.rand(4, 3)
    want = emb.embedding.weight.detach()[ids] * vals.unsqueeze(2)
    got = emb({"id": ids, "value": vals})
    np.testing.assert_array_equal(got.detach().numpy(), want.numpy())
    got.sum().backward()                                                      # dense table gradient, like nn.Embedding
    assert emb.embedding.weight.grad is not None and emb.embedding.weight.grad.shape == (50, 6)
    with pytest.raises(IndexError):
        emb({"id": torch.full((1, 3), 50), "value": torch.ones(1, 3)})
    head = MLP(18, 2, 8, 0.0).eval()
    with torch.no_grad():
        assert head(got.detach().view(4, -1)).shape == (4, 1)
    meta, sd, ids, vals, ref = load("g2_criteo_1h_a1.7_stress")
    m = build_model(meta, sd)
    with torch.no_grad():
        w = m.attn_layer(torch.from_numpy(ref["x_emb"]))
    assert _close(w.numpy(), ref["arm_weight"], 1e-6)


def test_out_of_range_id_raises_index_error_on_the_host_too():
    meta, sd, ids, vals, _ = load("g7_odd_1h_f13_e12_h7_a1.5")
    m = build_model(meta, sd)
    bad = ids.copy()
    bad[0, 0] = meta["ctor"]["nfeat"]
    with pytest.raises(IndexError):
        m(_x(bad, vals))


def test_nothing_under_the_package_imports_the_oracle():
    """the host branch is ATen ops: `grep -rn oracle arm-net_amd/` stays empty (import lines)"""
    pkg = os.path.join(os.path.dirname(GOLDEN), "..", "arm-net_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                for line in open(os.path.join(root, f)):
                    s = line.strip()
                    if s.startswith(("import ", "from ")):
                        assert "oracle" not in s, (f, s)


@pytest.mark.gpu
def test_mixed_placement_raises_instead_of_computing_elsewhere():
    """a device model with a host batch (or the reverse) is the caller's error: nothing is copied or computed on the
    other side behind their back"""
    from armnet_hip import native
    meta, sd, ids, vals, _ = load("g7_odd_1h_f13_e12_h7_a1.5")
    m = build_model(meta, sd, "cuda:0")
    with pytest.raises(native.ArmnetNativeError, match="no CPU fallback"):
        m(_x(ids, vals))
    m_host = build_model(meta, sd)
    x = _x(ids, vals)
    with pytest.raises((native.ArmnetNativeError, RuntimeError)):
        m_host({"id": x["id"].cuda(), "value": x["value"].cuda()})


def test_entmax_with_a_tensor_alpha_on_host_tensors_is_the_reference_bit_for_bit():
    """utils/entmax.py:31-36 (one alpha per row) on host tensors: the reference's ATen ops in the reference's order"""
    import numpy as np
    import torch
    from golden_util import load_entmax_row_alpha
    from utils.entmax import entmax_bisect
    for m, X, A, P in load_entmax_row_alpha():
        with torch.no_grad():
            got = entmax_bisect(torch.from_numpy(X), alpha=torch.from_numpy(A), dim=m["dim"], n_iter=m["n_iter"],
                                ensure_sum_one=m["ensure_sum_one"]).numpy()
        np.testing.assert_array_equal(got, P, err_msg=str(m))


def test_entmax_gradients_with_a_tensor_alpha_on_host_tensors_are_the_reference_bit_for_bit():
    """utils/entmax.py:70-98: dX and d alpha (per-row, partially broadcast and 0-dim alpha) for a random dY"""
    import numpy as np
    import torch
    from golden_util import load_entmax_row_alpha_grads
    from utils.entmax import entmax_bisect
    for m, X, A, dY, dX, dA in load_entmax_row_alpha_grads():
        Xt, At = torch.from_numpy(X).requires_grad_(True), torch.from_numpy(A).requires_grad_(True)
        (entmax_bisect(Xt, alpha=At, dim=m["dim"], n_iter=m["n_iter"], ensure_sum_one=m["ensure_sum_one"]) * torch.from_numpy(dY)).sum().backward()
        np.testing.assert_array_equal(Xt.grad.numpy(), dX, err_msg=str(m))
        np.testing.assert_array_equal(At.grad.numpy(), dA, err_msg=str(m))
