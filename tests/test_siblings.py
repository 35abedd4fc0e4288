"""GC-ARM and AFN on the fused kernels: golden vectors captured from the reference's
models/gc_arm.py and models/afn.py (tests/golden/s1_*, s2_*), the CPU oracle's restatement held to them, and the HIP
path (armnet_gc_fused_fwd_f32 / armnet_afn_fused_fwd_f32 + the HIP prediction head) held to both."""
import glob
import os

import numpy as np
import pytest
import torch

from golden_util import GOLDEN, load
from oracle import armnet_oracle as orc
from tol_util import TOL, assert_close, elem_excess, logit_excess, logit_term_scale, rel_err

DEV = "cuda:0"
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "s[12]_*.npz")))


def _build(meta, sd=None, device=None):
    c = meta["ctor"]
    if meta["variant"] == "gc":
        from models.gc_arm import GC_ARMModel
        m = GC_ARMModel(c["nfield"], c["nfeat"], c["nemb"], c["nhead"], c["alpha"], c["nhid"], c["mlp_nlayer"],
                        c["mlp_nhid"], c["dropout"], c["ensemble"], c["deep_nlayer"], c["deep_nhid"])
    else:
        from models.afn import AFNModel
        m = AFNModel(c["nfield"], c["nfeat"], c["nemb"], c["nhid"], c["mlp_nlayer"], c["mlp_nhid"], c["dropout"],
                     c["ensemble"], c["deep_nlayer"], c["deep_nhid"])
    if sd is not None:
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    m.eval()
    return m.to(device) if device is not None else m


def _term_scale(sd, ids, ref):
    """magnitude of the terms each logit is summed from (head on x_arm, + the ensemble branch where the model has one)"""
    x_deep = None
    if "ensemble_layer.weight" in sd:
        x_deep = sd["deep_embedding.embedding.weight"][ids] * ref["vals_clamped"][..., None]
    return logit_term_scale(sd, ref["x_arm"], x_deep)


def _oracle(meta, sd, ids, vals):
    return (orc.forward_gc_arm if meta["variant"] == "gc" else orc.forward_afn)(meta["ctor"], sd, ids, vals)


def test_fixture_families_are_present():
    assert sum(n.startswith("s1_") for n in CASES) >= 5 and sum(n.startswith("s2_") for n in CASES) >= 4


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name):
    meta, sd, ids, vals, ref = load(name)
    got = _oracle(meta, sd, ids, vals)
    np.testing.assert_array_equal(got["vals_clamped"], ref["vals_clamped"])
    assert_close(got["x_arm"], ref["x_arm"], TOL, name + " block")
    # logits: data-derived per-sample bar (AFN's exp(Linear(log x)) reaches 1e2..1e3): tol_util.logit_excess
    assert logit_excess(got["logits"], ref["logits"], _term_scale(sd, ids, ref)) <= 1.0, name
    if "table_after" in ref:
        np.testing.assert_array_equal(got["table_after"], ref["table_after"])


@pytest.mark.parametrize("name", CASES)
def test_product_module_has_the_reference_state_dict(name):
    """same keys in the same order as the reference module that produced the fixture (strict load both ways)"""
    meta, sd, _, _, _ = load(name)
    m = _build(meta, sd)
    assert list(m.state_dict().keys()) == list(sd.keys())


GRAD_CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "s3_grad_*.npz")))


def test_sibling_gradient_fixtures_are_present():
    assert sum("gcarm" in n for n in GRAD_CASES) >= 3 and sum("afn" in n for n in GRAD_CASES) >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("name", GRAD_CASES)
def test_sibling_training_step_matches_reference_gradients(name):
    """round-2 verdict, missing 3 / next 9: `train.py --model gc_arm | afn` (train.py:108-114: BCEWithLogitsLoss,
    backward) on the device — every parameter's gradient against the reference's own step (captured by
    tests/golden/make_golden.py), the BatchNorm running statistics after the step, AFN's clipped table"""
    meta, sd, ids, vals, ref = load(name)
    m = _build(meta, sd, DEV)
    m.train(meta["train"])
    x = {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
    y = torch.from_numpy(ref["target"]).to(DEV)
    logits = m(x)
    loss = torch.nn.BCEWithLogitsLoss()(logits, y)
    loss.backward()

    def close(got, want, rtol):
        want = np.asarray(want, dtype=np.float64)
        return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want))) / max(float(np.abs(want).max()), 1e-12) <= rtol

    assert tuple(logits.shape) == ref["logits"].shape
    assert close(logits.detach().cpu().numpy(), ref["logits"], 2e-5), "logits"
    assert abs(float(loss.detach()) - float(ref["loss"])) <= 2e-6
    np.testing.assert_array_equal(x["value"].cpu().numpy(), ref["vals_clamped"])
    if "table_after" in ref:
        np.testing.assert_array_equal(m.embedding.embedding.weight.detach().cpu().numpy(), ref["table_after"])
    gmax = max(float(np.abs(ref["grad/" + k]).max()) for k, _ in m.named_parameters())
    worst = {}
    for k, p in m.named_parameters():
        g = ref["grad/" + k]
        if float(np.abs(g).max()) < 1e-5 * gmax:
            # analytically zero up to the BatchNorm eps: a bias in front of a train-mode BatchNorm, and afn.bias, which
            # scales exp(.) per channel right in front of afn_bn (2e-7 against gradients of 1e-1: cancellation noise in
            # the reference's own fp32 step)
            continue
        assert p.grad is not None, k
        worst[k] = float(np.max(np.abs(p.grad.cpu().numpy().astype(np.float64) - g))) / max(float(np.abs(g).max()), 1e-12)
    print(name, {k: f"{v:.1e}" for k, v in worst.items() if v > 1e-5})
    for k, err in worst.items():
        assert err <= 5e-5, f"grad of {k}: rel err {err:.2e}"
    if meta["train"]:
        got = m.state_dict()
        for k in [k for k in ref if k.startswith("after/") and "running_" in k]:
            np.testing.assert_allclose(got[k[6:]].cpu().numpy(), ref[k], rtol=2e-5, atol=2e-6, err_msg=k)


def _analytically_zero(m):
    """parameters whose gradient is zero up to the BatchNorm eps in a training step (cancellation noise in any fp32
    implementation, the reference's included): a bias in front of a train-mode BatchNorm — the hidden Linears' biases,
    arm_bn.bias / afn_bn.bias (followed by Linear -> BatchNorm) — and afn.bias (a per-channel scale of exp(.) in front of afn_bn)"""
    names = {"arm_bn.bias", "afn_bn.bias", "afn.bias"}
    for pref, seq in m.named_modules():
        if isinstance(seq, torch.nn.Sequential):
            mods = list(seq)
            for i in range(len(mods) - 1):
                if isinstance(mods[i], torch.nn.Linear) and isinstance(mods[i + 1], torch.nn.BatchNorm1d):
                    names.add(f"{pref}.{i}.bias")
    return names


GC_FUSED_SHAPES = [  # nfield, nemb, nhead, arm_hid, alpha, batch  (one / several neuron slices, padded nemb, every solver)
    (39, 16, 2, 32, 1.7, 512), (10, 10, 1, 20, 2.0, 300), (22, 32, 2, 8, 1.5, 257), (5, 8, 3, 7, 1.0, 130),
    (43, 16, 1, 70, 2.5, 96), (48, 12, 4, 40, 2.0, 64), (3, 4, 1, 1, 1.3, 33), (30, 27, 2, 24, 1.5, 200),
    (22, 64, 2, 20, 2.0, 100), (39, 48, 1, 33, 1.7, 70),          # nemb 33..64: one 16-neuron pass per launch
    (10, 100, 2, 16, 1.7, 200), (22, 128, 1, 20, 2.0, 100), (30, 72, 2, 8, 1.0, 64), (32, 65, 1, 33, 1.5, 70),   # nemb 65..128 (round 6)
    (5, 96, 3, 7, 2.5, 130),
]


@pytest.mark.gpu
@pytest.mark.parametrize("F,E,K,H,alpha,B", GC_FUSED_SHAPES)
def test_gc_fused_training_step_matches_the_composed_device_ops(F, E, K, H, alpha, B):
    """round-3 verdict, missing 3: GC-ARM's training step through armnet_gc_fused_bwd_f32 (matrix cores, no [B, K*H, F]
    tensor in memory) against the same step through the composed device ops (the path the reference-gradient fixtures
    pinned in round 3): logits, every gradient, both BatchNorm layers' running statistics"""
    from armnet_hip import native
    from armnet_hip.siblings import GC_ARMModel
    assert native.gc_fused_bwd_supported(F, E, K * H)
    nfeat = 997
    g = torch.Generator().manual_seed(F * 131 + E)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 1.2 - 0.1).to(DEV)              # some outside [0.001, 1]: the clamp is live
    y = (torch.rand(B, generator=g) > 0.5).float().to(DEV)
    outs = []
    for fused in (False, True):
        torch.manual_seed(7)
        m = GC_ARMModel(F, nfeat, E, K, alpha, H, 2, 32, 0.0, True, 1, 16).to(DEV).train()
        with torch.no_grad():                                                # trained-like: wide gates, non-trivial affines
            m.embedding.embedding.weight.mul_(4.0)
            m.attn_layers.Q.mul_(3.0)
            for bn in (m.emb_bn, m.arm_bn):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.3, 0.3)
        m.fused_training = fused
        zero_names = _analytically_zero(m)
        assert m._fused_training_ok(ids, vals) == fused
        v = vals.clone()
        logits = m({"id": ids, "value": v})
        torch.nn.BCEWithLogitsLoss()(logits, y).backward()
        outs.append((logits.detach(), {k: p.grad.clone() for k, p in m.named_parameters()},
                     {k: b.clone() for k, b in m.named_buffers()}, v))
    (l0, g0, b0, v0), (l1, g1, b1, v1) = outs
    assert torch.equal(v0, v1)
    assert float((l0 - l1).abs().max()) <= 2e-5 * max(1.0, float(l0.abs().max())), "logits"
    gmax = max(float(t.abs().max()) for t in g0.values())
    rtol = 5e-4 if alpha > 2 else 5e-5                                       # alpha > 2: p^(2-alpha) near p = 0
    worst = {}
    for k in g0:
        if k in zero_names or float(g0[k].abs().max()) < 1e-5 * gmax:
            continue
        bar = rtol * float(g0[k].abs().max()) + 2e-6 * gmax
        worst[k] = float((g0[k] - g1[k]).abs().max()) / bar
    print({k: f"{v:.2f}" for k, v in worst.items() if v > 0.3})
    for k, v in worst.items():
        assert v <= 1.0, f"grad of {k}: {v:.2f} of the bar"
    for k in b0:
        if "running" in k:
            torch.testing.assert_close(b1[k], b0[k], rtol=2e-5, atol=2e-6)
        else:
            assert torch.equal(b0[k], b1[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("F,E,O,B", [(39, 16, 64, 512), (10, 10, 20, 300), (22, 32, 40, 257), (5, 8, 7, 130), (43, 16, 70, 96),
                                     (48, 12, 100, 64), (3, 4, 1, 33), (30, 27, 24, 200), (22, 64, 40, 100), (39, 48, 33, 70),
                                     (10, 100, 10, 200), (22, 128, 40, 100), (32, 65, 33, 70), (13, 96, 16, 130), (5, 77, 7, 64)])
def test_afn_fused_training_step_matches_the_composed_device_ops(F, E, O, B):
    """AFN's training step through armnet_afn_fused_bwd_f32 against the composed device ops (see the GC-ARM test above)"""
    from armnet_hip import native
    from armnet_hip.siblings import AFNModel
    assert native.afn_fused_bwd_supported(F, E, O)
    nfeat = 997
    g = torch.Generator().manual_seed(F * 131 + E)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 1.2 - 0.1).to(DEV)
    y = (torch.rand(B, generator=g) > 0.5).float().to(DEV)
    outs = []
    for fused in (False, True):
        torch.manual_seed(7)
        m = AFNModel(F, nfeat, E, O, 2, 32, 0.0, True, 1, 16).to(DEV).train()
        with torch.no_grad():
            for bn in (m.emb_bn, m.afn_bn):
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.3, 0.3)
            m.afn.bias.uniform_(-0.2, 0.2)
        m.fused_training = fused
        zero_names = _analytically_zero(m)
        assert m._fused_training_ok(ids, vals) == fused
        v = vals.clone()
        logits = m({"id": ids, "value": v})
        torch.nn.BCEWithLogitsLoss()(logits, y).backward()
        outs.append((logits.detach(), {k: p.grad.clone() for k, p in m.named_parameters()},
                     {k: b.clone() for k, b in m.named_buffers()}, v))
    (l0, g0, b0, v0), (l1, g1, b1, v1) = outs
    assert torch.equal(v0, v1)
    assert float((l0 - l1).abs().max()) <= 2e-5 * max(1.0, float(l0.abs().max())), "logits"
    gmax = max(float(t.abs().max()) for t in g0.values())
    worst = {}
    for k in g0:
        if k in zero_names or float(g0[k].abs().max()) < 1e-5 * gmax:
            continue
        bar = 5e-5 * float(g0[k].abs().max()) + 2e-6 * gmax
        worst[k] = float((g0[k] - g1[k]).abs().max()) / bar
    print({k: f"{v:.2f}" for k, v in worst.items() if v > 0.3})
    for k, v in worst.items():
        assert v <= 1.0, f"grad of {k}: {v:.2f} of the bar"
    for k in b0:
        if "running" in k:
            torch.testing.assert_close(b1[k], b0[k], rtol=2e-5, atol=2e-6)
        else:
            assert torch.equal(b0[k], b1[k]), k


@pytest.mark.gpu
def test_sibling_backward_kernels_against_a_float64_restatement_over_shapes():
    """armnet_gc_fused_bwd_f32 / armnet_afn_fused_bwd_f32 on ~330 of the shapes of tools/sibling_bwd_scan.py (every nfield
    1..48, nemb 4..64, one to five neuron slices, every solver there) against the backward written out in float64"""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "sibling_bwd_scan", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "sibling_bwd_scan.py"))
    scan = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(scan)
    g = torch.Generator().manual_seed(0)
    n = 0
    for F in (1, 2, 3, 4, 5, 9, 16, 17, 22, 31, 39, 40, 43, 48):
        for E in (4, 7, 10, 16, 17, 31, 32, 33, 47, 64):
            if (F + E) % 2:
                continue
            for O, kind, alpha in ((1, "gc", 2.0), (20, "gc", 1.7), (70, "gc", 1.0), (70, "gc", 1.5), (20, "afn", 1.0), (70, "afn", 1.0)):
                assert scan.case(kind, F, E, O, alpha, 37, g) == 0.0, (kind, F, E, O, alpha)
                n += 1
    assert n > 300
    for F in (1, 4, 5, 10, 17, 22, 31, 32):                                      # the E = 128 kernel family (round 6): nfield <= 32
        for E in (65, 72, 77, 96, 100, 128):
            if (F + E) % 2:
                continue
            for O, kind, alpha in ((1, "gc", 2.0), (20, "gc", 1.7), (40, "gc", 1.0), (33, "gc", 1.5), (20, "afn", 1.0), (40, "afn", 1.0)):
                assert scan.case(kind, F, E, O, alpha, 37, g) == 0.0, (kind, F, E, O, alpha)
                n += 1
    assert n > 400
    for B in (1, 2, 5, 64, 65, 1025, 4099):                                      # batch sizes around the wave / block granularity
        for F, E, O, kind, alpha in ((39, 16, 64, "gc", 1.7), (22, 32, 33, "gc", 1.5), (5, 64, 17, "gc", 1.0),
                                     (39, 16, 64, "afn", 1.0), (7, 64, 16, "afn", 1.0), (10, 100, 20, "gc", 1.7),
                                     (22, 128, 17, "afn", 1.0)):
            assert scan.case(kind, F, E, O, alpha, B, g) == 0.0, (kind, F, E, O, alpha, B)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gc", "afn"])
def test_sibling_backward_without_batchnorm_coefficients_is_the_identity_coefficients(kind):
    """coefA/B/C = NULL (dy is the gradient of the block's own output) against explicit {1, 0, 0} coefficients: same
    launch, same arithmetic — bit-equal outputs except the float-atomic accumulators (tolerance), int32 ids as well"""
    from armnet_hip import native
    B, F, E, O, nfeat, alpha = 301, 22, 16, 40, 503, 1.7
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(torch.int32).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
    table = (torch.rand(nfeat, E, generator=g) * 0.9 + 0.05).to(DEV)
    qf = (torch.randn(O, E, generator=g) * 0.8).to(DEV)
    values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
    es, et = (torch.rand(F, generator=g) + 0.5).to(DEV), (torch.randn(F, generator=g) * 0.3).to(DEV)
    z, dz = torch.rand(B, O, E, generator=g).to(DEV), torch.randn(B, O, E, generator=g).to(DEV)
    one, zero = torch.ones(O, device=DEV), torch.zeros(O, device=DEV)
    outs = []
    for coefs in ((None, None, None), (one, zero, zero)):
        d_table, d_y = torch.zeros(nfeat, E, device=DEV), torch.empty(B, F, E, device=DEV)
        if kind == "gc":
            d_v, d_q = torch.zeros(O, F, device=DEV), torch.zeros(O, E, device=DEV)
            native.gc_fused_bwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, es, et, z, dz, *coefs, d_table, d_v, d_q, d_y)
            outs.append((d_y, d_table, d_v, d_q))
        else:
            d_w, d_b = torch.zeros(O, F, device=DEV), torch.zeros(O, device=DEV)
            native.afn_fused_bwd(B, F, E, O, 0, ids, vals, table, values, es, et, z, dz, *coefs, d_w, d_b, d_y)
            outs.append((d_y, d_w, d_b))
    assert torch.equal(outs[0][0], outs[1][0])                               # d_y: plain stores
    for a, b in zip(outs[0][1:], outs[1][1:]):
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-6)
    with pytest.raises(native.ArmnetNativeError):                            # all three or none
        if kind == "gc":
            native.gc_fused_bwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, es, et, z, dz, one, None, None,
                                torch.zeros(nfeat, E, device=DEV), torch.zeros(O, F, device=DEV), torch.zeros(O, E, device=DEV),
                                torch.empty(B, F, E, device=DEV))
        else:
            native.afn_fused_bwd(B, F, E, O, 0, ids, vals, table, values, es, et, z, dz, one, None, None,
                                 torch.zeros(O, F, device=DEV), torch.zeros(O, device=DEV), torch.empty(B, F, E, device=DEV))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gc", "afn"])
def test_sibling_fused_step_with_frozen_parameters(kind):
    """round-4 advisor finding: a frozen embedding table / frozen attention parameters get no gradient (and no dense
    zeros + scatter pass is spent on them); the other gradients equal the all-trainable step's; a bad id raises before the
    BatchNorm layers count the batch"""
    from armnet_hip.siblings import AFNModel, GC_ARMModel
    F, E, nfeat, B = 13, 8, 211, 96
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = torch.rand(B, F, generator=g).to(DEV)
    y = (torch.rand(B, generator=g) > 0.5).float().to(DEV)
    grads = []
    for freeze in (False, True):
        torch.manual_seed(9)
        m = (GC_ARMModel(F, nfeat, E, 2, 1.7, 8, 1, 16, 0.0, False, 1, 16) if kind == "gc"
             else AFNModel(F, nfeat, E, 24, 1, 16, 0.0, False, 1, 16)).to(DEV).train()
        frozen = [m.embedding.embedding.weight] + ([m.attn_layers.Q] if kind == "gc" else [])
        if freeze:
            for p in frozen:
                p.requires_grad_(False)
        assert m._fused_training_ok(ids, vals)
        torch.nn.BCEWithLogitsLoss()(m({"id": ids, "value": vals.clone()}), y).backward()
        grads.append({k: (None if p.grad is None else p.grad.clone()) for k, p in m.named_parameters()})
        if freeze:
            assert all(p.grad is None for p in frozen)
            tracked = int(m.emb_bn.num_batches_tracked)
            bad = ids.clone()
            bad[3, 2] = nfeat
            m.check_ids = "sync"                           # raise inside the call, before any statistic is touched
            with pytest.raises(IndexError):
                m({"id": bad, "value": vals.clone()})
            assert int(m.emb_bn.num_batches_tracked) == tracked
            m.check_ids = True
    gmax = max(float(g_.abs().max()) for g_ in grads[0].values() if g_ is not None)
    for k, gfree in grads[1].items():
        if gfree is not None and float(grads[0][k].abs().max()) >= 1e-5 * gmax:     # (analytically zero ones: rounding noise)
            err = float((gfree - grads[0][k]).abs().max()) / float(grads[0][k].abs().max())
            assert err <= 2e-5, (k, err)                                             # float-atomic order


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gc", "afn"])
def test_sibling_graphed_train_step_equals_eager_steps(kind):
    """the fused sibling training step inside armnet_hip.modules.GraphedTrainStep (one hipGraph: no host sync, no
    allocation inside the autograd.Functions that a capture cannot replay) against the same SGD steps run eagerly"""
    from armnet_hip.modules import GraphedTrainStep
    from armnet_hip.siblings import AFNModel, GC_ARMModel
    F, E, nfeat, B = 22, 16, 503, 192
    g = torch.Generator().manual_seed(11)
    batches = [(torch.randint(0, nfeat, (B, F), generator=g).to(DEV), torch.rand(B, F, generator=g).to(DEV),
                (torch.rand(B, generator=g) > 0.5).float().to(DEV)) for _ in range(4)]
    lossf = torch.nn.BCEWithLogitsLoss()
    results = []
    for graphed in (False, True):
        torch.manual_seed(3)
        m = (GC_ARMModel(F, nfeat, E, 2, 1.7, 24, 1, 32, 0.0, False, 1, 16) if kind == "gc"
             else AFNModel(F, nfeat, E, 40, 1, 32, 0.0, False, 1, 16)).to(DEV).train()
        assert m._fused_training_ok(batches[0][0], batches[0][1])
        opt = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
        losses = []
        if graphed:
            state = {k: v.clone() for k, v in m.state_dict().items()}
            step = GraphedTrainStep(m, opt, lossf, *batches[0])
            m.load_state_dict(state)
            for st in opt.state.values():
                st["momentum_buffer"].zero_()
            for ids, vals, y in batches:
                losses.append(float(step(ids, vals.clone(), y)))
        else:
            for ids, vals, y in batches:
                opt.zero_grad(set_to_none=True)
                loss = lossf(m({"id": ids, "value": vals.clone()}), y)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
        results.append((losses, {k: v.clone() for k, v in m.state_dict().items()}))
    (l0, s0), (l1, s1) = results
    np.testing.assert_allclose(l1, l0, rtol=2e-4)
    for k in s0:
        if k.endswith("num_batches_tracked"):
            continue
        a, b = s0[k].float(), s1[k].float()
        assert float((a - b).abs().max()) <= 2e-3 * max(float(a.abs().max()), 1e-3), k


@pytest.mark.gpu
@pytest.mark.parametrize("map_kind", [0, 1])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
@pytest.mark.parametrize("B,F,E", [(77, 13, 10), (1, 3, 4), (4099, 39, 16), (300, 48, 32)])
def test_gather_map_stats_is_gather_then_map_then_bn_stats(map_kind, idt, B, F, E):
    """armnet_gather_map_stats_f32 against armnet_gather_scale_f32 + torch.exp / log + armnet_bn_stats_f32 (through the
    finalised mean / rstd: the shifted sums themselves depend on the pivot only through rounding)"""
    from armnet_hip import native
    nfeat = 211
    g = torch.Generator().manual_seed(B + map_kind)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(idt).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
    table = (torch.rand(nfeat, E, generator=g) * 0.8 + 1e-2).to(DEV)
    w, b = torch.rand(F, generator=g).to(DEV) + 0.5, torch.randn(F, generator=g).to(DEV)
    x = torch.empty(B, F, E, device=DEV)
    native.gather_scale(B * F, E, ids, vals, table, x, None)
    want = torch.exp(x) if map_kind == 0 else torch.log(x)
    rm0, rv0 = torch.zeros(F, device=DEV), torch.ones(F, device=DEV)
    rm1, rv1 = rm0.clone(), rv0.clone()
    ref = native.bn_train_stats(want, w, b, rm0, rv0, 0.1, 1e-5)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    out, sbuf = native.gather_map_stats(ids, vals, table, map_kind, status)
    got = native.bn_train_stats(out, w, b, rm1, rv1, 0.1, 1e-5, stats=sbuf)
    assert int(status.item()) == 0
    torch.testing.assert_close(out, want, rtol=2e-6, atol=1e-6)
    if B * E > 1:
        for a, c in zip(got + (rm1, rv1), ref + (rm0, rv0)):
            torch.testing.assert_close(a, c, rtol=3e-5, atol=3e-6)
    bad = ids.clone()
    bad[0, 0] = nfeat
    native.gather_map_stats(bad, vals, table, map_kind, status)
    assert int(status.item()) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("map_kind", [0, 1])
@pytest.mark.parametrize("idt", [torch.int64, torch.int32])
def test_bn_bwd_scatter_is_apply_times_derivative_scattered(map_kind, idt):
    """armnet_bn_bwd_scatter_f32 against armnet_bn_bwd_apply_f32 + the derivative of exp / log + armnet_scatter_add_f32"""
    from armnet_hip import native
    B, F, E, nfeat = 77, 13, 10, 211
    g = torch.Generator().manual_seed(5 + map_kind)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(idt).to(DEV)
    vals = torch.rand(B, F, generator=g).to(DEV)
    x = (torch.rand(B, F, E, generator=g) * 2 + 0.05).to(DEV)
    t = torch.exp(x) if map_kind == 0 else torch.log(x)
    dy = torch.randn(B, F, E, generator=g).to(DEV)
    cA, cB, cC = (torch.randn(F, generator=g).to(DEV) for _ in range(3))
    want = torch.zeros(nfeat, E, device=DEV)
    dt = native.bn_backward_apply(t, dy, cA, cB, cC)
    native.scatter_add(ids, vals, (dt * t if map_kind == 0 else dt / x).view(B * F, E), want)
    got = torch.zeros(nfeat, E, device=DEV)
    native.bn_bwd_scatter(ids, vals, t, dy, cA, cB, cC, map_kind, got)
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["gc", "afn"])
def test_sibling_adam_steps_reduce_the_loss(kind):
    """train.py-shaped loop: Adam + per-parameter gradient clamp hooks (train.py:61-65)"""
    name = "s3_grad_gcarm_k2_a1.7_train_b256" if kind == "gc" else "s3_grad_afn_h16_train_b256"
    meta, sd, ids, vals, ref = load(name)
    m = _build(meta, sd, DEV).train()
    for p in m.parameters():
        p.register_hook(lambda g: g.clamp(-1.0, 1.0))
    opt = torch.optim.Adam(m.parameters(), lr=3e-3)
    idt, y = torch.from_numpy(ids).to(DEV), torch.from_numpy(ref["target"]).to(DEV)
    losses = []
    for _ in range(25):
        loss = torch.nn.BCEWithLogitsLoss()(m({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)}), y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.8 * losses[0], losses
    m.eval()                                               # and the trained weights serve through the fused kernels
    with torch.no_grad():
        yy = m({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)})
    assert bool(torch.isfinite(yy).all())


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_forward_matches_reference(name):
    meta, sd, ids, vals, ref = load(name)
    m = _build(meta, sd, DEV)
    x = {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
    with torch.no_grad():
        y = m(x)
        if meta["variant"] == "gc":
            block = m.arm_block(x["id"], x["value"].clone())
        else:
            block = m.afn_block(x["id"], x["value"].clone())
    assert tuple(y.shape) == ref["logits"].shape
    np.testing.assert_array_equal(x["value"].cpu().numpy(), ref["vals_clamped"])      # in-place clamp
    assert_close(block.cpu().numpy(), ref["x_arm"], TOL, name + " block")
    e = logit_excess(y.cpu().numpy(), ref["logits"], _term_scale(sd, ids, ref))
    print(f"{name}: logits at {e:.3f} x the data-derived bar")
    assert e <= 1.0
    if "table_after" in ref:                                                           # embedding_clip side effect
        np.testing.assert_array_equal(m.embedding.embedding.weight.detach().cpu().numpy(), ref["table_after"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["s1_gcarm_criteo_k2_a1.7_stress", "s1_gcarm_frappe_k4_a1.5_ens_stress",
                                  "s2_afn_criteo_h32_stress", "s2_afn_odd_f7_e5_h600"])
def test_hip_block_matches_oracle_on_a_larger_batch(name):
    meta, sd, _, _, _ = load(name)
    c = meta["ctor"]
    g = torch.Generator().manual_seed(17)
    B = 1000 + 37
    ids = torch.randint(0, c["nfeat"], (B, c["nfield"]), generator=g)
    vals = torch.rand(B, c["nfield"], generator=g) * 1.2 - 0.1
    m = _build(meta, sd, DEV)
    with torch.no_grad():
        y = m({"id": ids.to(DEV), "value": vals.clone().to(DEV)})
        block = (m.arm_block if meta["variant"] == "gc" else m.afn_block)(ids.to(DEV), vals.clone().to(DEV))
    want = _oracle(meta, sd, ids.numpy(), vals.numpy())
    assert_close(block.cpu().numpy(), want["x_arm"], TOL, name)
    scale = max(1.0, float(np.max(np.abs(want["x_arm"]))))
    assert rel_err(y.cpu().numpy(), want["logits"]) <= TOL * scale


@pytest.mark.gpu
def test_out_of_range_id_raises_indexerror_for_the_siblings():
    for name in ("s1_gcarm_criteo_k2_a2.0_stress", "s2_afn_criteo_h32_stress"):
        meta, sd, ids, vals, _ = load(name)
        m = _build(meta, sd, DEV)
        bad = ids.copy()
        bad[2, 5] = meta["ctor"]["nfeat"]
        x = {"id": torch.from_numpy(bad).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
        with torch.no_grad():
            m(x)                                           # round 6: the report is deferred (block.IdStatus) ...
        with pytest.raises(IndexError):
            m.poll()                                       # ... to poll() or the next call
        m.check_ids = "sync"
        with pytest.raises(IndexError), torch.no_grad():
            m(x)


def _grid_model(variant, F, E, K, nhid, alpha, seed):
    ctor = dict(nfield=F, nfeat=997, nemb=E, nhead=K, alpha=alpha, nhid=nhid, mlp_nlayer=1, mlp_nhid=16, dropout=0.0,
                ensemble=False, deep_nlayer=1, deep_nhid=8)
    torch.manual_seed(seed)
    m = _build({"variant": variant, "ctor": ctor})
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                                  # trained-like statistics: non-trivial BN affines, wide gates
        for name, p in m.named_parameters():
            if "bn" in name:
                p.copy_(torch.rand(p.shape, generator=g) * 0.8 + 0.6 if name.endswith("weight")
                        else torch.randn(p.shape, generator=g) * 0.2)
        for name, b in m.named_buffers():
            if name.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.3)
            if name.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 1.5 + 0.5)
        if variant == "gc":
            m.embedding.embedding.weight.mul_(6.0)
            m.attn_layers.Q.mul_(3.0)
    return ctor, m


# (variant, nfield, nemb, nhead, nhid, alpha): every padded width, one- and two-sample wave groups, neuron slices,
# odd widths, every sparse map of the GC-ARM mode
GRID = [("gc", 39, 16, 2, 16, 2.0), ("gc", 39, 16, 1, 24, 1.5), ("gc", 22, 10, 4, 8, 1.7), ("gc", 10, 5, 1, 7, 1.0),
        ("gc", 13, 32, 2, 20, 2.5), ("gc", 43, 64, 1, 33, 1.3), ("gc", 3, 4, 1, 1, 2.0), ("gc", 30, 24, 3, 100, 1.5),
        ("afn", 39, 16, 1, 32, 0.0), ("afn", 22, 10, 1, 600, 0.0), ("afn", 7, 5, 1, 9, 0.0), ("afn", 13, 32, 1, 40, 0.0),
        ("afn", 43, 64, 1, 17, 0.0), ("afn", 48, 7, 1, 3, 0.0),
        # the 128-wide family (nemb 65...128): two- to twelve-quad field counts, odd widths, neuron slices
        ("gc", 10, 100, 2, 16, 1.7), ("gc", 39, 72, 1, 24, 2.0), ("gc", 22, 128, 4, 8, 1.0), ("gc", 43, 65, 1, 33, 2.5),
        ("gc", 5, 96, 3, 50, 1.3), ("afn", 10, 100, 1, 10, 0.0), ("afn", 39, 96, 1, 16, 0.0), ("afn", 48, 128, 1, 40, 0.0),
        ("afn", 13, 65, 1, 300, 0.0)]


@pytest.mark.gpu
@pytest.mark.parametrize("variant,F,E,K,nhid,alpha", GRID)
def test_matrix_core_and_generic_block_match_the_oracle(variant, F, E, K, nhid, alpha):
    """The siblings run as modes of the matrix-core kernel wherever the ARM block does; the shape-agnostic kernel stays
    the fallback.  Both are held to the oracle on the same batch (ragged size: short last wave group)."""
    from armnet_hip import native
    ctor, m = _grid_model(variant, F, E, K, nhid, alpha, seed=F * 100 + E)
    O = K * nhid if variant == "gc" else nhid
    assert native.sibling_kernel_kind(variant == "afn", F, E, O) == 1
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    B = 515
    ids = torch.randint(0, ctor["nfeat"], (B, F), generator=g)
    vals = torch.rand(B, F, generator=g) * 1.2 - 0.1
    want = _oracle({"variant": variant, "ctor": ctor}, sd, ids.numpy(), vals.numpy())
    m = m.to(DEV)
    if variant == "afn":
        m.embedding_clip()                                 # forward() does this before the block (afn.py:56)
    outs = []
    for flags in (0, native.F_FORCE_GENERIC):
        m.kernel_flags = flags
        v = vals.clone().to(DEV)
        with torch.no_grad():
            block = (m.arm_block if variant == "gc" else m.afn_block)(ids.to(DEV), v)
        np.testing.assert_array_equal(v.cpu().numpy(), want["vals_clamped"])
        assert_close(block.cpu().numpy(), want["x_arm"], TOL, f"{variant} flags={flags}")
        outs.append(block)
    assert torch.isfinite(outs[0]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("variant,F,E,K,nhid,alpha", [("gc", 39, 16, 1, 32, 2.0), ("gc", 39, 16, 2, 16, 1.7),
                                                      ("afn", 39, 16, 1, 32, 0.0), ("afn", 22, 32, 1, 48, 0.0)])
def test_sibling_modes_at_a_batch_that_fills_the_persistent_grid(variant, F, E, K, nhid, alpha):
    """B = 20 011 > 8 192: every wave takes several groups, so the software pipeline's steady state (prefetched rows,
    raw ids two groups ahead, the clamped re-read past the end, a short last group) runs in the sibling modes too"""
    ctor, m = _grid_model(variant, F, E, K, nhid, alpha, seed=7)
    sd = {k: v.numpy() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(11)
    B = 20011
    ids = torch.randint(0, ctor["nfeat"], (B, F), generator=g)
    vals = torch.rand(B, F, generator=g) * 1.2 - 0.1
    want = _oracle({"variant": variant, "ctor": ctor}, sd, ids.numpy(), vals.numpy())
    m = m.to(DEV)
    with torch.no_grad():
        y = m({"id": ids.to(DEV), "value": vals.clone().to(DEV)})
        v = vals.clone().to(DEV)
        block = (m.arm_block if variant == "gc" else m.afn_block)(ids.to(DEV), v)
        block32 = (m.arm_block if variant == "gc" else m.afn_block)(ids.to(DEV).int(), vals.clone().to(DEV))
    np.testing.assert_array_equal(v.cpu().numpy(), want["vals_clamped"])
    assert_close(block.cpu().numpy(), want["x_arm"], TOL, f"{variant} B={B}")
    assert torch.equal(block, block32)                                   # int32 ids: the same kernel, bit for bit
    scale = max(1.0, float(np.max(np.abs(want["x_arm"]))))
    assert rel_err(y.cpu().numpy(), want["logits"]) <= TOL * scale
