"""GPU parity tests proper: every call goes through the C ABI (libarmnet_hip.so) on cuda:0 and is
compared with (i) the golden vectors captured from the real reference and (ii) the CPU oracle on
the same inputs.  Tolerance is the one BASELINE.json's north_star states: 1e-5 fp32."""
import numpy as np
import pytest
import torch

from golden_util import load, load_entmax, model_cases
from model_util import build_model
from oracle import armnet_oracle as orc
from tol_util import elem_excess, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EVAL_CASES = [n for n in model_cases() if "train" not in n]
TOL = 1e-5


def _rel_err(got, ref):
    """the larger of the max-normalised error and the worst ELEMENTWISE error |got - ref| / max(1, |ref|): `<= TOL`
    means every element is within 1e-5 of its own magnitude (absolute below 1), north_star's "within 1e-5 fp32" """
    return max(rel_err(got, ref), elem_excess(got, ref, TOL) * TOL)


def _logit_err(y, ref, meta, sd=None):
    """logits: elementwise 1e-5 like everything else.  In the wide-exponent regime the head's inputs reach 1e3..1e4 and an
    O(1) logit is a cancelling sum of terms that large — no fp32 evaluation with another summation order (the reference on
    another BLAS included) holds 1e-5 absolute there — so the bar comes from the DATA, per sample: 1e-5 of the logit plus
    LOGIT_ULPS units of fp32 roundoff of the magnitude of the terms that logit is summed from (tol_util.logit_excess;
    round 2 divided by one global max |x_arm| instead, 100-1000x looser).  Returned in units of TOL."""
    if str(meta.get("regime", "")).startswith("wide"):
        from tol_util import logit_excess, logit_term_scale
        e = logit_excess(y, ref["logits"], logit_term_scale(sd, ref["x_arm"]))
        print(f"wide-regime logits at {e:.3f} x the data-derived bar")
        return e * TOL
    return _rel_err(y, ref["logits"])


def _run(name, flags=0, id_dtype=torch.int64):
    from armnet_hip import native  # noqa: F401
    meta, sd, ids, vals, ref = load(name)
    meta["_sd"] = sd
    m = build_model(meta, sd, DEV)
    m.kernel_flags = flags
    x = {"id": torch.from_numpy(ids).to(DEV).to(id_dtype), "value": torch.from_numpy(vals.copy()).to(DEV),
         "y": torch.zeros(ids.shape[0])}
    with torch.no_grad():
        x_arm = m.arm_block(x["id"], x["value"].clone())
        y = m(x)
    return meta, ref, x, x_arm, y


@pytest.mark.parametrize("name", EVAL_CASES)
def test_model_forward_matches_reference(name):
    meta, ref, x, x_arm, y = _run(name)
    assert tuple(y.shape) == ref["logits"].shape                      # 0-dim when B == 1 (armnet_1h.py:98)
    assert _rel_err(x_arm.cpu().numpy().reshape(ref["x_arm"].shape), ref["x_arm"]) <= TOL
    assert _logit_err(y.cpu().numpy(), ref, meta, meta["_sd"]) <= TOL
    np.testing.assert_array_equal(x["value"].cpu().numpy(), ref["vals_clamped"])   # in-place clamp


@pytest.mark.parametrize("name", EVAL_CASES)
def test_generic_kernel_matches_reference(name):
    from armnet_hip import native
    meta, ref, x, x_arm, y = _run(name, flags=native.F_FORCE_GENERIC)
    assert _rel_err(x_arm.cpu().numpy().reshape(ref["x_arm"].shape), ref["x_arm"]) <= TOL
    assert _logit_err(y.cpu().numpy(), ref, meta, meta["_sd"]) <= TOL


@pytest.mark.parametrize("name", [n for n in EVAL_CASES if "wide" in n])
def test_hip_head_is_as_accurate_as_the_fp32_gemm_path_on_wide_inputs(name):
    """round-2 verdict, weak 4 / advisor: on the wide-exponent fixtures (head inputs up to 4.5e3) the bf16x3 matrix-core
    head (armnet_mlp_head_f32) is compared with the fp32 hipBLASLt path (hip_head = False) on the SAME input — the
    reference's own x_arm — against a float64 evaluation of the head: its worst error may be at most 2x the GEMM path's
    (plus a quarter ulp of the term magnitude; measured: 0.02-0.17 ulp against 0.02-0.12), so a precision regression of the split cannot hide behind the scaled bar"""
    import copy
    from tol_util import U32, logit_term_scale
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd, DEV)
    x = torch.from_numpy(ref["x_arm"].reshape(ref["x_arm"].shape[0], -1)).to(DEV)
    with torch.no_grad():
        want = copy.deepcopy(m.mlp.mlp).double().eval()(x.double()).cpu().numpy().reshape(-1)
        assert "armnet_mlp_head_f32" in m.mlp.eval_path()
        got = m.mlp(x).cpu().numpy().reshape(-1).astype(np.float64)
        m.mlp.hip_head = False
        blas = m.mlp(x).cpu().numpy().reshape(-1).astype(np.float64)
    scale = logit_term_scale(sd, ref["x_arm"])
    e_hip, e_blas = np.abs(got - want) / scale, np.abs(blas - want) / scale        # in units of the term magnitude
    print(f"{name}: HIP head {e_hip.max() / U32:.2f} ulp, hipBLASLt fp32 {e_blas.max() / U32:.2f} ulp of the term magnitude")
    assert e_hip.max() <= 2.0 * e_blas.max() + 0.25 * U32


@pytest.mark.parametrize("name", [n for n in EVAL_CASES if "a1.0" not in n])
def test_faithful_bisection_matches_reference(name):
    from armnet_hip import native
    meta, ref, x, x_arm, y = _run(name, flags=native.F_FAITHFUL_BISECT)
    assert _rel_err(x_arm.cpu().numpy().reshape(ref["x_arm"].shape), ref["x_arm"]) <= TOL


@pytest.mark.parametrize("name", ["g2_criteo_1h_a2.0_stress", "g3_criteo_mh4_a1.7_stress"])
def test_int32_ids(name):
    meta, ref, x, x_arm, y = _run(name, id_dtype=torch.int32)
    assert _rel_err(y.cpu().numpy(), ref["logits"]) <= TOL


@pytest.mark.parametrize("name", ["g2_criteo_1h_a1.7_stress", "g7_odd_mh3_f7_e8_h5_a2.0"])
def test_against_oracle_on_fresh_random_inputs(name):
    """HIP path vs the CPU oracle on inputs the fixtures do not hold (bigger batch, ragged tail)."""
    meta, sd, ids, vals, _ = load(name)
    c = meta["ctor"]
    g = torch.Generator().manual_seed(123)
    B = 1000 + 37
    ids = torch.randint(0, c["nfeat"], (B, c["nfield"]), generator=g)
    vals = torch.rand(B, c["nfield"], generator=g) * 1.2 - 0.1
    m = build_model(meta, sd, DEV)
    with torch.no_grad():
        got = m.arm_block(ids.to(DEV), vals.clone().to(DEV)).cpu().numpy()
    v = vals.numpy().copy()
    want = orc.arm_block(meta["variant"], ids.numpy(), v, sd, float(c["alpha"]))
    assert _rel_err(got, want) <= TOL


def test_out_of_range_id_raises_indexerror():
    """round 6 (round-5 verdict, weak 3): the DEFAULT path has no host sync per forward.  An id outside [0, nfeat) is flagged
    by the kernel in a pinned host word (it reads row 0) and raised as IndexError at the model's next call or at poll() —
    the shape of the reference's GPU behaviour (nn.Embedding's asynchronous device-side assert, layers.py:20);
    check_ids = "sync" raises before forward returns (the rounds 1-5 behaviour), False switches the test off."""
    meta, sd, ids, vals, _ = load("g2_criteo_1h_a2.0_stress")
    m = build_model(meta, sd, DEV)
    good = lambda: {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
    for bad_id in (meta["ctor"]["nfeat"], -1, 2 ** 40):
        bad = ids.copy()
        bad[3, 7] = bad_id
        x = {"id": torch.from_numpy(bad).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
        with torch.no_grad():
            y = m(x)                                       # deferred: returns (sample 3 read row 0)
        assert y.shape == (ids.shape[0],)
        with pytest.raises(IndexError):
            m.poll()                                       # synchronises, then reads the flag
        m.poll()                                           # the report is consumed: nothing left to raise
        with torch.no_grad():
            m(x)
            torch.cuda.synchronize()
            with pytest.raises(IndexError):
                m(good())                                  # ... or it surfaces at the next call (host-memory read only)
            y_ok = m(good())                               # and that call was not launched: the model is usable again
        m.poll()
        m.check_ids = "sync"
        with pytest.raises(IndexError), torch.no_grad():
            m(x)
        m.check_ids = False
        with torch.no_grad():
            m(x)
        m.poll()                                           # unchecked: nothing is ever reported
        m.check_ids = True
    with torch.no_grad():
        g = m.make_graphed(torch.from_numpy(ids).to(DEV), torch.from_numpy(vals.copy()).to(DEV))
        g(x)                                               # the captured kernels keep the range test
        with pytest.raises(IndexError):
            m.poll()
        assert torch.equal(g(good()), y_ok)
    m.poll()


def test_forward_accepts_ids_vals_pair_and_noncontiguous_values():
    meta, sd, ids, vals, ref = load("g2_criteo_1h_a1.5_stress")
    m = build_model(meta, sd, DEV)
    vt = torch.from_numpy(vals.copy()).to(DEV).t().contiguous().t()     # non-contiguous view
    assert not vt.is_contiguous()
    with torch.no_grad():
        y = m(torch.from_numpy(ids).to(DEV), vt)
    assert _rel_err(y.cpu().numpy(), ref["logits"]) <= TOL
    np.testing.assert_array_equal(vt.cpu().numpy(), ref["vals_clamped"])


def test_embedding_layer_matches_reference():
    from models.layers import Embedding
    meta, sd, ids, vals, ref = load("g2_criteo_1h_a1.7_stress")
    emb = Embedding(meta["ctor"]["nfeat"], meta["ctor"]["nemb"])
    emb.load_state_dict({"embedding.weight": torch.from_numpy(sd["embedding.embedding.weight"])})
    emb = emb.to(DEV)
    x = {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(ref["vals_clamped"]).to(DEV)}
    with torch.no_grad():
        out = emb(x)
    np.testing.assert_array_equal(out.cpu().numpy(), ref["x_emb"])     # gather * value: bit exact
    out2 = emb(x)                                                      # autograd path: same forward, dense table grad
    out2.sum().backward()
    np.testing.assert_array_equal(out2.detach().cpu().numpy(), ref["x_emb"])
    want = torch.zeros_like(emb.embedding.weight).index_add_(0, x["id"].reshape(-1),
                                                              x["value"].reshape(-1, 1).expand(-1, out.shape[-1]))
    torch.testing.assert_close(emb.embedding.weight.grad, want)


def test_entmax_matches_reference_vectors():
    from utils.entmax import EntmaxBisect, entmax_bisect
    for m, X, P in load_entmax():
        got = entmax_bisect(torch.from_numpy(X).to(DEV), alpha=m["alpha"], dim=-1, n_iter=m["n_iter"],
                            ensure_sum_one=m.get("ensure_sum_one", True)).cpu().numpy()
        bar = 2e-6 if m["alpha"] <= 2.0 else 2e-5
        assert float(np.max(np.abs(got - P))) <= bar, m
    X = torch.randn(7, 5, 39, generator=torch.Generator().manual_seed(1))
    want = orc.entmax_bisect(X.permute(0, 2, 1).numpy(), 1.7)          # dim=1 handled by movedim
    got = EntmaxBisect(1.7, dim=1)(X.to(DEV)).permute(0, 2, 1).cpu().numpy()
    assert float(np.max(np.abs(got - want))) <= 2e-6


def test_entmax_with_a_tensor_alpha_matches_reference_vectors():
    """round 6 (round-5 verdict, next 7): utils/entmax.py:31-36 — alpha as a tensor that broadcasts over every dimension but
    `dim`: one alpha per row through armnet_entmax_rows_f32 (the reference's bisection with per-row alpha - 1, 1 / (alpha - 1),
    (1 / d)^(alpha - 1); t^r on the hardware exp2 / log2 pair: the bars of the scalar-alpha bisection)"""
    from golden_util import load_entmax_row_alpha
    from utils.entmax import entmax_bisect
    for m, X, A, P in load_entmax_row_alpha():
        with torch.no_grad():
            got = entmax_bisect(torch.from_numpy(X).to(DEV), alpha=torch.from_numpy(A).to(DEV), dim=m["dim"], n_iter=m["n_iter"],
                                ensure_sum_one=m["ensure_sum_one"]).cpu().numpy()
        assert got.shape == P.shape
        hi = np.broadcast_to(A > 2.0, P.shape) if A.ndim == P.ndim else np.broadcast_to((A > 2.0), P.shape)
        err = np.abs(got - P)
        assert float(err[~hi].max(initial=0.0)) <= 2e-6 and float(err[hi].max(initial=0.0)) <= 2e-5, m
    # ... and against the oracle's per-row restatement on inputs of another size (2 100 rows of 39 and of 7)
    g = torch.Generator().manual_seed(3)
    for d in (39, 7):
        X = torch.randn(7, 300, d, generator=g) * 2.0
        A = 1.05 + 0.95 * torch.rand(7, 300, 1, generator=g)
        with torch.no_grad():
            got = entmax_bisect(X.to(DEV), alpha=A.to(DEV)).cpu().numpy()
        want = orc.entmax_bisect_rows(X.numpy(), A.numpy())
        assert float(np.max(np.abs(got - want))) <= 2e-6
    # gradients with respect to X and to alpha (entmax.py:70-98) against the reference's own, for a random dY: the forward is the
    # HIP per-row map, the backward the reference's formulas as device tensor ops on its output
    from golden_util import load_entmax_row_alpha_grads
    for m, X, A, dY, dX, dA in load_entmax_row_alpha_grads():
        Xt, At = torch.from_numpy(X).to(DEV).requires_grad_(True), torch.from_numpy(A).to(DEV).requires_grad_(True)
        Y = entmax_bisect(Xt, alpha=At, dim=m["dim"], n_iter=m["n_iter"], ensure_sum_one=m["ensure_sum_one"])
        (Y * torch.from_numpy(dY).to(DEV)).sum().backward()
        ex = float(np.max(np.abs(Xt.grad.cpu().numpy() - dX))) / max(1.0, float(np.max(np.abs(dX))))
        ea = float(np.max(np.abs(At.grad.cpu().numpy() - dA))) / max(1.0, float(np.max(np.abs(dA))))
        print(f"{m['key']}: dX {ex:.2e}, d alpha {ea:.2e} (of the largest element)")
        assert tuple(At.grad.shape) == A.shape and ex <= 2e-5 and ea <= 2e-4, (m, ex, ea)


def test_entmax_edge_rows_on_device():
    from utils.entmax import entmax_bisect
    one_hot = entmax_bisect(torch.tensor([[5.0, 0.0, -1.0, 0.5]], device=DEV), 1.5).cpu().numpy()
    np.testing.assert_array_equal(one_hot, [[1, 0, 0, 0]])
    uni = entmax_bisect(torch.zeros(1, 8, device=DEV), 1.7).cpu().numpy()
    np.testing.assert_allclose(uni, np.full((1, 8), 0.125, np.float32), rtol=3e-7)
    nan = entmax_bisect(torch.tensor([[float("inf"), 0.0, 1.0]], device=DEV), 1.5).cpu().numpy()
    assert np.isnan(nan).all()
    masked = entmax_bisect(torch.tensor([[float("-inf"), 0.0, 1.0]], device=DEV), 2.0).cpu().numpy()
    np.testing.assert_allclose(masked, [[0.0, 0.0, 1.0]], atol=1e-7)
    sm = entmax_bisect(torch.tensor([[0.0, 1.0, 2.0]], device=DEV), 1.0).cpu().numpy()
    np.testing.assert_allclose(sm, torch.softmax(torch.tensor([[0.0, 1.0, 2.0]]), -1).numpy(), rtol=1e-6)


def test_attention_submodule_surface():
    """SparseAttention / SparseAttLayer called on their own return the reference's arm_weight."""
    for name in ("g2_criteo_1h_a1.7_stress", "g3_criteo_mh4_a2.0_stress"):
        meta, sd, ids, vals, ref = load(name)
        m = build_model(meta, sd, DEV)
        with torch.no_grad():
            w = m.attn_layer(torch.from_numpy(ref["x_emb"]).to(DEV))
        assert _rel_err(w.cpu().numpy(), ref["arm_weight"]) <= TOL


@pytest.mark.parametrize("R", [1, 2, 8])
@pytest.mark.parametrize("dtype", [torch.int64, torch.int32])
def test_shard_route_kernel_contract(R, dtype):
    """armnet_shard_route_ids: counts, grouping by owner, local indices, perm is a permutation."""
    from armnet_hip.sharded import HipShardOps
    nfeat, n = 100003, 39 * 1237
    ids = torch.randint(0, nfeat, (n,), generator=torch.Generator().manual_seed(R)).to(dtype)
    counts, send_local, perm = HipShardOps().route(ids.to(DEV), R, nfeat)
    counts, send_local, perm = counts.cpu().numpy(), send_local.cpu().numpy(), perm.cpu().numpy()
    idn = ids.numpy().astype(np.int64)
    np.testing.assert_array_equal(counts, np.bincount(idn % R, minlength=R))
    assert sorted(perm.tolist()) == list(range(n))
    owner_of_pos = np.repeat(np.arange(R), counts)
    np.testing.assert_array_equal(owner_of_pos[perm], idn % R)
    np.testing.assert_array_equal(send_local[perm], idn // R)
    counts2, send2, perm2 = (t.cpu().numpy() for t in HipShardOps().route(ids.to(DEV), R, nfeat))
    np.testing.assert_array_equal(perm, perm2)                    # deterministic


@pytest.mark.parametrize("R", [1, 2, 8])
def test_shard_route_unique_kernel_contract(R):
    """armnet_shard_route_unique_ids: every distinct id once, grouped by owner, rows[perm] == table[ids]."""
    from armnet_hip.sharded import HipShardOps
    nfeat, n = 5003, 39 * 811                                     # ~6 lookups per row: heavy duplication
    ids = torch.randint(0, nfeat, (n,), generator=torch.Generator().manual_seed(R))
    counts, send_local, perm = (t.cpu().numpy() for t in HipShardOps().route(ids.to(DEV), R, nfeat, dedup=True))
    idn = ids.numpy()
    nu = int(counts.sum())
    assert nu == np.unique(idn).size
    owner_of_slot = np.repeat(np.arange(R), counts)
    np.testing.assert_array_equal(owner_of_slot[perm], idn % R)
    np.testing.assert_array_equal(send_local[:nu][perm], idn // R)
    assert np.unique(np.stack([owner_of_slot, send_local[:nu]]), axis=1).shape[1] == nu      # no duplicates sent


@pytest.mark.parametrize("name", ["g2_criteo_1h_a2.0_stress", "g4_criteo_1h_e64_a1.7_stress", "g3_criteo_mh4_a2.0_stress"])
def test_row_sharded_path_is_bit_equal_to_replicated(name):
    """world_size 1 exercises route -> gather -> fused kernel over (rows, perm): sharding only moves rows,
    so the result must equal the replicated-table result bit for bit (SURVEY.md §4 iii)."""
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd, DEV)
    g = torch.Generator().manual_seed(5)
    B = 777
    idt = torch.randint(0, meta["ctor"]["nfeat"], (B, meta["ctor"]["nfield"]), generator=g).to(DEV)
    vt = torch.rand(B, meta["ctor"]["nfield"], generator=g).to(DEV)
    with torch.no_grad():
        want = m.arm_block(idt, vt.clone())
        m.shard_embedding()
        for dedup in (False, True):
            m._shard.dedup = dedup
            got = m.arm_block(idt, vt.clone())
            assert torch.equal(got, want), f"dedup={dedup}"


@pytest.mark.parametrize("name", ["g2_criteo_1h_a1.7_stress", "g7_odd_1h_f13_e12_h7_a1.5"])
def test_from_rows_entry_point_is_bit_equal(name):
    """armnet_fused_fwd_from_rows_f32 (pre-gathered unscaled rows) == the gather-fused entry point."""
    from armnet_hip.block import arm_block_forward
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd, DEV)
    idt = torch.from_numpy(ids).to(DEV)
    with torch.no_grad():
        want = m.arm_block(idt, torch.from_numpy(vals.copy()).to(DEV))
        qf, sc, sh = m._folded.q_fold, m._folded.bn_scale, m._folded.bn_shift
        rows = m.embedding.embedding.weight[idt].contiguous()                    # [B,F,E] unscaled
        got = arm_block_forward(None, torch.from_numpy(vals.copy()).to(DEV), None, qf, m.attn_layer.values, sc, sh,
                                m.alpha, rows=rows)
    assert torch.equal(got, want)


def _random_model(variant, F, nfeat, E, alpha, H, K, seed):
    """a product module with stressed random weights (no fixture): its own state_dict feeds the oracle"""
    meta = {"variant": variant, "ctor": dict(nfield=F, nfeat=nfeat, nemb=E, alpha=alpha, nhid=H, d_k=E, nhead=K,
                                             mlp_nlayer=1, mlp_nhid=8, dropout=0.0, ensemble=False, deep_nlayer=1,
                                             deep_nhid=8)}
    torch.manual_seed(seed)
    m = build_model(meta)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        m.embedding.embedding.weight.copy_(torch.rand(nfeat, E, generator=g) * 1.6 - 0.8)
        m.attn_layer.query.mul_(4.0)
        m.attn_layer.values.copy_(torch.randn(m.attn_layer.values.shape, generator=g) * 0.4)
        m.arm_bn.running_mean.copy_(torch.rand(K * H, generator=g) + 0.5)
        m.arm_bn.running_var.copy_(torch.rand(K * H, generator=g) + 0.5)
        m.arm_bn.weight.copy_(torch.rand(K * H, generator=g) + 0.5)
        m.arm_bn.bias.copy_(torch.randn(K * H, generator=g) * 0.2)
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    return meta, m.to(DEV), sd


# (variant, nfield, nemb, nhid, nhead, alpha): every staging family of the MFMA kernel (nemb padded to 16/32/64,
# whole and partial 16-byte chunks), every quarter-step count, neuron counts that are not multiples of 16, > 256 neurons
SHAPE_SWEEP = [
    ("1h", 1, 4, 1, 1, 2.0), ("1h", 2, 4, 3, 1, 1.5), ("1h", 3, 10, 128, 1, 2.0), ("1h", 5, 6, 17, 1, 1.7),
    ("1h", 8, 16, 16, 1, 2.0), ("1h", 9, 14, 33, 1, 1.0), ("mh", 12, 10, 20, 3, 2.0), ("1h", 16, 16, 48, 1, 1.5),
    ("1h", 17, 20, 5, 1, 2.0), ("mh", 21, 20, 9, 2, 1.7), ("1h", 24, 24, 40, 1, 2.0), ("1h", 25, 28, 31, 1, 1.5),
    ("1h", 31, 32, 64, 1, 1.0), ("mh", 33, 10, 64, 4, 2.0), ("1h", 39, 10, 128, 1, 2.0), ("1h", 40, 12, 100, 1, 1.7),
    ("mh", 43, 10, 64, 8, 1.5), ("1h", 47, 8, 19, 1, 2.0), ("1h", 48, 16, 300, 1, 2.0), ("1h", 7, 36, 12, 1, 2.0),
    ("1h", 13, 48, 20, 1, 1.5), ("1h", 22, 64, 32, 1, 2.0), ("mh", 30, 40, 10, 2, 1.7), ("1h", 44, 64, 24, 1, 1.0),
    ("1h", 39, 60, 18, 1, 2.0), ("1h", 6, 64, 7, 1, 1.5),
    # odd nemb: rows are only 4-byte aligned, the partial last chunk holds 1 or 3 floats
    ("1h", 39, 5, 32, 1, 2.0), ("1h", 10, 7, 20, 1, 1.5), ("mh", 22, 9, 16, 2, 1.7), ("1h", 13, 11, 40, 1, 2.0),
    ("1h", 43, 15, 9, 1, 1.0), ("1h", 8, 17, 24, 1, 2.0), ("1h", 24, 33, 16, 1, 1.5), ("1h", 39, 63, 12, 1, 2.0),
    # nemb 65..128 (round 4; README.md:32-42 runs Frappe at nemb 100): every NQ family of the E = 128 instantiations
    ("1h", 10, 100, 10, 1, 1.7), ("1h", 39, 96, 32, 1, 2.0), ("mh", 22, 72, 16, 2, 1.5), ("1h", 3, 128, 40, 1, 2.0),
    ("1h", 30, 65, 20, 1, 1.0), ("1h", 48, 128, 16, 1, 2.0), ("1h", 43, 101, 9, 1, 1.7), ("1h", 17, 67, 33, 1, 2.5),
    ("1h", 39, 120, 300, 1, 2.0),
]


@pytest.mark.parametrize("variant,F,E,H,K,alpha", SHAPE_SWEEP)
def test_shape_sweep_against_oracle(variant, F, E, H, K, alpha):
    """the fused block on shapes the fixtures do not hold, vs the CPU oracle; also the MFMA and the generic kernel
    must agree with each other (both within TOL of the oracle), and int32 ids must be bit-equal to int64 ids"""
    from armnet_hip import native
    nfeat = 97
    assert native.fused_kernel_kind(F, E, K * H, alpha) == 1     # the sweep is about the matrix-core kernel
    meta, m, sd = _random_model(variant, F, nfeat, E, alpha, H, K, seed=1000 + F * 7 + E)
    g = torch.Generator().manual_seed(F * 131 + E)
    B = 71                                               # odd: the last wave-group is short
    ids = torch.randint(0, nfeat, (B, F), generator=g)
    vals = torch.rand(B, F, generator=g) * 1.2 - 0.1
    v = vals.numpy().copy()
    want = orc.arm_block(variant, ids.numpy(), v, sd, float(alpha))
    with torch.no_grad():
        got = m.arm_block(ids.to(DEV), vals.clone().to(DEV))
        got32 = m.arm_block(ids.to(DEV).to(torch.int32), vals.clone().to(DEV))
        m.kernel_flags = native.F_FORCE_GENERIC
        gen = m.arm_block(ids.to(DEV), vals.clone().to(DEV))
    assert _rel_err(got.cpu().numpy(), want) <= TOL
    assert _rel_err(gen.cpu().numpy(), want) <= TOL
    assert torch.equal(got, got32)


def test_nonfinite_embedding_row_poisons_only_the_samples_that_use_it():
    """a NaN / inf embedding row makes the reference's output non-finite for exactly the samples whose ids hit
    it; nfield = 39 has a pad row per sample in the kernel's tile, which must not carry it to the group neighbour"""
    meta, m, sd = _random_model("1h", 39, 97, 16, 2.0, 32, 1, seed=5)
    g = torch.Generator().manual_seed(6)
    B = 64
    ids = torch.randint(2, 97, (B, 39), generator=g)
    vals = torch.rand(B, 39, generator=g)
    ids[0, 0] = 0                # first element of a wave-group: what a pad lane re-reads
    ids[9, 38] = 1
    with torch.no_grad():
        m.embedding.embedding.weight[0, 3] = float("nan")
        m.embedding.embedding.weight[1, :] = float("inf")
        got = m.arm_block(ids.to(DEV), vals.clone().to(DEV)).cpu().numpy()
    sd["embedding.embedding.weight"] = m.embedding.embedding.weight.detach().cpu().numpy()
    want = orc.arm_block("1h", ids.numpy(), vals.numpy().copy(), sd, 2.0)
    bad = np.zeros(B, bool); bad[[0, 9]] = True
    assert (~np.isfinite(want[bad])).any(axis=(1, 2)).all() and np.isfinite(want[~bad]).all()
    assert np.isfinite(got[~bad]).all()
    assert (np.isfinite(got) == np.isfinite(want)).all()
    assert _rel_err(got[~bad], want[~bad]) <= TOL


def test_nan_value_poisons_only_its_own_sample():
    """clamp_(NaN) stays NaN in the reference (armnet_1h.py:81) and makes that sample's neurons NaN;
    the other samples of the same wave-group are untouched (compared with the oracle)."""
    meta, sd, ids, vals, _ = load("g2_criteo_1h_a2.0_stress")
    m = build_model(meta, sd, DEV)
    g = torch.Generator().manual_seed(9)
    B = 64
    idt = torch.randint(0, meta["ctor"]["nfeat"], (B, 39), generator=g)
    vt = torch.rand(B, 39, generator=g)
    vt[5, 7] = float("nan")
    with torch.no_grad():
        got = m.arm_block(idt.to(DEV), vt.clone().to(DEV)).cpu().numpy()
    v = vt.numpy().copy()
    want = orc.arm_block("1h", idt.numpy(), v, sd, 2.0)
    assert np.isnan(got[5]).all() and np.isnan(want[5]).all()
    ok = np.ones(B, bool); ok[5] = False
    assert np.isfinite(got[ok]).all()
    assert _rel_err(got[ok], want[ok]) <= TOL


def test_hipgraph_captured_forward_matches_eager():
    """GraphedForward: static buffers + hipGraph replay give the eager logits and the clamp side effect."""
    meta, sd, ids, vals, ref = load("g2_criteo_1h_a2.0_stress_mlp256")
    m = build_model(meta, sd, DEV)
    idt = torch.from_numpy(ids).to(DEV)
    g = m.make_graphed(idt, torch.from_numpy(vals.copy()).to(DEV))
    for shift in (0, 1):                                   # replay twice with different inputs
        idx = torch.roll(idt, shift, 0)
        v = torch.roll(torch.from_numpy(vals.copy()).to(DEV), shift, 0)
        v2 = v.clone()
        with torch.no_grad():
            want = m({"id": idx, "value": v2})
        got = g({"id": idx, "value": v})
        assert torch.equal(got, want)
        assert torch.equal(v, v2)
    assert _rel_err(g({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)}).cpu().numpy(), ref["logits"]) <= TOL


def test_device_resident_loader_feeds_the_model():
    """data_loader.DeviceLoader: the split lives in HBM, batches are slices; a train.py-style eval loop over it."""
    import os
    from data_loader import DeviceLoader, LibsvmDataset
    from golden_util import GOLDEN
    ds = LibsvmDataset(os.path.join(GOLDEN, "libsvm_small.libsvm"), 10)
    meta, sd, _, _, _ = load("g1_frappe_1h_a1.7_stress")
    m = build_model(meta, sd, DEV)
    seen, outs = 0, []
    for batch in DeviceLoader(ds, batch_size=32, device=DEV):
        assert batch["id"].is_cuda and batch["value"].is_cuda
        with torch.no_grad():
            outs.append(m(batch))
        seen += batch["y"].numel()
    assert seen == ds.nsamples
    with torch.no_grad():
        want = m({"id": ds.feat_id[: ds.nsamples].to(DEV), "value": ds.feat_value[: ds.nsamples].clone().to(DEV)})
    # the MLP's hipBLASLt GEMMs pick batch-size-dependent kernels: equal to rounding, not bit for bit
    torch.testing.assert_close(torch.cat(outs), want, rtol=1e-5, atol=1e-6)
    shuffled = [b["y"].numel() for b in DeviceLoader(ds, batch_size=48, shuffle=True, device=DEV, drop_last=True)]
    assert shuffled == [48, 48]


@pytest.mark.parametrize("E", [4, 7, 10, 16, 20, 32, 64])
def test_matrix_core_kernel_agrees_with_generic_kernel_for_every_nfield(E):
    """every nfield 1..48 x neuron counts x alpha for one nemb family, matrix-core kernel vs the shape-agnostic one
    (itself pinned by the golden vectors) on random stressed inputs.  This scan is what caught an XDL-write ->
    inline-asm-read hazard that only some (nemb, nfield) instantiations scheduled badly."""
    from armnet_hip import native
    B, nfeat = 37, 53
    worst, n = 0.0, 0
    for F in range(1, 49):
        for O in (7, 24, 40):
            for alpha in (1.0, 1.5, 2.0):
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
                assert err <= TOL, f"nfield={F} nemb={E} neurons={O} alpha={alpha}: {err}"
                worst, n = max(worst, err), n + 1
    assert n == 48 * 9


def test_misaligned_buffers_and_empty_batch():
    """buffers need only their natural 4-byte alignment (the kernels' 16-byte global accesses are unaligned-mode
    dwordx4): a table / output at a 4-byte storage offset gives the same result; an empty batch is a no-op"""
    from armnet_hip import native
    F, E, O, alpha, B, nfeat = 39, 16, 32, 2.0, 100, 211
    g = torch.Generator().manual_seed(4)
    big = (torch.rand(nfeat * E + 1, generator=g) * 1.6 - 0.8).to(DEV)
    table_off = big[1:].view(nfeat, E)                    # storage offset of one float
    table = table_off.clone()                             # same values, allocator-aligned
    assert table_off.data_ptr() % 16 == 4
    qf = (torch.randn(O, E, generator=g) * 0.8).to(DEV)
    values = (torch.randn(O, F, generator=g) * 0.4).to(DEV)
    ids = torch.randint(0, nfeat, (B, F), generator=g).to(DEV)
    vals = (torch.rand(B, F, generator=g) * 0.999 + 1e-3).to(DEV)
    sc, sh = (torch.rand(O, generator=g) + 0.5).to(DEV), torch.randn(O, generator=g).to(DEV)
    want = torch.empty(B, O, E, device=DEV)
    native.fused_fwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, sc, sh, want)
    got = torch.empty(B, O, E, device=DEV)
    native.fused_fwd(B, F, E, O, alpha, 50, 0, ids, vals, table_off, qf, values, sc, sh, got)
    assert _rel_err(got.cpu().numpy(), want.cpu().numpy()) <= TOL
    out_big = torch.zeros(B * O * E + 1, device=DEV)
    out_off = out_big[1:].view(B, O, E)
    native.fused_fwd(B, F, E, O, alpha, 50, 0, ids, vals, table, qf, values, sc, sh, out_off)
    assert _rel_err(out_off.cpu().numpy(), want.cpu().numpy()) <= TOL
    # empty batch: nothing is launched, nothing is touched
    e_ids = torch.empty(0, F, dtype=torch.int64, device=DEV)
    e_vals = torch.empty(0, F, device=DEV)
    e_out = torch.empty(0, O, E, device=DEV)
    native.fused_fwd(0, F, E, O, alpha, 50, 0, e_ids, e_vals, table, qf, values, sc, sh, e_out)
    torch.cuda.synchronize()


def test_roc_auc_on_device_matches_sklearn():
    """the same check as the CPU suite's, with the tensors on the GPU (sort / cumsum / bincount on the device)"""
    from sklearn.metrics import roc_auc_score
    from armnet_hip.metrics import roc_auc_device
    g = torch.Generator().manual_seed(12)
    y = (torch.rand(200000, generator=g) > 0.7).float()
    p = torch.round((torch.randn(200000, generator=g) + y) * 64) / 64
    got = roc_auc_device(p.to(DEV), y.to(DEV))
    assert got.is_cuda and abs(float(got) - roc_auc_score(y.numpy(), p.numpy())) <= 1e-10


def test_integration_md_binding_stub_runs_as_written():
    """INTEGRATION.md section 2 shows the ctypes stub a maintainer of the reference would add: execute that very
    code (library path patched to the in-tree build) against a module with the reference's attribute names and
    compare with the product path."""
    import os
    import re
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    md = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = next(b for b in blocks if "models/_armnet_hip.py" in b)
    lib = os.path.join(root, "arm-net_amd", "lib", "libarmnet_hip.so")
    stub = re.sub(r'ctypes\.CDLL\("[^"]*"\)', f'ctypes.CDLL("{lib}")', stub)
    mod = types.ModuleType("_armnet_hip_stub")
    exec(compile(stub, "INTEGRATION.md", "exec"), mod.__dict__)
    meta, sd, ids, vals, ref = load("g2_criteo_1h_a2.0_stress")
    m = build_model(meta, sd, DEV)
    x = {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
    with torch.no_grad():
        mod.fold(m)
        got = mod.arm_block(m, x)
        torch.cuda.synchronize()
    assert _rel_err(got.cpu().numpy().reshape(ref["x_arm"].shape), ref["x_arm"]) <= TOL
    np.testing.assert_array_equal(x["value"].cpu().numpy(), ref["vals_clamped"])


@pytest.mark.parametrize("name", ["g2_criteo_1h_a2.0_stress", "g3_criteo_mh4_a1.7_stress", "g9_criteo_1h_h128_e10_a2.0",
                                  "g9_movielens_1h_h128_e10_a2.5_ens", "g7_odd_mh2_f22_e11_h20_a1.5"])
def test_hip_abi_against_its_cpu_twin_on_identical_arguments(name):
    """ONE set of arguments (raw parameters -> fold -> fused forward) handed to the HIP entry points and to their
    `_cpu` twins (oracle/armnet_cpu_twins.c): folded parameters and outputs agree"""
    from armnet_hip import native
    meta, sd, ids, vals, _ = load(name)
    c = meta["ctor"]
    mh = meta["variant"] == "mh"
    K, H, E, D, F = (c["nhead"] if mh else 1), c["nhid"], c["nemb"], c["d_k"], c["nfield"]
    O = K * H
    bw = sd["attn_layer.bilinear_w"] if mh else sd["attn_layer.bilinear_w.weight"]
    bn = [sd["arm_bn." + k] for k in ("weight", "bias", "running_mean", "running_var")]
    qf_c, sc_c, sh_c = orc.twin_fold_params(1 if mh else 0, K, H, E, D, bw, sd["attn_layer.query"], *bn)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    qf, sc, sh = torch.empty(O, E, device=DEV), torch.empty(O, device=DEV), torch.empty(O, device=DEV)
    native.fold_params(native.MULTI_HEAD if mh else native.ONE_HEAD, K, H, E, D, t(bw), t(sd["attn_layer.query"]),
                       *[t(a) for a in bn], 1e-5, qf, sc, sh)
    assert _rel_err(qf.cpu().numpy(), qf_c) <= 1e-6 and _rel_err(sc.cpu().numpy(), sc_c) <= 1e-6
    assert _rel_err(sh.cpu().numpy(), sh_c) <= 1e-6
    g = torch.Generator().manual_seed(77)
    B = 257
    ids_b = torch.randint(0, c["nfeat"], (B, F), generator=g)
    vals_b = torch.rand(B, F, generator=g) * 1.2 - 0.1
    v_c = vals_b.numpy().copy()
    want, status = orc.twin_fused_fwd(ids_b.numpy(), v_c, sd["embedding.embedding.weight"], qf_c, sd["attn_layer.values"],
                                      sc_c, sh_c, float(c["alpha"]), flags=1)
    v_g = vals_b.clone().to(DEV)
    out = torch.empty(B, O, E, device=DEV)
    native.fused_fwd(B, F, E, O, float(c["alpha"]), 50, native.F_WRITE_CLAMPED_VALS, ids_b.to(DEV), v_g,
                     t(sd["embedding.embedding.weight"]), qf, t(sd["attn_layer.values"]).reshape(O, F).contiguous(), sc, sh, out)
    assert status == 0 and _rel_err(out.cpu().numpy(), want) <= TOL
    np.testing.assert_array_equal(v_g.cpu().numpy(), v_c)


@pytest.mark.parametrize("R,dedup,cap_factor", [(1, False, 1.25), (2, False, 1.25), (8, False, 1.25), (8, True, 1.25),
                                                (4, False, 0.5)])
def test_shard_pad_route_kernel_contract(R, dedup, cap_factor):
    """armnet_shard_pad_route: R equal slots; rows gathered through (send_pad, perm_pad) reproduce table[ids]; the
    overflow flag is raised exactly when an owner's count exceeds the slot"""
    from armnet_hip.sharded import HipShardOps
    nfeat, n = 50021, 39 * 641
    ids = torch.randint(0, nfeat, (n,), generator=torch.Generator().manual_seed(R))
    ops = HipShardOps()
    counts, send_local, perm = ops.route(ids.to(DEV), R, nfeat, dedup=dedup)
    cap = int(cap_factor * n / R) + 16
    overflow = torch.zeros(1, device=DEV, dtype=torch.int32)
    send_pad, perm_pad = ops.pad_route(counts, send_local, perm, R, cap, overflow)
    c = counts.cpu().numpy()
    assert int(overflow.item()) == int((c > cap).any())
    assert send_pad.numel() == R * cap
    sp, pp, idn = send_pad.cpu().numpy(), perm_pad.cpu().numpy(), ids.numpy()
    if not (c > cap).any():
        owner = pp // cap
        np.testing.assert_array_equal(owner, idn % R)                  # the slot belongs to the id's owner
        np.testing.assert_array_equal(sp[pp], idn // R)                # and holds its local row index
        for o in range(R):                                             # unused entries: index 0
            assert (sp[o * cap + c[o]: (o + 1) * cap] == 0).all()


@pytest.mark.parametrize("R,dedup,cap_factor,dtype,nfeat,n", [
    (1, False, 1.25, torch.int64, 50021, 39 * 641), (2, False, 1.25, torch.int32, 50021, 39 * 641),
    (8, False, 1.25, torch.int64, 50021, 39 * 641), (3, False, 1.25, torch.int64, 50021, 39 * 641),
    (8, True, 1.25, torch.int64, 50021, 39 * 641), (1, True, 1.25, torch.int64, 5003, 39 * 811),
    (3, True, 1.25, torch.int32, 5003, 39 * 811), (8, True, 1.25, torch.int64, 1_000_000, 39 * 65536),
    (8, False, 1.06, torch.int64, 100_000_000, 39 * 8192), (4, False, 0.5, torch.int64, 50021, 39 * 641),
    (4, True, 0.2, torch.int64, 5003, 39 * 811), (2, True, 1.25, torch.int64, 7, 5), (64, False, 2.0, torch.int64, 1000, 4099)])
def test_shard_route_fixed_kernel_contract(R, dedup, cap_factor, dtype, nfeat, n):
    """armnet_shard_route_fixed (round 4): the slots of the fixed-capacity protocol in one call.  Contract, whatever order
    the positions inside a slot come out in: perm_pad[i] lies in the slot of id i's owner and send_pad there holds
    id i // R; counts[o] = requests to o (distinct ids with dedup); no position is handed out twice to different rows;
    unused entries hold 0; the overflow flag is raised exactly when a count exceeds the slot, and then every lookup that
    still fits keeps the contract; an out-of-range id is flagged and reads row 0."""
    from armnet_hip.sharded import HipShardOps
    ids = torch.randint(0, nfeat, (n,), generator=torch.Generator().manual_seed(R)).to(dtype)
    ops = HipShardOps()
    cap = max(1, int(cap_factor * (min(n, nfeat) if dedup else n) / R) + 16)
    overflow = torch.zeros(1, device=DEV, dtype=torch.int32)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    send_pad, perm_pad = ops.route_fixed(ids.to(DEV), R, nfeat, cap, dedup, overflow, status)
    assert int(status.item()) == 0 and send_pad.numel() == R * cap and perm_pad.numel() == n
    sp, pp, idn = send_pad.cpu().numpy(), perm_pad.cpu().numpy().astype(np.int64), ids.numpy().astype(np.int64)
    owner, local = idn % R, idn // R
    c = (np.array([np.unique(local[owner == o]).size for o in range(R)]) if dedup else np.bincount(owner, minlength=R))
    assert int(overflow.item()) == int((c > cap).any())
    np.testing.assert_array_equal(pp // cap, owner)                     # always: the slot belongs to the id's owner
    if not (c > cap).any():
        np.testing.assert_array_equal(sp[pp], local)
        key = owner * (nfeat // R + 2) + local                           # one position per distinct (owner, local) ...
        first = {}
        for k, p in zip(key[:200000].tolist(), pp[:200000].tolist()):
            assert first.setdefault(k, p) == p or not dedup
        if dedup:
            assert np.unique(pp).size == np.unique(key).size
        else:
            assert np.unique(pp).size == n                               # ... or per lookup
        for o in range(R):                                               # unused entries: index 0
            assert (sp[o * cap + c[o]: (o + 1) * cap] == 0).all()
    else:
        inside = pp % cap != 0                                           # surplus lookups of an overflowing slot point at entry 0
        assert (sp[pp][inside] == local[inside]).all()
        for o in range(R):                                               # a slot that overflowed is full; the others are exact
            held = np.unique(pp[(owner == o) & (inside | (sp[pp] == local))]).size
            assert held == min(c[o], cap), (o, held, c[o], cap)
    if dedup and not (c > cap).any():                                    # requests inside a slot: sorted by local row index
        for o in range(R):
            seg = sp[o * cap: o * cap + c[o]]
            assert (np.diff(seg) > 0).all()
    bad = ids.clone()
    bad[n // 2] = nfeat
    status.zero_()
    ops.route_fixed(bad.to(DEV), R, nfeat, cap, dedup, overflow, status)
    assert int(status.item()) == 1


@pytest.mark.parametrize("R,dedup,dtype,nfeat,n,hot", [
    (8, False, torch.int64, 1_000_000, 39 * 8192, 65536), (8, True, torch.int64, 1_000_000, 39 * 8192, 65536),
    (3, True, torch.int32, 5003, 39 * 811, 700), (2, False, torch.int32, 50021, 39 * 641, 1), (1, True, torch.int64, 5003, 4099, 5003),
    (4, False, torch.int64, 50021, 39 * 641, 50021), (8, True, torch.int64, 100_000_000, 39 * 8192, 1 << 20)])
def test_shard_route_fixed_hot_rows_contract(R, dedup, dtype, nfeat, n, hot):
    """armnet_shard_route_fixed_hot (round 5): ids below hot_rows are replicated on every rank — perm_pad = R * cap + id for
    them, no slot entry, no count; the cold ids keep the contract of armnet_shard_route_fixed exactly (the same slots as
    routing the cold ids alone would give, up to positions inside a slot); all three forms of the position gather agree"""
    from armnet_hip.sharded import HipShardOps, wait_perm
    u = torch.rand(n, generator=torch.Generator().manual_seed(R + hot), dtype=torch.float64)
    ids = (nfeat ** u - 1).clamp_(0, nfeat - 1).to(dtype)               # bench.py's skewed stream
    ids[:3] = torch.tensor([0, max(hot - 1, 0), min(hot, nfeat - 1)]).to(dtype)
    ops = HipShardOps()
    idn = ids.numpy().astype(np.int64)
    cold = idn >= hot
    owner, local = idn % R, idn // R
    c = (np.array([np.unique(local[cold & (owner == o)]).size for o in range(R)]) if dedup
         else np.bincount(owner[cold], minlength=R))
    cap = int(c.max()) + 16
    overflow = torch.zeros(1, device=DEV, dtype=torch.int32)
    status = torch.zeros(1, device=DEV, dtype=torch.int32)
    idd = ids.to(DEV)
    send_pad, perm_pad = ops.route_fixed(idd, R, nfeat, cap, dedup, overflow, status, hot_rows=hot)
    assert int(status.item()) == 0 and int(overflow.item()) == 0
    sp, pp = send_pad.cpu().numpy(), perm_pad.cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(pp[~cold], R * cap + idn[~cold])         # hot: straight into the appended hot rows
    np.testing.assert_array_equal(pp[cold] // cap, owner[cold])             # cold: a slot of the id's owner ...
    np.testing.assert_array_equal(sp[pp[cold]], local[cold])                # ... that asks for the id's row
    for o in range(R):
        used = sp[o * cap: (o + 1) * cap]
        assert (used[c[o]:] == 0).all()                                     # hot ids took no slot entry
        if dedup:
            assert (np.diff(used[:c[o]]) > 0).all()
    if dedup:
        assert np.unique(pp[cold]).size == sum(c)
        for kw in ({"defer_perm": True}, {"perm_with_gather": True}):       # the other two forms of the position gather
            sp1, pp1 = ops.route_fixed(idd, R, nfeat, cap, True, overflow, status, hot_rows=hot, **kw)
            pend = getattr(pp1, "_armnet_pending", None)
            if pend is not None:
                ops.gather_perm(sp1, torch.zeros(max(int(local.max()) + 1, 1), 4, device=DEV), pend)
            wait_perm(pp1)
            assert torch.equal(sp1, send_pad) and torch.equal(pp1, perm_pad), kw
    else:
        assert np.unique(pp[cold]).size == int(cold.sum())


@pytest.mark.parametrize("R,dtype", [(1, torch.int64), (8, torch.int32), (3, torch.int64)])
def test_position_gather_on_a_side_stream_gives_the_in_order_result(R, dtype):
    """armnet_shard_route_fixed(perm_pad = NULL) + armnet_shard_route_fixed_perm on a side stream (what ranks > 1 do beside
    their exchanges): the same perm_pad as the one-call form, step after step on the same workspace — the next route
    waits for the previous gather, which still reads the position table"""
    from armnet_hip.sharded import HipShardOps, wait_perm
    nfeat, n = 200_003, 39 * 4099
    ops = HipShardOps()
    cap = (nfeat + R - 1) // R
    overflow = torch.zeros(1, device=DEV, dtype=torch.int32)
    for step in range(4):
        ids = torch.randint(0, nfeat, (n,), generator=torch.Generator().manual_seed(10 * R + step)).to(dtype).to(DEV)
        sp0, pp0 = ops.route_fixed(ids, R, nfeat, cap, True, overflow)
        sp1, pp1 = ops.route_fixed(ids, R, nfeat, cap, True, overflow, defer_perm=True)
        assert getattr(pp1, "_armnet_ready", None) is not None
        wait_perm(pp1)
        assert torch.equal(sp0, sp1) and torch.equal(pp0, pp1), step
    assert int(overflow.item()) == 0


@pytest.mark.parametrize("R,dtype,E", [(1, torch.int64, 16), (8, torch.int32, 10), (3, torch.int64, 7), (2, torch.int64, 64)])
def test_owner_gather_and_position_gather_in_one_launch(R, dtype, E):
    """armnet_shard_gather_perm_f32 (the owner-side gather and the position gather of the de-duplicating route as one launch)
    against armnet_gather_scale_f32 + the one-call route: same rows bit for bit, same perm_pad; 16- / 8- / 4-byte chunks;
    an index outside the shard reads row 0; step after step on the same workspace"""
    from armnet_hip import native
    from armnet_hip.sharded import HipShardOps
    nfeat, n = 200_003, 39 * 4099
    ops = HipShardOps()
    cap = (nfeat + R - 1) // R
    overflow = torch.zeros(1, device=DEV, dtype=torch.int32)
    shard = torch.randn(cap + 3, E, generator=torch.Generator().manual_seed(E)).to(DEV)
    for step in range(3):
        ids = torch.randint(0, nfeat, (n,), generator=torch.Generator().manual_seed(10 * R + step)).to(dtype).to(DEV)
        sp0, pp0 = ops.route_fixed(ids, R, nfeat, cap, True, overflow)
        sp1, pp1 = ops.route_fixed(ids, R, nfeat, cap, True, overflow, perm_with_gather=True)
        pending = pp1._armnet_pending
        idx = sp1.clone()
        if step == 2:
            idx[5] = shard.shape[0] + 7                          # (never produced by the routing: contract of the entry point)
        rows1 = ops.gather_perm(idx, shard, pending)
        want = ops.gather(torch.where(idx < shard.shape[0], idx, torch.zeros_like(idx)), shard)
        assert torch.equal(sp0, sp1) and torch.equal(pp0, pp1) and torch.equal(rows1, want), step
    assert int(overflow.item()) == 0


def test_mark_epochs_give_the_filled_map_result_over_a_full_cycle():
    """armnet_shard_route_fixed_epoch: 300 steps on one workspace with the byte map zeroed once per 255 steps (epochs 0, 2, ..,
    255, 0, ..) against the same steps with a fill per step — same slots, same positions, every step (a stale mark of an
    earlier step would add a request)"""
    from armnet_hip.sharded import HipShardOps
    nfeat, n, R = 50_021, 8191, 4
    a, b = HipShardOps(), HipShardOps()
    b.mark_epochs = False
    cap = (nfeat + R - 1) // R
    overflow = torch.zeros(1, device=DEV, dtype=torch.int32)
    g = torch.Generator().manual_seed(77)
    for step in range(300):
        ids = torch.randint(0, nfeat, (n,), generator=g).to(DEV)
        sa, pa = a.route_fixed(ids, R, nfeat, cap, True, overflow)
        sb, pb = b.route_fixed(ids, R, nfeat, cap, True, overflow)
        assert torch.equal(sa, sb) and torch.equal(pa, pb), step
    # a workspace whose layout changes starts a new cycle
    sa, pa = a.route_fixed(ids, 2, nfeat, (nfeat + 1) // 2, True, overflow)
    sb, pb = b.route_fixed(ids, 2, nfeat, (nfeat + 1) // 2, True, overflow)
    assert torch.equal(sa, sb) and torch.equal(pa, pb)
    assert int(overflow.item()) == 0


def test_row_sharded_step_is_bit_equal_with_the_fused_and_the_round3_routing():
    """both routings of the fixed protocol (armnet_shard_route_fixed | route + pad_route) feed the same rows to the fused
    kernel: bit-equal outputs, with and without de-duplication"""
    meta, sd, ids, vals, ref = load("g2_criteo_1h_a2.0_stress")
    m = build_model(meta, sd, DEV)
    g = torch.Generator().manual_seed(11)
    B = 4099
    idt = torch.randint(0, meta["ctor"]["nfeat"], (B, meta["ctor"]["nfield"]), generator=g).to(DEV)
    vt = torch.rand(B, meta["ctor"]["nfield"], generator=g).to(DEV)
    with torch.no_grad():
        want = m.arm_block(idt, vt.clone())
        m.shard_embedding()
        m._shard.whole_shard = False
        for dedup in (False, True):
            for fused in (True, False):
                for gwp in ((True, False) if (dedup and fused) else (True,)):       # position gather inside the owner gather's launch | own launch
                    m._shard.dedup, m._shard.fused_route, m._shard.gather_with_perm = dedup, fused, gwp
                    got = m.arm_block(idt, vt.clone())
                    assert m._shard.last_path == "fixed" and not m._shard.overflowed()
                    assert torch.equal(got, want), (dedup, fused, gwp)


@pytest.mark.parametrize("name", ["g5_avazu_mh4_ens_a1.7_stress", "g5_avazu_1h_ens_a2.0_fresh", "g12_criteo_mh4_h8_e10_a2.0_ens_mlp500_dnn500",
                                  "g9_frappe_mh4_h4_e10_a1.5_ens"])
def test_ensemble_forward_runs_without_a_torch_op_in_the_tail(name):
    """round-3 verdict, item 7: the ensemble tail (armnet.py:93-99: cat([y, y_deep]) -> Linear(2, 1)) is folded into the
    final Linears of the two head launches, which write / accumulate one logits buffer.  The ensemble Linear's forward
    is made to raise: the eval-mode forward must not call it; logits against the reference's, and against the composed
    path (torch cat + Linear) on the same model; a changed ensemble weight is picked up."""
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd, DEV)
    idt = torch.from_numpy(ids).to(DEV)
    with torch.no_grad():
        m.mlp.hip_head = False                                        # composed path: torch tail
        want = m({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)})
        m.mlp.hip_head = True
        m.invalidate_folded()
        called = []
        h = m.ensemble_layer.register_forward_pre_hook(lambda mod, inp: called.append(1))
        got = m({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)})
        assert not called, "the ensemble Linear ran as a torch op"
        y = got.cpu().numpy()
        assert elem_excess(y, ref["logits"], 1e-5) <= 1.0, "logits against the reference's, elementwise 1e-5"
        assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max()))
        m.ensemble_layer.weight.mul_(torch.tensor([[2.0, -1.0]], device=DEV))      # in-place: the version counter moves
        m.ensemble_layer.bias.add_(0.25)
        got2 = m({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)})
        h.remove()
        m.mlp.hip_head = False
        m.invalidate_folded()
        want2 = m({"id": idt, "value": torch.from_numpy(vals.copy()).to(DEV)})
        assert float((got2 - want2).abs().max()) <= 4e-6 * max(1.0, float(want2.abs().max()))
        assert float((got2 - got).abs().max()) > 1e-3


@pytest.mark.gpu
def test_forwards_in_flight_give_the_sequential_results():
    """armnet_hip.serving.InFlight: consecutive batches on alternating streams, bit-equal to one-at-a-time execution"""
    from armnet_hip.serving import InFlight
    meta, sd, _, _, _ = load("g2_criteo_1h_a2.0_stress")
    c = meta["ctor"]
    m = build_model(meta, sd, DEV)
    m.check_ids = False
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randint(0, c["nfeat"], (3000 + 17 * k, c["nfield"]), generator=g).to(DEV),
                (torch.rand(3000 + 17 * k, c["nfield"], generator=g) * 1.2 - 0.1).to(DEV)) for k in range(7)]
    with torch.no_grad():
        want = [m({"id": i, "value": v.clone()}) for i, v in batches]
    for n in (1, 2, 3):
        fl = InFlight(m, n=n)
        hs = [fl.submit(i, v.clone()) for i, v in batches]
        got = [fl.result(h) for h in hs]
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got, want)), n
    blk = InFlight(m, n=2, call="arm_block")
    y = blk.result(blk.submit(batches[0][0], batches[0][1].clone()))
    with torch.no_grad():
        assert torch.equal(y, m.arm_block(batches[0][0], batches[0][1].clone()))


@pytest.mark.parametrize("name", [n for n in EVAL_CASES if n.startswith("g12_")])
def test_run_sh_500_wide_heads_take_the_hip_head_kernel(name):
    """round-2 verdict, missing 4: run.sh:18-19,44-45 build the Criteo models with --mlp_hid 500 / --dnn_hid 500; such
    heads used to fall back to hipBLASLt silently.  They now run as slices of <= 256 units of armnet_mlp_head_f32 and
    match both the reference's logits (elementwise 1e-5) and the fp32 GEMM path."""
    meta, sd, ids, vals, ref = load(name)
    m = build_model(meta, sd, DEV)
    heads = [m.mlp] + ([m.deep_mlp] if hasattr(m, "deep_mlp") else [])
    for h in heads:
        assert "armnet_mlp_head_f32" in h.eval_path() and len(h._hip_plan()) == 4       # 2 layers x 2 slices
    x = {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
    with torch.no_grad():
        y = m(x).cpu().numpy()
        for h in heads:
            h.hip_head = False
        y_blas = m({"id": x["id"], "value": torch.from_numpy(vals.copy()).to(DEV)}).cpu().numpy()
    assert _rel_err(y, ref["logits"]) <= TOL
    assert _rel_err(y, y_blas) <= TOL


@pytest.mark.parametrize("alpha", [1.3, 1.5, 1.7])
@pytest.mark.parametrize("F,spread", [(39, 0.0), (39, 3e-4), (39, 3e-2), (48, 1e-3), (10, 1e-3)])
def test_first_order_newton_finish_on_near_tied_gate_rows(alpha, F, spread):
    """round-5 advisor finding (low): the first-order finish of the Newton solver (1 < alpha < 2: a step below `lin_tol` is
    taken without a confirming evaluation, p follows as p - r t^(r-1) step, clamped at 0) was only exercised by the fixtures'
    ordinary rows.  Adversarial rows here: every field of a sample looks up (nearly) the SAME embedding, so a row's gates are
    tied to within `spread` x their scale — many elements enter or leave the support within one step — at widths up to the
    kernel's 48 fields, with larger and smaller gate scales per neuron.  The block output is held to the oracle's literal
    bisection at the usual bar, with the finish on AND switched off at run time (ARMNET_F_NO_LIN_FINISH)."""
    from armnet_hip import native
    from models.armnet_1h import ARMNetModel
    torch.manual_seed(int(alpha * 10) + F)
    E, H, nfeat, B = 16, 32, 64, 203
    m = ARMNetModel(F, nfeat, E, alpha, H, E, 1, 16, 0.0, False, 1, 16)
    g = torch.Generator().manual_seed(F)
    with torch.no_grad():
        base = torch.randn(1, E, generator=g) * 0.5
        m.embedding.embedding.weight.copy_(base + spread * torch.randn(nfeat, E, generator=g))
        m.attn_layer.query.mul_(torch.logspace(-1, 1.2, H).unsqueeze(1))          # gate scale 0.1 ... 16 over the neurons
        m.arm_bn.running_mean.uniform_(0.5, 1.5, generator=g)
        m.arm_bn.running_var.uniform_(0.5, 2.0, generator=g)
    m = m.eval().to(DEV)
    ids = torch.randint(0, nfeat, (B, F), generator=g)
    vals = 1.0 - 1e-3 * torch.rand(B, F, generator=g)                              # values ~ 1: the ties survive the scaling
    vals[::7] = torch.rand(F, generator=g)                                          # ... and some ordinary rows
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    want = orc.arm_block("1h", ids.numpy(), vals.numpy().copy(), sd, alpha)
    outs = {}
    for name, fl in (("lin", 0), ("no_lin", native.F_NO_LIN_FINISH)):
        m.kernel_flags = fl
        with torch.no_grad():
            outs[name] = m.arm_block(ids.to(DEV), vals.clone().to(DEV)).cpu().numpy()
        e = _rel_err(outs[name], want)
        print(f"alpha={alpha} F={F} spread={spread:g} {name}: {e:.2e}")
        assert e <= TOL, (name, e)
    assert np.isfinite(outs["lin"]).all()


def test_host_branch_warns_once_on_a_box_with_a_gpu():
    """round-5 advisor finding (low): a model that was never moved to the GPU runs the reference's ATen chain on the host —
    the reference's own device semantics, but on a GPU box most likely a forgotten .cuda(): one RuntimeWarning per model"""
    import warnings
    from models.armnet_1h import ARMNetModel
    m = ARMNetModel(10, 50, 8, 1.7, 4, 8, 1, 8, 0.0, False, 1, 8).eval()
    x = {"id": torch.randint(0, 50, (5, 10)), "value": torch.rand(5, 10)}
    with pytest.warns(RuntimeWarning, match="host memory"), torch.no_grad():
        m(x)
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter("error")
        m(x)                                                                        # once
        m2 = ARMNetModel(10, 50, 8, 1.7, 4, 8, 1, 8, 0.0, False, 1, 8).eval()
        m2.allow_host = True
        m2(x)


def test_copied_and_pickled_models_get_their_own_id_report():
    """the deferred id report lives in pinned host memory: copy.deepcopy / pickle of a model must not carry a plain copy of it"""
    import copy
    import io
    meta, sd, ids, vals, _ = load("g2_criteo_1h_a2.0_stress")
    m = build_model(meta, sd, DEV)
    x = lambda: {"id": torch.from_numpy(ids).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)}
    with torch.no_grad():
        y = m(x())
        m2 = copy.deepcopy(m)
        assert m2._id_status is not m._id_status and m2.embedding._id_status is m2._id_status   # shared inside the copy
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        m3 = torch.load(buf, weights_only=False)
        for mm in (m2, m3):
            assert torch.equal(mm(x()), y)
            bad = ids.copy()
            bad[0, 0] = -5
            mm({"id": torch.from_numpy(bad).to(DEV), "value": torch.from_numpy(vals.copy()).to(DEV)})
            with pytest.raises(IndexError):
                mm.poll()
        m.poll()                                            # the original's report is untouched
