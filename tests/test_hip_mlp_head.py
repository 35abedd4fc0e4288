"""armnet_mlp_head_f32: the eval-mode prediction head models/layers.py:68-88 as one HIP kernel on the
16-bit matrix cores (round 6: fp16 x 2 split, three cross products, in-kernel fallback to the bf16 x 3 split with six;
fp32 accumulate).  First gate: logits within 1e-5 of a
float64 evaluation of the same nn.Sequential, elementwise — the fp32 hipBLASLt path is held to the same bar beside it
so the two error levels can be compared."""
import numpy as np
import pytest
import torch

from tol_util import TOL, elem_excess

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _make_head(K0, nlayers, nhid, seed, bn_regime="stress", noutput=1):
    from armnet_hip.modules import _MLP
    torch.manual_seed(seed)
    m = _MLP(K0, nlayers, nhid, 0.1, noutput)             # dropout is the identity in eval mode
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for bn in [x for x in m.modules() if isinstance(x, torch.nn.BatchNorm1d)]:
            if bn_regime == "stress":
                bn.running_mean.copy_(torch.randn(bn.running_mean.shape, generator=g) * 0.3)
                bn.running_var.copy_(torch.rand(bn.running_var.shape, generator=g) * 1.5 + 0.25)
                bn.weight.copy_(torch.rand(bn.weight.shape, generator=g) + 0.5)
                bn.bias.copy_(torch.randn(bn.bias.shape, generator=g) * 0.2)
    return m.eval()


def _ref64(m, x):
    import copy
    with torch.no_grad():
        return copy.deepcopy(m.mlp).double().eval()(x.double())


# (K0, nlayers, nhid): the headline head (512 -> 256 -> 256 -> 1), config 3 (2048 inputs), the ensemble's deep MLP
# (nfield * nemb inputs), every tile count (hidden widths padded to 32/64/128/256), widths that are not multiples of 32
# or 4, an input width with a partial last k-step, one / three / four hidden layers (chained launches)
HEADS = [(512, 2, 256), (2048, 2, 256), (704, 2, 256), (320, 2, 32), (130, 1, 16), (100, 3, 64), (37, 1, 10),
         (512, 2, 128), (17, 4, 8), (640, 2, 200), (1280, 1, 256), (390, 3, 16),
         # wider than 256 (round 3; run.sh:18-19,44-45 build --mlp_hid 500 / --dnn_hid 500): slices of <= 256 units, one
         # hidden layer per launch, the final Linear accumulated over the slices of the last hidden layer
         (1280, 2, 500), (390, 2, 500), (512, 1, 300), (64, 3, 600), (100, 2, 257), (2048, 2, 512)]


# the kernel runs with 4-wave blocks (128 samples) while all of them fit the chip at once (B <= 32 768) and with 8-wave
# blocks (256 samples) beyond: both sides of the switch, ragged last blocks
BIG = [(512, 2, 256, 33001), (512, 2, 256, 32768), (320, 2, 32, 40007), (130, 1, 16, 33001), (640, 2, 200, 33001),
       (17, 4, 8, 50000), (100, 3, 64, 32769), (1280, 2, 500, 33001)]


@pytest.mark.parametrize("K0,nlayers,nhid,B", BIG)
def test_mlp_head_matches_float64_evaluation_on_both_block_sizes(K0, nlayers, nhid, B):
    m = _make_head(K0, nlayers, nhid, seed=K0 + nhid).to(DEV)
    g = torch.Generator().manual_seed(B)
    x = (torch.rand(B, K0, generator=g) * 3.0 - 1.0).to(DEV)
    want = _ref64(m, x).cpu().numpy()
    with torch.no_grad():
        got = m(x).cpu().numpy()
        small = torch.cat([m(x[i:i + 8192]) for i in range(0, B, 8192)]).cpu().numpy()   # the 4-wave kernel, in pieces
    e = elem_excess(got, want, TOL)
    assert e <= 1.0, f"worst element at {e:.3f} x the 1e-5 bar"
    np.testing.assert_array_equal(got, small)            # same arithmetic per sample whatever the block size


@pytest.mark.parametrize("K0,nlayers,nhid", HEADS)
@pytest.mark.parametrize("B", [1, 37, 1000])
def test_mlp_head_matches_float64_evaluation(K0, nlayers, nhid, B):
    m = _make_head(K0, nlayers, nhid, seed=K0 + nhid).to(DEV)
    assert m._hip_plan() is not None and "armnet_mlp_head_f32" in m.eval_path()
    if nhid > 256:                                         # sliced: every launch at most 256 units wide
        assert all(n1 - n0 <= 256 and n == 1 for _, n, _, n0, n1 in m._hip_plan())
        assert [hf for _, _, hf, _, _ in m._hip_plan() if hf] == [1] + [2] * ((len(m._hip_plan()) // nlayers) - 1)
    g = torch.Generator().manual_seed(B)
    x = (torch.rand(B, K0, generator=g) * 3.0 - 1.0).to(DEV)           # post-BatchNorm neurons: O(1), both signs
    want = _ref64(m, x).cpu().numpy()
    with torch.no_grad():
        got = m(x).cpu().numpy()
        m.hip_head = False
        blas = m(x).cpu().numpy()
    assert got.shape == (B, 1)
    e_hip, e_blas = elem_excess(got, want, TOL), elem_excess(blas, want, TOL)
    print(f"K0={K0} nlayers={nlayers} nhid={nhid} B={B}: HIP head {e_hip * TOL:.2e}, hipBLASLt fp32 {e_blas * TOL:.2e}")
    assert e_hip <= 1.0, f"worst element at {e_hip:.3f} x the 1e-5 bar (hipBLASLt fp32 path: {e_blas:.3f} x)"


@pytest.mark.parametrize("K0,nlayers,nhid", [(512, 2, 256), (2048, 2, 256), (704, 2, 256), (320, 2, 32), (100, 3, 64), (1280, 2, 500)])
def test_both_operand_splits_sit_at_the_fp32_gemm_error_level(K0, nlayers, nhid):
    """round 6: fp16 x 2 (three products, the default) and bf16 x 3 (six, `bf16x3 = True` / ARMNET_MLP_F_BF16X3) against a
    float64 evaluation, the fp32 hipBLASLt path beside them"""
    B = 2000
    m = _make_head(K0, nlayers, nhid, seed=K0 + nhid).to(DEV)
    x = (torch.rand(B, K0, generator=torch.Generator().manual_seed(5)) * 3.0 - 1.0).to(DEV)
    want = _ref64(m, x).cpu().numpy()
    err = {}
    with torch.no_grad():
        err["f16x2"] = float(np.max(np.abs(m(x).cpu().numpy() - want)))
        m.bf16x3 = True
        assert "bf16x3" in m.eval_path()
        err["bf16x3"] = float(np.max(np.abs(m(x).cpu().numpy() - want)))
        m.hip_head = False
        err["blas"] = float(np.max(np.abs(m(x).cpu().numpy() - want)))
    print(f"K0={K0} nlayers={nlayers} nhid={nhid}: " + ", ".join(f"{k} {v:.2e}" for k, v in err.items()))
    scale = max(1.0, float(np.abs(want).max()))
    assert err["f16x2"] <= TOL * scale and err["bf16x3"] <= TOL * scale
    assert err["f16x2"] <= 2.0 * max(err["blas"], err["bf16x3"], 1e-7 * scale)


@pytest.mark.parametrize("B", [300, 40000])
def test_blocks_outside_the_fp16_range_redo_in_bf16x3(B):
    """the reference has no clamp behind exp (armnet_1h.py:86): a first-layer input may be anything.  A block that meets
    |x| > 4 062, inf or -inf runs the bf16 x 3 split — bit-equal to the forced bf16 x 3 launch for its samples —, every other
    block keeps its fp16 x 2 result; a NaN poisons its own sample only"""
    K0, nlayers, nhid = 512, 2, 256
    m = _make_head(K0, nlayers, nhid, seed=9).to(DEV)
    x = (torch.rand(B, K0, generator=torch.Generator().manual_seed(B)) * 3.0 - 1.0).to(DEV)
    blk = 128 if B <= 32768 else 256
    with torch.no_grad():
        plain = m(x).clone()
        xb = x.clone()
        xb[5, 17] = 5.0e3                     # block 0
        xb[B - 1, 500] = -7.0e4               # last block
        xb[blk + 3, 0] = float("nan")         # block 1: stays fp16 x 2
        auto = m(xb).clone()
        m.bf16x3 = True
        forced = m(xb).clone()
        m.bf16x3 = False
        want = _ref64(m, xb)
    last0 = (B - 1) // blk * blk
    redo = torch.zeros(B, dtype=torch.bool, device=DEV)
    redo[:blk] = True
    redo[last0:] = True
    assert torch.equal(auto[redo], forced[redo])
    keep = ~redo
    keep[blk + 3] = False
    assert torch.equal(auto[keep], plain[keep])
    assert bool(torch.isnan(auto[blk + 3]).all()) and bool(torch.isfinite(auto[keep]).all())
    fin = torch.isfinite(want[:, 0])
    assert elem_excess(auto[fin].cpu().numpy(), want[fin].cpu().numpy(), TOL) <= 1.0
    with torch.no_grad():                     # +inf: non-finite logits for that sample, as in the reference
        xi = x.clone()
        xi[7, 3] = float("inf")
        yi = m(xi)
    assert not bool(torch.isfinite(yi[7]).any()) and torch.equal(yi[blk:], plain[blk:])


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 30.0, 4e3, 1e5])
def test_mlp_head_keeps_fp32_accuracy_over_the_input_range(scale):
    """inputs from 1e-3 to the wide-exponent regime's 4e3: the bf16 split has fp32's exponent range, so the error
    relative to the magnitude of the terms is the same at every scale; compared with the fp32 GEMM path's own error"""
    K0, nlayers, nhid, B = 512, 2, 256, 513
    m = _make_head(K0, nlayers, nhid, seed=3).to(DEV)
    g = torch.Generator().manual_seed(7)
    x = ((torch.rand(B, K0, generator=g) * 2.0 - 1.0) * scale).to(DEV)
    want = _ref64(m, x).cpu().numpy()
    with torch.no_grad():
        got = m(x).cpu().numpy()
        m.hip_head = False
        blas = m(x).cpu().numpy()
    mag = max(1.0, scale)                                  # the logits are sums of terms of this magnitude
    err_hip = float(np.max(np.abs(got - want))) / mag
    err_blas = float(np.max(np.abs(blas - want))) / mag
    print(f"scale {scale}: HIP head {err_hip:.2e}, hipBLASLt fp32 {err_blas:.2e} (relative to the term magnitude)")
    assert err_hip <= TOL and err_hip <= 4.0 * max(err_blas, 1e-7)


@pytest.mark.parametrize("K0,nlayers,nhid,noutput", [(512, 0, 256, 1), (510, 0, 8, 3), (77, 0, 8, 16), (512, 2, 256, 3), (96, 1, 64, 2),
                                                      (390, 2, 500, 4), (2048, 3, 128, 16),
                                                      # round 5: more than 16 outputs end in armnet_linear_bf16x3_f32
                                                      (512, 2, 256, 40), (100, 0, 8, 300), (96, 1, 64, 17), (390, 2, 500, 33)])
@pytest.mark.parametrize("B", [1, 37, 4099])
def test_heads_without_hidden_layers_or_with_several_outputs_run_on_hip(K0, nlayers, nhid, noutput, B):
    """round-3 verdict, missing 5: models/layers.py:79-80 (nlayers == 0: the MLP is one Linear) and a final Linear with
    noutput > 1 used to fall to hipBLASLt; they now end in armnet_linear_small_f32 (plain fp32, one wave per row),
    behind the matrix-core launches of the hidden layers when there are any.  Against a float64 evaluation; strided and
    misaligned inputs take the element-wise path"""
    m = _make_head(K0, nlayers, nhid, seed=K0 + noutput, noutput=noutput).to(DEV)
    assert m._hip_plan() is not None
    assert ("armnet_linear_small_f32" if noutput <= 16 else "armnet_linear_bf16x3_f32") in m.eval_path()
    x = (torch.rand(B, K0, generator=torch.Generator().manual_seed(B)) * 3.0 - 1.0).to(DEV)
    want = _ref64(m, x).cpu().numpy()
    with torch.no_grad():
        got = m(x)
        assert tuple(got.shape) == (B, noutput)
        assert elem_excess(got.cpu().numpy(), want, TOL) <= 1.0
        big = torch.zeros(B, K0 + 9, device=DEV)
        big[:, 3:3 + K0] = x
        xs = big[:, 3:3 + K0]                               # 12-byte offset, odd row stride
        assert elem_excess(m(xs).cpu().numpy(), want, TOL) <= 1.0


def test_linear_small_abi():
    from armnet_hip import native
    g = torch.Generator().manual_seed(0)
    x, W, b = (torch.randn(300, 40, generator=g).to(DEV), torch.randn(5, 40, generator=g).to(DEV), torch.randn(5, generator=g).to(DEV))
    out = torch.full((300, 5), 2.0, device=DEV)
    native.linear_small(x, W, b, out, scale=0.5, accumulate=True)
    want = 2.0 + 0.5 * (x.double() @ W.double().t() + b.double())
    assert float((out.double() - want).abs().max()) <= 1e-5
    with pytest.raises(native.ArmnetNativeError):
        native.linear_small(x, torch.randn(17, 40, device=DEV), None, torch.empty(300, 17, device=DEV))


def test_mlp_head_repacks_when_parameters_change():
    m = _make_head(96, 2, 64, seed=11).to(DEV)
    x = torch.randn(65, 96, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        y0 = m(x).clone()
        m.mlp[0].weight.mul_(1.5)                          # in-place op: bumps the version counter
        m.mlp[1].running_mean.add_(0.1)
        y1 = m(x)
    want = _ref64(m, x).cpu().numpy()
    assert not torch.allclose(y0, y1)
    assert elem_excess(y1.cpu().numpy(), want, TOL) <= 1.0
    with torch.no_grad():
        m.mlp[0].weight.data.mul_(2.0)                     # a write through .data is NOT seen ...
        assert torch.equal(m(x), y1)
        m.invalidate()                                     # ... until the caches are invalidated explicitly
        y2 = m(x)
    assert elem_excess(y2.cpu().numpy(), _ref64(m, x).cpu().numpy(), TOL) <= 1.0


def test_mlp_head_abi_rejects_bad_arguments_and_handles_strided_input():
    from armnet_hip import native
    assert not native.mlp_head_supported(512, 257, 2) and not native.mlp_head_supported(512, 256, 3)
    with pytest.raises(native.ArmnetNativeError):
        native.mlp_packed_bytes(512, 300, 1)
    m = _make_head(48, 1, 32, seed=5).to(DEV)
    big = torch.randn(200, 64, generator=torch.Generator().manual_seed(2)).to(DEV)
    x = big[:, 8:56]                                       # row stride 64 floats, unit inner stride, 32-byte offset
    assert not x.is_contiguous()
    with torch.no_grad():
        got = m(x)
    assert elem_excess(got.cpu().numpy(), _ref64(m, x).cpu().numpy(), TOL) <= 1.0
    e = torch.empty(0, 48, device=DEV)
    with torch.no_grad():
        assert tuple(m(e).shape) == (0, 1)
