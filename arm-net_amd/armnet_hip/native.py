"""ctypes binding of libarmnet_hip.so (the C ABI declared in include/armnet_hip.h).

There is NO CPU fallback here: if the HIP library cannot be loaded the import of
the product path fails loudly.  Tensors are passed as raw device pointers and
the current torch HIP stream; PyTorch is used only for memory and streams.
"""
import ctypes
import os
import subprocess

import torch  # noqa: F401  (must be imported first: the library binds to torch's libamdhip64.so.7)

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("ARMNET_HIP_LIB", os.path.join(_PKG, "lib", "libarmnet_hip.so"))  # env override: A/B builds
CSRC = os.path.join(_PKG, "csrc")
ABI_VERSION = 7

OK, ERR_BAD_ARG, ERR_UNSUPPORTED, ERR_ID_RANGE, ERR_HIP = 0, -1, -2, -3, -4
ID_I64, ID_I32 = 0, 1
ONE_HEAD, MULTI_HEAD, GC_ARM = 0, 1, 2
F_WRITE_CLAMPED_VALS, F_FAITHFUL_BISECT, F_FORCE_GENERIC, F_NO_LIN_FINISH, F_FP32_CONTRACTIONS = 0x1, 0x2, 0x4, 0x8, 0x10

EXPORTS = (
    "armnet_abi_version", "armnet_strerror", "armnet_last_hip_error", "armnet_fold_params_f32",
    "armnet_fused_fwd_f32", "armnet_fused_fwd_from_rows_f32", "armnet_gather_scale_f32",
    "armnet_clamp_vals_f32", "armnet_entmax_f32", "armnet_shard_route_ws_bytes", "armnet_shard_route_ids",
    "armnet_fused_bwd_f32", "armnet_shard_route_unique_ws_bytes", "armnet_shard_route_unique_ids",
    "armnet_fused_kernel_kind", "armnet_fused_bwd_bn_f32", "armnet_bn_stats_f32", "armnet_bn_finalize_f32",
    "armnet_bn_apply_f32", "armnet_bn_bwd_reduce_f32", "armnet_bn_bwd_coef_f32", "armnet_bn_bwd_apply_f32",
    "armnet_scatter_add_f32", "armnet_mlp_head_supported", "armnet_mlp_packed_bytes", "armnet_mlp_pack_layer_f32",
    "armnet_mlp_head_f32", "armnet_gc_fused_fwd_f32", "armnet_afn_fused_fwd_f32", "armnet_fold_bn_f32",
    "armnet_abs_clamp_min_f32", "armnet_shard_pad_route", "armnet_shard_direct_perm",
    "armnet_shard_route_fixed_ws_bytes", "armnet_shard_route_fixed", "armnet_shard_route_fixed_perm",
    "armnet_linear_small_f32", "armnet_entmax_bwd_f32", "armnet_gc_fused_bwd_supported", "armnet_gc_fused_bwd_f32",
    "armnet_afn_fused_bwd_supported", "armnet_afn_fused_bwd_f32", "armnet_bn_bwd_scatter_f32",
    "armnet_gather_map_stats_f32", "armnet_shard_gather_perm_f32",
    "armnet_shard_route_fixed_epoch",
    "armnet_shard_route_fixed_hot", "armnet_shard_route_fixed_perm_hot", "armnet_shard_gather_perm_hot_f32",
    "armnet_linear_bf16x3_f32", "armnet_mlp_head_ex_f32", "armnet_entmax_rows_f32", "armnet_sibling_kernel_kind",
)

_lib = None


class ArmnetNativeError(RuntimeError):
    pass


def build(verbose=False):
    """Compile arm-net_amd/csrc for gfx950 with hipcc (cross-compiles without a GPU)."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-C", CSRC, "-j8"], stdout=out)
    return LIB_PATH


def load():
    """Load the HIP library or raise.  Never substitutes anything else."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ArmnetNativeError(
            f"{LIB_PATH} is missing: the ARM-Net HIP kernels are not built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {CSRC}`). "
            "There is no CPU fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise ArmnetNativeError(f"{LIB_PATH} does not export {name}")
    lib.armnet_strerror.restype = ctypes.c_char_p
    lib.armnet_shard_route_ws_bytes.restype = ctypes.c_int64
    lib.armnet_shard_route_unique_ws_bytes.restype = ctypes.c_int64
    lib.armnet_shard_route_fixed_ws_bytes.restype = ctypes.c_int64
    lib.armnet_last_hip_error.restype = ctypes.c_char_p
    lib.armnet_mlp_packed_bytes.restype = ctypes.c_int64
    if lib.armnet_abi_version() != ABI_VERSION:
        raise ArmnetNativeError(f"ABI version mismatch: library {lib.armnet_abi_version()} != binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    lib = load()
    msg = lib.armnet_strerror(rc).decode()
    if rc == ERR_HIP:
        msg += ": " + lib.armnet_last_hip_error().decode()
    if rc == ERR_ID_RANGE:
        raise IndexError(msg)
    raise ArmnetNativeError(f"armnet_hip call failed ({rc}): {msg}")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _on:
    """Device guard of one native call: every tensor argument must live on ONE HIP device; that device is made
    current for the duration of the call, so that `_stream()`, the library's CU-count query and its
    hipFuncSetAttribute calls all act on the device that owns the pointers (torch ops get this from their own
    DeviceGuard; raw ctypes launches do not)."""

    def __init__(self, *tensors):
        dev = None
        for t in tensors:
            if t is None:
                continue
            if not t.is_cuda and t.is_pinned():
                continue        # pinned host memory is mapped into every device (block.IdStatus' deferred flag word)
            if not t.is_cuda:
                raise ArmnetNativeError(f"expected a tensor on the HIP device, got one on {t.device} (no CPU fallback)")
            if dev is None:
                dev = t.device
            elif t.device != dev:
                raise ArmnetNativeError(f"tensor arguments live on different devices ({dev} and {t.device})")
        if dev is None:
            raise ArmnetNativeError("no device tensor among the arguments")
        self._ctx = torch.cuda.device(dev)

    def __enter__(self):
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        return self._ctx.__exit__(*exc)


def _dev_f32(t, name):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise ArmnetNativeError(f"{name}: expected a contiguous float32 tensor on the HIP device, got "
                                f"{t.dtype} on {t.device} (contiguous={t.is_contiguous()})")
    return t


def _id_type(ids):
    if ids.dtype == torch.int64:
        return ID_I64
    if ids.dtype == torch.int32:
        return ID_I32
    raise ArmnetNativeError(f"ids must be int64 or int32, got {ids.dtype}")


def fold_params(variant, K, H, E, D, bilinear_w, query, bn_w, bn_b, bn_mean, bn_var, eps,
                q_fold, bn_scale, bn_shift):
    ts = (bilinear_w, query, bn_w, bn_b, bn_mean, bn_var, q_fold, bn_scale, bn_shift)
    for n, t in zip(("bilinear_w", "query", "bn_w", "bn_b", "bn_mean", "bn_var", "q_fold", "bn_scale", "bn_shift"), ts):
        _dev_f32(t, n)
    with _on(*ts):
        check(load().armnet_fold_params_f32(variant, K, H, E, D, _ptr(bilinear_w), _ptr(query), _ptr(bn_w),
                                            _ptr(bn_b), _ptr(bn_mean), _ptr(bn_var), ctypes.c_float(eps),
                                            _ptr(q_fold), _ptr(bn_scale), _ptr(bn_shift), _stream()))


def fused_kernel_kind(F, E, O, alpha, n_iter=50, flags=0):
    """1 = the matrix-core kernel serves this shape, 0 = the generic kernel (host-only query)"""
    rc = load().armnet_fused_kernel_kind(int(F), int(E), int(O), ctypes.c_float(alpha), int(n_iter),
                                         ctypes.c_uint32(flags))
    if rc < 0:
        check(rc)
    return rc


def sibling_kernel_kind(afn, F, E, O):
    """1: GC-ARM's (afn = False) / AFN's (afn = True) fused forward runs on the matrix-core kernel for this block shape, 0: on
    the shape-agnostic one"""
    rc = load().armnet_sibling_kernel_kind(int(bool(afn)), int(F), int(E), int(O))
    if rc < 0:
        check(rc)
    return rc


def _ids_ok(ids):
    if not (ids.is_cuda and ids.is_contiguous()):
        raise ArmnetNativeError("ids must be a contiguous tensor on the HIP device")
    return ids


def fused_fwd(B, F, E, O, alpha, n_iter, flags, ids, vals, table, q_fold, values, bn_scale, bn_shift, out,
              id_status=None):
    _ids_ok(ids)
    ts = (vals, table, q_fold, values, bn_scale, bn_shift, out)
    for n, t in zip(("vals", "table", "q_fold", "values", "bn_scale", "bn_shift", "out"), ts):
        _dev_f32(t, n)
    with _on(ids, id_status, *ts):
        check(load().armnet_fused_fwd_f32(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                          ctypes.c_uint32(flags), _ptr(ids), _id_type(ids), _ptr(vals), _ptr(table),
                                          ctypes.c_int64(table.shape[0]), _ptr(q_fold), _ptr(values),
                                          _ptr(bn_scale), _ptr(bn_shift), _ptr(out), _ptr(id_status), _stream()))


def fused_fwd_from_rows(B, F, E, O, alpha, n_iter, flags, rows, vals, q_fold, values, bn_scale, bn_shift, out):
    ts = (rows, vals, q_fold, values, bn_scale, bn_shift, out)
    for n, t in zip(("rows", "vals", "q_fold", "values", "bn_scale", "bn_shift", "out"), ts):
        _dev_f32(t, n)
    with _on(*ts):
        check(load().armnet_fused_fwd_from_rows_f32(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                                    ctypes.c_uint32(flags), _ptr(rows), _ptr(vals), _ptr(q_fold),
                                                    _ptr(values), _ptr(bn_scale), _ptr(bn_shift), _ptr(out),
                                                    _stream()))


def gather_scale(n_rows, E, ids, vals, table, out, id_status=None):
    _ids_ok(ids)
    _dev_f32(table, "table"); _dev_f32(out, "out")
    if vals is not None:
        _dev_f32(vals, "vals")
    with _on(ids, vals, table, out, id_status):
        check(load().armnet_gather_scale_f32(ctypes.c_int64(n_rows), E, _ptr(ids), _id_type(ids), _ptr(vals),
                                             _ptr(table), ctypes.c_int64(table.shape[0]), _ptr(out),
                                             _ptr(id_status), _stream()))


def clamp_vals(vals):
    _dev_f32(vals, "vals")
    with _on(vals):
        check(load().armnet_clamp_vals_f32(_ptr(vals), ctypes.c_int64(vals.numel()), _stream()))


def entmax_rows(rows, d, alpha_rows, n_iter, ensure_sum_one, X, P):
    """armnet_entmax_rows_f32: one alpha per row (alpha_rows [rows] float32 on the device, all > 1), the reference's bisection"""
    _dev_f32(X, "X"); _dev_f32(P, "P"); _dev_f32(alpha_rows, "alpha_rows")
    with _on(X, P, alpha_rows):
        check(load().armnet_entmax_rows_f32(ctypes.c_int64(rows), d, _ptr(alpha_rows), int(n_iter), int(bool(ensure_sum_one)),
                                            _ptr(X), _ptr(P), _stream()))


def entmax(rows, d, alpha, n_iter, ensure_sum_one, flags, X, P):
    _dev_f32(X, "X"); _dev_f32(P, "P")
    with _on(X, P):
        check(load().armnet_entmax_f32(ctypes.c_int64(rows), d, ctypes.c_float(alpha), int(n_iter),
                                       int(bool(ensure_sum_one)), ctypes.c_uint32(flags), _ptr(X), _ptr(P), _stream()))


def entmax_bwd(rows, d, alpha, Y, dY, dX):
    _dev_f32(Y, "Y"); _dev_f32(dY, "dY"); _dev_f32(dX, "dX")
    with _on(Y, dY, dX):
        check(load().armnet_entmax_bwd_f32(ctypes.c_int64(rows), int(d), ctypes.c_float(alpha), _ptr(Y), _ptr(dY), _ptr(dX),
                                           _stream()))


def shard_route_ws_bytes(n, R):
    return int(load().armnet_shard_route_ws_bytes(ctypes.c_int64(n), int(R)))


def _i32_ok(**ts):
    for name, t in ts.items():
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise ArmnetNativeError(f"{name}: expected a contiguous int32 tensor on the HIP device")


def shard_route_ids(n, ids, R, nfeat, counts, send_local, perm, workspace, id_status=None):
    _ids_ok(ids)
    _i32_ok(counts=counts, send_local=send_local, perm=perm)
    with _on(ids, counts, send_local, perm, workspace, id_status):
        check(load().armnet_shard_route_ids(ctypes.c_int64(n), _ptr(ids), _id_type(ids), int(R), ctypes.c_int64(nfeat),
                                            _ptr(counts), _ptr(send_local), _ptr(perm), _ptr(workspace),
                                            ctypes.c_int64(workspace.numel() * workspace.element_size()),
                                            _ptr(id_status), _stream()))


def fused_bwd(B, F, E, O, alpha, n_iter, flags, ids, vals, table, q_fold, values, z, dz, d_table, d_values, d_qfold):
    _ids_ok(ids)
    ts = (vals, table, q_fold, values, z, dz, d_table, d_values, d_qfold)
    for n, t in zip(("vals", "table", "q_fold", "values", "z", "dz", "d_table", "d_values", "d_qfold"), ts):
        _dev_f32(t, n)
    with _on(ids, *ts):
        check(load().armnet_fused_bwd_f32(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                          ctypes.c_uint32(flags), _ptr(ids), _id_type(ids), _ptr(vals), _ptr(table),
                                          ctypes.c_int64(table.shape[0]), _ptr(q_fold), _ptr(values), _ptr(z), _ptr(dz),
                                          _ptr(d_table), _ptr(d_values), _ptr(d_qfold), _stream()))


def scatter_add(ids, vals, grad, d_table):
    """d_table[ids[r], :] += grad[r, :] * vals[r] (backward of gather_scale with respect to the table)"""
    _ids_ok(ids)
    _dev_f32(grad, "grad"); _dev_f32(d_table, "d_table")
    if vals is not None:
        _dev_f32(vals, "vals")
    n, E = ids.numel(), d_table.shape[1]
    with _on(ids, vals, grad, d_table):
        check(load().armnet_scatter_add_f32(ctypes.c_int64(n), E, _ptr(ids), _id_type(ids), _ptr(vals), _ptr(grad),
                                            ctypes.c_int64(d_table.shape[0]), _ptr(d_table), _stream()))


def fused_bwd_bn(B, F, E, O, alpha, n_iter, flags, ids, vals, table, q_fold, values, z, dy, coefA, coefB, coefC,
                 d_table, d_values, d_qfold):
    _ids_ok(ids)
    ts = (vals, table, q_fold, values, z, dy, coefA, coefB, coefC, d_table, d_values, d_qfold)
    for n, t in zip(("vals", "table", "q_fold", "values", "z", "dy", "coefA", "coefB", "coefC", "d_table", "d_values",
                     "d_qfold"), ts):
        _dev_f32(t, n)
    with _on(ids, *ts):
        check(load().armnet_fused_bwd_bn_f32(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                             ctypes.c_uint32(flags), _ptr(ids), _id_type(ids), _ptr(vals), _ptr(table),
                                             ctypes.c_int64(table.shape[0]), _ptr(q_fold), _ptr(values), _ptr(z),
                                             _ptr(dy), _ptr(coefA), _ptr(coefB), _ptr(coefC),
                                             _ptr(d_table), _ptr(d_values), _ptr(d_qfold), _stream()))


def gc_fused_bwd_supported(F, E, O):
    return bool(load().armnet_gc_fused_bwd_supported(int(F), int(E), int(O)))


def gc_fused_bwd(B, F, E, O, alpha, n_iter, flags, ids, vals, table, q_fold, values, emb_scale, emb_shift, z, dy,
                 coefA, coefB, coefC, d_table, d_values, d_qfold, d_y):
    """armnet_gc_fused_bwd_f32 (include/armnet_hip.h): GC-ARM's block backward; coefA/B/C all None or all tensors"""
    _ids_ok(ids)
    ts = (vals, table, q_fold, values, emb_scale, emb_shift, z, dy, d_table, d_values, d_qfold, d_y)
    for n, t in zip(("vals", "table", "q_fold", "values", "emb_scale", "emb_shift", "z", "dy", "d_table", "d_values",
                     "d_qfold", "d_y"), ts):
        _dev_f32(t, n)
    coefs = (coefA, coefB, coefC)
    if any(c is not None for c in coefs):
        if any(c is None for c in coefs):
            raise ArmnetNativeError("coefA / coefB / coefC: all three (armnet_bn_bwd_coef_f32) or none")
        for n, t in zip(("coefA", "coefB", "coefC"), coefs):
            _dev_f32(t, n)
    if d_y.numel() != B * F * E:
        raise ArmnetNativeError(f"d_y must hold B*F*E = {B * F * E} floats, got {d_y.numel()}")
    with _on(ids, *ts):
        check(load().armnet_gc_fused_bwd_f32(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                             ctypes.c_uint32(flags), _ptr(ids), _id_type(ids), _ptr(vals), _ptr(table),
                                             ctypes.c_int64(table.shape[0]), _ptr(q_fold), _ptr(values), _ptr(emb_scale),
                                             _ptr(emb_shift), _ptr(z), _ptr(dy), _ptr(coefA), _ptr(coefB), _ptr(coefC),
                                             _ptr(d_table), _ptr(d_values), _ptr(d_qfold), _ptr(d_y), _stream()))


def _ncl(x):
    if x.dim() == 2:
        return x.shape[0], x.shape[1], 1
    if x.dim() == 3:
        return x.shape[0], x.shape[1], x.shape[2]
    raise ArmnetNativeError(f"BatchNorm1d input must be 2-D or 3-D, got {x.dim()}-D")


def bn_forward_train(x, weight, bias, running_mean, running_var, momentum, eps, relu):
    """training-mode BatchNorm1d forward (+ optional ReLU): returns y, mean, rstd, scale, shift"""
    _dev_f32(x, "x")
    N, C, L = _ncl(x)
    with _on(x, weight, bias, running_mean, running_var):
        buf = torch.zeros(6, C, device=x.device, dtype=torch.float32)      # stats[2], mean, rstd, scale, shift
        st = _stream()
        lib = load()
        check(lib.armnet_bn_stats_f32(ctypes.c_int64(N), C, L, _ptr(x), _ptr(buf), st))
        check(lib.armnet_bn_finalize_f32(C, ctypes.c_int64(N * L), _ptr(buf), _ptr(x), L, _ptr(weight), _ptr(bias),
                                         ctypes.c_float(eps), ctypes.c_float(momentum), _ptr(running_mean),
                                         _ptr(running_var), _ptr(buf[2]), _ptr(buf[3]), _ptr(buf[4]), _ptr(buf[5]), st))
        y = torch.empty_like(x)
        check(lib.armnet_bn_apply_f32(ctypes.c_int64(N), C, L, _ptr(x), _ptr(buf[4]), _ptr(buf[5]), int(bool(relu)),
                                      _ptr(y), st))
    return y, buf[2], buf[3], buf[4], buf[5]


def afn_fused_bwd_supported(F, E, O):
    return bool(load().armnet_afn_fused_bwd_supported(int(F), int(E), int(O)))


def afn_fused_bwd(B, F, E, O, flags, ids, vals, table, weight, emb_scale, emb_shift, z, dy, coefA, coefB, coefC, d_weight,
                  d_bias, d_y):
    """armnet_afn_fused_bwd_f32 (include/armnet_hip.h): AFN's block backward; coefA/B/C all None or all tensors"""
    _ids_ok(ids)
    ts = (vals, table, weight, emb_scale, emb_shift, z, dy, d_weight, d_bias, d_y)
    for n, t in zip(("vals", "table", "weight", "emb_scale", "emb_shift", "z", "dy", "d_weight", "d_bias", "d_y"), ts):
        _dev_f32(t, n)
    if any(c is not None for c in (coefA, coefB, coefC)):
        if any(c is None for c in (coefA, coefB, coefC)):
            raise ArmnetNativeError("coefA / coefB / coefC: all three (armnet_bn_bwd_coef_f32) or none")
        for n, t in zip(("coefA", "coefB", "coefC"), (coefA, coefB, coefC)):
            _dev_f32(t, n)
    if d_y.numel() != B * F * E:
        raise ArmnetNativeError(f"d_y must hold B*F*E = {B * F * E} floats, got {d_y.numel()}")
    with _on(ids, *ts):
        check(load().armnet_afn_fused_bwd_f32(ctypes.c_int64(B), F, E, O, ctypes.c_uint32(flags), _ptr(ids), _id_type(ids),
                                              _ptr(vals), _ptr(table), ctypes.c_int64(table.shape[0]), _ptr(weight),
                                              _ptr(emb_scale), _ptr(emb_shift), _ptr(z), _ptr(dy), _ptr(coefA), _ptr(coefB),
                                              _ptr(coefC), _ptr(d_weight), _ptr(d_bias), _ptr(d_y), _stream()))


def bn_bwd_scatter(ids, vals, t, dy, coefA, coefB, coefC, map_kind, d_table):
    """armnet_bn_bwd_scatter_f32: t [B,F,E] = exp(x) (map_kind 0) or log(x) (1), dy its BatchNorm output's gradient"""
    _ids_ok(ids)
    ts = (vals, t, dy, coefA, coefB, coefC, d_table)
    for n, x in zip(("vals", "t", "dy", "coefA", "coefB", "coefC", "d_table"), ts):
        _dev_f32(x, n)
    B, F, E = t.shape
    if dy.shape != t.shape or vals.numel() != B * F or ids.numel() != B * F or d_table.shape[1] != E:
        raise ArmnetNativeError("bn_bwd_scatter: shapes of ids / vals / t / dy / d_table disagree")
    with _on(ids, *ts):
        check(load().armnet_bn_bwd_scatter_f32(ctypes.c_int64(B * F), F, E, _ptr(ids), _id_type(ids), _ptr(vals), _ptr(t),
                                               _ptr(dy), _ptr(coefA), _ptr(coefB), _ptr(coefC), int(map_kind),
                                               ctypes.c_int64(d_table.shape[0]), _ptr(d_table), _stream()))


def bn_train_stats(x, weight, bias, running_mean, running_var, momentum, eps, stats=None):
    """the statistics half of bn_forward_train (running statistics updated, nothing normalised): mean, rstd, scale,
    shift of THIS batch — for a consumer that applies the affine itself (the siblings' fused blocks, siblings.py).
    `stats`: a [6, C] buffer whose first two rows already hold the shifted sums of x (gather_map_stats); None = run the pass"""
    _dev_f32(x, "x")
    N, C, L = _ncl(x)
    with _on(x, weight, bias, running_mean, running_var):
        st = _stream()
        lib = load()
        if stats is None:
            buf = torch.zeros(6, C, device=x.device, dtype=torch.float32)
            check(lib.armnet_bn_stats_f32(ctypes.c_int64(N), C, L, _ptr(x), _ptr(buf), st))
        else:
            buf = stats
        check(lib.armnet_bn_finalize_f32(C, ctypes.c_int64(N * L), _ptr(buf), _ptr(x), L, _ptr(weight), _ptr(bias),
                                         ctypes.c_float(eps), ctypes.c_float(momentum), _ptr(running_mean),
                                         _ptr(running_var), _ptr(buf[2]), _ptr(buf[3]), _ptr(buf[4]), _ptr(buf[5]), st))
    return buf[2], buf[3], buf[4], buf[5]


def gather_map_stats(ids, vals, table, map_kind, id_status=None):
    """armnet_gather_map_stats_f32: returns (out [B,F,E] = exp / log of the scaled rows, the [6, F] buffer for bn_train_stats)"""
    _ids_ok(ids)
    _dev_f32(vals, "vals"); _dev_f32(table, "table")
    B, F = vals.shape
    E = table.shape[1]
    with _on(ids, vals, table, id_status):
        out = torch.empty(B, F, E, device=vals.device, dtype=torch.float32)
        buf = torch.zeros(6, F, device=vals.device, dtype=torch.float32)
        check(load().armnet_gather_map_stats_f32(ctypes.c_int64(B), F, E, _ptr(ids), _id_type(ids), _ptr(vals), _ptr(table),
                                                 ctypes.c_int64(table.shape[0]), int(map_kind), _ptr(out), _ptr(buf),
                                                 _ptr(id_status), _stream()))
    return out, buf


def bn_backward_coef(x, dy, weight, mean, rstd, relu_scale=None, relu_shift=None):
    """backward reductions of training-mode BatchNorm1d: returns d_weight, d_bias, coefA, coefB, coefC with
    dx = coefA * dy + coefC * x + coefB (dy masked by the recomputed ReLU when relu_scale/shift are given)"""
    _dev_f32(x, "x"); _dev_f32(dy, "dy")
    N, C, L = _ncl(x)
    with _on(x, dy, weight, mean, rstd, relu_scale, relu_shift):
        buf = torch.zeros(7, C, device=x.device, dtype=torch.float32)   # sums[2], d_weight, d_bias, A, B, C
        st = _stream()
        lib = load()
        check(lib.armnet_bn_bwd_reduce_f32(ctypes.c_int64(N), C, L, _ptr(x), _ptr(dy), _ptr(mean), _ptr(rstd),
                                           _ptr(relu_scale), _ptr(relu_shift), _ptr(buf), st))
        check(lib.armnet_bn_bwd_coef_f32(C, ctypes.c_int64(N * L), _ptr(buf), _ptr(weight), _ptr(mean), _ptr(rstd),
                                         _ptr(buf[2]), _ptr(buf[3]), _ptr(buf[4]), _ptr(buf[5]), _ptr(buf[6]), st))
    return buf[2], buf[3], buf[4], buf[5], buf[6]


def bn_backward_apply(x, dy, coefA, coefB, coefC, relu_scale=None, relu_shift=None):
    N, C, L = _ncl(x)
    with _on(x, dy, coefA, coefB, coefC, relu_scale, relu_shift):
        dx = torch.empty_like(x)
        check(load().armnet_bn_bwd_apply_f32(ctypes.c_int64(N), C, L, _ptr(x), _ptr(dy), _ptr(coefA), _ptr(coefB),
                                             _ptr(coefC), _ptr(relu_scale), _ptr(relu_shift), _ptr(dx), _stream()))
    return dx


def shard_route_unique_ws_bytes(R, nfeat):
    return int(load().armnet_shard_route_unique_ws_bytes(int(R), ctypes.c_int64(nfeat)))


def shard_route_unique_ids(n, ids, R, nfeat, counts, send_local, perm, workspace, id_status=None):
    _ids_ok(ids)
    _i32_ok(counts=counts, send_local=send_local, perm=perm)
    with _on(ids, counts, send_local, perm, workspace, id_status):
        check(load().armnet_shard_route_unique_ids(ctypes.c_int64(n), _ptr(ids), _id_type(ids), int(R),
                                                   ctypes.c_int64(nfeat), _ptr(counts), _ptr(send_local), _ptr(perm),
                                                   ctypes.c_void_p(0), _ptr(workspace),
                                                   ctypes.c_int64(workspace.numel() * workspace.element_size()),
                                                   _ptr(id_status), _stream()))


def mlp_head_supported(K0, nhid, n_hidden):
    return bool(load().armnet_mlp_head_supported(int(K0), int(nhid), int(n_hidden)))


def mlp_packed_bytes(K0, nhid, n_hidden):
    n = int(load().armnet_mlp_packed_bytes(int(K0), int(nhid), int(n_hidden)))
    if n < 0:
        raise ArmnetNativeError(f"no MLP-head kernel for input width {K0}, hidden width {nhid}, {n_hidden} hidden layers")
    return n


def mlp_pack_layer(K0, nhid, n_hidden, slot, W, b, bn, packed):
    """bn: None or (weight, bias, running_mean, running_var, eps)"""
    _dev_f32(W, "W")
    ts = [W, b, packed] + (list(bn[:4]) if bn is not None else [])
    with _on(*ts):
        bw, bb, bm, bv = (bn[0], bn[1], bn[2], bn[3]) if bn is not None else (None, None, None, None)
        check(load().armnet_mlp_pack_layer_f32(int(K0), int(nhid), int(n_hidden), int(slot), _ptr(W), int(W.shape[1]),
                                               _ptr(b), _ptr(bw), _ptr(bb), _ptr(bm), _ptr(bv),
                                               ctypes.c_float(bn[4] if bn is not None else 0.0), _ptr(packed),
                                               _stream()))


def linear_small(x, W, bias, out, scale=1.0, accumulate=False):
    """out[b, n] (+)= (bias[n] + x[b, :] . W[n, :]) * scale for N <= 16 outputs (armnet_linear_small_f32)"""
    _dev_f32(W, "W")
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise ArmnetNativeError("x: expected a float32 [B, K] tensor with unit inner stride on the HIP device")
    if not (out.is_cuda and out.dtype == torch.float32 and out.dim() == 2 and out.stride(1) == 1):
        raise ArmnetNativeError("out: expected a float32 [B, N] tensor with unit inner stride on the HIP device")
    B, K = x.shape
    N = W.shape[0]
    ldx = x.stride(0) if B > 1 else max(x.stride(0), K)
    ldo = out.stride(0) if B > 1 else max(out.stride(0), N)
    with _on(x, W, bias, out):
        check(load().armnet_linear_small_f32(ctypes.c_int64(B), int(K), int(N), _ptr(x), ctypes.c_int64(ldx), _ptr(W),
                                             _ptr(bias), ctypes.c_float(scale), _ptr(out), ctypes.c_int64(ldo),
                                             int(bool(accumulate)), _stream()))


MLP_F_BF16X3 = 0x1


def mlp_head(B, K0, nhid, n_hidden, has_final, x, packed, out, flags=0):
    """x: [B, >= K0] float32 with unit inner stride whose row stride covers 16 * ceil(K0 / 16) floats (columns past K0
    readable and finite); out: [B] logits (has_final: 1 = write, 2 = add this slice's share) or [B, >= nhid] hidden
    activations (a column slice of a wider buffer qualifies)"""
    if not (out.is_cuda and out.dtype == torch.float32 and (out.dim() == 1 or out.stride(1) == 1)):
        raise ArmnetNativeError("out: expected a float32 tensor with unit inner stride on the HIP device")
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1):
        raise ArmnetNativeError("x: expected a float32 [B, K0] tensor with unit inner stride on the HIP device")
    ldx = x.stride(0) if B > 1 else max(x.stride(0), x.shape[1])
    ldo = 0 if has_final else (out.stride(0) if B > 1 else max(out.stride(0), out.shape[1]))
    with _on(x, packed, out):
        check(load().armnet_mlp_head_ex_f32(ctypes.c_int64(B), int(K0), int(nhid), int(n_hidden), int(has_final),
                                            _ptr(x), ctypes.c_int64(ldx), _ptr(packed), _ptr(out), ctypes.c_int64(ldo),
                                            ctypes.c_uint32(flags), _stream()))


def linear_bf16x3(x, packed, out, K, N):
    """out[:, :N] = bias + x[:, :K] @ W.T on the head's bf16x3 matrix-core path (armnet_linear_bf16x3_f32); `packed` from
    mlp_pack_layer(K, N, 1, 0, W, bias, None, packed).  x / out: float32, unit inner stride, row strides >= the padded widths"""
    for t, n in ((x, "x"), (out, "out")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1):
            raise ArmnetNativeError(f"{n}: expected a float32 2-d tensor with unit inner stride on the HIP device")
    B = x.shape[0]
    ldx = x.stride(0) if B > 1 else max(x.stride(0), x.shape[1])
    ldo = out.stride(0) if B > 1 else max(out.stride(0), out.shape[1])
    with _on(x, packed, out):
        check(load().armnet_linear_bf16x3_f32(ctypes.c_int64(B), int(K), int(N), _ptr(x), ctypes.c_int64(ldx), _ptr(packed),
                                              _ptr(out), ctypes.c_int64(ldo), _stream()))


def gc_fused_fwd(B, F, E, O, alpha, n_iter, flags, ids, vals, table, q_fold, values, emb_scale, emb_shift, bn_scale,
                 bn_shift, out, id_status=None):
    _ids_ok(ids)
    ts = (vals, table, q_fold, values, emb_scale, emb_shift, bn_scale, bn_shift, out)
    for n, t in zip(("vals", "table", "q_fold", "values", "emb_scale", "emb_shift", "bn_scale", "bn_shift", "out"), ts):
        _dev_f32(t, n)
    with _on(ids, id_status, *ts):
        check(load().armnet_gc_fused_fwd_f32(ctypes.c_int64(B), F, E, O, ctypes.c_float(alpha), int(n_iter),
                                             ctypes.c_uint32(flags), _ptr(ids), _id_type(ids), _ptr(vals), _ptr(table),
                                             ctypes.c_int64(table.shape[0]), _ptr(q_fold), _ptr(values),
                                             _ptr(emb_scale), _ptr(emb_shift), _ptr(bn_scale), _ptr(bn_shift),
                                             _ptr(out), _ptr(id_status), _stream()))


def afn_fused_fwd(B, F, E, O, flags, ids, vals, table, weight, bias, emb_scale, emb_shift, bn_scale, bn_shift, out,
                  id_status=None):
    _ids_ok(ids)
    ts = (vals, table, weight, bias, emb_scale, emb_shift, bn_scale, bn_shift, out)
    for n, t in zip(("vals", "table", "weight", "bias", "emb_scale", "emb_shift", "bn_scale", "bn_shift", "out"), ts):
        _dev_f32(t, n)
    with _on(ids, id_status, *ts):
        check(load().armnet_afn_fused_fwd_f32(ctypes.c_int64(B), F, E, O, ctypes.c_uint32(flags), _ptr(ids),
                                              _id_type(ids), _ptr(vals), _ptr(table), ctypes.c_int64(table.shape[0]),
                                              _ptr(weight), _ptr(bias), _ptr(emb_scale), _ptr(emb_shift),
                                              _ptr(bn_scale), _ptr(bn_shift), _ptr(out), _ptr(id_status), _stream()))


def fold_bn(weight, bias, mean, var, eps, scale, shift):
    ts = (weight, bias, mean, var, scale, shift)
    for n, t in zip(("weight", "bias", "mean", "var", "scale", "shift"), ts):
        _dev_f32(t, n)
    with _on(*ts):
        check(load().armnet_fold_bn_f32(int(weight.numel()), _ptr(weight), _ptr(bias), _ptr(mean), _ptr(var),
                                        ctypes.c_float(eps), _ptr(scale), _ptr(shift), _stream()))


def abs_clamp_min(t, lo):
    """t <- max(|t|, lo) in place (afn.py:74-77)"""
    _dev_f32(t, "t")
    with _on(t):
        check(load().armnet_abs_clamp_min_f32(_ptr(t), ctypes.c_int64(t.numel()), ctypes.c_float(lo), _stream()))


def shard_pad_route(n, R, cap, counts, send_local, perm, send_pad, perm_pad, overflow):
    _i32_ok(counts=counts, send_local=send_local, perm=perm, send_pad=send_pad, perm_pad=perm_pad, overflow=overflow)
    with _on(counts, send_local, perm, send_pad, perm_pad, overflow):
        check(load().armnet_shard_pad_route(ctypes.c_int64(n), int(R), ctypes.c_int64(cap), _ptr(counts),
                                            _ptr(send_local), _ptr(perm), _ptr(send_pad), _ptr(perm_pad),
                                            _ptr(overflow), _stream()))


def shard_route_fixed_ws_bytes(R, nfeat, dedup):
    return int(load().armnet_shard_route_fixed_ws_bytes(int(R), ctypes.c_int64(nfeat), int(bool(dedup))))


def shard_route_fixed(n, ids, R, nfeat, cap, dedup, send_pad, perm_pad, counts, overflow, workspace=None, id_status=None,
                      epoch=0, hot=(0, 0)):
    """routing of the fixed-capacity protocol in one call: send_pad [R*cap], perm_pad [n], counts [R], overflow flag.
    epoch (dedup only): 0 = the mark map is zeroed by the call; 2..255 = caller-managed mark epoch, no fill
    (armnet_shard_route_fixed_epoch: valid after a call with epoch 0 or a smaller epoch on the same workspace).
    hot = (hot_rows, hot_base): ids < hot_rows are replicated on every rank and not routed, perm_pad = hot_base + id
    (armnet_shard_route_fixed_hot)"""
    _ids_ok(ids)
    _i32_ok(send_pad=send_pad, counts=counts, overflow=overflow)
    if perm_pad is not None:                 # None (dedup only): the position gather is left to shard_route_fixed_perm
        _i32_ok(perm_pad=perm_pad)
    with _on(ids, send_pad, perm_pad, counts, overflow, workspace, id_status):
        check(load().armnet_shard_route_fixed_hot(
            ctypes.c_int64(n), _ptr(ids), _id_type(ids), int(R), ctypes.c_int64(nfeat), ctypes.c_int64(cap),
            int(bool(dedup)), _ptr(send_pad), _ptr(perm_pad), _ptr(counts), _ptr(overflow), _ptr(id_status),
            _ptr(workspace), ctypes.c_int64(workspace.numel() * workspace.element_size() if workspace is not None else 0),
            int(epoch), ctypes.c_int64(hot[0]), ctypes.c_int64(hot[1]), _stream()))


def shard_gather_perm(idx, table, out, ids, R, nfeat, perm_pad, workspace, hot=(0, 0)):
    """armnet_shard_gather_perm_f32: out[j] = table[idx[j]] and perm_pad[i] = position of ids[i] (workspace of a preceding
    shard_route_fixed(dedup=True, perm_pad=None)) in one launch"""
    _ids_ok(ids)
    _dev_f32(table, "table"); _dev_f32(out, "out")
    if idx.dtype != torch.int32 or perm_pad.dtype != torch.int32:
        raise ArmnetNativeError("shard_gather_perm: idx and perm_pad must be int32")
    with _on(idx, table, out, ids, perm_pad, workspace):
        check(load().armnet_shard_gather_perm_hot_f32(ctypes.c_int64(idx.numel()), table.shape[1], _ptr(idx), _ptr(table),
                                                      ctypes.c_int64(table.shape[0]), _ptr(out), ctypes.c_int64(ids.numel()),
                                                      _ptr(ids), _id_type(ids), int(R), ctypes.c_int64(nfeat), _ptr(perm_pad),
                                                      _ptr(workspace), ctypes.c_int64(workspace.numel()),
                                                      ctypes.c_int64(hot[0]), ctypes.c_int64(hot[1]), _stream()))


def shard_route_fixed_perm(n, ids, R, nfeat, perm_pad, workspace, hot=(0, 0)):
    """perm_pad[i] = position of id i's row, from the workspace a shard_route_fixed(dedup=True, perm_pad=None) call left"""
    _ids_ok(ids)
    _i32_ok(perm_pad=perm_pad)
    with _on(ids, perm_pad, workspace):
        check(load().armnet_shard_route_fixed_perm_hot(
            ctypes.c_int64(n), _ptr(ids), _id_type(ids), int(R), ctypes.c_int64(nfeat), _ptr(perm_pad), _ptr(workspace),
            ctypes.c_int64(workspace.numel() * workspace.element_size()), ctypes.c_int64(hot[0]), ctypes.c_int64(hot[1]),
            _stream()))


def shard_direct_perm(n, ids, R, nfeat, perm, id_status=None):
    """perm[i] = (id % R) * ceil(nfeat / R) + id // R: address of id i's row in the all-gathered shards"""
    _ids_ok(ids)
    _i32_ok(perm=perm)
    with _on(ids, perm, id_status):
        check(load().armnet_shard_direct_perm(ctypes.c_int64(n), _ptr(ids), _id_type(ids), int(R),
                                              ctypes.c_int64(nfeat), _ptr(perm), _ptr(id_status), _stream()))
