"""Tensor-level wrappers over the C ABI: parameter folding cache + fused-block launch.

PyTorch here is plumbing (device memory, streams); all arithmetic of the hot path runs in the HIP kernels.  Device
tensors never leave them: there is no fallback.  A model that was never moved to the GPU, called with host tensors, runs
the reference's ATen op chain instead (host_ops.py — a dispatch on where the tensors live, like the reference's own
`models/model_utils.py:86`); a model and a batch on DIFFERENT devices raise.
"""
import torch

from . import host_ops, native


def _require_cuda(t, what):
    if not t.is_cuda:
        raise native.ArmnetNativeError(
            f"{what} is on {t.device} while the rest of the call is on the GPU: the ARM-Net HIP kernels take device "
            "tensors only and nothing is copied or computed elsewhere behind the caller's back (no CPU fallback). "
            "Move the model and the batch to the same device (model.cuda(), batch['id'].cuda(), ...).")


class IdStatus:
    """The out-of-range-id report of one module, DEFERRED — the shape of the reference's GPU behaviour: nn.Embedding on a
    device (models/layers.py:20) raises nothing at the call, a device-side assert surfaces at a later synchronisation, and
    train.py:117-121 never pays a sync for it.  One int32 word in PINNED HOST memory (mapped into every device): a kernel
    that meets an id outside [0, nfeat) stores 1 into it (the id reads row 0: memory-safe, the sample's output is
    garbage); the host reads the word WITHOUT touching the device —
        * at the module's next call (`raise_if_set`): IndexError for an earlier call whose kernels have finished,
        * at `poll()`: after synchronising the device — the answer for everything enqueued so far,
        * `check_ids = "sync"` on the module polls behind every call (the rounds 1-5 behaviour: one host sync per forward).
    No allocation, fill, copy or `.item()` per call."""

    def __init__(self):
        self._word = None

    # a copied / pickled module (copy.deepcopy for an EMA twin, torch.save(model)) gets a FRESH report of its own: the word is
    # pinned memory tied to this process, and a plain copy of it would not be pinned
    def __deepcopy__(self, memo):
        return IdStatus()

    def __getstate__(self):
        return {}

    def __setstate__(self, state):
        self._word = None

    def word(self):
        if self._word is None or not self._word.is_pinned():
            self._word = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self._word

    def raise_if_set(self):
        w = self._word
        if w is not None and int(w[0]) != 0:                 # a host-memory read
            w.zero_()
            raise IndexError("index out of range in self")

    def poll(self, device=None):
        if self._word is not None:
            torch.cuda.synchronize(device)
            self.raise_if_set()


def _id_mode(check_ids, status):
    """(flag word for the kernels | None, poll behind the call?) of a `check_ids` setting: False / None = unchecked,
    True = deferred (IdStatus), "sync" = IndexError before the call returns"""
    if not check_ids:
        return None, False
    if status is None:                                       # a stand-alone call without a module: nothing to defer to
        status, check_ids = IdStatus(), "sync"
    status.raise_if_set()
    return status, check_ids == "sync"


class ArmBlockParams:
    """Folded parameters of one ARM block (q_fold, bn_scale, bn_shift), refreshed when the
    source tensors change (tracked by their autograd version counters and storage pointers)."""

    def __init__(self):
        self.key = None
        self.q_fold = self.bn_scale = self.bn_shift = None

    def get(self, variant, K, H, E, D, bilinear_w, query, bn):
        src = (bilinear_w, query, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        key = tuple((t.data_ptr(), t._version) for t in src) + (variant, K, H, E, D, float(bn.eps))
        if key != self.key:
            dev = query.device
            O = K * H
            if self.q_fold is None or self.q_fold.shape != (O, E) or self.q_fold.device != dev:
                self.q_fold = torch.empty(O, E, device=dev, dtype=torch.float32)
                self.bn_scale = torch.empty(O, device=dev, dtype=torch.float32)
                self.bn_shift = torch.empty(O, device=dev, dtype=torch.float32)
            native.fold_params(variant, K, H, E, D, bilinear_w.detach().contiguous(), query.detach().contiguous(),
                               bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var,
                               float(bn.eps), self.q_fold, self.bn_scale, self.bn_shift)
            self.key = key
        return self.q_fold, self.bn_scale, self.bn_shift


def arm_block_forward(ids, vals, table, q_fold, values, bn_scale, bn_shift, alpha, n_iter=50,
                      write_clamped_vals=True, check_ids=True, flags=0, rows=None, out=None, status=None):
    """Fused a2..a9.  Returns out [B, O, E] (post-BN).  ``vals`` is clamped in place
    when write_clamped_vals (the reference's side effect, armnet_1h.py:81).  check_ids: the kernel's range test of the
    ids is on; an out-of-range id raises IndexError — through `status` (an IdStatus: deferred to the owner's next call /
    poll(), no host sync here) or, with check_ids == "sync" or no `status`, before this call returns (one host sync)."""
    _require_cuda(vals, "x['value']")
    if not vals.is_contiguous() or vals.dtype != torch.float32:
        raise native.ArmnetNativeError("x['value'] must be a contiguous float32 tensor (clamped in place)")
    B, F = vals.shape
    O, E = q_fold.shape
    values2d = values.detach().reshape(O, F)
    if out is None:
        out = torch.empty(B, O, E, device=vals.device, dtype=torch.float32)
    elif tuple(out.shape) != (B, O, E) or not out.is_contiguous():
        raise native.ArmnetNativeError(f"out must be a contiguous [{B}, {O}, {E}] tensor")
    fl = flags | (native.F_WRITE_CLAMPED_VALS if write_clamped_vals else 0)
    if rows is not None:
        native.fused_fwd_from_rows(B, F, E, O, alpha, n_iter, fl, rows, vals, q_fold, values2d, bn_scale,
                                   bn_shift, out)
        return out
    _require_cuda(ids, "x['id']")
    ids = ids if ids.is_contiguous() else ids.contiguous()
    status, sync = _id_mode(check_ids, status)
    native.fused_fwd(B, F, E, O, alpha, n_iter, fl, ids, vals, table.detach(), q_fold, values2d, bn_scale,
                     bn_shift, out, None if status is None else status.word())
    if sync:
        status.poll(vals.device)
    return out


def embedding_forward(ids, vals, table, check_ids=True, status=None):
    """layers.py:15-21 — table[ids] * vals.unsqueeze(2) -> [B, F, E].  check_ids / status: see arm_block_forward."""
    if host_ops.on_host(ids, table) and (vals is None or not vals.is_cuda):
        return host_ops.embedding(ids, vals, table)                  # host tensors: the reference's own ops
    _require_cuda(ids, "x['id']")
    _require_cuda(table, "the embedding table")
    shape = tuple(ids.shape)
    E = table.shape[1]
    ids_c = ids.contiguous()
    n = ids_c.numel()
    out = torch.empty(*shape, E, device=ids.device, dtype=torch.float32)
    v = None
    if vals is not None:
        v = vals.contiguous()
        if v.dtype != torch.float32:
            v = v.float()
    status, sync = _id_mode(check_ids, status)
    native.gather_scale(n, E, ids_c, v, table.detach(), out, None if status is None else status.word())
    if sync:
        status.poll(ids.device)
    return out


def _entmax_raw(X, alpha, dim, n_iter, ensure_sum_one, flags):
    if not X.is_cuda:                                                # host tensor: utils/entmax.py:29-68 on ATen ops
        with torch.no_grad():
            return host_ops.entmax_bisect(X, float(alpha), dim, n_iter, ensure_sum_one)
    if X.dtype != torch.float32:
        raise native.ArmnetNativeError(f"entmax: float32 only, got {X.dtype}")
    nd = X.dim()
    dim = dim % nd
    Xt = X.movedim(dim, -1).contiguous() if dim != nd - 1 else X.contiguous()
    d = Xt.shape[-1]
    P = torch.empty_like(Xt)
    if Xt.numel():
        native.entmax(Xt.numel() // d, d, float(alpha), n_iter, ensure_sum_one, flags, Xt, P)
    return P.movedim(-1, dim) if dim != nd - 1 else P


class _EntmaxFn(torch.autograd.Function):
    """forward: armnet_entmax_f32; backward: the Jacobian-vector product of utils/entmax.py:70-80 on the saved output
    (softmax's for alpha == 1).  alpha is a float here — a float hyper-parameter on every call path of the reference's models —;
    a TENSOR alpha (per row, or one that wants its gradient, entmax.py:82-98) takes _EntmaxRowsFn below."""

    @staticmethod
    def forward(ctx, X, alpha, dim, n_iter, ensure_sum_one, flags):
        Y = _entmax_raw(X, alpha, dim, n_iter, ensure_sum_one, flags)
        ctx.save_for_backward(Y)
        ctx.alpha, ctx.dim = float(alpha), dim
        return Y

    @staticmethod
    def backward(ctx, dY):
        Y, = ctx.saved_tensors
        d = Y.shape[ctx.dim]
        if Y.is_cuda and Y.dtype == torch.float32 and dY.dtype == torch.float32 and 2 * d * 65 * 4 <= 64 * 1024:
            # one HIP pass (armnet_entmax_bwd_f32) instead of six ATen passes over the [.., d] tensors
            nd = Y.dim()
            last = ctx.dim % nd == nd - 1
            Yc = (Y if last else Y.movedim(ctx.dim, -1)).contiguous()
            dYc = (dY if last else dY.movedim(ctx.dim, -1)).contiguous()
            dX = torch.empty_like(Yc)
            if Yc.numel():
                native.entmax_bwd(Yc.numel() // d, d, ctx.alpha, Yc, dYc, dX)
            return (dX if last else dX.movedim(-1, ctx.dim)), None, None, None, None, None
        if ctx.alpha == 1.0:
            dX = Y * (dY - (Y * dY).sum(ctx.dim, keepdim=True))
        else:
            gppr = torch.where(Y > 0, Y ** (2.0 - ctx.alpha), torch.zeros((), device=Y.device, dtype=Y.dtype))
            dX = dY * gppr
            q = dX.sum(ctx.dim, keepdim=True) / gppr.sum(ctx.dim, keepdim=True)
            dX = dX - q * gppr
        return dX, None, None, None, None, None


def _entmax_rows_raw(X, al, dim, n_iter, ensure_sum_one):
    """X and al (alpha expanded to X's shape with extent 1 along `dim`): the per-row sparse map, no autograd"""
    if not X.is_cuda:
        from . import host_ops
        with torch.no_grad():
            return host_ops.entmax_bisect(X, al, dim, n_iter, ensure_sum_one)
    if X.dtype != torch.float32:
        raise native.ArmnetNativeError(f"entmax: float32 only, got {X.dtype}")
    nd = X.dim()
    last = dim == nd - 1
    Xt = (X if last else X.movedim(dim, -1)).contiguous()
    at = (al if last else al.movedim(dim, -1)).contiguous().view(-1)
    d = Xt.shape[-1]
    P = torch.empty_like(Xt)
    if Xt.numel():
        native.entmax_rows(Xt.numel() // d, d, at, n_iter, ensure_sum_one, Xt, P)
    return P if last else P.movedim(-1, dim)


class _EntmaxRowsFn(torch.autograd.Function):
    """entmax with a TENSOR alpha, differentiable in X and in alpha (utils/entmax.py:70-98): forward = the per-row map
    (armnet_entmax_rows_f32 on the device, the reference's op chain on host tensors); backward = the reference's formulas on
    the saved output as tensor ops on whatever device the tensors live on — the Jacobian-vector product with gppr = Y^(2 - alpha)
    on the support, and the alpha gradient from the Shannon terms (entmax.py:82-98; "ensure alpha is not close to 1")."""

    @staticmethod
    def forward(ctx, X, al, dim, n_iter, ensure_sum_one):
        Y = _entmax_rows_raw(X.detach(), al.detach(), dim, n_iter, ensure_sum_one)
        ctx.save_for_backward(Y, al.detach())
        ctx.dim = dim
        return Y

    @staticmethod
    def backward(ctx, dY):
        Y, al = ctx.saved_tensors
        dim = ctx.dim
        zero = Y.new_zeros(())
        gppr = torch.where(Y > 0, Y ** (2 - al), zero)
        gs = gppr.sum(dim, keepdim=True)
        dX = dY * gppr
        dX = dX - (dX.sum(dim, keepdim=True) / gs) * gppr
        d_al = None
        if ctx.needs_input_grad[1]:
            S = torch.where(Y > 0, Y * torch.log(Y), zero)
            ent = S.sum(dim, keepdim=True)
            Ysk = gppr / gs
            d_al = (dY * (Y - Ysk) / ((al - 1) ** 2) - dY * (S - Ysk * ent) / (al - 1)).sum(dim, keepdim=True)
        return dX, d_al, None, None, None


def entmax_rows_forward(X, alpha, dim=-1, n_iter=50, ensure_sum_one=True):
    """utils/entmax.py:31-36 with a TENSOR alpha: broadcast over every dimension of X but `dim` (its extent along `dim` must
    be 1), every row solved with its own alpha by the reference's bisection (armnet_entmax_rows_f32).  Differentiable in X and
    in alpha (round 6: _EntmaxRowsFn — the expand to X's shape is a torch op, so autograd reduces the gradient to alpha's own
    shape exactly as it does for the reference)."""
    nd = X.dim()
    dim = dim % nd
    shape = list(X.shape)
    shape[dim] = 1
    al = alpha.to(dtype=X.dtype, device=X.device).expand(*shape)       # entmax.py:33-36
    if torch.is_grad_enabled() and (X.requires_grad or al.requires_grad):
        return _EntmaxRowsFn.apply(X, al, dim, n_iter, ensure_sum_one)
    return _entmax_rows_raw(X, al, dim, n_iter, ensure_sum_one)


def entmax_forward(X, alpha=1.5, dim=-1, n_iter=50, ensure_sum_one=True, flags=0):
    """utils/entmax.py:134 — alpha-entmax over `dim` (softmax when alpha == 1); differentiable in X."""
    if torch.is_grad_enabled() and X.requires_grad:
        return _EntmaxFn.apply(X, alpha, dim, n_iter, ensure_sum_one, flags)
    return _entmax_raw(X, alpha, dim, n_iter, ensure_sum_one, flags)
