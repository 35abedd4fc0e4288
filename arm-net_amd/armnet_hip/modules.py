"""nn.Module plumbing shared by models/armnet.py and models/armnet_1h.py.

The two public ARMNetModel classes keep the reference's constructor signatures,
sub-module names and state_dict keys so weights move freely
between the reference and this build; everything between the input dict and the
MLP head's input runs in ONE fused HIP kernel (armnet_fused_fwd_f32).
"""
import torch
import torch.nn as nn

from . import host_ops, native
from .block import ArmBlockParams, IdStatus, _require_cuda, arm_block_forward, embedding_forward, entmax_forward


class _GatherScaleFn(torch.autograd.Function):
    """table[ids] * vals with a dense table gradient (what nn.Embedding + multiply gives the reference)."""

    @staticmethod
    def forward(ctx, table, ids, vals, check_ids):
        ctx.save_for_backward(ids, vals)
        ctx.nfeat = table.shape[0]
        check_ids, status = check_ids if isinstance(check_ids, tuple) else (check_ids, None)
        return embedding_forward(ids, vals, table, check_ids=check_ids, status=status)

    @staticmethod
    def backward(ctx, g):
        ids, vals = ctx.saved_tensors
        E = g.shape[-1]
        d_table = torch.zeros(ctx.nfeat, E, device=g.device, dtype=g.dtype)
        native.scatter_add(ids.contiguous(), vals.contiguous(), g.contiguous(), d_table)
        return d_table, None, None, None


class _ArmBlockFn(torch.autograd.Function):
    """Pre-BatchNorm exponential neurons z [B,O,E] with the HIP forward (identity affine) and the HIP
    backward (armnet_fused_bwd_f32); BatchNorm1d and the MLP head stay with torch autograd."""

    @staticmethod
    def forward(ctx, table, bilinear_w, query, values, ids, vals, cfg):
        variant, K, H, E, D, alpha, n_iter, flags, check_ids = cfg
        qf, one, zero = _fold_train(variant, K, H, E, D, bilinear_w, query)
        check_ids, status = check_ids if isinstance(check_ids, tuple) else (check_ids, None)
        z = arm_block_forward(ids, vals, table, qf, values, one, zero, alpha, n_iter=n_iter,
                              write_clamped_vals=True, check_ids=check_ids, flags=flags, status=status)
        ctx.save_for_backward(table, bilinear_w, query, values, ids, vals, qf, z)
        ctx.cfg = cfg
        return z

    @staticmethod
    def backward(ctx, dz):
        table, bilinear_w, query, values, ids, vals, qf, z = ctx.saved_tensors
        variant, K, H, E, D, alpha, n_iter, flags, _ = ctx.cfg
        B, F = vals.shape
        O = K * H
        d_table = torch.zeros_like(table)
        d_values, d_qf = _param_grad_buffers(O, F, E, dz.device)
        native.fused_bwd(B, F, E, O, alpha, n_iter, flags, ids.contiguous(), vals, table.detach(), qf,
                         values.detach().reshape(O, F).contiguous(), z, dz.contiguous(), d_table, d_values, d_qf)
        return _arm_block_grads((variant, K, H, E, D), (bilinear_w, query, values), d_qf, d_table, d_values) + (None, None, None)


_CONSTS = {}


def _unit_affine(dev, O):
    """(ones[O], zeros[O], scratch[O], scratch[O]) kept per device and width: the identity BatchNorm affine of the
    training forward and the unused fold outputs (small per-step allocations and fills add up at batch 4 096)"""
    key = (str(dev), O)
    if key not in _CONSTS:
        _CONSTS[key] = (torch.ones(O, device=dev), torch.zeros(O, device=dev), torch.empty(O, device=dev),
                        torch.empty(O, device=dev))
    return _CONSTS[key]


def _fold_train(variant, K, H, E, D, bilinear_w, query):
    dev = query.device
    O = K * H
    one, zero, sc, sh = _unit_affine(dev, O)
    qf = torch.empty(O, E, device=dev, dtype=torch.float32)
    native.fold_params(variant, K, H, E, D, bilinear_w.detach().contiguous(), query.detach().contiguous(),
                       one, zero, zero, one, 0.0, qf, sc, sh)
    return qf, one, zero


def _param_grad_buffers(O, F, E, dev):
    """d_values [O,F] and d_q_fold [O,E] as views of ONE zero-filled buffer"""
    buf = torch.zeros(O * (F + E), device=dev, dtype=torch.float32)
    return buf[:O * F].view(O, F), buf[O * F:].view(O, E)


def _arm_block_grads(ctx_cfg, tensors, d_qf, d_table, d_values):
    """chain rule through the parameter fold (q_fold from bilinear_w and query)"""
    variant, K, H, E, D = ctx_cfg
    bilinear_w, query, values = tensors
    d_qf = d_qf * (float(D) ** -0.5)
    if variant == native.ONE_HEAD:                      # q_fold = scale * query @ W,  W = bilinear_w [D,E]
        d_q = d_qf @ bilinear_w.t()
        d_w = query.t() @ d_qf
    else:                                               # q_fold[k,o,e] = scale * sum_y W[k,e,y] query[k,o,y]
        g3 = d_qf.view(K, H, E)
        d_q = torch.einsum("koe,key->koy", g3, bilinear_w)
        d_w = torch.einsum("koe,koy->key", g3, query)
    return d_table, d_w, d_q, d_values.reshape(values.shape)


class _ArmBlockBNFn(torch.autograd.Function):
    """Training step of the block INCLUDING its training-mode BatchNorm1d (armnet_1h.py:85 / armnet.py:88-89):
    forward = fused kernel (pre-BN neurons z) + batch statistics + normalise, all HIP; backward = the BatchNorm
    reductions over (z, dy), then armnet_fused_bwd_bn_f32, which forms dz from dy inside the kernel."""

    @staticmethod
    def forward(ctx, table, bilinear_w, query, values, bn_weight, bn_bias, ids, vals, cfg, bn_state):
        variant, K, H, E, D, alpha, n_iter, flags, check_ids = cfg
        running_mean, running_var, momentum, eps = bn_state
        qf, one, zero = _fold_train(variant, K, H, E, D, bilinear_w, query)
        check_ids, status = check_ids if isinstance(check_ids, tuple) else (check_ids, None)
        z = arm_block_forward(ids, vals, table, qf, values, one, zero, alpha, n_iter=n_iter,
                              write_clamped_vals=True, check_ids=check_ids, flags=flags, status=status)
        y, mean, rstd, _, _ = native.bn_forward_train(z, bn_weight.detach(), bn_bias.detach(), running_mean,
                                                      running_var, momentum, eps, relu=False)
        ctx.save_for_backward(table, bilinear_w, query, values, bn_weight, ids, vals, qf, z, mean, rstd)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        table, bilinear_w, query, values, bn_weight, ids, vals, qf, z, mean, rstd = ctx.saved_tensors
        variant, K, H, E, D, alpha, n_iter, flags, _ = ctx.cfg
        B, F = vals.shape
        O = K * H
        dy = dy.contiguous()
        d_bnw, d_bnb, cA, cB, cC = native.bn_backward_coef(z, dy, bn_weight.detach(), mean, rstd)
        d_table = torch.zeros_like(table)
        d_values, d_qf = _param_grad_buffers(O, F, E, dy.device)
        native.fused_bwd_bn(B, F, E, O, alpha, n_iter, flags, ids.contiguous(), vals, table.detach(), qf,
                            values.detach().reshape(O, F).contiguous(), z, dy, cA, cB, cC, d_table, d_values, d_qf)
        d_table, d_w, d_q, d_v = _arm_block_grads((variant, K, H, E, D), (bilinear_w, query, values), d_qf, d_table,
                                                  d_values)
        return d_table, d_w, d_q, d_v, d_bnw, d_bnb, None, None, None, None


class _BatchNormTrainFn(torch.autograd.Function):
    """training-mode BatchNorm1d (+ optional fused ReLU) on [N,C] / [N,C,L] with the HIP passes of bn_kernels.hip"""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, relu):
        x = x.contiguous()
        y, mean, rstd, scale, shift = native.bn_forward_train(x, weight.detach(), bias.detach(), running_mean,
                                                              running_var, momentum, eps, relu)
        ctx.save_for_backward(x, weight, mean, rstd, scale, shift)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, mean, rstd, scale, shift = ctx.saved_tensors
        dy = dy.contiguous()
        rs, rt = (scale, shift) if ctx.relu else (None, None)
        d_w, d_b, cA, cB, cC = native.bn_backward_coef(x, dy, weight.detach(), mean, rstd, rs, rt)
        dx = native.bn_backward_apply(x, dy, cA, cB, cC, rs, rt) if ctx.needs_input_grad[0] else None
        return dx, d_w, d_b, None, None, None, None, None


class _LinearSplitKFn(torch.autograd.Function):
    """nn.Linear whose weight gradient dW = dy^T x is computed split-K: with B = 65 536 rows and a [256, 512]
    result the plain GEMM has 16 output tiles for 256 CUs (hipBLASLt: 0.6 ms = 28 TFLOP/s); as a batched GEMM
    over S row-chunks plus a sum it fills the chip.  Same fp32 products, a different (pairwise-like) summation
    order."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return torch.addmm(bias, x, weight.t())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        B = x.shape[0]
        dx = dy @ weight if ctx.needs_input_grad[0] else None
        S = 1
        while S < 64 and B % (2 * S) == 0 and B // (2 * S) >= 1024:
            S *= 2
        if S > 1:
            dw = torch.bmm(dy.reshape(S, B // S, -1).transpose(1, 2), x.reshape(S, B // S, -1)).sum(0)
        else:
            dw = dy.t() @ x
        return dx, dw, (dy.sum(0) if ctx.needs_input_grad[2] else None)


def _linear_mfma_ok(x, weight):
    """shapes the bf16x3 matrix-core Linear (armnet_linear_bf16x3_f32) takes for BOTH the forward and the input gradient:
    device float32, widths in whole 16-float k-steps (every head width BASELINE.json builds: 512 / 2048 / 704 -> 256)"""
    N, K = weight.shape
    return (x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2
            and x.shape[0] >= 2048 and K % 16 == 0 and N % 16 == 0 and N >= 32)


def _linear_mfma(x, W, bias):
    """bias + x @ W.T for W [N, K] through armnet_linear_bf16x3_f32, in slices of <= 256 outputs; the weights are packed
    (three bf16 planes in the kernel's operand order) per call: they change with every optimizer step"""
    B, K = x.shape
    N = W.shape[0]
    out = torch.empty(B, N, device=x.device, dtype=torch.float32)
    for n0 in range(0, N, 256):
        n1 = min(N, n0 + 256)
        blob = torch.zeros(native.mlp_packed_bytes(K, n1 - n0, 1), device=x.device, dtype=torch.uint8)
        native.mlp_pack_layer(K, n1 - n0, 1, 0, W[n0:n1].contiguous(), None if bias is None else bias[n0:n1].contiguous(),
                              None, blob)
        native.linear_bf16x3(x, blob, out[:, n0:], K, n1 - n0)
    return out


class _LinearMfmaFn(torch.autograd.Function):
    """nn.Linear of the TRAINING head on the bf16 matrix cores (round 5): forward x W^T + b and the input gradient dY W as
    armnet_linear_bf16x3_f32 launches (three-way bf16 split of both operands, six cross products, fp32 accumulate: the
    error class of an fp32 GEMM at 6/16 of its matrix-pipe time); the weight gradient dY^T X stays the split-K batched
    GEMM of _LinearSplitKFn (its contraction runs over the samples: both operands would need a run-time split AND a
    transpose — not built)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return _linear_mfma(x, weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        B = x.shape[0]
        dx = _linear_mfma(dy, weight.detach().t(), None) if ctx.needs_input_grad[0] else None
        S = 1
        while S < 64 and B % (2 * S) == 0 and B // (2 * S) >= 1024:
            S *= 2
        if S > 1:
            dw = torch.bmm(dy.reshape(S, B // S, -1).transpose(1, 2), x.reshape(S, B // S, -1)).sum(0)
        else:
            dw = dy.t() @ x
        return dx, dw, (dy.sum(0) if ctx.needs_input_grad[2] else None)


class HipBatchNorm1d(nn.BatchNorm1d):
    """nn.BatchNorm1d (same parameters, buffers and state_dict keys) whose TRAINING forward/backward on the GPU
    run as the HBM-bound HIP passes of bn_kernels.hip; eval mode and anything unusual (no affine, no running
    stats, cumulative momentum, non-fp32, CPU) fall through to torch."""

    def _hip_ok(self, x):
        return (self.training and x.is_cuda and x.dtype == torch.float32 and self.affine and self.track_running_stats
                and self.momentum is not None and x.dim() in (2, 3) and x.numel() // x.shape[1] > 1)

    def forward(self, x, relu=False):
        if not self._hip_ok(x):
            y = super().forward(x)
            return torch.relu(y) if relu else y
        self.num_batches_tracked.add_(1)
        return _BatchNormTrainFn.apply(x, self.weight, self.bias, self.running_mean, self.running_var,
                                       float(self.momentum), float(self.eps), bool(relu))


class HipEmbedding(nn.Module):
    """Field embedding lookup scaled by the field value (reference: models/layers.py:8-21).

    forward(x) with x = {'id': Long[B,F], 'value': Float[B,F]} -> Float[B,F,E] computed by
    armnet_gather_scale_f32.  The parameter lives at ``embedding.weight`` like the reference's."""

    def __init__(self, nfeat, nemb):
        super().__init__()
        self.embedding = nn.Embedding(nfeat, nemb)
        nn.init.xavier_uniform_(self.embedding.weight)
        self.check_ids = True          # True: IndexError for an out-of-range id at the NEXT call / poll() (block.IdStatus);
        self._id_status = IdStatus()   # "sync": before the call returns (one host sync); False: unchecked

    def poll(self):
        """synchronise and raise IndexError if any call so far met an id outside [0, nfeat) (block.IdStatus)"""
        self._id_status.poll(self.embedding.weight.device)

    def forward(self, x, check_ids=None):
        check = self.check_ids if check_ids is None else check_ids
        if host_ops.on_host(x["id"], x["value"], self.embedding.weight):     # never moved to the GPU: the reference's ops
            return host_ops.embedding(x["id"], x["value"], self.embedding.weight)
        if torch.is_grad_enabled() and self.embedding.weight.requires_grad:
            return _GatherScaleFn.apply(self.embedding.weight, x["id"], x["value"], (check, self._id_status))
        return embedding_forward(x["id"], x["value"], self.embedding.weight, check_ids=check, status=self._id_status)


def build_mlp(ninput, nlayers, nhid, dropout, noutput=1):
    """Prediction head (reference: models/layers.py:68-88): nlayers x (Linear, BatchNorm1d, ReLU,
    Dropout) then Linear(., noutput); with nlayers == 0 a single Linear(ninput, noutput)."""
    stack = []
    width = ninput
    for _ in range(nlayers):
        stack += [nn.Linear(width, nhid), HipBatchNorm1d(nhid), nn.ReLU(), nn.Dropout(p=dropout)]
        width = nhid
    stack.append(nn.Linear(width, noutput))
    return nn.Sequential(*stack)


class SparseGateBase(nn.Module):
    """Owner of the attention parameters (bilinear_w, query, values) of one ARM block.

    Calling it on x [B,F,E] returns the per-field value weights (armnet_1h.py:25-34 /
    armnet.py:26-36): gates by a dense contraction, the sparse map by armnet_entmax_f32.  The model
    forward does NOT go through here — it uses the fused kernel — this is the stand-alone surface."""

    alpha: float
    n_iter = 50

    def _gates(self, x):
        raise NotImplementedError

    def forward(self, x):
        gates = self._gates(x)
        p = entmax_forward(gates, self.alpha, dim=-1, n_iter=self.n_iter)
        return p * self.values


class ArmNetBase(nn.Module):
    """Common forward of both ARM-Net variants."""

    variant = native.ONE_HEAD

    def _init_common(self, nfield, nfeat, nemb, nhead, nhid, alpha, mlp_nlayer, mlp_nhid, dropout, ensemble,
                     deep_nlayer, deep_nhid, noutput, attn_layer):
        # construction order == the reference's (armnet_1h.py:59-74): same seed -> same initial weights
        self.nfield, self.nfeat, self.nemb = nfield, nfeat, nemb
        self.nhead, self.nhid, self.alpha = nhead, nhid, float(alpha)
        self.embedding = HipEmbedding(nfeat, nemb)
        self.attn_layer = attn_layer()
        self.arm_bn = HipBatchNorm1d(nhead * nhid)
        self.mlp = _MLP(nhead * nhid * nemb, mlp_nlayer, mlp_nhid, dropout, noutput=noutput)
        if ensemble:
            self.deep_embedding = HipEmbedding(nfeat, nemb)
            self.deep_mlp = _MLP(nfield * nemb, deep_nlayer, deep_nhid, dropout, noutput=noutput)
            self.ensemble_layer = nn.Linear(2 * noutput, noutput)
            nn.init.constant_(self.ensemble_layer.weight, 0.5)
            nn.init.constant_(self.ensemble_layer.bias, 0.0)
        self._folded = ArmBlockParams()
        self.check_ids = True          # IndexError on out-of-range ids: True = DEFERRED to this model's next call or
        self._id_status = self.embedding._id_status   # poll() (no host sync per forward — the reference's GPU behaviour, layers.py:20);
                                       # "sync" = before forward returns (one host sync per call); False = unchecked
        self.n_iter = 50               # utils/entmax.py:239 default, never overridden by the reference
        self.kernel_flags = 0          # native.F_* bits for testing (faithful bisection, generic kernel)

    # -- the fused block ------------------------------------------------------------------------
    def _d_k(self):
        raise NotImplementedError

    def arm_block(self, ids, vals, out=None):
        """ids [B,F], vals [B,F] (clamped in place) -> post-BN exponential neurons [B, O, E]
        (written into `out` when given: inference with a replicated table only).

        Inference (eval mode under no_grad, or frozen parameters): ONE fused kernel, BN folded.
        Otherwise (training, or eval with autograd on): the same kernel yields the pre-BN neurons inside
        an autograd.Function whose backward is armnet_fused_bwd_f32; arm_bn then runs as a torch module
        (batch statistics + running-stat update in train mode, armnet_1h.py:85).

        A model that was never moved to the GPU, called with host tensors, runs the reference's ATen op chain
        (host_ops.arm_block + the arm_bn module; differentiable) — the dispatch the reference itself makes
        (model_utils.py:86).  Model and batch on different devices raise."""
        at = self.attn_layer
        bw = at.bilinear_w.weight if self.variant == native.ONE_HEAD else at.bilinear_w
        if host_ops.on_host(ids, vals, self.embedding.embedding.weight) and getattr(self, "_shard", None) is None:
            host_ops.note_host_branch(self)
            z = host_ops.arm_block(self.variant == native.ONE_HEAD, ids, vals, self.embedding.embedding.weight, bw,
                                   at.query, at.values, self.alpha, n_iter=self.n_iter)
            y = self.arm_bn(z)
            if out is not None:
                out.copy_(y)
                return out
            return y
        _require_cuda(vals, "x['value']")
        _require_cuda(ids, "x['id']")
        _require_cuda(self.embedding.embedding.weight, "the model (call model.cuda())")
        needs_grad = torch.is_grad_enabled() and any(
            p.requires_grad for p in (self.embedding.embedding.weight, bw, at.query, at.values))
        if self.training or needs_grad:
            if getattr(self, "_shard", None) is not None:
                raise NotImplementedError("training with a row-sharded table is not supported")
            cfg = (self.variant, self.nhead, self.nhid, self.nemb, self._d_k(), self.alpha, self.n_iter,
                   self.kernel_flags, (self.check_ids, self._id_status))
            bn = self.arm_bn
            if (bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
                    and vals.shape[0] * self.nemb > 1):
                bn.num_batches_tracked.add_(1)
                return _ArmBlockBNFn.apply(self.embedding.embedding.weight, bw, at.query, at.values, bn.weight, bn.bias,
                                           ids, vals, cfg, (bn.running_mean, bn.running_var, float(bn.momentum),
                                                            float(bn.eps)))
            z = _ArmBlockFn.apply(self.embedding.embedding.weight, bw, at.query, at.values, ids, vals, cfg)
            return bn(z)
        qf, sc, sh = self._folded.get(self.variant, self.nhead, self.nhid, self.nemb, self._d_k(), bw, at.query,
                                      self.arm_bn)
        if getattr(self, "_shard", None) is not None:
            from .sharded import sharded_arm_block
            self._refresh_shard()
            return sharded_arm_block(self._shard, ids, vals, qf, at.values, sc, sh, self.alpha, n_iter=self.n_iter,
                                     write_clamped_vals=True, flags=self.kernel_flags, check_ids=self.check_ids)
        return arm_block_forward(ids, vals, self.embedding.embedding.weight, qf, at.values, sc, sh, self.alpha,
                                 n_iter=self.n_iter, write_clamped_vals=True, check_ids=self.check_ids,
                                 flags=self.kernel_flags, out=out, status=self._id_status)

    def poll(self):
        """Synchronise the device and raise IndexError if any forward of this model so far met an id outside [0, nfeat)
        — the deferred report of `check_ids = True` (block.IdStatus; the reference's GPU path reports the same way: a
        device-side assert at a later synchronisation, layers.py:20).  (A row-sharded table checks per call instead:
        sharded_arm_block's all-reduced flag.)"""
        self._id_status.poll(self.embedding.embedding.weight.device)

    def shard_embedding(self, group=None, release_full=False, hot_rows=0, data_groups=None):
        """Row-shard the ARM embedding table over the process group (multi-GPU): this rank
        keeps a COPY of rows i = rank (mod world); every later inference arm_block() call fetches rows by
        all-to-all.  The full table must be resident when this is called.

        The full parameter `embedding.embedding.weight` stays resident by default (state_dict, training and
        un-sharding keep working) and the shard is re-cut whenever the parameter changes (load_state_dict, an
        optimizer step: tracked by its storage pointer and version counter; writes through `.data` are not
        seen — call shard_embedding() again after those).  With release_full=True the parameter's storage is
        replaced by an empty [0, nemb] tensor afterwards — that is what frees the memory; the shard is then the
        only copy (state_dict no longer holds the table).

        data_groups (round 6): further process groups over the same ranks, one per step a serving loop keeps in flight on its
        own stream — the exchanges of consecutive steps then run on different communicators and overlap (RowShardedTable).

        hot_rows = N (round 5): rows [0, N) — the head of a frequency-ordered id space, where skewed click logs put most
        lookups — are also kept REPLICATED on every rank (N * nemb * 4 bytes) and served without crossing a link
        (sharded.RowShardedTable).

        Training with a row-sharded table is not supported (SURVEY.md §8e scopes the sharded lookup to inference: the
        training-mode BatchNorm and the table gradient would need their own collectives): arm_block() raises
        NotImplementedError in train mode / with autograd on while a shard is attached."""
        import torch.distributed as dist
        from .sharded import RowShardedTable, shard_rows
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        p = self.embedding.embedding.weight
        w = p.detach()
        self._shard = RowShardedTable(shard_rows(w, rank, world), w.shape[0], group, hot_rows=hot_rows)
        self._shard.data_groups = list(data_groups) if data_groups else None    # one communicator per stream in flight (sharded.py)
        self._shard_src = None if release_full else (p.data_ptr(), p._version)
        if release_full:
            p.data = torch.empty(0, w.shape[1], device=w.device, dtype=w.dtype)
        return self

    def _refresh_shard(self):
        """re-cut the local shard when the full table changed since shard_embedding() (no-op after release_full)"""
        src = getattr(self, "_shard_src", None)
        if src is None:
            return
        p = self.embedding.embedding.weight
        if (p.data_ptr(), p._version) != src:
            from .sharded import shard_rows
            sh = self._shard
            sh.table_local = shard_rows(p.detach(), sh.rank, sh.world)   # the setter also drops the whole-shard copy
            self._shard_src = (p.data_ptr(), p._version)

    def invalidate_folded(self):
        """Drop every cached parameter fold (q_fold, the BatchNorm affines, the MLPs' folded / split weights).
        The caches key on (storage pointer, version counter) of their sources, which catches optimizer steps,
        load_state_dict and in-place ops — but NOT writes through `.data` (p.data.copy_(), some EMA /
        weight-averaging utilities): call this after such a write.  train()/eval() call it too."""
        self._folded.key = None
        for m in self.modules():
            if isinstance(m, _MLP):
                m.invalidate()
        if getattr(self, "_shard", None) is not None and getattr(self, "_shard_src", None) is not None:
            self._shard_src = (0, -1)

    def warm_caches(self, heads=True):
        """Build every lazily produced parameter cache NOW, on the current stream: q_fold + the BatchNorm affine, the
        re-cut of the local shard, and (heads) the MLPs' packed / folded weights.  Whoever then forks work onto other
        streams (serving.InFlight) reads caches that already exist instead of racing with the stream that first needs
        them (round-3 advisor finding: one hook instead of attribute probing)."""
        with torch.no_grad():
            at = self.attn_layer
            bw = at.bilinear_w.weight if self.variant == native.ONE_HEAD else at.bilinear_w
            self._folded.get(self.variant, self.nhead, self.nhid, self.nemb, self._d_k(), bw, at.query, self.arm_bn)
            if getattr(self, "_shard", None) is not None:
                self._refresh_shard()
            if heads and not self.training:
                _warm_heads(self)

    def train(self, mode=True):
        self.invalidate_folded()
        return super().train(mode)

    def make_graphed(self, ids, vals):
        """Capture this model's inference forward for the shape of (ids, vals) in a hipGraph."""
        return GraphedForward(self, ids, vals)

    def forward(self, x, vals=None):
        """x = {'id': Long[B,F], 'value': Float[B,F], ...} -> logits Float[B] (armnet_1h.py:76-98).
        Also accepts forward(ids, vals).  x['value'] is clamped to [1e-3, 1] IN PLACE."""
        if vals is not None:
            x = {"id": x, "value": vals}
        ids, v = x["id"], x["value"]
        if v.dtype != torch.float32:
            raise native.ArmnetNativeError(f"x['value'] must be float32, got {v.dtype}")
        v_run = v if v.is_contiguous() else v.contiguous()
        x_arm = self.arm_block(ids, v_run)                       # [B, O, E]
        if v_run is not v:
            v.copy_(v_run)                                       # keep the visible clamp side effect
        fused_tail = _fused_ensemble_tail(self, x_arm.view(x_arm.shape[0], -1), ids, v)
        if fused_tail is not None:
            return fused_tail.squeeze()
        y = self.mlp(x_arm.view(x_arm.shape[0], -1))            # [B, noutput]
        if hasattr(self, "ensemble_layer"):
            # ids were validated by the fused call; sees the clamped values (armnet.py:94)
            x_deep = self.deep_embedding({"id": ids, "value": v}, check_ids=False)
            y_deep = self.deep_mlp(x_deep.view(x_deep.shape[0], -1))
            yy = torch.cat([y, y_deep], dim=1)
            if self.training and yy.shape[0] >= 2048:           # tiny-N, huge-K weight gradient: split-K
                y = _LinearSplitKFn.apply(yy, self.ensemble_layer.weight, self.ensemble_layer.bias)
            else:
                y = self.ensemble_layer(yy)
        return y.squeeze()


class GraphedForward:
    """hipGraph-captured inference of one ARM-Net module at a fixed batch shape (serving / small batches,
    where per-call host overhead — allocation, ctypes, three launches + the MLP's GEMMs — exceeds the
    kernels).  Inputs are copied into static device buffers, the captured graph is replayed, and the static
    logits buffer is returned (clone it if it must outlive the next call).  The in-place clamp of
    x['value'] is applied to the static copy and mirrored back to the caller's tensor."""

    def __init__(self, model, ids, vals, warmup=3):
        if model.training:
            raise RuntimeError("GraphedForward captures the inference path: call model.eval() first")
        self.model = model
        self.ids = ids.clone()
        self.vals = vals.clone()
        check = model.check_ids
        model.check_ids = bool(check)               # no host sync inside a capture: the range test stays live in the captured
        try:                                        # kernels (block.IdStatus' pinned word), its report is read in __call__
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):
                    model({"id": self.ids, "value": self.vals})
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self.out = model({"id": self.ids, "value": self.vals})
        finally:
            model.check_ids = check

    def __call__(self, x, vals=None):
        ids, v = (x["id"], x["value"]) if vals is None else (x, vals)
        if ids.shape != self.ids.shape or v.shape != self.vals.shape:
            raise ValueError(f"graph was captured for batch shape {tuple(self.ids.shape)}, got {tuple(ids.shape)}")
        if self.model.check_ids:
            self.model._id_status.raise_if_set()    # an earlier replay's out-of-range id (host-memory read, no sync)
        self.ids.copy_(ids)
        self.vals.copy_(v)
        self.graph.replay()
        v.copy_(self.vals)                          # the reference's visible clamp side effect
        if self.model.check_ids == "sync":
            self.model.poll()
        return self.out


class GraphedTrainStep:
    """hipGraph-captured training step (forward, loss, backward, optimizer step) of one ARM-Net module at a fixed
    batch shape.  At the reference's default batch size (train.py: 4096) a step is ~60 short kernels and the host
    (Python, allocator, launches) takes longer than the GPU; replaying one graph removes that.

        step = GraphedTrainStep(model, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True), loss_fn, ids, vals, y)
        loss = step(ids, vals, y)          # static loss tensor, overwritten by the next call

    The optimizer must be capture-safe (torch: ``capturable=True``).  An out-of-range id (it reads row 0) is reported like
    the eager step's: IndexError at the next call or model.poll() (block.IdStatus).  x['value'] is clamped in the static copy and
    mirrored back to the caller's tensor like the eager step."""

    def __init__(self, model, optimizer, loss_fn, ids, vals, y, warmup=3):
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures the training path: call model.train() first")
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.ids, self.vals, self.y = ids.clone(), vals.clone(), y.clone()
        check = model.check_ids
        model.check_ids = bool(check)               # no host sync inside a capture (see GraphedForward)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    self._eager()
            torch.cuda.current_stream().wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            self.opt.zero_grad(set_to_none=True)
            with torch.cuda.graph(self.graph):
                self.loss = self._eager(zero=False)
        finally:
            model.check_ids = check                 # later eager calls validate ids again

    def _eager(self, zero=True):
        if zero:
            self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(self.model({"id": self.ids, "value": self.vals}), self.y)
        loss.backward()
        self.opt.step()
        return loss.detach()

    def __call__(self, ids, vals, y):
        if ids.shape != self.ids.shape or vals.shape != self.vals.shape:
            raise ValueError(f"graph was captured for batch shape {tuple(self.ids.shape)}, got {tuple(ids.shape)}")
        if self.model.check_ids:
            self.model._id_status.raise_if_set()    # an earlier step's out-of-range id (host-memory read, no sync)
        self.ids.copy_(ids)
        self.vals.copy_(vals)
        self.y.copy_(y)
        self.graph.replay()
        vals.copy_(self.vals)
        if self.model.check_ids == "sync":
            self.model.poll()
        return self.loss


class _MLP(nn.Module):
    """reference: models/layers.py:68-88 (state_dict keys mlp.<i>.*).

    Training mode on the GPU: hipBLASLt Linear, then BatchNorm1d + ReLU fused into the HIP passes of
    bn_kernels.hip (HipBatchNorm1d).  Eval-mode inference on the GPU: the whole head is ONE hand-written kernel
    (armnet_mlp_head_f32, csrc/mlp_head.hip: BatchNorm folded into the weights, W' = W * s, b' = b * s + t with
    s = gamma / sqrt(var + eps), t = beta - mean * s; operands split into three bf16 slices, six cross products on the
    bf16 matrix cores, fp32 accumulate; packed weights refreshed when any source tensor's version counter moves) for
    heads with 1..n hidden layers and one output — every head the reference builds; hidden widths above 256 (run.sh's
    500) run as slices of <= 256 units through the same kernel (_hip_plan).  Anything else (no hidden layer, several
    outputs, `hip_head = False`) takes one hipBLASLt GEMM per folded
    (Linear, BatchNorm1d, ReLU, Dropout) group with a bias+ReLU epilogue."""

    def __init__(self, ninput, nlayers, nhid, dropout, noutput=1):
        super().__init__()
        self.mlp = build_mlp(ninput, nlayers, nhid, dropout, noutput)
        self._fold_key = None
        self._folded = None
        self.fold_eval = True
        self.hip_head = True           # eval-mode inference through armnet_mlp_head_f32 where it has a kernel
        self.bf16x3 = False            # True: force the bf16 x 3 operand split (six products; the rounds 2-5 kernel) instead
                                       # of fp16 x 2 (three products, in-kernel fallback to bf16 x 3 out of the fp16 range)
        self.mfma_train = False        # training-mode Linear forward / dX through armnet_linear_bf16x3_f32 (round 5).  OFF:
                                       # measured 2.13 ms against 2.09 ms per step at B = 65 536 and 0.94 against 0.75 ms
                                       # graphed at B = 4 096 (profiles/r05_train_step_times.txt) — one layer per launch
                                       # (training BatchNorm needs the batch's pre-activations, so nothing chains in
                                       # registers) re-streams and re-packs the weights for 85 us where hipBLASLt's fp32
                                       # GEMM takes 60-110, and dX of a 512-wide input is two slices
        self._dims = (ninput, nlayers, nhid, noutput)
        self._pack_key = {}
        self._packed = {}              # ens flag -> [(K0, n_hidden, has_final, blob)] one entry per launch

    def eval_path(self):
        """which code runs the eval-mode head (reported by bench.py)"""
        if self.hip_head and self._hip_plan() is not None:
            last = "armnet_linear_small_f32" if self._dims[3] <= 16 else "armnet_linear_bf16x3_f32"
            if self._dims[1] == 0:
                return f"{last}: the head is one Linear (nlayers = 0)" + (", a plain fp32 HIP kernel" if self._dims[3] <= 16 else "")
            split = ("bf16x3-split operands on v_mfma_f32_32x32x16_bf16 (6 cross products, fp32 accumulate)" if self.bf16x3 else
                     "fp16x2-split operands on v_mfma_f32_32x32x16_f16 (3 cross products, fp32 accumulate; blocks whose "
                     "inputs leave the fp16 range redo in bf16x3 inside the launch)")
            return (f"armnet_mlp_head_f32: ONE HIP kernel, {split}, hidden layers chained in registers"
                    + ("" if self._dims[3] == 1 else f"; final Linear with several outputs by {last}"))
        return "torch/hipBLASLt fp32 GEMMs, BatchNorm folded into the weights, bias+ReLU epilogue"

    def invalidate(self):
        """forget the folded / packed eval-mode weights (see ArmNetBase.invalidate_folded)"""
        self._fold_key = None
        self._folded = None
        self._pack_key = {}
        self._packed = {}

    HIP_MAX_SLICE = 256            # hidden units per launch of the head kernel

    def _hip_plan(self):
        """launch plan of the HIP head: [(first hidden layer, hidden layers fused, has_final, n0, n1)], or None when
        there is no kernel for this head (no hidden layer, more than one output).  Hidden width <= 256: up to two hidden
        layers (+ the final Linear) per launch.  Wider (run.sh:18-19,44-45 build mlp_hid / dnn_hid 500): one hidden
        layer per launch, cut into slices [n0, n1) of <= 256 units that write their columns of a [B, nhid] buffer; in
        the last hidden layer the first slice writes its share of the final Linear (has_final 1) and the others add
        theirs (has_final 2)."""
        ninput, nlayers, nhid, noutput = self._dims
        if noutput < 1:
            return None
        if nlayers == 0:
            return []                                  # one Linear(ninput, noutput): armnet_linear_small_f32 alone (round 4)
        if nhid < 1:
            return None
        one = noutput == 1                             # several outputs: the hidden layers here, the final Linear(nhid, noutput)
        S = self.HIP_MAX_SLICE                         # by armnet_linear_small_f32 on the last hidden activations (round 4)
        plan, i = [], 0
        if nhid <= S:
            if not native.mlp_head_supported(ninput, nhid, 1):
                return None
            while i < nlayers:
                n = 2 if nlayers - i >= 2 else 1
                plan.append((i, n, 1 if (i + n == nlayers and one) else 0, 0, nhid))
                i += n
            return plan
        nsl = (nhid + S - 1) // S
        w = ((nhid + nsl - 1) // nsl + 31) // 32 * 32            # equal slices, whole 32-unit tiles
        for i in range(nlayers):
            for k, n0 in enumerate(range(0, nhid, w)):
                last = i + 1 == nlayers and one
                plan.append((i, 1, (1 if k == 0 else 2) if last else 0, n0, min(nhid, n0 + w)))
        return plan

    def _groups(self):
        """[(Linear, BatchNorm1d)] of the hidden layers and the final Linear"""
        mods = list(self.mlp)
        hidden = [(mods[i], mods[i + 1]) for i in range(0, len(mods) - 1, 4)]
        return hidden, mods[-1]

    def _pack(self, ens=None):
        """packed eval-mode weights of the launch plan.  ens = (ensemble Linear(2, 1), column): the ensemble tail
        (armnet.py:97-99: cat([y, y_deep]) -> Linear(2, 1)) folded into this head's final Linear — its weights scaled by
        the ensemble weight of this head's column, the ensemble bias added to column 0's bias — so that the two heads
        of an ensemble model write and accumulate ONE logits buffer and no torch op is left (round-3 verdict, item 7)"""
        mods = list(self.mlp)
        src = [p for m in mods for p in list(m.parameters()) + list(m.buffers())]
        if ens is not None:
            src += [ens[0].weight, ens[0].bias]
        ek = None if ens is None else int(ens[1])
        key = tuple((t.data_ptr(), t._version) for t in src)
        if key != self._pack_key.get(ek):
            ninput, nlayers, nhid, _ = self._dims
            hidden, last = self._groups()
            dev = last.weight.device
            packed = []
            with torch.no_grad():
                for first, n, has_final, n0, n1 in self._hip_plan():
                    K0 = ninput if first == 0 else nhid
                    wid = n1 - n0                                   # units of this launch (a slice of a wider layer)
                    blob = torch.zeros(native.mlp_packed_bytes(K0, wid, n), device=dev, dtype=torch.uint8)
                    for slot in range(n):
                        lin, bn = hidden[first + slot]
                        native.mlp_pack_layer(K0, wid, n, slot, lin.weight.detach()[n0:n1].contiguous(),
                                              lin.bias.detach()[n0:n1].contiguous(),
                                              (bn.weight.detach()[n0:n1].contiguous(), bn.bias.detach()[n0:n1].contiguous(),
                                               bn.running_mean[n0:n1].contiguous(), bn.running_var[n0:n1].contiguous(),
                                               float(bn.eps)), blob)
                    if has_final:                                   # the bias rides with the first slice only
                        wl, bl = last.weight.detach()[:, n0:n1], (last.bias.detach() if has_final == 1 else None)
                        if ens is not None:
                            we = ens[0].weight.detach()[0, ek]
                            wl = wl * we
                            if bl is not None:
                                bl = bl * we + (ens[0].bias.detach() if ek == 0 else 0.0)
                        native.mlp_pack_layer(K0, wid, n, 2, wl.contiguous(), bl, None, blob)
                    packed.append((first, K0, n, has_final, n0, n1, blob))
            self._packed[ek], self._pack_key[ek] = packed, key
        return self._packed[ek]

    def _hip_forward(self, x, ens=None, logits=None):
        """ens / logits: see _pack — with `logits` given this head ADDS its (ensemble-weighted) output to them"""
        ninput, nlayers, nhid, noutput = self._dims
        B = x.shape[0]
        hidden, last = self._groups()
        if nlayers == 0:                               # layers.py:79-80: the MLP is one Linear
            return self._final_linear(x, last)
        NP = (nhid + 15) // 16 * 16
        cur, cur_layer = x, 0                          # activations feeding hidden layer `cur_layer`
        flags = native.MLP_F_BF16X3 if self.bf16x3 else 0
        nxt = None
        add = logits is not None
        for first, K0, n, has_final, n0, n1, blob in self._pack(ens):
            if has_final and add:
                has_final = 2                          # accumulate into the other head's logits (its packed bias is added too)
            if first != cur_layer:                     # the previous layer's slices are complete
                cur, cur_layer, nxt = nxt, first, None
            KP = (K0 + 15) // 16 * 16
            if cur.shape[1] < KP and (cur.stride(0) < KP or B == 1):
                # the kernel reads whole 16-float k-steps: pad odd widths with zeros (heads whose input width is a
                # multiple of 16 — every BASELINE.json configuration — take the activations as they are)
                cur = torch.nn.functional.pad(cur, (0, KP - cur.shape[1]))
            if has_final:
                if logits is None:
                    logits = torch.empty(B, device=x.device, dtype=torch.float32)
                native.mlp_head(B, K0, n1 - n0, n, has_final, cur, blob, logits, flags)
            else:
                if nxt is None:
                    nxt = torch.zeros(B, NP, device=x.device, dtype=torch.float32)   # pad columns stay zero
                native.mlp_head(B, K0, n1 - n0, n, 0, cur, blob, nxt[:, n0:], flags)
        if noutput != 1:                               # layers.py:86-87 with several outputs, on the last hidden activations
            return self._final_linear(nxt[:, :nhid], last)
        return logits.view(B, 1)

    def _final_linear(self, x, last):
        """the head's last Linear(K, noutput) alone (layers.py:79-80 with nlayers == 0, layers.py:86-87 with several outputs):
        up to 16 outputs one wave per row in plain fp32 (armnet_linear_small_f32), more on the matrix cores
        (armnet_linear_bf16x3_f32, slices of 256 outputs; round 5 — the reference builds no such head for ARM-Net, but
        `noutput` is a constructor argument, armnet_1h.py:44)"""
        B, K = x.shape
        N = self._dims[3]
        if N <= 16:
            y = torch.empty(B, N, device=x.device, dtype=torch.float32)
            native.linear_small(x, last.weight.detach().contiguous(), last.bias.detach(), y)
            return y
        KP = (K + 15) // 16 * 16                       # the kernel reads whole 16-float k-steps of every row
        if x.stride(1) != 1 or (B > 1 and x.stride(0) < KP) or (B == 1 and KP != K):
            x = torch.nn.functional.pad(x, (0, KP - K))[:, :K]
        key = tuple((t.data_ptr(), t._version) for t in (last.weight, last.bias))
        if self._pack_key.get("final") != key:         # packed bf16 planes of the final Linear, per slice of 256 outputs
            blobs = []
            with torch.no_grad():
                for n0 in range(0, N, 256):
                    n1 = min(N, n0 + 256)
                    blob = torch.zeros(native.mlp_packed_bytes(K, n1 - n0, 1), device=x.device, dtype=torch.uint8)
                    native.mlp_pack_layer(K, n1 - n0, 1, 0, last.weight.detach()[n0:n1].contiguous(),
                                          last.bias.detach()[n0:n1].contiguous(), None, blob)
                    blobs.append((n0, n1, blob))
            self._packed["final"], self._pack_key["final"] = blobs, key
        y = torch.empty(B, N, device=x.device, dtype=torch.float32)
        for n0, n1, blob in self._packed["final"]:
            native.linear_bf16x3(x, blob, y[:, n0:], K, n1 - n0)
        return y

    def _fold(self):
        mods = list(self.mlp)
        src = [p for m in mods for p in list(m.parameters()) + list(m.buffers())]
        key = tuple((t.data_ptr(), t._version) for t in src)
        if key != self._fold_key:
            layers = []
            i = 0
            with torch.no_grad():
                while i < len(mods):
                    lin = mods[i]
                    if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d):   # HipBatchNorm1d is one
                        bn = mods[i + 1]
                        s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
                        t = bn.bias - bn.running_mean * s
                        layers.append(((lin.weight * s[:, None]).t().contiguous(), lin.bias * s + t, True))
                        i += 4                          # Linear, BatchNorm1d, ReLU, Dropout
                    else:
                        layers.append((lin.weight.t().contiguous(), lin.bias.clone(), False))
                        i += 1
            self._folded, self._fold_key = layers, key
        return self._folded

    def forward(self, x):
        if self.training and x.is_cuda:
            # training: Linear by hipBLASLt, then BatchNorm1d + ReLU as ONE HIP pass each way (bn_kernels.hip)
            mods = list(self.mlp)
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.Linear) and m.bias is not None and x.dim() == 2 and x.shape[0] >= 2048:
                    xc = x.contiguous()
                    if self.mfma_train and _linear_mfma_ok(xc, m.weight):
                        h = _LinearMfmaFn.apply(xc, m.weight, m.bias)      # bf16x3 matrix cores: forward and dX
                    else:
                        h = _LinearSplitKFn.apply(xc, m.weight, m.bias)
                else:
                    h = m(x)
                if (isinstance(m, nn.Linear) and i + 2 < len(mods) and isinstance(mods[i + 1], HipBatchNorm1d)
                        and isinstance(mods[i + 2], nn.ReLU)):
                    x = mods[i + 1](h, relu=True)
                    i += 3
                else:
                    x = h
                    i += 1
            return x
        if self.training or torch.is_grad_enabled() or not x.is_cuda or not self.fold_eval:
            return self.mlp(x)                   # autograd in eval mode / CPU: the plain nn.Sequential
        if self.hip_head and x.dim() == 2 and x.dtype == torch.float32 and self._hip_plan() is not None:
            return self._hip_forward(x if x.stride(1) == 1 else x.contiguous())
        for wt, b, relu in self._fold():
            x = torch._addmm_activation(b, x, wt) if relu else torch.addmm(b, x, wt)
        return x


def _warm_heads(model):
    """pack / fold the eval-mode weights of every prediction head of `model` on the current stream"""
    for sub in model.modules():
        if isinstance(sub, _MLP):
            if sub.hip_head and sub._hip_plan() is not None:
                sub._pack()
            elif sub.fold_eval:
                sub._fold()
    ens = getattr(model, "ensemble_layer", None)
    if ens is not None and tuple(ens.weight.shape) == (1, 2):      # the packs of the fused ensemble tail
        for col, head in enumerate((model.mlp, model.deep_mlp)):
            if head.fold_eval and head.hip_head and head._hip_plan() is not None:
                head._pack((ens, col))


def _fused_ensemble_tail(model, x, ids, v):
    """Eval-mode inference of an ENSEMBLE model whose two heads both run on armnet_mlp_head_f32: logits [B, 1] with the
    ensemble tail (armnet.py:93-99 / armnet_1h.py:90-96: second lookup, deep_mlp, cat, Linear(2, 1)) as the second HIP
    lookup and two head launches into one logits buffer — the ensemble Linear is folded into the heads' final Linears
    (_MLP._pack), no torch op runs.  None when this is not that case (no ensemble, training / autograd, a head without a
    kernel): the caller then takes the composed path."""
    if not hasattr(model, "ensemble_layer") or model.training or torch.is_grad_enabled() or not x.is_cuda:
        return None
    m1, m2, ens = model.mlp, model.deep_mlp, model.ensemble_layer
    if tuple(ens.weight.shape) != (1, 2) or x.dtype != torch.float32:
        return None
    for m in (m1, m2):
        if not (m.fold_eval and m.hip_head and m._hip_plan() and m._dims[3] == 1):     # (a non-empty launch plan: hidden layers)
            return None
    x_deep = model.deep_embedding({"id": ids, "value": v}, check_ids=False)      # sees the clamped values (armnet.py:94)
    logits = m1._hip_forward(x if x.stride(1) == 1 else x.contiguous(), ens=(ens, 0))
    m2._hip_forward(x_deep.view(x_deep.shape[0], -1), ens=(ens, 1), logits=logits.view(-1))
    return logits
