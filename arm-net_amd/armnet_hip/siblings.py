"""Sibling models of ARM-Net on the same HIP kernels: GC-ARM and AFN, inference.

Both share ARM-Net's skeleton — clamp, embedding lookup * value, a per-sample [neurons, fields] x [fields, nemb]
contraction, BatchNorm, MLP head, optional DNN ensemble — and differ in how the [neurons, fields] weights arise and in
the element-wise maps around the contraction:

    GC-ARM (models/gc_arm.py)  weights = entmax(gates + global context) * values; contraction over emb_bn(exp(x)); no
                               outer exp
    AFN (models/afn.py)        weights = afn.weight (fixed); contraction over emb_bn(log(x)); bias, outer exp

Eval-mode inference runs on armnet_gc_fused_fwd_f32 / armnet_afn_fused_fwd_f32 + the HIP prediction head.  Training
(train.py:108-114 with --model gc_arm / afn):
  * round 4, nemb <= 64 and nfield <= 48 (round 6: also nemb 65..128 with nfield <= 32) — the block as ONE autograd.Function around the fused kernels (_GcBlockFn /
    _AfnBlockFn): armnet_gather_map_stats_f32 (lookup * value, exp / log, emb_bn's batch sums), the fused forward with this
    batch's emb_bn affine, HIP BatchNorm passes for the block's BatchNorm, armnet_gc_fused_bwd_f32 / armnet_afn_fused_bwd_f32
    on the matrix cores, emb_bn's backward sums and armnet_bn_bwd_scatter_f32.  No [B, K*H, F] tensor is written.
  * otherwise (round 3) — the reference's op chain on the device with autograd: HIP kernels for the lookup and its
    scatter-add gradient (armnet_gather_scale_f32 / armnet_scatter_add_f32), the in-place clamp / clip, the sparse map
    (armnet_entmax_f32, backward armnet_entmax_bwd_f32 = utils/entmax.py:70-80 on the saved output), every training-mode
    BatchNorm1d (bn_kernels.hip); the small contractions go to hipBLASLt through torch (split-K weight gradients).
The head's Linear + BatchNorm + ReLU passes are shared with ARM-Net (modules.py)."""
import torch
import torch.nn as nn

from . import host_ops, native
from .block import IdStatus, _require_cuda
from .modules import HipBatchNorm1d, HipEmbedding, _LinearSplitKFn, _MLP


class _VersionKey:
    """cache key over a set of tensors: (storage pointer, version counter) — see ArmNetBase.invalidate_folded"""

    def __init__(self):
        self.key = None

    def changed(self, tensors, extra=()):
        key = tuple((t.data_ptr(), t._version) for t in tensors) + tuple(extra)
        if key != self.key:
            self.key = key
            return True
        return False


class SiblingBase(nn.Module):
    """shared plumbing: ensemble branch, logits, id checking, folded BatchNorm affines"""

    def _init_tail(self, nfield, nfeat, nemb, ninput, mlp_layers, mlp_hid, dropout, ensemble, deep_layers, deep_hid):
        self.mlp = _MLP(ninput, mlp_layers, mlp_hid, dropout)
        if ensemble:
            self.deep_embedding = HipEmbedding(nfeat, nemb)
            self.deep_mlp = _MLP(nfield * nemb, deep_layers, deep_hid, dropout)
            self.ensemble_layer = nn.Linear(2, 1)
            nn.init.constant_(self.ensemble_layer.weight, 0.5)
            nn.init.constant_(self.ensemble_layer.bias, 0.)
        self.check_ids = True          # True: deferred IndexError (next call / poll()); "sync": per call; False: unchecked
        self._id_status = self.embedding._id_status = IdStatus()    # one report for the model and its lookup module
        self.kernel_flags = 0
        self._fold_key = _VersionKey()
        self._folds = None

    def invalidate_folded(self):
        self._fold_key.key = None
        for m in self.modules():
            if isinstance(m, _MLP):
                m.invalidate()

    def warm_caches(self, heads=True):
        """build the lazily produced folds (q fold, emb_bn / arm_bn / afn_bn affines, the heads' packed weights) on the
        current stream — see ArmNetBase.warm_caches"""
        from .modules import _warm_heads
        with torch.no_grad():
            if not self.training:
                self._fold()
                if heads:
                    _warm_heads(self)

    def train(self, mode=True):
        self.invalidate_folded()
        return super().train(mode)

    def _bn_affine(self, bn):
        dev = bn.weight.device
        sc, sh = torch.empty_like(bn.weight), torch.empty_like(bn.weight)
        native.fold_bn(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, float(bn.eps), sc, sh)
        return sc, sh

    def _inputs(self, x, vals):
        if vals is not None:
            x = {"id": x, "value": vals}
        ids, v = x["id"], x["value"]
        if not self._on_host(ids, v):
            _require_cuda(v, "x['value']")
            _require_cuda(ids, "x['id']")
            _require_cuda(self.embedding.embedding.weight, "the model (call model.cuda())")
        if v.dtype != torch.float32:
            raise native.ArmnetNativeError(f"x['value'] must be float32, got {v.dtype}")
        v_run = v if v.is_contiguous() else v.contiguous()
        ids = ids if ids.is_contiguous() else ids.contiguous()
        return ids, v, v_run

    def _on_host(self, ids, v):
        """a model that was never moved to the GPU, called with host tensors: the reference's ATen op chain from this
        module's own sub-modules (host_ops.py; the reference dispatches the same way, model_utils.py:86)"""
        host = host_ops.on_host(ids, v, self.embedding.embedding.weight)
        if host:
            host_ops.note_host_branch(self)
        return host

    def _needs_autograd(self):
        """training mode, or eval mode with autograd on and a trainable parameter: the composed differentiable path"""
        return self.training or (torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))

    def _lookup_train(self, ids, v_run):
        """value clamp in place (x['value'].clamp_(0.001, 1.)) + differentiable lookup * value (layers.py:15-21)"""
        native.clamp_vals(v_run)
        return self.embedding({"id": ids, "value": v_run}, check_ids=self.check_ids)

    def _finish(self, block, ids, v, v_run):
        if v_run is not v:
            v.copy_(v_run)                                       # keep the visible clamp side effect
        if self.ensemble:
            from .modules import _fused_ensemble_tail
            fused = _fused_ensemble_tail(self, block.view(block.shape[0], -1), ids, v)
            if fused is not None:
                return fused.squeeze(1)
        y = self.mlp(block.view(block.shape[0], -1))
        if self.ensemble:
            x_deep = self.deep_embedding({"id": ids, "value": v}, check_ids=False)   # ids validated by the fused call
            y_deep = self.deep_mlp(x_deep.view(x_deep.shape[0], -1))
            y = self.ensemble_layer(torch.cat([y, y_deep], dim=1))
        return y.squeeze(1)

    def _status(self, dev):
        """the flag word the kernels of this call write (block.IdStatus: the report is deferred to the next call / poll()
        unless check_ids == "sync"); an earlier call's report is raised here, from host memory"""
        if not self.check_ids:
            return None
        self._id_status.raise_if_set()
        return self._id_status.word()

    def _raise_if_bad(self, status):
        if status is not None and self.check_ids == "sync":
            self._id_status.poll(self.embedding.embedding.weight.device)

    def poll(self):
        """synchronise and raise IndexError if any forward so far met an id outside [0, nfeat) (block.IdStatus)"""
        self._id_status.poll(self.embedding.embedding.weight.device)


def _bn_state(bn):
    """(running_mean, running_var, momentum, eps) as the HIP BatchNorm passes take them (HipBatchNorm1d._hip_ok holds)"""
    return bn.running_mean, bn.running_var, float(bn.momentum), float(bn.eps)


class _GcBlockFn(torch.autograd.Function):
    """GC-ARM's block in training mode (gc_arm.py:86-94 under train.py:108-114), fused (round 4):
        forward   lookup * value -> exp -> emb_bn batch statistics (HIP passes) -> armnet_gc_fused_fwd_f32 with THIS batch's
                  emb_bn affine and an identity arm_bn -> arm_bn training passes
        backward  arm_bn reductions -> armnet_gc_fused_bwd_f32 (gates path straight into the table gradient, d_values,
                  d_qfold, and the gradient of emb_bn's output) -> emb_bn backward passes -> * exp(x) -> scatter-add
    The [B, K*H, F] gate / weight tensors of the composed path are never written."""

    @staticmethod
    def forward(ctx, table, bilinear, Q, values, emb_w, emb_b, arm_w, arm_b, ids, vals, cfg, emb_state, arm_state):
        from .modules import _unit_affine
        K, H, E, alpha, n_iter, flags, check_ids = cfg
        B, F = vals.shape
        O = K * H
        dev = vals.device
        check_ids, ist = check_ids
        if check_ids:
            ist.raise_if_set()
        native.clamp_vals(vals)
        ex, sbuf = native.gather_map_stats(ids, vals, table.detach(), 0, ist.word() if check_ids else None)   # exp(lookup * value) + emb_bn's batch sums
        if check_ids == "sync":
            ist.poll(dev)
        e_mean, e_rstd, e_scale, e_shift = native.bn_train_stats(ex, emb_w.detach(), emb_b.detach(), *emb_state, stats=sbuf)
        one, zero, sc, sh = _unit_affine(dev, O)
        qf = torch.empty(O, E, device=dev, dtype=torch.float32)
        native.fold_params(native.GC_ARM, K, H, E, E, bilinear.detach().contiguous(), Q.detach().contiguous(),
                           one, zero, zero, one, 0.0, qf, sc, sh)
        vflat = values.detach().reshape(O, F).contiguous()
        z = torch.empty(B, O, E, device=dev, dtype=torch.float32)
        native.gc_fused_fwd(B, F, E, O, alpha, n_iter, flags, ids, vals, table.detach(), qf, vflat, e_scale, e_shift,
                            one, zero, z, None)
        y, a_mean, a_rstd, _, _ = native.bn_forward_train(z, arm_w.detach(), arm_b.detach(), *arm_state, relu=False)
        ctx.save_for_backward(table, bilinear, Q, values, emb_w, arm_w, ids, vals, qf, vflat, z, ex, e_mean, e_rstd,
                              e_scale, e_shift, a_mean, a_rstd)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        from .modules import _param_grad_buffers
        (table, bilinear, Q, values, emb_w, arm_w, ids, vals, qf, vflat, z, ex, e_mean, e_rstd, e_scale, e_shift, a_mean,
         a_rstd) = ctx.saved_tensors
        K, H, E, alpha, n_iter, flags, _ = ctx.cfg
        B, F = vals.shape
        O = K * H
        dy = dy.contiguous()
        d_aw, d_ab, cA, cB, cC = native.bn_backward_coef(z, dy, arm_w.detach(), a_mean, a_rstd)
        d_table = torch.zeros_like(table)
        d_values, d_qf = _param_grad_buffers(O, F, E, dy.device)
        d_y = torch.empty(B, F, E, device=dy.device, dtype=torch.float32)
        native.gc_fused_bwd(B, F, E, O, alpha, n_iter, flags, ids, vals, table.detach(), qf, vflat, e_scale, e_shift, z, dy,
                            cA, cB, cC, d_table, d_values, d_qf, d_y)
        d_ew, d_eb, eA, eB, eC = native.bn_backward_coef(ex, d_y, emb_w.detach(), e_mean, e_rstd)
        need = ctx.needs_input_grad            # frozen parameters (round-4 advisor finding): no pass / contraction for them
        if need[0]:
            native.bn_bwd_scatter(ids, vals, ex, d_y, eA, eB, eC, 0, d_table)     # emb_bn backward, * exp(x), * value, scatter-add
        g3 = d_qf.view(K, H, E)                                              # q_fold[k,o,x] = sum_y bilinear[k,x,y] Q[k,o,y]
        d_bil = torch.einsum("kox,koy->kxy", g3, Q) if need[1] else None
        d_Q = torch.einsum("kox,kxy->koy", g3, bilinear) if need[2] else None
        return (d_table if need[0] else None, d_bil, d_Q, d_values.reshape(values.shape) if need[3] else None, d_ew, d_eb,
                d_aw, d_ab, None, None, None, None, None)


class _AfnBlockFn(torch.autograd.Function):
    """AFN's block in training mode (afn.py:61-67 under train.py:108-114), fused (round 4): lookup * value -> log -> emb_bn batch
    statistics -> armnet_afn_fused_fwd_f32 with THIS batch's emb_bn affine and an identity afn_bn -> afn_bn training passes;
    backward: afn_bn reductions -> armnet_afn_fused_bwd_f32 (d afn.weight, d afn.bias, the gradient of emb_bn's output) ->
    emb_bn backward passes -> / x -> scatter-add.  The table is already clipped (embedding_clip)."""

    @staticmethod
    def forward(ctx, table, weight, bias, emb_w, emb_b, afn_w, afn_b, ids, vals, cfg, emb_state, afn_state):
        from .modules import _unit_affine
        O, E, flags, check_ids = cfg
        B, F = vals.shape
        dev = vals.device
        check_ids, ist = check_ids
        if check_ids:
            ist.raise_if_set()
        native.clamp_vals(vals)
        lg, sbuf = native.gather_map_stats(ids, vals, table.detach(), 1, ist.word() if check_ids else None)   # log(lookup * value) + emb_bn's batch sums
        if check_ids == "sync":
            ist.poll(dev)
        l_mean, l_rstd, l_scale, l_shift = native.bn_train_stats(lg, emb_w.detach(), emb_b.detach(), *emb_state, stats=sbuf)
        one, zero, _, _ = _unit_affine(dev, O)
        wc = weight.detach().contiguous()
        z = torch.empty(B, O, E, device=dev, dtype=torch.float32)
        native.afn_fused_fwd(B, F, E, O, flags, ids, vals, table.detach(), wc, bias.detach(), l_scale, l_shift, one, zero, z,
                             None)
        y, a_mean, a_rstd, _, _ = native.bn_forward_train(z, afn_w.detach(), afn_b.detach(), *afn_state, relu=False)
        ctx.save_for_backward(table, emb_w, afn_w, ids, vals, wc, z, lg, l_mean, l_rstd, l_scale, l_shift, a_mean, a_rstd)
        ctx.cfg = cfg
        return y

    @staticmethod
    def backward(ctx, dy):
        table, emb_w, afn_w, ids, vals, wc, z, lg, l_mean, l_rstd, l_scale, l_shift, a_mean, a_rstd = ctx.saved_tensors
        O, E, flags, _ = ctx.cfg
        B, F = vals.shape
        dy = dy.contiguous()
        d_aw, d_ab, cA, cB, cC = native.bn_backward_coef(z, dy, afn_w.detach(), a_mean, a_rstd)
        buf = torch.zeros(O * (F + 1), device=dy.device, dtype=torch.float32)
        d_weight, d_bias = buf[:O * F].view(O, F), buf[O * F:]
        d_y = torch.empty(B, F, E, device=dy.device, dtype=torch.float32)
        native.afn_fused_bwd(B, F, E, O, flags, ids, vals, table.detach(), wc, l_scale, l_shift, z, dy, cA, cB, cC, d_weight,
                             d_bias, d_y)
        d_ew, d_eb, eA, eB, eC = native.bn_backward_coef(lg, d_y, emb_w.detach(), l_mean, l_rstd)
        d_table = None
        if ctx.needs_input_grad[0]:            # a frozen table: no dense zeros, no scatter pass (round-4 advisor finding)
            d_table = torch.zeros_like(table)
            native.bn_bwd_scatter(ids, vals, lg, d_y, eA, eB, eC, 1, d_table)     # emb_bn backward, / x = * exp(-log x), * value, scatter-add
        return d_table, d_weight, d_bias, d_ew, d_eb, d_aw, d_ab, None, None, None, None, None


class GC_SparseAttLayer(nn.Module):
    """Sparse attention with global context (gc_arm.py:6-48): Q [nhead, nhid, nemb], bilinear [nhead, nemb, nemb],
    values [nhead, nhid, nfield].  Called on x [B,F,E] it returns the attention weights [B,K,O,F] (stand-alone surface;
    the model forward uses the fused kernel)."""

    def __init__(self, nhead, nfield, nemb, nhid, alpha=1.5):
        super().__init__()
        from utils.entmax import EntmaxBisect
        self.alpha = float(alpha)
        self.sparsemax = nn.Softmax(dim=-1) if alpha == 1. else EntmaxBisect(alpha, dim=-1)
        self.Q = nn.Parameter(torch.zeros(nhead, nhid, nemb))
        nn.init.xavier_uniform_(self.Q, gain=1.414)
        self.bilinear = nn.Parameter(torch.zeros(nhead, nemb, nemb))
        nn.init.xavier_uniform_(self.bilinear, gain=1.414)
        self.values = nn.Parameter(torch.zeros(nhead, nhid, nfield))
        nn.init.xavier_uniform_(self.values, gain=1.414)

    def forward(self, x):
        t = torch.matmul(x.unsqueeze(1), self.bilinear.unsqueeze(0))                    # [B,K,F,E]
        gates = torch.matmul(self.Q.unsqueeze(0), t.transpose(2, 3))                    # [B,K,O,F]
        gates = gates + gates.sum(-1, keepdim=True)                                     # global context
        return self.sparsemax(gates) * self.values


class GC_ARMModel(SiblingBase):
    """Adaptive Relation Modeling Network + Global Context (gc_arm.py:50-105), positional constructor of
    model_utils.py:83-85: GC_ARMModel(nfield, nfeat, nemb, nhead, alpha, arm_hid, mlp_layers, mlp_hid, dropout,
    ensemble, deep_layers, deep_hid)."""

    def __init__(self, nfield, nfeat, nemb, nhead, alpha, arm_hid, mlp_layers, mlp_hid, dropout, ensemble, deep_layers,
                 deep_hid):
        super().__init__()
        self.nfield, self.nfeat, self.nemb = nfield, nfeat, nemb
        self.nhead, self.arm_hid = nhead, arm_hid
        self.ensemble = ensemble
        self.alpha = float(alpha)
        self.n_iter = 50
        self.dropout = nn.Dropout(p=dropout)
        self.embedding = HipEmbedding(nfeat, nemb)
        self.emb_bn = HipBatchNorm1d(nfield)
        self.attn_layers = GC_SparseAttLayer(nhead, nfield, nemb, arm_hid, alpha)
        self.arm_bn = HipBatchNorm1d(nhead * arm_hid)
        self._init_tail(nfield, nfeat, nemb, nhead * arm_hid * nemb, mlp_layers, mlp_hid, dropout, ensemble, deep_layers,
                        deep_hid)

    def _fold(self):
        at = self.attn_layers
        src = [at.Q, at.bilinear] + [t for bn in (self.emb_bn, self.arm_bn)
                                      for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
        if self._fold_key.changed(src) or self._folds is None:
            K, H, E = self.nhead, self.arm_hid, self.nemb
            dev = at.Q.device
            qf = torch.empty(K * H, E, device=dev)
            sc, sh = torch.empty(K * H, device=dev), torch.empty(K * H, device=dev)
            bn = self.arm_bn
            native.fold_params(native.GC_ARM, K, H, E, E, at.bilinear.detach().contiguous(), at.Q.detach().contiguous(),
                               bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, float(bn.eps),
                               qf, sc, sh)
            es, et = self._bn_affine(self.emb_bn)
            self._folds = (qf, sc, sh, es, et)
        return self._folds

    def arm_block(self, ids, vals):
        """ids [B,F], vals [B,F] (clamped in place) -> arm_bn(arm) [B, nhead*arm_hid, nemb] (gc_arm.py:86-94)"""
        B, F = vals.shape
        O, E = self.nhead * self.arm_hid, self.nemb
        qf, sc, sh, es, et = self._fold()
        out = torch.empty(B, O, E, device=vals.device, dtype=torch.float32)
        status = self._status(vals.device)
        native.gc_fused_fwd(B, F, E, O, self.alpha, self.n_iter, self.kernel_flags | native.F_WRITE_CLAMPED_VALS, ids,
                            vals, self.embedding.embedding.weight.detach(), qf,
                            self.attn_layers.values.detach().reshape(O, F), es, et, sc, sh, out, status)
        self._raise_if_bad(status)
        return out

    def forward(self, x, vals=None):
        """x = {'id': Long[B,F], 'value': Float[B,F]} -> logits Float[B] (gc_arm.py:82-105: squeeze(1))"""
        ids, v, v_run = self._inputs(x, vals)
        if self._on_host(ids, v):                                                # gc_arm.py:86-94 on host tensors
            host_ops.clamp_vals_(v_run)
            x_emb = self.embedding({"id": ids, "value": v_run})
            x_exp = self.emb_bn(torch.exp(x_emb))
            arm = torch.einsum("bfe,bkof->bkoe", x_exp, self.attn_layers(x_emb))
            arm = self.arm_bn(arm.reshape(arm.shape[0], -1, self.nemb))
            return self._finish(arm, ids, v, v_run)
        if self._needs_autograd():
            return self._finish(self._arm_block_autograd(ids, v_run), ids, v, v_run)
        return self._finish(self.arm_block(ids, v_run), ids, v, v_run)

    fused_training = True        # developer switch: False keeps the composed device ops for every shape

    def _fused_training_ok(self, ids, v_run):
        """training mode (batch statistics in both BatchNorm1d layers, affine parameters present) on device float32
        tensors and a shape the matrix-core backward has a kernel for; everything else runs the composed ops below"""
        plain = all(bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
                    for bn in (self.emb_bn, self.arm_bn))
        w = self.embedding.embedding.weight
        on_dev = ids.is_cuda and v_run.is_cuda and w.is_cuda and v_run.dtype == torch.float32 and w.dtype == torch.float32
        return (self.fused_training and self.training and plain and on_dev
                and native.gc_fused_bwd_supported(v_run.shape[1], self.nemb, self.nhead * self.arm_hid))

    def _arm_block_autograd(self, ids, v_run):
        """gc_arm.py:86-94 as differentiable device ops (train mode: batch statistics in both BatchNorm1d layers)"""
        from .block import entmax_forward
        B = v_run.shape[0]
        K, H, E = self.nhead, self.arm_hid, self.nemb
        at = self.attn_layers
        if self._fused_training_ok(ids, v_run):
            cfg = (K, H, E, self.alpha, self.n_iter, self.kernel_flags, (self.check_ids, self._id_status))
            w = self.embedding.embedding.weight
            out = _GcBlockFn.apply(w, at.bilinear, at.Q, at.values, self.emb_bn.weight, self.emb_bn.bias,
                                   self.arm_bn.weight, self.arm_bn.bias, ids, v_run, cfg, _bn_state(self.emb_bn),
                                   _bn_state(self.arm_bn))
            self.emb_bn.num_batches_tracked.add_(1)      # only a forward that did not raise (a bad id) counts as a batch
            self.arm_bn.num_batches_tracked.add_(1)
            return out
        x_emb = self._lookup_train(ids, v_run)                                   # [B,F,E]
        x_exp = self.emb_bn(torch.exp(x_emb))                                    # channel = field (gc_arm.py:89)
        qb = torch.einsum("kxy,koy->kox", at.bilinear, at.Q).reshape(K * H, E)   # parameter-only fold of the bilinear form
        # (the weight gradient of this contraction is a [K*H, E] result reduced over B*F rows: split-K, like the heads')
        zb = qb.new_zeros(K * H)
        gates = _LinearSplitKFn.apply(x_emb.reshape(B * x_emb.shape[1], E), qb, zb).view(B, -1, K * H)      # [B, F, K*H]
        # global context (gc_arm.py:37-41): the bilinear form of the FIELD SUM of x — the reference's own formulation, and a
        # reduction over [B, F, E] instead of the 4x larger gate tensor
        gc = _LinearSplitKFn.apply(x_emb.sum(1), qb, zb)                                                    # [B, K*H]
        gates = (gates + gc.unsqueeze(1)).transpose(1, 2)                                                   # [B, K*H, F]
        p = entmax_forward(gates.contiguous(), self.alpha, dim=-1, n_iter=self.n_iter)
        w = p * at.values.reshape(1, K * H, -1)
        arm = torch.bmm(w, x_exp)                                                # [B, K*H, E]
        return self.arm_bn(arm)


class AFNModel(SiblingBase):
    """Adaptive Factorization Network (afn.py:5-77), positional constructor of model_utils.py:40-42:
    AFNModel(nfield, nfeat, nemb, afn_hid, mlp_layers, mlp_hid, dropout, ensemble, deep_layers, deep_hid)."""

    def __init__(self, nfield, nfeat, nemb, afn_hid, mlp_layers, mlp_hid, dropout, ensemble, deep_layers, deep_hid):
        super().__init__()
        self.nfield, self.nfeat = nfield, nfeat
        self.nemb, self.afn_hid = nemb, afn_hid
        self.ensemble = ensemble
        self.dropout = nn.Dropout(p=dropout)
        self.embedding = HipEmbedding(nfeat, nemb)
        self.emb_bn = HipBatchNorm1d(nfield)
        self.afn = nn.Linear(nfield, afn_hid)
        self.afn_bn = HipBatchNorm1d(afn_hid)
        nn.init.normal_(self.afn.weight, std=0.1)
        nn.init.constant_(self.afn.bias, 0.)
        self._init_tail(nfield, nfeat, nemb, afn_hid * nemb, mlp_layers, mlp_hid, dropout, ensemble, deep_layers, deep_hid)
        self._clip_key = _VersionKey()

    def embedding_clip(self):
        """keep AFN embeddings positive (afn.py:74-77): weight.abs_().clamp_(min=1e-4) in place.  Idempotent, so it
        is only re-run when the table changed since the last clip."""
        w = self.embedding.embedding.weight
        if self._clip_key.changed([w]):
            with torch.no_grad():
                if w.is_cuda:
                    native.abs_clamp_min(w.detach(), 1e-4)       # raw in-place write: the version counter is unmoved
                else:
                    w.abs_().clamp_(min=1e-4)
                    self._clip_key.changed([w])

    def _fold(self):
        src = [t for bn in (self.emb_bn, self.afn_bn) for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
        if self._fold_key.changed(src) or self._folds is None:
            self._folds = self._bn_affine(self.emb_bn) + self._bn_affine(self.afn_bn)
        return self._folds

    def afn_block(self, ids, vals):
        """ids [B,F], vals [B,F] (clamped in place) -> afn_bn(exp(afn(emb_bn(log x)))) [B, afn_hid, nemb]
        (afn.py:56-67)"""
        B, F = vals.shape
        O, E = self.afn_hid, self.nemb
        es, et, sc, sh = self._fold()
        out = torch.empty(B, O, E, device=vals.device, dtype=torch.float32)
        status = self._status(vals.device)
        native.afn_fused_fwd(B, F, E, O, self.kernel_flags | native.F_WRITE_CLAMPED_VALS, ids, vals,
                             self.embedding.embedding.weight.detach(), self.afn.weight.detach().contiguous(),
                             self.afn.bias.detach(), es, et, sc, sh, out, status)
        self._raise_if_bad(status)
        return out

    def forward(self, x, vals=None):
        """x = {'id': Long[B,F], 'value': Float[B,F]} -> logits Float[B] (afn.py:49-72: squeeze(1))"""
        ids, v, v_run = self._inputs(x, vals)
        self.embedding_clip()
        if self._on_host(ids, v):                                                # afn.py:56-69 on host tensors
            host_ops.clamp_vals_(v_run)
            x_log = self.emb_bn(torch.log(self.embedding({"id": ids, "value": v_run}))).transpose(1, 2)
            afn = self.afn_bn(torch.exp(self.afn(x_log)).transpose(1, 2))
            return self._finish(self.dropout(afn), ids, v, v_run)
        if self._needs_autograd():
            return self._finish(self._afn_block_autograd(ids, v_run), ids, v, v_run)
        return self._finish(self.afn_block(ids, v_run), ids, v, v_run)

    fused_training = True        # developer switch: False keeps the composed device ops for every shape

    def _fused_training_ok(self, ids, v_run):
        plain = all(bn.training and bn.affine and bn.track_running_stats and bn.momentum is not None
                    for bn in (self.emb_bn, self.afn_bn))
        w = self.embedding.embedding.weight
        on_dev = ids.is_cuda and v_run.is_cuda and w.is_cuda and v_run.dtype == torch.float32 and w.dtype == torch.float32
        return (self.fused_training and self.training and plain and on_dev
                and native.afn_fused_bwd_supported(v_run.shape[1], self.nemb, self.afn_hid))

    def _afn_block_autograd(self, ids, v_run):
        """afn.py:61-69 as differentiable device ops; Dropout (afn.py:69) acts on the block's output"""
        if self._fused_training_ok(ids, v_run):
            cfg = (self.afn_hid, self.nemb, self.kernel_flags, (self.check_ids, self._id_status))
            afn = _AfnBlockFn.apply(self.embedding.embedding.weight, self.afn.weight, self.afn.bias, self.emb_bn.weight,
                                    self.emb_bn.bias, self.afn_bn.weight, self.afn_bn.bias, ids, v_run, cfg,
                                    _bn_state(self.emb_bn), _bn_state(self.afn_bn))
            self.emb_bn.num_batches_tracked.add_(1)      # only a forward that did not raise (a bad id) counts as a batch
            self.afn_bn.num_batches_tracked.add_(1)
            return self.dropout(afn)
        x_emb = self._lookup_train(ids, v_run)                                   # [B,F,E], positive after the clip
        x_log = self.emb_bn(torch.log(x_emb))                                    # channel = field
        Bq, Fq, Eq = x_log.shape                                                 # afn.py:64: Linear over the fields; its weight
        lin = _LinearSplitKFn.apply(x_log.transpose(1, 2).reshape(Bq * Eq, Fq), self.afn.weight, self.afn.bias)   # gradient is a
        afn = torch.exp(lin.view(Bq, Eq, -1))                                    # [O, F] result reduced over B*E rows: split-K
        afn = self.afn_bn(afn.transpose(1, 2).contiguous())                      # [B,O,E]
        return self.dropout(afn)
