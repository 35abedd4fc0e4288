"""models.armnet_1h — one-head ARM-Net on MI355X (drop-in for the reference's models/armnet_1h.py).

Same class names, constructor order and state_dict keys as the reference (armnet_1h.py:8-74);
forward({'id','value'}) -> logits runs the fused HIP block + the MLP head.
"""
import torch
import torch.nn as nn

from armnet_hip import native
from utils.entmax import EntmaxBisect
from armnet_hip.modules import ArmNetBase, SparseGateBase


class SparseAttention(SparseGateBase):
    """Shared-bilinear ("one-head") sparse attention parameters (armnet_1h.py:8-23):
    bilinear_w = Linear(nemb -> d_k, no bias), query [nhid, d_k], values [nhid, nfield]."""

    def __init__(self, nfield, d_k, nhid, nemb, alpha=1.5):
        super().__init__()
        self.alpha = float(alpha)
        # the reference keeps its normaliser as a sub-module of this name (no parameters: no state_dict keys)
        self.sparsemax = nn.Softmax(dim=-1) if alpha == 1. else EntmaxBisect(alpha, dim=-1)
        self.scale = d_k ** -0.5
        self.bilinear_w = nn.Linear(nemb, d_k, bias=False)
        self.query = nn.Parameter(torch.zeros(nhid, d_k))
        self.values = nn.Parameter(torch.zeros(nhid, nfield))
        self.reset_parameters()

    def reset_parameters(self):
        for p in (self.query, self.values):
            nn.init.xavier_uniform_(p, gain=1.414)

    def _gates(self, x):
        keys = x @ self.bilinear_w.weight.t()                     # [B,F,D]
        return (keys @ self.query.t()).transpose(1, 2) * self.scale   # [B,H,F]


class ARMNetModel(ArmNetBase):
    """Adaptive Relation Modeling Network, one-head variant.

    ARMNetModel(nfield, nfeat, nemb, alpha, nhid, d_k, mlp_nlayer, mlp_nhid, dropout, ensemble,
                deep_nlayer, deep_nhid, noutput=1)   — positional order of armnet_1h.py:42-44,
    as called by the reference's model factory (model_utils.py:47-49)."""

    variant = native.ONE_HEAD

    def __init__(self, nfield, nfeat, nemb, alpha, nhid, d_k, mlp_nlayer, mlp_nhid, dropout, ensemble,
                 deep_nlayer, deep_nhid, noutput=1):
        super().__init__()
        self.d_k = d_k
        self._init_common(nfield, nfeat, nemb, 1, nhid, alpha, mlp_nlayer, mlp_nhid, dropout, ensemble,
                          deep_nlayer, deep_nhid, noutput,
                          lambda: SparseAttention(nfield, d_k, nhid, nemb, alpha))

    def _d_k(self):
        return self.d_k
