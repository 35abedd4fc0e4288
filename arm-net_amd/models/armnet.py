"""models.armnet — multi-head ARM-Net on MI355X (drop-in for the reference's models/armnet.py).

Same class names, constructor order and state_dict keys as the reference (armnet.py:8-75).  For the
fused kernel a K-head block is a one-head block with O = nhead*nhid neurons (channel = k*nhid + o,
the order arm_bn sees after the reference's rearrange, armnet.py:88).
"""
import torch
import torch.nn as nn

from armnet_hip import native
from utils.entmax import EntmaxBisect
from armnet_hip.modules import ArmNetBase, SparseGateBase


class SparseAttLayer(SparseGateBase):
    """Per-head bilinear sparse attention parameters (armnet.py:8-24):
    bilinear_w [nhead, nemb, d_k], query [nhead, nhid, d_k], values [nhead, nhid, nfield]."""

    def __init__(self, nhead, nfield, nemb, d_k, nhid, alpha=1.5):
        super().__init__()
        self.alpha = float(alpha)
        # the reference keeps its normaliser as a sub-module of this name (no parameters: no state_dict keys)
        self.sparsemax = nn.Softmax(dim=-1) if alpha == 1. else EntmaxBisect(alpha, dim=-1)
        self.scale = d_k ** -0.5
        self.bilinear_w = nn.Parameter(torch.zeros(nhead, nemb, d_k))
        self.query = nn.Parameter(torch.zeros(nhead, nhid, d_k))
        self.values = nn.Parameter(torch.zeros(nhead, nhid, nfield))
        self.reset_parameters()

    def reset_parameters(self):
        for p in (self.bilinear_w, self.query, self.values):
            nn.init.xavier_uniform_(p, gain=1.414)

    def _gates(self, x):
        t = torch.matmul(x.unsqueeze(1), self.bilinear_w.unsqueeze(0))               # [B,K,F,D]
        return torch.matmul(self.query.unsqueeze(0), t.transpose(2, 3)) * self.scale  # [B,K,H,F]


class ARMNetModel(ArmNetBase):
    """Adaptive Relation Modeling Network, multi-head variant.

    ARMNetModel(nfield, nfeat, nemb, nhead, alpha, nhid, mlp_nlayer, mlp_nhid, dropout, ensemble,
                deep_nlayer, deep_nhid, noutput=1)   — positional order of armnet.py:44-46,
    as called by the reference's model factory (model_utils.py:44-46); d_k = nemb (armnet.py:66)."""

    variant = native.MULTI_HEAD

    def __init__(self, nfield, nfeat, nemb, nhead, alpha, nhid, mlp_nlayer, mlp_nhid, dropout, ensemble,
                 deep_nlayer, deep_nhid, noutput=1):
        super().__init__()
        self._init_common(nfield, nfeat, nemb, nhead, nhid, alpha, mlp_nlayer, mlp_nhid, dropout, ensemble,
                          deep_nlayer, deep_nhid, noutput,
                          lambda: SparseAttLayer(nhead, nfield, nemb, nemb, nhid, alpha))

    def _d_k(self):
        return self.nemb
