"""utils.entmax — alpha-entmax on MI355X (drop-in for the forward of the reference's utils/entmax.py).

``entmax_bisect(X, alpha, dim, n_iter, ensure_sum_one)`` and ``EntmaxBisect(alpha, dim, n_iter)``
keep the reference's names, argument meaning and defaults (entmax.py:134,238-275) and dispatch to
armnet_entmax_f32.  With n_iter >= 24 and alpha <= 2 the HIP kernel solves the same threshold root by
Newton/Michelot iterations (result within ~1e-6 of the 50-step bisection); otherwise it runs the
reference's bisection step for step.  Differentiable in X (the Jacobian-vector product of
entmax.py:70-80 on the saved output).  A TENSOR alpha (entmax.py:31-36: one alpha per row; or any alpha tensor that requires
grad) runs the reference's bisection per row and is differentiable in X AND in alpha (entmax.py:82-98) (round 6).
"""
import torch.nn as nn

from armnet_hip.block import entmax_forward, entmax_rows_forward


def entmax_bisect(X, alpha=1.5, dim=-1, n_iter=50, ensure_sum_one=True):
    if not isinstance(alpha, (int, float)):
        if alpha.numel() != 1 or alpha.requires_grad:
            # entmax.py:31-36: a tensor alpha, broadcast over every dimension but `dim` — one alpha per row — or an alpha that
            # wants its gradient (entmax.py:82-98) (round 6)
            return entmax_rows_forward(X, alpha, dim=dim, n_iter=n_iter, ensure_sum_one=ensure_sum_one)
        alpha = float(alpha)
    return entmax_forward(X, float(alpha), dim=dim, n_iter=n_iter, ensure_sum_one=ensure_sum_one)


class EntmaxBisect(nn.Module):
    def __init__(self, alpha=1.5, dim=-1, n_iter=50):
        super().__init__()
        self.alpha, self.dim, self.n_iter = alpha, dim, n_iter

    def forward(self, X):
        return entmax_bisect(X, alpha=self.alpha, dim=self.dim, n_iter=self.n_iter)
