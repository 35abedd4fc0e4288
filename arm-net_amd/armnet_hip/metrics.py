"""On-device ROC-AUC (the reference's train.py:120 moves every batch to the host and calls
sklearn through utils/utils.py:85-106 `roc_auc_compute_fn`).

`roc_auc_device(y_pred, y_target)` computes the same number as sklearn.metrics.roc_auc_score (ties share their
average rank — the Mann-Whitney form) with a sort and prefix sums on the tensors' own device and returns a 0-dim
float64 tensor without a host sync; `roc_auc_compute_fn` keeps the reference helper's name and contract (Python
float; 0. and a printed message when only one class is present)."""
import torch


def roc_auc_device(y_pred, y_target):
    s = y_pred.detach().reshape(-1)
    t = y_target.detach().reshape(-1).to(s.device)
    if s.numel() != t.numel():
        raise ValueError(f"y_pred has {s.numel()} scores, y_target {t.numel()} labels")
    s, order = torch.sort(s)
    pos = (t[order] > 0).to(torch.float64)
    n = s.numel()
    # average 1-based rank of every tie group
    new_group = torch.ones(n, dtype=torch.bool, device=s.device)
    if n > 1:
        new_group[1:] = s[1:] != s[:-1]
    gid = torch.cumsum(new_group.to(torch.int64), 0) - 1
    counts = torch.bincount(gid).to(torch.float64)
    start = torch.cumsum(counts, 0) - counts
    avg_rank = (start + (counts + 1.0) * 0.5)[gid]
    n_pos = pos.sum()
    n_neg = n - n_pos
    u = (avg_rank * pos).sum() - n_pos * (n_pos + 1.0) * 0.5
    return u / (n_pos * n_neg)            # nan when a class is missing (0/0)


def roc_auc_compute_fn(y_pred, y_target):
    """utils/utils.py:85-106 on the device of the inputs; one host sync for the returned float."""
    auc = float(roc_auc_device(y_pred, y_target))
    if auc != auc:
        print('ValueError: Only one class present in y_true. ROC AUC score is not defined in that case.')
        return 0.
    return auc
