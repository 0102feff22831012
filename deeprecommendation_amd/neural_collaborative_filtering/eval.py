"""Evaluation loop of the reference (eval.py:25-182) without plots / W&B: sequential batches through
``Dataset.do_forward`` (so any model of this package or of the reference plugs in), sum-MSE, and per-user NDCG /
min-max adjusted NDCG at cut-offs 5 / 10 / 20."""
from math import sqrt

import numpy as np
import torch
from torch.utils.data import DataLoader

from .datasets.base import PointwiseDataset


def _dcg_per_user(user_idx, n_users, y_true, y_score, k):
    """DCG@k of every user at once, with sklearn's tie handling (dcg_score(ignore_ties=False)): inside a user, items
    with equal scores share the average gain of their tie group.  Fully vectorised: one lexsort over all rows."""
    order = np.lexsort((-y_score, user_idx))                 # by user, then by descending score
    u, t, sc = user_idx[order], y_true[order], y_score[order]
    first_of_user = np.r_[True, u[1:] != u[:-1]]
    start = np.maximum.accumulate(np.where(first_of_user, np.arange(len(u)), 0))
    rank = np.arange(len(u)) - start                          # 0-based rank inside the user
    disc = np.where(rank < k, 1.0 / np.log2(rank + 2.0), 0.0)
    new_group = first_of_user | np.r_[True, sc[1:] != sc[:-1]]
    gid = np.cumsum(new_group) - 1
    cnt = np.bincount(gid)
    gain = np.bincount(gid, weights=t) / cnt                  # mean gain of each tie group
    dsum = np.bincount(gid, weights=disc)                     # discounts the group's positions collect
    return np.bincount(u[new_group], weights=gain * dsum, minlength=n_users)


def eval_ranking(samples_with_preds, cutoff=10):
    """eval.py:25-75.  ``samples_with_preds``: DataFrame with userId, rating, prediction columns.  Users with a single
    row (:38) or with ideal DCG == worst DCG (:57) are ignored; returns (mean NDCG, mean adjusted NDCG).
    The reference loops over users calling sklearn three times each; this computes all users in one pass."""
    users = samples_with_preds['userId'].to_numpy()
    rating = samples_with_preds['rating'].to_numpy(dtype=np.float64)
    pred = samples_with_preds['prediction'].to_numpy(dtype=np.float64)
    _, uidx, counts = np.unique(users, return_inverse=True, return_counts=True)
    n = len(counts)
    dcg = _dcg_per_user(uidx, n, rating, pred, cutoff)
    ideal = _dcg_per_user(uidx, n, rating, rating, cutoff)
    worst = _dcg_per_user(uidx, n, rating, 5.0 - rating, cutoff)
    keep = (counts > 1) & (ideal != worst)
    ndcg = np.where(ideal[keep] != 0, dcg[keep] / np.where(ideal[keep] != 0, ideal[keep], 1.0), 0.0)
    adj = (dcg[keep] - worst[keep]) / (ideal[keep] - worst[keep])
    return float(np.mean(ndcg)), float(np.mean(adj))


def eval_model(model, test_dataset: PointwiseDataset, batch_size, ranking=False, device=None, verbose=False):
    """eval.py:78-182 (metrics only).  Returns a dict: predictions, mse, rmse, ndcg@k / adj_ndcg@k for k = 5, 10, 20."""
    assert isinstance(test_dataset, PointwiseDataset), 'Should only be testing on pointwise datasets.'
    device = device or next(model.parameters()).device
    model.to(device)
    loader = DataLoader(test_dataset, batch_size=batch_size, collate_fn=test_dataset.use_collate())  # sequential order
    graph = test_dataset.get_graph(device)
    extra = [] if graph is None else [graph]
    model.eval()
    fitted, total = [], 0.0
    with torch.no_grad():
        for batch in loader:
            out, y = test_dataset.__class__.do_forward(model, batch, device, *extra)
            if not ranking:
                total += test_dataset.calculate_loss(out, y.to(device)).item()
            fitted.append(out.detach().cpu().numpy())
    if device is not None and torch.device(device).type == "cuda":
        from .. import native
        native.check_oob(torch.device(device))  # an out-of-range id anywhere in the run raises IndexError here
    pred = np.concatenate(fitted).astype(np.float64).reshape(-1)
    res = {"predictions": pred}
    if not ranking:
        res["mse"] = total / len(test_dataset)
        res["rmse"] = sqrt(res["mse"])
    frame = test_dataset.samples.assign(prediction=pred)
    for k in (5, 10, 20):
        res[f"ndcg@{k}"], res[f"adj_ndcg@{k}"] = eval_ranking(frame, cutoff=k)
    if verbose:
        print({k: v for k, v in res.items() if k != "predictions"})
    return res
