"""Evaluation loop of the reference (eval.py:25-182) without plots / W&B: sequential batches through
``Dataset.do_forward`` (so any model of this package or of the reference plugs in), sum-MSE, and per-user NDCG /
min-max adjusted NDCG at cut-offs 5 / 10 / 20."""
from math import sqrt

import numpy as np
import torch
from torch.utils.data import DataLoader

from .datasets.base import PointwiseDataset


def _dcg(y_true, y_score, k):
    """DCG@k with sklearn's tie handling (ties in the scores share the average gain of their tie group)."""
    n = y_true.shape[0]
    disc = 1.0 / np.log2(np.arange(n) + 2.0)
    disc[k:] = 0.0
    cum = np.cumsum(disc)
    _, inv, cnt = np.unique(-y_score, return_inverse=True, return_counts=True)
    gain = np.bincount(inv, weights=y_true, minlength=len(cnt)) / cnt
    ends = np.cumsum(cnt) - 1
    dsum = np.diff(np.concatenate(([0.0], cum[ends])))
    return float(np.dot(gain, dsum))


def eval_ranking(samples_with_preds, cutoff=10):
    """eval.py:25-75.  ``samples_with_preds``: DataFrame with userId, rating, prediction columns.  Users with a single
    row (:38) or with ideal DCG == worst DCG (:57) are ignored; returns (mean NDCG, mean adjusted NDCG)."""
    users = samples_with_preds['userId'].to_numpy()
    rating = samples_with_preds['rating'].to_numpy(dtype=np.float64)
    pred = samples_with_preds['prediction'].to_numpy(dtype=np.float64)
    order = np.argsort(users, kind='stable')
    _, starts = np.unique(users[order], return_index=True)
    bounds = np.append(starts, len(order))
    ndcgs, adjs = [], []
    for a, b in zip(bounds[:-1], bounds[1:]):
        if b - a <= 1:
            continue
        rows = order[a:b]
        t, p = rating[rows], pred[rows]
        dcg, ideal, worst = _dcg(t, p, cutoff), _dcg(t, t, cutoff), _dcg(t, 5.0 - t, cutoff)
        if ideal == worst:
            continue
        ndcgs.append(dcg / ideal if ideal != 0 else 0.0)
        adjs.append((dcg - worst) / (ideal - worst))
    return float(np.mean(ndcgs)), float(np.mean(adjs))


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
