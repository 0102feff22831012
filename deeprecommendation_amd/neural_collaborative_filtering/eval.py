"""Evaluation loop of the reference (eval.py:25-182) without plots / W&B: sequential batches through
``Dataset.do_forward`` (so any model of this package or of the reference plugs in), sum-MSE, and per-user NDCG /
min-max adjusted NDCG at cut-offs 5 / 10 / 20."""
from math import sqrt

import numpy as np
import torch
from torch.utils.data import DataLoader

from .datasets.base import PointwiseDataset
from .util import cap_host_threads


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


def _dcg_orders(uidx, y_score):
    """Rows ordered by user, then by descending score (ties in input order) — the np.lexsort of _dcg_per_user as two
    stable device sorts; independent of the cut-off, so one ordering serves every k."""
    by_score = torch.argsort(y_score, descending=True, stable=True)
    return by_score[torch.argsort(uidx[by_score], stable=True)]


def _dcg_per_user_torch(order, uidx, n_users, y_true, y_score, cutoffs):
    """_dcg_per_user on tensors (any device, float64), for several cut-offs from one ordering."""
    u, t, sc = uidx[order], y_true[order], y_score[order]
    n = u.numel()
    idx = torch.arange(n, device=u.device)
    first = torch.ones(n, dtype=torch.bool, device=u.device)
    first[1:] = u[1:] != u[:-1]
    rank = idx - torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values   # 0-based rank inside the user
    new_group = first.clone()
    new_group[1:] |= sc[1:] != sc[:-1]
    gid = torch.cumsum(new_group, 0) - 1
    n_groups = int(gid[-1].item()) + 1 if n else 0
    cnt = torch.bincount(gid, minlength=n_groups).to(torch.float64)
    gain = torch.bincount(gid, weights=t, minlength=n_groups) / cnt     # mean gain of each tie group
    owner = u[new_group]
    out = []
    for k in cutoffs:
        disc = torch.where(rank < k, 1.0 / torch.log2(rank.to(torch.float64) + 2.0), torch.zeros((), dtype=torch.float64, device=u.device))
        dsum = torch.bincount(gid, weights=disc, minlength=n_groups)    # discounts the group's positions collect
        out.append(torch.bincount(owner, weights=gain * dsum, minlength=n_users))
    return out


def eval_ranking_device(users, rating, pred, cutoffs=(5, 10, 20), device=None):
    """eval_ranking for several cut-offs with the sorts and segment sums on ``device`` (three stable sort pairs in all,
    instead of nine host lexsorts): ``{k: (mean NDCG, mean adjusted NDCG)}``.  Same definition and tie handling as
    eval_ranking; float64 throughout (bincount's atomic adds make the last bits run-dependent)."""
    device = torch.device(device) if device is not None else (pred.device if torch.is_tensor(pred) else torch.device("cpu"))
    as_t = lambda x, dt: (x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))).to(device=device, dtype=dt).reshape(-1)
    users, rating, pred = as_t(users, torch.int64), as_t(rating, torch.float64), as_t(pred, torch.float64)
    if users.numel() == 0:
        return {k: (float("nan"), float("nan")) for k in cutoffs}
    _, uidx, counts = torch.unique(users, return_inverse=True, return_counts=True)
    n = counts.numel()
    dcg = _dcg_per_user_torch(_dcg_orders(uidx, pred), uidx, n, rating, pred, cutoffs)
    ideal = _dcg_per_user_torch(_dcg_orders(uidx, rating), uidx, n, rating, rating, cutoffs)
    worst = _dcg_per_user_torch(_dcg_orders(uidx, 5.0 - rating), uidx, n, rating, 5.0 - rating, cutoffs)
    res = {}
    for j, k in enumerate(cutoffs):
        keep = (counts > 1) & (ideal[j] != worst[j])
        d, i, w = dcg[j][keep], ideal[j][keep], worst[j][keep]
        ndcg = torch.where(i != 0, d / torch.where(i != 0, i, torch.ones_like(i)), torch.zeros_like(i))
        res[k] = (float(ndcg.mean().item()), float(((d - w) / (i - w)).mean().item())) if d.numel() else (float("nan"), float("nan"))
    return res


RESIDENT_CHUNK_BATCHES = 64   # batches uploaded per copy on the side stream


def _resident_batches(res, batch_size, device):
    """Yield device-resident batches ``(*inputs[s:e], y[s:e])`` of whole-file host tensors.  Uploads go RESIDENT_CHUNK_BATCHES
    batches at a time through two reused sets of pinned staging + device buffers on a side stream, one chunk ahead of the
    kernels that consume them: the compute stream waits on the chunk's copy event, the copy stream on the event that marks
    the previous user of the buffer set as enqueued-and-done.  No per-sample Python, no per-batch host-to-device copy, no
    allocation or page-pinning per chunk, and the staging copy is a plain memcpy (an OpenMP copy on the thread that enqueues
    GPU work is what exposed the thread-pool throttling util.cap_host_threads guards against)."""
    host = (*res.tensors, res.targets)
    n = len(res.targets)
    if n == 0:
        return
    chunk = min(n, max(1, RESIDENT_CHUNK_BATCHES) * batch_size)
    side, main = torch.cuda.Stream(device), torch.cuda.current_stream(device)
    sets = 2 if n > chunk else 1
    pinned = [[torch.empty(chunk, dtype=t.dtype).pin_memory() for t in host] for _ in range(sets)]
    devbuf = [[torch.empty(chunk, dtype=t.dtype, device=device) for t in host] for _ in range(sets)]
    copied, consumed = [None] * sets, [None] * sets
    # devbuf came from the caching allocator on the compute stream: its blocks may still be in use by kernels queued there
    # (a direct eval_model call after other work) — the copy stream must not write them before those have finished
    side.wait_stream(main)

    def upload(ci):
        b, c0 = ci % sets, ci * chunk
        m = min(n, c0 + chunk) - c0
        if copied[b] is not None:
            copied[b].synchronize()                 # the staging buffers' previous upload has left the host
        for p, t in zip(pinned[b], host):
            np.copyto(p[:m].numpy(), t[c0:c0 + m].numpy())   # plain memcpy: no OpenMP region on the enqueueing thread
        with torch.cuda.stream(side):
            if consumed[b] is not None:
                side.wait_event(consumed[b])        # kernels reading this device buffer set have finished
            for d, p in zip(devbuf[b], pinned[b]):
                d[:m].copy_(p[:m], non_blocking=True)
            copied[b] = torch.cuda.Event()
            copied[b].record(side)
        return b, m

    n_chunks = (n + chunk - 1) // chunk
    nxt = upload(0)
    for ci in range(n_chunks):
        b, m = nxt
        main.wait_event(copied[b])
        # the next upload reuses the OTHER set; with one chunk ahead its previous reader is the chunk before this one
        nxt = upload(ci + 1) if ci + 1 < n_chunks else None
        dev = [d[:m] for d in devbuf[b]]
        if res.on_chunk is not None:
            dev = [*res.on_chunk(*dev[:-1]), dev[-1]]  # e.g. raw ids -> table positions, one gather per chunk on the GPU
        for s in range(0, m, batch_size):
            batch = tuple(t[s:s + batch_size] for t in dev)
            yield batch if res.on_batch is None else res.on_batch(*batch)
        consumed[b] = torch.cuda.Event()
        consumed[b].record(main)


def eval_model(model, test_dataset: PointwiseDataset, batch_size, ranking=False, device=None, verbose=False, resident=None,
               cutoffs=(5, 10, 20)):
    """eval.py:78-182 (metrics only).  Returns a dict: predictions, mse, rmse, ndcg@k / adj_ndcg@k for k in ``cutoffs``
    (the reference reports 5, 10, 20).

    ``resident`` (None = when possible): datasets whose inputs are index ids (`Dataset.resident_inputs`) are evaluated
    without the DataLoader — same batches, same ``do_forward`` plug-in call, same per-batch loss sums accumulated in
    double like the reference's ``loss.item()`` additions — but ids are uploaded in large chunks on a copy stream, the
    loss and the predictions stay on the GPU, and the host synchronises ONCE at the end instead of twice per batch."""
    assert isinstance(test_dataset, PointwiseDataset), 'Should only be testing on pointwise datasets.'
    cap_host_threads()
    device = device or next(model.parameters()).device
    model.to(device)
    on_gpu = torch.device(device).type == "cuda"
    host = test_dataset.resident_inputs(torch.device(device), batch_size) if (resident is not False and on_gpu) else None
    if resident and host is None:
        raise ValueError("resident evaluation needs a CUDA device and a dataset with resident_inputs()")
    graph = test_dataset.get_graph(device)
    extra = [] if graph is None else [graph]
    model.eval()
    fitted, total, pred_dev = [], 0.0, None
    with torch.no_grad():
        if host is not None:
            total_dev = torch.zeros((), dtype=torch.float64, device=device)
            outs = []
            for batch in _resident_batches(host, batch_size, torch.device(device)):
                out, y = test_dataset.__class__.do_forward(model, batch, device, *extra)
                if not ranking:
                    total_dev += test_dataset.calculate_loss(out, y).double()
                outs.append(out)
            total = float(total_dev.item())
            pred_dev = torch.cat(outs).reshape(-1) if outs else None
            fitted = [pred_dev.cpu().numpy()] if outs else [np.zeros((0, 1), dtype=np.float32)]
        else:
            loader = DataLoader(test_dataset, batch_size=batch_size, collate_fn=test_dataset.use_collate())  # sequential order
            for batch in loader:
                out, y = test_dataset.__class__.do_forward(model, batch, device, *extra)
                if not ranking:
                    total += test_dataset.calculate_loss(out, y.to(device)).item()
                fitted.append(out.detach().cpu().numpy())
    if on_gpu:
        from .. import native
        native.check_oob(torch.device(device))  # an out-of-range id anywhere in the run raises IndexError here
    pred = np.concatenate(fitted).astype(np.float64).reshape(-1)
    res = {"predictions": pred}
    if not ranking:
        res["mse"] = total / len(test_dataset)
        res["rmse"] = sqrt(res["mse"])
    ranked = None
    if np.issubdtype(test_dataset.samples['userId'].to_numpy().dtype, np.integer):
        # sorts and segment sums of the ranking metrics on the model's device (nine host lexsorts cost 1.3-4.4 s per
        # million samples; three device sort pairs serve all three cut-offs)
        ranked = eval_ranking_device(test_dataset.samples['userId'].to_numpy(), test_dataset.samples['rating'].to_numpy(dtype=np.float64),
                                     pred_dev if pred_dev is not None else pred, tuple(cutoffs), device)
    else:
        frame = test_dataset.samples.assign(prediction=pred)
    for k in cutoffs:
        res[f"ndcg@{k}"], res[f"adj_ndcg@{k}"] = ranked[k] if ranked is not None else eval_ranking(frame, cutoff=k)
    if verbose:
        print({k: v for k, v in res.items() if k != "predictions"})
    return res
