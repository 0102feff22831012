"""Point-wise training loop of the reference (train.py:21-227: epochs of zero_grad -> do_forward -> loss -> backward ->
step, validation loss + NDCG per epoch, checkpoint of the best validation loss, patience-based early stopping), with the
same arguments and the same returned ``monitored_metrics``; tqdm / W&B output is reduced to optional ``wandb.log`` calls.

What differs is how batches reach the GPU.  For datasets whose batch is a pure function of the sample rows
(``Dataset.resident_inputs``: index ids) the whole training file is uploaded ONCE, every epoch draws a permutation on the
device and a batch is a gather of ids — no per-sample ``__getitem__``, no collate, no per-batch host-to-device copy — and
the running loss stays on the GPU (the reference synchronises with ``loss.item()`` and copies ``out`` to the host every
batch, train.py:107-110).  Other datasets run the reference-shaped ``DataLoader(shuffle=True)`` loop.  The default
optimiser on a GPU is ``FusedAdam`` (torch.optim.Adam's update as one kernel per tensor).  Pair-wise (BPR) training with
negative sampling (train.py:79-82, datasets/base.py:45-99) is not mirrored.
"""
import numpy as np
import torch
from torch import optim
from torch.utils.data import DataLoader

from .datasets.base import PointwiseDataset
from .eval import eval_model
from .util import cap_host_threads, load_model


class EarlyStopping:
    """The reference's stopping rule (train.py:157-210) as a small state machine over the validation loss.

    * a new overall best: remember it (the caller checkpoints), strikes back to 0, budget back to ``max_patience``;
    * otherwise: one strike if the loss also rose against the PREVIOUS epoch, else one strike is forgiven (never below
      0); the budget shrinks by one regardless; stop when strikes exceed ``patience`` or the budget is used up.
    """

    def __init__(self, patience=3, max_patience=5):
        self.patience, self.full_budget = patience, max_patience
        self.budget, self.strikes = max_patience, 0
        self.best, self.previous, self.best_epoch = None, None, -1

    def update(self, value: float, epoch: int) -> str:
        """'best' (checkpoint now), 'stop', or 'continue'."""
        if self.best is None or value < self.best:
            self.best, self.best_epoch = value, epoch
            self.strikes, self.budget = 0, self.full_budget
            verdict = "best"
        else:
            if self.previous is not None and value > self.previous:
                self.strikes += 1
            else:
                self.strikes = max(0, self.strikes - 1)
            self.budget -= 1
            verdict = "stop" if (self.strikes > self.patience or self.budget <= 0) else "continue"
        self.previous = value
        return verdict


def _resident_training_inputs(dataset, device, batch_size):
    """(inputs on the device, targets on the device) for the whole file, or None."""
    if torch.device(device).type != "cuda":
        return None
    res = dataset.resident_inputs(torch.device(device), batch_size)
    if res is None or res.on_batch is not None:
        return None
    dev = [t.pin_memory().to(device, non_blocking=True) for t in (*res.tensors, res.targets)]
    inputs = list(res.on_chunk(*dev[:-1])) if res.on_chunk is not None else dev[:-1]
    return inputs, dev[-1]


def train_model(model, train_dataset, val_dataset: PointwiseDataset, lr, weight_decay, batch_size, val_batch_size, early_stop,
                final_model_path='final_model.pt', checkpoint_model_path='temp.pt', max_epochs=100, patience=3, max_patience=5,
                optimizer=None, ndcg_cutoff=10, wandb=None, num_workers=0, device=None, resident=None, shuffle=True, verbose=True):
    """train.py:21-227 for point-wise datasets.  Extra keyword arguments: ``device`` (default: cuda:0 when there is one),
    ``resident`` (None = device-resident batches when the dataset allows, False = DataLoader loop), ``shuffle`` (tests)."""
    if not isinstance(train_dataset, PointwiseDataset) or not isinstance(val_dataset, PointwiseDataset):
        raise NotImplementedError("only point-wise training is mirrored (pair-wise BPR training stays with the reference loop)")
    device = torch.device(device) if device is not None else torch.device("cuda:0" if torch.cuda.is_available() else "cpu")
    cap_host_threads()
    model.to(device)
    if not model.is_dataset_compatible(train_dataset.__class__) or not model.is_dataset_compatible(val_dataset.__class__):
        raise Exception('Model used is incompatible with this dataset.')
    say = print if verbose else (lambda *a, **k: None)
    say('Training size:', len(train_dataset), ' - Validation size:', len(val_dataset))

    if optimizer is None:
        if device.type == "cuda":
            from ..optim import FusedAdam
            optimizer = FusedAdam(model.parameters(), lr=lr, weight_decay=weight_decay)
        else:
            optimizer = optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)

    train_graph = train_dataset.get_graph(device)
    extra = [] if train_graph is None else [train_graph]
    held = _resident_training_inputs(train_dataset, device, batch_size) if resident is not False else None
    if resident and held is None:
        raise ValueError("resident training needs a CUDA device and a dataset with on-device batches")
    loader = None
    if held is None:
        loader = DataLoader(train_dataset, batch_size=batch_size, shuffle=shuffle, collate_fn=train_dataset.use_collate(), num_workers=num_workers)

    stopper = EarlyStopping(patience, max_patience)
    monitored_metrics = {'train_loss': [], 'val_loss': [], 'val_ndcg': []}
    best_ndcg = -1.0
    do_forward = train_dataset.__class__.do_forward
    n = len(train_dataset)

    for epoch in range(max_epochs):
        say(f'\nEpoch {epoch + 1}')
        model.train()
        if held is not None:
            inputs, targets = held
            order = torch.randperm(n, device=device) if shuffle else None
            running = torch.zeros((), dtype=torch.float64, device=device)
            for s in range(0, n, batch_size):
                pick = order[s:s + batch_size] if order is not None else slice(s, s + batch_size)
                batch = (*[t[pick] for t in inputs], targets[pick])
                optimizer.zero_grad()
                out, y = do_forward(model, batch, device, *extra)
                loss = train_dataset.calculate_loss(out, y)
                loss.backward()
                optimizer.step()
                running += loss.detach().double()
            train_sum_loss = float(running.item())   # the epoch's only host synchronisation
        else:
            train_sum_loss = 0.0
            for batch in loader:
                optimizer.zero_grad()
                out, y = do_forward(model, batch, device, *extra)
                loss = train_dataset.calculate_loss(out, y.to(device))
                loss.backward()
                optimizer.step()
                train_sum_loss += loss.detach().item()
        train_loss = train_sum_loss / max(1, n)
        monitored_metrics['train_loss'].append(train_loss)
        say(f'Training loss: {train_loss:.4f}')

        val = eval_model(model, val_dataset, val_batch_size, ranking=False, device=device, resident=resident, cutoffs=(ndcg_cutoff,))
        val_loss, val_ndcg, val_adj = val["mse"], val[f"ndcg@{ndcg_cutoff}"], val[f"adj_ndcg@{ndcg_cutoff}"]
        monitored_metrics['val_loss'].append(val_loss)
        monitored_metrics['val_ndcg'].append(val_ndcg)
        val_dataset.samples['prediction'] = val["predictions"]   # as train.py:137 leaves it
        say(f'Validation loss: {val_loss:.4f} - Validation NDCG@{ndcg_cutoff}: {val_ndcg:.4f}, adj-NDCG@{ndcg_cutoff}: {val_adj:.4f}')
        best_ndcg = max(best_ndcg, val_ndcg) if not np.isnan(val_ndcg) else best_ndcg
        if wandb is not None:
            wandb.log({'train_loss': train_loss, f'val_ndcg@{ndcg_cutoff}': val_ndcg, f'val_adj_ndcg@{ndcg_cutoff}': val_adj,
                       'epoch': epoch + 1, 'val_loss': val_loss})

        if early_stop:
            verdict = stopper.update(val_loss, epoch)
            if verdict == "best":
                model.save_model(checkpoint_model_path)
            elif verdict == "stop" or epoch == max_epochs - 1:
                say(f'{"Early stopping" if verdict == "stop" else "Last epoch"} at epoch {epoch + 1}: restoring the checkpoint of epoch '
                    f'{stopper.best_epoch + 1} (val loss {stopper.best:.4f}).')
                state, _ = load_model(checkpoint_model_path, map_location=device)
                model.load_state_dict(state)
                model.eval()
                if verdict == "stop":
                    break
            say(f'Patience remaining: {patience - stopper.strikes}')

    if wandb is not None:
        logs = {'best_ndcg@10': best_ndcg}
        if stopper.best is not None:
            logs['best_val_loss'] = stopper.best
        wandb.log(logs)
    if final_model_path is not None:
        model.save_model(final_model_path)
    return monitored_metrics
