"""Point-wise dataset contract of the reference (datasets/base.py:8-42): (userId, movieId, rating) triplets, a
``use_collate()`` hook, a static ``do_forward(model, batch, device, ...)`` and ``calculate_loss``.
Ranking datasets (BPR loss + negative sampling, base.py:45-99) belong to training and are not mirrored."""
import pandas as pd
from torch import nn
from torch.utils.data import Dataset


class ResidentInputs:
    """Whole-file model inputs for eval_model's device-resident loop: host ``tensors`` (sample order) + ``targets``;
    ``on_chunk(*device_tensors) -> inputs`` runs once per uploaded chunk (raw ids -> table positions), ``on_batch(*inputs,
    y) -> batch`` builds the tuple ``do_forward`` receives (default: ``(*inputs, y)``)."""

    def __init__(self, tensors, targets, on_chunk=None, on_batch=None):
        self.tensors, self.targets, self.on_chunk, self.on_batch = tuple(tensors), targets, on_chunk, on_batch


class PointwiseDataset(Dataset):
    def __init__(self, file_or_frame, use_bce_loss=False):
        self.samples = file_or_frame if isinstance(file_or_frame, pd.DataFrame) else pd.read_csv(str(file_or_frame) + '.csv')
        self.use_bce_loss = use_bce_loss
        self.loss_fn = nn.BCEWithLogitsLoss(reduction='sum') if use_bce_loss else nn.MSELoss(reduction='sum')
        # column arrays once: the reference's per-sample DataFrame.iloc (base.py:25) is 14 ms per 512-batch
        self._u = self.samples['userId'].to_numpy()
        self._i = self.samples['movieId'].to_numpy()
        self._r = self.samples['rating'].to_numpy()

    def __getitem__(self, item):
        r = self._r[item]
        return self._u[item], self._i[item], r / 5.0 if self.use_bce_loss else r

    def __len__(self):
        return len(self.samples)

    def calculate_loss(self, y_pred, y_true):
        return self.loss_fn(y_pred, y_true.view(-1, 1).float())

    def get_graph(self, device):
        return None

    def use_collate(self):
        return None

    def resident_inputs(self, device=None, batch_size=None):
        """A ResidentInputs for datasets whose batch is a pure function of the sample rows (index ids), such that
        ``do_forward(model, batch, ...)`` over its batches equals the DataLoader loop; ``None`` (default): batches need the
        host collate (dense profiles).  eval_model uses it to skip the per-sample Python loop."""
        return None

    def _targets(self):
        import torch
        r = self._r / 5.0 if self.use_bce_loss else self._r
        return torch.as_tensor(r, dtype=torch.float32)

    @staticmethod
    def do_forward(*args, **kwargs):
        raise NotImplementedError
