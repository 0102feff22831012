"""FixedPointwiseDataset (reference datasets/fixed_datasets.py:8-30): fixed user / item vectors from a ContentProvider.
The collate emits int64 tensors when the provider returns positions (table path) and float tensors otherwise."""
import numpy as np
import torch

from .base import PointwiseDataset, ResidentInputs


def _to_tensor(x):
    x = np.asarray(x)
    return torch.from_numpy(x.astype(np.int64)) if np.issubdtype(x.dtype, np.integer) and x.ndim == 1 else torch.as_tensor(x, dtype=torch.float32)


class FixedPointwiseDataset(PointwiseDataset):
    def __init__(self, file_or_frame, content_provider):
        super().__init__(file_or_frame)
        self.content_provider = content_provider

    def use_collate(self):
        cp = self.content_provider

        def custom_collate(batch):
            users, items, targets = zip(*batch)
            return _to_tensor(cp.get_user_profile(userID=users)), _to_tensor(cp.get_item_profile(itemID=items)), torch.as_tensor(np.asarray(targets), dtype=torch.float32)

        return custom_collate

    def resident_inputs(self, device=None, batch_size=None):
        cp = self.content_provider
        probe_u, probe_i = np.asarray(cp.get_user_profile(userID=self._u[:1])), np.asarray(cp.get_item_profile(itemID=self._i[:1]))
        if not (np.issubdtype(probe_u.dtype, np.integer) and probe_u.ndim == 1 and np.issubdtype(probe_i.dtype, np.integer) and probe_i.ndim == 1):
            return None  # dense profile rows: built batch by batch
        lookup = cp.device_lookup(device) if device is not None and hasattr(cp, "device_lookup") else None
        if lookup is not None and np.issubdtype(self._u.dtype, np.integer) and np.issubdtype(self._i.dtype, np.integer):
            return ResidentInputs((_to_tensor(self._u), _to_tensor(self._i)), self._targets(), on_chunk=lookup)  # raw ids; positions resolved on the GPU
        return ResidentInputs((_to_tensor(cp.get_user_profile(userID=self._u)), _to_tensor(cp.get_item_profile(itemID=self._i))), self._targets())

    @staticmethod
    def do_forward(model, batch, device):
        user_vec, item_vec, y_batch = batch
        if user_vec.dtype != torch.int64:
            user_vec, item_vec = user_vec.float(), item_vec.float()
        return model(user_vec.to(device), item_vec.to(device)), y_batch
