"""DynamicPointwiseDataset (reference datasets/dynamic_datasets.py:6-40): the user profile is built per batch from the
user's rated items by the provider's ``collate_interacted_items``."""
import numpy as np
import torch

from .base import PointwiseDataset, ResidentInputs
from ..models.attention_ncf import SparseRatings


def _dev(x, device):
    if isinstance(x, SparseRatings):
        return SparseRatings(x.rowptr.to(device), x.col.to(device), x.val.to(device), x.num_items,
                             None if x.pair_row is None else x.pair_row.to(device), x.pairs_per_row_hint)
    return x.float().to(device)


class DynamicPointwiseDataset(PointwiseDataset):
    def __init__(self, file_or_frame, dynamic_provider):
        super().__init__(file_or_frame)
        self.dynamic_provider = dynamic_provider

    def use_collate(self):
        return lambda batch: self.dynamic_provider.collate_interacted_items(batch, for_ranking=False)

    def resident_inputs(self, device=None, batch_size=None):
        """Raw (user id, candidate id) columns; each batch becomes the collate's 6-tuple ON THE GPU from the provider's
        device-resident state (feature table, every user's rated set as one CSR): see SparseDynamicProvider.device_state."""
        dp = self.dynamic_provider
        if device is None or not hasattr(dp, "device_state") or not getattr(dp, "sparse", False):
            return None
        state = dp.device_state(device)
        if state is None or not (np.issubdtype(self._u.dtype, np.integer) and np.issubdtype(self._i.dtype, np.integer)):
            return None
        # pairs per distinct user in a batch, estimated on a few batches of the file (decides grouped vs per-pair kernel)
        bs = int(batch_size or 512)
        starts = np.unique(np.linspace(0, max(0, len(self._u) - bs), num=8).astype(np.int64))
        hint = float(np.mean([len(self._u[s:s + bs]) / max(1, len(np.unique(self._u[s:s + bs]))) for s in starts])) if len(self._u) else 1.0

        def on_batch(upos, cpos, y):
            return state.batch_at(upos, cpos, y, hint)

        # id -> position once per uploaded chunk (a dozen small torch kernels), the batch tuple per batch
        return ResidentInputs((torch.as_tensor(self._u, dtype=torch.int64), torch.as_tensor(self._i, dtype=torch.int64)),
                              self._targets(), on_chunk=state.positions, on_batch=on_batch)

    @staticmethod
    def do_forward(model, batch, device, return_attention_weights=False):
        cand_ids, rated_ids, candidate_items, rated_items, user_matrix, y_batch = batch
        res = model(candidate_items.float().to(device), rated_items.float().to(device), _dev(user_matrix, device),
                    return_attention_weights=return_attention_weights)
        if return_attention_weights:
            out, att = res
            return out, y_batch, cand_ids, rated_ids, att, user_matrix
        return res, y_batch
