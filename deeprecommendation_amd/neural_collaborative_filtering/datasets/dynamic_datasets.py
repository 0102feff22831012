"""DynamicPointwiseDataset (reference datasets/dynamic_datasets.py:6-40): the user profile is built per batch from the
user's rated items by the provider's ``collate_interacted_items``."""
from .base import PointwiseDataset
from ..models.attention_ncf import SparseRatings


def _dev(x, device):
    if isinstance(x, SparseRatings):
        return SparseRatings(x.rowptr.to(device), x.col.to(device), x.val.to(device), x.num_items,
                             None if x.pair_row is None else x.pair_row.to(device))
    return x.float().to(device)


class DynamicPointwiseDataset(PointwiseDataset):
    def __init__(self, file_or_frame, dynamic_provider):
        super().__init__(file_or_frame)
        self.dynamic_provider = dynamic_provider

    def use_collate(self):
        return lambda batch: self.dynamic_provider.collate_interacted_items(batch, for_ranking=False)

    @staticmethod
    def do_forward(model, batch, device, return_attention_weights=False):
        cand_ids, rated_ids, candidate_items, rated_items, user_matrix, y_batch = batch
        res = model(candidate_items.float().to(device), rated_items.float().to(device), _dev(user_matrix, device),
                    return_attention_weights=return_attention_weights)
        if return_attention_weights:
            out, att = res
            return out, y_batch, cand_ids, rated_ids, att, user_matrix
        return res, y_batch
