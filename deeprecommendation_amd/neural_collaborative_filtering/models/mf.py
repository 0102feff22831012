"""MF — drop-in for reference models/mf.py: two embedding Linears and a row-wise dot product (mf.py:28-32)."""
import torch
from torch import nn

from ... import native
from ..util import require_gpu, row_major_embedding_, use_native
from .base import NCF
from .basic_ncf import _ScoringMixin


class MF(_ScoringMixin, NCF):
    compatible_datasets = ("FixedPointwiseDataset", "FixedRankingDataset")

    def __init__(self, item_dim, user_dim, item_emb=128, user_emb=128):
        super().__init__()
        self.kwargs = {'item_dim': item_dim, 'user_dim': user_dim, 'item_emb': item_emb, 'user_emb': user_emb}
        self.item_embeddings = nn.Sequential(row_major_embedding_(nn.Linear(item_dim, item_emb)))
        self.user_embeddings = nn.Sequential(row_major_embedding_(nn.Linear(user_dim, user_emb)))

    def get_model_parameters(self) -> dict:
        return self.kwargs

    def forward(self, X_user, X_item):
        indexed = X_user.dtype == torch.int64 and X_user.dim() == 1
        if not use_native(self):
            if indexed:
                ue, ie = self.user_embeddings[0], self.item_embeddings[0]
                u, i = ue.weight.t()[X_user] + ue.bias, ie.weight.t()[X_item] + ie.bias
            else:
                u, i = self.user_embeddings(X_user), self.item_embeddings(X_item)
            return torch.bmm(u.unsqueeze(1), i.unsqueeze(2)).view(-1, 1)
        require_gpu(X_user, X_item)
        cache = self._refresh()  # one parameter fingerprint per forward
        if indexed:
            return native.gather_dot(self._table("user", self.user_embeddings[0], cache), X_user.contiguous(),
                                     self._table("item", self.item_embeddings[0], cache), X_item.contiguous())
        ue, ie = self.user_embeddings[0], self.item_embeddings[0]
        u = native.linear(X_user.float().contiguous(), self._dense_weight("user", ue, cache), ue.bias.detach())
        i = native.linear(X_item.float().contiguous(), self._dense_weight("item", ie, cache), ie.bias.detach())
        return native.gather_dot(u, None, i, None, B=u.shape[0])
