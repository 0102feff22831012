"""Plugin surface of the scoring path: what a user of the framework implements to feed their own content to the models.

Three provider kinds exist, one per model family (the names and method signatures are the contract the datasets and
evaluation loop call, identical to the upstream project's so existing provider subclasses keep working):

  ContentProvider          fixed per-user / per-item vectors        -> BasicNCF, MF
  DynamicContentProvider   user profile built from rated items      -> AttentionNCF
  GraphContentProvider     user-item interaction graph + node ids   -> GraphNCF

A provider may return *positions* (int64, rank of the id among the sorted unique ids) instead of float rows from
``get_user_profile`` / ``get_item_profile``; the datasets then take the embedding-table path on the GPU.
"""
from abc import ABC, abstractmethod


class _CatalogueInfo(ABC):
    """Sizes every provider reports."""

    @abstractmethod
    def get_num_items(self) -> int:
        """Number of distinct items known to the provider."""

    @abstractmethod
    def get_num_users(self) -> int:
        """Number of distinct users known to the provider."""


class _ItemContent(_CatalogueInfo):
    @abstractmethod
    def get_item_profile(self, itemID):
        """Vector(s) for one item id or a tuple of ids (one row per id)."""

    @abstractmethod
    def get_item_feature_dim(self) -> int:
        """Width of an item vector (== model ``item_dim``)."""


class ContentProvider(_ItemContent):
    @abstractmethod
    def get_user_profile(self, userID):
        """Vector(s) for one user id or a tuple of ids (one row per id)."""


class DynamicContentProvider(_ItemContent):
    @abstractmethod
    def collate_interacted_items(self, batch, for_ranking: bool):
        """collate_fn for a batch of (userId, itemId, target | item2Id) samples.  Returns the 6-tuple
        ``(candidate_ids, rated_ids, candidate_items (B, F), rated_items (I, F), user_matrix, targets_or_items2)``
        where ``user_matrix`` is the (B, I) matrix of normalised ratings of the batch's users over the union of
        their rated items (0 = unrated), column order == row order of ``rated_items`` — or its CSR form
        (``models.attention_ncf.SparseRatings``)."""


class GraphContentProvider(_CatalogueInfo):
    @abstractmethod
    def get_user_nodeID(self, userID) -> int:
        """Graph node id of a user (users come after all items)."""

    @abstractmethod
    def get_item_nodeID(self, itemID) -> int:
        """Graph node id of an item."""

    @abstractmethod
    def get_graph(self):
        """The interaction graph (``models.gnn_ncf.GraphData``)."""
