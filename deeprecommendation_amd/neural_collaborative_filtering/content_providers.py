"""Plugin surface of the reference (neural_collaborative_filtering/content_providers.py:4-62): the three abstract
provider classes a user extends.  Same method names and meanings; no PyG import (``get_graph`` returns a
``GraphData``, which carries the attributes the reference's PyG ``Data`` object does)."""


class ContentProvider:
    """Fixed per-item / per-user input vectors (reference content_providers.py:4-19).  ``get_*_profile`` is called
    with a tuple of ids per batch and returns something stackable: float rows, or — for the table path — int64
    positions (rank of the id among the sorted unique ids)."""

    def get_item_profile(self, itemID):
        raise NotImplementedError

    def get_user_profile(self, userID):
        raise NotImplementedError

    def get_num_items(self):
        raise NotImplementedError

    def get_num_users(self):
        raise NotImplementedError

    def get_item_feature_dim(self):
        raise NotImplementedError


class DynamicContentProvider:
    """User profile built from the user's rated items (reference content_providers.py:22-44)."""

    def get_item_profile(self, itemID):
        raise NotImplementedError

    def get_num_items(self):
        raise NotImplementedError

    def get_num_users(self):
        raise NotImplementedError

    def get_item_feature_dim(self):
        raise NotImplementedError

    def collate_interacted_items(self, batch, for_ranking: bool):
        """Returns (candidate_ids, rated_ids, candidate_items (B,F), rated_items (I,F), user_matrix (B,I) or a
        SparseRatings, targets | items2) — the 6-tuple consumed at datasets/dynamic_datasets.py:27,57."""
        raise NotImplementedError


class GraphContentProvider:
    """Graph + node ids (reference content_providers.py:47-62)."""

    def get_num_items(self):
        raise NotImplementedError

    def get_num_users(self):
        raise NotImplementedError

    def get_user_nodeID(self, userID) -> int:
        raise NotImplementedError

    def get_item_nodeID(self, itemID) -> int:
        raise NotImplementedError

    def get_graph(self):
        raise NotImplementedError
