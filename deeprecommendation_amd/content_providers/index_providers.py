"""Index-returning counterparts of the reference's concrete providers (src/content_providers/*.py).

The reference builds a dense one-hot row per sample with sklearn's LabelBinarizer re-fitted per batch
(src/util.py:5-10, one_hot_provider.py:17-21).  At 1 M users that row is 4 MB per sample; the table path needs only
the COLUMN of the 1 — the rank of the id among the sorted unique ids (one_hot_provider.py:14-15, LabelBinarizer's
``classes_`` order).  These providers return that position (int64) and keep the reference's method names, so
datasets / eval loops written against ``ContentProvider`` keep working.
"""
import numpy as np
import torch

from ..neural_collaborative_filtering.content_providers import ContentProvider, DynamicContentProvider, GraphContentProvider
from ..neural_collaborative_filtering.models.attention_ncf import RowsOf, SparseRatings
from ..neural_collaborative_filtering.models.gnn_ncf import GraphData


_LUT_MIN_QUERY = 4096          # below this many ids a binary search is cheaper than building / touching a table
_LUT_MAX_SLOTS = 1 << 27       # 1 GiB of int64 at most
_luts = {}                     # id(sorted array) -> (array, lo, table): rank of every integer id in [lo, hi], -1 if absent


def _lut_for(sorted_ids: np.ndarray):
    hit = _luts.get(id(sorted_ids))
    if hit is not None and hit[0] is sorted_ids:
        return hit[1], hit[2]
    if len(sorted_ids) == 0 or not np.issubdtype(sorted_ids.dtype, np.integer):
        return None
    lo, hi = int(sorted_ids[0]), int(sorted_ids[-1])
    slots = hi - lo + 1
    if slots > _LUT_MAX_SLOTS or slots > 16 * len(sorted_ids) + 1024:
        return None  # sparse id space: keep the binary search
    table = np.full(slots, -1, dtype=np.int64)
    table[sorted_ids - lo] = np.arange(len(sorted_ids), dtype=np.int64)
    _luts[id(sorted_ids)] = (sorted_ids, lo, table)
    return lo, table


def _positions(sorted_ids: np.ndarray, ids) -> np.ndarray:
    """Rank of each id among the sorted unique ids (LabelBinarizer's ``classes_`` order); KeyError for an unknown id.
    Whole-file queries over a dense integer id space go through a direct table (one gather per id; measured at 8 M
    queries into 1 M ids: 1.98 s with the binary search — 98 % of a device-resident evaluation pass)."""
    ids = np.atleast_1d(np.asarray(ids))
    if len(ids) >= _LUT_MIN_QUERY and np.issubdtype(ids.dtype, np.integer):
        lut = _lut_for(sorted_ids)
        if lut is not None:
            lo, table = lut
            if int(ids.min()) >= lo and int(ids.max()) < lo + len(table):   # else: the search below names the culprits
                pos = table[ids - lo] if lo else table[ids]
                if int(pos.min()) >= 0:
                    return pos
    pos = np.searchsorted(sorted_ids, ids)
    bad = (pos >= len(sorted_ids)) | (sorted_ids[np.minimum(pos, len(sorted_ids) - 1)] != ids)
    if bad.any():
        raise KeyError(f"unknown id(s): {ids[bad][:5].tolist()}")
    return pos.astype(np.int64)


class IndexProvider(ContentProvider):
    """Same id universe as the reference's OneHotProvider (all ids of the full utility matrix, sorted), but
    ``get_*_profile`` returns int64 positions instead of one-hot rows."""

    def __init__(self, user_ids, item_ids):
        self.all_item_ids = np.array(sorted(np.unique(np.asarray(item_ids))))
        self.all_user_ids = np.array(sorted(np.unique(np.asarray(user_ids))))

    @classmethod
    def from_csv(cls, full_matrix_csv):
        import pandas as pd
        m = pd.read_csv(full_matrix_csv)
        return cls(m['userId'].values, m['movieId'].values)

    def get_item_profile(self, itemID):
        return _positions(self.all_item_ids, itemID)

    def get_user_profile(self, userID):
        return _positions(self.all_user_ids, userID)

    def get_num_items(self):
        return len(self.all_item_ids)

    def get_num_users(self):
        return len(self.all_user_ids)

    def get_item_feature_dim(self):
        return self.get_num_items()

    def device_lookup(self, device):
        """id -> position ON THE GPU: ``f(user_ids, item_ids) -> (user_pos, item_pos)`` over int64 device tensors, or
        None when an id space is too sparse for a direct table.  A host core resolves ~5 M random ids per second
        (binary search or table alike: it is cache-miss bound); the same gather is microseconds on the GPU, so whole-file
        evaluation uploads RAW ids.  An unknown id maps to -1, which every HIP kernel reports through the out-of-range
        flag (IndexError from native.check_oob) instead of the host path's KeyError."""
        cache = self.__dict__.setdefault("_device_luts", {})
        key = str(device)
        if key not in cache:
            luts = [_lut_for(self.all_user_ids), _lut_for(self.all_item_ids)]
            cache[key] = None if any(l is None for l in luts) else [(lo, torch.from_numpy(t).to(device)) for lo, t in luts]
        tabs = cache[key]
        if tabs is None:
            return None

        def one(ids, lo, table):
            rel = ids - lo
            pos = table[rel.clamp(0, table.numel() - 1)]
            return torch.where((rel >= 0) & (rel < table.numel()), pos, torch.full_like(pos, -1))

        return lambda user_ids, item_ids: (one(user_ids, *tabs[0]), one(item_ids, *tabs[1]))


class OneHotProvider(IndexProvider):
    """Dense one-hot rows like the reference's OneHotProvider (small scale only), without sklearn."""

    def _onehot(self, pos, n):
        out = np.zeros((len(pos), n), dtype=np.int64)
        out[np.arange(len(pos)), pos] = 1
        return out

    def get_item_profile(self, itemID):
        return self._onehot(_positions(self.all_item_ids, itemID), self.get_num_items())

    def get_user_profile(self, userID):
        return self._onehot(_positions(self.all_user_ids, userID), self.get_num_users())


class IndexGraphProvider(GraphContentProvider):
    """Vectorised counterpart of GraphProvider + create_graph (graph_providers.py:10-118) with one-hot node features
    (OneHotGraphProvider): items are nodes 0..I-1 in sorted id order, user node id = I + rank (:79-80); edge
    user->item weighs ``rating - (mean_user + 2.5)/2`` and item->user ``rating - (mean_item + 2.5)/2`` (:31-47);
    ``binary=True`` keeps only edges with rating >= that neutral value and drops the weights."""

    def __init__(self, all_user_ids, all_item_ids, inter_users, inter_items, inter_ratings, binary=False):
        self.binary = binary
        self.all_items = np.array(sorted(np.unique(np.asarray(all_item_ids))))
        self.all_users = np.array(sorted(np.unique(np.asarray(all_user_ids))))
        I = len(self.all_items)
        u = np.asarray(inter_users)
        it = np.asarray(inter_items)
        r = np.asarray(inter_ratings, dtype=np.float64)
        upos = _positions(self.all_users, u)
        ipos = _positions(self.all_items, it)
        # per-user / per-item mean rating over THESE interactions (groupby().mean(), :16-17)
        usum = np.bincount(upos, weights=r, minlength=len(self.all_users))
        ucnt = np.maximum(np.bincount(upos, minlength=len(self.all_users)), 1)
        isum = np.bincount(ipos, weights=r, minlength=I)
        icnt = np.maximum(np.bincount(ipos, minlength=I), 1)
        user_avg = ((usum / ucnt)[upos] + 2.5) / 2
        item_avg = ((isum / icnt)[ipos] + 2.5) / 2
        unode = upos + I
        k1 = np.ones(len(r), bool) if not binary else r >= user_avg
        k2 = np.ones(len(r), bool) if not binary else r >= item_avg
        self._pos_parts = (unode, ipos, k1, k2)
        self.graph = GraphData(
            item_features=None, user_features=None,
            user2item_edge_index=torch.from_numpy(np.stack([unode[k1], ipos[k1]])),
            item2user_edge_index=torch.from_numpy(np.stack([ipos[k2], unode[k2]])),
            user2item_edge_attr=None if binary else torch.from_numpy((r - user_avg)[k1]).float(),
            item2user_edge_attr=None if binary else torch.from_numpy((r - item_avg)[k2]).float(),
            num_items=I, num_users=len(self.all_users))

    def get_num_items(self):
        return len(self.all_items)

    def get_num_users(self):
        return len(self.all_users)

    def get_item_dim(self):
        return self.get_num_items()

    def get_user_dim(self):
        return self.get_num_users()

    def get_user_nodeID(self, userID):
        pos = _positions(self.all_users, userID) + self.get_num_items()
        return pos if np.ndim(userID) else int(pos[0])

    def get_item_nodeID(self, itemID):
        pos = _positions(self.all_items, itemID)
        return pos if np.ndim(itemID) else int(pos[0])

    def get_graph(self) -> GraphData:
        return self.graph

    def pos_df(self):
        """The reference's ``pos_df`` (graph_providers.py:24,37,46,54-55): MultiIndex (Id1, Id2) -> position of the edge inside its
        direction's edge list, rows in the reference's order (per interaction: the user->item row, then the item->user row).
        Built on demand — the models here mask target edges by key (``PreparedGraph.masked_coef``), not through this frame."""
        import pandas as pd
        unode, ipos, k1, k2 = self._pos_parts
        n = len(unode)
        rows = np.concatenate([np.stack([unode[k1], ipos[k1], np.arange(int(k1.sum()))], 1),
                               np.stack([ipos[k2], unode[k2], np.arange(int(k2.sum()))], 1)])
        order = np.argsort(np.concatenate([2 * np.arange(n)[k1], 2 * np.arange(n)[k2] + 1]), kind="stable")
        return pd.DataFrame(rows[order], columns=["Id1", "Id2", "pos"]).set_index(["Id1", "Id2"], inplace=False)


class SparseDynamicProvider(DynamicContentProvider):
    """Counterpart of DynamicProfilesProvider (dynamic_profiles_provider.py:10-73) over in-memory arrays.

    ``item_ids`` (I,), ``item_features`` (I, F); per user: arrays of rated item ids (sorted by id, the ordering the
    reference relies on, :64-66), ratings and the user's mean rating.  ``collate_interacted_items`` emits the same
    6-tuple; with ``sparse=True`` the user matrix is a SparseRatings (CSR) instead of the dense (B, I) float matrix.
    Values are ``rating - (mean + 2.5)/2`` (:66); entries equal to 0.0 are dropped exactly as the reference's
    ``user_matrix != 0`` does (attention_ncf.py:158).
    """

    def __init__(self, item_ids, item_features, user_ids, user_rated_items, user_ratings, user_mean_ratings, sparse=True):
        order = np.argsort(np.asarray(item_ids))
        self.item_ids = np.asarray(item_ids)[order]
        self.features = torch.as_tensor(np.asarray(item_features)[order], dtype=torch.float32)
        self.user_pos = {u: k for k, u in enumerate(np.asarray(user_ids).tolist())}
        self.user_rated_items = [np.asarray(x) for x in user_rated_items]
        self.user_ratings = [np.asarray(x, dtype=np.float64) for x in user_ratings]
        self.user_mean = np.asarray(user_mean_ratings, dtype=np.float64)
        self.sparse = sparse

    def get_item_profile(self, itemID):
        return self.features[torch.from_numpy(_positions(self.item_ids, itemID))]

    def get_num_items(self):
        return len(self.item_ids)

    def get_num_users(self):
        return len(self.user_pos)

    def get_item_feature_dim(self):
        return self.features.shape[1]

    def device_state(self, device):
        """Everything the collate reads, resident on ``device`` (cached): see _DynamicDeviceState.  None when an id space
        is not a dense integer range."""
        cache = self.__dict__.setdefault("_device_states", {})
        key = str(device)
        if key not in cache:
            try:
                cache[key] = _DynamicDeviceState(self, device)
            except ValueError:
                cache[key] = None
        return cache[key]

    def collate_interacted_items(self, batch, for_ranking: bool, ignore_ratings=False):
        users, cand, third = zip(*batch)
        cand_ids = np.asarray(cand)
        candidate_items = self.get_item_profile(cand_ids)
        targets_or_items2 = self.get_item_profile(np.asarray(third)) if for_ranking else torch.as_tensor(np.asarray(third), dtype=torch.float32)
        ups_all = np.asarray([self.user_pos[u] for u in users])
        # one CSR row per DISTINCT user of the batch (the dense matrix of dynamic_profiles_provider.py:60-70 repeats a
        # user's row for each of their samples); pair_row maps samples to rows
        ups, pair_row = np.unique(ups_all, return_inverse=True)
        rated_ids = np.unique(np.concatenate([self.user_rated_items[p] for p in ups]))  # sorted (:59)
        cols, vals, counts = [], [], []
        for p in ups:
            c = np.searchsorted(rated_ids, self.user_rated_items[p])
            v = np.ones(len(c)) if ignore_ratings else self.user_ratings[p] - (self.user_mean[p] + 2.5) / 2
            v = v.astype(np.float32)
            keep = v != 0
            cols.append(c[keep]); vals.append(v[keep]); counts.append(int(keep.sum()))
        rowptr = torch.zeros(len(ups) + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(torch.as_tensor(counts), 0)
        ratings = SparseRatings(rowptr, torch.as_tensor(np.concatenate(cols), dtype=torch.int32),
                                torch.as_tensor(np.concatenate(vals), dtype=torch.float32), len(rated_ids),
                                pair_row=torch.as_tensor(pair_row, dtype=torch.int64))
        user_matrix = ratings if self.sparse else ratings.to_dense(ratings.val)
        return cand_ids, rated_ids, candidate_items, self.get_item_profile(rated_ids), user_matrix, targets_or_items2


class _DynamicDeviceState:
    """SparseDynamicProvider on the GPU: the (I, F) feature table, id -> position tables, and EVERY user's rated set as
    one CSR (row = user position, col = position in the whole sorted catalogue, val = rating - (mean + 2.5)/2, zeros
    dropped — the rows ``collate_interacted_items`` builds, against the full catalogue instead of the batch's union of
    rated items: columns of items nobody in the batch rated are masked out by the reference anyway, attention_ncf.py:158).
    ``batch`` assembles the collate's 6-tuple from raw id tensors with two gathers; nothing is rebuilt per batch, and
    since ``rated_items`` is the same tensor every time the model's catalogue projections are computed once."""

    def __init__(self, prov, device):
        self.device = torch.device(device)
        uid = np.asarray(list(prov.user_pos.keys()))
        upos = np.asarray(list(prov.user_pos.values()), dtype=np.int64)
        if not (np.issubdtype(uid.dtype, np.integer) and np.issubdtype(prov.item_ids.dtype, np.integer)) or len(uid) == 0:
            raise ValueError("non-integer ids")
        ilut = _lut_for(prov.item_ids)
        order = np.argsort(uid)
        self._sorted_uid = uid[order]      # kept alive: _lut_for caches by object identity
        ulut = _lut_for(self._sorted_uid)
        if ilut is None or ulut is None:
            raise ValueError("sparse id space")
        utable = np.where(ulut[1] >= 0, upos[order][np.maximum(ulut[1], 0)], -1)   # rank among sorted ids -> provider position
        self.user_lo, self.user_table = ulut[0], torch.from_numpy(utable).to(self.device)
        self.item_lo, self.item_table = ilut[0], torch.from_numpy(ilut[1]).to(self.device)
        self.features = prov.features.to(self.device).contiguous()
        n_users = len(prov.user_rated_items)
        counts = np.zeros(n_users, dtype=np.int64)
        cols, vals = [], []
        for p in range(n_users):
            c = _positions(prov.item_ids, prov.user_rated_items[p]) if len(prov.user_rated_items[p]) else np.zeros(0, np.int64)
            v = (prov.user_ratings[p] - (prov.user_mean[p] + 2.5) / 2).astype(np.float32)
            keep = v != 0
            cols.append(c[keep]); vals.append(v[keep]); counts[p] = int(keep.sum())
        rowptr = np.zeros(n_users + 1, dtype=np.int64)
        np.cumsum(counts, out=rowptr[1:])
        self.rowptr = torch.from_numpy(rowptr).to(self.device)
        self.col = torch.from_numpy(np.concatenate(cols).astype(np.int32) if cols else np.zeros(0, np.int32)).to(self.device)
        self.val = torch.from_numpy(np.concatenate(vals) if vals else np.zeros(0, np.float32)).to(self.device)
        self.num_items = len(prov.item_ids)

    def _lookup(self, ids, lo, table):
        rel = ids - lo
        inside = (rel >= 0) & (rel < table.numel())
        pos = torch.where(inside, table[rel.clamp(0, table.numel() - 1)], torch.full_like(rel, -1))
        bad = pos < 0
        if ids.is_cuda:
            from .. import native
            native._oob_flag(ids.device).add_(bad.any().to(torch.int32))   # sticky: IndexError at the next check_oob, no sync here
        elif bool(bad.any()):
            raise KeyError(f"unknown id(s): {ids[bad][:5].tolist()}")
        return pos.clamp_min(0)

    COMPACT_MIN_USERS = 4096   # user bases above this get a chunk-local CSR (the grouping pass is O(rows of the CSR))

    def positions(self, users, cands):
        """Raw id tensors -> (CSR row of each sample, catalogue position); done once per uploaded CHUNK of batches.

        With a large user base the chunk's distinct users are compacted into a chunk-local CSR (two size read-backs per
        chunk, none per batch): the pair-grouping pass and the grouped kernel's grid bound scale with the rows of the CSR
        they are handed, and a batch touches a few dozen of possibly millions of users."""
        upos, cpos = self._lookup(users, self.user_lo, self.user_table), self._lookup(cands, self.item_lo, self.item_table)
        if self.rowptr.numel() - 1 <= self.COMPACT_MIN_USERS:
            self._chunk_csr = (self.rowptr, self.col, self.val)
            return upos, cpos
        uniq, local_row = torch.unique(upos, return_inverse=True)
        start = self.rowptr[:-1][uniq]
        counts = self.rowptr[1:][uniq] - start
        rowptr = torch.zeros(uniq.numel() + 1, dtype=torch.int64, device=upos.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        owner = torch.repeat_interleave(torch.arange(uniq.numel(), device=upos.device), counts)
        src = start[owner] + (torch.arange(owner.numel(), device=upos.device) - rowptr[:-1][owner])
        self._chunk_csr = (rowptr, self.col[src], self.val[src])
        return local_row, cpos

    def batch_at(self, rows, cpos, y, pairs_per_row_hint=None):
        """The collate's 6-tuple for one batch of the chunk last passed to ``positions`` (cand_ids carries catalogue
        positions, rated_ids is None: the rated list is the whole catalogue in provider order; the candidate rows stay an
        unevaluated selection of it)."""
        rowptr, col, val = self._chunk_csr
        ratings = SparseRatings(rowptr, col, val, self.num_items, pair_row=rows, pairs_per_row_hint=pairs_per_row_hint)
        return cpos, None, RowsOf(self.features, cpos), self.features, ratings, y

    def batch(self, users, cands, y, pairs_per_row_hint=None):
        return self.batch_at(*self.positions(users, cands), y, pairs_per_row_hint)
