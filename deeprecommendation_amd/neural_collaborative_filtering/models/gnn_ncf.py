"""GraphNCF / LightGCNConv — drop-in for reference models/gnn_ncf.py without PyG.

Scoring (eval, no-grad) runs on the HIP kernels:
  * the per-EDGE Linear of the reference's message() (gnn_ncf.py:91-93) is hoisted to a per-NODE Linear
    (W(x)[src] == W(x[src])) — one MFMA GEMM over N rows instead of E rows;
  * propagate(aggr='add') becomes a CSR-by-destination SpMM with a wavefront segmented reduction
    (ncf_spmm_csr); the edge coefficient w_e * dis[src] * dis[dst] (gnn_ncf.py:47-50,54,58,91) is computed once
    per graph (ncf_degree_accumulate + ncf_edge_coef);
  * the layer mean (gnn_ncf.py:351) is a running sum fused into the SpMM epilogue + one division;
  * the propagated node table does not depend on the batch in eval mode, so it is cached per (graph, weights)
    instead of being recomputed on every forward as the reference does (gnn_ncf.py:298-351);
  * readout = fused gather(item) ‖ gather(user) -> MLP (gnn_ncf.py:354-362) or gather-dot (:365).
Node numbering follows the reference: items first, user node id = num_items + rank (graph_providers.py:79-80).

Training (module.training / autograd recording) keeps to differentiable torch ops, including the reference's
train-only edge masking, node dropout and message dropout (gnn_ncf.py:246-296,314-333,369-378).
LightGATConv (a SURVEY §8(f) "next" row) scores on the HIP path too: edge softmax kernel + the same SpMM.
"""
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from ... import native
from ..util import build_MLP_layers, require_gpu, use_native
from .base import GNN_NCF
from .basic_ncf import _ScoringMixin

SEGMENT_EDGES = 512  # destination rows longer than this are split into segments summed in order (load balance)


class GraphData:
    """Minimal stand-in for the PyG ``Data`` object the reference builds in graph_providers.py:58-66.

    Attributes (same names): item_features, user_features, user2item_edge_index (2,E) int64 [src; dst],
    item2user_edge_index, user2item_edge_attr / item2user_edge_attr ((E,) float or None), pos_df (unused here).
    ``item_features`` / ``user_features`` may be ``None`` = one-hot identity features (OneHotGraphProvider's
    torch.eye, graph_providers.py:123-127, which cannot be materialised at 1 M users); then
    ``num_items`` / ``num_users`` give the node counts.
    """

    def __init__(self, item_features=None, user_features=None, user2item_edge_index=None, item2user_edge_index=None,
                 user2item_edge_attr=None, item2user_edge_attr=None, pos_df=None, num_items=None, num_users=None):
        self.item_features = item_features
        self.user_features = user_features
        self.user2item_edge_index = user2item_edge_index
        self.item2user_edge_index = item2user_edge_index
        self.user2item_edge_attr = user2item_edge_attr
        self.item2user_edge_attr = item2user_edge_attr
        self.pos_df = pos_df
        self.num_items = int(num_items if num_items is not None else item_features.shape[0])
        self.num_users = int(num_users if num_users is not None else user_features.shape[0])
        self._device_copies = {}
        self._prepared = {}

    _TENSORS = ("item_features", "user_features", "user2item_edge_index", "item2user_edge_index",
                "user2item_edge_attr", "item2user_edge_attr")

    @property
    def device(self):
        return self.user2item_edge_index.device

    def to(self, device):
        """Cached per device: the reference calls graph.to(device) on every forward (gnn_datasets.py:28)."""
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if self.device == device:
            return self
        g = self._device_copies.get(device)
        if g is None:
            kw = {k: (None if getattr(self, k) is None else getattr(self, k).to(device)) for k in self._TENSORS}
            g = GraphData(pos_df=self.pos_df, num_items=self.num_items, num_users=self.num_users, **kw)
            self._device_copies[device] = g
        return g


class PreparedGraph:
    """Per-(graph, hetero) device structures for the SpMM: combined CSR by destination split into segments."""

    def __init__(self, graph: GraphData, hetero: bool, seg_len: int = SEGMENT_EDGES):
        u2i, i2u = graph.user2item_edge_index, graph.item2user_edge_index
        require_gpu(u2i, i2u)
        N = graph.num_items + graph.num_users
        dev = u2i.device
        self.N, self.hetero = N, hetero
        # degree over cat(u2i, i2u) destinations (:41,:48).  The CSR build needs the per-row counts anyway (bincount);
        # the float-atomic ncf_degree_accumulate kernel gives the same numbers but serialises on hub items
        # (69 ms for 50 M Zipf edges vs 2 ms for uniform ones), so the prepared graph reuses the counts.
        counts = torch.bincount(torch.cat([u2i[1], i2u[1]]), minlength=N)
        if counts.numel() != N:
            raise IndexError(f"edge destination id out of range for a graph of {N} nodes")
        # sources are range-checked here, once per graph (the SpMM kernels skip a bad source silently; PyG's
        # x[edge_index[0]] raises IndexError in the reference, gnn_ncf.py:74-94)
        for ei in (u2i, i2u):
            if ei.shape[1] and (int(ei[0].min()) < 0 or int(ei[0].max()) >= N):
                raise IndexError(f"edge source id out of range for a graph of {N} nodes")
        deg = counts.to(torch.float32)
        self.deg = deg
        c1 = native.edge_coef(u2i[0].contiguous(), u2i[1].contiguous(), graph.user2item_edge_attr, deg)
        c2 = native.edge_coef(i2u[0].contiguous(), i2u[1].contiguous(), graph.item2user_edge_attr, deg)
        # Which hoisted table does an edge read?  hetero: u2i edges read W_u2i(x[src]), i2u edges W_i2u(x[src]).
        # On a bipartite graph in reference numbering every u2i source is a user node (>= num_items) and every i2u
        # source an item node, so ONE (N, D) table with rows [0, I) = W_i2u(x) and rows [I, N) = W_u2i(x) serves
        # both; otherwise the two tables are stacked (2N rows) and i2u sources are offset by N.
        self.split = None
        src2 = i2u[0]
        self.z_rows = N
        if hetero:
            I = graph.num_items
            bip = (u2i.shape[1] == 0 or int(u2i[0].min()) >= I) and (i2u.shape[1] == 0 or int(i2u[0].max()) < I)
            if bip:
                self.split = I
            else:
                src2 = i2u[0] + N
                self.z_rows = 2 * N
        src = torch.cat([u2i[0], src2])
        dst = torch.cat([u2i[1], i2u[1]])
        coef = torch.cat([c1, c2])
        order = torch.argsort(dst, stable=True)  # keeps the reference's edge order inside each destination
        self.col = src[order].to(torch.int32).contiguous()
        self.coef = coef[order].contiguous()
        # raw edge weights in CSR order (LightGAT uses weight * attention, no degree normalisation, :173-176)
        a1, a2 = graph.user2item_edge_attr, graph.item2user_edge_attr
        self.attr = None if (a1 is None or a2 is None) else torch.cat([a1, a2])[order].float().contiguous()
        # per-destination softmax groups must not mix edge types: true for non-hetero and for bipartite hetero graphs
        self.type_pure = (not hetero) or self.split is not None
        rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dev)
        rowptr[1:] = torch.cumsum(counts, 0)
        self.rowptr = rowptr
        # rows longer than seg_len are split into segments; hub rows are finished by an ordered log-depth tree
        self.csr = native.SegmentedCSR(rowptr, self.col, self.coef, seg_len=seg_len)
        self.segptr, self.row_of = self.csr.levels[0][0], self.csr.levels[0][1]
        self.counts = counts
        self._seg_len = seg_len
        self._transposed = None
        self._train = None
        self._edges = (u2i, i2u)          # references only: the training step derives its per-edge keys on first use
        native.check_oob(dev)

    def softmax_segments(self):
        """(segptr, row_of, seg_first) for the segmented edge softmax (LightGAT): seg_first[r] = first segment of destination r.  None
        when no row was split (segments == rows)."""
        if self.row_of is None:
            return None
        if getattr(self, "_seg_first", None) is None:
            self._seg_first = torch.searchsorted(self.row_of.to(torch.int64), torch.arange(self.N + 1, device=self.row_of.device)).contiguous()
        return self.segptr, self.row_of, self._seg_first

    def transposed(self):
        """(CSR by SOURCE, edge ids): entry k of the transpose is entry eid[k] of the CSR by destination — the backward of the
        aggregation, dz[src] += coef_e * dy[dst], is the same segmented SpMM on it.  Built on first use (training only)."""
        if self._transposed is None:
            dst_of = torch.repeat_interleave(torch.arange(self.N, device=self.col.device), self.counts)
            order = torch.argsort(self.col, stable=True)
            counts_t = torch.bincount(self.col.long(), minlength=self.z_rows)
            rowptr_t = torch.zeros(self.z_rows + 1, dtype=torch.int64, device=self.col.device)
            rowptr_t[1:] = torch.cumsum(counts_t, 0)
            eid = order.to(torch.int32).contiguous()
            csr_t = native.SegmentedCSR(rowptr_t, dst_of[order].to(torch.int32).contiguous(), None, seg_len=self._seg_len)
            self._transposed = (csr_t, eid)
        return self._transposed

    def train_state(self):
        """Per-graph tensors of the training step, built on first use: destination and source of every CSR entry and its
        (user, item) key for the target-edge masking (gnn_ncf.py:314-320, 369-378)."""
        if self._train is None:
            u2i, i2u = self._edges
            order = torch.argsort(torch.cat([u2i[1], i2u[1]]), stable=True)     # the CSR's own order (see __init__)
            pair_key = torch.cat([u2i[0] * self.N + u2i[1], i2u[1] * self.N + i2u[0]])[order]
            dst_of = torch.repeat_interleave(torch.arange(self.N, device=self.col.device), self.counts)
            self._train = (dst_of, self.col.long(), pair_key)
        return self._train

    def masked_coef(self, user_ids, item_ids):
        """Coefficients of one training batch with the batch's target edges removed in both directions (gnn_ncf.py:314-320):
        the degrees are those of the REMAINING edges (the reference recomputes them from the masked edge lists, :47-50), a
        removed edge gets coefficient 0 — the CSR itself stays as it is."""
        dst_of, src, pair_key = self.train_state()
        masked = torch.isin(pair_key, user_ids.long() * self.N + item_ids.long())
        deg = (self.counts - torch.bincount(dst_of[masked], minlength=self.N)).to(torch.float32)
        dis = deg.pow(-0.5)
        dis[dis == float('inf')] = 0
        norm = dis[src] * dis[dst_of]
        coef = norm if self.attr is None else self.attr * norm
        return torch.where(masked, torch.zeros_like(coef), coef).contiguous()


class _ConvBase(nn.Module):
    def __init__(self, in_channels, out_channels, hetero, dropout, attention: bool):
        super().__init__()
        self.hetero = hetero

        def lin():
            seq = nn.Sequential(nn.Linear(in_channels, out_channels), nn.Dropout(dropout))
            nn.init.xavier_uniform_(seq[0].weight)
            return seq

        if hetero:
            self.user2item_W, self.item2user_W = lin(), lin()
            if attention:
                self.user2item_AttNet = nn.Sequential(nn.Linear(in_channels * 2, 1))
                self.item2user_AttNet = nn.Sequential(nn.Linear(in_channels * 2, 1))
        else:
            self.W = lin()
            if attention:
                self.AttNet = nn.Sequential(nn.Linear(in_channels * 2, 1))


def _scatter_add(ei, messages, N):
    return torch.zeros((N, messages.shape[1]), dtype=messages.dtype, device=messages.device).index_add_(0, ei[1], messages)


class LightGCNConv(_ConvBase):
    """reference gnn_ncf.py:13-94.  Parameter names (user2item_W.0.*, item2user_W.0.*, W.0.*) are kept."""

    def __init__(self, in_channels, out_channels, hetero, dropout=0.1, **kwargs):
        super().__init__(in_channels, out_channels, hetero, dropout, attention=False)

    # ---- HIP scoring path -------------------------------------------------------------------------------
    def hoisted(self, x: torch.Tensor, prep: PreparedGraph) -> torch.Tensor:
        """Z such that message(e) = coef_e * Z[col_e]: the per-edge Linear applied once per node."""
        if not self.hetero:
            l = self.W[0]
            return native.linear(x, l.weight.detach(), l.bias.detach())
        lu, li = self.user2item_W[0], self.item2user_W[0]
        N, D = x.shape[0], lu.out_features
        z = torch.empty((prep.z_rows, D), dtype=torch.float32, device=x.device)
        if prep.split is not None:
            I = prep.split
            if I > 0:
                native.linear(x[:I], li.weight.detach(), li.bias.detach(), out=z[:I])   # item sources -> item2user_W
            if N > I:
                native.linear(x[I:], lu.weight.detach(), lu.bias.detach(), out=z[I:])   # user sources -> user2item_W
        else:
            native.linear(x, lu.weight.detach(), lu.bias.detach(), out=z[:N])
            native.linear(x, li.weight.detach(), li.bias.detach(), out=z[N:])
        return z

    def propagate_native(self, x, prep: PreparedGraph, acc_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
        return prep.csr.spmm(self.hoisted(x, prep), acc_sum=acc_sum)

    # ---- torch path (training; same maths as the reference, scatter via index_add_) -----------------------
    def forward(self, x, user2item_edge_index, item2user_edge_index, user2item_edge_attr=None, item2user_edge_attr=None):
        N = x.size(0)
        to_ = torch.cat([user2item_edge_index[1], item2user_edge_index[1]])
        deg = torch.zeros(N, dtype=x.dtype, device=x.device).scatter_add_(0, to_, torch.ones(to_.numel(), dtype=x.dtype, device=x.device))
        dis = deg.pow(-0.5)
        dis[dis == float('inf')] = 0

        def msgs(ei, attr, W):
            norm = dis[ei[0]] * dis[ei[1]]
            wx = W(x[ei[0]])
            return (attr.view(-1, 1) * norm.view(-1, 1) * wx) if attr is not None else norm.view(-1, 1) * wx

        if self.hetero:
            return (_scatter_add(user2item_edge_index, msgs(user2item_edge_index, user2item_edge_attr, self.user2item_W), N)
                    + _scatter_add(item2user_edge_index, msgs(item2user_edge_index, item2user_edge_attr, self.item2user_W), N))
        ei = torch.cat([user2item_edge_index, item2user_edge_index], dim=1)
        attr = None
        if user2item_edge_attr is not None and item2user_edge_attr is not None:
            attr = torch.cat([user2item_edge_attr, item2user_edge_attr])
        return _scatter_add(ei, msgs(ei, attr, self.W), N)


def _segment_softmax(scores, index, N):
    """PyG softmax(src, index): exp(src - max_group) / (sum_group + 1e-16)."""
    mx = torch.full((N, scores.shape[1]), -float('inf'), dtype=scores.dtype, device=scores.device)
    mx = mx.scatter_reduce(0, index.view(-1, 1).expand_as(scores), scores, reduce='amax', include_self=True)
    ex = torch.exp(scores - mx[index])
    den = torch.zeros((N, scores.shape[1]), dtype=scores.dtype, device=scores.device).index_add_(0, index, ex)
    return ex / (den[index] + 1e-16)


class LightGATConv(_ConvBase):
    """reference gnn_ncf.py:97-177.  HIP scoring path: the per-edge Linear is hoisted per node like LightGCN; the
    attention score AttNet(cat(x_j, x_i)) = w_j·x_j + (w_i·x_i + b) only needs its SOURCE half inside the per-destination
    softmax (the rest is constant in a group), so one GEMV gives s[n] = w_j·x[n], ncf_edge_softmax_csr turns it into the
    per-edge coefficient weight·alpha, and the aggregation is the same segmented SpMM as LightGCN."""

    def __init__(self, in_channels, out_channels, hetero, dropout=0.1, **kwargs):
        super().__init__(in_channels, out_channels, hetero, dropout, attention=True)

    hoisted = LightGCNConv.hoisted

    def propagate_native(self, x, prep: "PreparedGraph", acc_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not prep.type_pure:
            raise NotImplementedError("LightGAT on a hetero graph whose destinations mix edge types runs on the torch path only")
        D = x.shape[1]
        s = torch.empty((x.shape[0], 1), dtype=torch.float32, device=x.device)
        if not self.hetero:
            native.linear(x, self.AttNet[0].weight.detach()[:, :D].contiguous(), None, out=s)
        else:
            I = prep.split  # items are the sources of item->user edges, users of user->item edges
            if I > 0:
                native.linear(x[:I], self.item2user_AttNet[0].weight.detach()[:, :D].contiguous(), None, out=s[:I])
            if x.shape[0] > I:
                native.linear(x[I:], self.user2item_AttNet[0].weight.detach()[:, :D].contiguous(), None, out=s[I:])
        coef = native.edge_softmax_csr(prep.rowptr, prep.col, prep.attr, s.view(-1), segments=prep.softmax_segments())
        return prep.csr.spmm(self.hoisted(x, prep), acc_sum=acc_sum, coef=coef)

    def forward(self, x, user2item_edge_index, item2user_edge_index, user2item_edge_attr=None, item2user_edge_attr=None):
        N = x.size(0)

        def msgs(ei, attr, W, att):
            x_j, x_i = x[ei[0]], x[ei[1]]
            a = _segment_softmax(att(torch.cat([x_j, x_i], dim=1)), ei[1], N)
            m = a * W(x_j)
            return attr.view(-1, 1) * m if attr is not None else m

        if self.hetero:
            return (_scatter_add(user2item_edge_index, msgs(user2item_edge_index, user2item_edge_attr, self.user2item_W, self.user2item_AttNet), N)
                    + _scatter_add(item2user_edge_index, msgs(item2user_edge_index, item2user_edge_attr, self.item2user_W, self.item2user_AttNet), N))
        ei = torch.cat([user2item_edge_index, item2user_edge_index], dim=1)
        attr = None
        if user2item_edge_attr is not None and item2user_edge_attr is not None:
            attr = torch.cat([user2item_edge_attr, item2user_edge_attr])
        return _scatter_add(ei, msgs(ei, attr, self.W, self.AttNet), N)


class GraphNCF(_ScoringMixin, GNN_NCF):
    compatible_datasets = ("GraphPointwiseDataset", "GraphRankingDataset")

    def __init__(self, item_dim, user_dim, num_gnn_layers: int, hetero, node_emb=64, mlp_dense_layers=None,
                 dropout_rate=0.2, use_dot_product=False, concat=False, message_dropout=None, node_dropout=None,
                 convType='LightGCN'):
        super().__init__()
        if mlp_dense_layers is None:
            mlp_dense_layers = [256, 128]
        self.kwargs = {'item_dim': item_dim, 'user_dim': user_dim, 'node_emb': node_emb, 'num_gnn_layers': num_gnn_layers,
                       'mlp_dense_layers': mlp_dense_layers, 'use_dot_product': use_dot_product, 'dropout_rate': dropout_rate,
                       'message_dropout': message_dropout, 'node_dropout': node_dropout, 'hetero': hetero, 'concat': concat,
                       'convType': convType}
        self.concat = concat
        self.hetero = hetero
        self.message_dropout = message_dropout
        self.node_dropout = node_dropout
        self.item_embeddings = nn.Sequential(nn.Linear(item_dim, node_emb))
        self.user_embeddings = nn.Sequential(nn.Linear(user_dim, node_emb))
        self.convType = convType
        if convType == 'LightGCN':
            conv = LightGCNConv(node_emb, node_emb, dropout=dropout_rate / 2, hetero=hetero)
        elif convType == 'LightGAT':
            conv = LightGATConv(node_emb, node_emb, dropout=dropout_rate / 2, hetero=hetero)
        else:
            raise ValueError('Invalid convType.')
        self.gnn_convs = nn.ModuleList([conv for _ in range(num_gnn_layers)])  # shared weights (gnn_ncf.py:227)
        if use_dot_product:
            self.MLP = None
        else:
            self.MLP = build_MLP_layers(node_emb * (num_gnn_layers + 1) * 2 if concat else node_emb * 2,
                                        mlp_dense_layers, dropout_rate=dropout_rate)

    def get_model_parameters(self) -> dict:
        return self.kwargs

    def important_hypeparams(self) -> str:
        return '_' + self.convType

    # ------------------------------------------------------------------------------------------ HIP scoring
    def _node_table0(self, graph: GraphData) -> torch.Tensor:
        """graph_emb of gnn_ncf.py:300-304: items first, then users."""
        I, U = graph.num_items, graph.num_users
        D = self.item_embeddings[0].out_features
        dev = graph.device
        x0 = torch.empty((I + U, D), dtype=torch.float32, device=dev)
        for feats, lin, lo, hi in ((graph.item_features, self.item_embeddings[0], 0, I),
                                   (graph.user_features, self.user_embeddings[0], I, I + U)):
            if feats is None:  # one-hot identity features: Linear(eye)[i] = W[:, i] + b
                if lin.in_features != hi - lo:
                    raise ValueError("one-hot features need in_features == number of nodes")
                x0[lo:hi] = lin.weight.detach().t() + lin.bias.detach()
            elif hi > lo:
                native.linear(feats.float().contiguous(), lin.weight.detach(), lin.bias.detach(), out=x0[lo:hi])
        return x0

    def propagate_all(self, graph: GraphData, cache=None) -> torch.Tensor:
        """combined_graph_emb of gnn_ncf.py:336-351 for the whole graph (cached per graph + weights)."""
        cache = self._refresh() if cache is None else cache
        key = ("combined", id(graph))
        if key in cache:
            return cache[key][1]
        pk = ("prep", self.hetero)
        if pk not in graph._prepared:
            graph._prepared[pk] = PreparedGraph(graph, self.hetero)
        prep = graph._prepared[pk]
        conv = self.gnn_convs[0]
        L = len(self.gnn_convs)
        x = self._node_table0(graph)
        if self.concat:
            hs = [x]
            for _ in range(L):
                x = conv.propagate_native(x, prep)
                hs.append(x)
            combined = torch.cat(hs, dim=1).contiguous()  # gnn_ncf.py:349
        else:
            acc = x.clone()
            for _ in range(L):
                x = conv.propagate_native(x, prep, acc_sum=acc)
            combined = native.scale_rows(acc, float(L + 1))  # gnn_ncf.py:351
        cache[key] = (graph, combined)  # keep the graph alive so id() stays unique
        return combined

    def forward(self, graph, userIds, itemIds, device=None, mask_targets=True):
        if not use_native(self):
            return self._forward_train(graph, userIds, itemIds, device, mask_targets)
        require_gpu(userIds, itemIds)
        graph = graph.to(userIds.device)
        cache = self._refresh()  # one parameter fingerprint per forward
        combined = self.propagate_all(graph, cache)
        userIds, itemIds = userIds.long().contiguous(), itemIds.long().contiguous()
        if self.MLP is not None:
            return self._score(combined, itemIds, combined, userIds, cache=cache)  # cat(item, user): gnn_ncf.py:361
        return native.gather_dot(combined, userIds, combined, itemIds)  # gnn_ncf.py:365

    # ------------------------------------------------------------------------------------------ torch training path
    def _features(self, feats, lin):
        if feats is None:
            return lin.weight.t() + lin.bias
        return lin(feats)

    @staticmethod
    def _drop_edges(keep, ei, attr):
        return ei[:, keep], (attr[keep] if attr is not None else None)

    def _hip_training_possible(self, graph, userIds) -> bool:
        """The training step runs on the HIP autograd blocks for LightGCN layers on CUDA tensors; node / message dropout (edge
        sets redrawn per batch on the host, gnn_ncf.py:246-296) and LightGAT keep to the torch ops, as do hetero graphs whose
        sources are not split items / users (two stacked hoisted tables)."""
        return (userIds.is_cuda and not getattr(self, "train_with_torch_ops", False) and self.convType == 'LightGCN'
                and not (self.training and ((self.node_dropout or 0.0) > 0.0 or (self.message_dropout or 0.0) > 0.0)))

    def _forward_train_hip(self, graph, userIds, itemIds, mask_targets):
        """gnn_ncf.py:298-367 with autograd recording, on the HIP blocks: per-node hoisted Linear (LinearFn), aggregation
        (SpmmFn, with the per-edge message dropout of gnn_ncf.py:22-31 regenerated inside the kernel), row gather + MLP."""
        from ...autograd import GatherConcatFn, LinearFn, SpmmFn, mlp_train
        dev = userIds.device
        graph = graph.to(dev)
        pk = ("prep", self.hetero)
        if pk not in graph._prepared:
            graph._prepared[pk] = PreparedGraph(graph, self.hetero)
        prep = graph._prepared[pk]
        if prep.z_rows != prep.N:
            return None                                    # stacked hoisted tables: torch path
        I = graph.num_items

        def feats(f, lin):
            if f is None:                                  # one-hot identity features: Linear(eye) = W^T + b
                return lin.weight.t() + lin.bias
            return LinearFn.apply(f.float(), lin.weight, lin.bias, False)

        x = torch.vstack([feats(graph.item_features, self.item_embeddings[0]), feats(graph.user_features, self.user_embeddings[0])])
        userIds, itemIds = userIds.long().contiguous(), itemIds.long().contiguous()
        coef = prep.masked_coef(userIds, itemIds) if (self.training and mask_targets) else prep.coef
        conv = self.gnn_convs[0]
        p = float((conv.W if not conv.hetero else conv.user2item_W)[1].p) if self.training else 0.0
        seed0 = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0 else 0     # host generator: no device sync
        hs = [x]
        for layer in range(len(self.gnn_convs)):
            if not conv.hetero:
                z = LinearFn.apply(x, conv.W[0].weight, conv.W[0].bias, False)
            else:                                          # rows [0, I) are item sources (item2user_W), the rest user sources
                z = torch.cat((LinearFn.apply(x[:I], conv.item2user_W[0].weight, conv.item2user_W[0].bias, False),
                               LinearFn.apply(x[I:], conv.user2item_W[0].weight, conv.user2item_W[0].bias, False)), dim=0)
            x = SpmmFn.apply(z, prep, coef, (p, seed0 + 7919 * layer) if p > 0 else None)
            hs.append(x)
        combined = torch.cat(hs, dim=1) if self.concat else torch.mean(torch.stack(hs, dim=0), dim=0)
        if self.MLP is not None:
            return mlp_train(self.MLP, GatherConcatFn.apply(combined, itemIds, combined, userIds))   # cat(item, user): :361
        item_emb, user_emb = combined[itemIds], combined[userIds]
        return torch.bmm(user_emb.unsqueeze(1), item_emb.unsqueeze(2)).view(-1, 1)

    def _forward_train(self, graph, userIds, itemIds, device, mask_targets):
        if self._hip_training_possible(graph, userIds):
            out = self._forward_train_hip(graph, userIds, itemIds, mask_targets)
            if out is not None:
                return out
        dev = userIds.device
        graph = graph.to(dev)
        x = torch.vstack([self._features(graph.item_features, self.item_embeddings[0]),
                          self._features(graph.user_features, self.user_embeddings[0])])
        N = x.shape[0]
        u2i, i2u = graph.user2item_edge_index, graph.item2user_edge_index
        a1, a2 = graph.user2item_edge_attr, graph.item2user_edge_attr
        if self.training and mask_targets:
            # drop the edges that are targets of this batch, both directions (gnn_ncf.py:314-320, 369-378)
            key = userIds.long() * N + itemIds.long()
            u2i, a1 = self._drop_edges(~torch.isin(u2i[0] * N + u2i[1], key), u2i, a1)
            i2u, a2 = self._drop_edges(~torch.isin(i2u[1] * N + i2u[0], key), i2u, a2)
        if self.training and self.node_dropout is not None and self.node_dropout > 0.0:
            # keep (1-p) of the nodes outside the batch plus every batch node; keep edges with both ends kept (:281-296)
            batch_nodes = torch.unique(torch.cat((itemIds, userIds)))
            others = np.setdiff1d(np.arange(N), batch_nodes.cpu().numpy())
            kept = np.random.choice(others, size=int((1.0 - self.node_dropout) * len(others)), replace=False)
            keep_node = torch.zeros(N, dtype=torch.bool, device=dev)
            keep_node[torch.as_tensor(kept, device=dev)] = True
            keep_node[batch_nodes] = True
            u2i, a1 = self._drop_edges(keep_node[u2i[0]] & keep_node[u2i[1]], u2i, a1)
            i2u, a2 = self._drop_edges(keep_node[i2u[0]] & keep_node[i2u[1]], i2u, a2)
        if self.training and self.message_dropout is not None and self.message_dropout > 0.0:
            if u2i.shape[1] == i2u.shape[1] and a1 is not None and a2 is not None:  # symmetric: one mask (:254-264)
                keep = (F.dropout(torch.ones(u2i.shape[1]), self.message_dropout, True) > 0).to(dev)
                u2i, a1 = self._drop_edges(keep, u2i, a1)
                i2u, a2 = self._drop_edges(keep, i2u, a2)
            else:
                k1 = (F.dropout(torch.ones(u2i.shape[1]), self.message_dropout, True) > 0).to(dev)
                u2i, a1 = self._drop_edges(k1, u2i, a1)
                k2 = (F.dropout(torch.ones(i2u.shape[1]), self.message_dropout, True) > 0).to(dev)
                i2u, a2 = self._drop_edges(k2, i2u, a2)
        hs = [x]
        for conv in self.gnn_convs:
            x = conv(x, u2i, i2u, a1, a2)
            hs.append(x)
        combined = torch.cat(hs, dim=1) if self.concat else torch.mean(torch.stack(hs, dim=0), dim=0)
        item_emb, user_emb = combined[itemIds], combined[userIds]
        if self.MLP is not None:
            return self.MLP(torch.cat((item_emb, user_emb), dim=1))
        return torch.bmm(user_emb.unsqueeze(1), item_emb.unsqueeze(2)).view(-1, 1)
