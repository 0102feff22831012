"""CPU oracle for the NCF scoring hot path — TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (PyTorch fp32 on the host, no GPU, no native code) of the
forward arithmetic of michaelbzms/DeepRecommendation's models.  It is the *checker*:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``deeprecommendation_amd/`` imports it, and the product path
raises when the HIP library is missing instead of falling back to this file.

Every function cites the reference file:line it follows (paths relative to
``/root/reference/src/neural_collaborative_filtering``).  All functions are *functional*:
weights come in as a ``state`` dict whose keys are the reference's ``state_dict`` keys
(``item_embeddings.0.weight``, ``MLP.3.weight`` …), so a reference checkpoint can be fed
in unchanged.

Pinning (tests/test_oracle_golden.py):
  * BasicNCF / MF / AttentionNCF / build_MLP_layers / eval_ranking: pinned against golden
    vectors produced by importing the reference itself in the build container
    (tests/golden/make_golden.py -> tests/golden/*.npz).
  * LightGCNConv / GraphNCF: **parity unpinned** — the arithmetic partly lives in PyG 2.0.4 +
    torch-scatter 2.0.9 (reference env.yml:122,138), neither of which is installed and
    neither of which can be fetched.  The restatement follows PyG 2.0.4's published
    MessagePassing semantics (flow source_to_target: x_j = x[edge_index[0]], aggregation =
    scatter-sum over edge_index[1]; degree = scatter_add of ones; update = identity) and is
    covered by self-consistency tests only (dense-adjacency formulation vs edge list).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# util.py:5-18  build_MLP_layers
# --------------------------------------------------------------------------------------
def mlp_linear_indices(n_hidden: int, dropout_rate) -> List[int]:
    """Sequential indices of the Linear modules built by util.py:5-18.

    Layout is ``Linear, [ReLU, Dropout?, Linear]*``: with dropout the Linears sit at
    0, 3, 6 …; with ``dropout_rate is None`` (util.py:16) at 0, 2, 4 ….
    There are ``n_hidden + 1`` Linears (the last one has out_features = 1, util.py:10).
    """
    step = 3 if dropout_rate is not None else 2
    return [step * i for i in range(n_hidden + 1)]


def mlp_weights(state: State, prefix: str = "MLP") -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """Collect (weight, bias) of every Linear under ``prefix`` in Sequential order."""
    idx = sorted({int(k.split(".")[1]) for k in state if k.startswith(prefix + ".") and k.endswith(".weight")})
    return [(state[f"{prefix}.{i}.weight"], state[f"{prefix}.{i}.bias"]) for i in idx]


def mlp_forward(x: torch.Tensor, layers: Sequence[Tuple[torch.Tensor, torch.Tensor]]) -> torch.Tensor:
    """util.py:12-17 in eval mode: Linear, then (ReLU, Dropout=identity, Linear) per extra layer.

    No activation after the last Linear.
    """
    h = F.linear(x, layers[0][0], layers[0][1])
    for w, b in layers[1:]:
        h = F.linear(torch.relu(h), w, b)
    return h


# --------------------------------------------------------------------------------------
# models/basic_ncf.py:37-42, models/mf.py:28-32
# --------------------------------------------------------------------------------------
def basic_ncf_forward(state: State, X_user: torch.Tensor, X_item: torch.Tensor) -> torch.Tensor:
    """models/basic_ncf.py:37-42 — dense (reference-faithful) formulation."""
    user_emb = F.linear(X_user, state["user_embeddings.0.weight"], state["user_embeddings.0.bias"])  # :38
    item_emb = F.linear(X_item, state["item_embeddings.0.weight"], state["item_embeddings.0.bias"])  # :39
    combined = torch.cat((user_emb, item_emb), dim=1)  # :40  user first
    return mlp_forward(combined, mlp_weights(state))  # :41


def embedding_table(weight: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """``Linear(onehot(i)) == W[:, i] + b`` (basic_ncf.py:27-32 applied to a one-hot row from
    src/util.py:5-10).  Table row i = W[:, i] + b — one fp32 rounding, the same one the dense
    GEMM performs (every other product is an exact 0)."""
    return weight.t().contiguous() + bias


def basic_ncf_forward_indexed(state: State, user_pos: torch.Tensor, item_pos: torch.Tensor) -> torch.Tensor:
    """Table formulation of basic_ncf.py:37-42 for one-hot inputs: positions are the column of the
    one-hot row (= rank of the id among the sorted unique ids, one_hot_provider.py:14-15)."""
    tu = embedding_table(state["user_embeddings.0.weight"], state["user_embeddings.0.bias"])
    ti = embedding_table(state["item_embeddings.0.weight"], state["item_embeddings.0.bias"])
    combined = torch.cat((tu[user_pos], ti[item_pos]), dim=1)
    return mlp_forward(combined, mlp_weights(state))


def _bf16(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)


def basic_ncf_forward_indexed_bf16(state: State, user_pos: torch.Tensor, item_pos: torch.Tensor) -> torch.Tensor:
    """BASELINE config 5 arithmetic (no reference counterpart — the reference is fp32 only): tables and MLP weight
    matrices rounded to bf16 (round-to-nearest-even), products accumulated in fp32, hidden activations rounded to
    bf16 where they feed the next matrix product, biases / the final 1-wide layer / the output in fp32.
    Evaluated in fp64 on the rounded operands so that only the accumulation order differs from the GPU."""
    tu = _bf16(embedding_table(state["user_embeddings.0.weight"], state["user_embeddings.0.bias"]))
    ti = _bf16(embedding_table(state["item_embeddings.0.weight"], state["item_embeddings.0.bias"]))
    layers = mlp_weights(state)
    h = torch.cat((tu[user_pos], ti[item_pos]), dim=1).double()
    for li, (w, b) in enumerate(layers[:-1]):
        h = torch.relu(h @ _bf16(w).double().t() + b.double())
        if li < len(layers) - 2:
            h = _bf16(h.float()).double()  # feeds the next MFMA as a bf16 operand
    w, b = layers[-1]
    return (h @ w.double().t() + b.double()).float()


def mf_forward(state: State, X_user: torch.Tensor, X_item: torch.Tensor) -> torch.Tensor:
    """models/mf.py:28-32 — dot product of the two embeddings, shape (B, 1)."""
    user_emb = F.linear(X_user, state["user_embeddings.0.weight"], state["user_embeddings.0.bias"])
    item_emb = F.linear(X_item, state["item_embeddings.0.weight"], state["item_embeddings.0.bias"])
    return torch.bmm(user_emb.unsqueeze(1), item_emb.unsqueeze(2)).view(-1, 1)  # :31


def mf_forward_indexed(state: State, user_pos: torch.Tensor, item_pos: torch.Tensor) -> torch.Tensor:
    tu = embedding_table(state["user_embeddings.0.weight"], state["user_embeddings.0.bias"])
    ti = embedding_table(state["item_embeddings.0.weight"], state["item_embeddings.0.bias"])
    return torch.bmm(tu[user_pos].unsqueeze(1), ti[item_pos].unsqueeze(2)).view(-1, 1)


# --------------------------------------------------------------------------------------
# models/attention_ncf.py:136-224 (eval mode)
# --------------------------------------------------------------------------------------
def attention_ncf_forward(state: State, candidate_items: torch.Tensor, rated_items: torch.Tensor,
                          user_matrix: torch.Tensor, use_cos_sim_instead: bool = False,
                          return_attention_weights: bool = False, training: bool = False):
    """models/attention_ncf.py:136-224; ``training`` = ``self.training`` with every Dropout at p = 0 and
    ``message_dropout=None`` (the deterministic train mode): the only train-only step left is the target mask (:195-205).

    Reference-faithful: materialises all B*I candidate/rated pairs (:154-155), keeps the valid ones
    (``user_matrix != 0``, :158-159), scores them with AttentionNet (:176-179) or cosine similarity
    (:162-173), scatters into a (B, I) matrix of -inf (:182,192), row-softmax + nan_to_num (:208-209),
    multiplies by the signed ratings (:212), aggregates the rated items' *feature* rows (:213),
    then UserEmbeddings (:216), cat(candidate_emb, user_emb) (:219), MLP (:222).
    """
    I = rated_items.shape[0]
    B = candidate_items.shape[0]
    Wi, bi = state["ItemEmbeddings.0.weight"], state["ItemEmbeddings.0.bias"]
    cand_emb = F.linear(candidate_items, Wi, bi)  # :150
    rated_emb = F.linear(rated_items, Wi, bi)  # :151
    cand_full = cand_emb.repeat_interleave(I, dim=0)  # :154
    rated_full = rated_emb.repeat(B, 1)  # :155
    valid = user_matrix != 0  # :158
    cand_v = cand_full.view(B, I, -1)[valid]
    rated_v = rated_full.view(B, I, -1)[valid]
    if use_cos_sim_instead:
        a = F.normalize(cand_v, p=2, dim=1)  # :167
        b = F.normalize(rated_v, p=2, dim=1)  # :168
        att_out = torch.bmm(a.unsqueeze(1), b.unsqueeze(2)).view(-1)  # :170
    else:
        att_in = torch.cat((cand_v, rated_v), dim=1)  # :176
        att_layers = mlp_weights(state, "AttentionNet")  # Linear,(ReLU,Dropout,Linear)? :112-122
        att_out = mlp_forward(att_in, att_layers).view(-1)  # :179
    scores = -float("inf") * torch.ones((B, I), dtype=torch.float32)  # :182
    scores[valid] = att_out  # :192
    if training:
        # :195-205 — a candidate whose EMBEDDING equals a rated item's (isclose, atol 1e-5, every element) is removed from
        # its own softmax row, whether or not the user rated it
        same = torch.isclose(cand_full, rated_full, atol=1e-5).all(dim=1).view(B, I)  # :199
        scores[same] = -float("inf")  # :203
    scores = F.softmax(scores, dim=1)  # :208
    scores = scores.nan_to_num(nan=0.0, posinf=0.0, neginf=0.0)  # :209
    attended = torch.mul(scores, user_matrix)  # :212
    user_feat = torch.matmul(attended, rated_items)  # :213
    user_emb = F.linear(user_feat, state["UserEmbeddings.0.weight"], state["UserEmbeddings.0.bias"])  # :216
    combined = torch.cat((cand_emb, user_emb), dim=1)  # :219  candidate first
    out = mlp_forward(combined, mlp_weights(state))  # :222
    return (out, scores.detach()) if return_attention_weights else out  # :224


def mse_sum_loss(y_pred: torch.Tensor, y_true: torch.Tensor) -> torch.Tensor:
    """datasets/base.py:19-20,31-32: ``nn.MSELoss(reduction='sum')(y_pred, y_true.view(-1, 1).float())``."""
    return F.mse_loss(y_pred, y_true.view(-1, 1).float(), reduction="sum")


def loss_and_grads(forward, state: State, y: torch.Tensor):
    """Reference training step up to ``loss.backward()`` (train.py:99-107) through a functional forward of this file:
    returns (out, loss, {state key: d loss / d parameter}).  ``forward(state)`` must return the (B, 1) prediction."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in state.items()}
    out = forward(leaf)
    loss = mse_sum_loss(out, y)
    loss.backward()
    return out.detach(), loss.detach(), {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}


# --------------------------------------------------------------------------------------
# models/gnn_ncf.py:13-94 LightGCNConv, :298-367 GraphNCF.forward (eval)  — PARITY UNPINNED
# --------------------------------------------------------------------------------------
def pyg_degree(index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.degree (PyG 2.0.4): zeros(N).scatter_add_(0, index, ones)."""
    out = torch.zeros(num_nodes, dtype=torch.float32)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=torch.float32))


def pyg_propagate_add(edge_index: torch.Tensor, messages: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """MessagePassing.propagate(aggr='add') (PyG 2.0.4, flow source_to_target, node_dim=-2):
    scatter-sum of the per-edge messages over edge_index[1]; update() is the identity."""
    out = torch.zeros((num_nodes, messages.shape[1]), dtype=messages.dtype)
    return out.index_add_(0, edge_index[1], messages)


def lightgcn_conv(x: torch.Tensor, conv_state: State, hetero: bool,
                  user2item_edge_index: torch.Tensor, item2user_edge_index: torch.Tensor,
                  user2item_edge_attr: Optional[torch.Tensor] = None,
                  item2user_edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """models/gnn_ncf.py:39-94 in eval mode (Dropout = identity), per-EDGE Linear as the reference does.

    ``conv_state`` keys: ``user2item_W.0.weight/bias``, ``item2user_W.0.weight/bias`` (hetero, :22-29)
    or ``W.0.weight/bias`` (:33-36).
    """
    N = x.size(0)
    total_edges = torch.cat([user2item_edge_index, item2user_edge_index], dim=1)  # :41
    total_attr = None
    if user2item_edge_attr is not None and item2user_edge_attr is not None:
        total_attr = torch.cat([user2item_edge_attr, item2user_edge_attr], dim=0)  # :44
    from_, to_ = total_edges  # :47
    deg = pyg_degree(to_, N)  # :48
    dis = deg.pow(-0.5)  # :49
    dis[dis == float("inf")] = 0  # :50

    def message(ei, weight, norm, W, b):  # :74-94
        x_j = x[ei[0]]
        wx = F.linear(x_j, W, b)
        if weight is not None:
            return weight.view(-1, 1) * norm.view(-1, 1) * wx  # :91
        return norm.view(-1, 1) * wx  # :93

    if hetero:
        n1 = dis[user2item_edge_index[0]] * dis[user2item_edge_index[1]]  # :54
        out1 = pyg_propagate_add(user2item_edge_index,
                                 message(user2item_edge_index, user2item_edge_attr, n1,
                                         conv_state["user2item_W.0.weight"], conv_state["user2item_W.0.bias"]), N)
        n2 = dis[item2user_edge_index[0]] * dis[item2user_edge_index[1]]  # :58
        out2 = pyg_propagate_add(item2user_edge_index,
                                 message(item2user_edge_index, item2user_edge_attr, n2,
                                         conv_state["item2user_W.0.weight"], conv_state["item2user_W.0.bias"]), N)
        return out1 + out2  # :62
    norm = dis[from_] * dis[to_]  # :66
    return pyg_propagate_add(total_edges, message(total_edges, total_attr, norm,
                                                  conv_state["W.0.weight"], conv_state["W.0.bias"]), N)  # :69


def pyg_softmax(src: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.softmax (PyG 2.0.4): exp(src - max_group) / (sum_group + 1e-16), groups by ``index``."""
    mx = torch.full((num_nodes, src.shape[1]), -float("inf"), dtype=src.dtype)
    mx = mx.scatter_reduce(0, index.view(-1, 1).expand_as(src), src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    den = torch.zeros((num_nodes, src.shape[1]), dtype=src.dtype).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def lightgat_conv(x: torch.Tensor, conv_state: State, hetero: bool,
                  user2item_edge_index: torch.Tensor, item2user_edge_index: torch.Tensor,
                  user2item_edge_attr: Optional[torch.Tensor] = None,
                  item2user_edge_attr: Optional[torch.Tensor] = None) -> torch.Tensor:
    """models/gnn_ncf.py:128-177 in eval mode, per-edge formulation as the reference does (PARITY UNPINNED: PyG absent)."""
    N = x.size(0)

    def prop(ei, attr, pre):
        x_j, x_i = x[ei[0]], x[ei[1]]
        a = F.linear(torch.cat([x_j, x_i], dim=1), conv_state[f"{pre}AttNet.0.weight"], conv_state[f"{pre}AttNet.0.bias"])  # :158-167
        a = pyg_softmax(a, ei[1], N)  # :171
        wname = pre + "W" if pre else "W"
        wx = F.linear(x_j, conv_state[f"{wname}.0.weight"], conv_state[f"{wname}.0.bias"])
        msg = attr.view(-1, 1) * a * wx if attr is not None else a * wx  # :173-176
        return pyg_propagate_add(ei, msg, N)

    if hetero:
        return (prop(user2item_edge_index, user2item_edge_attr, "user2item_")
                + prop(item2user_edge_index, item2user_edge_attr, "item2user_"))  # :132-138
    ei = torch.cat([user2item_edge_index, item2user_edge_index], dim=1)
    attr = None
    if user2item_edge_attr is not None and item2user_edge_attr is not None:
        attr = torch.cat([user2item_edge_attr, item2user_edge_attr], dim=0)
    return prop(ei, attr, "")


def graph_ncf_forward(state: State, hetero: bool, num_gnn_layers: int, concat: bool, use_dot_product: bool,
                      item_features: torch.Tensor, user_features: torch.Tensor,
                      user2item_edge_index: torch.Tensor, item2user_edge_index: torch.Tensor,
                      user2item_edge_attr: Optional[torch.Tensor], item2user_edge_attr: Optional[torch.Tensor],
                      userIds: torch.Tensor, itemIds: torch.Tensor, convType: str = "LightGCN") -> torch.Tensor:
    """models/gnn_ncf.py:298-367 with ``self.training == False`` (no masking / dropout, :314-333 skipped).

    The conv weights are shared by every layer (:227) — state keys are ``gnn_convs.0.*`` (the ModuleList
    registers the same module L times; index 0 is taken).
    """
    item_emb = F.linear(item_features, state["item_embeddings.0.weight"], state["item_embeddings.0.bias"])  # :300
    user_emb = F.linear(user_features, state["user_embeddings.0.weight"], state["user_embeddings.0.bias"])  # :301
    graph_emb = torch.vstack([item_emb, user_emb])  # :304 items first
    conv_state = {k[len("gnn_convs.0."):]: v for k, v in state.items() if k.startswith("gnn_convs.0.")}
    hs = [graph_emb]
    for _ in range(num_gnn_layers):  # :337
        conv = lightgcn_conv if convType == "LightGCN" else lightgat_conv
        graph_emb = conv(graph_emb, conv_state, hetero, user2item_edge_index, item2user_edge_index,
                         user2item_edge_attr, item2user_edge_attr)
        hs.append(graph_emb)
    if concat:
        combined = torch.cat(hs, dim=1)  # :349
    else:
        combined = torch.mean(torch.stack(hs, dim=0), dim=0)  # :351
    item_e = combined[itemIds]  # :354
    user_e = combined[userIds]  # :357
    if not use_dot_product:
        return mlp_forward(torch.cat((item_e, user_e), dim=1), mlp_weights(state))  # :361-362 item first
    return torch.bmm(user_e.unsqueeze(1), item_e.unsqueeze(2)).view(-1, 1)  # :365


def lightgcn_conv_dense(x, conv_state, hetero, u2i_ei, i2u_ei, u2i_attr=None, i2u_attr=None):
    """Independent formulation used only to cross-check ``lightgcn_conv`` (self-consistency, float64):
    build the dense normalised adjacency A[dst, src] += w * dis[src] * dis[dst] and compute A @ (x W^T + b)."""
    N = x.size(0)
    xd = x.double()
    to_all = torch.cat([u2i_ei[1], i2u_ei[1]])
    deg = torch.zeros(N, dtype=torch.float64)
    for t in to_all.tolist():
        deg[t] += 1.0
    dis = torch.where(deg > 0, deg.pow(-0.5), torch.zeros_like(deg))

    def adj(ei, attr):
        A = torch.zeros((N, N), dtype=torch.float64)
        for e in range(ei.shape[1]):
            s, d = int(ei[0, e]), int(ei[1, e])
            w = 1.0 if attr is None else float(attr[e])
            A[d, s] += w * float(dis[s]) * float(dis[d])
        return A

    if hetero:
        z1 = xd @ conv_state["user2item_W.0.weight"].double().t() + conv_state["user2item_W.0.bias"].double()
        z2 = xd @ conv_state["item2user_W.0.weight"].double().t() + conv_state["item2user_W.0.bias"].double()
        return adj(u2i_ei, u2i_attr) @ z1 + adj(i2u_ei, i2u_attr) @ z2
    z = xd @ conv_state["W.0.weight"].double().t() + conv_state["W.0.bias"].double()
    ei = torch.cat([u2i_ei, i2u_ei], dim=1)
    attr = None if (u2i_attr is None or i2u_attr is None) else torch.cat([u2i_attr, i2u_attr])
    return adj(ei, attr) @ z


# --------------------------------------------------------------------------------------
# eval.py:25-75 eval_ranking (caller-side metric, restated without sklearn)
# --------------------------------------------------------------------------------------
def _dcg_at_k(y_true: np.ndarray, y_score: np.ndarray, k: int) -> float:
    """sklearn.metrics.dcg_score(ignore_ties=False) for one sample: ties in y_score share the mean
    gain of their tie group (sklearn _tie_averaged_dcg); discount 1/log2(rank+1), zero past k."""
    n = len(y_true)
    discount = 1.0 / (np.log(np.arange(n) + 2) / np.log(2))
    if k is not None:
        discount[k:] = 0
    discount_cumsum = np.cumsum(discount)
    _, inv, counts = np.unique(-y_score, return_inverse=True, return_counts=True)
    ranked = np.zeros(len(counts))
    np.add.at(ranked, inv, y_true)
    ranked /= counts
    groups = np.cumsum(counts) - 1
    discount_sums = np.empty(len(counts))
    discount_sums[0] = discount_cumsum[groups[0]]
    discount_sums[1:] = np.diff(discount_cumsum[groups])
    return float((ranked * discount_sums).sum())


def eval_ranking(user_ids: np.ndarray, ratings: np.ndarray, predictions: np.ndarray, cutoff: int = 10):
    """eval.py:25-75: per-user NDCG@cutoff and min-max 'adj-NDCG' (:62); users with <= 1 rows (:38) or
    ideal == worst DCG (:57) are skipped; the mean over the remaining users is returned (:69-70)."""
    ndcgs, adj = [], []
    order = np.argsort(user_ids, kind="stable")
    uids, starts = np.unique(user_ids[order], return_index=True)
    bounds = list(starts) + [len(order)]
    for g in range(len(uids)):
        rows = order[bounds[g]:bounds[g + 1]]
        if len(rows) <= 1:
            continue
        y_pred = predictions[rows].astype(np.float64)
        y_true = ratings[rows].astype(np.float64)
        dcg = _dcg_at_k(y_true, y_pred, cutoff)
        ideal = _dcg_at_k(y_true, y_true, cutoff)
        worst = _dcg_at_k(y_true, 5.0 - y_true, cutoff)
        ndcg = dcg / ideal if ideal != 0 else 0.0  # sklearn ndcg_score: 0 when all gains are 0
        if ideal == worst:
            continue
        ndcgs.append(ndcg)
        adj.append((dcg - worst) / (ideal - worst))
    return float(np.mean(ndcgs)), float(np.mean(adj))
