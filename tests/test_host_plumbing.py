"""CPU tests of the host-side mirror: providers, datasets, eval metrics, model contract (no GPU compute)."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden
from deeprecommendation_amd.content_providers.index_providers import (IndexGraphProvider, IndexProvider, OneHotProvider,
                                                                        SparseDynamicProvider)
from deeprecommendation_amd.neural_collaborative_filtering import eval as E
from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
from deeprecommendation_amd.neural_collaborative_filtering.datasets.gnn_datasets import GraphPointwiseDataset
from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF, SparseRatings
from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF
from deeprecommendation_amd.neural_collaborative_filtering.util import build_MLP_layers, load_model


def test_index_provider_matches_label_binarizer_columns():
    """Position == column of the 1 in the reference's one-hot row (src/util.py:5-10: LabelBinarizer fitted on the
    sorted unique ids)."""
    from sklearn.preprocessing import LabelBinarizer
    rng = np.random.default_rng(0)
    ids = rng.choice(10_000, 300, replace=False)
    p = IndexProvider(user_ids=ids, item_ids=ids[:50])
    q = rng.choice(ids, 64)
    lb = LabelBinarizer().fit(np.array(sorted(np.unique(ids))))
    assert np.array_equal(p.get_user_profile(tuple(q)), lb.transform(q).argmax(1))
    oh = OneHotProvider(user_ids=ids, item_ids=ids[:50])
    assert np.array_equal(oh.get_user_profile(tuple(q)), lb.transform(q))
    with pytest.raises(KeyError):
        p.get_user_profile((123456,))
    assert p.get_num_users() == 300 and p.get_num_items() == 50 and p.get_item_feature_dim() == 50


def test_eval_ranking_matches_reference_goldens():
    _, a, _ = load_golden("g5_eval_ranking")
    df = pd.DataFrame({"userId": a["userId"], "rating": a["rating"], "prediction": a["prediction"]})
    for k, key in ((5, "ndcg5"), (10, "ndcg10")):
        nd, adj = E.eval_ranking(df, cutoff=k)
        assert abs(nd - a[key][0]) < 1e-12 and abs(adj - a[key][1]) < 1e-12
    _, c, _ = load_golden("cfg1_basic_ml1m")
    df = pd.DataFrame({"userId": c["user_pos"] + 1, "rating": c["rating"], "prediction": c["out"].reshape(-1).astype(np.float64)})
    for k in (5, 10, 20):
        nd, adj = E.eval_ranking(df, cutoff=k)
        assert abs(nd - c[f"ndcg{k}"][0]) < 1e-12 and abs(adj - c[f"ndcg{k}"][1]) < 1e-12


def _loop_graph(users, items, ratings, all_users, all_items, binary):
    """Straight restatement of create_graph's loop (graph_providers.py:19-47) for the test."""
    all_items, all_users = sorted(set(all_items)), sorted(set(all_users))
    inode = {i: k for k, i in enumerate(all_items)}
    unode = {u: len(all_items) + k for k, u in enumerate(all_users)}
    df = pd.DataFrame({"userId": users, "movieId": items, "rating": ratings})
    um, im = df.groupby("userId")["rating"].mean(), df.groupby("movieId")["rating"].mean()
    e1, a1, e2, a2 = [], [], [], []
    for u, i, r in zip(users, items, ratings):
        ua, ia = (um.loc[u] + 2.5) / 2, (im.loc[i] + 2.5) / 2
        if not binary or r >= ua:
            e1.append([unode[u], inode[i]]); a1.append(r - ua)
        if not binary or r >= ia:
            e2.append([inode[i], unode[u]]); a2.append(r - ia)
    return np.array(e1).T, np.array(a1), np.array(e2).T, np.array(a2)


@pytest.mark.parametrize("binary", [False, True])
def test_index_graph_provider_matches_reference_loop(binary):
    rng = np.random.default_rng(1)
    users = rng.integers(100, 140, 300)
    items = rng.integers(7, 30, 300) * 3
    ratings = rng.integers(1, 11, 300) * 0.5
    gp = IndexGraphProvider(np.arange(100, 145), np.arange(0, 100), users, items, ratings, binary=binary)
    e1, a1, e2, a2 = _loop_graph(users, items, ratings, np.arange(100, 145), np.arange(0, 100), binary)
    g = gp.get_graph()
    assert np.array_equal(g.user2item_edge_index.numpy(), e1) and np.array_equal(g.item2user_edge_index.numpy(), e2)
    if binary:
        assert g.user2item_edge_attr is None and g.item2user_edge_attr is None
    else:
        assert np.allclose(g.user2item_edge_attr.numpy(), a1, rtol=0, atol=1e-6)
        assert np.allclose(g.item2user_edge_attr.numpy(), a2, rtol=0, atol=1e-6)
    assert gp.get_user_nodeID(100) == 100 and gp.get_item_nodeID(0) == 0   # items first, users after (:79-80)
    assert g.num_items == 100 and g.num_users == 45


@pytest.mark.parametrize("kind", ["weighted", "binary"])
def test_index_graph_provider_matches_reference_create_graph_golden(kind):
    """g8: the graph the REFERENCE's own create_graph (content_providers/graph_providers.py:10-66) builds from the same
    interactions — edge lists in the same order, edge weights, and the pos_df rows."""
    _, a, _ = load_golden("g8_create_graph_" + kind)
    gp = IndexGraphProvider(a["all_users"], a["all_items"], a["userId"], a["movieId"], a["rating"], binary=(kind == "binary"))
    g = gp.get_graph()
    assert np.array_equal(g.user2item_edge_index.numpy(), a["user2item_edge_index"])
    assert np.array_equal(g.item2user_edge_index.numpy(), a["item2user_edge_index"])
    if kind == "binary":
        assert g.user2item_edge_attr is None and g.item2user_edge_attr is None
        assert g.user2item_edge_index.shape[1] < len(a["userId"])          # the rating >= neutral filter dropped edges
    else:
        # the reference rounds python floats (float64 arithmetic) to float32 once: torch.tensor(list, dtype=float)
        assert np.array_equal(g.user2item_edge_attr.numpy(), a["user2item_edge_attr"])
        assert np.array_equal(g.item2user_edge_attr.numpy(), a["item2user_edge_attr"])
    assert np.array_equal(gp.pos_df().reset_index()[["Id1", "Id2", "pos"]].to_numpy(), a["pos"])


def test_sparse_dynamic_provider_matches_dense_reference_form():
    rng = np.random.default_rng(2)
    item_ids = np.arange(1000, 1040)
    feats = rng.random((40, 9)).astype(np.float32)
    user_ids = [5, 9, 11]
    rated = [np.sort(rng.choice(item_ids, n, replace=False)) for n in (6, 0, 13)]
    ratings = [rng.integers(1, 11, len(r)) * 0.5 for r in rated]
    means = [float(np.mean(r)) if len(r) else 3.0 for r in ratings]
    ratings[0][2] = (means[0] + 2.5) / 2  # a rating exactly at the neutral value -> entry 0.0 -> dropped (:158)
    p = SparseDynamicProvider(item_ids, feats, user_ids, rated, ratings, means, sparse=True)
    batch = [(5, 1003, 4.0), (9, 1001, 2.5), (11, 1010, 5.0), (5, 1039, 1.0)]
    cand_ids, rated_ids, cand, rated_feats, um, y = p.collate_interacted_items(batch, for_ranking=False)
    assert isinstance(um, SparseRatings)
    # dense restatement of dynamic_profiles_provider.py:55-71
    ref_rated = np.sort(np.unique(np.concatenate([rated[0], rated[1], rated[2], rated[0]])))
    dense = np.zeros((4, len(ref_rated)), dtype=np.float32)
    for row, uix in enumerate([0, 1, 2, 0]):
        dense[row, np.searchsorted(ref_rated, rated[uix])] = (ratings[uix] - (means[uix] + 2.5) / 2).astype(np.float32)
    assert np.array_equal(rated_ids, ref_rated)
    assert torch.equal(um.to_dense(um.val), torch.from_numpy(dense))
    assert int(um.rowptr[2] - um.rowptr[1]) == 0                      # user without ratings: empty row
    assert torch.equal(cand, torch.from_numpy(feats[[3, 1, 10, 39]]))
    assert torch.equal(rated_feats, torch.from_numpy(feats[ref_rated - 1000]))
    p2 = SparseDynamicProvider(item_ids, feats, user_ids, rated, ratings, means, sparse=False)
    assert torch.equal(p2.collate_interacted_items(batch, False)[4], torch.from_numpy(dense))


def test_dataset_collate_and_model_contract(tmp_path):
    df = pd.DataFrame({"userId": [3, 5, 3, 9], "movieId": [10, 20, 20, 10], "rating": [4.0, 2.5, 3.0, 5.0]})
    df.to_csv(tmp_path / "test.csv", index=False)
    prov = IndexProvider([3, 5, 9], [10, 20])
    ds = FixedPointwiseDataset(str(tmp_path / "test"), prov)       # file name without '.csv', as the reference passes it
    assert len(ds) == 4 and ds[1] == (5, 20, 2.5)
    u, i, y = ds.use_collate()([ds[k] for k in range(4)])
    assert u.dtype == torch.int64 and u.tolist() == [0, 1, 0, 2] and i.tolist() == [0, 1, 1, 0] and y.dtype == torch.float32
    m = BasicNCF(item_dim=2, user_dim=3, item_emb=8, user_emb=8, mlp_dense_layers=[16])
    assert m.is_dataset_compatible(FixedPointwiseDataset) and not m.is_dataset_compatible(GraphPointwiseDataset)
    RefLike = type("FixedRankingDataset", (), {})                   # the reference's class, matched by name
    assert m.is_dataset_compatible(RefLike)
    assert GraphNCF(2, 3, 2, True, node_emb=8).is_dataset_compatible(GraphPointwiseDataset)
    # checkpoint format [state_dict, kwargs] round-trips (reference models/base.py:18-19, util.py:21-34)
    m.save_model(tmp_path / "m.pt")
    m2 = load_model(tmp_path / "m.pt", BasicNCF)
    assert m2.kwargs == m.kwargs and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # embedding weights are stored id-major (transpose views) but keep the nn.Linear shape and state_dict entry: a plain
    # contiguous nn.Linear — what the reference's BasicNCF holds — loads the saved tensor as is
    w = m.user_embeddings[0].weight
    assert tuple(w.shape) == (8, 3) and not w.is_contiguous() and w.t().is_contiguous()
    saved, _ = torch.load(tmp_path / "m.pt", weights_only=True)
    ref_like = torch.nn.Linear(3, 8)
    ref_like.load_state_dict({"weight": saved["user_embeddings.0.weight"], "bias": saved["user_embeddings.0.bias"]})
    assert ref_like.weight.is_contiguous() and torch.equal(ref_like.weight, w)
    x = torch.eye(3)
    assert torch.equal(ref_like(x), m.user_embeddings(x))
    # training step stays on differentiable torch ops and works on CPU
    m.train()
    out = m(u, i)
    ds.calculate_loss(out, y).backward()
    assert m.user_embeddings[0].weight.grad is not None
    keys = list(build_MLP_layers(10, [8, 4], dropout_rate=None).state_dict().keys())
    assert keys == ["0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias"]


def test_attention_and_graph_training_paths_run_on_cpu():
    torch.manual_seed(0)
    att = AttentionNCF(item_dim=12, item_emb=8, user_emb=8, att_dense=4, mlp_dense_layers=[8], message_dropout=0.2).train()
    cand, rated = torch.rand(5, 12), torch.rand(7, 12)
    um = torch.randn(5, 7) * (torch.rand(5, 7) < 0.5)
    out = att(cand, rated, um)
    out.sum().backward()
    assert out.shape == (5, 1) and att.ItemEmbeddings[0].weight.grad is not None
    gp = IndexGraphProvider(np.arange(6), np.arange(4), [0, 1, 2, 3, 4, 5, 0], [0, 1, 2, 3, 0, 1, 2], [4.0, 3, 5, 1, 2, 4.5, 3.5])
    g = GraphNCF(item_dim=4, user_dim=6, num_gnn_layers=2, hetero=True, node_emb=8, mlp_dense_layers=[8],
                 message_dropout=0.1, node_dropout=0.1).train()
    users = torch.tensor([gp.get_user_nodeID(0), gp.get_user_nodeID(3)])
    items = torch.tensor([gp.get_item_nodeID(0), gp.get_item_nodeID(3)])
    out = g(gp.get_graph(), users, items, torch.device("cpu"))
    out.sum().backward()
    assert out.shape == (2, 1) and g.gnn_convs[0].user2item_W[0].weight.grad is not None


def test_segmented_csr_tree_structure_cpu():
    """native.SegmentedCSR is pure index plumbing (torch ops) until spmm() is called: its level structure can be checked
    on CPU.  Level 0 partitions every row's edges into <= seg_len pieces; each further level groups the previous
    level's partial sums of the still-split rows into <= fan pieces; the last level has exactly one segment per row."""
    from deeprecommendation_amd.native import SegmentedCSR
    counts = torch.tensor([0, 3, 1300, 7, 0, 70000, 512, 513])
    rowptr = torch.zeros(len(counts) + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    E = int(rowptr[-1])
    csr = SegmentedCSR(rowptr, torch.zeros(E, dtype=torch.int32), None, seg_len=512, fan=8)
    segptr, row_of, _ = csr.levels[0]
    assert int(segptr[0]) == 0 and int(segptr[-1]) == E
    lens = segptr[1:] - segptr[:-1]
    assert int(lens.max()) <= 512 and bool((lens >= 0).all())
    per_row = torch.bincount(row_of.long(), weights=lens.double(), minlength=len(counts))
    assert torch.equal(per_row.long(), counts)                       # every edge in exactly one segment of its row
    assert torch.equal(torch.bincount(row_of.long(), minlength=len(counts)) >= 1, torch.ones(len(counts), dtype=torch.bool))
    assert bool((row_of[1:] >= row_of[:-1]).all())                   # a row's segments are consecutive
    prev_nseg = torch.bincount(row_of.long(), minlength=len(counts))
    assert len(csr.levels) == 4                                      # 70000 edges: 137 partials -> 18 -> 3 -> 1
    for segptr, row_of, edge_ids in csr.levels[1:]:
        lens = segptr[1:] - segptr[:-1]
        assert int(lens.max()) <= 8
        rows = torch.unique(row_of.long())
        assert set(rows.tolist()) == set(torch.nonzero(prev_nseg > 1).flatten().tolist())   # exactly the still-split rows
        assert int(lens.sum()) == int(edge_ids.numel()) == int(prev_nseg[prev_nseg > 1].sum())
        assert bool((edge_ids[1:] > edge_ids[:-1]).all())            # partials are consumed in index (= edge) order
        prev_nseg = torch.bincount(row_of.long(), minlength=len(counts))
    assert int(prev_nseg.max()) == 1


def test_from_dense_shares_exactly_equal_rows():
    """SparseRatings.from_dense: repeated rows (the reference's datasets repeat a user's row per sample) share one CSR
    row after an element-by-element verification; to_dense / expanded() give the original matrix back; a batch of
    distinct rows keeps one row per pair."""
    g = torch.Generator().manual_seed(0)
    rows = torch.zeros(5, 40)
    mask = torch.rand(5, 40, generator=g) < 0.3
    rows[mask] = (torch.randint(1, 11, (5, 40), generator=g).float() * 0.5 - 2.9)[mask]
    rows[3] = 0.0                                            # a user without ratings
    pair_row = torch.randint(0, 5, (64,), generator=g)
    um = rows[pair_row]
    sr = SparseRatings.from_dense(um)
    assert sr.pair_row is not None and sr.rowptr.numel() - 1 == int(torch.unique(pair_row).numel())
    assert torch.equal(sr.to_dense(sr.val), um)
    ex = sr.expanded()
    assert ex.pair_row is None and torch.equal(ex.to_dense(ex.val), um)
    plain = SparseRatings.from_dense(um, share_identical_rows=False)
    assert torch.equal(ex.rowptr, plain.rowptr) and torch.equal(ex.col, plain.col) and torch.equal(ex.val, plain.val)
    distinct = SparseRatings.from_dense(torch.rand(16, 40, generator=g))
    assert distinct.pair_row is None


def test_gpu_only_helpers_fail_loudly_on_cpu():
    """No CPU fallback: the HIP-side helpers added for grouped attention, graph replay and the fused optimiser raise on
    CPU tensors / without a GPU instead of computing something else."""
    from deeprecommendation_amd import native
    from deeprecommendation_amd.optim import FusedAdam
    assert native.default_pairs_per_wg(4096) == 32 and native.default_pairs_per_wg(600) == 16 and native.default_pairs_per_wg(10) == 8
    assert native.attn_grouped_supported(native.ATT_MLP, 128, 64) and not native.attn_grouped_supported(native.ATT_MLP, 130, 64)
    p = torch.nn.Parameter(torch.zeros(4, 4))
    p.grad = torch.ones(4, 4)
    with pytest.raises(RuntimeError, match="GPU"):
        FusedAdam([p]).step()
    with pytest.raises(ValueError):
        FusedAdam([p], lr=-1.0)
    if not torch.cuda.is_available():
        from deeprecommendation_amd.graphs import GraphedForward
        with pytest.raises(RuntimeError, match="GPU"):
            GraphedForward(lambda x: x, [torch.zeros(2)])
        with pytest.raises(RuntimeError):
            native.group_pairs(torch.zeros(4, dtype=torch.int64), 2, 8)


def test_resident_inputs_match_the_dataloader_batches():
    """Dataset.resident_inputs() (what eval_model keeps on the GPU) == the concatenation of the DataLoader's collated
    batches, for index providers and graph datasets; dense-profile providers have none; BCE targets are scaled alike."""
    import pandas as pd
    from torch.utils.data import DataLoader
    from deeprecommendation_amd.content_providers.index_providers import IndexGraphProvider, IndexProvider, OneHotProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.gnn_datasets import GraphPointwiseDataset
    rng = np.random.default_rng(0)
    n = 257
    frame = pd.DataFrame({"userId": rng.integers(10, 60, n) * 3, "movieId": rng.integers(1, 30, n) * 7, "rating": rng.integers(1, 11, n) * 0.5})
    users, items = np.arange(10, 60) * 3, np.arange(1, 30) * 7
    for ds in (FixedPointwiseDataset(frame, IndexProvider(users, items)),
               GraphPointwiseDataset(frame, IndexGraphProvider(users, items, frame["userId"].values, frame["movieId"].values, frame["rating"].values))):
        res = ds.resident_inputs()
        (u, i), y = res.tensors, res.targets
        assert res.on_chunk is None and res.on_batch is None  # no device given: positions resolved on the host
        batches = list(DataLoader(ds, batch_size=50, collate_fn=ds.use_collate()))
        assert torch.equal(u, torch.cat([b[0] for b in batches])) and torch.equal(i, torch.cat([b[1] for b in batches]))
        assert torch.equal(y, torch.cat([b[2] for b in batches])) and u.dtype == torch.int64 and y.dtype == torch.float32
    assert FixedPointwiseDataset(frame, OneHotProvider(users, items)).resident_inputs() is None
    bce = FixedPointwiseDataset(frame, IndexProvider(users, items))
    bce.use_bce_loss = True
    assert torch.allclose(bce.resident_inputs().targets, torch.as_tensor(frame["rating"].values / 5.0, dtype=torch.float32))
    # the on-GPU id lookup (run here on the CPU device): raw ids -> the same positions; unknown ids -> -1
    prov = IndexProvider(users, items)
    f = prov.device_lookup(torch.device("cpu"))
    raw = FixedPointwiseDataset(frame, prov).resident_inputs(torch.device("cpu"))
    assert raw.on_chunk is not None and torch.equal(raw.tensors[0], torch.as_tensor(frame["userId"].values))
    up, ip = raw.on_chunk(*raw.tensors)
    assert torch.equal(up, torch.from_numpy(prov.get_user_profile(frame["userId"].values)))
    assert torch.equal(ip, torch.from_numpy(prov.get_item_profile(frame["movieId"].values)))
    up, ip = f(torch.tensor([30, 31, 29, 10 ** 9, -7]), torch.tensor([7, 8, 0, 203, 210]))
    assert up.tolist() == [0, -1, -1, -1, -1] and ip.tolist() == [0, -1, -1, 28, -1]
    sparse = IndexProvider(np.array([1, 10 ** 9]), items)
    assert sparse.device_lookup(torch.device("cpu")) is None
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    with pytest.raises(ValueError):
        eval_model(BasicNCF(item_dim=29, user_dim=50, item_emb=8, user_emb=8, mlp_dense_layers=[16]), bce, 32, device="cpu", resident=True)


def test_dynamic_provider_device_state_equals_the_collate():
    """SparseDynamicProvider.device_state (run here on the CPU device): candidate rows and the users' rated sets of a
    batch assembled from the resident whole-catalogue CSR == collate_interacted_items, column for column on the batch's
    union of rated items and zero elsewhere; unknown ids raise."""
    rng = np.random.default_rng(0)
    I, F, U = 40, 6, 15
    item_ids = np.arange(100, 100 + I) * 2
    feats = rng.normal(size=(I, F)).astype(np.float32)
    user_ids = rng.permutation(np.arange(7, 7 + U))
    rated = [np.sort(rng.choice(item_ids, rng.integers(1, 9), replace=False)) for _ in range(U)]
    ratings = [rng.integers(1, 11, len(r)) * 0.5 for r in rated]
    means = np.array([r.mean() for r in ratings])
    prov = SparseDynamicProvider(item_ids, feats, user_ids, rated, ratings, means, sparse=True)
    st = prov.device_state(torch.device("cpu"))
    assert st is prov.device_state(torch.device("cpu"))  # cached
    users, cands, tg = rng.choice(user_ids, 20), rng.choice(item_ids, 20), rng.random(20)
    _, rated_ids, candidate_items, _, um, _ = prov.collate_interacted_items(list(zip(users.tolist(), cands.tolist(), tg.tolist())), for_ranking=False)
    got = st.batch(torch.as_tensor(users), torch.as_tensor(cands), torch.as_tensor(tg), pairs_per_row_hint=2.5)
    assert torch.equal(got[2].materialise(), candidate_items) and got[2].table is got[3] is st.features and got[4].pairs_per_row == 2.5
    assert got[2].float().to(torch.device("cpu")) is got[2] and got[2].shape == candidate_items.shape
    local, whole = um.to_dense(um.val), got[4].to_dense(got[4].val)
    colpos = np.searchsorted(prov.item_ids, rated_ids)
    assert torch.equal(whole[:, colpos], local)
    rest = np.ones(I, bool)
    rest[colpos] = False
    assert float(whole[:, rest].abs().sum()) == 0.0
    # large user bases: the chunk's distinct users are compacted into a chunk-local CSR; same dense matrix
    st.COMPACT_MIN_USERS = 0
    compact = st.batch(torch.as_tensor(users), torch.as_tensor(cands), torch.as_tensor(tg))
    assert compact[4].rowptr.numel() - 1 == len(np.unique(users)) < U
    assert torch.equal(compact[4].to_dense(compact[4].val), whole)
    del st.COMPACT_MIN_USERS
    with pytest.raises(KeyError):
        st.batch(torch.tensor([9999]), torch.tensor([200]), torch.tensor([0.0]))
    sparse_ids = SparseDynamicProvider(np.array([1, 10 ** 9]), feats[:2], user_ids, [np.array([1])] * U, [np.array([3.0])] * U, means)
    assert sparse_ids.device_state(torch.device("cpu")) is None


@pytest.mark.parametrize("n,n_users", [(2000, 100), (30000, 2000), (7, 7), (300, 1), (0, 1)])
def test_eval_ranking_device_equals_eval_ranking(n, n_users):
    """The tensor implementation of the ranking metrics (run here on the CPU device; eval_model runs it on the GPU)
    against the numpy one that is pinned to the reference's NDCG goldens: heavy score ties, single-row users, users whose
    ideal and worst DCG coincide, all three cut-offs from one ordering."""
    rng = np.random.default_rng(n + n_users)
    users = rng.integers(0, n_users, n) * 7 + 3
    rating = rng.integers(1, 11, n) * 0.5
    pred = np.round(rng.normal(3, 1, n), 1).astype(np.float32).astype(np.float64)
    got = E.eval_ranking_device(users, rating, pred, (5, 10, 20))
    if n == 0:
        assert all(np.isnan(v[0]) for v in got.values())
        return
    frame = pd.DataFrame({"userId": users, "rating": rating, "prediction": pred})
    for k in (5, 10, 20):
        assert np.allclose(got[k], E.eval_ranking(frame, cutoff=k), rtol=1e-12, atol=1e-12, equal_nan=True)


def test_early_stopping_follows_the_reference_rule():
    """EarlyStopping against a literal replay of train.py:157-210's counters (point-wise branch) on random loss curves."""
    from deeprecommendation_amd.neural_collaborative_filtering.train import EarlyStopping
    rng = np.random.default_rng(4)
    for trial in range(200):
        patience, max_patience = int(rng.integers(0, 4)), int(rng.integers(1, 7))
        losses = np.round(rng.random(30) + np.linspace(1, 0.5, 30) * rng.random(), 2)   # rounding: exact ties occur
        es = EarlyStopping(patience, max_patience)
        times, best, prev, budget = 0, None, None, max_patience
        for epoch, v in enumerate(losses):
            if best is None or v < best:
                best, times, budget, want = v, 0, max_patience, "best"
            else:
                if prev is not None and v > prev:
                    times += 1
                else:
                    times = max(0, times - 1)
                budget -= 1
                want = "stop" if (times > patience or budget <= 0) else "continue"
            prev = v
            got = es.update(float(v), epoch)
            assert got == want, (trial, epoch)
            assert es.best == best and es.strikes == times
            if want == "stop":
                break


def test_host_thread_cap_never_raises_the_count_and_respects_the_share():
    """util.cap_host_threads: torch's intra-op pool is lowered to the CPU share (affinity mask capped by the cgroup quota)
    when it exceeds it, and never raised."""
    from deeprecommendation_amd.neural_collaborative_filtering.util import cap_host_threads, host_cpu_share
    share = host_cpu_share()
    assert 1 <= share <= (os.cpu_count() or 1)
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(1)
        assert cap_host_threads() == 1                      # below the share: untouched
        torch.set_num_threads(share + 3)
        assert cap_host_threads() == share == torch.get_num_threads()
    finally:
        torch.set_num_threads(before)
