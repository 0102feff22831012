"""GPU end-to-end: provider -> dataset -> eval_model (the reference's evaluate_model.py call pattern) on the HIP path."""
import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden
from oracle import ncf_oracle as O
from test_gpu_basic import assert_close

pytestmark = pytest.mark.gpu


def test_cfg1_eval_model_end_to_end(gpu):
    """BASELINE configs[0] through the whole plumbing: IndexProvider -> FixedPointwiseDataset -> DataLoader(512) ->
    do_forward -> BasicNCF (HIP) -> MSE + NDCG, against the metrics the REFERENCE produced on the same data."""
    from deeprecommendation_amd.content_providers.index_providers import IndexProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    state, a, kw = load_golden("cfg1_basic_ml1m")
    U, I = kw["user_dim"], kw["item_dim"]
    frame = pd.DataFrame({"userId": a["user_pos"] + 1, "movieId": a["item_pos"] + 1, "rating": a["rating"]})
    ds = FixedPointwiseDataset(frame, IndexProvider(np.arange(1, U + 1), np.arange(1, I + 1)))
    m = BasicNCF(**kw)
    m.load_state_dict(state)
    res = eval_model(m.to(gpu), ds, batch_size=512, ranking=False, device=gpu)
    assert_close(torch.from_numpy(res["predictions"]).float().view(-1, 1), torch.from_numpy(a["out"]))
    assert abs(res["mse"] - float(a["mse"])) <= 1e-5 * float(a["mse"])
    for k in (5, 10, 20):
        assert abs(res[f"ndcg@{k}"] - a[f"ndcg{k}"][0]) < 1e-4 and abs(res[f"adj_ndcg@{k}"] - a[f"ndcg{k}"][1]) < 1e-4


def test_graph_pipeline_end_to_end(gpu):
    from deeprecommendation_amd.content_providers.index_providers import IndexGraphProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.gnn_datasets import GraphPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF
    rng = np.random.default_rng(3)
    n = 3000
    key = np.unique(rng.integers(0, 200, n) * 1000 + rng.integers(0, 60, n))
    users, items = key // 1000 + 1, key % 1000 + 1
    ratings = rng.integers(1, 11, len(key)) * 0.5
    gp = IndexGraphProvider(np.arange(1, 201), np.arange(1, 61), users, items, ratings)
    test = pd.DataFrame({"userId": users[::3], "movieId": items[::3], "rating": ratings[::3]})
    ds = GraphPointwiseDataset(test, gp)
    torch.manual_seed(1)
    m = GraphNCF(item_dim=60, user_dim=200, num_gnn_layers=2, hetero=True, node_emb=64, mlp_dense_layers=[128]).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = gp.get_graph()
    ref = O.graph_ncf_forward(state, True, 2, False, False, torch.eye(60), torch.eye(200), g.user2item_edge_index,
                              g.item2user_edge_index, g.user2item_edge_attr, g.item2user_edge_attr,
                              torch.as_tensor(gp.get_user_nodeID(test["userId"].values)), torch.as_tensor(gp.get_item_nodeID(test["movieId"].values)))
    res = eval_model(m.to(gpu), ds, batch_size=128, device=gpu)
    assert_close(torch.from_numpy(res["predictions"]).float().view(-1, 1), ref)
    assert 0.0 < res["ndcg@10"] <= 1.0


def test_attention_pipeline_end_to_end(gpu):
    from deeprecommendation_amd.content_providers.index_providers import SparseDynamicProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.dynamic_datasets import DynamicPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    rng = np.random.default_rng(4)
    I, F, nu = 80, 30, 25
    item_ids = np.arange(500, 500 + I)
    feats = rng.random((I, F)).astype(np.float32)
    rated = [np.sort(rng.choice(item_ids, rng.integers(1, 30), replace=False)) for _ in range(nu)]
    ratings = [rng.integers(1, 11, len(r)) * 0.5 for r in rated]
    means = [float(r.mean()) for r in ratings]
    test = pd.DataFrame({"userId": rng.integers(0, nu, 300), "movieId": rng.choice(item_ids, 300), "rating": rng.integers(1, 11, 300) * 0.5})
    torch.manual_seed(2)
    m = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=128, mlp_dense_layers=[256, 128]).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    outs = {}
    for sparse in (True, False):
        prov = SparseDynamicProvider(item_ids, feats, np.arange(nu), rated, ratings, means, sparse=sparse)
        outs[sparse] = eval_model(m.to(gpu), DynamicPointwiseDataset(test, prov), batch_size=64, device=gpu)["predictions"]
    # CSR ratings == dense user_matrix.  The two forms may group the pairs differently (shared CSR rows vs rows found equal in the
    # dense matrix) and so take kernels with another order of the softmax partial sums: equal to fp32 rounding, not bit for bit
    assert_close(torch.from_numpy(outs[True]), torch.from_numpy(outs[False]))
    # oracle on the first batch, dense reference formulation
    prov = SparseDynamicProvider(item_ids, feats, np.arange(nu), rated, ratings, means, sparse=False)
    batch = [tuple(test.iloc[k][["userId", "movieId", "rating"]]) for k in range(64)]
    _, _, cand, rated_f, um, _ = prov.collate_interacted_items([(int(u), int(i), r) for u, i, r in batch], False)
    ref = O.attention_ncf_forward(state, cand, rated_f, um)
    assert_close(torch.from_numpy(outs[True][:64]).float().view(-1, 1), ref)


def test_hip_graph_replay_matches_eager(gpu):
    """GraphedForward (deeprecommendation_amd/graphs.py): a captured AttentionNCF step on shared-row ratings and a
    captured BasicNCF scoring step replay bit-identically to the eager calls, on the captured batch and on new ones
    (incl. a batch with fewer users / entries than the captured capacity)."""
    from deeprecommendation_amd.graphs import GraphedForward
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF, SparseRatings
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    torch.manual_seed(3)
    I, F, B = 300, 48, 256
    m = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=128, mlp_dense_layers=[256, 128]).eval().to(gpu)
    g = torch.Generator().manual_seed(4)
    rated = torch.rand(I, F, generator=g).to(gpu)

    def batch(seed, users):
        gg = torch.Generator().manual_seed(seed)
        counts = torch.randint(5, 90, (users,), generator=gg)
        rowptr = torch.zeros(users + 1, dtype=torch.int64)
        rowptr[1:] = torch.cumsum(counts, 0)
        col = torch.cat([torch.randperm(I, generator=gg)[:int(c)].sort().values for c in counts]).to(torch.int32)
        val = torch.randint(1, 11, (int(rowptr[-1]),), generator=gg).float() * 0.5 - 2.9
        pair_row = torch.randint(0, users, (B,), generator=gg)
        return torch.rand(B, F, generator=gg).to(gpu), SparseRatings(rowptr.to(gpu), col.to(gpu), val.to(gpu), I, pair_row=pair_row.to(gpu))

    cand0, r0 = batch(1, 12)
    with torch.no_grad():
        graphed = GraphedForward(lambda c, rt, um: m(c, rt, um), [cand0, rated, r0])
        for seed, users in ((1, 12), (2, 12), (3, 7)):      # the last one: fewer rows and entries than captured
            cand, r = batch(seed, users)
            if r.col.numel() > r0.col.numel():
                continue
            eager = m(cand, rated, r)
            replay = graphed(cand, rated, r).clone()
            assert torch.equal(eager, replay)
    # BasicNCF scoring
    bm = BasicNCF(item_dim=700, user_dim=3000, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128]).eval().to(gpu)
    u0 = torch.randint(0, 3000, (4096,), generator=g).to(gpu)
    i0 = torch.randint(0, 700, (4096,), generator=g).to(gpu)
    with torch.no_grad():
        gb = GraphedForward(lambda u, i: bm(u, i), [u0, i0])
        u1 = torch.randint(0, 3000, (4096,), generator=g).to(gpu)
        i1 = torch.randint(0, 700, (4096,), generator=g).to(gpu)
        assert torch.equal(bm(u1, i1), gb(u1, i1).clone())
        assert torch.equal(bm(u0, i0), gb(u0, i0).clone())


@pytest.mark.parametrize("n,batch", [(10_000, 512), (70_000, 1000), (513, 512), (5, 64)])
def test_resident_eval_equals_dataloader_eval(gpu, monkeypatch, n, batch):
    """eval_model's device-resident path (ids uploaded in chunks on a copy stream, loss and predictions kept on the GPU)
    against the reference-shaped DataLoader loop: same batches through the same do_forward -> identical predictions and
    the same per-batch loss sums; several upload chunks (chunk = 4 batches here) and a ragged last chunk / batch."""
    from deeprecommendation_amd.content_providers.index_providers import IndexProvider
    from deeprecommendation_amd.neural_collaborative_filtering import eval as E
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    monkeypatch.setattr(E, "RESIDENT_CHUNK_BATCHES", 4)
    rng = np.random.default_rng(n + batch)
    U, I = 3000, 800
    frame = pd.DataFrame({"userId": rng.integers(1, U + 1, n), "movieId": rng.integers(1, I + 1, n),
                          "rating": rng.integers(1, 11, n) * 0.5})
    ds = FixedPointwiseDataset(frame, IndexProvider(np.arange(1, U + 1), np.arange(1, I + 1)))
    torch.manual_seed(0)
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128]).to(gpu)
    fast = E.eval_model(m, ds, batch_size=batch, device=gpu, resident=True)
    slow = E.eval_model(m, ds, batch_size=batch, device=gpu, resident=False)
    assert np.array_equal(fast["predictions"], slow["predictions"])
    assert abs(fast["mse"] - slow["mse"]) <= 1e-12 * slow["mse"]
    for k in (5, 10, 20):
        # same predictions -> same metric up to the order of the float64 atomic adds of the device bincount; nan: no user with 2+ rows
        assert np.isclose(fast[f"ndcg@{k}"], slow[f"ndcg@{k}"], rtol=1e-12, atol=0, equal_nan=True)


def test_resident_eval_graph_dataset_and_refusals(gpu):
    from deeprecommendation_amd.content_providers.index_providers import IndexGraphProvider, OneHotProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.gnn_datasets import GraphPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF
    rng = np.random.default_rng(5)
    key = np.unique(rng.integers(0, 150, 4000) * 1000 + rng.integers(0, 50, 4000))
    users, items = key // 1000 + 1, key % 1000 + 1
    ratings = rng.integers(1, 11, len(key)) * 0.5
    gp = IndexGraphProvider(np.arange(1, 151), np.arange(1, 51), users, items, ratings)
    test = pd.DataFrame({"userId": users[::2], "movieId": items[::2], "rating": ratings[::2]})
    ds = GraphPointwiseDataset(test, gp)
    torch.manual_seed(2)
    m = GraphNCF(item_dim=50, user_dim=150, num_gnn_layers=2, hetero=False, node_emb=64, mlp_dense_layers=[128]).to(gpu)
    fast = eval_model(m, ds, batch_size=100, device=gpu, resident=True)
    slow = eval_model(m, ds, batch_size=100, device=gpu, resident=False)
    assert np.array_equal(fast["predictions"], slow["predictions"]) and abs(fast["mse"] - slow["mse"]) <= 1e-12 * slow["mse"]
    # dense one-hot profiles have no resident form: auto falls back to the loader, an explicit request is refused
    dense = FixedPointwiseDataset(test, OneHotProvider(np.arange(1, 151), np.arange(1, 51)))
    assert dense.resident_inputs() is None
    b = BasicNCF(item_dim=50, user_dim=150, item_emb=32, user_emb=32, mlp_dense_layers=[128]).to(gpu)
    eval_model(b, dense, batch_size=100, device=gpu)
    with pytest.raises(ValueError):
        eval_model(b, dense, batch_size=100, device=gpu, resident=True)
    # an id the provider does not know: KeyError from the host lookup (loader path), IndexError from the kernels' out-of-range
    # flag when the lookup runs on the GPU (resident path) — never a silent score
    from deeprecommendation_amd.content_providers.index_providers import IndexProvider
    n = 6000
    frame = pd.DataFrame({"userId": rng.integers(1, 151, n), "movieId": rng.integers(1, 51, n), "rating": np.full(n, 3.0)})
    frame.loc[n - 3, "userId"] = 999
    bad = FixedPointwiseDataset(frame, IndexProvider(np.arange(1, 151), np.arange(1, 51)))
    b2 = BasicNCF(item_dim=50, user_dim=150, item_emb=64, user_emb=64, mlp_dense_layers=[128]).to(gpu)
    with pytest.raises(KeyError):
        eval_model(b2, bad, batch_size=512, device=gpu, resident=False)
    with pytest.raises(IndexError):
        eval_model(b2, bad, batch_size=512, device=gpu, resident=True)


@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("att_dense", [16, None])
def test_resident_eval_attention_dataset(gpu, monkeypatch, att_dense, compact):
    """DynamicPointwiseDataset through the device-resident loop (whole-catalogue CSR + feature table on the GPU, raw ids
    uploaded) against the DataLoader loop with the host collate: same scores within 1e-5 (the rated set is the whole
    catalogue instead of the batch's union: another GEMM shape for the catalogue projections, same sums per user)."""
    from deeprecommendation_amd.content_providers import index_providers
    from deeprecommendation_amd.content_providers.index_providers import SparseDynamicProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.dynamic_datasets import DynamicPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    if compact:   # the path user bases above 4096 take: chunk-local CSR of the chunk's distinct users
        monkeypatch.setattr(index_providers._DynamicDeviceState, "COMPACT_MIN_USERS", 0)
    rng = np.random.default_rng(11)
    I, F, U = 300, 40, 60
    item_ids = np.arange(1, I + 1) * 3
    feats = rng.normal(size=(I, F)).astype(np.float32)
    user_ids = rng.permutation(np.arange(1000, 1000 + U))
    rated = [np.sort(rng.choice(item_ids, rng.integers(1, 40), replace=False)) for _ in range(U)]
    ratings = [rng.integers(1, 11, len(r)) * 0.5 for r in rated]
    means = np.array([r.mean() for r in ratings])
    prov = SparseDynamicProvider(item_ids, feats, user_ids, rated, ratings, means, sparse=True)
    n = 3000
    # samples sorted by user (many pairs per user in a batch -> grouped kernel) and shuffled (-> per-pair kernel)
    for shuffle in (False, True):
        u = np.repeat(user_ids, n // U) if not shuffle else rng.choice(user_ids, n)
        frame = pd.DataFrame({"userId": u, "movieId": rng.choice(item_ids, len(u)), "rating": rng.integers(1, 11, len(u)) * 0.5})
        ds = DynamicPointwiseDataset(frame, prov)
        torch.manual_seed(3)
        m = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=att_dense, mlp_dense_layers=[128]).to(gpu)
        batch = 200 if not shuffle else 30
        fast = eval_model(m, ds, batch_size=batch, device=gpu, resident=True)
        slow = eval_model(m, ds, batch_size=batch, device=gpu, resident=False)
        assert_close(torch.from_numpy(fast["predictions"]).float().view(-1, 1), torch.from_numpy(slow["predictions"]).float().view(-1, 1))
        assert abs(fast["mse"] - slow["mse"]) <= 1e-5 * slow["mse"]
    frame.loc[7, "movieId"] = 5  # not a catalogue id
    with pytest.raises(IndexError):
        eval_model(m, DynamicPointwiseDataset(frame, prov), batch_size=64, device=gpu, resident=True)


def test_back_to_back_graph_replays_of_every_model_family(gpu):
    """A captured forward of each model family replayed a dozen times WITHOUT a synchronisation in between still equals the eager
    forward (nodes of a replayed graph that lose their order show only from the second back-to-back replay on: the
    hipMemsetAsync trap of DESIGN §4.15 — the library's launches are all kernel nodes now)."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF, SparseRatings
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, GraphNCF
    from deeprecommendation_amd.neural_collaborative_filtering.models.mf import MF
    g = torch.Generator().manual_seed(8)

    def replayed(fn):
        with torch.no_grad():
            ref = fn().clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                out = fn()
            for _ in range(12):
                gr.replay()
            torch.cuda.synchronize()
            assert torch.equal(out, ref)

    torch.manual_seed(2)
    bm = BasicNCF(item_dim=700, user_dim=3000, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128]).eval().to(gpu)
    u = torch.randint(0, 3000, (5000,), generator=g).to(gpu)
    i = torch.randint(0, 700, (5000,), generator=g).to(gpu)
    replayed(lambda: bm(u, i))
    mf = MF(item_dim=700, user_dim=3000, item_emb=64, user_emb=64).eval().to(gpu)
    replayed(lambda: mf(u, i))
    # GraphNCF: propagation + readout
    n_items, n_users = 60, 400
    key = torch.unique(torch.randint(0, n_users, (5000,), generator=g) * n_items + torch.randint(0, n_items, (5000,), generator=g))
    uu, ii = key // n_items + n_items, key % n_items
    a = torch.randn(uu.numel(), generator=g)
    graph = GraphData(user2item_edge_index=torch.stack([uu, ii]).to(gpu), item2user_edge_index=torch.stack([ii, uu]).to(gpu),
                      user2item_edge_attr=a.to(gpu), item2user_edge_attr=a.clone().to(gpu), num_items=n_items, num_users=n_users)
    gm = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=2, hetero=True, node_emb=64, mlp_dense_layers=[128]).eval().to(gpu)
    gu = (torch.randint(0, n_users, (1000,), generator=g) + n_items).to(gpu)
    gi = torch.randint(0, n_items, (1000,), generator=g).to(gpu)
    replayed(lambda: gm(graph, gu, gi))
    # AttentionNCF on provider-style shared-row ratings (the cfg 3 path: candidate kernel + grouping, entry-split attention, tail)
    I, F, B, users = 300, 96, 2048, 16
    am = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=128, mlp_dense_layers=[256, 128]).eval().to(gpu)
    rated = (torch.rand(I, F, generator=g) < 0.2).float().to(gpu)
    counts = torch.randint(20, 150, (users,), generator=g)
    rowptr = torch.zeros(users + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    col = torch.cat([torch.randperm(I, generator=g)[:int(c)].sort().values for c in counts]).to(torch.int32)
    val = torch.randint(1, 11, (int(rowptr[-1]),), generator=g).float() * 0.5 - 2.9
    r = SparseRatings(rowptr.to(gpu), col.to(gpu), val.to(gpu), I, pair_row=torch.randint(0, users, (B,), generator=g).to(gpu))
    cand = rated[torch.randint(0, I, (B,), generator=g).to(gpu)].contiguous()
    replayed(lambda: am(cand, rated, r))
