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
    assert np.array_equal(outs[True], outs[False])  # CSR ratings == dense user_matrix
    # oracle on the first batch, dense reference formulation
    prov = SparseDynamicProvider(item_ids, feats, np.arange(nu), rated, ratings, means, sparse=False)
    batch = [tuple(test.iloc[k][["userId", "movieId", "rating"]]) for k in range(64)]
    _, _, cand, rated_f, um, _ = prov.collate_interacted_items([(int(u), int(i), r) for u, i, r in batch], False)
    ref = O.attention_ncf_forward(state, cand, rated_f, um)
    assert_close(torch.from_numpy(outs[True][:64]).float().view(-1, 1), ref)
