"""GPU tests of the sharded / partitioned classes at world size 1 (HIP local compute, no process group):
they must reduce to the single-GPU product path."""
import pytest
import torch


pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sharded_basic_ncf_world1(gpu, dtype):
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.neural_collaborative_filtering.util import mlp_linears
    from deeprecommendation_amd.sharded import ShardedBasicNCF
    torch.manual_seed(3)
    U, I, E = 5000, 900, 128
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=[256, 128]).eval().to(gpu)
    m.set_scoring_dtype(dtype)
    g = torch.Generator().manual_seed(1)
    u = torch.randint(0, U, (4000,), generator=g).to(gpu)
    i = torch.randint(0, I, (4000,), generator=g).to(gpu)
    u[:500] = 7  # duplicates: de-duplicated before the (here trivial) exchange
    with torch.no_grad():
        ref = m(u, i)
        lins = mlp_linears(m.MLP)
        sh = ShardedBasicNCF(m._table("user", m.user_embeddings[0]), U, m._table("item", m.item_embeddings[0]), I,
                             [l.weight for l in lins], [l.bias for l in lins])
        out = sh(u, i)
        sh2 = ShardedBasicNCF(m._table("user", m.user_embeddings[0]), U, m._table("item", m.item_embeddings[0]), I,
                              [l.weight for l in lins], [l.bias for l in lins], replicate_items=True)
        out2 = sh2(u, i)
    assert torch.equal(out, ref) and torch.equal(out2, ref)  # same rows, same kernel -> bit-identical


@pytest.mark.parametrize("mode", ["dst", "edge"])
def test_partitioned_lightgcn_world1(gpu, mode):
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, GraphNCF
    from deeprecommendation_amd.sharded import PartitionedLightGCN
    g = torch.Generator().manual_seed(2)
    n_items, n_users, D = 60, 400, 64
    key = torch.unique(torch.randint(0, n_users, (5000,), generator=g) * n_items + torch.randint(0, n_items, (5000,), generator=g))
    u, i = key // n_items + n_items, key % n_items
    a = torch.randn(u.numel(), generator=g)
    graph = GraphData(user2item_edge_index=torch.stack([u, i]).to(gpu), item2user_edge_index=torch.stack([i, u]).to(gpu),
                      user2item_edge_attr=a.to(gpu), item2user_edge_attr=a.clone().to(gpu), num_items=n_items, num_users=n_users)
    torch.manual_seed(4)
    model = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=3, hetero=True, node_emb=D, mlp_dense_layers=[128]).eval().to(gpu)
    with torch.no_grad():
        ref = model.propagate_all(graph)
        part = PartitionedLightGCN(model, graph, mode=mode)
        out = part.propagate(model._node_table0(graph))
    assert torch.equal(out, ref)  # world 1: same CSR, same kernels
