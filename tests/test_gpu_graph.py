"""GPU parity tests for K4/K5 (LightGCN SpMM + layer mean) and GraphNCF against the CPU oracle.

GraphNCF parity is UNPINNED against the real reference (PyG 2.0.4 cannot be imported — see oracle header): the
oracle follows PyG's published semantics and is cross-checked against a dense-adjacency formulation on CPU.
"""
import pytest
import torch

from oracle import ncf_oracle as O
from test_gpu_basic import assert_close

pytestmark = pytest.mark.gpu


def _bipartite(n_items, n_users, n_inter, seed, binary=False, hub=None):
    g = torch.Generator().manual_seed(seed)
    u = torch.randint(0, n_users, (n_inter,), generator=g)
    i = torch.randint(0, n_items, (n_inter,), generator=g)
    if hub is not None:
        i[: n_inter // 2] = hub  # one very popular item: exercises the long-row segment split
    key = torch.unique(u * n_items + i)
    u, i = key // n_items + n_items, key % n_items
    u2i = torch.stack([u, i])
    i2u = torch.stack([i, u])
    if binary:
        return u2i, i2u, None, None
    return u2i, i2u, torch.randn(u.numel(), generator=g), torch.randn(u.numel(), generator=g)


@pytest.mark.parametrize("hetero", [True, False])
@pytest.mark.parametrize("binary", [True, False])
@pytest.mark.parametrize("concat,dot,hidden", [(False, False, [128]), (True, False, [32]), (False, True, None)])
@pytest.mark.parametrize("onehot", [True, False])
def test_graph_ncf_vs_oracle(gpu, hetero, binary, concat, dot, hidden, onehot):
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF, GraphData
    n_items, n_users, D, L = 40, 70, 64, 3
    u2i, i2u, a1, a2 = _bipartite(n_items, n_users, 900, seed=3, binary=binary)
    torch.manual_seed(11)
    if onehot:
        fi, fu, idim, udim = None, None, n_items, n_users
    else:
        fi, fu, idim, udim = torch.rand(n_items, 23), torch.rand(n_users, 17), 23, 17
    m = GraphNCF(item_dim=idim, user_dim=udim, num_gnn_layers=L, hetero=hetero, node_emb=D,
                 mlp_dense_layers=hidden, use_dot_product=dot, concat=concat).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    B = 200
    users = torch.randint(0, n_users, (B,), generator=g) + n_items
    items = torch.randint(0, n_items, (B,), generator=g)
    ref = O.graph_ncf_forward(state, hetero, L, concat, dot,
                              torch.eye(n_items) if onehot else fi, torch.eye(n_users) if onehot else fu,
                              u2i, i2u, a1, a2, users, items)
    graph = GraphData(item_features=fi, user_features=fu, user2item_edge_index=u2i, item2user_edge_index=i2u,
                      user2item_edge_attr=a1, item2user_edge_attr=a2, num_items=n_items, num_users=n_users)
    m.to(gpu)
    with torch.no_grad():
        out = m(graph.to(gpu), users.to(gpu), items.to(gpu), gpu)
        out2 = m(graph, users.to(gpu), items.to(gpu), gpu)  # cached propagation, graph.to() inside
    assert_close(out, ref)
    assert torch.equal(out, out2)


def test_graph_long_rows_are_split_and_deterministic(gpu, monkeypatch):
    """A hub item with thousands of in-edges: split into 64-edge segments + ordered fix-up pass; result equals the
    oracle and is bitwise reproducible."""
    from deeprecommendation_amd.neural_collaborative_filtering.models import gnn_ncf as G
    monkeypatch.setattr(G, "SEGMENT_EDGES", 64)
    n_items, n_users, D = 30, 3000, 128
    u2i, i2u, a1, a2 = _bipartite(n_items, n_users, 6000, seed=9, hub=7)
    torch.manual_seed(2)
    m = G.GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=2, hetero=True, node_emb=D, mlp_dense_layers=[256, 128]).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    users = torch.arange(0, 500) + n_items
    items = torch.arange(0, 500) % n_items
    ref = O.graph_ncf_forward(state, True, 2, False, False, torch.eye(n_items), torch.eye(n_users), u2i, i2u, a1, a2, users, items)

    def run():
        graph = G.GraphData(user2item_edge_index=u2i.to(gpu), item2user_edge_index=i2u.to(gpu), user2item_edge_attr=a1.to(gpu),
                            item2user_edge_attr=a2.to(gpu), num_items=n_items, num_users=n_users)
        prep = G.PreparedGraph(graph, True, seg_len=64)
        assert prep.row_of is not None and int((prep.row_of == 7).sum()) > 10
        graph._prepared[("prep", True)] = prep
        mm = G.GraphNCF(**m.kwargs).eval()
        mm.load_state_dict(state)
        mm.to(gpu)
        with torch.no_grad():
            return mm(graph, users.to(gpu), items.to(gpu), gpu)

    o1, o2 = run(), run()
    # the hub row sums ~3000 signed terms (and the fp32 CPU oracle adds them in another order): the summation-order difference
    # scales with sum|terms|, far above the cancelled results, so the absolute part of the bar is 1e-5 of the largest output here
    assert_close(o1, ref, floor=1.0)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("D", [4, 32, 64, 100, 128, 256])
def test_spmm_kernel_shapes(gpu, D):
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(D)
    N, Nz, E = 257, 300, 5000
    dst = torch.randint(0, N, (E,), generator=g)
    dst[dst == 5] = 6  # row 5 has no edges
    order = torch.argsort(dst, stable=True)
    col = torch.randint(0, Nz, (E,), generator=g)[order].to(torch.int32)
    coef = torch.randn(E, generator=g)
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(dst, minlength=N), 0)
    z = torch.randn(Nz, D, generator=g)
    ref = torch.zeros(N, D, dtype=torch.float64).index_add_(0, dst[order], coef.double()[:, None] * z.double()[col.long()])
    acc = torch.ones(N, D, device=gpu)
    y = native.spmm_csr(rowptr.to(gpu), None, col.to(gpu), coef.to(gpu), z.to(gpu), N, acc_sum=acc)
    assert_close(y, ref.float())
    assert float(y[5].abs().sum()) == 0.0
    assert_close(acc, (ref + 1).float())
    y1 = native.spmm_csr(rowptr.to(gpu), None, col.to(gpu), None, z.to(gpu), N)  # binary graph: coef = 1
    ref1 = torch.zeros(N, D, dtype=torch.float64).index_add_(0, dst[order], z.double()[col.long()])
    assert_close(y1, ref1.float())


def test_degree_and_coef_match_torch(gpu):
    from deeprecommendation_amd import native
    u2i, i2u, a1, a2 = _bipartite(50, 80, 2000, seed=1)
    N = 130
    deg = torch.zeros(N, device=gpu)
    native.degree_accumulate(u2i[1].to(gpu), N, deg)
    native.degree_accumulate(i2u[1].to(gpu), N, deg)
    ref_deg = O.pyg_degree(torch.cat([u2i[1], i2u[1]]), N)
    assert torch.equal(deg.cpu(), ref_deg)
    dis = ref_deg.pow(-0.5)
    dis[dis == float("inf")] = 0
    coef = native.edge_coef(u2i[0].to(gpu), u2i[1].to(gpu), a1.to(gpu), deg)
    ref = a1 * (dis[u2i[0]] * dis[u2i[1]])
    assert torch.allclose(coef.cpu(), ref, rtol=2e-7, atol=0)


@pytest.fixture(scope="module")
def full_graph(gpu):
    """BASELINE configs[3] at FULL size (the graph bench.py's cfg-4 workload builds): 1 M users + 100 k items, 50 M interactions
    (user uniform, item Zipf(1.0)) = 100 M directed edges; built once for the full-size tests of this module."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, PreparedGraph
    g = torch.Generator(device=gpu).manual_seed(11)
    I, U, n = 100_000, 1_000_000, 50_000_000
    ranks = torch.arange(1, I + 1, device=gpu, dtype=torch.float64)
    p = (1.0 / ranks)
    items = torch.multinomial((p / p.sum()).float(), n, replacement=True, generator=g)
    users = torch.randint(0, U, (n,), device=gpu, generator=g) + I
    attr = torch.randn(n, device=gpu, generator=g)
    graph = GraphData(user2item_edge_index=torch.stack([users, items]), item2user_edge_index=torch.stack([items, users]),
                      user2item_edge_attr=attr, item2user_edge_attr=attr.clone(), num_items=I, num_users=U)
    prep = PreparedGraph(graph, hetero=True)
    yield graph, prep, g
    del graph, prep
    torch.cuda.empty_cache()


def test_full_size_lightgat_properties(gpu, full_graph):
    """LightGAT's per-destination edge softmax (gnn_ncf.py:158-176; PyG softmax) at FULL cfg-4 size — 100 M edges, hub items with
    millions of in-edges: the attention weights of every non-empty destination are positive and sum to 1, the edge weight enters as
    a plain factor, a shift of all scores changes nothing, and the whole layer (scores -> softmax -> SpMM) is bitwise repeatable."""
    from deeprecommendation_amd import native
    graph, prep, g = full_graph
    I, U, D = graph.num_items, graph.num_users, 128
    N = I + U
    assert prep.type_pure
    s = torch.randn(N, device=gpu, generator=g) * 2.0
    seg = prep.softmax_segments()
    assert seg is not None                                   # hub rows are split: the segmented softmax is the one LightGAT runs
    alpha = native.edge_softmax_csr(prep.rowptr, prep.col, None, s, segments=seg)
    assert alpha.numel() == prep.col.numel() and float(alpha.min()) >= 0.0
    slow = native.edge_softmax_csr(prep.rowptr, prep.col, None, s)       # one wave per destination (the small-graph form): same function
    assert float((slow - alpha).abs().max()) <= 1e-5 * float(alpha.max())
    del slow
    counts = prep.rowptr[1:] - prep.rowptr[:-1]
    # per-destination sums from a float64 prefix sum (no atomics: a hub item has millions of in-edges on ONE address)
    cs = torch.zeros(alpha.numel() + 1, dtype=torch.float64, device=gpu)
    torch.cumsum(alpha.double(), 0, out=cs[1:])
    sums = cs[prep.rowptr[1:]] - cs[prep.rowptr[:-1]]
    del cs
    nonempty = counts > 0
    err = float((sums[nonempty] - 1.0).abs().max())
    assert err <= 2e-5, f"attention weights of a destination do not sum to 1: {err:.3e}"      # hub rows add 4 M fp32 terms
    if bool((~nonempty).any()):
        assert float(sums[~nonempty].abs().max()) == 0.0
    del sums
    assert torch.equal(alpha, native.edge_softmax_csr(prep.rowptr, prep.col, None, s, segments=seg))   # run to run, bit for bit
    shifted = native.edge_softmax_csr(prep.rowptr, prep.col, None, s + 3.0, segments=seg)
    assert float((shifted - alpha).abs().max()) <= 1e-5 * float(alpha.max())
    del shifted
    coef = native.edge_softmax_csr(prep.rowptr, prep.col, prep.attr, s, segments=seg)
    ref = prep.attr * alpha
    assert float((coef - ref).abs().max()) <= 2e-6 * float(ref.abs().max())
    del ref, alpha
    z = torch.randn(N, D, device=gpu, generator=g)
    y1 = prep.csr.spmm(z, coef=coef).clone()
    y2 = prep.csr.spmm(z, coef=coef)
    assert torch.equal(y1, y2) and bool(torch.isfinite(y1).all())


def test_full_size_lightgcn_properties(gpu, full_graph):
    """BASELINE configs[3] at FULL size (1.1 M nodes, D = 128, 50 M interactions = 100 M directed edges, Zipf items —
    the graph bench.py's cfg-4 workload builds): linearity  SpMM(a z1 + b z2) == a SpMM(z1) + b SpMM(z2)  within fp32 rounding,
    column checksum sum_n y[n] == sum_e coef_e z[col_e], and bitwise run-to-run determinism."""
    from deeprecommendation_amd import native
    graph, prep, g = full_graph
    I, U, D, n = graph.num_items, graph.num_users, 128, 50_000_000
    assert prep.row_of is not None and prep.split == I
    assert prep.col.numel() == 2 * n
    N = I + U
    z1 = torch.randn(N, D, device=gpu, generator=g)
    z2 = torch.randn(N, D, device=gpu, generator=g)
    assert len(prep.csr.levels) >= 3  # the hub items need the multi-level ordered tree
    y1 = prep.csr.spmm(z1).clone()
    y1b = prep.csr.spmm(z1)
    assert torch.equal(y1, y1b)
    # the tree and the simple in-call serial fix-up add the same partials in the same order -> bitwise equal
    part = torch.empty((prep.row_of.numel(), D), device=gpu)
    y1c = native.spmm_csr(prep.segptr, prep.row_of, prep.col, prep.coef, z1, N, partial=part, fixup=True)
    assert_close(y1c, y1.cpu())
    y2 = prep.csr.spmm(z2).clone()
    y12 = prep.csr.spmm(0.5 * z1 - 2.0 * z2)
    lin = 0.5 * y1 - 2.0 * y2
    scale = float(lin.abs().max())
    assert float((y12 - lin).abs().max()) <= 2e-5 * scale
    # checksum over all destinations, fp64 on the device with plain torch ops
    chk = torch.zeros(D, dtype=torch.float64, device=gpu)
    for s in range(0, prep.col.numel(), 4_000_000):
        c = prep.col[s:s + 4_000_000].long()
        chk += (prep.coef[s:s + 4_000_000].double()[:, None] * z1[c].double()).sum(0)
    got = y1.double().sum(0)
    assert float((got - chk).abs().max()) <= 1e-6 * float(chk.abs().max()) + 1e-3


def test_edge_endpoint_out_of_range_raises(gpu):
    """A source or destination node id outside the graph raises IndexError when the graph is prepared (the reference's
    x[edge_index[0]] / scatter raise inside PyG, gnn_ncf.py:74-94); the SpMM kernels themselves never see a bad id."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, PreparedGraph
    u2i, i2u, a1, a2 = _bipartite(20, 30, 200, seed=5)
    N = 50
    for which, row in (("u2i", 0), ("u2i", 1), ("i2u", 0)):
        e1, e2 = u2i.clone(), i2u.clone()
        (e1 if which == "u2i" else e2)[row, 3] = N + 7
        graph = GraphData(user2item_edge_index=e1.to(gpu), item2user_edge_index=e2.to(gpu), user2item_edge_attr=a1.to(gpu),
                          item2user_edge_attr=a2.to(gpu), num_items=20, num_users=30)
        with pytest.raises(IndexError):
            PreparedGraph(graph, hetero=False)


@pytest.mark.parametrize("hetero", [True, False])
@pytest.mark.parametrize("binary", [True, False])
def test_lightgat_vs_oracle(gpu, hetero, binary):
    """LightGAT scoring on the HIP path (edge-softmax kernel + SpMM) against the per-edge oracle restatement
    (PyG softmax semantics; parity unpinned like all of GraphNCF), with a hub item to exercise the split rows."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF, GraphData
    n_items, n_users, D, L = 30, 900, 64, 2
    u2i, i2u, a1, a2 = _bipartite(n_items, n_users, 3000, seed=4, binary=binary, hub=3)
    torch.manual_seed(12)
    m = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=L, hetero=hetero, node_emb=D, mlp_dense_layers=[128],
                 convType="LightGAT").eval()
    with torch.no_grad():  # make the attention non-trivial
        for n_, p_ in m.named_parameters():
            if "AttNet" in n_ and n_.endswith("weight"):
                p_.mul_(8.0)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    users = torch.randint(0, n_users, (300,), generator=g) + n_items
    items = torch.randint(0, n_items, (300,), generator=g)
    ref = O.graph_ncf_forward(state, hetero, L, False, False, torch.eye(n_items), torch.eye(n_users), u2i, i2u, a1, a2,
                              users, items, convType="LightGAT")
    graph = GraphData(user2item_edge_index=u2i, item2user_edge_index=i2u, user2item_edge_attr=a1, item2user_edge_attr=a2,
                      num_items=n_items, num_users=n_users)
    m.to(gpu)
    with torch.no_grad():
        out = m(graph.to(gpu), users.to(gpu), items.to(gpu), gpu)
    assert_close(out, ref)
    # the training (torch) path of the same module agrees with the oracle too
    m.cpu().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    out_t = m(graph, users, items, torch.device("cpu"), mask_targets=False)
    assert_close(out_t, ref)


def test_graph_readout_with_folded_first_layer(gpu):
    """GraphNCF readout through the opt-in folded path (PA / PB both derived from the propagated node table)."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF, GraphData
    n_items, n_users, D = 50, 120, 64
    u2i, i2u, a1, a2 = _bipartite(n_items, n_users, 1500, seed=8)
    torch.manual_seed(3)
    m = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=2, hetero=True, node_emb=D, mlp_dense_layers=[256, 128]).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    users = torch.randint(0, n_users, (400,), generator=g) + n_items
    items = torch.randint(0, n_items, (400,), generator=g)
    ref = O.graph_ncf_forward(state, True, 2, False, False, torch.eye(n_items), torch.eye(n_users), u2i, i2u, a1, a2, users, items)
    graph = GraphData(user2item_edge_index=u2i, item2user_edge_index=i2u, user2item_edge_attr=a1, item2user_edge_attr=a2,
                      num_items=n_items, num_users=n_users)
    m.to(gpu)
    with torch.no_grad():
        plain = m(graph.to(gpu), users.to(gpu), items.to(gpu), gpu)
        m.set_fold_first_layer(True)
        folded = m(graph.to(gpu), users.to(gpu), items.to(gpu), gpu)
    assert any(isinstance(k, tuple) and k[0] == "folded" and v is not None for k, v in m._native_cache.items())
    assert_close(plain, ref)
    assert_close(folded, ref)


@pytest.mark.parametrize("weighted", [True, False])
def test_segmented_edge_softmax_equals_row_form(gpu, weighted):
    """ncf_edge_softmax_segmented (three passes over the SpMM's segments) == ncf_edge_softmax_csr (one wave per destination) on a graph
    with hub rows split into many segments, empty rows and sources out of range; and == the PyG softmax definition in float64."""
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(5)
    N, E = 300, 40_000
    dst = torch.cat([torch.zeros(9000, dtype=torch.int64), torch.full((3000,), 7, dtype=torch.int64), torch.randint(10, N - 5, (E - 12000,), generator=g)])
    src = torch.randint(0, N, (E,), generator=g)
    order = torch.argsort(dst, stable=True)
    dst, src = dst[order], src[order]
    counts = torch.bincount(dst, minlength=N)
    rowptr = torch.zeros(N + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    attr = torch.randn(E, generator=g) if weighted else None
    s = torch.randn(N, generator=g) * 3
    col = src.to(torch.int32)
    col[5] = -1
    col[20000] = N + 3
    csr = native.SegmentedCSR(rowptr.to(gpu), col.to(gpu), None, seg_len=512)
    segptr, row_of, _ = csr.levels[0]
    assert row_of is not None
    seg_first = torch.searchsorted(row_of.to(torch.int64), torch.arange(N + 1, device=gpu))
    a_seg = native.edge_softmax_csr(rowptr.to(gpu), col.to(gpu), None if attr is None else attr.to(gpu), s.to(gpu), segments=(segptr, row_of, seg_first))
    a_row = native.edge_softmax_csr(rowptr.to(gpu), col.to(gpu), None if attr is None else attr.to(gpu), s.to(gpu))
    assert float((a_seg - a_row).abs().max()) <= 1e-6 * float(a_row.abs().max())
    okc = (col >= 0) & (col < N)
    sc = torch.where(okc, s[col.long().clamp(0, N - 1)].double(), torch.full((E,), -float("inf"), dtype=torch.float64))
    mx = torch.full((N,), -float("inf"), dtype=torch.float64).scatter_reduce(0, dst, sc, reduce="amax", include_self=True)
    ex = torch.where(okc, torch.exp(sc - mx[dst]), torch.zeros(E, dtype=torch.float64))
    den = torch.zeros(N, dtype=torch.float64).index_add_(0, dst, ex)
    ref = ex / (den[dst] + 1e-16)
    if attr is not None:
        ref = ref * attr.double()
    assert_close(a_seg, ref.float())
