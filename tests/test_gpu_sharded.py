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


def test_partitioned_lightgat_world1(gpu):
    """LightGAT through PartitionedLightGCN's destination-block form at world size 1 (HIP kernels, scores packed beside Z):
    equal to the model's own propagation."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, GraphNCF
    from deeprecommendation_amd.sharded import PartitionedLightGCN
    from test_gpu_basic import assert_close
    g = torch.Generator().manual_seed(2)
    n_items, n_users, D = 60, 400, 64
    key = torch.unique(torch.randint(0, n_users, (5000,), generator=g) * n_items + torch.randint(0, n_items, (5000,), generator=g))
    u, i = key // n_items + n_items, key % n_items
    a = torch.randn(u.numel(), generator=g)
    for hetero in (True, False):
        graph = GraphData(user2item_edge_index=torch.stack([u, i]).to(gpu), item2user_edge_index=torch.stack([i, u]).to(gpu),
                          user2item_edge_attr=a.to(gpu), item2user_edge_attr=a.clone().to(gpu), num_items=n_items, num_users=n_users)
        torch.manual_seed(4)
        model = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=3, hetero=hetero, node_emb=D, mlp_dense_layers=[128],
                         convType="LightGAT").eval().to(gpu)
        with torch.no_grad():
            ref = model.propagate_all(graph)
            part = PartitionedLightGCN(model, graph, mode="dst")
            out = part.propagate(model._node_table0(graph))
        assert_close(out, ref, rtol=2e-6)          # same kernels; the strided Z of the packed buffer may reorder nothing: fp32 rounding at most


@pytest.mark.parametrize("world,cap,B", [(8, 1200, 8192), (2, 4096, 4097), (3, 40, 1000), (1, 16, 100), (64, 3, 5000)])
def test_bucket_ids_kernel(gpu, world, cap, B):
    """ncf_bucket_ids against its definition (include/ncf_abi.h): every kept id sits in its owner's bucket as a local row,
    slot[p] points at it, counts are exact, unused slots are 0, ids over capacity / out of range are dropped and flagged."""
    from deeprecommendation_amd import native
    total = 100_003
    rpr = (total + world - 1) // world
    g = torch.Generator().manual_seed(world * 1000 + B)
    idx = torch.randint(0, total, (B,), generator=g)
    idx[5], idx[77] = -3, total + 9          # two ids outside the table
    d = idx.to(gpu)
    send = torch.full((world * cap,), -7, dtype=torch.int64, device=gpu)
    slot = torch.full((B,), -7, dtype=torch.int64, device=gpu)
    counts = torch.full((world,), -7, dtype=torch.int32, device=gpu)
    overflow = torch.zeros(1, dtype=torch.int32, device=gpu)
    native.bucket_ids(d, rpr, total, world, cap, send, slot, counts, overflow)
    with pytest.raises(IndexError):
        native.check_oob(gpu)
    send, slot, counts = send.cpu(), slot.cpu(), counts.cpu()
    ok = (idx >= 0) & (idx < total)
    owner = torch.where(ok, idx // rpr, torch.zeros_like(idx))
    expect = torch.bincount(owner[ok], minlength=world)
    assert torch.equal(counts.long(), expect)
    assert int(overflow.item()) == int(bool((expect > cap).any()))
    kept = slot >= 0
    assert not bool(kept[~ok].any())
    assert torch.equal(slot[kept] // cap, owner[kept])                       # the right bucket
    assert torch.equal(send[slot[kept]], (idx - owner * rpr)[kept])          # holding the right local row
    assert torch.unique(slot[kept]).numel() == int(kept.sum())               # one pair per slot
    assert torch.equal(torch.bincount(owner[kept], minlength=world), torch.minimum(expect, torch.tensor(cap)))
    used = torch.zeros(world * cap, dtype=torch.bool)
    used[slot[kept]] = True
    assert bool((send[~used] == 0).all())                                    # padding names local row 0
    for o in range(world):                                                   # a bucket is filled from its start
        n = min(int(expect[o]), cap)
        assert bool(used[o * cap:o * cap + n].all()) and not bool(used[o * cap + n:(o + 1) * cap].any())


@pytest.mark.parametrize("world,cap,B,zipf", [(8, 1200, 8192, False), (8, 700, 8192, True), (2, 4096, 4097, False), (3, 40, 1000, True),
                                              (1, 16, 100, False), (64, 3, 5000, False), (8, 16, 0, False)])
def test_bucket_ids_dedup_kernel_and_bucket_gather(gpu, world, cap, B, zipf):
    """ncf_bucket_ids_dedup against its definition (include/ncf_abi.h): every DISTINCT valid id is listed once in its owner's bucket
    (as a local row, buckets filled from their start, header = count), every pair of that id points at that slot, counts are the
    distinct counts, ids over capacity / out of range are dropped and flagged; and ncf_gather_buckets returns exactly the rows the
    slots name (padding untouched)."""
    from deeprecommendation_amd import native
    total = 100_003
    rpr = (total + world - 1) // world
    g = torch.Generator().manual_seed(world * 1000 + B + 7)
    if zipf:       # heavy repeats
        idx = (torch.rand(B, generator=g) ** 6 * total).long().clamp(0, total - 1)
    else:
        idx = torch.randint(0, total, (B,), generator=g)
    if B > 100:
        idx[5], idx[77] = -3, total + 9          # two ids outside the table
        idx[10:20] = idx[9]                      # a run of repeats
    d = idx.to(gpu)
    H = native.bucket_dedup_table_slots(B)
    send = torch.full((world * (cap + 1),), -7, dtype=torch.int64, device=gpu)
    slot = torch.full((max(B, 1),), -7, dtype=torch.int64, device=gpu)
    counts = torch.full((world,), -7, dtype=torch.int32, device=gpu)
    overflow = torch.zeros(1, dtype=torch.int32, device=gpu)
    hk, hv = torch.empty(H, dtype=torch.int64, device=gpu), torch.empty(H, dtype=torch.int64, device=gpu)
    native.bucket_ids_dedup(d, rpr, total, world, cap, send, slot, counts, overflow, hk, hv)
    if B > 100:
        with pytest.raises(IndexError):
            native.check_oob(gpu)
    else:
        native.check_oob(gpu)
    send_c, slot_c, counts_c = send.cpu().view(world, cap + 1), slot.cpu()[:B], counts.cpu()
    ok = (idx >= 0) & (idx < total)
    uniq = torch.unique(idx[ok])
    expect = torch.bincount(uniq // rpr, minlength=world)
    assert torch.equal(counts_c.long(), expect)
    assert int(overflow.item()) == int(bool((expect > cap).any()))
    assert torch.equal(send_c[:, 0], torch.minimum(expect, torch.tensor(cap)))          # bucket headers
    kept = slot_c >= 0
    assert not bool(kept[~ok].any())
    owner = torch.where(ok, idx // rpr, torch.zeros_like(idx))
    assert torch.equal(slot_c[kept] // cap, owner[kept])                                   # the right bucket
    k = slot_c[kept] % cap
    assert bool((k < send_c[owner[kept], 0]).all())                                        # inside the bucket's filled prefix
    assert torch.equal(send_c[owner[kept], 1 + k], (idx - owner * rpr)[kept])              # holding the right local row
    # one slot per distinct id: pairs with equal ids share it, distinct ids never do
    assert torch.unique(slot_c[kept]).numel() == torch.unique(idx[kept]).numel()
    assert torch.unique(torch.stack([slot_c[kept], idx[kept]]), dim=1).shape[1] == torch.unique(slot_c[kept]).numel()
    if not bool((expect > cap).any()):
        assert bool(kept[ok].all())
    # the owner's side (all buckets "received" by one table here): rows of exactly the listed ids
    tab = torch.randn(rpr, 128, generator=g).to(torch.bfloat16).to(gpu)
    out = torch.full((world * cap, 128), 7.0, dtype=torch.bfloat16, device=gpu)
    native.gather_buckets(tab, send, world, cap, out)
    native.check_oob(gpu)
    out_c, tab_c = out.cpu(), tab.cpu()
    for o in range(world):
        n = int(send_c[o, 0])
        assert torch.equal(out_c[o * cap:o * cap + n], tab_c[send_c[o, 1:1 + n]])
        assert bool((out_c[o * cap + n:(o + 1) * cap] == 7.0).all())                       # padding: neither gathered nor written
    if B:
        assert torch.equal(out_c[slot_c[kept]], tab_c[(idx - owner * rpr)[kept]])           # what a pair reads back is its row


def _two_rank_worker(rank, world, store, exchange, replicate):
    """Both ranks on cuda:0, gloo transport (device tensors staged through the host): the HIP local compute, the device
    bucketing and the two-stream pipeline of ShardedBasicNCF, checked bit for bit against the unsharded fused kernel."""
    import torch.distributed as dist
    from deeprecommendation_amd import native
    from deeprecommendation_amd.sharded import RowShardedTable, ShardedBasicNCF
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        g = torch.Generator().manual_seed(0)
        U, I, E = 40_001, 5_003, 128
        tu = (torch.randn(U, E, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        ti = (torch.randn(I, E, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        dims = [2 * E, 256, 128, 1]
        ws = [(torch.randn(dims[k + 1], dims[k], generator=g) / dims[k] ** 0.5).to(dev) for k in range(3)]
        bs = [(torch.randn(dims[k + 1], generator=g) * 0.1).to(dev) for k in range(3)]
        packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
        ulo, uhi = RowShardedTable.shard_bounds(U, world, rank)
        ilo, ihi = RowShardedTable.shard_bounds(I, world, rank)
        model = ShardedBasicNCF(tu[ulo:uhi].contiguous(), U, ti if replicate else ti[ilo:ihi].contiguous(), I, ws, bs,
                                replicate_items=replicate, dtype=torch.bfloat16, exchange=exchange)
        gb = torch.Generator().manual_seed(10 + rank)
        B, n = 6000, 5
        ups = [torch.randint(0, U, (B,), generator=gb).to(dev) for _ in range(n)]
        ips = [torch.randint(0, I, (B,), generator=gb).to(dev) for _ in range(n)]
        outs = []
        t = model.submit(ups[0], ips[0])
        for k in range(n):
            nxt = model.submit(ups[k + 1], ips[k + 1]) if k + 1 < n else None
            outs.append(model.score(t))
            t = nxt
        model.check()
        for k in range(n):
            ref = native.score_fused(tu, ups[k], ti, ips[k], packed)
            assert torch.equal(outs[k], ref), f"rank {rank} batch {k} ({exchange})"
        if exchange == "bounded":
            assert model._xstream is not None and model.users.cap % 64 == 0
            st = model.wire_stats()["users"]
            assert st["lookups"] == n and st["ids_asked"] == n * B and 0 < st["ids_listed"] <= st["ids_asked"]
            assert 0.0 < st["padding_fraction"] < 0.5 and st["remote_rows_needed"] < st["ids_listed"]
            # overflow -> every rank raises -> grow on every rank -> repeat -> clean check (no stale flags left behind)
            from deeprecommendation_amd.sharded import ExchangeOverflow
            model.users.set_capacity(64)
            model(ups[0], ips[0])
            with pytest.raises(ExchangeOverflow):
                model.check()
            model.check()
            model.grow_capacity(80.0)
            out = model(ups[0], ips[0])
            model.check()
            assert torch.equal(out, native.score_fused(tu, ups[0], ti, ips[0], packed))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("exchange,replicate", [("bounded", True), ("bounded", False), ("unique", True)])
def test_sharded_basic_ncf_two_ranks_one_gpu(gpu, exchange, replicate):
    import os
    import tempfile
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_two_rank_worker, args=(2, os.path.join(d, "store"), exchange, replicate), nprocs=2, join=True)


def _two_rank_graph_worker(rank, world, store, mode):
    import torch.distributed as dist
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, GraphNCF
    from deeprecommendation_amd.sharded import PartitionedLightGCN
    from test_gpu_basic import assert_close
    dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        g = torch.Generator().manual_seed(2)
        n_items, n_users, D = 300, 4000, 128
        it = (torch.rand(60_000, generator=g) ** 3 * n_items).long()          # skewed items: blocks are edge-balanced
        key = torch.unique(torch.randint(0, n_users, (60_000,), generator=g) * n_items + it)
        u, i = key // n_items + n_items, key % n_items
        a = torch.randn(u.numel(), generator=g)
        graph = GraphData(user2item_edge_index=torch.stack([u, i]).to(dev), item2user_edge_index=torch.stack([i, u]).to(dev),
                          user2item_edge_attr=a.to(dev), item2user_edge_attr=a.clone().to(dev), num_items=n_items, num_users=n_users)
        torch.manual_seed(4)
        model = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=3, hetero=True, node_emb=D, mlp_dense_layers=[128]).eval().to(dev)
        with torch.no_grad():
            ref = model.propagate_all(graph)
            part = PartitionedLightGCN(model, graph, mode=mode)
            out = part.propagate(model._node_table0(graph))
        if mode == "dst":
            assert part.bounds[1] not in (0, n_items + n_users)
            assert torch.equal(out, ref)          # every row is summed by the same kernel in the same order
        else:
            assert_close(out, ref.cpu())          # partial sums of the two ranks are added by the all-reduce
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["dst", "edge"])
def test_partitioned_lightgcn_two_ranks_one_gpu(gpu, mode):
    import os
    import tempfile
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_two_rank_graph_worker, args=(2, os.path.join(d, "store"), mode), nprocs=2, join=True)


class _LoopbackComm:
    """Stand-in for sharded.Comm that moves bytes on the device only (this rank plays every peer): the results are not a real
    exchange, the ORDER OF OPERATIONS on the streams is — what the no-host-synchronisation test needs."""

    def __init__(self, world, rank):
        self.world, self.rank, self.on, self.backend, self.group = world, rank, True, "loopback", None

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        assert out_splits is None and in_splits is None        # the bounded exchange never needs sizes
        out.view(self.world, -1).copy_(inp.view(self.world, -1)[self.rank].unsqueeze(0).expand(self.world, -1))
        return out

    def all_reduce(self, t, op=None):
        return t


def test_bounded_exchange_has_no_host_synchronisation(gpu):
    """SURVEY §8e / VERDICT item 5: once the capacity is agreed, a pipelined step of the row-sharded scoring path (device-side owner
    bucketing, equal-split exchanges on the second stream, fused scoring on the first) must not synchronise the host at all.
    torch's sync debug mode raises on ANY synchronising call (.item(), .tolist(), .cpu(), nonzero, ...)."""
    from deeprecommendation_amd.sharded import RowShardedTable, ShardedBasicNCF
    world, rank = 4, 1
    g = torch.Generator().manual_seed(0)
    U, I, E, B = 40_000, 5_000, 128, 8192
    ulo, uhi = RowShardedTable.shard_bounds(U, world, rank)
    tu = (torch.randn(uhi - ulo, E, generator=g) * 0.05).to(torch.bfloat16).to(gpu)
    ti = (torch.randn(I, E, generator=g) * 0.05).to(torch.bfloat16).to(gpu)
    dims = [2 * E, 256, 128, 1]
    ws = [(torch.randn(dims[k + 1], dims[k], generator=g) / dims[k] ** 0.5).to(gpu) for k in range(3)]
    bs = [(torch.randn(dims[k + 1], generator=g) * 0.1).to(gpu) for k in range(3)]
    model = ShardedBasicNCF(tu, U, ti, I, ws, bs, replicate_items=True, dtype=torch.bfloat16, exchange="bounded", comm=_LoopbackComm(world, rank))
    assert model._xstream is not None
    ups = [torch.randint(0, U, (B,), generator=g).to(gpu) for _ in range(6)]
    ips = [torch.randint(0, I, (B,), generator=g).to(gpu) for _ in range(6)]
    model.negotiate_capacity(ups[0], ips[0])               # the one host read
    model(ups[0], ips[0])                                  # buffers allocated
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        t = model.submit(ups[0], ips[0])
        outs = []
        for k in range(6):
            nxt = model.submit(ups[(k + 1) % 6], ips[(k + 1) % 6])
            outs.append(model.score(t))
            t = nxt
        model.score(t)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert all(torch.isfinite(o).all() for o in outs)
    model.check()                                          # (this one synchronises: end of the pass)


def _rccl_world1_worker(rank, world, store):
    """RCCL itself (backend "nccl"), world size 1 — all a one-GPU box can give: the collectives of Comm on DEVICE tensors with the
    dtypes and shapes the exchange ships (int32 id buckets with equal splits, bf16 / fp32 rows, the MAX of the flags, the fp32 SUM
    of a node table, a barrier).  At world size 1 every collective is the identity: what is checked is that RCCL takes the calls."""
    import torch.distributed as dist
    from deeprecommendation_amd.sharded import Comm
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", init_method=f"file://{store}", rank=rank, world_size=world, device_id=dev)
    try:
        comm = Comm()
        assert comm.backend == "nccl" and comm.world == 1 and not comm._staged(torch.empty(1, device=dev))
        ids = torch.arange(4 * 65, dtype=torch.int32, device=dev)
        got = comm.all_to_all(torch.empty_like(ids), ids)
        rows = torch.randn(4 * 64, 128, device=dev).to(torch.bfloat16)
        got_rows = comm.all_to_all(torch.empty_like(rows), rows)
        rows32 = torch.randn(300, 64, device=dev)
        got32 = comm.all_to_all(torch.empty_like(rows32), rows32, [300], [300])
        side = torch.cuda.Stream(device=dev)                     # the exchange runs on a second stream
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            got_side = comm.all_to_all(torch.empty_like(ids), ids)
        torch.cuda.current_stream(dev).wait_stream(side)
        flags = torch.tensor([0, 1, 0], dtype=torch.int32, device=dev)
        comm.all_reduce(flags, dist.ReduceOp.MAX)
        table = torch.randn(1000, 128, device=dev)
        ref = table.clone()
        comm.all_reduce(table)
        dist.barrier(device_ids=[0])
        torch.cuda.synchronize()
        assert torch.equal(got, ids) and torch.equal(got_rows, rows) and torch.equal(got32, rows32) and torch.equal(got_side, ids)
        assert flags.tolist() == [0, 1, 0] and torch.equal(table, ref)
        assert comm.exchange_blocks(table, [0, 1000]) is table
    finally:
        dist.destroy_process_group()


def test_comm_collectives_over_rccl_world1(gpu):
    import os
    import tempfile
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_rccl_world1_worker, args=(1, os.path.join(d, "store")), nprocs=1, join=True)
