"""Secondary workloads of BASELINE.json (bench.py's `other_configs`, or alone with `--workload`):

  cfg3   AttentionNCF forward: 100k-item catalogue, 256 rated items per user, IE = UE = 64, A = 128, F = 2094, B = 4096
         (user, candidate) pairs per step; catalogue projections precomputed once (eval-time precompute, SURVEY §8d cfg 3).
         Unit: pairs/s.  Replicas only: single-GPU entry.
  cfg4   GraphNCF 3-layer LightGCN propagation: 1 M users + 100 k items, 50 M interactions = 100 M directed edges
         (items ~ Zipf(1.0)), D = 128, hetero, mean readout.  Unit: directed edges/s (3 layers x 100 M per step).
         N > 1: PartitionedLightGCN, destination blocks + all-gather and edge split + all-reduce side by side.
  cfg5   BasicNCF 100 M users x 10 M items, emb 128 bf16, MLP 256-256-128-1.  N = 1: both tables on the one GPU (no
         exchange).  N > 1: user table row-sharded, all-to-all exchange over RCCL; pipelined / serial / unique side by side.
  train2 training step on the cfg-2 shape.
Every function returns the JSON object (rank 0; None elsewhere) with the same contract as bench.py's line.
"""
import os
import time

import torch


def _time_steps(step, warmup, steps, ctx=None):
    """(wall seconds of exactly `steps` steps, warm-up steps actually run).  Untimed: the warm-up steps, continued until
    ~30 ms have been enqueued so that the clock has settled on this workload's kernels (short steps only; DVFS takes
    milliseconds).  Barrier + synchronize on both sides; the caller takes the max over ranks."""
    t_w = time.perf_counter()
    k = 0
    # with several ranks the number of warm-up steps must not depend on a rank's own clock: every step of a sharded form is a
    # collective, and ranks that leave the loop after different counts deadlock
    fixed = None
    if ctx is not None and getattr(ctx, "world", 1) > 1:
        rehearsal = getattr(ctx, "dist", None) is not None and ctx.dist.get_backend() != "nccl"    # gloo stages tensors through the host
        fixed = max(warmup, 3 if rehearsal else 50)
    while (k < fixed) if fixed is not None else (k < warmup or (time.perf_counter() - t_w < 0.03 and k < 2000)):
        step(k)
        k += 1
    if ctx is not None:
        ctx.barrier()
    else:
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for j in range(steps):
        step(j)
    if ctx is not None:
        ctx.barrier()
    else:
        torch.cuda.synchronize()
    return time.perf_counter() - t0, k


def _cpu_median_s(fn, reps, warm=2):
    """Median wall time of fn() on the host cores (torch CPU threads = the process's CPU share)."""
    import bench
    torch.set_num_threads(bench.host_cores())
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], bench.host_cores()


# ---------------------------------------------------------------------------------------------------- cfg 4
def _cfg4_graph(device):
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData
    I, U, n = 100_000, 1_000_000, 50_000_000
    g = torch.Generator(device=device).manual_seed(11)     # the same graph on every rank
    p = 1.0 / torch.arange(1, I + 1, device=device, dtype=torch.float64)
    items = torch.multinomial((p / p.sum()).float(), n, replacement=True, generator=g)
    users = torch.randint(0, U, (n,), device=device, generator=g) + I
    attr = torch.randint(1, 11, (n,), device=device, generator=g).float() * 0.5 - 3.0
    return GraphData(user2item_edge_index=torch.stack([users, items]), item2user_edge_index=torch.stack([items, users]),
                     user2item_edge_attr=attr, item2user_edge_attr=attr.clone(), num_items=I, num_users=U)


def _cfg4_model(device, L=3, D=128, conv="LightGCN"):
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphNCF
    I, U = 100_000, 1_000_000
    torch.manual_seed(1234)
    with torch.device("meta"):
        model = GraphNCF(item_dim=I, user_dim=U, num_gnn_layers=L, hetero=True, node_emb=D, mlp_dense_layers=[256, 128], convType=conv)
    model = model.to_empty(device=device).eval()
    gg = torch.Generator(device=device).manual_seed(1234)  # the same weights on every rank
    with torch.no_grad():
        for prm in model.parameters():
            if prm.dim() == 2:
                bound = (6.0 / (prm.shape[0] + prm.shape[1])) ** 0.5 if prm.shape[0] == prm.shape[1] else 0.05
                prm.uniform_(-bound, bound, generator=gg)
            else:
                prm.uniform_(-0.05, 0.05, generator=gg)
    return model


def _cfg4_cpu_baseline(model, D):
    # CPU oracle, reference formulation (per-EDGE Linear + scatter-add, gnn_ncf.py:39-94) on a graph 100x smaller:
    # 10 000 users x 1 000 items, 500 000 interactions (the full graph's per-edge messages alone are 51 GB)
    from oracle import ncf_oracle as O
    gc = torch.Generator().manual_seed(11)
    Ic, Uc, nc = 1_000, 10_000, 500_000
    pc_ = 1.0 / torch.arange(1, Ic + 1, dtype=torch.float64)
    it = torch.multinomial((pc_ / pc_.sum()).float(), nc, replacement=True, generator=gc)
    us_ = torch.randint(0, Uc, (nc,), generator=gc) + Ic
    at = torch.randint(1, 11, (nc,), generator=gc).float() * 0.5 - 3.0
    xc = torch.randn(Ic + Uc, D, generator=gc) * 0.05
    cs = {k: v.detach().cpu() for k, v in model.gnn_convs[0].state_dict().items()}
    sec, cores = _cpu_median_s(lambda: O.lightgcn_conv(xc, cs, True, torch.stack([us_, it]), torch.stack([it, us_]), at, at), reps=20, warm=5)
    return {"value": 2 * nc / sec, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"reference formulation (per-edge Linear + index_add) of ONE layer on {Uc} users x {Ic} items, "
                      f"{2 * nc} directed edges, D={D}, median of 20 after 5 warm-ups ({sec * 1e3:.0f} ms each), torch CPU fp32"}


def run_cfg4(args, ctx):
    import bench
    if ctx.world > 1:
        return run_cfg4_partitioned(args, ctx)
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import PreparedGraph
    device = ctx.device
    I, U, D, L = 100_000, 1_000_000, 128, 3
    graph = _cfg4_graph(device)
    model = _cfg4_model(device, L, D)
    t0 = time.perf_counter()
    prep = PreparedGraph(graph, True)
    graph._prepared[("prep", True)] = prep
    torch.cuda.synchronize()
    prep_s = time.perf_counter() - t0
    E = int(prep.col.numel())
    N = I + U

    def step(k):
        model._native_cache = {}  # force a full re-propagation (the product caches it across batches)
        model._native_ver = None
        with torch.no_grad():
            return model.propagate_all(graph)

    wall, warm = _time_steps(step, args.warmup, args.steps)
    # dominant kernel: one SpMM layer (edge pass + the ordered partial-sum tree of the hub rows)
    conv = model.gnn_convs[0]
    x = model._node_table0(graph)
    z = conv.hoisted(x, prep)
    y = torch.empty((N, D), device=device)
    kt = bench.kernel_time("ncf::spmm_seg_kernel", "cfg4", lambda: prep.csr.spmm(z, y=y), reps=20, settle=5)
    # one SpMM layer = the edge pass + the (small) ordered partial-sum tree of the hub rows: live timings cover the whole layer,
    # rocprof's average is per spmm_seg_kernel LAUNCH (edge pass and tree levels mixed), so the layer's time stays live here
    us = kt["us_back_to_back"]
    gemm_us = bench.back_to_back_us(lambda: conv.hoisted(x, prep), reps=20, settle=5)
    bytes_per_edge = D * 4 + 4 + 4
    alg = E * bytes_per_edge + N * D * 4  # + the output rows written once
    gbs = alg / (us * 1e-6) / 1e9
    dig = bench.profile_digest("ncf::spmm_layer", "cfg4") or {}   # tools/summarize_profile.py: the layer's launches summed
    traffic = dig.get("traffic")
    hbm_gbs_live = None if traffic is None else traffic / (us * 1e-6) / 1e9
    # the counter bytes come from the committed profile: divide them by THAT profile's own layer time (same run, same clock);
    # the live layer time gives frac_live beside it
    prof_us = dig.get("rocprof_avg_us")
    hbm_gbs = hbm_gbs_live if (traffic is None or not prof_us) else traffic / (prof_us * 1e-6) / 1e9
    line = {"metric": "LightGCN propagated directed edges/sec", "value": L * E * args.steps / wall, "unit": "edges/s", "n_gpus": 1,
            "steps": args.steps, "warmup": warm, "warmup_requested": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg4: GraphNCF {L}-layer LightGCN, {U} users x {I} items, {E} directed edges (Zipf items), D={D}, hetero, mean",
                       "graph_prep_s": prep_s, "segments": int(prep.segptr.numel() - 1)},
            "roofline": {"kernel": f"spmm_seg_kernel<32> x{len(prep.csr.levels)} levels (edge pass + ordered partial tree), one layer",
                         "bound": "hbm",
                         # frac is the COUNTER figure when the PMC profile of this context is committed: bytes that actually crossed
                         # the L2's memory side (FETCH x2 + WRITE) per layer over the layer's time; the algorithmic rate beside it
                         "achieved": hbm_gbs if hbm_gbs is not None else gbs, "peak": bench.PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": (hbm_gbs if hbm_gbs is not None else gbs) / bench.PEAK_HBM_GBS,
                         "frac_basis": ("PMC bytes per layer / the same profile's layer time" if prof_us else "PMC bytes per layer / live layer time") if hbm_gbs is not None else "ALGORITHMIC bytes / live layer time (no committed PMC profile)",
                         "frac_live": None if hbm_gbs_live is None else hbm_gbs_live / bench.PEAK_HBM_GBS,
                         "traffic": traffic, "us_per_launch": us, "us_isolated_events": kt["us_isolated_events"],
                         "rocprof_layer_us": dig.get("rocprof_avg_us"), "profile": dig.get("profile"),
                         "algorithmic_bytes_per_edge": bytes_per_edge, "algorithmic_bytes_per_launch": alg,
                         "algorithmic_GBps": gbs, "algorithmic_frac_of_hbm_peak": gbs / bench.PEAK_HBM_GBS,
                         "hoisted_gemm_us": gemm_us,
                         "note": "the algorithmic rate (520 B per edge) can exceed the HBM peak: the 51 MB of item rows — half of all source "
                                 "reads — are served by the Infinity Cache and never reach HBM; what did is the counter figure"}}
    if not getattr(args, "no_variants", False):
        # LightGAT (gnn_ncf.py:97-177) on the same graph: per layer one GEMV for the source scores, the per-destination edge softmax
        # (two passes over 4-byte scalars per row) and the same SpMM with the softmax as its coefficients
        try:
            from deeprecommendation_amd import native
            gat = _cfg4_model(device, L, D, conv="LightGAT")

            def gstep(k):
                gat._native_cache = {}
                gat._native_ver = None
                with torch.no_grad():
                    return gat.propagate_all(graph)

            wall_g, warm_g = _time_steps(gstep, args.warmup, args.steps)
            sv = torch.randn(N, device=device)
            sm_us = bench.back_to_back_us(lambda: native.edge_softmax_csr(prep.rowptr, prep.col, prep.attr, sv, segments=prep.softmax_segments()), reps=10, settle=3)
            sm_bytes = E * (4 + 4 + 4 + 4 + 4) + N * 16       # col + attr read, gathered score twice, coefficient written
            line["variants"] = {"lightgat": {"edges_per_s": L * E * args.steps / wall_g, "ms_per_step": wall_g / args.steps * 1e3,
                                             "vs_lightgcn": wall / wall_g, "edge_softmax_us_per_layer": sm_us,
                                             "edge_softmax_GBps_algorithmic": sm_bytes / (sm_us * 1e-6) / 1e9,
                                             "edge_softmax_frac_of_hbm_peak": sm_bytes / (sm_us * 1e-6) / 1e9 / bench.PEAK_HBM_GBS,
                                             "note": "same SpMM kernel with per-layer coefficients; the softmax passes are the added cost"}}
            del gat, sv
        except Exception as exc:  # noqa: BLE001
            line["variants"] = {"lightgat": {"error": f"{type(exc).__name__}: {exc}"}}
    if not getattr(args, "no_cpu_baseline", False):
        line["cpu_baseline"] = _cfg4_cpu_baseline(model, D)
    return line


def run_cfg4_partitioned(args, ctx):
    """cfg 4 at N > 1: every rank builds the same graph, keeps its partition, and one step = the full
    3-layer propagation including the collectives.  Both partitionings are timed; value = the faster one."""
    from deeprecommendation_amd.sharded import PartitionedLightGCN
    device = ctx.device
    I, U, D, L = 100_000, 1_000_000, 128, 3
    graph = _cfg4_graph(device)
    model = _cfg4_model(device, L, D)
    E = 2 * int(graph.user2item_edge_index.shape[1])
    x0 = model._node_table0(graph)
    res = {}
    for mode in ("dst", "edge"):
        t0 = time.perf_counter()
        part = PartitionedLightGCN(model, graph, mode=mode)
        torch.cuda.synchronize()
        build_s = time.perf_counter() - t0

        def step(k):
            with torch.no_grad():
                return part.propagate(x0)

        wall, warm = _time_steps(step, args.warmup, args.steps, ctx)
        wall = ctx.max_over_ranks(wall)
        res[mode] = {"ms_per_step": wall / args.steps * 1e3, "edges_per_s": L * E * args.steps / wall, "build_s": build_s,
                     "local_edges_rank0": part.local_edges, "rows_rank0": part.hi - part.lo, "warmup": warm}
        del part
        torch.cuda.empty_cache()
    if ctx.rank != 0:
        return None
    best = min(res, key=lambda m: res[m]["ms_per_step"])
    return {"metric": "LightGCN propagated directed edges/sec", "value": res[best]["edges_per_s"], "unit": "edges/s",
            "n_gpus": ctx.world, "steps": args.steps, "warmup": res[best]["warmup"], "ms_per_step": res[best]["ms_per_step"],
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg4: GraphNCF {L}-layer LightGCN, {U} users x {I} items, {E} directed edges (Zipf items), D={D}, hetero, mean",
                       "parallelism": f"{best} (the faster of the two below) x{ctx.world}; the graph is fixed: strong scaling"},
            "partitionings": {"dst_blocks_all_gather": res["dst"], "edge_split_all_reduce": res["edge"],
                              "collective_bytes_per_layer": {"dst": f"{(I + U) * D * 4 * (ctx.world - 1) // ctx.world} received per rank (direct block exchange, all links)",
                                                            "edge": f"{2 * (ctx.world - 1) * (I + U) * D * 4 // ctx.world} moved per rank (ring all-reduce of the (N, D) partial sums)"}},
            "roofline": None}


# ---------------------------------------------------------------------------------------------------- cfg 3
CFG3_DIMS = (100_000, 4096, 256, 2094, 64, 64, 128)    # I, B, nnz, F, IE, UE, A (SURVEY §8d cfg 3)


def cfg3_workload(device, users=64, per_pair=False, n_batches=4):
    """BASELINE config 3's synthetic workload (also run, at full size, by tests/test_gpu_attention.py): a 100 k-item catalogue
    of F = 2094 sparse binary features, B = 4096 (user, candidate) pairs per batch drawn from ``users`` distinct users with
    exactly 256 rated items each (sampled without replacement, values r - 2.9, r in {0.5..5}), pairs in random user order.  The
    provider hands over ONE CSR row per distinct user + pair_row (SparseDynamicProvider.collate_interacted_items), so the model
    takes the LDS-tiled grouped kernel when users repeat; ``per_pair`` expands to one CSR row per pair (what a dense user_matrix
    of distinct rows converts to: the per-pair kernel).  ``users`` = B gives 4096 distinct users (no sharing at all)."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF, SparseRatings
    I, B, nnz, Fdim, IE, UE, A = CFG3_DIMS
    torch.manual_seed(7)
    model = AttentionNCF(item_dim=Fdim, item_emb=IE, user_emb=UE, att_dense=A, mlp_dense_layers=[256, 128]).eval().to(device)
    g = torch.Generator(device=device).manual_seed(7)
    catalogue = (torch.rand(I, Fdim, device=device, generator=g) < 0.02).float()
    batches = []
    for _ in range(n_batches):
        cand = catalogue[torch.randint(0, I, (B,), device=device, generator=g)].contiguous()
        if users <= 256:
            col = torch.stack([torch.randperm(I, device=device, generator=g)[:nnz].sort().values for _ in range(users)])
        else:   # many users: 256 distinct columns per row from one argsort of random keys per chunk of rows
            col = torch.cat([torch.rand(256, I, device=device, generator=g).topk(nnz, dim=1).indices.sort(dim=1).values
                             for _ in range((users + 255) // 256)])[:users]
        who = torch.randint(0, users, (B,), device=device, generator=g) if users < B else torch.randperm(B, device=device, generator=g)
        val = torch.randint(1, 11, (users * nnz,), device=device, generator=g).float() * 0.5 - 2.9
        rowptr = torch.arange(0, (users + 1) * nnz, nnz, device=device, dtype=torch.int64)
        r = SparseRatings(rowptr, col.reshape(-1).to(torch.int32).contiguous(), val, I, pair_row=who)
        batches.append((cand, r.expanded() if per_pair else r))
    return model, catalogue, batches


def run_cfg3(args, ctx):
    import bench
    from deeprecommendation_amd import native
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF, SparseRatings
    device = ctx.device
    I, B, nnz, Fdim, IE, UE, A = CFG3_DIMS
    per_pair = os.environ.get("NCF_CFG3_PER_PAIR") == "1"
    n_users = int(os.environ.get("NCF_CFG3_USERS", "64"))
    model, catalogue, batches = cfg3_workload(device, users=n_users, per_pair=per_pair)
    with torch.no_grad():
        model.precompute_catalog(catalogue)

        def step(k):
            cand, r = batches[k % len(batches)]
            return model(cand, catalogue, r)

        wall_eager, warm = _time_steps(step, args.warmup, args.steps)
        # the step is a dozen 5-60 us kernels: also replay it as ONE HIP graph launch per step (each step copies its batch
        # into the graph's static buffers first) and report the faster form
        wall = wall_eager
        wall_graph = None
        graph_err = None
        if os.environ.get("NCF_CFG3_NO_GRAPH") != "1":
            try:
                from deeprecommendation_amd.graphs import GraphedForward
                graphed = GraphedForward(lambda c, cat, um: model(c, cat, um), [batches[0][0], catalogue, batches[0][1]])
                graphed.static_in[1] = catalogue                      # the catalogue never changes: no copy per step

                def gstep(k):
                    cand, r = batches[k % len(batches)]
                    return graphed(cand, catalogue, r)

                ref = step(1).clone()
                if not torch.equal(ref, gstep(1)):
                    raise RuntimeError("graph replay differs from the eager step")
                wall_graph, _ = _time_steps(gstep, args.warmup, args.steps)
                wall = min(wall_eager, wall_graph)                   # a GPU-bound step gains nothing (the copies cost)
            except Exception as exc:   # never take the line down: fall back to the eager timing
                graph_err = str(exc)
                wall = wall_eager
        # third form: the resident batches captured IN ORDER in one HIP graph (no copies: one launch = len(batches) steps) — what is
        # left when the host is out of the loop entirely
        wall_graph_nb = None
        graph_nb_err = None
        steps_per_graph_launch = len(batches)
        if os.environ.get("NCF_CFG3_NO_GRAPH") != "1":
            try:
                nbt = len(batches)
                side = torch.cuda.Stream(device=device)
                side.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(side):
                    for k in range(nbt):
                        step(k)
                torch.cuda.current_stream(device).wait_stream(side)
                # steps per graph launch: several passes over the resident batches (the gap between two graph launches is shared by more
                # steps), the largest multiple of the batch count up to 32 that divides the requested number of steps
                per_launch = next((pl for pl in range(32 // nbt * nbt, nbt - 1, -nbt) if args.steps % pl == 0), nbt)
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    gouts = [step(k % nbt) for k in range(per_launch)]
                gr.replay()
                torch.cuda.synchronize()
                if not (torch.equal(gouts[1], step(1)) and torch.equal(gouts[per_launch - 1], step(nbt - 1))):
                    raise RuntimeError("graph replay differs from the eager step")
                full = (args.steps // per_launch) * per_launch
                steps_per_graph_launch = per_launch

                def gstep_nb(j):                   # steps 0 .. full-1 in graph launches of per_launch steps, the remainder one by one
                    if j < full:
                        if j % per_launch == 0:
                            gr.replay()
                    else:
                        step(j)
                wall_graph_nb, _ = _time_steps(gstep_nb, args.warmup, args.steps)
                # after the back-to-back replays of the timed region: the outputs still equal the eager step's (a replayed graph whose
                # nodes lost their order shows only from the second replay on — the hipMemsetAsync trap of DESIGN §4.15)
                if not (torch.equal(gouts[1], step(1)) and torch.equal(gouts[per_launch - 1], step((per_launch - 1) % nbt))):
                    wall_graph_nb = None
                    raise RuntimeError("graph replay differs from the eager step after the timed replays")
                wall = min(wall, wall_graph_nb)
                del gr, gouts
            except Exception as exc:   # noqa: BLE001
                graph_nb_err = str(exc)
        # fourth form (reported beside `value`, never as `value`): the same resident batches with TWO steps in flight — even steps on one
        # stream, odd ones on another, captured in one HIP graph.  The steps are independent (an evaluation pass scores batch after batch),
        # and each of the three kernels of a step leaves CUs idle (launch ramps, dependent loads, partly filled rounds of workgroups): a
        # second step fills them.  Throughput, not latency: a step still takes what it takes.
        wall_two = None
        two_err = None
        if os.environ.get("NCF_CFG3_NO_GRAPH") != "1":
            try:
                nbt = len(batches)
                cap = torch.cuda.Stream(device=device)
                lanes = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
                cap.wait_stream(torch.cuda.current_stream(device))
                with torch.cuda.stream(cap):
                    for k in range(nbt):
                        step(k)
                torch.cuda.current_stream(device).wait_stream(cap)
                torch.cuda.synchronize()
                gr2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr2, stream=cap):
                    outs2 = [None] * nbt
                    for ln in lanes:
                        ln.wait_stream(cap)                         # fork
                    for k in range(nbt):
                        with torch.cuda.stream(lanes[k % 2]):
                            outs2[k] = step(k)
                    for ln in lanes:
                        cap.wait_stream(ln)                         # join
                gr2.replay()
                torch.cuda.synchronize()
                for k in range(nbt):
                    if not torch.equal(outs2[k], step(k)):
                        raise RuntimeError("two-stream graph replay differs from the eager step")
                full2 = (args.steps // nbt) * nbt

                def gstep2(j):
                    if j < full2:
                        if j % nbt == 0:
                            gr2.replay()
                    else:
                        step(j)
                wall_two, _ = _time_steps(gstep2, args.warmup, args.steps)
                del gr2, outs2
            except Exception as exc:   # noqa: BLE001
                two_err = str(exc)
        # dominant kernel
        rated_emb, pr, proj = model.precompute_catalog(catalogue)
        cand, r = batches[0]
        li = model.ItemEmbeddings[0]
        cand_emb = native.linear(cand, li.weight.detach(), li.bias.detach())
        wc, _, b0 = model._att_split()
        pc = native.linear(cand_emb, wc, b0)
        w1, b1 = model._refresh()["att_out"]
        bias_u = model.UserEmbeddings[0].bias.detach()
        rs = batches[0][1] if not per_pair else None
        rx = r.expanded()
        us_pp = bench.back_to_back_us(lambda: native.attn_forward(native.ATT_MLP_SCALED, pc, pr, w1, b1, rx.rowptr, rx.col, rx.val, proj, out_bias=bias_u), reps=50, settle=10)
        ppw = native.default_pairs_per_wg(B)
        nsplit = 1
        us_g = us_g_b2b = us_group_prep = None
        if not per_pair:
            grouping = (native.group_pairs(rs.pair_row, rs.rowptr.numel() - 1, ppw), ppw)

            def grouped():
                # what the model's forward launches: the entry-split kernel alone (its partials are merged by the tail kernel)
                return native.attn_forward_grouped(native.ATT_MLP_SCALED, pc, pr, w1, b1, rs.rowptr, rs.col, rs.val, rs.pair_row, proj, out_bias=bias_u,
                                                   grouping=grouping, leave_partials=True)

            nsplit = native.default_attn_nsplit(B, rs.rowptr.numel() - 1, rs.col.numel(), ppw)
            kt = bench.kernel_time("ncf::attn_split_kernel", "cfg3", grouped, reps=50, settle=20)
            us_g, us_g_b2b = kt["us"], kt["us_back_to_back"]
            us_group_prep = bench.back_to_back_us(lambda: native.group_pairs(rs.pair_row, rs.rowptr.numel() - 1, ppw), reps=50, settle=5)
        else:
            kt = bench.kernel_time("ncf::attn_kernel", "cfg3", lambda: native.attn_forward(native.ATT_MLP_SCALED, pc, pr, w1, b1, rx.rowptr, rx.col, rx.val, proj, out_bias=bias_u), reps=50, settle=10)
        lin_us = bench.back_to_back_us(lambda: native.linear(cand, li.weight.detach(), li.bias.detach()), reps=50, settle=5)
        cand_us = None
        if native.attn_candidates_supported(Fdim, IE, A) and rs is not None:
            prow = rs.pair_row.to(torch.int64).contiguous()
            wpk = native.PackedCandidateWeight(li.weight.detach())         # what the model's forward uses (packed once per weight version)
            cand_us = bench.back_to_back_us(lambda: native.attn_candidates(cand, wpk, li.bias.detach(), wc, b0, prow, rs.rowptr.numel() - 1, ppw),
                                            reps=50, settle=5)
        # second workload: every pair its own user (4096 distinct rated sets): nothing to share, the per-pair kernel
        distinct = None
        if not per_pair and not getattr(args, "no_variants", False) and os.environ.get("NCF_CFG3_NO_DISTINCT") != "1":

            model_d, catalogue_d, batches_d = cfg3_workload(device, users=B, n_batches=2)
            model_d.precompute_catalog(catalogue_d)

            def step_d(k):
                c_, r_ = batches_d[k % len(batches_d)]
                return model_d(c_, catalogue_d, r_)

            wall_d, _ = _time_steps(step_d, args.warmup, args.steps)
            rxd = batches_d[0][1].expanded()
            us_d = bench.back_to_back_us(lambda: native.attn_forward(native.ATT_MLP_SCALED, pc, pr, w1, b1, rxd.rowptr, rxd.col, rxd.val, proj, out_bias=bias_u), reps=30, settle=5)
            bpp_d = nnz * (A * 4 + UE * 4 + 4 + 4 + 3 * 4)
            distinct = {"workload": f"the same batch shape with {B} DISTINCT users (one rated set per pair: the model dispatches the per-pair kernel)",
                        "pairs_per_s": B * args.steps / wall_d, "ms_per_step": wall_d / args.steps * 1e3,
                        "roofline": {"kernel": "attn_kernel<3> (one wave per pair)", "bound": "cache bandwidth (tables of 51 + 26 MB sit in L2 / the Infinity Cache)",
                                     "us_per_launch": us_d, "algorithmic_bytes_per_pair": bpp_d, "achieved": bpp_d * B / (us_d * 1e-6) / 1e9,
                                     "unit": "GB/s of gathered rows", "peak": bench.PEAK_HBM_GBS, "frac": bpp_d * B / (us_d * 1e-6) / 1e9 / bench.PEAK_HBM_GBS,
                                     "valu_TFLOPs": nnz * (4 * A + 2 * UE + 8) * B / (us_d * 1e-6) / 1e12,
                                     "note": "frac is against the 8 TB/s HBM figure although the rows come from cache: the guide measures 8.6 TB/s chip-wide for random rows out of the Infinity Cache"}}
            del model_d, catalogue_d, batches_d
    us = kt["us"]
    # What the kernel executes per (pair, rated entry): A x (add, max, fma) for the score (4 flop per a) + UE x fma for the
    # aggregation (2 flop per feature) + the softmax arithmetic — the reformulated attention (AttentionNet.0 split at the cat
    # boundary, UserEmbeddings linearity), NOT the reference's per-pair (2 IE -> A) GEMM.  Its operands come from LDS / cache
    # (rows are staged once per workgroup), so the bound is the fp32 VECTOR issue rate: 157.3 TFLOP/s (= the fp32 MFMA figure).
    flop_per_pair = nnz * (4 * A + 2 * UE + 8)
    tf = flop_per_pair * B / (us * 1e-6) / 1e12
    bytes_per_pair_pp = nnz * (A * 4 + UE * 4 + 4 + 4 + 3 * 4)   # pr row + projected row + col + val + weights r/w
    bytes_per_pair_g = nnz * (A * 4 + UE * 4 + 4 + 4) / ppw + A * 4 + UE * 4 + 8
    bpp = bytes_per_pair_pp if per_pair else bytes_per_pair_g
    line = {"metric": "AttentionNCF scored pairs/sec", "value": B * args.steps / wall, "unit": "pairs/s", "n_gpus": 1,
            "steps": args.steps, "warmup": warm, "warmup_requested": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg3: AttentionNCF, catalogue {I}, {nnz} rated/user, IE=UE={IE}, A={A}, F={Fdim}, B={B} "
                                   f"({n_users} users per batch, pairs in random user order; "
                                   f"{'one CSR row per pair: per-pair kernel' if per_pair else 'one CSR row per user + pair_row: LDS-tiled grouped kernel'}); "
                                   "catalogue projections precomputed; attention net split + UserEmbeddings linearity; "
                                   + ("steps captured in ONE HIP graph over the resident batches (no copies), one launch per "
                                      f"{steps_per_graph_launch} steps" if wall_graph_nb is not None and wall == wall_graph_nb
                                      else "step = batch copied into static buffers + ONE HIP-graph launch" if wall_graph is not None and wall == wall_graph and wall_graph < wall_eager
                                      else "step enqueued kernel by kernel from Python"),
                       "reference_formulation_mfma_bound_pairs_per_s": 157.3e12 / (nnz * (2 * 2 * IE * A + 2 * A) + 2 * (128 * 256 + 256 * 128 + 128)),
                       "eager_ms_per_step": wall_eager / args.steps * 1e3, "eager_pairs_per_s": B * args.steps / wall_eager,
                       "graph_replay_ms_per_step": None if wall_graph is None else wall_graph / args.steps * 1e3,
                       "graph_of_resident_batches_ms_per_step": None if wall_graph_nb is None else wall_graph_nb / args.steps * 1e3,
                       "graph_of_resident_batches_error": graph_nb_err,
                       "two_steps_in_flight": None if wall_two is None else {
                           "ms_per_step": wall_two / args.steps * 1e3, "pairs_per_s": B * args.steps / wall_two,
                           "what": "the same resident batches, even steps on one stream and odd steps on another, one HIP graph: independent steps "
                                   "fill each other's idle CUs.  A throughput figure (evaluation passes, serving); `value` stays the one-step-at-a-time rate"},
                       "two_steps_in_flight_error": two_err,
                       "graph_error": graph_err},
            "roofline": {"kernel": "attn_kernel<0>" if per_pair else f"attn_split_kernel<3,{ppw // 4},64> x {nsplit} slices of each rated set", "bound": "valu", "achieved": tf,
                         "peak": bench.PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / bench.PEAK_F32_MFMA_TFLOPS,
                         "traffic": kt["traffic"], "us_per_launch": us, "us_per_launch_basis": kt["basis"], "us_back_to_back": kt["us_back_to_back"],
                         "us_isolated_events": kt["us_isolated_events"],
                         "algorithmic_flop_per_pair": flop_per_pair, "algorithmic_bytes_per_pair": bpp,
                         "algorithmic_bytes_per_launch": bpp * B, "hbm_GBps_at_this_rate": bpp * B / (us * 1e-6) / 1e9,
                         "per_pair_kernel_us": us_pp, "grouped_kernel_us": us_g, "grouping_prep_us": us_group_prep,
                         "pairs_per_workgroup": ppw, "slices_per_rated_set": nsplit, "candidate_linear_us": lin_us,
                         "candidate_kernel_us": cand_us,
                         "rocprof_avg_us": kt["rocprof_avg_us"], "profile": kt["profile"], "profile_check": kt["profile_check"],
                         "note": "bound = fp32 vector (VALU) issue: the contract's hbm/mfma pair does not describe this kernel — its tiles come "
                                 "from L2 / the Infinity Cache once per workgroup (hbm_GBps_at_this_rate is what it NEEDS, a few % of HBM), "
                                 "and relu sits between the add and the dot, so the matrix cores cannot take the (pair, entry, a) loop; "
                                 "peak = 157.3 TFLOP/s fp32 vector = the fp32 MFMA figure"}}
    if distinct is not None:
        line["variants"] = {"distinct_users": distinct}
        try:
            line["variants"]["reference_eval_shape"] = _cfg3_reference_eval_shape(device, args)
        except Exception as exc:  # noqa: BLE001
            line["variants"]["reference_eval_shape"] = {"error": f"{type(exc).__name__}: {exc}"}
    if not getattr(args, "no_cpu_baseline", False):
        # CPU oracle, reference formulation (materialised candidate x rated pairs, attention_ncf.py:154-213) on a down-scaled
        # sample: 64 pairs of 16 users against those users' own rated items (the full 4096 x 100k pair grid is 2·B·I·IE floats)
        from oracle import ncf_oracle as O
        state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        gcpu = torch.Generator().manual_seed(3)
        users = 16
        colc = torch.stack([torch.randperm(I, generator=gcpu)[:nnz].sort().values for _ in range(users)])
        rated_ids = torch.unique(colc)
        rated_cpu = catalogue[rated_ids.to(device)].cpu()
        who_c = torch.randint(0, users, (64,), generator=gcpu)
        um = torch.zeros(64, rated_ids.numel())
        pos = torch.searchsorted(rated_ids, colc)
        for b_ in range(64):
            um[b_, pos[who_c[b_]]] = torch.randint(1, 11, (nnz,), generator=gcpu).float() * 0.5 - 2.9
        cand_cpu = catalogue[torch.randint(0, I, (64,), generator=gcpu).to(device)].cpu()
        sec, cores = _cpu_median_s(lambda: O.attention_ncf_forward(state, cand_cpu, rated_cpu, um), reps=20, warm=5)
        line["cpu_baseline"] = {"value": 64 / sec, "unit": "pairs/s", "cores": cores, "kind": "port",
                                "sample": f"reference formulation on 64 pairs of {users} users x {nnz} rated against their {int(rated_ids.numel())} rated items, "
                                          f"median of 20 forwards after 5 warm-ups ({sec * 1e3:.0f} ms each), torch CPU fp32"}
    return line


def _cfg3_reference_eval_shape(device, args):
    """The reference's OWN evaluation call (eval.py:120-127 -> dynamic_datasets.py:24-40): batch of 512 samples, the union of the batch
    users' rated items as `rated_items` (I = 1174: the whole catalogue of its dataset), F = 2094 features, a DENSE (512, 1174)
    user_matrix with a user's row repeated for each of their samples, model dims of the shipped checkpoints (IE = UE = 128, A = 128,
    MLP [256, 128]).  The dense matrix goes through the on-stream CSR conversion (no host read); CPU baseline = the oracle's
    reference formulation on the same batch."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    F, I, B, users = 2094, 1174, 512, 64
    torch.manual_seed(21)
    model = AttentionNCF(item_dim=F, item_emb=128, user_emb=128, att_dense=128, mlp_dense_layers=[256, 128]).eval().to(device)
    g = torch.Generator(device=device).manual_seed(22)
    rated = ((torch.rand(I, F, device=device, generator=g) < 0.02).float() + torch.rand(I, F, device=device, generator=g) * 0.1)
    batches = []
    for _ in range(4):
        rows = torch.zeros(users, I, device=device)
        mask = torch.rand(users, I, device=device, generator=g) < 0.125                  # ~146 rated items per user (SURVEY §6 probe)
        rows[mask] = (torch.randint(1, 11, (users, I), device=device, generator=g).float() * 0.5 - 2.9)[mask]
        who = torch.arange(B, device=device) // (B // users)                              # test files are grouped by user
        cand = rated[torch.randint(0, I, (B,), device=device, generator=g)].contiguous()
        batches.append((cand, rows[who].contiguous()))
    with torch.no_grad():
        def step(k):
            c, um = batches[k % 4]
            return model(c, rated, um)
        wall_eager, warm = _time_steps(step, args.warmup, args.steps)
        out = step(0).clone()
        # the forward reads nothing back (the dense matrix is converted on the stream), so it replays as ONE HIP graph launch per batch:
        # the batch is copied into the graph's static buffers, the catalogue tensor is shared
        wall_graph, graph_err = None, None
        try:
            from deeprecommendation_amd.graphs import GraphedForward
            graphed = GraphedForward(lambda c, r_, um: model(c, r_, um), [batches[0][0], rated, batches[0][1]])
            graphed.static_in[1] = rated

            def gstep(k):
                c, um = batches[k % 4]
                return graphed(c, rated, um)

            if not torch.equal(gstep(0), out):
                raise RuntimeError("graph replay differs from the eager step")
            wall_graph, _ = _time_steps(gstep, args.warmup, args.steps)
        except Exception as exc:   # noqa: BLE001
            graph_err = str(exc)
        wall = wall_eager if wall_graph is None else min(wall_eager, wall_graph)
    res = {"workload": f"model(candidates ({B}, {F}), rated_items ({I}, {F}), DENSE user_matrix ({B}, {I}): {users} users x {B // users} samples), "
                       "IE = UE = 128, att_dense = 128, MLP [256, 128]; "
                       + ("step = batch copied into static buffers + ONE HIP-graph launch" if wall_graph is not None and wall == wall_graph
                          else "step enqueued from Python"),
           "pairs_per_s": B * args.steps / wall, "ms_per_step": wall / args.steps * 1e3,
           "eager_ms_per_step": wall_eager / args.steps * 1e3, "graph_replay_ms_per_step": None if wall_graph is None else wall_graph / args.steps * 1e3,
           "graph_error": graph_err}
    if not getattr(args, "no_cpu_baseline", False):
        from oracle import ncf_oracle as O
        state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        c0, um0 = batches[0][0].cpu(), batches[0][1].cpu()
        rated_c = rated.cpu()
        ref = O.attention_ncf_forward(state, c0, rated_c, um0)
        res["max_rel_err_vs_oracle"] = float((out.cpu() - ref).abs().max() / ref.abs().max())
        sec, cores = _cpu_median_s(lambda: O.attention_ncf_forward(state, c0, rated_c, um0), reps=5, warm=1)
        res["cpu_baseline"] = {"value": B / sec, "unit": "pairs/s", "cores": cores, "kind": "port",
                               "sample": f"the same batch through the oracle's reference formulation (attention_ncf.py:136-224), median of 5 forwards ({sec * 1e3:.0f} ms each), torch CPU fp32"}
    return res


# ---------------------------------------------------------------------------------------------------- cfg 5
def _cfg5_cpu_baseline(ws, bs, E, B):
    """CPU oracle, table formulation with bf16 operands (oracle.basic_ncf_forward_indexed_bf16's arithmetic) on down-scaled
    tables (4 M users x 1 M items: the CPU rate does not depend on the table height once it is far beyond the caches)."""
    from oracle import ncf_oracle as O
    Uc, Ic = 4_000_000, 1_000_000
    g = torch.Generator().manual_seed(5)
    tu = (torch.randn(Uc, E, generator=g) * 0.05).to(torch.bfloat16)
    ti = (torch.randn(Ic, E, generator=g) * 0.05).to(torch.bfloat16)
    layers = [(w.detach().cpu().to(torch.bfloat16).float() if k < len(ws) - 1 else w.detach().cpu().float(), b.detach().cpu().float())
              for k, (w, b) in enumerate(zip(ws, bs))]
    iu = torch.randint(0, Uc, (B,), generator=g)
    ii = torch.randint(0, Ic, (B,), generator=g)

    def step():
        x = torch.cat((tu[iu], ti[ii]), dim=1).float()
        h = torch.relu(torch.nn.functional.linear(x, *layers[0])).to(torch.bfloat16).float()   # hidden layer 1 re-rounded to bf16
        h = torch.relu(torch.nn.functional.linear(h, *layers[1]))
        return torch.nn.functional.linear(h, *layers[2])

    with torch.no_grad():
        sec, cores = _cpu_median_s(step, reps=20, warm=3)
    return {"value": B / sec, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"table formulation, bf16 tables {Uc} x {E} / {Ic} x {E} (down-scaled from 100 M / 10 M rows) and bf16-rounded weights, "
                      f"fp32 accumulate, batch {B}, median of 20 after 3 warm-ups ({sec * 1e3:.0f} ms each), torch CPU"}


def run_cfg5(args, ctx):
    """BASELINE config 5: BasicNCF, 100 M users x 10 M items, emb 128 bf16, tables row-sharded over the ranks with
    all-to-all (RCCL) exchange; local batch 65 536 per rank (weak scaling).  At world size 1 there is no exchange."""
    import bench
    from deeprecommendation_amd import native
    from deeprecommendation_amd.sharded import ExchangeOverflow, RowShardedTable, ShardedBasicNCF
    device, world, rank = ctx.device, ctx.world, ctx.rank
    U, I, E, B = 100_000_000, 10_000_000, 128, 65_536
    if os.environ.get("NCF_CFG5_SMALL") == "1":      # rehearsals on a shared GPU: 1/50 of the rows
        U, I = 2_000_000, 200_000
    strong = os.environ.get("NCF_CFG5_STRONG") == "1"   # fixed global batch of 65 536 pairs (SURVEY 8d cfg 5, second form)
    if strong:
        B = B // world
    if os.environ.get("NCF_CFG5_LOCAL_BATCH"):           # its own profile context (profiles/<tag>_cfg5_b<B>_*): the 1 M-pair point
        B = int(os.environ["NCF_CFG5_LOCAL_BATCH"])
    ctxname = "cfg5" if B == 65_536 else f"cfg5_b{B}"
    big = not getattr(args, "no_variants", False) and B == 65_536
    replicate = os.environ.get("NCF_REPLICATE_ITEMS", "1") == "1"
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    ulo, uhi = RowShardedTable.shard_bounds(U, world, rank)
    ilo, ihi = (0, I) if replicate else RowShardedTable.shard_bounds(I, world, rank)

    def table(rows):
        t = torch.empty((rows, E), dtype=torch.bfloat16, device=device)
        for s in range(0, rows, 4_000_000):
            t[s:s + 4_000_000] = (torch.randn((min(4_000_000, rows - s), E), device=device, generator=g) * 0.05).to(torch.bfloat16)
        return t

    tu, ti = table(uhi - ulo), table(ihi - ilo)
    gw = torch.Generator(device=device).manual_seed(99)  # same MLP on every rank
    dims = [2 * E, 256, 128, 1]
    ws = [(torch.rand((dims[k + 1], dims[k]), device=device, generator=gw) * 2 - 1) / dims[k] ** 0.5 for k in range(3)]
    bs = [(torch.rand((dims[k + 1],), device=device, generator=gw) * 2 - 1) / dims[k] ** 0.5 for k in range(3)]
    gi = torch.Generator(device=device).manual_seed(2024 + rank)
    batches = [(torch.randint(0, U, (B,), device=device, generator=gi), torch.randint(0, I, (B,), device=device, generator=gi))
               for _ in range(8)]
    nb = len(batches)
    forms = {}

    def timed(model, pipelined):
        if pipelined:
            state = {"t": None}

            def step(k):   # the exchange of step k+1 is enqueued before step k is scored
                if state["t"] is None:
                    state["t"] = model.submit(*batches[k % nb])
                nxt = model.submit(*batches[(k + 1) % nb])
                out = model.score(state["t"])
                state["t"] = nxt
                return out
        else:
            def step(k):
                return model(*batches[k % nb])
        wall, warm = _time_steps(step, args.warmup, args.steps, ctx)
        if pipelined and state["t"] is not None:
            model.score(state["t"])        # drain the last submitted exchange
        return ctx.max_over_ranks(wall), warm

    def timed_checked(model, pipelined):
        """Time a bounded form, then check() — a COLLECTIVE, every rank raises the same exception.  A pass with dropped pairs is never
        reported: on ExchangeOverflow every rank grows the capacity (same value) and the pass is timed again (at most 3 times);
        any other error (an id out of range) is recorded and the form is left out of `best`."""
        info = {}
        for attempt in range(4):
            model.wire_stats()                     # reset the counters: they describe the timed pass
            wall, warm = timed(model, pipelined)
            try:
                model.check()
                info.update({"ms_per_step": wall / args.steps * 1e3, "pairs_per_s": world * B * args.steps / wall, "warmup": warm,
                             "capacity": {"users": model.users.cap, "items": None if model.items is None else model.items.cap},
                             "capacity_regrown": attempt})
                if world > 1:
                    st = model.wire_stats()
                    su = st["users"]
                    steps_seen = max(su["lookups"], 1)
                    info["wire_rank0"] = {k: {"row_bytes_on_wire_per_step": v["row_bytes_on_wire"] / steps_seen,
                                              "row_bytes_needed_per_step": v["row_bytes_needed"] / steps_seen,
                                              "id_bytes_on_wire_per_step": v["id_bytes_on_wire"] / steps_seen,
                                              "padding_fraction": v["padding_fraction"],
                                              "duplicates_removed_fraction": v["duplicates_removed_fraction"]} for k, v in st.items()}
                return wall, warm, info
            except ExchangeOverflow:
                model.grow_capacity()
            except IndexError as exc:
                return None, None, {"error": f"IndexError: {exc}"}
        return None, None, {"error": "ExchangeOverflow after three capacity growths"}

    model = ShardedBasicNCF(tu, U, ti, I, ws, bs, replicate_items=replicate, dtype=torch.bfloat16, exchange="bounded")
    caps = None
    if world > 1:
        caps = model.negotiate_capacity(*batches[0])     # the ONE host read of the bounded exchange
    wall, warm, info = timed_checked(model, pipelined=world > 1)
    forms["bounded_pipelined" if world > 1 else "single_gpu"] = info
    if wall is None:
        raise RuntimeError(f"cfg5 main form failed: {info}")
    main_wall, main_warm = wall, warm
    step_form = "one launch per step from Python"
    if world == 1:
        # The step is ONE ~16 us kernel, about what Python + ctypes need to enqueue it: the same steps captured in a HIP graph
        # (the 8 resident batches in order, no copies: one graph launch = 8 steps) take the host out of the loop.
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                for k in range(nb):
                    model(*batches[k])
            torch.cuda.current_stream(device).wait_stream(side)
            # steps per launch: the largest divisor of the requested step count up to 32 (the batches are cycled)
            spl = next(pl for pl in range(min(args.steps, 32), 0, -1) if args.steps % pl == 0)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                gouts = [model(*batches[k % nb]) for k in range(spl)]
            graph.replay()
            torch.cuda.synchronize()
            for k in (0, spl - 1):
                if not torch.equal(gouts[k], model(*batches[k % nb])):
                    raise RuntimeError("graph replay differs from the eager step")

            def gstep(j):                      # one graph launch per spl steps
                if j % spl == 0:
                    graph.replay()
            wall_g, _ = _time_steps(gstep, args.warmup, args.steps, ctx)
            for k in (0, spl - 1):               # still equal after the timed, back-to-back replays
                if not torch.equal(gouts[k], model(*batches[k % nb])):
                    raise RuntimeError("graph replay differs from the eager step after the timed replays")
            wall_g = ctx.max_over_ranks(wall_g)
            forms["single_gpu_graph"] = {"ms_per_step": wall_g / args.steps * 1e3, "pairs_per_s": B * args.steps / wall_g,
                                         "steps_per_graph_launch": spl}
            if wall_g < main_wall:
                main_wall = wall_g
                step_form = f"HIP graph of {spl} steps (the resident batches in order, no copies) per launch"
            del graph, gouts
        except Exception as exc:  # noqa: BLE001
            forms["single_gpu_graph"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world > 1:
        for name, kw, pipe in (("bounded_serial", {"exchange": "bounded"}, False),
                               ("bounded_pipelined_no_dedup", {"exchange": "bounded", "dedup": False}, True),
                               ("unique_dedup_host_sizes", {"exchange": "unique"}, False)):
            try:
                m2 = ShardedBasicNCF(tu, U, ti, I, ws, bs, replicate_items=replicate, dtype=torch.bfloat16, **kw)
                if kw["exchange"] == "bounded":
                    m2.negotiate_capacity(*batches[0])
                    _, _, forms[name] = timed_checked(m2, pipelined=pipe)
                else:
                    w2, _ = timed(m2, pipelined=pipe)        # exact splits: an id out of range raises inside the step
                    forms[name] = {"ms_per_step": w2 / args.steps * 1e3, "pairs_per_s": world * B * args.steps / w2,
                                   "exchange_stats_rank0": m2.users.last_stats}
                del m2
            except Exception as exc:  # noqa: BLE001 — a side-by-side form must not cost the main one
                forms[name] = {"error": f"{type(exc).__name__}: {exc}"}
        best = min((f for f in forms if "ms_per_step" in forms[f]), key=lambda f: forms[f]["ms_per_step"])
        main_wall = forms[best]["ms_per_step"] * 1e-3 * args.steps
    # dominant kernel at this rank: the bf16 fused kernel straight off this rank's shards (random local rows from HBM)
    gl = torch.Generator(device=device).manual_seed(77 + rank)
    lu = torch.randint(0, tu.shape[0], (B,), device=device, generator=gl)
    li = torch.randint(0, ti.shape[0], (B,), device=device, generator=gl)
    out = torch.empty((B, 1), device=device)
    kt = bench.kernel_time("ncf::score_ws8_bf16_kernel", ctxname, lambda: native.score_fused(tu, lu, ti, li, model.packed, out=out), reps=100, settle=60)
    us, us_b2b = kt["us"], kt["us_back_to_back"]
    flop = 2 * (256 * 256 + 256 * 128 + 128)
    tf = flop * B / (us * 1e-6) / 1e12
    # the same kernel on a 16x larger local batch (north_star: ">= 50 % MFMA utilisation on the MLP at emb_dim = 128"); skipped
    # under the profiler (--no-variants): the same kernel name at another size would pollute rocprof's per-kernel average —
    # that point has its own context (NCF_CFG5_LOCAL_BATCH=1048576 -> profiles/<tag>_cfg5_b1048576_*)
    big_variants = {}
    if big:
        for BL, reps in ((16 * B, 30), (64 * B, 12)):
            lu2 = torch.randint(0, tu.shape[0], (BL,), device=device, generator=gl)
            li2 = torch.randint(0, ti.shape[0], (BL,), device=device, generator=gl)
            outl = torch.empty((BL, 1), device=device)
            ktl = bench.kernel_time("ncf::score_ws8_bf16_kernel", f"cfg5_b{BL}", lambda: native.score_fused(tu, lu2, ti, li2, model.packed, out=outl), reps=reps, settle=10)
            tfl = flop * BL / (ktl["us"] * 1e-6) / 1e12
            bv = {"kernel": "score_ws8_bf16_kernel<256,true>", "us_per_launch": ktl["us"], "us_per_launch_basis": ktl["basis"],
                  "us_back_to_back": ktl["us_back_to_back"], "rocprof_avg_us": ktl["rocprof_avg_us"], "profile": ktl["profile"], "profile_check": ktl["profile_check"],
                  "pairs_per_s": BL / (ktl["us"] * 1e-6), "achieved_TFLOPs": tfl, "frac_of_bf16_mfma_peak": tfl / bench.PEAK_BF16_MFMA_TFLOPS,
                  "frac_back_to_back": flop * BL / (ktl["us_back_to_back"] * 1e-6) / 1e12 / bench.PEAK_BF16_MFMA_TFLOPS,
                  "hbm_GBps_at_this_rate": 532 * BL / (ktl["us"] * 1e-6) / 1e9}
            dg = bench.profile_digest("ncf::score_ws8_bf16_kernel", f"cfg5_b{BL}")
            if dg and dg.get("held_clock_GHz"):
                bv["held_clock_GHz_under_counters"] = dg["held_clock_GHz"]
                bv["mfma_busy_frac_of_held_cycles"] = dg.get("mfma_busy_frac_of_held_cycles")
                bv["note"] = ("the 2.5 PF peak assumes 2.4 GHz; under this kernel the chip holds the clock above (GRBM_GUI_ACTIVE / 8 / "
                              "kernel time in the PMC pass of the same context); the same binary on all-zero operands runs at 0.56 "
                              "of 2.5 PF (DESIGN 4.2 c)")
            big_variants[f"local_batch_{BL}"] = bv
            del lu2, li2, outl
    if rank != 0:
        return None
    line = {"metric": "scored user-item pairs/sec", "value": world * B * args.steps / main_wall, "unit": "pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": main_warm, "warmup_requested": args.warmup, "ms_per_step": main_wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"cfg5: BasicNCF {U} users x {I} items, emb_dim={E} bf16, local batch {B}, MLP 256-256-128-1, "
                                   f"user table row-sharded x{world}, item table {'replicated' if replicate else 'row-sharded'}",
                       "step": step_form,
                       "exchange": "none (both tables on the one GPU)" if world == 1 else
                                   "device-side owner bucketing into fixed-capacity buffers, equal-split all-to-alls, step t+1's exchange "
                                   "on a second stream under step t's MLP; value = the fastest form below"},
            "exchange_forms": forms,
            "roofline": {"kernel": "score_ws8_bf16_kernel<256,true>", "bound": "mfma", "achieved": tf, "peak": bench.PEAK_BF16_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": tf / bench.PEAK_BF16_MFMA_TFLOPS, "traffic": kt["traffic"], "us_per_launch": us,
                         "us_per_launch_basis": kt["basis"], "us_back_to_back": us_b2b, "us_isolated_events": kt["us_isolated_events"],
                         "algorithmic_flop_per_pair": flop, "algorithmic_bytes_per_pair": 532,
                         "algorithmic_bytes_per_launch": 532 * B, "hbm_GBps_at_this_rate": 532 * B / (us * 1e-6) / 1e9,
                         "rocprof_avg_us": kt["rocprof_avg_us"], "profile": kt["profile"], "profile_check": kt["profile_check"]},
            "variants": big_variants}
    if world == 1 and big:
        # What "2.5 PFLOP/s" is worth on THIS chip, now: the bf16 MFMA rate it sustains on random operands (it lowers its clock under
        # matrix load), measured by the library's probe kernel in this process — and every bf16 figure of the line as a fraction of it.
        try:
            pr = native.probe_mfma_bf16(device, seconds=2.0)
            prl = native.probe_mfma_bf16(device, seconds=1.0, with_lds=True)
            rl = line["roofline"]
            rl["sustained_mfma_probe"] = {"TFLOPs": pr["TFLOPs"], "clock_GHz": pr["clock_GHz"], "cycles_per_mfma_per_simd": pr["cycles_per_mfma_per_simd"],
                                          "with_lds_operand_reads": {"TFLOPs": prl["TFLOPs"], "clock_GHz": prl["clock_GHz"]},
                                          "what": pr["mfma"] + f", {pr['launches']} back-to-back launches of {pr['us_per_launch'] / 1e3:.2f} ms, the last 4 timed",
                                          "frac_of_spec_peak": pr["TFLOPs"] / bench.PEAK_BF16_MFMA_TFLOPS}
            rl["frac_of_sustained_mfma"] = tf / pr["TFLOPs"]
            for bv in big_variants.values():
                bv["frac_of_sustained_mfma"] = bv["achieved_TFLOPs"] / pr["TFLOPs"]
        except Exception as exc:  # noqa: BLE001 — calibration must never cost the line
            line["roofline"]["sustained_mfma_probe"] = {"error": f"{type(exc).__name__}: {exc}"}
    if world == 1 and not getattr(args, "no_cpu_baseline", False):
        line["cpu_baseline"] = _cfg5_cpu_baseline(ws, bs, E, B)
    return line


# ---------------------------------------------------------------------------------------------------- train2
def run_train2(args, ctx):
    """Training step on the cfg-2 shape: BasicNCF 1 M x 100 k, emb 64, batch 65 536, MLP [256,128],
    dropout 0.2, MSE-sum loss, Adam over every parameter (the reference's optimiser, train.py:55).  Forward + backward
    of the gather and Linear(+ReLU) layers on the HIP autograd blocks, next to the same step with plain torch ops
    (rocBLAS / ATen) on the same GPU.  Unit: trained pairs/s."""
    import bench
    from deeprecommendation_amd.optim import FusedAdam
    device = ctx.device
    res = {}
    modes = ("hip_blocks_fused_adam", "hip_blocks", "torch_ops", "torch_ops_fused_adam")
    if os.environ.get("NCF_TRAIN2_ONLY"):
        modes = (os.environ["NCF_TRAIN2_ONLY"],)
    warm = 0
    for mode in modes:
        model = bench.make_model(device).train()
        model.train_with_torch_ops = mode.startswith("torch_ops")
        if mode == "hip_blocks_fused_adam":
            opt = FusedAdam(model.parameters(), lr=1e-3)
        elif mode == "torch_ops_fused_adam":
            try:
                opt = torch.optim.Adam(model.parameters(), lr=1e-3, fused=True)   # torch's own fused kernel, for reference
            except Exception:
                res[mode] = None
                continue
        else:
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        batches = bench.make_batches(device, 0)
        g = torch.Generator(device=device).manual_seed(5)
        y = torch.rand((bench.B, 1), device=device, generator=g) * 5

        def step(k):
            iu, ii = batches[k % len(batches)]
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(model(iu, ii), y, reduction="sum")
            loss.backward()
            opt.step()

        wall, warm = _time_steps(step, args.warmup, args.steps)
        res[mode] = wall / args.steps
        del model, opt
        torch.cuda.empty_cache()
    best = res.get("hip_blocks_fused_adam") or next(iter(res.values()))
    return {"metric": "trained user-item pairs/sec (forward + backward + Adam)", "value": bench.B / best, "unit": "pairs/s",
            "n_gpus": 1, "steps": args.steps, "warmup": warm, "warmup_requested": args.warmup, "ms_per_step": best * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "train2: BasicNCF 1M x 100k, emb 64, batch 65536, MLP 128-256-128-1, dropout 0.2, MSE, Adam (dense)",
                       "step": "HIP blocks (column gather / scatter, Linear + ReLU fwd / dgrad / wgrad / bias grad) + FusedAdam (ncf_adam_step)",
                       "same_step_hip_blocks_with_torch_adam_ms": res["hip_blocks"] * 1e3 if "hip_blocks" in res else None,
                       "same_step_with_torch_ops_ms": res["torch_ops"] * 1e3 if "torch_ops" in res else None,
                       "same_step_with_torch_ops_and_torch_fused_adam_ms": None if res.get("torch_ops_fused_adam") is None else res["torch_ops_fused_adam"] * 1e3,
                       "note": "both ways the step is dominated by dense full-table work (dense table gradients + dense Adam over 71 M "
                               "parameters, ~2.5 ms), which the reference's Linear-layout embeddings imply; the HIP blocks cover the MLP "
                               "forward / dgrad / wgrad / bias-grad / ReLU mask"}}


# ---------------------------------------------------------------------------------------------------- cfg 2 at emb_dim = 128, fp32
def run_cfg2_emb128(args, ctx):
    """north_star's "MFMA utilisation on the MLP at emb_dim = 128" in fp32 (its own rocprofv3 context): BasicNCF 1 M x 100 k tables of
    128 fp32 columns (512 MB + 51 MB), MLP 256-256-128-1, batch 65 536 — the K0 = 256 instance of the fused kernel."""
    import bench
    from deeprecommendation_amd import native
    device = ctx.device
    U, I, E, B, H = bench.U, bench.I, 128, bench.B, bench.HIDDEN
    g = torch.Generator(device=device).manual_seed(11)
    tu = torch.empty((U, E), device=device).normal_(0.0, 0.05, generator=g)
    ti = torch.empty((I, E), device=device).normal_(0.0, 0.05, generator=g)
    dims = [2 * E, H[0], H[1], 1]
    ws = [(torch.rand((dims[k + 1], dims[k]), device=device, generator=g) * 2 - 1) / dims[k] ** 0.5 for k in range(3)]
    bs = [(torch.rand((dims[k + 1],), device=device, generator=g) * 2 - 1) / dims[k] ** 0.5 for k in range(3)]
    packed = native.PackedMLP(ws, bs)
    batches = [(torch.randint(0, U, (B,), device=device, generator=g), torch.randint(0, I, (B,), device=device, generator=g)) for _ in range(8)]
    out = torch.empty((B, 1), device=device)

    def step(k):
        return native.score_fused(tu, batches[k % 8][0], ti, batches[k % 8][1], packed, out=out)

    wall, warm = _time_steps(step, args.warmup, args.steps)
    cyc = [0]

    def fused():
        cyc[0] = (cyc[0] + 1) % 8
        native.score_fused(tu, batches[cyc[0]][0], ti, batches[cyc[0]][1], packed, out=out)

    kt = bench.kernel_time("ncf::score_fused_f32_kernel<256, 256, 128>", "cfg2_emb128", fused)
    flop = 2 * (2 * E * H[0] + H[0] * H[1] + H[1])
    tf = flop * B / (kt["us"] * 1e-6) / 1e12
    return {"metric": "scored user-item pairs/sec", "value": B * args.steps / wall, "unit": "pairs/s", "n_gpus": 1, "steps": args.steps,
            "warmup": warm, "warmup_requested": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg2 at emb_dim=128: BasicNCF {U} users x {I} items, emb_dim={E} fp32, batch={B}, MLP 256-256-128-1 (kernel calls, no model wrapper)"},
            "roofline": {"kernel": "score_fused_f32_kernel<256,256,128>", "bound": "mfma", "achieved": tf, "peak": bench.PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": tf / bench.PEAK_F32_MFMA_TFLOPS, "traffic": kt["traffic"], "us_per_launch": kt["us"],
                         "us_per_launch_basis": kt["basis"], "us_back_to_back": kt["us_back_to_back"], "rocprof_avg_us": kt["rocprof_avg_us"],
                         "profile": kt["profile"], "profile_check": kt["profile_check"],
                         "algorithmic_flop_per_pair": flop, "algorithmic_bytes_per_pair": 2 * E * 4 + 16 + 4,
                         "algorithmic_bytes_per_launch": (2 * E * 4 + 16 + 4) * B}}


# name -> (function, (default steps, default warm-up))
WORKLOADS = {"cfg2_emb128": (run_cfg2_emb128, (200, 300)), "cfg3": (run_cfg3, (64, 5)), "cfg4": (run_cfg4, (5, 1)), "cfg5": (run_cfg5, (100, 10)), "train2": (run_train2, (20, 3))}
