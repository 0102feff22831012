"""bench.py — BASELINE.json headline metric: scored user-item pairs/s of the NCF scoring hot path on MI355X.

Workload of the headline line (config.workload = "cfg2"): BASELINE.json configs[1] — BasicNCF, 1 M users x 100 k items,
emb_dim 64, batch 65 536, fp32, MLP [256, 128] -> 1; synthetic seeded tables / weights / indices (no datasets offline).
A "step" = one forward of one batch through the product path (BasicNCF.forward on int64 positions ->
ncf_score_fused through the C ABI), inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment STARTS the N ranks itself: this process, before it has made
any GPU call, runs the torch.distributed.run command above as a child and exits with its code (a process that has
touched the GPU is never re-exec'd).  Under torchrun, WORLD_SIZE must equal --gpus.

N > 1, headline line: replicas (pairs are independent; tables + MLP replicated, the batch is split, no
data-path collective) -> weak scaling, value = N * 65 536 * K / max-over-ranks time.

`other_configs` (default run only; `--workload cfg2` prints the headline line alone, `--workload cfgK` that config alone):
  N = 1: cfg 3 (AttentionNCF), cfg 4 (LightGCN propagation, 100 M directed edges), cfg 5 (bf16 emb 128, the 28 GB of
         tables on one GPU), each with its own `roofline` and `cpu_baseline`;
  N > 1: the two configs that SHARD — cfg 5 (row-sharded tables, all-to-all over RCCL; pipelined / serial / de-duplicating
         exchange side by side) and cfg 4 (partitioned LightGCN: destination blocks + all-gather and edge split +
         all-reduce side by side).
Every secondary config runs after the headline measurement, in the same processes, and a failure there is recorded in
its entry instead of taking the line down.

Every `roofline.frac` = algorithmic work per launch / the kernel's ISOLATED launch duration (HIP events around single
launches on the launch stream, a synchronisation between launches — what rocprofv3's serialised kernel trace
reports, and what `profiles/<round>_*` hold for the same command); the back-to-back figure (a launch's ramp overlapping
the previous launch's tail) is kept as `us_back_to_back`.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

U, I, E, B = 1_000_000, 100_000, 64, 65_536
HIDDEN = [256, 128]
FLOP_PER_PAIR = 2 * (2 * E * HIDDEN[0] + HIDDEN[0] * HIDDEN[1] + HIDDEN[1])      # 131 328
FUSED_BYTES_PER_PAIR = 2 * E * 4 + 2 * 8 + 4                                       # 532 B   
GATHER_BYTES_PER_PAIR = 2 * (2 * E * 4) + 2 * 8                                    # 1040 B: rows read + rows written + ids
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
N_BATCHES = 16  # distinct index batches cycled through, so no step re-reads the previous step's rows from cache
PROFILE_TAG = "r03"
PROFILE_TAGS = ("r03", "r02")   # profiles/<tag>_*: the rocprofv3 summaries of THIS round's bench command (tools/profile*.sh)


class Ctx:
    """Process placement of one bench run."""

    def __init__(self, device, rank, world, dist):
        self.device, self.rank, self.world, self.dist = device, rank, world, dist

    def barrier(self):
        torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.device if self.dist.get_backend() == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())


def make_model(device):
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    torch.manual_seed(1234)
    with torch.device("meta"):
        model = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=HIDDEN)
    model = model.to_empty(device=device).eval()
    g = torch.Generator(device=device).manual_seed(1234)
    with torch.no_grad():
        # tables ~ N(0, 0.05^2) (SURVEY §8d cfg 2).  Linear weight is [E, U]: table row i = W[:, i] + b.
        for lin in (model.user_embeddings[0], model.item_embeddings[0]):
            lin.weight.normal_(0.0, 0.05, generator=g)
            lin.bias.normal_(0.0, 0.05, generator=g)
        for m in model.MLP:
            if isinstance(m, torch.nn.Linear):
                bound = 1.0 / (m.in_features ** 0.5)  # nn.Linear default init range
                m.weight.uniform_(-bound, bound, generator=g)
                m.bias.uniform_(-bound, bound, generator=g)
    return model


def make_batches(device, rank, zipf_users=False):
    g = torch.Generator().manual_seed(2024 + rank)
    if zipf_users:
        return [(zipf_indices(U, B, 1.05, device, 2024 + 31 * k + rank), torch.randint(0, I, (B,), generator=g).to(device))
                for k in range(N_BATCHES)]
    return [(torch.randint(0, U, (B,), generator=g).to(device), torch.randint(0, I, (B,), generator=g).to(device))
            for _ in range(N_BATCHES)]


def profile_digest(kernel_prefix, context):
    """HBM traffic per launch from the committed rocprofv3 PMC passes of THIS bench context
    (profiles/<PROFILE_TAG>_<context>_digest.json, written by tools/summarize_profile.py: FETCH_SIZE x2-corrected +
    WRITE_SIZE, per the gfx950 guide) and rocprof's own mean duration of the same kernel.  None when that profile is not
    committed (never another context's)."""
    d = None
    for tag in PROFILE_TAGS:      # this round's profile of the context; an earlier round's only where the kernel did not change
        path = os.path.join(ROOT, "profiles", f"{tag}_{context}_digest.json")
        try:
            d = json.load(open(path))
            break
        except (OSError, ValueError):
            continue
    if d is None:
        return None
    for name, k in d.get("kernels", {}).items():
        if name.startswith(kernel_prefix):
            out = {"rocprof_avg_us": k.get("rocprof_avg_us"), "profile": os.path.basename(path)}
            if k.get("gui_active_cycles_sum8xcd") and k.get("rocprof_avg_us"):
                # GRBM_GUI_ACTIVE sums the 8 XCDs; the quotient is the clock the chip held (reads high on launches under ~0.2 ms)
                out["held_clock_GHz"] = k["gui_active_cycles_sum8xcd"] / 8 / k["rocprof_avg_us"] / 1e3
                if k.get("mfma_busy_cycles"):
                    out["mfma_busy_frac_of_held_cycles"] = k["mfma_busy_cycles"] / 1024 / (k["gui_active_cycles_sum8xcd"] / 8)
            if "fetch_bytes" in k or "write_bytes" in k:
                out["traffic"] = k.get("fetch_bytes", 0.0) + k.get("write_bytes", 0.0)
                out["fetch_bytes"], out["write_bytes"] = k.get("fetch_bytes"), k.get("write_bytes")
            return out
    return None


def host_cores():
    """CPU share of this process: the cgroup quota when there is one (the GPU box gives 16 of 256 cores), else the
    affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


# ---------------------------------------------------------------------------------------------------- kernel timing
def back_to_back_us(fn, reps=100, settle=60):
    """Mean duration of `reps` back-to-back launches (HIP events on the launch stream = torch's current stream)."""
    for _ in range(settle):  # the chip takes milliseconds to settle its clock after a change of kernel
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def isolated_us(fn, reps=100, settle=60):
    """Mean duration of single launches, the stream drained between them (HIP events around each launch): the kernel's
    isolated duration, the figure rocprofv3's serialised kernel trace reports."""
    for _ in range(settle):
        fn()
    torch.cuda.synchronize()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for e0, e1 in pairs:
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in pairs) * 1e3 / reps


def kernel_time(kernel_prefix, context, fn, reps=100, settle=60):
    """The duration of one launch of the dominant kernel:
      us (= us_back_to_back) — measured LIVE in this run: HIP events on the launch stream around `reps` launches in a row
                               (a launch's ramp overlaps its predecessor's tail).  roofline.frac is computed from this.
      rocprof_avg_us        — rocprofv3's serialised kernel trace of this same bench context, committed under profiles/:
                               the cross-check (`profile_check`), flagged when the two differ by more than 10 %.  A slower
                               box or a regressed kernel moves `frac`; the committed figure cannot hide it."""
    b2b = back_to_back_us(fn, reps, settle)
    # (isolated_us() is not run here: its ~5-8 us of dispatch + event overhead per launch says nothing about the kernel, and under
    # rocprofv3 its drained-stream launches — the chip clocks down in the gaps — would only pollute the profiler's average)
    dig = profile_digest(kernel_prefix, context) or {}
    prof = dig.get("rocprof_avg_us")
    check = None
    if prof:
        check = {"rocprof_avg_us": prof, "profile": dig.get("profile"), "live_over_profile": b2b / prof,
                 "agrees_within_10pct": bool(abs(b2b / prof - 1.0) <= 0.10),
                 "note": "rocprofv3 serialises launches (each pays its own ramp and drain) and lowers the clock: it reads 3-10 % longer than back-to-back launches"}
    return {"us": b2b, "basis": f"HIP events around {reps} back-to-back launches on the launch stream, live in this run",
            "us_back_to_back": b2b, "us_isolated_events": None, "rocprof_avg_us": prof, "profile": dig.get("profile"),
            "profile_check": check,
            "traffic": dig.get("traffic"), "fetch_bytes": dig.get("fetch_bytes"), "write_bytes": dig.get("write_bytes")}


def cpu_baseline(model, seconds=12.0):
    """CPU oracle (table formulation of the reference forward, PyTorch CPU fp32) on the host cores: same tables,
    same batch size; bounded to ~`seconds` of CPU work."""
    from oracle import ncf_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    tu = O.embedding_table(state["user_embeddings.0.weight"], state["user_embeddings.0.bias"])
    ti = O.embedding_table(state["item_embeddings.0.weight"], state["item_embeddings.0.bias"])
    layers = O.mlp_weights(state)
    g = torch.Generator().manual_seed(99)
    iu = torch.randint(0, U, (B,), generator=g)
    ii = torch.randint(0, I, (B,), generator=g)

    def step():
        return O.mlp_forward(torch.cat((tu[iu], ti[ii]), dim=1), layers)

    with torch.no_grad():
        step()
        n, t0 = 0, time.perf_counter()
        while True:
            step()
            n += 1
            dt = time.perf_counter() - t0
            if dt >= seconds or n >= 200:
                break
    return {"value": n * B / dt, "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{n} batches of {B} pairs, table formulation (W^T[idx]+b gather -> MLP {2*E}-{HIDDEN[0]}-{HIDDEN[1]}-1), "
                      f"torch CPU fp32, {cores} threads, {dt:.1f} s"}


def cpu_baseline_dense_cfg1(iters=60):
    """the reference-FAITHFUL formulation (dense one-hot rows through nn.Linear, basic_ncf.py:37-42)
    at the largest size that fits comfortably — BASELINE configs[0]: 6040 x 3706, emb 32, MLP [256], batch 512.
    (At cfg 2 the dense input alone would be 262 TB.)  Median of `iters` timed forwards after 5 warm-ups."""
    from oracle import ncf_oracle as O
    torch.manual_seed(0)
    Uc, Ic, Ec, Bc = 6040, 3706, 32, 512
    state = {"user_embeddings.0.weight": torch.randn(Ec, Uc) * 0.01, "user_embeddings.0.bias": torch.zeros(Ec),
             "item_embeddings.0.weight": torch.randn(Ec, Ic) * 0.01, "item_embeddings.0.bias": torch.zeros(Ec),
             "MLP.0.weight": torch.randn(256, 2 * Ec) * 0.1, "MLP.0.bias": torch.zeros(256),
             "MLP.3.weight": torch.randn(1, 256) * 0.1, "MLP.3.bias": torch.zeros(1)}
    g = torch.Generator().manual_seed(1)
    xu = torch.nn.functional.one_hot(torch.randint(0, Uc, (Bc,), generator=g), Uc).float()
    xi = torch.nn.functional.one_hot(torch.randint(0, Ic, (Bc,), generator=g), Ic).float()
    ts = []
    with torch.no_grad():
        for k in range(iters + 5):
            t0 = time.perf_counter()
            O.basic_ncf_forward(state, xu, xi)
            if k >= 5:
                ts.append(time.perf_counter() - t0)
    med = sorted(ts)[len(ts) // 2]
    return {"value": Bc / med, "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"dense one-hot formulation at cfg-1 scale ({Uc} x {Ic}, emb {Ec}, MLP [256], batch {Bc}), median of {iters} forwards"}


def zipf_indices(n_rows, count, alpha, device, seed):
    """Zipf(alpha) ranks over n_rows ids by inverse-CDF sampling on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    w = 1.0 / torch.arange(1, n_rows + 1, device=device, dtype=torch.float64) ** alpha
    cdf = torch.cumsum(w, 0)
    u = torch.rand(count, device=device, generator=g, dtype=torch.float64) * cdf[-1]
    rank = torch.searchsorted(cdf, u).clamp_(max=n_rows - 1)
    perm = torch.randperm(n_rows, device=device, generator=g)  # popular ids scattered over the table
    return perm[rank]


# ---------------------------------------------------------------------------------------------------- cfg 2 (headline)
def run_cfg2(args, ctx):
    from deeprecommendation_amd import native
    device, rank, world = ctx.device, ctx.rank, ctx.world
    model = make_model(device)
    if args.fold:
        model.set_fold_first_layer(True)
    batches = make_batches(device, rank, getattr(args, "zipf_users", False))
    ctxname = "cfg2zipf" if getattr(args, "zipf_users", False) else "cfg2"

    def step(k):
        iu, ii = batches[k % N_BATCHES]
        return model(iu, ii)

    warm = max(args.warmup, 300)
    with torch.no_grad():
        # untimed: the requested warm-up steps, topped up to >= 300 so that the chip's clock has settled on this kernel
        # before the timed region (DVFS takes milliseconds to settle; measured: 50 timed steps straight after 5 warm-ups
        # read 805 M pairs/s, after the settling 905 M) — the timed region is exactly --steps steps either way
        for k in range(warm):
            step(k)
        ctx.barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        ctx.barrier()
        elapsed = time.perf_counter() - t0
        native.check_oob(device)
        # The same steps also as HIP-graph launches of up to 32 steps over the resident index batches (every rank its own graph: replicas
        # share nothing).  A step is one kernel either way; what the graph removes is the host's part of the FIRST step of the timed
        # region (the GPU idles behind the synchronisation until Python has walked a whole forward) — 3-7 % of a 20-step run, nothing
        # at 400 steps.  With several ranks every decision below is taken collectively (a rank that could not capture takes all of them
        # back to the eager figure), so the ranks never disagree on the number of barriers.
        elapsed_eager = ctx.max_over_ranks(elapsed)
        elapsed_graph, graph_err, steps_per_launch = None, None, None
        if os.environ.get("NCF_CFG2_NO_GRAPH") != "1":
            gr = gouts = None
            spl = next(pl for pl in range(min(args.steps, 32), 0, -1) if args.steps % pl == 0)
            try:                                   # capture: no collective in here
                cur = torch.cuda.current_stream(device)
                side = torch.cuda.Stream(device=device)
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    step(0)
                cur.wait_stream(side)
                torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    gouts = [step(k) for k in range(spl)]
                gr.replay()
                torch.cuda.synchronize()
                if not torch.equal(gouts[spl - 1], step(spl - 1)):
                    raise RuntimeError("graph replay differs from the eager step")
            except Exception as exc:   # noqa: BLE001 — never take the line down: the eager timing stands
                graph_err = str(exc)
                gr = None
            if ctx.max_over_ranks(0.0 if gr is not None else 1.0) == 0.0:      # every rank has its graph
                for _ in range(max(1, 300 // spl)):
                    gr.replay()
                ctx.barrier()
                t0 = time.perf_counter()
                for _ in range(args.steps // spl):
                    gr.replay()
                ctx.barrier()
                local = time.perf_counter() - t0
                still = True
                try:
                    native.check_oob(device)
                    still = bool(torch.equal(gouts[spl - 1], step(spl - 1)) and torch.equal(gouts[0], step(0)))   # after the timed replays
                except Exception as exc:   # noqa: BLE001
                    still, graph_err = False, str(exc)
                if not still and graph_err is None:
                    graph_err = "graph replay differs from the eager step after the timed replays"
                bad = ctx.max_over_ranks(0.0 if still else 1.0)
                local = ctx.max_over_ranks(local)
                if bad == 0.0:
                    elapsed_graph, steps_per_launch = local, spl
            elif graph_err is None:
                graph_err = "another rank could not capture its graph"
            del gr, gouts
    elapsed = elapsed_eager if elapsed_graph is None else min(elapsed_eager, elapsed_graph)

    # --- roofline of the dominant kernel (the kernels run on torch's current stream, so torch.cuda.Event brackets them)
    packed = model._packed_mlp()
    tu, ti = model._table("user", model.user_embeddings[0]), model._table("item", model.item_embeddings[0])
    outbuf = torch.empty((B, 1), dtype=torch.float32, device=device)
    cyc = [0]

    def fused():
        k = cyc[0] = (cyc[0] + 1) % N_BATCHES
        native.score_fused(tu, batches[k][0], ti, batches[k][1], packed, out=outbuf)

    kt_f = kernel_time("ncf::score_fused_f32_kernel<128, 256, 128>", ctxname, fused)
    achieved_tf = FLOP_PER_PAIR * B / (kt_f["us"] * 1e-6) / 1e12
    fold_info = None
    if args.fold:
        PA, PB, tail = model._folded(tu, ti, "MLP")
        fus = back_to_back_us(lambda: native.score_folded(PA, batches[1][0], PB, batches[1][1], tail, out=outbuf))
        ex_flop = 2 * (HIDDEN[0] * HIDDEN[1] + HIDDEN[1])
        fold_info = {"kernel": "score_folded_direct_kernel<256,128>", "bound": "mfma", "us_per_launch": fus,
                     "executed_flop_per_pair": ex_flop, "achieved": ex_flop * B / (fus * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": ex_flop * B / (fus * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "algorithmic_bytes_per_pair": 2 * HIDDEN[0] * 4 + 2 * 8 + 4, "traffic": None,
                     "note": "opt-in folded first layer: executes 65 792 of the 131 328 FLOP/pair; the roofline object "
                             "above is the default (unfolded) kernel measured in the same process"}

    gbuf = torch.empty((B, 2 * E), dtype=torch.float32, device=device)

    def gather():
        k = cyc[0] = (cyc[0] + 1) % N_BATCHES
        native.gather_concat(tu, batches[k][0], ti, batches[k][1], out=gbuf)

    kt_g = kernel_time("ncf::gather_concat", ctxname, gather)
    gather_gbs = GATHER_BYTES_PER_PAIR * B / (kt_g["us"] * 1e-6) / 1e9

    variants = None
    if not getattr(args, "no_variants", False):
        # ---- SURVEY §8d cfg 2 variants: Zipf(1.05) user ids, and the [256]-only MLP (train_model.py:40-42 default) ----
        zu = zipf_indices(U, B, 1.05, device, 2024)
        zi = batches[0][1]
        zipf_fused_us = back_to_back_us(lambda: native.score_fused(tu, zu, ti, zi, packed, out=outbuf))
        lin = [m for m in model.MLP if isinstance(m, torch.nn.Linear)]
        g2 = torch.Generator(device=device).manual_seed(7)
        w_last = (torch.rand((1, HIDDEN[0]), device=device, generator=g2) * 2 - 1) / HIDDEN[0] ** 0.5
        packed256 = native.PackedMLP([lin[0].weight, w_last], [lin[0].bias, lin[2].bias])
        h256_us = back_to_back_us(lambda: native.score_fused(tu, batches[1][0], ti, batches[1][1], packed256, out=outbuf))
        zipf_gather_us = back_to_back_us(lambda: native.gather_concat(tu, zu, ti, zi, out=gbuf))
        dig_z = profile_digest("ncf::gather_concat", "cfg2zipf") or {}
        flop256 = 2 * (2 * E * HIDDEN[0] + HIDDEN[0])
        variants = {
            "zipf_1.05_users": {"fused_us_per_launch": zipf_fused_us, "fused_pairs_per_s": B / (zipf_fused_us * 1e-6),
                                "gather_us_per_launch": zipf_gather_us, "gather_rocprof_avg_us": dig_z.get("rocprof_avg_us"),
                                "gather_profile": dig_z.get("profile"),
                                "gather_GBps_algorithmic": GATHER_BYTES_PER_PAIR * B / (zipf_gather_us * 1e-6) / 1e9,
                                "note": "back-to-back HIP-event timings; hot rows are cache hits, so the algorithmic rate is not an HBM figure"},
            "mlp_256_only": {"fused_us_per_launch": h256_us, "fused_pairs_per_s": B / (h256_us * 1e-6),
                             "flop_per_pair": flop256, "frac_of_fp32_mfma_peak": flop256 * B / (h256_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS},
        }
        # ---- north_star's ">= 50 % MFMA utilisation on the MLP at emb_dim = 128", in fp32: the same kernel's K0 = 256 instance on
        # 1 M x 128 and 100 k x 128 fp32 tables (512 MB + 51 MB), MLP 256-256-128-1 — 196 864 FLOP and 1044 B per pair ----
        try:
            E2 = 128
            g3 = torch.Generator(device=device).manual_seed(11)
            tu2 = torch.empty((U, E2), device=device).normal_(0.0, 0.05, generator=g3)
            ti2 = torch.empty((I, E2), device=device).normal_(0.0, 0.05, generator=g3)
            w0 = (torch.rand((HIDDEN[0], 2 * E2), device=device, generator=g3) * 2 - 1) / (2 * E2) ** 0.5
            packed128 = native.PackedMLP([w0, lin[1].weight, lin[2].weight], [lin[0].bias, lin[1].bias, lin[2].bias])
            e128_us = back_to_back_us(lambda: native.score_fused(tu2, batches[1][0], ti2, batches[1][1], packed128, out=outbuf))
            flop128 = 2 * (2 * E2 * HIDDEN[0] + HIDDEN[0] * HIDDEN[1] + HIDDEN[1])
            dig128 = profile_digest("ncf::score_fused_f32_kernel<256, 256, 128>", "cfg2_emb128") or {}
            variants["emb128_fp32"] = {"kernel": "score_fused_f32_kernel<256,256,128>", "bound": "mfma", "us_per_launch": e128_us,
                                       "pairs_per_s": B / (e128_us * 1e-6), "flop_per_pair": flop128, "bytes_per_pair": 2 * E2 * 4 + 16 + 4,
                                       "achieved": flop128 * B / (e128_us * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": flop128 * B / (e128_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                       "rocprof_avg_us": dig128.get("rocprof_avg_us"), "profile": dig128.get("profile"), "traffic": dig128.get("traffic")}
            del tu2, ti2, packed128
        except Exception as exc:  # noqa: BLE001
            variants["emb128_fp32"] = {"error": f"{type(exc).__name__}: {exc}"}
        # ---- what the chip delivers for the standalone gather's two access patterns, measured now (library probes): a streaming
        # copy of the same volume and random 256-byte row reads of the user table ----
        try:
            ceil = native.probe_gather_ceilings(tu, batches[2][0], copy_bytes=B * 2 * E * 4)
            ceil["gather_frac_of_copy_ceiling"] = gather_gbs / ceil["copy"]["GBps"]
            ceil["note"] = ("the gather reads 2 random 256-byte rows and writes 512 contiguous bytes per pair; a streaming copy of the same bytes is "
                            "the most this chip moves for that volume, random 256-byte row reads (read bytes only) what the access pattern allows")
            variants["gather_measured_ceilings"] = ceil
        except Exception as exc:  # noqa: BLE001
            variants["gather_measured_ceilings"] = {"error": f"{type(exc).__name__}: {exc}"}
        # opt-in variant (BasicNCF.set_fold_first_layer): layer 1 folded into 256-wide tables; reported beside the default
        # line, never as `value` (its results agree with the oracle to 1e-5 but are not bit-identical to the default kernel's)
        if fold_info is None and native.folded_supported(HIDDEN[0], HIDDEN[1]):
            try:
                PAv, PBv, tailv = model._folded(tu, ti, "MLP")
                fv_us = back_to_back_us(lambda: native.score_folded(PAv, batches[1][0], PBv, batches[1][1], tailv, out=outbuf))
                variants["folded_first_layer_opt_in"] = {"fused_us_per_launch": fv_us, "fused_pairs_per_s": B / (fv_us * 1e-6),
                                                         "executed_flop_per_pair": 2 * (HIDDEN[0] * HIDDEN[1] + HIDDEN[1]),
                                                         "bytes_per_pair": 2 * HIDDEN[0] * 4 + 2 * 8 + 4,
                                                         "table_bytes": int(PAv.numel() + PBv.numel()) * 4}
                del PAv, PBv
            except Exception as exc:  # the variant must never take the default line down
                variants["folded_first_layer_opt_in"] = {"error": str(exc)}

    if rank != 0:
        return None
    total_pairs = world * B * args.steps
    line = {
        "metric": "scored user-item pairs/sec",
        "value": total_pairs / elapsed,
        "unit": "pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": warm,
        "warmup_requested": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "cfg2: BasicNCF 1M users x 100k items, emb_dim=64, batch=65536/GPU, fp32, MLP 128-256-128-1"
                               + (" [Zipf(1.05) user ids]" if ctxname == "cfg2zipf" else "")
                               + (" [first MLP layer FOLDED into 256-wide tables: 65 792 executed FLOP and 2068 B per pair]" if args.fold else ""),
                   "parallelism": f"replicas x{world} (tables+MLP replicated, batch split, no collective)",
                   "step_form": ("HIP-graph launches of %d steps over the resident index batches" % steps_per_launch
                                 if elapsed_graph is not None and elapsed_graph <= elapsed_eager else "one forward per step, enqueued from Python"),
                   "eager_ms_per_step": elapsed_eager / args.steps * 1e3,
                   "graph_ms_per_step": None if elapsed_graph is None else elapsed_graph / args.steps * 1e3, "graph_error": graph_err},
        "roofline": {"kernel": "score_fused_f32_kernel<128,256,128>", "bound": "mfma", "achieved": achieved_tf,
                     "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / PEAK_F32_MFMA_TFLOPS,
                     "traffic": kt_f["traffic"], "us_per_launch": kt_f["us"], "us_per_launch_basis": kt_f["basis"],
                     "us_back_to_back": kt_f["us_back_to_back"], "us_isolated_events": kt_f["us_isolated_events"],
                     "rocprof_avg_us": kt_f["rocprof_avg_us"], "profile": kt_f["profile"], "profile_check": kt_f["profile_check"],
                     "frac_from_timed_region": FLOP_PER_PAIR * B / (elapsed / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "algorithmic_flop_per_pair": FLOP_PER_PAIR, "algorithmic_bytes_per_pair": FUSED_BYTES_PER_PAIR,
                     "algorithmic_bytes_per_launch": FUSED_BYTES_PER_PAIR * B,
                     "note": "frac = algorithmic FLOP per launch / us_per_launch; frac_from_timed_region divides by the driver-visible "
                             "step time instead (one launch per step plus the host's enqueue gaps: a lower bound)"},
        "gather_roofline": {"kernel": "gather_concat_persistent<32> (standalone K1)", "bound": "hbm", "achieved": gather_gbs,
                            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gather_gbs / PEAK_HBM_GBS,
                            "traffic": kt_g["traffic"], "us_per_launch": kt_g["us"], "us_per_launch_basis": kt_g["basis"],
                            "us_back_to_back": kt_g["us_back_to_back"], "us_isolated_events": kt_g["us_isolated_events"],
                            "rocprof_avg_us": kt_g["rocprof_avg_us"], "profile": kt_g["profile"], "profile_check": kt_g["profile_check"],
                            "ids": "Zipf(1.05) users" if ctxname == "cfg2zipf" else "uniform",
                            "algorithmic_bytes_per_pair": GATHER_BYTES_PER_PAIR, "algorithmic_bytes_per_launch": GATHER_BYTES_PER_PAIR * B,
                            "frac_of_measured_copy_ceiling": ((variants or {}).get("gather_measured_ceilings") or {}).get("gather_frac_of_copy_ceiling")},
    }
    if variants is not None:
        line["variants"] = variants
    if fold_info is not None:
        line["folded_roofline"] = fold_info
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(model)
        line["cpu_baseline_reference_formulation"] = cpu_baseline_dense_cfg1()
    return line


# ---------------------------------------------------------------------------------------------------- launch / main
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """--gpus N without a torchrun environment: start the N ranks as children (torch.distributed.run), relay their
    output (rank 0 prints the one JSON line) and exit with their code.  Nothing in this process has touched the GPU:
    torch.cuda.device_count() does not initialise it on this image."""
    have = torch.cuda.device_count()
    if have < args.gpus and os.environ.get("NCF_BENCH_SINGLE_DEVICE") != "1":
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {have} GPU(s)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # the host driver only supports dmabuf IPC (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // args.gpus)))
    return subprocess.call(cmd, env=env)


def init_ctx(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus is not None and args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, "
                         "or run `python bench.py --gpus N` alone and let it start the ranks")
    if os.environ.get("NCF_BENCH_ANNOUNCE") == "1":
        print(f"bench.py: rank {rank} of {world} (local rank {local_rank})", file=sys.stderr, flush=True)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    if os.environ.get("NCF_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0  # rehearsal of the multi-process flow on a one-GPU box (all ranks share cuda:0)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("NCF_BENCH_FORCE_DIST") == "1":  # FORCE_DIST: exercise the RCCL init/barrier path at world size 1
        import torch.distributed as dist
        backend = os.environ.get("NCF_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm; gloo only for rehearsals
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return Ctx(device, rank, world, dist)


def _guarded(name, fn, args, ctx):
    """A secondary config never takes the line down: its failure is recorded (every rank reaches the next config)."""
    t0 = time.perf_counter()
    try:
        res = fn(args, ctx)
    except Exception as exc:  # noqa: BLE001
        import traceback
        res = {"error": f"{type(exc).__name__}: {exc}", "traceback_tail": traceback.format_exc()[-600:]}
    torch.cuda.synchronize()
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    if isinstance(res, dict):
        res["bench_wall_s"] = time.perf_counter() - t0
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fold", action="store_true",
                    help="opt-in: fold the first MLP layer into the tables (BasicNCF.set_fold_first_layer); NOT the default line")
    ap.add_argument("--no-variants", action="store_true", help="cfg2: skip the variants block (profiling runs: one id distribution per run)")
    ap.add_argument("--zipf-users", action="store_true", help="cfg2: Zipf(1.05) user ids instead of uniform ones (its own profile context)")
    ap.add_argument("--workload", default=None, choices=["cfg2", "cfg2_emb128", "cfg3", "cfg4", "cfg5", "train2"],
                    help="one config alone (cfg2 = the headline line without other_configs); default: headline + other_configs")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ:
        if args.gpus is not None and args.gpus > 1:
            sys.exit(launch_ranks(args, sys.argv[1:]))
        args.gpus = 1
    # torch sizes its OpenMP pool from the visible CPUs (256 on the GPU box, of which the cgroup grants 16): surplus workers
    # spinning after a host-side parallel region get the enqueueing thread throttled — keep the pool within the share
    torch.set_num_threads(min(torch.get_num_threads(), host_cores()))
    ctx = init_ctx(args)
    from deeprecommendation_amd import native
    native.load_library()
    import bench_extra
    user_steps, user_warmup = args.steps, args.warmup

    def sized(steps, warmup):
        a = argparse.Namespace(**vars(args))
        a.steps = user_steps if user_steps is not None else steps
        a.warmup = user_warmup if user_warmup is not None else warmup
        return a

    if args.workload not in (None, "cfg2"):
        fn, (st, wu) = bench_extra.WORKLOADS[args.workload]
        line = fn(sized(st, wu), ctx)
    else:
        line = run_cfg2(sized(400, 40), ctx)
        if args.workload is None:
            others = {}
            if ctx.rank == 0:
                line["other_configs"] = others
            names = ("cfg3", "cfg4", "cfg5") if ctx.world == 1 else ("cfg5", "cfg4")
            budget_end = time.monotonic() + SECONDARY_BUDGET_S   # ONE budget for all secondary configs (they take ~40 s at one GPU)
            for name in names:
                fn, (st, wu) = bench_extra.WORKLOADS[name]
                a = argparse.Namespace(**vars(args))
                a.steps, a.warmup = st, wu          # the secondary configs keep their own step counts
                left = budget_end - time.monotonic()
                if left < 15.0:                     # every rank takes the same branch only approximately: the deadline below is the
                    left = 15.0                     # safety net, so keep calling (a config that cannot finish is abandoned there)
                # a secondary config that hangs (a collective whose peer died) must not cost the headline measurement: past
                # the budget every rank leaves, rank 0 printing the line with what it has
                guard = _Deadline(left, name, line if ctx.rank == 0 else None, others)
                res = _guarded(name, fn, a, ctx)
                guard.cancel()
                if ctx.rank == 0:
                    others[name] = res
    if ctx.rank == 0:
        print(json.dumps(line), flush=True)
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


SECONDARY_BUDGET_S = 300.0   # after the headline: the driver allows the whole run 600 s


class _Deadline:
    def __init__(self, seconds, name, line, others):
        import threading
        self.name, self.line, self.others, self.seconds = name, line, others, seconds
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        self.timer.cancel()

    def _fire(self):
        if self.line is not None:
            self.others[self.name] = {"error": f"no result within the {self.seconds:.0f} s left of the secondary-config budget: abandoned"}
            print(json.dumps(self.line), flush=True)
        os._exit(0)


if __name__ == "__main__":
    main()
