"""bench.py — BASELINE.json headline metric: scored user-item pairs/s of the NCF scoring hot path on MI355X.

Workload (config.workload = "cfg2"): BASELINE.json configs[1] — BasicNCF, 1 M users x 100 k items, emb_dim 64,
batch 65 536, fp32, MLP [256, 128] -> 1; synthetic seeded tables / weights / indices (no datasets offline).
A "step" = one forward of one batch through the product path (BasicNCF.forward on int64 positions ->
ncf_score_fused through the C ABI), inputs already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

N > 1: replicas (pairs are independent; tables + MLP replicated, the batch is split, no data-path
collective) -> weak scaling, value = N * 65 536 * K / max-over-ranks time.

The JSON line carries `roofline` (dominant kernel = the fused gather+MLP kernel, fp32-MFMA-bound, timed live with
HIP events on the launch stream), `gather_roofline` (the standalone K1 gather kernel vs the 8 TB/s HBM peak — the
second half of BASELINE's metric) and `cpu_baseline` (the CPU oracle's table formulation timed on the host cores,
rank 0, N = 1 only, bounded sample).
"""
import argparse
import json
import os
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
PEAK_HBM_GBS = 8000.0
N_BATCHES = 16  # distinct index batches cycled through, so no step re-reads the previous step's rows from cache


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


def make_batches(device, rank):
    g = torch.Generator().manual_seed(2024 + rank)
    return [(torch.randint(0, U, (B,), generator=g).to(device), torch.randint(0, I, (B,), generator=g).to(device))
            for _ in range(N_BATCHES)]


def profile_digest(kernel_prefix):
    """HBM traffic per launch from the committed rocprofv3 PMC pass (profiles/<tag>_digest.json, written by
    tools/summarize_profile.py: FETCH_SIZE x2-corrected + WRITE_SIZE, per the gfx950 guide) and rocprof's own mean
    duration of the same kernel.  None when no profile of this kernel is committed."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_digest.json"))):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        for name, k in d.get("kernels", {}).items():
            if name.startswith(kernel_prefix) and "fetch_bytes" in k:
                best = {"traffic": k.get("fetch_bytes", 0.0) + k.get("write_bytes", 0.0),
                        "rocprof_avg_us": k.get("rocprof_avg_us"), "profile": os.path.basename(path)}
    return best


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fold", action="store_true",
                    help="opt-in: fold the first MLP layer into the tables (BasicNCF.set_fold_first_layer); NOT the default line")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg3", "cfg4", "cfg5", "train2"],
                    help="cfg2 = BASELINE headline (default); cfg3 / cfg4 = the other single-GPU configs (bench_extra.py)")
    args = ap.parse_args()
    # torch sizes its OpenMP pool from the visible CPUs (256 on the GPU box, of which the cgroup grants 16): surplus workers
    # spinning after a host-side parallel region get the enqueueing thread throttled — keep the pool within the share
    torch.set_num_threads(min(torch.get_num_threads(), host_cores()))
    if args.workload != "cfg2":
        import bench_extra
        return bench_extra.main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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

    from deeprecommendation_amd import native
    native.load_library()
    model = make_model(device)
    if args.fold:
        model.set_fold_first_layer(True)
    batches = make_batches(device, rank)

    def step(k):
        iu, ii = batches[k % N_BATCHES]
        return model(iu, ii)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        # untimed: the requested warm-up steps, topped up to >= 300 so that the chip's clock has settled on this kernel
        # before the timed region (DVFS takes milliseconds to settle; measured: 50 timed steps straight after 5 warm-ups
        # read 805 M pairs/s, after the settling 905 M) — the timed region is exactly --steps steps either way
        for k in range(max(args.warmup, 300)):
            out = step(k)
        barrier()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        t0 = time.perf_counter()
        ev[0].record()
        for k in range(args.steps):
            out = step(k)
        ev[1].record()
        barrier()
        elapsed = time.perf_counter() - t0
        native.check_oob(device)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # --- roofline of the dominant kernel: per-launch duration from HIP events around back-to-back launches on the
    # launch stream (the kernels run on torch's current stream, so torch.cuda.Event brackets exactly them).
    iu, ii = batches[0]
    packed = model._packed_mlp()
    tu, ti = model._table("user", model.user_embeddings[0]), model._table("item", model.item_embeddings[0])
    outbuf = torch.empty((B, 1), dtype=torch.float32, device=device)
    reps = 100
    for k in range(60):
        native.score_fused(tu, batches[k % N_BATCHES][0], ti, batches[k % N_BATCHES][1], packed, out=outbuf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(reps):
        native.score_fused(tu, batches[k % N_BATCHES][0], ti, batches[k % N_BATCHES][1], packed, out=outbuf)
    e1.record()
    torch.cuda.synchronize()
    fused_us = e0.elapsed_time(e1) * 1e3 / reps
    achieved_tf = FLOP_PER_PAIR * B / (fused_us * 1e-6) / 1e12
    fold_info = None
    if args.fold:
        PA, PB, tail = model._folded(tu, ti, "MLP")
        for k in range(5):
            native.score_folded(PA, batches[k % N_BATCHES][0], PB, batches[k % N_BATCHES][1], tail, out=outbuf)
        e0.record()
        for k in range(reps):
            native.score_folded(PA, batches[k % N_BATCHES][0], PB, batches[k % N_BATCHES][1], tail, out=outbuf)
        e1.record()
        torch.cuda.synchronize()
        fus = e0.elapsed_time(e1) * 1e3 / reps
        ex_flop = 2 * (HIDDEN[0] * HIDDEN[1] + HIDDEN[1])
        fold_info = {"kernel": "score_folded_direct_kernel<256,128>", "bound": "mfma", "us_per_launch": fus,
                     "executed_flop_per_pair": ex_flop, "achieved": ex_flop * B / (fus * 1e-6) / 1e12, "peak": PEAK_F32_MFMA_TFLOPS,
                     "unit": "TFLOP/s", "frac": ex_flop * B / (fus * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "algorithmic_bytes_per_pair": 2 * HIDDEN[0] * 4 + 2 * 8 + 4, "traffic": None,
                     "note": "opt-in folded first layer: executes 65 792 of the 131 328 FLOP/pair; the roofline object "
                             "above is the default (unfolded) kernel measured in the same process"}

    gbuf = torch.empty((B, 2 * E), dtype=torch.float32, device=device)
    for k in range(60):  # settle the clock on this (memory-bound) kernel before timing it
        native.gather_concat(tu, batches[k % N_BATCHES][0], ti, batches[k % N_BATCHES][1], out=gbuf)
    e0.record()
    for k in range(reps):
        native.gather_concat(tu, batches[k % N_BATCHES][0], ti, batches[k % N_BATCHES][1], out=gbuf)
    e1.record()
    torch.cuda.synchronize()
    gather_us = e0.elapsed_time(e1) * 1e3 / reps
    gather_gbs = GATHER_BYTES_PER_PAIR * B / (gather_us * 1e-6) / 1e9

    # ---- SURVEY §8d cfg 2 variants: Zipf(1.05) user ids, and the [256]-only MLP (train_model.py:40-42 default) ----
    def per_launch(fn, n=100):
        for _ in range(60):  # the chip takes milliseconds to settle its clock after a change of kernel: warm up long enough
            fn()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    zu = zipf_indices(U, B, 1.05, device, 2024)
    zi = batches[0][1]
    per_launch(lambda: native.score_fused(tu, batches[2][0], ti, batches[2][1], packed, out=outbuf))  # back to this kernel's clock state
    zipf_fused_us = per_launch(lambda: native.score_fused(tu, zu, ti, zi, packed, out=outbuf))
    lin = [m for m in model.MLP if isinstance(m, torch.nn.Linear)]
    g2 = torch.Generator(device=device).manual_seed(7)
    w_last = (torch.rand((1, HIDDEN[0]), device=device, generator=g2) * 2 - 1) / HIDDEN[0] ** 0.5
    packed256 = native.PackedMLP([lin[0].weight, w_last], [lin[0].bias, lin[2].bias])
    h256_us = per_launch(lambda: native.score_fused(tu, batches[1][0], ti, batches[1][1], packed256, out=outbuf))
    zipf_gather_us = per_launch(lambda: native.gather_concat(tu, zu, ti, zi, out=gbuf))
    # opt-in variant (BasicNCF.set_fold_first_layer): layer 1 folded into 256-wide tables; reported beside the default
    # line, never as `value` (its results agree with the oracle to 1e-5 but are not bit-identical to the default kernel's)
    fold_variant = None
    if fold_info is None and native.folded_supported(HIDDEN[0], HIDDEN[1]):
        try:
            PAv, PBv, tailv = model._folded(tu, ti, "MLP")
            fv_us = per_launch(lambda: native.score_folded(PAv, batches[1][0], PBv, batches[1][1], tailv, out=outbuf))
            fold_variant = {"fused_us_per_launch": fv_us, "fused_pairs_per_s": B / (fv_us * 1e-6),
                            "executed_flop_per_pair": 2 * (HIDDEN[0] * HIDDEN[1] + HIDDEN[1]),
                            "bytes_per_pair": 2 * HIDDEN[0] * 4 + 2 * 8 + 4,
                            "table_bytes": int(PAv.numel() + PBv.numel()) * 4}
            del PAv, PBv
        except Exception as exc:  # the variant must never take the default line down
            fold_variant = {"error": str(exc)}

    if rank == 0:
        dig_f = profile_digest("ncf::score_fused_f32_kernel<128, 256, 128>") or {}
        dig_g = profile_digest("ncf::gather_concat_vec16") or {}
        total_pairs = world * B * args.steps
        line = {
            "metric": "scored user-item pairs/sec",
            "value": total_pairs / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "cfg2: BasicNCF 1M users x 100k items, emb_dim=64, batch=65536/GPU, fp32, MLP 128-256-128-1"
                                   + (" [first MLP layer FOLDED into 256-wide tables: 65 792 executed FLOP and 2068 B per pair]" if args.fold else ""),
                       "parallelism": f"replicas x{world} (tables+MLP replicated, batch split, no collective)"},
            "roofline": {"kernel": "score_fused_f32_kernel<128,256,128>", "bound": "mfma", "achieved": achieved_tf,
                         "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / PEAK_F32_MFMA_TFLOPS,
                         "traffic": dig_f.get("traffic"), "us_per_launch": fused_us,
                         "algorithmic_flop_per_pair": FLOP_PER_PAIR, "algorithmic_bytes_per_pair": FUSED_BYTES_PER_PAIR,
                         "algorithmic_bytes_per_launch": FUSED_BYTES_PER_PAIR * B,
                         "rocprof_avg_us_isolated": dig_f.get("rocprof_avg_us"), "profile": dig_f.get("profile"),
                         "note": "us_per_launch = HIP events around 100 back-to-back launches (a launch's ramp overlaps the "
                                 "previous launch's tail); rocprof serialises dispatches and reports the isolated duration"},
            "gather_roofline": {"kernel": "gather_concat_vec16<32,1>", "bound": "hbm", "achieved": gather_gbs,
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gather_gbs / PEAK_HBM_GBS,
                                "traffic": dig_g.get("traffic"), "us_per_launch": gather_us,
                                "algorithmic_bytes_per_pair": GATHER_BYTES_PER_PAIR,
                                "algorithmic_bytes_per_launch": GATHER_BYTES_PER_PAIR * B,
                                "rocprof_avg_us_isolated": dig_g.get("rocprof_avg_us"), "profile": dig_g.get("profile")},
        }
        flop256 = 2 * (2 * E * HIDDEN[0] + HIDDEN[0])
        line["variants"] = {
            "zipf_1.05_users": {"fused_us_per_launch": zipf_fused_us, "fused_pairs_per_s": B / (zipf_fused_us * 1e-6),
                                "gather_us_per_launch": zipf_gather_us,
                                "gather_GBps_algorithmic": GATHER_BYTES_PER_PAIR * B / (zipf_gather_us * 1e-6) / 1e9},
            "mlp_256_only": {"fused_us_per_launch": h256_us, "fused_pairs_per_s": B / (h256_us * 1e-6),
                             "flop_per_pair": flop256, "frac_of_fp32_mfma_peak": flop256 * B / (h256_us * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS},
        }
        if fold_variant is not None:
            line["variants"]["folded_first_layer_opt_in"] = fold_variant
        if fold_info is not None:
            line["folded_roofline"] = fold_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model)
            line["cpu_baseline_reference_formulation"] = cpu_baseline_dense_cfg1()
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
