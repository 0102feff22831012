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
        for k in range(args.warmup):
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
    for k in range(5):
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
    for k in range(5):
        native.gather_concat(tu, batches[k % N_BATCHES][0], ti, batches[k % N_BATCHES][1], out=gbuf)
    e0.record()
    for k in range(reps):
        native.gather_concat(tu, batches[k % N_BATCHES][0], ti, batches[k % N_BATCHES][1], out=gbuf)
    e1.record()
    torch.cuda.synchronize()
    gather_us = e0.elapsed_time(e1) * 1e3 / reps
    gather_gbs = GATHER_BYTES_PER_PAIR * B / (gather_us * 1e-6) / 1e9

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
        if fold_info is not None:
            line["folded_roofline"] = fold_info
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(model)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
