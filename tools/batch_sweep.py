"""Dev tool (GPU box): per-launch time of the fused kernels vs batch size (asymptotic efficiency beyond B = 65536)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)


def run(dtype, E, hidden, U=4_000_000, I=1_000_000):
    tu = (torch.randn(U, E, device=dev, generator=g) * 0.05).to(dtype)
    ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(dtype)
    dims = [2 * E] + hidden + [1]
    ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(len(dims) - 1)]
    packed = native.PackedMLP(ws, bs, dtype=dtype)
    flop = 2 * sum(dims[i] * dims[i + 1] for i in range(len(dims) - 1))
    peak = 157.3 if dtype == torch.float32 else 2500.0
    for B in (4096, 16384, 65536, 262144, 1048576, 4194304):
        iu = torch.randint(0, U, (B,), device=dev, generator=g)
        ii = torch.randint(0, I, (B,), device=dev, generator=g)
        out = torch.empty(B, 1, device=dev)
        for _ in range(3):
            native.score_fused(tu, iu, ti, ii, packed, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps):
            native.score_fused(tu, iu, ti, ii, packed, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        tf = flop * B / us / 1e6
        print(f"{str(dtype):16s} E={E} MLP{dims} B={B:8d}: {us:9.1f} us  {B/us:8.1f} Mpairs/s  {tf:7.1f} TFLOP/s = {100*tf/peak:5.1f}% of peak  {(2*E*tu.element_size()+20)*B/us/1e3:7.0f} GB/s")


run(torch.bfloat16, 128, [256, 128])
run(torch.float32, 128, [256, 128])
run(torch.float32, 64, [256, 128])
