"""Dev tool (GPU box): row-streaming GEMM, one tile per wave (NCF_LINEAR_KERNEL=rs) vs the persistent form whose A ring
runs across tiles (=rsp), for the tall shapes of the hot path; checks that both give the same result."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)


def per_launch(fn, reps=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


SHAPES = os.environ.get("AB_SHAPES")
shapes = [tuple(int(v) for v in t.split("x")) for t in SHAPES.split(",")] if SHAPES else None
for M, K, N in shapes or ((1_100_000, 128, 128), (65536, 128, 256), (65536, 256, 128), (65536, 256, 256), (100_000, 128, 128), (16384, 2096, 64), (65536, 64, 128)):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    res = {}
    for mode in ("rs", "rsp"):
        os.environ["NCF_LINEAR_KERNEL"] = mode
        out = native.linear(x, w, b)
        res[mode] = (per_launch(lambda: native.linear(x, w, b)), out)
    flop = 2.0 * M * K * N
    byt = 4.0 * (M * K + M * N)
    md2 = float((res["rs"][1] - res["rsp"][1]).abs().max())
    print(f"M={M:8d} K={K:5d} N={N:4d}: rs {res['rs'][0]:8.1f} us ({flop/res['rs'][0]/1e6:6.1f} TF, {byt/res['rs'][0]/1e3:6.0f} GB/s)   "
          f"rsp {res['rsp'][0]:8.1f} us ({flop/res['rsp'][0]/1e6:6.1f} TF, {byt/res['rsp'][0]/1e3:6.0f} GB/s)   maxdiff={md2:.2e}", flush=True)
