"""Dev tool (GPU box): A/B of ncf_linear under different values of one environment switch.

    AB_VAR=NCF_LINEAR_KS AB_VALUES=4,8,16 AB_SHAPES=4096x2094x64,4096x64x128 python tools/ab_linear_env.py

Interleaved rounds in one process; prints the median per-launch time of every value and the max |diff| against the
first value and against a float64 product."""
import os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
var = os.environ["AB_VAR"]
values = os.environ["AB_VALUES"].split(",")
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["AB_SHAPES"].split(",")]


def per_launch(fn, reps=40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for M, K, N in shapes:
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    ref = (x.double() @ w.double().t() + b.double())
    times = {v: [] for v in values}
    outs = {}
    for rnd in range(7):
        for v in values:
            if v == "-":
                os.environ.pop(var, None)
            else:
                os.environ[var] = v
            if rnd == 0:
                outs[v] = native.linear(x, w, b)
                per_launch(lambda: native.linear(x, w, b), 20)
            times[v].append(per_launch(lambda: native.linear(x, w, b)))
    os.environ.pop(var, None)
    line = "  ".join(f"{var}={v}: {statistics.median(times[v]):7.1f} us (d0 {float((outs[v] - outs[values[0]]).abs().max()):.1e}, "
                     f"rel64 {float(((outs[v] - ref).abs().max() / ref.abs().max())):.1e})" for v in values)
    print(f"M={M} K={K} N={N}: {line}", flush=True)
