"""Dev tool (GPU box): the skinny-deep Linear (AttentionNCF's 4096 x 2094 -> 64 candidate layer) with 4 and 8 K-slices, timed back to back."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
for M, K, N in ((4096, 2094, 64), (2048, 2094, 64), (4096, 2094, 128), (8192, 2094, 64), (4096, 1030, 64)):
    x = torch.randn(M, K, device=dev, generator=g)
    w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
    b = torch.randn(N, device=dev, generator=g)
    ref = (x.double() @ w.double().t() + b.double()).float()
    for ks in ("4", "8", "auto"):
        native.set_option("linear_kslices", ks)
        out = native.linear(x, w, b)
        err = (out - ref).abs().max().item()
        for _ in range(20):
            native.linear(x, w, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for r in range(5):
            e0.record()
            for _ in range(100):
                native.linear(x, w, b)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 10)
        print(f"{M} x {K} -> {N}, kslices {ks:4s}: {sorted(ts)[2]:7.2f} us   max err {err:.2e}", flush=True)
native.set_option("linear_kslices", "auto")
