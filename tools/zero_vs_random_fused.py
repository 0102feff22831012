"""Dev tool (GPU box): the fp32 fused scoring kernel (cfg-2 shape) on random and on all-zero operands — same instruction stream and
ids; the difference is the clock the chip holds (a power / DVFS check)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
U, I, E, B = 1_000_000, 100_000, 64, 65536
dims = [128, 256, 128, 1]
ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
batches = [(torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g)) for _ in range(16)]
out = torch.empty(B, 1, device=dev)
res = {}
for mode in ("random", "zero", "random", "zero"):
    tu = torch.randn(U, E, device=dev, generator=g) * 0.05
    ti = torch.randn(I, E, device=dev, generator=g) * 0.05
    w2 = ws
    if mode == "zero":
        tu.zero_(); ti.zero_()
        w2 = [w * 0 for w in ws]
    packed = native.PackedMLP(w2, bs)
    for k in range(400):
        native.score_fused(tu, batches[k % 16][0], ti, batches[k % 16][1], packed, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(5):
        e0.record()
        for k in range(200):
            native.score_fused(tu, batches[k % 16][0], ti, batches[k % 16][1], packed, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 5)
    print(f"{mode:6s}: {sorted(ts)[2]:.2f} us per launch", flush=True)
