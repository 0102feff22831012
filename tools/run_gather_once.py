"""Dev tool (GPU box): launches of the standalone gather (cfg-2 shape) for rocprofv3 --kernel-trace --stats, back to back or one at a
time (a synchronize after each): python3 tools/run_gather_once.py b2b|sync [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "b2b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
U, I, E, B = 1_000_000, 100_000, 64, 65536
tu = torch.randn(U, E, device=dev, generator=g)
ti = torch.randn(I, E, device=dev, generator=g)
batches = [(torch.randint(0, U, (B,), device=dev, generator=g), torch.randint(0, I, (B,), device=dev, generator=g)) for _ in range(16)]
out = torch.empty(B, 2 * E, device=dev)
for k in range(300):
    native.gather_concat(tu, batches[k % 16][0], ti, batches[k % 16][1], out=out)
torch.cuda.synchronize()
for k in range(reps):
    native.gather_concat(tu, batches[k % 16][0], ti, batches[k % 16][1], out=out)
    if mode == "sync":
        torch.cuda.synchronize()
torch.cuda.synchronize()
