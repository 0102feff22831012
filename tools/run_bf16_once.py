"""Dev tool (GPU box): a few launches of one bf16 scoring kernel for rocprofv3 (--kernel-trace / --pmc).
   python3 tools/run_bf16_once.py <kernel> [B] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

kernel = sys.argv[1] if len(sys.argv) > 1 else "ws8"
Bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 1048576
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
U, I, E = 4_000_000, 1_000_000, 128
g = torch.Generator(device=dev).manual_seed(1)
tu = (torch.randn(U, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
dims = [256, 256, 128, 1]
ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
if os.environ.get("AB_ZERO") == "1":      # all-zero operands: same instruction stream and ids, far less switching power
    tu.zero_(); ti.zero_()
    ws = [w * 0 for w in ws]
packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
iu = torch.randint(0, U, (Bsz,), device=dev, generator=g)
ii = torch.randint(0, I, (Bsz,), device=dev, generator=g)
out = torch.empty(Bsz, 1, device=dev)
native.set_option("bf16_kernel", kernel)
for _ in range(reps):
    native.score_fused(tu, iu, ti, ii, packed, out=out)
torch.cuda.synchronize()
