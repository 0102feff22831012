"""Dev tool (GPU box, under rocprofv3 --kernel-trace --stats): a handful of launches of the entry-split attention at cfg-3 shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")
B, users, nnz, A, F, I = 4096, 64, 256, 128, 64, 100_000
ppw = int(os.environ.get("AB_PPW", "32"))
ns = int(os.environ.get("AB_NSPLIT", "4"))
g = torch.Generator(device=dev).manual_seed(1)
f = 2.0 ** -native.ATT_SCALE_LOG2
pr = torch.randn(I, A, device=dev, generator=g) * 0.3 * f
pc = torch.randn(B, A, device=dev, generator=g) * 0.3 * f
feat = torch.randn(I, F, device=dev, generator=g)
w1 = torch.randn(A, device=dev, generator=g) * 0.2 / f
col = torch.stack([torch.randperm(I, device=dev, generator=g)[:nnz].sort().values for _ in range(users)]).reshape(-1).to(torch.int32)
val = torch.randint(1, 11, (users * nnz,), device=dev, generator=g).float() * 0.5 - 2.9
rowptr = torch.arange(0, (users + 1) * nnz, nnz, device=dev, dtype=torch.int64)
who = torch.randint(0, users, (B,), device=dev, generator=g)
if os.environ.get("AB_EVEN") == "1":      # exactly B / users pairs per user: no partly filled groups
    who = torch.arange(B, device=dev) % users
grouping = (native.group_pairs(who, users, ppw), ppw)
for _ in range(int(os.environ.get("AB_REPS", "40"))):
    native.attn_forward_grouped(native.ATT_MLP_SCALED, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping, nsplit=ns, leave_partials=True)
torch.cuda.synchronize()
