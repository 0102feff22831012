"""Dev tool (GPU box): per-launch time of the grouped (LDS-tiled) attention kernel vs the per-pair kernel for a few
shapes and pairs-per-workgroup settings (cfg-3 base: 4096 pairs, 64 users, 256 rated, A = 128, Fdim = 64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402
from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings  # noqa: E402

dev = torch.device("cuda:0")


def per_launch(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def case(B=4096, users=64, nnz=256, A=128, F=64, I=100_000):
    g = torch.Generator(device=dev).manual_seed(1)
    pr = torch.randn(I, A, device=dev, generator=g) * 0.3
    pc = torch.randn(B, A, device=dev, generator=g) * 0.3
    feat = torch.randn(I, F, device=dev, generator=g)
    w1 = torch.randn(A, device=dev, generator=g) * 0.2
    col = torch.stack([torch.randperm(I, device=dev, generator=g)[:nnz].sort().values for _ in range(users)]).reshape(-1).to(torch.int32)
    val = torch.randint(1, 11, (users * nnz,), device=dev, generator=g).float() * 0.5 - 2.9
    rowptr = torch.arange(0, (users + 1) * nnz, nnz, device=dev, dtype=torch.int64)
    who = torch.randint(0, users, (B,), device=dev, generator=g)
    sr = SparseRatings(rowptr, col, val, I, pair_row=who)
    ex = sr.expanded()
    t_pp = per_launch(lambda: native.attn_forward(native.ATT_MLP, pc, pr, w1, 0.1, ex.rowptr, ex.col, ex.val, feat))
    res = [f"B={B} users={users} nnz={nnz} A={A} F={F}: per-pair {t_pp:7.1f} us |"]
    for ppw in (8, 16, 32):
        grouping = (native.group_pairs(who, users, ppw), ppw)
        t = per_launch(lambda: native.attn_forward_grouped(native.ATT_MLP, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping))
        res.append(f"ppw={ppw}: {t:7.1f} us")
    print("  ".join(res), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "threshold":
    # pairs per rated set at which the grouped kernel overtakes the per-pair kernel (SparseRatings.GROUPED_MIN_PAIRS_PER_ROW)
    for users in (4096, 2048, 1024, 512, 256, 128):
        case(users=users)
    sys.exit(0)
case()
case(nnz=64)
case(A=64)
case(users=4)
case(B=16384, users=64)
case(B=65536, users=1, nnz=256)
