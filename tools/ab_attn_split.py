"""Dev tool (GPU box): the entry-split grouped attention kernel (attn_split.hip) against round 2's scalar-operand kernel and the
per-pair kernel, interleaved in one process: us per launch and max relative difference, for a few shapes / ppw / nsplit."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402
from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings  # noqa: E402

dev = torch.device("cuda:0")


def per_launch(fn, reps=60):
    for _ in range(10):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def case(B=4096, users=64, nnz=256, A=128, F=64, I=100_000, splits=(1, 2, 4), ppws=(32, 64)):
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
    M = native.ATT_MLP_SCALED
    ex = SparseRatings(rowptr, col, val, I, pair_row=who).expanded()
    ref, _ = native.attn_forward(M, pc, pr, w1, 0.1, ex.rowptr, ex.col, ex.val, feat)
    t_pp = per_launch(lambda: native.attn_forward(M, pc, pr, w1, 0.1, ex.rowptr, ex.col, ex.val, feat), reps=20)
    res = [f"B={B} users={users} nnz={nnz} A={A} F={F}: per-pair {t_pp:6.1f} |"]
    for ppw in ppws:
        grouping = (native.group_pairs(who, users, ppw), ppw)
        res.append(f"ppw={ppw}")
        if ppw <= 32:
            native.set_option("attn_grouped_kernel", "scalar")
            t = per_launch(lambda: native.attn_forward_grouped(M, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping))
            o = native.attn_forward_grouped(M, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping)
            d = ((o - ref).abs().max() / ref.abs().max()).item()
            res.append(f"r2-scalar {t:6.1f} ({d:.0e})")
            native.set_option("attn_grouped_kernel", "auto")
        for ns in splits:
            t = per_launch(lambda: native.attn_forward_grouped(M, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping, nsplit=ns))
            o2 = native.attn_forward_grouped(M, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping, nsplit=ns)
            o3 = native.attn_forward_grouped(M, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping, nsplit=ns)
            d = ((o2 - ref).abs().max() / ref.abs().max()).item()
            res.append(f"split{ns} {t:6.1f} ({d:.0e}{'' if torch.equal(o2, o3) else ' NONDET'})")
    ns = native.default_attn_nsplit(B, users, col.numel(), native.default_pairs_per_wg(B))
    res.append(f"| default ppw={native.default_pairs_per_wg(B)} nsplit={ns}")
    print("  ".join(res), flush=True)


if __name__ == "__main__":
    case()
    case(nnz=64)
    case(nnz=1000, splits=(1, 4, 8))
    case(A=64)
    case(F=128)
    case(users=4)
    case(users=1024, splits=(1, 2, 4))
    case(B=16384, users=64, splits=(1, 2))
    case(B=65536, users=1, nnz=256, splits=(1,))
