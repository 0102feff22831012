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
    outs = {}
    for ppw in (8, 16, 32):
        grouping = (native.group_pairs(who, users, ppw), ppw)
        for kern in ("lds", "scalar"):   # first form vs scalar-operand form, interleaved in one process
            native.set_option("attn_grouped_kernel", kern)
            t = per_launch(lambda: native.attn_forward_grouped(native.ATT_MLP, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping))
            outs[kern] = native.attn_forward_grouped(native.ATT_MLP, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping)
            res.append(f"ppw={ppw} {kern}: {t:7.1f} us")
            if kern == "scalar":                               # the same with operands scaled by 2^-64 / 2^64: relu as a clamp
                f = 2.0 ** -native.ATT_SCALE_LOG2
                pcs, prs, w1s = pc * f, pr * f, w1 / f
                t = per_launch(lambda: native.attn_forward_grouped(native.ATT_MLP_SCALED, pcs, prs, w1s, 0.1, rowptr, col, val, who, feat, grouping=grouping))
                o3 = native.attn_forward_grouped(native.ATT_MLP_SCALED, pcs, prs, w1s, 0.1, rowptr, col, val, who, feat, grouping=grouping)
                res.append(f"scaled/clamp: {t:7.1f} us (bit-equal to unscaled: {torch.equal(o3, outs['scalar'])})")
        if "scalar" in outs:
            d = (outs["lds"] - outs["scalar"]).abs().max().item() / outs["lds"].abs().max().item()
            res.append(f"(rel diff {d:.1e})")
    native.set_option("attn_grouped_kernel", "auto")
    print("  ".join(res), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    # a handful of launches of each form at cfg 3 (for rocprofv3 --pmc passes)
    g = torch.Generator(device=dev).manual_seed(1)
    B, users, nnz, A, F, I = 4096, 64, 256, 128, 64, 100_000
    pr = torch.randn(I, A, device=dev, generator=g) * 0.3
    pc = torch.randn(B, A, device=dev, generator=g) * 0.3
    feat = torch.randn(I, F, device=dev, generator=g)
    w1 = torch.randn(A, device=dev, generator=g) * 0.2
    col = torch.stack([torch.randperm(I, device=dev, generator=g)[:nnz].sort().values for _ in range(users)]).reshape(-1).to(torch.int32)
    val = torch.randint(1, 11, (users * nnz,), device=dev, generator=g).float() * 0.5 - 2.9
    rowptr = torch.arange(0, (users + 1) * nnz, nnz, device=dev, dtype=torch.int64)
    who = torch.randint(0, users, (B,), device=dev, generator=g)
    ppw = int(os.environ.get("AB_PPW", "16"))
    grouping = (native.group_pairs(who, users, ppw), ppw)
    f = 2.0 ** -native.ATT_SCALE_LOG2
    pcs, prs, w1s = pc * f, pr * f, w1 / f
    for kern, ppw in (("lds", 32), ("scalar", 32)):        # round 1's kernel and this round's, each at its own best group size
        grouping = (native.group_pairs(who, users, ppw), ppw)
        native.set_option("attn_grouped_kernel", kern)
        for _ in range(20):
            native.attn_forward_grouped(native.ATT_MLP_SCALED, pcs, prs, w1s, 0.1, rowptr, col, val, who, feat, grouping=grouping)
    torch.cuda.synchronize()
    sys.exit(0)
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
