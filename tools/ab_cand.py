"""Dev tool (GPU box): ncf_attn_candidates (one launch: candidate ItemEmbeddings + candidate half of AttentionNet.0 + grouping) against
the three launches it replaces (two ncf_linear_forward + ncf_group_pairs): max relative difference, grouping equality, us per call."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")


def per_launch(fn, reps=100):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def case(B=4096, K=2094, N1=64, N2=128, users=64, ppw=32):
    g = torch.Generator(device=dev).manual_seed(5)
    x = (torch.rand(B, K, device=dev, generator=g) < 0.02).float() + torch.rand(B, K, device=dev, generator=g) * 0.01
    Wi = torch.randn(N1, K, device=dev, generator=g) / K ** 0.5
    bi = torch.randn(N1, device=dev, generator=g) * 0.1
    Wc = (torch.randn(N2, N1, device=dev, generator=g) / N1 ** 0.5).contiguous()
    b0 = torch.randn(N2, device=dev, generator=g) * 0.1
    who = torch.randint(0, users, (B,), device=dev, generator=g)
    e_ref = native.linear(x, Wi, bi)
    p_ref = native.linear(e_ref, Wc, b0)
    e64 = x.double() @ Wi.double().t() + bi.double()
    p64 = e64 @ Wc.double().t() + b0.double()
    emb, pc, grp = native.attn_candidates(x, Wi, bi, Wc, b0, who, users, ppw)
    de = ((emb.double() - e64).abs().max() / e64.abs().max()).item()
    dp = ((pc.double() - p64).abs().max() / p64.abs().max()).item()
    de0 = ((e_ref.double() - e64).abs().max() / e64.abs().max()).item()
    dp0 = ((p_ref.double() - p64).abs().max() / p64.abs().max()).item()
    g0 = native.group_pairs(who, users, ppw)
    same = torch.equal(g0[0], grp[0]) and torch.equal(g0[2], grp[2]) and torch.equal(g0.wg_row[:int(g0[2][-1])], grp.wg_row[:int(grp[2][-1])])
    # pair lists: the same SET per row
    ok_sets = True
    gp = g0[0].tolist()
    for r in range(users):
        a, b = sorted(g0[1][gp[r]:gp[r + 1]].tolist()), sorted(grp[1][gp[r]:gp[r + 1]].tolist())
        ok_sets &= a == b
    emb2, pc2, _ = native.attn_candidates(x, Wi, bi, Wc, b0, who, users, ppw)
    t_new = per_launch(lambda: native.attn_candidates(x, Wi, bi, Wc, b0, who, users, ppw))
    t_new_ng = per_launch(lambda: native.attn_candidates(x, Wi, bi, Wc, b0))
    wpk = native.PackedCandidateWeight(Wi)
    wpk.use_packed = True                  # also where the wrapper would pick the LDS-staged kernel (N1 = 128, large batches)
    emb3, pc3, grp3 = native.attn_candidates(x, wpk, bi, Wc, b0, who, users, ppw)
    t_pk = per_launch(lambda: native.attn_candidates(x, wpk, bi, Wc, b0, who, users, ppw))
    t_pk_ng = per_launch(lambda: native.attn_candidates(x, wpk, bi, Wc, b0))
    print(f"   packed Wi: {t_pk:6.1f} us (no grouping {t_pk_ng:6.1f}) bit-identical {torch.equal(emb3, emb) and torch.equal(pc3, pc)}", flush=True)
    t_l1 = per_launch(lambda: native.linear(x, Wi, bi))
    t_l2 = per_launch(lambda: native.linear(e_ref, Wc, b0))
    t_g = per_launch(lambda: native.group_pairs(who, users, ppw))
    print(f"B={B} K={K} N1={N1} N2={N2}: fused {t_new:6.1f} us (no grouping {t_new_ng:6.1f}) | linear1 {t_l1:6.1f} + linear2 {t_l2:6.1f} + group {t_g:6.1f} = {t_l1 + t_l2 + t_g:6.1f} | "
          f"rel err vs f64: emb {de:.1e} (old {de0:.1e}) pc {dp:.1e} (old {dp0:.1e}) | grouping equal {same and ok_sets} | repeatable {torch.equal(emb, emb2) and torch.equal(pc, pc2)}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "prof":        # under rocprofv3: the cfg-3 shape only, one form per process
        B, K, N1, N2, users, ppw = 4096, 2094, 64, 128, 64, int(os.environ.get("AB_PPW", "32"))
        g = torch.Generator(device=dev).manual_seed(5)
        x = (torch.rand(B, K, device=dev, generator=g) < 0.02).float()
        Wi = torch.randn(N1, K, device=dev, generator=g) / K ** 0.5
        bi = torch.randn(N1, device=dev, generator=g) * 0.1
        Wc = (torch.randn(N2, N1, device=dev, generator=g) / N1 ** 0.5).contiguous()
        b0 = torch.randn(N2, device=dev, generator=g) * 0.1
        who = torch.randint(0, users, (B,), device=dev, generator=g)
        wpk = native.PackedCandidateWeight(Wi)
        for _ in range(60):
            if sys.argv[2] == "packed":
                native.attn_candidates(x, wpk, bi, Wc, b0, who, users, ppw)
            elif sys.argv[2] == "packed_ng":
                native.attn_candidates(x, wpk, bi, Wc, b0)
            elif sys.argv[2] == "fused":
                native.attn_candidates(x, Wi, bi, Wc, b0, who, users, ppw)
            elif sys.argv[2] == "fused_ng":
                native.attn_candidates(x, Wi, bi, Wc, b0)
            else:
                e = native.linear(x, Wi, bi)
                native.linear(e, Wc, b0)
                native.group_pairs(who, users, ppw)
        torch.cuda.synchronize()
        sys.exit(0)
    case()
    case(B=4100, K=2094)
    case(B=512, K=2094, N1=128, N2=128, users=40)
    case(B=4096, K=2094, N1=128, N2=128, users=64)
    case(B=16384, K=2094, N1=128, N2=128, users=64)
    case(B=16384, K=2094, N1=64, N2=128, users=64)
    case(B=4096, K=1030)
    case(B=8192, K=2094)
    case(B=333, K=50, N1=64, N2=16, users=7)
    case(B=4096, K=64, N1=64, N2=256)
