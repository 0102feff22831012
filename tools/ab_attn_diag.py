"""Dev tool (GPU box): cfg-3-shaped entry-split attention on the diagnostic builds of attn_split.hip (tools/ab_build.sh diagN
attn_split.hip -DATT_SPLIT_DIAG=N), one child process per library: where does the kernel's time go?"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from deeprecommendation_amd import native
    dev = torch.device("cuda:0")
    B, users, nnz, A, F, I = 4096, 64, 256, 128, 64, 100_000
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
    out = []
    for ppw, ns in ((32, 4), (32, 1), (64, 2)):
        grouping = (native.group_pairs(who, users, ppw), ppw)
        fn = lambda: native.attn_forward_grouped(native.ATT_MLP_SCALED, pc, pr, w1, 0.1, rowptr, col, val, who, feat, grouping=grouping, nsplit=ns, leave_partials=True)
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out.append(f"ppw={ppw} nsplit={ns}: {e0.elapsed_time(e1) * 10:6.1f} us")
    print(sys.argv[2], "  ".join(out), flush=True)
    sys.exit(0)
for name in ["", "diag1", "diag2", "diag3", "diag4", "diag5", "diag6"]:
    lib = os.path.join(ROOT, "deeprecommendation_amd", f"libncf_hip_{name}.so" if name else "libncf_hip.so")
    if os.path.exists(lib):
        subprocess.run([sys.executable, __file__, "child", name or "shipped"], env=dict(os.environ, NCF_HIP_LIBRARY=lib))
