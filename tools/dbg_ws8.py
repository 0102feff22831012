"""Dev tool (GPU box): which outputs of the 8-wave bf16 kernel differ from the 4-wave kernel's, by unit and position (pipeline debugging)."""
import sys, torch
sys.path.insert(0, '/root/repo')
from deeprecommendation_amd import native
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
E = 128
U, I = 5000, 3000
tu = (torch.randn(U, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
dims = [256, 256, 128, 1]
ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
for B in (64, 128, 64 * 256 * 2, 64 * 256 * 5 + 17):
    iu = torch.randint(0, U, (B,), device=dev, generator=g)
    ii = torch.randint(0, I, (B,), device=dev, generator=g)
    native.set_option("bf16_kernel", "ws")
    a = native.score_fused(tu, iu, ti, ii, packed).clone()
    native.set_option("bf16_kernel", "ws8")
    b = native.score_fused(tu, iu, ti, ii, packed).clone()
    bad = ((a - b).abs() > 1e-4).flatten().nonzero().flatten()
    print("B", B, "bad", bad.numel(), "first", bad[:40].tolist())
    if bad.numel():
        t = bad // 64
        print("  tiles with errors:", torch.unique(t)[:20].tolist(), " in-tile positions:", torch.unique(bad % 64).tolist())
