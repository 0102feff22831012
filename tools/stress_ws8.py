"""Dev tool (GPU box): repeated runs of the 8-wave bf16 kernel against the 4-wave kernel over many batch sizes / seeds (race hunting:
the kernel's hand-counted vmcnt waits and one-barrier-per-phase pipeline must hold for every units-per-workgroup count)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    bad = 0
    n = 0
    for E in (128, 64):
        g = torch.Generator(device=dev).manual_seed(E)
        U, I = 3_000_000, 700_000          # beyond the caches: rows arrive late, like config 5's
        tu = (torch.randn(U, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
        dims = [2 * E, 256, 128, 1]
        ws = [torch.randn(dims[k + 1], dims[k], device=dev, generator=g) / dims[k] ** 0.5 for k in range(3)]
        bs = [torch.randn(dims[k + 1], device=dev, generator=g) * 0.1 for k in range(3)]
        packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
        sizes = [1, 31, 32, 33, 64, 8191, 8192, 8193] + [32 * cus * k + r for k in (1, 2, 3, 4, 5, 6, 7, 9, 16, 40) for r in (0, 17, -15)] + [1 << 20]
        for r in range(rounds):
            for B in sizes:
                iu = torch.randint(0, U, (B,), device=dev, generator=g)
                ii = torch.randint(0, I, (B,), device=dev, generator=g)
                native.set_option("bf16_kernel", "ws")
                ref = native.score_fused(tu, iu, ti, ii, packed).clone()
                native.set_option("bf16_kernel", "ws8")
                for rep in range(3):
                    out = native.score_fused(tu, iu, ti, ii, packed)
                    n += 1
                    err = (out - ref).abs().max().item()
                    if not err <= 2e-6 * max(1.0, ref.abs().max().item()):
                        bad += 1
                        print(f"MISMATCH E={E} B={B} round {r} rep {rep}: max err {err:.3e}", flush=True)
        del tu, ti
        torch.cuda.empty_cache()
    native.set_option("bf16_kernel", "auto")
    print(f"stress_ws8: {n} launches, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
