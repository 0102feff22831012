"""Dev tool (GPU box): interleaved in-process timing of the bf16 scoring kernels selected through ncf_set_option
(cfg-5 shape: E = 128, MLP 256-256-128-1).   python tools/ab_bf16_opt.py [kernels...] ; AB_B="65536,1048576" AB_U=4000000"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402


def main():
    kernels = sys.argv[1:] or ["ws", "ws8"]
    dev = torch.device("cuda:0")
    U, I, E = int(os.environ.get("AB_U", 4_000_000)), int(os.environ.get("AB_I", 1_000_000)), 128
    sizes = [int(x) for x in os.environ.get("AB_B", "65536,1048576,4194304").split(",")]
    g = torch.Generator(device=dev).manual_seed(1)
    tu = torch.empty(U, E, dtype=torch.bfloat16, device=dev)
    for s0 in range(0, U, 4_000_000):
        tu[s0:s0 + 4_000_000] = (torch.randn(min(4_000_000, U - s0), E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    dims = [256, 256, 128, 1]
    ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
    if os.environ.get("AB_ZERO") == "1":      # all-zero operands: same cycles, far less switching power (a power / clock check)
        tu.zero_(); ti.zero_()
        ws = [w * 0 for w in ws]
    if os.environ.get("AB_ZERO") == "2":      # zero tables only
        tu.zero_(); ti.zero_()
    packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
    flop = 2 * (256 * 256 + 256 * 128 + 128)
    for Bsz in sizes:
        batches = [(torch.randint(0, U, (Bsz,), device=dev, generator=g), torch.randint(0, I, (Bsz,), device=dev, generator=g)) for _ in range(4)]
        out = torch.empty(Bsz, 1, device=dev)
        outs = {}
        for k in kernels:
            native.set_option("bf16_kernel", k)
            native.score_fused(tu, batches[0][0], ti, batches[0][1], packed, out=out)
            torch.cuda.synchronize()
            outs[k] = out.clone()
        for k in kernels[1:]:
            d = (outs[k] - outs[kernels[0]]).abs().max().item()
            print(f"B={Bsz}: max |{k} - {kernels[0]}| = {d:.3e} (scale {outs[kernels[0]].abs().max().item():.3f})", flush=True)
        reps = max(10, min(200, int(6e7 / Bsz)))
        times = {k: [] for k in kernels}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(6):
            for k in kernels:
                native.set_option("bf16_kernel", k)
                for j in range(3):
                    native.score_fused(tu, batches[j % 4][0], ti, batches[j % 4][1], packed, out=out)
                e0.record()
                for j in range(reps):
                    native.score_fused(tu, batches[j % 4][0], ti, batches[j % 4][1], packed, out=out)
                e1.record()
                torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) * 1e3 / reps)
        for k in kernels:
            t = sorted(times[k])
            med = t[len(t) // 2]
            print(f"B={Bsz} {k:7s}: median {med:9.2f} us  min {t[0]:9.2f}  -> {Bsz * flop / med / 1e6:7.1f} TFLOP/s = {Bsz * flop / med / 1e6 / 2500:.3f} of 2.5 PF; {Bsz / med:.0f} M pairs/s", flush=True)
    native.set_option("bf16_kernel", "auto")


if __name__ == "__main__":
    main()
