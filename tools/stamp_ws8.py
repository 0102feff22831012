"""Dev tool (GPU box): phase stamps of the 8-wave bf16 kernel (diagnostic build, -DNCF_BF16_STAMP=1).
   bash tools/ab_build.sh stamp mlp_bf16.hip,mlp_bf16_ws8.hip -DNCF_BF16_STAMP=1 && NCF_HIP_LIBRARY=.../libncf_hip_stamp.so python tools/stamp_ws8.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    U, I, E, Bsz = 4_000_000, 1_000_000, 128, int(os.environ.get("AB_B", 1048576))
    g = torch.Generator(device=dev).manual_seed(1)
    tu = (torch.randn(U, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    dims = [256, 256, 128, 1]
    ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
    packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
    iu = torch.randint(0, U, (Bsz,), device=dev, generator=g)
    ii = torch.randint(0, I, (Bsz,), device=dev, generator=g)
    out = torch.empty(Bsz, 1, device=dev)
    lib = native.load_library()
    dbg = torch.zeros(256 * 512, dtype=torch.int64, device=dev)
    lib.ncf_dev_set_bf16_debug_buffer.restype = None
    lib.ncf_dev_set_bf16_debug_buffer.argtypes = [ctypes.c_void_p]
    native.set_option("bf16_kernel", "ws8")
    lib.ncf_dev_set_bf16_debug_buffer(None)
    for _ in range(50):
        native.score_fused(tu, iu, ti, ii, packed, out=out)
    lib.ncf_dev_set_bf16_debug_buffer(dbg.data_ptr())
    native.score_fused(tu, iu, ti, ii, packed, out=out)
    torch.cuda.synchronize()
    lib.ncf_dev_set_bf16_debug_buffer(None)
    d = dbg.view(256, 8, 8, 8).cpu().double()       # block, wave, phase 16..23, stamp
    a, b = d[:, :4], d[:, 4:]
    per = a[:, :, 1:, 0] - a[:, :, :-1, 0]
    print(f"phase period: median {per.median().item():.0f} cycles, mean {per.mean().item():.0f}  (each stamp costs the wave ~150-200 cycles)")
    for nm, x, i0, i1 in (("A stream (32 or 64 MFMAs + pack)", a, 0, 1), ("A locate + row DMAs", a, 1, 4), ("A vmcnt wait", a, 4, 2), ("A barrier wait", a, 2, 3),
                          ("B head + k-steps 0..5 (locate, ids DMA, row DMAs at 1 and 4)", b, 0, 4), ("B k-steps 6..15", b, 4, 1),
                          ("B vmcnt wait", b, 1, 2), ("B barrier wait", b, 2, 3)):
        v = x[..., i1] - x[..., i0]
        print(f"{nm}: median {v.median().item():.0f}, mean {v.mean().item():.0f}")
    print(f"B phase start - A phase start: median {(b[..., 0].mean(1) - a[..., 0].mean(1)).median().item():.0f}")


if __name__ == "__main__":
    main()
