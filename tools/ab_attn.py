"""Dev tool (GPU box): interleaved in-process A/B of ncf_attn_forward built with different -D flags (cfg-3 shape)."""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402
from deeprecommendation_amd.csrc import build as B  # noqa: E402


def build_variant(i, flags):
    out = os.path.join(ROOT, "gpurun_out", "ab")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, f"libattn_v{i}.so")
    srcs = [os.path.join(B.HERE, s) for s in ("abi.hip", "attn.hip")]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", lib] + flags.split() + srcs)
    return lib


def main():
    variants = sys.argv[1:] or [""]
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(7)
    I, Bsz, nnz, A, UE = 100_000, 4096, 256, 128, 64
    pr = torch.randn(I, A, device=dev, generator=g) * 0.3
    pc = torch.randn(Bsz, A, device=dev, generator=g) * 0.3
    feat = torch.randn(I, UE, device=dev, generator=g)
    w1 = torch.randn(A, device=dev, generator=g) * 0.2
    col = torch.stack([torch.randperm(I, device=dev, generator=g)[:nnz].sort().values for _ in range(64)])
    col = col[torch.randint(0, 64, (Bsz,), device=dev, generator=g)].reshape(-1).to(torch.int32).contiguous()
    val = torch.randint(1, 11, (Bsz * nnz,), device=dev, generator=g).float() * 0.5 - 2.9
    rowptr = torch.arange(0, (Bsz + 1) * nnz, nnz, device=dev, dtype=torch.int64)
    out = torch.empty(Bsz, UE, device=dev)
    wts = torch.empty(Bsz * nnz, device=dev)
    libs = []
    for i, fl in enumerate(variants):
        lib = ctypes.CDLL(build_variant(i, fl))
        lib.ncf_attn_forward.restype, lib.ncf_attn_forward.argtypes = native.SIGNATURES["ncf_attn_forward"]
        libs.append(lib)

    def run(lib):
        rc = lib.ncf_attn_forward(0, pc.data_ptr(), A, pr.data_ptr(), A, A, w1.data_ptr(), 0.1, rowptr.data_ptr(), col.data_ptr(),
                                  val.data_ptr(), Bsz, I, feat.data_ptr(), UE, UE, None, out.data_ptr(), UE, wts.data_ptr(),
                                  torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    outs = []
    for lib in libs:
        run(lib)
        torch.cuda.synchronize()
        outs.append(out.clone())
    for o in outs[1:]:
        print("max |diff| vs variant 0:", (o - outs[0]).abs().max().item())
    reps, rounds = 30, 8
    times = [[] for _ in libs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        for vi, lib in enumerate(libs):
            for k in range(3):
                run(lib)
            e0.record()
            for k in range(reps):
                run(lib)
            e1.record()
            torch.cuda.synchronize()
            times[vi].append(e0.elapsed_time(e1) * 1e3 / reps)
    for vi, fl in enumerate(variants):
        t = sorted(times[vi])
        med = t[len(t) // 2]
        print(f"variant {vi} [{fl or 'default'}]: median {med:.1f} us  min {t[0]:.1f} us  -> {Bsz*nnz*(A*4+UE*4+20)/med/1e3:.0f} GB/s gathered")


if __name__ == "__main__":
    main()
