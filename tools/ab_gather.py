"""Dev tool (GPU box): interleaved in-process A/B of ncf_gather_concat built with different -D flags (cfg 2 shape)."""
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
    lib = os.path.join(out, f"libgather_v{i}.so")
    srcs = [os.path.join(B.HERE, s) for s in ("abi.hip", "gather.hip")]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", lib] + flags.split() + srcs)
    return lib


def main():
    variants = sys.argv[1:] or [""]
    dev = torch.device("cuda:0")
    U, I, E, Bsz = 1_000_000, 100_000, 64, 65536
    g = torch.Generator(device=dev).manual_seed(1)
    tu = torch.randn(U, E, device=dev, generator=g)
    ti = torch.randn(I, E, device=dev, generator=g)
    batches = [(torch.randint(0, U, (Bsz,), device=dev, generator=g), torch.randint(0, I, (Bsz,), device=dev, generator=g)) for _ in range(16)]
    out = torch.empty(Bsz, 2 * E, device=dev)
    libs = []
    for i, fl in enumerate(variants):
        lib = ctypes.CDLL(build_variant(i, fl))
        lib.ncf_gather_concat.restype, lib.ncf_gather_concat.argtypes = native.SIGNATURES["ncf_gather_concat"]
        libs.append(lib)

    def run(lib, k):
        iu, ii = batches[k % 16]
        rc = lib.ncf_gather_concat(0, tu.data_ptr(), U, E, ti.data_ptr(), I, E, iu.data_ptr(), ii.data_ptr(), Bsz, E, E,
                                   out.data_ptr(), 2 * E, None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    for lib in libs:
        run(lib, 3)
        torch.cuda.synchronize()
        assert torch.equal(out, torch.cat((tu[batches[3][0]], ti[batches[3][1]]), 1))
    reps, rounds = 100, 8
    times = [[] for _ in libs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        for vi, lib in enumerate(libs):
            for k in range(5):
                run(lib, k)
            e0.record()
            for k in range(reps):
                run(lib, k)
            e1.record()
            torch.cuda.synchronize()
            times[vi].append(e0.elapsed_time(e1) * 1e3 / reps)
    for vi, fl in enumerate(variants):
        t = sorted(times[vi])
        med = t[len(t) // 2]
        print(f"variant {vi} [{fl or 'default'}]: median {med:.2f} us  min {t[0]:.2f} us  -> {1040*Bsz/med/1e3:.0f} GB/s ({1040*Bsz/med/8e6*100:.1f}% of 8 TB/s)")


if __name__ == "__main__":
    main()
