"""Dev tool (GPU box): interleaved in-process A/B of ncf_score_folded built with different -D flags (cfg-2 shape)."""
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
    lib = os.path.join(out, f"libfold_v{i}.so")
    srcs = [os.path.join(B.HERE, s) for s in ("abi.hip", "mlp_fused.hip", "mlp_bf16.hip", "mlp_bf16_ws8.hip")]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", lib] + flags.split() + srcs)
    return lib


def main():
    variants = sys.argv[1:] or [""]
    dev = torch.device("cuda:0")
    U, I, N1, N2, Bsz = 1_000_000, 100_000, 256, 128, 65536
    g = torch.Generator(device=dev).manual_seed(1)
    PA = torch.randn(U, N1, device=dev, generator=g) * 0.3
    PB = torch.randn(I, N1, device=dev, generator=g) * 0.3
    ws = [torch.randn(N2, N1, device=dev, generator=g) / 16, torch.randn(1, N2, device=dev, generator=g) / 11]
    bs = [torch.randn(N2, device=dev, generator=g) * 0.1, torch.randn(1, device=dev, generator=g)]
    batches = [(torch.randint(0, U, (Bsz,), device=dev, generator=g), torch.randint(0, I, (Bsz,), device=dev, generator=g)) for _ in range(16)]
    d = (ctypes.c_int * 3)(N1, N2, 1)
    libs = []
    for i, fl in enumerate(variants):
        lib = ctypes.CDLL(build_variant(i, fl))
        for name in ("ncf_mlp_packed_bytes", "ncf_mlp_pack", "ncf_score_folded"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = native.SIGNATURES[name]
        nbytes = lib.ncf_mlp_packed_bytes(0, 2, d)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        W = (ctypes.c_void_p * 2)(*[w.data_ptr() for w in ws])
        Bp = (ctypes.c_void_p * 2)(*[b.data_ptr() for b in bs])
        assert lib.ncf_mlp_pack(0, 2, d, W, Bp, blob.data_ptr(), nbytes, None) == 0
        libs.append((lib, blob))
    out = torch.empty(Bsz, 1, device=dev)

    def run(lib, blob, k):
        iu, ii = batches[k % 16]
        rc = lib.ncf_score_folded(0, PA.data_ptr(), U, N1, PB.data_ptr(), I, N1, iu.data_ptr(), ii.data_ptr(), Bsz, N1, N2,
                                  blob.data_ptr(), out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    outs = []
    for lib, blob in libs:
        run(lib, blob, 0)
        torch.cuda.synchronize()
        outs.append(out.clone())
    for o in outs[1:]:
        print("max |diff| vs variant 0:", (o - outs[0]).abs().max().item())
    reps, rounds = 50, 8
    times = [[] for _ in libs]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(rounds):
        for vi, (lib, blob) in enumerate(libs):
            for k in range(5):
                run(lib, blob, k)
            e0.record()
            for k in range(reps):
                run(lib, blob, k)
            e1.record()
            torch.cuda.synchronize()
            times[vi].append(e0.elapsed_time(e1) * 1e3 / reps)
    for vi, fl in enumerate(variants):
        t = sorted(times[vi])
        med = t[len(t) // 2]
        print(f"variant {vi} [{fl or 'default'}]: median {med:.2f} us  min {t[0]:.2f} us  -> {65792*Bsz/med/1e6:.1f} TFLOP/s executed, {2068*Bsz/med/1e3:.0f} GB/s")


if __name__ == "__main__":
    main()
