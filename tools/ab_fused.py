"""Dev tool (GPU box): interleaved in-process A/B of ncf_score_fused built with different -D flags.

    python tools/ab_fused.py "" "-DNCF_TOUCH=0" "-DNCF_X_DEPTH=4" ...

Each argument is a flag string; one .so per variant is compiled into gpurun_out/ab/ and timed over interleaved
rounds on the cfg-2 workload (rule: perf deltas only from interleaved rounds in ONE process on ONE device).
"""
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
    lib = os.path.join(out, f"libncf_v{i}.so")
    srcs = [os.path.join(B.HERE, s) for s in ("abi.hip", "mlp_fused.hip", "mlp_bf16.hip", "mlp_bf16_ws8.hip")]
    cmd = [B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", lib] + flags.split() + srcs
    subprocess.check_call(cmd)
    return lib


def main():
    variants = sys.argv[1:] or [""]
    dev = torch.device("cuda:0")
    U, I, E, Bsz = 1_000_000, 100_000, 64, int(os.environ.get("AB_B", 65536))
    g = torch.Generator(device=dev).manual_seed(1)
    tu = torch.randn(U, E, device=dev, generator=g) * 0.05
    ti = torch.randn(I, E, device=dev, generator=g) * 0.05
    dims = [128, 256, 128, 1]
    ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
    batches = [(torch.randint(0, U, (Bsz,), device=dev, generator=g), torch.randint(0, I, (Bsz,), device=dev, generator=g)) for _ in range(16)]
    if os.environ.get("AB_ZIPF"):
        import bench
        batches = [(bench.zipf_indices(U, Bsz, float(os.environ["AB_ZIPF"]), dev, 100 + k), b[1]) for k, b in enumerate(batches)]
    d = (ctypes.c_int * 4)(*dims)
    libs = []
    for i, fl in enumerate(variants):
        path = build_variant(i, fl)
        lib = native.load_library(path) if False else ctypes.CDLL(path)
        for name in ("ncf_mlp_packed_bytes", "ncf_mlp_pack", "ncf_score_fused"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = native.SIGNATURES[name]
        nbytes = lib.ncf_mlp_packed_bytes(0, 3, d)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        W = (ctypes.c_void_p * 3)(*[w.data_ptr() for w in ws])
        Bp = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bs])
        assert lib.ncf_mlp_pack(0, 3, d, W, Bp, blob.data_ptr(), nbytes, None) == 0
        libs.append((lib, blob))
    out = torch.empty(Bsz, 1, device=dev)
    outs = []

    def run(lib, blob, k):
        iu, ii = batches[k % 16]
        rc = lib.ncf_score_fused(0, tu.data_ptr(), U, E, ti.data_ptr(), I, E, iu.data_ptr(), ii.data_ptr(), Bsz, E, E, 3, d,
                                 blob.data_ptr(), out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0

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
        if "NCF_STAMP=1" in fl:
            lib, blob = libs[vi]
            ntiles = Bsz // 32
            dbg = torch.zeros(ntiles * 4, dtype=torch.int64, device=dev)
            lib.ncf_dev_set_debug_buffer.argtypes = [ctypes.c_void_p]
            lib.ncf_dev_set_debug_buffer(dbg.data_ptr())
            for k in range(20):
                run(lib, blob, k)
            torch.cuda.synchronize()
            dd = dbg.view(ntiles, 4).cpu().double() / 100.0   # us
            t0 = dd[:, 2].min()
            for nme, col in (("wave start", dd[:, 2]), ("end of layer 1", dd[:, 1]), ("end of layer 2", dd[:, 0]), ("wave end", dd[:, 3])):
                c = col - t0
                print(f"    {nme:16s}: median {c.median().item():6.2f} us  min {c.min().item():6.2f}  max {c.max().item():6.2f}")
            d1 = dd[:, 1] - dd[:, 2]
            d2 = dd[:, 0] - dd[:, 1]
            print(f"    per wave: start->L1 end median {d1.median().item():.2f} (min {d1.min().item():.2f} max {d1.max().item():.2f}); "
                  f"L2 median {d2.median().item():.2f} (min {d2.min().item():.2f} max {d2.max().item():.2f}); tail {(dd[:,3]-dd[:,0]).median().item():.2f}")
        t = sorted(times[vi])
        med = t[len(t) // 2]
        print(f"variant {vi} [{fl or 'default'}]: median {med:.2f} us  min {t[0]:.2f} us  -> {131328*Bsz/med/1e6:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
