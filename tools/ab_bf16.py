"""Dev tool (GPU box): interleaved in-process A/B of the bf16 fused kernel (cfg-5 shape: E = 128, MLP 256-256-128-1,
B = 65536) built with different -D flags."""
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
    lib = os.path.join(out, f"libbf16_v{i}.so")
    srcs = [os.path.join(B.HERE, s) for s in ("abi.hip", "mlp_fused.hip", "mlp_bf16.hip", "mlp_bf16_ws8.hip")]
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", lib] + flags.split() + srcs)
    return lib


def main():
    variants = sys.argv[1:] or [""]
    dev = torch.device("cuda:0")
    U, I, E, Bsz = int(os.environ.get("AB_U", 4_000_000)), int(os.environ.get("AB_I", 1_000_000)), 128, int(os.environ.get("AB_B", 65536))
    g = torch.Generator(device=dev).manual_seed(1)
    tu = torch.empty(U, E, dtype=torch.bfloat16, device=dev)
    for s0 in range(0, U, 4_000_000):   # chunked: a 100 M-row table is 25.6 GB in bf16, its fp32 source would be twice that
        tu[s0:s0 + 4_000_000] = (torch.randn(min(4_000_000, U - s0), E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    ti = (torch.randn(I, E, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    dims = [256, 256, 128, 1]
    ws = [torch.randn(dims[i + 1], dims[i], device=dev, generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], device=dev, generator=g) * 0.1 for i in range(3)]
    batches = [(torch.randint(0, U, (Bsz,), device=dev, generator=g), torch.randint(0, I, (Bsz,), device=dev, generator=g)) for _ in range(16)]
    d = (ctypes.c_int * 4)(*dims)
    libs = []
    for i, fl in enumerate(variants):
        lib = ctypes.CDLL(build_variant(i, fl))
        for name in ("ncf_mlp_packed_bytes", "ncf_mlp_pack", "ncf_score_fused"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = native.SIGNATURES[name]
        nbytes = lib.ncf_mlp_packed_bytes(1, 3, d)
        blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        W = (ctypes.c_void_p * 3)(*[w.data_ptr() for w in ws])
        Bp = (ctypes.c_void_p * 3)(*[b.data_ptr() for b in bs])
        assert lib.ncf_mlp_pack(1, 3, d, W, Bp, blob.data_ptr(), nbytes, None) == 0
        libs.append((lib, blob))
    out = torch.empty(Bsz, 1, device=dev)

    def run(lib, blob, k):
        iu, ii = batches[k % 16]
        rc = lib.ncf_score_fused(1, tu.data_ptr(), U, E, ti.data_ptr(), I, E, iu.data_ptr(), ii.data_ptr(), Bsz, E, E, 3, d,
                                 blob.data_ptr(), out.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
        assert rc == 0

    outs = []
    for lib, blob in libs:
        run(lib, blob, 0)
        torch.cuda.synchronize()
        outs.append(out.clone())
    for o in outs[1:]:
        print("max |diff| vs variant 0:", (o - outs[0]).abs().max().item())
    reps, rounds = 100, 8
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
        if "NCF_BF16_STAMP=1" in fl:
            lib, blob = libs[vi]
            ws = "NCF_BF16_WS=0" not in fl
            ntiles = 256 * 8 if ws else Bsz // 32
            dbg = torch.zeros(ntiles * 8 + 256, dtype=torch.int64, device=dev)
            lib.ncf_dev_set_bf16_debug_buffer.argtypes = [ctypes.c_void_p]
            lib.ncf_dev_set_bf16_debug_buffer(dbg.data_ptr())
            for k in range(10):
                run(lib, blob, k)
            torch.cuda.synchronize()
            dd = dbg[:ntiles * 8].view(ntiles, 8).cpu().double() / 100.0  # us (100 MHz realtime counter)
            if ws:
                d3 = dd.view(256, 8, 8)
                start = dbg[ntiles * 8:].cpu().double() / 100.0
                names = ["top", "A (L1 ct0)", "B (L1 ct1|pack0)", "alpha", "C (L2 ct0|pack1)+beta", "D (L2 ct1|dot0)", "E (dot1, red)"]
                raw = dbg[:ntiles * 8].view(256, 8, 8).cpu()
                for a_, b_ in ((0, 3), (4, 7)):
                    if int(raw[:, b_, 0].max()) > 0:
                        dc = (raw[:, b_, 7] - raw[:, a_, 7]).double()
                        dt = (raw[:, b_, 0] - raw[:, a_, 0]).double()
                        ok = dt > 0
                        print(f"    in-kernel clock over iterations {a_ if a_ < 4 else a_ + 96}..{b_ if b_ < 4 else b_ + 96}: "
                              f"{(dc[ok] / dt[ok] * 100).median().item():.0f} MHz")
                for slot in range(8):
                    if float(d3[:, slot, 0].max()) == 0:
                        continue
                    row = d3[:, slot, :]
                    rel = row[:, :7] - start[:, None]
                    seg = rel[:, 1:7] - rel[:, 0:6]
                    print(f"    iteration {slot if slot < 4 else slot + 96}: top at {rel[:, 0].median().item():7.2f} us after start; "
                          + "  ".join(f"{n} +{seg[:, i].median().item():.2f}" for i, n in enumerate(names[1:])))
            else:
                t0 = dd[:, 0].min()
                names = ["start", "after prologue barrier", "after layer 1", "after pack", "after layer 2", "end"]
                for i, nme in enumerate(names):
                    col = dd[:, i] - t0
                    print(f"    {nme:24s}: median {col.median().item():6.2f} us  min {col.min().item():6.2f}  max {col.max().item():6.2f}")
        t = sorted(times[vi])
        med = t[len(t) // 2]
        print(f"variant {vi} [{fl or 'default'}]: median {med:.2f} us  min {t[0]:.2f} us  -> {196864*Bsz/med/1e6:.0f} TFLOP/s, {532*Bsz/med/1e3:.0f} GB/s")


if __name__ == "__main__":
    main()
