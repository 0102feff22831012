"""Dev tool (GPU box): the persistent row-streaming GEMM built with different -D flags (ablations: NCF_RSP_ABLATE=1 no C
stores, 2 = W fragments loaded once per tile, 3 = both), timed interleaved in one process on 1.1 M x 128 x 128."""
import ctypes, os, subprocess, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native
from deeprecommendation_amd.csrc import build as B

dev = torch.device("cuda:0")
M, K, N = (int(x) for x in os.environ.get("AB_SHAPE", "1100000,128,128").split(","))
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(M, K, device=dev, generator=g)
w = torch.randn(N, K, device=dev, generator=g) / K ** 0.5
b = torch.randn(N, device=dev, generator=g)
out = torch.empty(M, N, device=dev)
libs = []
for i, fl in enumerate(sys.argv[1:] or [""]):
    so = os.path.join(ROOT, "gpurun_out", "ab", f"liblin_v{i}.so")
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call([B._hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-shared", "-o", so] + fl.split()
                          + [os.path.join(B.HERE, s) for s in ("abi.hip", "linear.hip")])
    lib = ctypes.CDLL(so)
    lib.ncf_linear_forward.restype, lib.ncf_linear_forward.argtypes = native.SIGNATURES["ncf_linear_forward"]
    libs.append((fl, lib))


def run(lib):
    rc = lib.ncf_linear_forward(0, x.data_ptr(), M, K, w.data_ptr(), b.data_ptr(), K, N, 0, out.data_ptr(), N, torch.cuda.current_stream().cuda_stream)
    assert rc == 0


times = [[] for _ in libs]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for r in range(6):
    for i, (fl, lib) in enumerate(libs):
        for _ in range(5):
            run(lib)
        e0.record()
        for _ in range(20):
            run(lib)
        e1.record()
        torch.cuda.synchronize()
        times[i].append(e0.elapsed_time(e1) * 1e3 / 20)
for i, (fl, lib) in enumerate(libs):
    t = sorted(times[i])[len(times[i]) // 2]
    print(f"[{fl or 'default'}]: {t:8.1f} us  {2.0*M*K*N/t/1e6:6.1f} TF  {(4.0*M*(K+N))/t/1e3:6.0f} GB/s", flush=True)
