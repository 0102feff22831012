"""Dev tool (GPU box): the two forms of ncf_gather_concat (one step per wave / persistent prefetching waves), interleaved in one
process at the cfg-2 shape and a few others; bit-equality of the outputs checked."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")


def case(U, I, E, Bsz, dtype=torch.float32, zipf=False):
    g = torch.Generator(device=dev).manual_seed(1)
    tu = torch.randn(U, E, device=dev, generator=g).to(dtype)
    ti = torch.randn(I, E, device=dev, generator=g).to(dtype)
    batches = [(torch.randint(0, U, (Bsz,), device=dev, generator=g), torch.randint(0, I, (Bsz,), device=dev, generator=g)) for _ in range(16)]
    out = torch.empty(Bsz, 2 * E, device=dev, dtype=dtype)
    elt = 4 if dtype == torch.float32 else 2
    nbytes = (2 * (2 * E * elt) + 16) * Bsz
    res = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for kern in ("step", "persistent"):
        native.set_option("gather_kernel", kern)
        native.gather_concat(tu, batches[3][0], ti, batches[3][1], out=out)
        assert torch.equal(out, torch.cat((tu[batches[3][0]], ti[batches[3][1]]), 1)), kern
    times = {"step": [], "persistent": []}
    for r in range(8):
        for kern in ("step", "persistent"):
            native.set_option("gather_kernel", kern)
            for k in range(20):
                native.gather_concat(tu, batches[k % 16][0], ti, batches[k % 16][1], out=out)
            e0.record()
            for k in range(100):
                native.gather_concat(tu, batches[k % 16][0], ti, batches[k % 16][1], out=out)
            e1.record()
            torch.cuda.synchronize()
            times[kern].append(e0.elapsed_time(e1) * 10)
    native.set_option("gather_kernel", "auto")
    msg = f"U={U} I={I} E={E} B={Bsz} {str(dtype)[6:]}:"
    for kern, t in times.items():
        t = sorted(t)
        med = t[len(t) // 2]
        msg += f"  {kern}: median {med:6.2f} us min {t[0]:6.2f} -> {nbytes / med / 1e3:5.0f} GB/s ({nbytes / med / 8e6 * 100:4.1f} %)"
    print(msg, flush=True)


case(1_000_000, 100_000, 64, 65536)
case(1_000_000, 100_000, 64, 262144)
case(1_000_000, 100_000, 64, 16384)
case(4_000_000, 1_000_000, 128, 65536, torch.bfloat16)
case(1_000_000, 100_000, 128, 65536)
