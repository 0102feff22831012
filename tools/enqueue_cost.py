"""Dev tool (GPU box): is the cfg-2 step bound by the GPU or by the host's enqueue rate?

Times N steps of model(iu, ii) three ways: host time to ENQUEUE them (no sync), wall time until the GPU has finished
them, and the same through native.score_fused on preallocated output (what bench.py's roofline leg times)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from deeprecommendation_amd import native  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    native.load_library()
    model = bench.make_model(dev)
    batches = bench.make_batches(dev, 0)
    n = 2000
    with torch.no_grad():
        for k in range(500):
            model(*batches[k % bench.N_BATCHES])
        torch.cuda.synchronize()
        for label, fn in (("model(iu, ii)", lambda k: model(*batches[k % bench.N_BATCHES])),):
            t0 = time.perf_counter()
            for k in range(n):
                fn(k)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{label}: enqueue {1e6 * (t1 - t0) / n:.1f} us/step, done {1e6 * (t2 - t0) / n:.1f} us/step")
        packed = model._packed_mlp()
        tu, ti = model._table("user", model.user_embeddings[0]), model._table("item", model.item_embeddings[0])
        out = torch.empty((bench.B, 1), device=dev)
        t0 = time.perf_counter()
        for k in range(n):
            b = batches[k % bench.N_BATCHES]
            native.score_fused(tu, b[0], ti, b[1], packed, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"native.score_fused(out=): enqueue {1e6 * (t1 - t0) / n:.1f} us/step, done {1e6 * (t2 - t0) / n:.1f} us/step")
        # host-only cost of the same calls while the GPU is idle-ish: tiny batch
        small = (batches[0][0][:32].contiguous(), batches[0][1][:32].contiguous())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            model(*small)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"model() on 32 pairs (host cost per call): {1e6 * (t1 - t0) / n:.1f} us")


if __name__ == "__main__":
    main()
