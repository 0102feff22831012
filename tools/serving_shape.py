"""Dev tool (GPU box): the reference web backend's call shape (webapp/backend.py:78-121) at config-3 sizes — ONE user scored
against a whole catalogue tensor, request after request: first request vs later requests (candidate projections kept for
the repeated catalogue tensor) vs the same with a fresh candidate tensor every request."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402
from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    native.load_library()
    I, F, NNZ = 65536, 2094, 256
    g = torch.Generator(device=dev).manual_seed(0)
    catalogue = (torch.rand(I, F, device=dev, generator=g) < 0.02).float()
    torch.manual_seed(0)
    model = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=128, mlp_dense_layers=[256, 128]).to(dev).eval()

    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings

    def user(seed):
        gg = torch.Generator(device=dev).manual_seed(seed)
        rated = torch.randperm(I, device=dev, generator=gg)[:NNZ].sort().values
        vals = torch.randint(1, 11, (NNZ,), device=dev, generator=gg).float() * 0.5 - 2.9
        return SparseRatings(torch.tensor([0, NNZ], device=dev), rated.to(torch.int32), vals, I,
                             pair_row=torch.zeros(I, dtype=torch.int64, device=dev))

    users = [user(k) for k in range(8)]      # built outside the timed region: only the model call is timed

    def request(k, cands):
        return model(cands, catalogue, users[k % 8])

    def timed(fn, n):
        for _ in range(3):
            fn(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for k in range(n):
            fn(k + 1)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    with torch.no_grad():
        model.precompute_catalog(catalogue)
        copies = [catalogue.clone() for _ in range(3)]
        fresh = timed(lambda k: request(k, copies[k % 3]), 12)     # another candidate tensor every request: nothing kept
        same = timed(lambda k: request(k, catalogue), 30)          # the backend's shape: the same tensor every request
        a, b = request(5, catalogue), request(5, copies[0])
        print(f"one user x {I} candidates (F = {F}): {fresh:.0f} us per request with a new candidate tensor each time, "
              f"{same:.0f} us with the catalogue tensor repeated; same scores: {bool(torch.equal(a, b))}")


if __name__ == "__main__":
    main()
