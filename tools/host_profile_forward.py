"""Dev tool (GPU box): where the HOST time of an eager AttentionNCF forward goes (cProfile over 300 forwards of the reference's
evaluation shape; the GPU work is asynchronous, so cumulative times are enqueue costs)."""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF  # noqa: E402

dev = torch.device("cuda:0")
F, I, B, users = 2094, 1174, 512, 64
IE = int(os.environ.get("HP_IE", "128"))
torch.manual_seed(21)
model = AttentionNCF(item_dim=F, item_emb=IE, user_emb=IE, att_dense=128, mlp_dense_layers=[256, 128]).eval().to(dev)
g = torch.Generator(device=dev).manual_seed(22)
rated = (torch.rand(I, F, device=dev, generator=g) < 0.02).float()
rows = torch.zeros(users, I, device=dev)
mask = torch.rand(users, I, device=dev, generator=g) < 0.125
rows[mask] = (torch.randint(1, 11, (users, I), device=dev, generator=g).float() * 0.5 - 2.9)[mask]
who = torch.arange(B, device=dev) // (B // users)
cands = [rated[torch.randint(0, I, (B,), device=dev, generator=g)].contiguous() for _ in range(4)]
um = rows[who].contiguous()
with torch.no_grad():
    for k in range(20):
        model(cands[k % 4], rated, um)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for k in range(300):
        model(cands[k % 4], rated, um)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
