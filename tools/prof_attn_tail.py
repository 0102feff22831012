"""Dev tool (GPU box, under rocprofv3): launches of ncf_attn_tail at the cfg-3 shape (4096 pairs, 4 slices, E = 64)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")
B, E, ns = 4096, 64, 4
g = torch.Generator(device=dev).manual_seed(1)
cand = torch.randn(B, E, device=dev, generator=g)
W1 = torch.randn(256, 2 * E, device=dev, generator=g) * 0.1
W2 = torch.randn(128, 256, device=dev, generator=g) * 0.1
b1, b2, w3 = torch.randn(256, device=dev, generator=g), torch.randn(128, device=dev, generator=g), torch.randn(128, device=dev, generator=g)
part = torch.randn(B, ns, E + 4, device=dev, generator=g)
part[:, :, 1] = part[:, :, 1].abs() + 0.1
ws = part.view(torch.uint8).view(-1)
ub = torch.randn(E, device=dev, generator=g)
if os.environ.get("AB_TAIL_PACKED", "1") == "1":      # what the model passes: the hidden layers' weights in MFMA operand order
    W1, W2 = native.PackedTailWeight(W1), native.PackedTailWeight(W2)
for _ in range(60):
    native.attn_tail(cand, native.AttnPartials(ws, ns, E, B), ub, W1, b1, W2, b2, w3, 0.1)
torch.cuda.synchronize()
