"""Dev tool (GPU box, diagnostic library built with -DATT_SC_STAMP): where a tile of the scalar-operand grouped attention kernel
spends its cycles (s_memtime sums of wave 0 of every workgroup; shares, not absolute times — the stamps forbid overlaps)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("NCF_HIP_LIBRARY", os.path.join(ROOT, "deeprecommendation_amd", "libncf_hip_stamp.so"))
from deeprecommendation_amd import native  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
B, users, nnz, A, F, I = int(os.environ.get("AB_B", 4096)), int(os.environ.get("AB_USERS", 64)), 256, 128, 64, 100_000
f = 2.0 ** -native.ATT_SCALE_LOG2
pr = torch.randn(I, A, device=dev, generator=g) * 0.3 * f
pc = torch.randn(B, A, device=dev, generator=g) * 0.3 * f
feat = torch.randn(I, F, device=dev, generator=g)
w1 = torch.randn(A, device=dev, generator=g) * 0.2 / f
col = torch.stack([torch.randperm(I, device=dev, generator=g)[:nnz].sort().values for _ in range(users)]).reshape(-1).to(torch.int32)
val = torch.randint(1, 11, (users * nnz,), device=dev, generator=g).float() * 0.5 - 2.9
rowptr = torch.arange(0, (users + 1) * nnz, nnz, device=dev, dtype=torch.int64)
who = torch.randint(0, users, (B,), device=dev, generator=g)
ppw = 16
grp_ptr, pair_ids, wg_ptr = native.group_pairs(who, users, ppw)
lib = native.load_library()
out = torch.empty((B, F), device=dev)
nblk = (B + ppw - 1) // ppw + min(users, B)
dbg = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
args = (native.ATT_MLP_SCALED, pc.data_ptr(), A, pr.data_ptr(), A, A, w1.data_ptr(), 0.1, rowptr.data_ptr(), col.data_ptr(), val.data_ptr(),
        users, I, grp_ptr.data_ptr(), pair_ids.data_ptr(), wg_ptr.data_ptr(), B, ppw, feat.data_ptr(), F, F, None, out.data_ptr(), F,
        None, dbg.data_ptr(), torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    rc = lib.ncf_attn_forward_grouped(*args)
    assert rc == 0, lib.ncf_last_error()
torch.cuda.synchronize()
d = dbg.view(-1, 8).cpu().double()
d = d[d.sum(1) > 0]
names = ["0 wait pr(t) + barrier", "1 issue feat DMA + col/val loads", "2 row reads (LDS -> regs)", "3 barrier + issue pr(t+1) DMA",
         "4 score loop (scalar loads + VALU)", "5 epilogue (max, exp, P, scl)", "6 wait feat + barrier", "7 rescale + MFMA"]
tot = d.sum(1).mean()
print(f"workgroups {d.shape[0]}, cycles per workgroup (wave 0, all tiles) {tot:.0f}")
for i, n in enumerate(names):
    print(f"  {n:40s} {d[:, i].mean():9.0f} cycles  {100 * d[:, i].mean() / tot:5.1f} %")
