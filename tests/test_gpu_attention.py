"""GPU parity tests for K3 (AttentionNCF item-item attention) against reference goldens and the CPU oracle."""
import os

import pytest
import torch

from conftest import load_golden
from oracle import ncf_oracle as O
from test_gpu_basic import assert_close

pytestmark = pytest.mark.gpu


def _model(kw, state, gpu):
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    m = AttentionNCF(**kw)
    m.load_state_dict(state)
    return m.eval().to(gpu)


@pytest.mark.parametrize("name", ["g3_att_dense8", "g3_att_none", "g3_att_cos", "g3_att_vec64", "g3_att_vec128"])
def test_attention_golden(gpu, name):
    state, a, kw = load_golden(name)
    m = _model(kw, state, gpu)
    with torch.no_grad():
        out, att = m(torch.from_numpy(a["candidate_items"]).to(gpu), torch.from_numpy(a["rated_items"]).to(gpu),
                     torch.from_numpy(a["user_matrix"]).to(gpu), return_attention_weights=True)
        out2 = m(torch.from_numpy(a["candidate_items"]).to(gpu), torch.from_numpy(a["rated_items"]).to(gpu),
                 torch.from_numpy(a["user_matrix"]).to(gpu))
    assert_close(out, torch.from_numpy(a["out"]))
    assert_close(att, torch.from_numpy(a["att"]))
    assert torch.equal(out, out2)
    if name.startswith("g3_att_vec"):
        return
    assert float(att[1].abs().sum()) == 0.0   # user without ratings: zeros, not NaN (attention_ncf.py:208-209)
    assert float(att[2, 3]) == 0.0            # exact-zero entry is unrated (attention_ncf.py:158)


def test_attention_shipped_checkpoint(gpu):
    """G4 with the reference's trained weights — only where /root/reference is mounted (never on the GPU box, since
    reference content does not travel); kept so that a GPU-equipped build container exercises it."""
    _, a, kw = load_golden("g4_att_shipped_ckpt")
    ck = os.path.join("/root/reference", str(a["ckpt_relpath"]))
    if not os.path.exists(ck):
        pytest.skip("reference checkpoint not present on this host")
    state, kwargs = torch.load(ck, map_location="cpu", weights_only=True)
    m = _model(kwargs, state, gpu)
    with torch.no_grad():
        out, att = m(torch.from_numpy(a["candidate_items"]).to(gpu), torch.from_numpy(a["rated_items"]).to(gpu),
                     torch.from_numpy(a["user_matrix"]).to(gpu), return_attention_weights=True)
    assert_close(out, torch.from_numpy(a["out"]))
    assert_close(att, torch.from_numpy(a["att"]))


def _random_case(B, I, Fdim, IE, UE, A, hidden, density, seed, **kw):
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    torch.manual_seed(seed)
    m = AttentionNCF(item_dim=Fdim, item_emb=IE, user_emb=UE, att_dense=A, mlp_dense_layers=hidden, **kw).eval()
    g = torch.Generator().manual_seed(seed + 1)
    rated = torch.rand(I, Fdim, generator=g)
    cand = torch.rand(B, Fdim, generator=g)
    um = torch.zeros(B, I)
    mask = torch.rand(B, I, generator=g) < density
    um[mask] = (torch.randint(1, 11, (B, I), generator=g).float() * 0.5 - 2.9)[mask]
    um[0] = 0.0
    return m, cand, rated, um


@pytest.mark.parametrize("cfg", [
    dict(B=64, I=300, Fdim=50, IE=64, UE=64, A=128, hidden=[256, 128], density=0.3),      # cfg-3 shaped (vector paths)
    dict(B=33, I=70, Fdim=37, IE=24, UE=40, A=10, hidden=[32], density=0.5),             # ragged: generic paths
    dict(B=17, I=129, Fdim=64, IE=128, UE=128, A=128, hidden=[256, 128], density=0.9),   # shipped-model dims
    dict(B=9, I=40, Fdim=20, IE=16, UE=16, A=None, hidden=[16], density=0.4),
    dict(B=9, I=40, Fdim=20, IE=64, UE=16, A=None, hidden=[16], density=0.4, use_cos_sim_instead=True),
])
def test_attention_vs_oracle(gpu, cfg):
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings
    cfg = dict(cfg)
    m, cand, rated, um = _random_case(seed=5, **cfg)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    ref_out, ref_att = O.attention_ncf_forward(state, cand, rated, um, use_cos_sim_instead=m.use_cos_sim_instead,
                                               return_attention_weights=True)
    m.to(gpu)
    with torch.no_grad():
        out, att = m(cand.to(gpu), rated.to(gpu), um.to(gpu), return_attention_weights=True)
        # CSR input == dense input, bit for bit
        out_csr = m(cand.to(gpu), rated.to(gpu), SparseRatings.from_dense(um.to(gpu)))
    assert_close(out, ref_out)
    assert_close(att, ref_att)
    assert torch.equal(out, out_csr)


def test_attention_weights_are_a_distribution(gpu):
    """Size-independent properties at cfg-3 scale (100k-item catalogue, 256 rated per user, B = 4096): every
    non-empty row's weights are positive and sum to 1; permuting the batch permutes the outputs bit-exactly."""
    from deeprecommendation_amd import native
    g = torch.Generator(device=gpu).manual_seed(7)
    I, B, nnz, A, UE = 100_000, 4096, 256, 128, 64
    pr = torch.randn(I, A, device=gpu, generator=g) * 0.3
    pc = torch.randn(B, A, device=gpu, generator=g) * 0.3
    feat = torch.randn(I, UE, device=gpu, generator=g)
    w1 = torch.randn(A, device=gpu, generator=g) * 0.2
    col = torch.stack([torch.randperm(I, device=gpu, generator=g)[:nnz].sort().values for _ in range(64)])
    col = col[torch.randint(0, 64, (B,), device=gpu, generator=g)].reshape(-1).to(torch.int32)
    val = (torch.randint(1, 11, (B * nnz,), device=gpu, generator=g).float() * 0.5 - 2.9)
    rowptr = torch.arange(0, (B + 1) * nnz, nnz, device=gpu, dtype=torch.int64)
    out, w = native.attn_forward(native.ATT_MLP, pc, pr, w1, 0.1, rowptr, col, val, feat)
    sums = w.view(B, nnz).sum(1)
    assert float((sums - 1).abs().max()) < 1e-5 and float(w.min()) >= 0.0
    # oracle on a slice
    sub = slice(0, 8)
    cc = col.view(B, nnz)[sub].long().cpu()
    sc = (torch.relu(pc[sub].cpu()[:, None, :] + pr.cpu()[cc]) * w1.cpu()).sum(-1) + 0.1
    ww = torch.softmax(sc.double(), 1)
    ref = ((ww * val.view(B, nnz)[sub].cpu().double())[:, :, None] * feat.cpu()[cc].double()).sum(1).float()
    assert_close(out[sub], ref)
    perm = torch.randperm(B, device=gpu, generator=g)
    out_p, _ = native.attn_forward(native.ATT_MLP, pc[perm].contiguous(), pr, w1, 0.1, rowptr,
                                   col.view(B, nnz)[perm].reshape(-1).contiguous(), val.view(B, nnz)[perm].reshape(-1).contiguous(), feat)
    assert torch.equal(out_p, out[perm])


def test_serving_shape_one_user_whole_catalogue(gpu):
    """The reference web backend's call shape (webapp/backend.py:78-121): candidates = every unseen item of the
    catalogue, rated_items = the user's rated items only, user_matrix = the same row repeated B times,
    return_attention_weights=True; top-k by score must agree with the oracle."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    torch.manual_seed(21)
    F, n_items, n_rated, k = 96, 1248, 37, 10
    m = AttentionNCF(item_dim=F, item_emb=128, user_emb=128, att_dense=128, mlp_dense_layers=[256, 128]).eval()
    state = {k_: v.clone() for k_, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    feats = (torch.rand(n_items, F, generator=g) < 0.1).float() + torch.rand(n_items, F, generator=g) * 0.05
    rated_idx = torch.randperm(n_items, generator=g)[:n_rated].sort().values
    ratings = torch.randint(1, 11, (n_rated,), generator=g).float() * 0.5
    unseen = torch.ones(n_items, dtype=torch.bool)
    unseen[rated_idx] = False
    cand = feats[unseen]
    rated = feats[rated_idx]
    row = ratings - (ratings.mean() + 2.5) / 2
    um = row.unsqueeze(0).repeat(cand.shape[0], 1)
    ref_out, ref_att = O.attention_ncf_forward(state, cand, rated, um, return_attention_weights=True)
    m.to(gpu)
    with torch.no_grad():
        out, att = m(cand.to(gpu), rated.to(gpu), um.to(gpu), return_attention_weights=True)
    assert_close(out, ref_out)
    assert_close(att, ref_att)
    assert att.shape == (cand.shape[0], n_rated)
    top_ref = torch.topk(ref_out.view(-1), k).indices
    top = torch.topk(out.view(-1).cpu(), k).indices
    assert torch.equal(top, top_ref)
