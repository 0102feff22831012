"""GPU parity tests for K3 (AttentionNCF item-item attention) against reference goldens and the CPU oracle."""
import os

import pytest
import torch

from conftest import load_golden
from oracle import ncf_oracle as O
from test_gpu_basic import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native(gpu):
    from deeprecommendation_amd import native as n
    n.load_library()
    return n


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
    assert_close(out2, torch.from_numpy(a["out"]))          # the plain call converts the dense matrix on the stream and may take other
    assert_close(out, out2)                                 # kernels (attention and MLP) than the call that returns weights: fp32 rounding
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


# ----------------------------------------------------------------------------- grouped (LDS-tiled) form
def _shared_rows(R, I, nnz_max, B, gpu, seed, empty_row=True, bad_cols=False):
    """R rated sets (ragged, one possibly empty) shared by B pairs in random order."""
    g = torch.Generator().manual_seed(seed)
    counts = torch.randint(1, nnz_max + 1, (R,), generator=g)
    if empty_row:
        counts[R // 2] = 0
    rowptr = torch.zeros(R + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    col = torch.cat([torch.randperm(I, generator=g)[:int(c)].sort().values for c in counts]).to(torch.int32)
    if bad_cols:
        col[::7] = I + 3      # out-of-range positions are masked entries (-inf score), never a fault
        col[3::11] = -1
    val = torch.randint(1, 11, (int(rowptr[-1]),), generator=g).float() * 0.5 - 2.9
    pair_row = torch.randint(0, R, (B,), generator=g)
    return rowptr.to(gpu), col.to(gpu), val.to(gpu), pair_row.to(gpu)


@pytest.mark.parametrize("mode", ["mlp", "cos"])
@pytest.mark.parametrize("A,Fdim,R,nnz_max,B,ppw", [
    (128, 64, 9, 300, 500, None),      # cfg-3 shaped: several 64-entry tiles per user, ragged last tile
    (64, 128, 3, 40, 37, 4),           # fewer entries than one tile; two output registers per lane
    (8, 20, 5, 130, 64, 16),           # Fdim not a multiple of 4 (scalar staging), 16 pairs per workgroup
    (256, 256, 2, 70, 9, 1),           # the largest tile the entry accepts (> 64 KB of LDS), one pair per workgroup
])
def test_attention_grouped_equals_per_pair_kernel(native, gpu, mode, A, Fdim, R, nnz_max, B, ppw):
    """ncf_attn_forward_grouped on shared rows == ncf_attn_forward on the expanded per-pair CSR (both within 1e-5 of a
    float64 evaluation), incl. an empty rated set, masked (out-of-range) positions and pairs in random user order."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings
    I = 400
    g = torch.Generator(device=gpu).manual_seed(A + Fdim)
    pr = torch.randn(I, A, device=gpu, generator=g) * 0.4
    pc = torch.randn(B, A, device=gpu, generator=g) * 0.4
    feat = torch.randn(I, Fdim, device=gpu, generator=g)
    bias = torch.randn(Fdim, device=gpu, generator=g)
    w1 = torch.randn(A, device=gpu, generator=g) * 0.3
    m = native.ATT_MLP if mode == "mlp" else native.ATT_COS
    rowptr, col, val, pair_row = _shared_rows(R, I, nnz_max, B, gpu, seed=R * 100 + B, bad_cols=(A == 128))
    out_g, w_g = native.attn_forward_grouped(m, pc, pr, w1 if mode == "mlp" else None, 0.25, rowptr, col, val, pair_row, feat,
                                             out_bias=bias, pairs_per_wg=ppw, return_weights=True)
    out_g2 = native.attn_forward_grouped(m, pc, pr, w1 if mode == "mlp" else None, 0.25, rowptr, col, val, pair_row, feat,
                                         out_bias=bias, pairs_per_wg=ppw)
    # asking for the weights does not change the scores: the weights come from round 2's one-workgroup-per-group kernel, the plain
    # call may take the entry-split kernel (another order of the softmax partial sums): equal to fp32 rounding, not bit for bit
    assert_close(out_g, out_g2)
    ex = SparseRatings(rowptr, col, val, I, pair_row=pair_row).expanded()
    out_p, w_p = native.attn_forward(m, pc, pr, w1 if mode == "mlp" else None, 0.25, ex.rowptr, ex.col, ex.val, feat, out_bias=bias)
    assert w_g.shape == w_p.shape
    assert_close(w_g, w_p)                                  # attention weights, expanded-CSR layout (:224)
    # float64 restatement on the CPU
    pcd, prd, fd, w1d = pc.cpu().double(), pr.cpu().double(), feat.cpu().double(), w1.cpu().double()
    ref = torch.zeros(B, Fdim, dtype=torch.float64)
    rp, cc, vv, prow = rowptr.cpu(), col.cpu().long(), val.cpu().double(), pair_row.cpu()
    for b in range(B):
        r = int(prow[b])
        c, v = cc[rp[r]:rp[r + 1]], vv[rp[r]:rp[r + 1]]
        ok = (c >= 0) & (c < I)
        c, v = c[ok], v[ok]
        if c.numel():
            if mode == "mlp":
                s = (torch.relu(pcd[b][None, :] + prd[c]) * w1d).sum(1) + 0.25
            else:
                s = (pcd[b][None, :] * prd[c]).sum(1)
            w = torch.softmax(s, 0)
            ref[b] = ((w * v)[:, None] * fd[c]).sum(0)
    ref += bias.cpu().double()
    assert_close(out_p, ref.float())
    assert_close(out_g, ref.float())
    empty = (rp[1:] - rp[:-1])[prow] == 0
    if bool(empty.any()):
        assert torch.equal(out_g.cpu()[empty], bias.cpu().expand(int(empty.sum()), Fdim))   # zeros + bias, not NaN


def test_attention_model_takes_the_grouped_path_for_shared_rows(gpu, monkeypatch):
    """AttentionNCF.forward with a shared-row SparseRatings (what SparseDynamicProvider collates) runs the LDS-tiled
    kernel when users repeat — also with return_attention_weights, whose (B, I) weights come from the same kernel — and
    agrees with the dense-matrix call and the oracle."""
    from deeprecommendation_amd import native
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings
    m, cand, rated, um_rows = _random_case(B=6, I=150, Fdim=40, IE=64, UE=64, A=128, hidden=[256, 128], density=0.4, seed=11)
    g = torch.Generator().manual_seed(2)
    B = 96
    pair_row = torch.randint(0, 6, (B,), generator=g)
    cand = torch.rand(B, 40, generator=g)
    um = um_rows[pair_row]                                  # the dense matrix repeats a user's row per pair
    state = {k: v.clone() for k, v in m.state_dict().items()}
    ref_out, ref_att = O.attention_ncf_forward(state, cand, rated, um, return_attention_weights=True)
    m.to(gpu)
    shared_rows = SparseRatings.from_dense(um_rows.to(gpu))
    shared = SparseRatings(shared_rows.rowptr, shared_rows.col, shared_rows.val, shared_rows.num_items, pair_row=pair_row.to(gpu))
    calls = []
    real = native.attn_forward_grouped
    monkeypatch.setattr(native, "attn_forward_grouped", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    with torch.no_grad():
        out_shared = m(cand.to(gpu), rated.to(gpu), shared)
        out_dense = m(cand.to(gpu), rated.to(gpu), um.to(gpu))
        out_w, att = m(cand.to(gpu), rated.to(gpu), shared, return_attention_weights=True)
    assert calls == [1, 1, 1]          # the dense matrix repeats 6 distinct rows 96 times: from_dense shares them too
    assert_close(out_shared, ref_out)
    assert_close(out_dense, ref_out)
    assert_close(out_w, ref_out)
    assert_close(att, ref_att)


def test_attention_grouped_large_batch_matches_per_pair(native, gpu):
    """40 000 pairs over 50 users: the multi-launch grouping path (ncf_group_pairs above 32 768 pairs) and 32 pairs per
    workgroup; the grouped kernel must agree with the per-pair kernel on the expanded CSR, and the grouping must be a
    permutation of the pairs, row by row."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings
    I, A, Fdim, R, B = 500, 64, 64, 50, 40000
    g = torch.Generator(device=gpu).manual_seed(3)
    pr = torch.randn(I, A, device=gpu, generator=g) * 0.4
    pc = torch.randn(B, A, device=gpu, generator=g) * 0.4
    feat = torch.randn(I, Fdim, device=gpu, generator=g)
    w1 = torch.randn(A, device=gpu, generator=g) * 0.3
    rowptr, col, val, pair_row = _shared_rows(R, I, 90, B, gpu, seed=9)
    grp_ptr, pair_ids, wg_ptr = native.group_pairs(pair_row, R, 32)
    assert torch.equal(torch.sort(pair_ids).values, torch.arange(B, device=gpu))
    counts = torch.bincount(pair_row, minlength=R)
    assert torch.equal(grp_ptr[1:] - grp_ptr[:-1], counts)
    assert torch.equal(wg_ptr[1:] - wg_ptr[:-1], (counts + 31) // 32)
    rows_of_listed = pair_row[pair_ids]
    assert bool((rows_of_listed[1:] >= rows_of_listed[:-1]).all())          # listed row by row
    small = native.group_pairs(pair_row[:3000].contiguous(), R, 8)          # single-workgroup path
    assert torch.equal(torch.sort(small[1]).values, torch.arange(3000, device=gpu))
    assert torch.equal(small[0][1:] - small[0][:-1], torch.bincount(pair_row[:3000], minlength=R))
    out_g = native.attn_forward_grouped(native.ATT_MLP, pc, pr, w1, -0.1, rowptr, col, val, pair_row, feat)
    ex = SparseRatings(rowptr, col, val, I, pair_row=pair_row).expanded()
    out_p, _ = native.attn_forward(native.ATT_MLP, pc, pr, w1, -0.1, ex.rowptr, ex.col, ex.val, feat)
    assert_close(out_g, out_p)
    native.check_oob(gpu)


@pytest.mark.parametrize("R", [1, 63, 1000, 1024, 1025, 4096, 4097, 9999, 32768, 40000])
@pytest.mark.parametrize("B,ppw", [(1, 8), (4096, 32), (30000, 16)])
def test_group_pairs_is_a_row_ordered_permutation(native, gpu, R, B, ppw):
    """ncf_group_pairs over the row counts of every dispatch: LDS counters (R <= 4096), global counters (R <= 32768, run
    lengths that do not divide R), the multi-launch path above; rows concentrated on a few ids and spread over all of them.
    grp_ptr / wg_ptr must be the exclusive scans of the row counts / per-row workgroup counts and pair_ids a permutation
    of the pairs listed row by row."""
    g = torch.Generator(device=gpu).manual_seed(R * 31 + B)
    for spread in (min(R, 7), R):
        pair_row = torch.randint(0, spread, (B,), device=gpu, generator=g)
        if spread == R and R > 1:
            pair_row[0], pair_row[-1] = R - 1, 0          # first and last row in use
        grp_ptr, pair_ids, wg_ptr = native.group_pairs(pair_row, R, ppw)
        counts = torch.bincount(pair_row, minlength=R)
        assert int(grp_ptr[0]) == 0 and int(wg_ptr[0]) == 0
        assert torch.equal(grp_ptr[1:] - grp_ptr[:-1], counts)
        assert torch.equal(wg_ptr[1:] - wg_ptr[:-1], (counts + ppw - 1) // ppw)
        assert torch.equal(torch.sort(pair_ids).values, torch.arange(B, device=gpu))
        listed = pair_row[pair_ids]
        assert bool((listed[1:] >= listed[:-1]).all())
    native.check_oob(gpu)


@pytest.mark.parametrize("cos", [False, True])
def test_candidates_as_rows_of_the_catalogue(native, gpu, cos):
    """AttentionNCF.forward with the candidates given as RowsOf(rated_items, index) (embeddings taken from the catalogue
    pass) == the same forward on the materialised feature rows; a RowsOf over ANOTHER table is materialised first."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF, RowsOf, SparseRatings
    from test_gpu_basic import assert_close
    g = torch.Generator().manual_seed(5)
    I, F, B, R = 500, 48, 300, 12
    feats = torch.randn(I, F, generator=g).to(gpu)
    torch.manual_seed(7)
    m = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=None if cos else 32, mlp_dense_layers=[128],
                     use_cos_sim_instead=cos).to(gpu).eval()
    counts = torch.randint(1, 40, (R,), generator=g)
    rowptr = torch.zeros(R + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(counts, 0)
    col = torch.cat([torch.randperm(I, generator=g)[:c].sort().values for c in counts.tolist()]).to(torch.int32)
    val = torch.randn(int(rowptr[-1]), generator=g).abs() + 0.1
    pair_row = torch.randint(0, R, (B,), generator=g)
    cand = torch.randint(0, I, (B,), generator=g).to(gpu)
    ratings = SparseRatings(rowptr.to(gpu), col.to(gpu), val.to(gpu), I, pair_row=pair_row.to(gpu))
    with torch.no_grad():
        dense = m(feats[cand], feats, ratings)
        lazy = m(RowsOf(feats, cand), feats, ratings)
        other = m(RowsOf(feats.clone(), cand), feats, ratings)
    assert_close(lazy, dense)
    assert_close(other, dense)
    assert lazy.shape == (B, 1)


def test_candidate_projections_are_kept_for_a_repeated_candidate_tensor(native, gpu):
    """Serving shape (webapp/backend.py:78-121): every request scores the SAME catalogue tensor for another user.  The
    second call reuses the candidate projections (no F -> IE Linear launched), gives the same scores as a fresh model, and
    an in-place edit of the tensor, a new tensor, or a weight update invalidate the kept entry."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    g = torch.Generator().manual_seed(8)
    I, F, B = 400, 48, 2048
    feats = torch.randn(I, F, generator=g).to(gpu)
    cands = torch.randn(B, F, generator=g).to(gpu)
    torch.manual_seed(4)
    m = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=32, mlp_dense_layers=[128]).to(gpu).eval()
    fresh = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=32, mlp_dense_layers=[128]).to(gpu).eval()
    fresh.load_state_dict(m.state_dict())

    def user_matrix(seed):
        gg = torch.Generator().manual_seed(seed)
        row = torch.where(torch.rand(I, generator=gg) < 0.2, torch.randn(I, generator=gg), torch.zeros(I))
        return row.repeat(B, 1).to(gpu)

    calls = []
    real = native.linear
    def counting(x, *a, **k):
        calls.append(tuple(x.shape))
        return real(x, *a, **k)
    with torch.no_grad():
        first = m(cands, feats, user_matrix(1))
        native.linear = counting
        try:
            second = m(cands, feats, user_matrix(2))
        finally:
            native.linear = real
        assert (B, F) not in calls and (B, 64) not in calls          # neither candidate Linear ran again
        assert torch.equal(second, fresh(cands, feats, user_matrix(2)))
        assert torch.equal(first, fresh(cands, feats, user_matrix(1)))
        cands[0].mul_(2.0)                                           # in-place edit -> recomputed
        assert torch.equal(m(cands, feats, user_matrix(2)), fresh(cands.clone(), feats, user_matrix(2)))
        other = cands.clone()
        assert torch.equal(m(other, feats, user_matrix(3)), fresh(other, feats, user_matrix(3)))
        m.ItemEmbeddings[0].bias.add_(0.5)                          # weight update -> everything derived is dropped
        fresh.load_state_dict(m.state_dict())
        assert torch.equal(m(other, feats, user_matrix(3)), fresh(other.clone(), feats, user_matrix(3)))


def test_cfg3_bench_workload_full_size_grouped_kernel(gpu):
    """The workload bench.py times for BASELINE config 3, at full size and through the model (bench_extra.cfg3_workload: 100 k-row
    catalogue, F = 2094, 4096 pairs of 64 users x 256 rated): the forward takes the LDS-tiled grouped kernel
    (attn_grouped_sc_kernel<NCF_ATT_MLP_SCALED, 32 pairs / workgroup, 8 waves>).  Checked (i) against the per-pair kernel on the
    SAME batch (one CSR row per pair, a different kernel and summation order), 1e-5; (ii) a 16-pair slice against the CPU oracle's
    reference formulation (attention_ncf.py:136-224) over the union of those users' rated items — the catalogue the reference's
    provider would hand over (dynamic_profiles_provider.py:55-71: columns nobody in the batch rated do not exist there, and are
    masked by `user_matrix != 0` here); (iii) with 4096 distinct users (no sharing: the model's per-pair dispatch) the same pairs
    give the same scores."""
    import bench_extra
    from deeprecommendation_amd import native
    I, B, nnz, Fdim, IE, UE, A = bench_extra.CFG3_DIMS
    model, catalogue, batches = bench_extra.cfg3_workload(gpu, users=64, n_batches=1)
    cand, r = batches[0]
    assert r.pair_row is not None and r.pairs_per_row >= 4 and native.default_pairs_per_wg(B) == 32
    with torch.no_grad():
        out_g, att_g = model(cand, catalogue, r, return_attention_weights=True)         # grouped kernel
        rx = r.expanded()
        assert rx.pair_row is None
        out_p, att_p = model(cand, catalogue, rx, return_attention_weights=True)        # per-pair kernel
    assert out_g.shape == (B, 1)
    assert_close(out_g, out_p)
    assert float((att_g - att_p).abs().max()) <= 1e-6
    sums = att_g.sum(1)
    assert float((sums - 1).abs().max()) < 1e-5
    # (ii) oracle on 16 pairs
    sl = torch.arange(0, B, B // 16, device=gpu)[:16]
    rows = r.pair_row[sl]
    cols = r.col.view(64, nnz)[rows].long()                       # (16, 256) catalogue positions
    rated_ids = torch.unique(cols)
    um = torch.zeros(16, rated_ids.numel(), device=gpu)
    um.scatter_(1, torch.searchsorted(rated_ids, cols), r.val.view(64, nnz)[rows])
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ref, ref_att = O.attention_ncf_forward(state, cand[sl].cpu(), catalogue[rated_ids].cpu(), um.cpu(), return_attention_weights=True)
    assert_close(out_g[sl], ref)
    assert_close(att_g[sl][:, rated_ids], ref_att)
    # (iii) 4096 distinct users: every pair has its own rated set (per-pair dispatch inside the model)
    model2, catalogue2, batches2 = bench_extra.cfg3_workload(gpu, users=B, n_batches=1)
    cand2, r2 = batches2[0]
    assert int(torch.unique(r2.pair_row).numel()) == B
    with torch.no_grad():
        out2 = model2(cand2, catalogue2, r2)
        out2x = model2(cand2, catalogue2, r2.expanded())
    assert_close(out2, out2x)
    sl2 = sl[:4]
    rows2 = r2.pair_row[sl2]
    cols2 = r2.col.view(B, nnz)[rows2].long()
    ids2 = torch.unique(cols2)
    um2 = torch.zeros(4, ids2.numel(), device=gpu)
    um2.scatter_(1, torch.searchsorted(ids2, cols2), r2.val.view(B, nnz)[rows2])
    state2 = {k: v.detach().cpu() for k, v in model2.state_dict().items()}
    assert_close(out2[sl2], O.attention_ncf_forward(state2, cand2[sl2].cpu(), catalogue2[ids2].cpu(), um2.cpu()))


@pytest.mark.parametrize("mode_name", ["mlp", "mlp_scaled", "cos"])
@pytest.mark.parametrize("A,Fdim", [(128, 64), (32, 64), (64, 128), (256, 128)])
@pytest.mark.parametrize("ppw,nsplit", [(32, 1), (32, 4), (16, 2), (32, 3), (5, 8), (64, 4), (64, 1), (40, 2)])
def test_entry_split_kernel_vs_per_pair(gpu, mode_name, A, Fdim, ppw, nsplit):
    """ncf_attn_forward_split (round 3: (group of pairs) x (slice of the rated set) workgroups + merge of the softmax partials)
    against the per-pair kernel on the same batch: rows of 0, 1, 63, 64, 65, 300 entries (empty slices, ragged last tiles, more
    slices than tiles), users with one pair and with many, a group size that is not a multiple of 4.  1e-5; bitwise repeatable."""
    from deeprecommendation_amd import native
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings
    g = torch.Generator(device=gpu).manual_seed(A * 7 + Fdim + ppw + nsplit)
    I, B = 700, 333
    lens = torch.tensor([0, 1, 63, 64, 65, 300, 129, 7, 256, 2], device=gpu)
    R = lens.numel()
    rowptr = torch.zeros(R + 1, dtype=torch.int64, device=gpu)
    rowptr[1:] = torch.cumsum(lens, 0)
    col = torch.cat([torch.randperm(I, device=gpu, generator=g)[:int(n)].sort().values for n in lens.tolist()]).to(torch.int32)
    val = torch.randint(1, 11, (col.numel(),), device=gpu, generator=g).float() * 0.5 - 2.9
    who = torch.randint(0, R, (B,), device=gpu, generator=g)
    who[:3] = torch.tensor([0, 1, 5], device=gpu)
    mode = {"mlp": native.ATT_MLP, "mlp_scaled": native.ATT_MLP_SCALED, "cos": native.ATT_COS}[mode_name]
    f = 2.0 ** -native.ATT_SCALE_LOG2 if mode_name == "mlp_scaled" else 1.0
    pr = torch.randn(I, A, device=gpu, generator=g) * 0.3 * f
    pc = torch.randn(B, A, device=gpu, generator=g) * 0.3 * f
    feat = torch.randn(I, Fdim, device=gpu, generator=g)
    w1 = None if mode_name == "cos" else torch.randn(A, device=gpu, generator=g) * 0.2 / f
    bias = torch.randn(Fdim, device=gpu, generator=g)
    assert native.attn_split_supported(mode, A, Fdim, ppw)
    ex = SparseRatings(rowptr, col, val, I, pair_row=who).expanded()
    ref, _ = native.attn_forward(mode, pc, pr, w1, 0.1, ex.rowptr, ex.col, ex.val, feat, out_bias=bias)
    out = native.attn_forward_grouped(mode, pc, pr, w1, 0.1, rowptr, col, val, who, feat, out_bias=bias, pairs_per_wg=ppw, nsplit=nsplit)
    assert_close(out, ref)
    empty = who == 0
    assert torch.equal(out[empty], bias.expand(int(empty.sum()), Fdim))          # no rated entry: weights 0, the bias alone (:208-209)
    out2 = native.attn_forward_grouped(mode, pc, pr, w1, 0.1, rowptr, col, val, who, feat, out_bias=bias, pairs_per_wg=ppw, nsplit=nsplit)
    assert torch.equal(out, out2)


def test_entry_split_kernel_out_of_catalogue_columns_and_default_nsplit(gpu):
    from deeprecommendation_amd import native
    g = torch.Generator(device=gpu).manual_seed(3)
    I, B, A, Fdim, R, nnz = 500, 2048, 128, 64, 16, 200
    rowptr = torch.arange(0, (R + 1) * nnz, nnz, device=gpu, dtype=torch.int64)
    col = torch.stack([torch.randperm(I, device=gpu, generator=g)[:nnz].sort().values for _ in range(R)]).reshape(-1).to(torch.int32)
    col[5] = -1            # columns outside the catalogue are ignored (weight 0), like the per-pair kernel
    col[nnz + 9] = I + 3
    val = torch.randn(R * nnz, device=gpu, generator=g)
    who = torch.randint(0, R, (B,), device=gpu, generator=g)
    pr, pc = torch.randn(I, A, device=gpu, generator=g) * 0.3, torch.randn(B, A, device=gpu, generator=g) * 0.3
    feat, w1 = torch.randn(I, Fdim, device=gpu, generator=g), torch.randn(A, device=gpu, generator=g) * 0.2
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import SparseRatings
    ex = SparseRatings(rowptr, col, val, I, pair_row=who).expanded()
    ref, _ = native.attn_forward(native.ATT_MLP, pc, pr, w1, -0.2, ex.rowptr, ex.col, ex.val, feat)
    assert native.default_attn_nsplit(B, R, col.numel(), native.default_pairs_per_wg(B)) > 1
    out = native.attn_forward_grouped(native.ATT_MLP, pc, pr, w1, -0.2, rowptr, col, val, who, feat)       # default ppw / nsplit
    assert_close(out, ref)


def _dense_batch(gpu, B, I, users, density, seed, with_edge_cases=True):
    g = torch.Generator().manual_seed(seed)
    rows = torch.zeros(users, I)
    mask = torch.rand(users, I, generator=g) < density
    rows[mask] = (torch.randint(1, 11, (users, I), generator=g).float() * 0.5 - 2.9)[mask]
    who = torch.randint(0, users, (B,), generator=g)
    um = rows[who].clone()
    if with_edge_cases and B >= 8:
        um[3] = 0.0                                   # a user with no ratings
        um[5] = um[4]                                 # two pairs with the very same row
        um[6] = um[4]; um[6, 0] = -0.0 if float(um[4, 0]) == 0.0 else um[4, 0]
        um[7] = um[4]; um[7, I - 1] += 0.5            # differs from row 4 in the LAST column only
    return um.to(gpu), who


@pytest.mark.parametrize("B,I,users,share", [(64, 50, 7, True), (512, 1174, 64, True), (512, 1174, 64, False), (300, 3000, 300, True), (1, 9, 1, True)])
def test_dense_to_csr_on_stream(gpu, B, I, users, share):
    """ncf_dense_csr_rows / _fill against the definition: pair b's CSR row lists exactly the non-zero entries of user_matrix[b]
    (attention_ncf.py:158) in column order; with sharing, identical rows use ONE row — that of the smallest pair index — and
    nothing else shares; -0.0 counts as unrated; sizes never touch the host (sync debug mode = error)."""
    from deeprecommendation_amd import native
    um, _ = _dense_batch(gpu, B, I, users, 0.15, seed=B + I)
    if B >= 8:
        um[2, 1] = -0.0
    native.dense_to_csr(um, share)                    # warm-up (allocator, library load)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        rowptr, col, val, pair_row = native.dense_to_csr(um, share)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    rowptr, col, val, pair_row, umc = rowptr.cpu(), col.cpu(), val.cpu(), pair_row.cpu(), um.cpu()
    assert int(rowptr[0]) == 0 and bool((rowptr[1:] >= rowptr[:-1]).all())
    first = {}
    for b in range(B):
        key = tuple(umc[b].tolist())                  # -0.0 == 0.0 as tuple elements? no: compare via the != 0 pattern below
        nz = (umc[b] != 0).nonzero().view(-1)
        r = int(pair_row[b])
        lo, hi = int(rowptr[r]), int(rowptr[r + 1])
        assert torch.equal(col[lo:hi].long(), nz), f"pair {b}"
        assert torch.equal(val[lo:hi], umc[b][nz]), f"pair {b}"
        if share:
            k = (tuple(nz.tolist()), tuple(umc[b][nz].tolist()))
            first.setdefault(k, b)
            assert r == first[k], f"pair {b} must use the row of the first identical pair"
        else:
            assert r == b
    reps = torch.unique(pair_row)
    assert int(rowptr[-1]) == sum(int((umc[int(r)] != 0).sum()) for r in reps)


@pytest.mark.parametrize("name", ["g3_att_dense8", "g3_att_vec64", "g3_att_vec128", "g3_att_cos", "g3_att_none"])
def test_attention_forward_dense_matrix_without_host_synchronisation(gpu, name):
    """The reference's own call shape — model(candidates, rated_items, DENSE user_matrix) — through the on-stream conversion:
    equal to the reference's golden output, and the whole forward enqueues without a single host read
    (torch.cuda.set_sync_debug_mode("error")) once the weight-derived caches exist."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    state, a, kw = load_golden(name)
    m = AttentionNCF(**kw)
    m.load_state_dict(state)
    m = m.eval().to(gpu)
    cand, rated, um = (torch.from_numpy(a[k]).to(gpu) for k in ("candidate_items", "rated_items", "user_matrix"))
    ref = torch.from_numpy(a["out"])
    with torch.no_grad():
        out = m(cand, rated, um)                      # first call: packs weights, projects the catalogue (may read the host)
        assert_close(out, ref)
        cand2, um2 = cand.clone(), um.clone()         # new tensors: no cache of candidate projections applies
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            out2 = m(cand2, rated, um2)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert_close(out2, ref)
        m.dense_user_matrix_on_stream = False         # round 2's conversion (host reads, exact kernel dispatch): same scores
        out3 = m(cand.clone(), rated, um.clone())
        assert_close(out3, out2)


def test_attention_forward_dense_matrix_reference_eval_shape(gpu):
    """B = 512 samples of 64 users (rows repeated as dynamic_datasets.py:24-40 builds them), I = 1174 rated items, F = 2094: the
    reference's evaluation batch.  On-stream conversion + grouped kernels vs the CPU oracle's reference formulation."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    torch.manual_seed(11)
    F, I, B = 2094, 1174, 512
    m = AttentionNCF(item_dim=F, item_emb=128, user_emb=128, att_dense=128, mlp_dense_layers=[256, 128]).eval()
    g = torch.Generator().manual_seed(12)
    rated = (torch.rand(I, F, generator=g) < 0.02).float() + torch.rand(I, F, generator=g) * 0.1
    cand = (torch.rand(B, F, generator=g) < 0.02).float() + torch.rand(B, F, generator=g) * 0.1
    um, _ = _dense_batch(gpu, B, I, 64, 0.125, seed=13)
    state = {k: v.clone() for k, v in m.state_dict().items()}
    sub = slice(0, 48)
    ref = O.attention_ncf_forward(state, cand[sub], rated, um.cpu()[sub])
    m = m.to(gpu)
    with torch.no_grad():
        out = m(cand.to(gpu), rated.to(gpu), um)
    assert_close(out[sub], ref)


@pytest.mark.parametrize("E,B,nsplit", [(64, 4096, 4), (64, 333, 1), (128, 1000, 3), (64, 17, 8), (128, 16, 1)])
def test_attention_tail_kernel(gpu, E, B, nsplit):
    """ncf_attn_tail — merge of the entry-split attention's softmax partials (+ UserEmbeddings' bias), cat(candidate_emb, user_emb)
    (attention_ncf.py:219), MLP (util.py:5-18) in one launch — against a float64 evaluation of the same formulas; also with
    finished user embeddings instead of partials, a ragged last tile, empty slices (m = -inf) and a pair whose slices are ALL empty."""
    from deeprecommendation_amd import native
    g = torch.Generator(device=gpu).manual_seed(E + B + nsplit)
    cand = torch.randn(B, E, device=gpu, generator=g)
    N1, N2 = 256, 128
    W1 = (torch.randn(N1, 2 * E, device=gpu, generator=g) / (2 * E) ** 0.5).contiguous()
    W2 = (torch.randn(N2, N1, device=gpu, generator=g) / N1 ** 0.5).contiguous()
    b1, b2 = torch.randn(N1, device=gpu, generator=g) * 0.1, torch.randn(N2, device=gpu, generator=g) * 0.1
    w3, b3 = torch.randn(N2, device=gpu, generator=g) / N2 ** 0.5, 0.37
    ubias = torch.randn(E, device=gpu, generator=g) * 0.1
    assert native.attn_tail_supported(E, E, N1, N2)

    def mlp64(x):
        h = torch.relu(x @ W1.double().t() + b1.double())
        h = torch.relu(h @ W2.double().t() + b2.double())
        return h @ w3.double().view(-1, 1) + b3

    # finished user embeddings
    user = torch.randn(B, E, device=gpu, generator=g)
    out = native.attn_tail(cand, user, None, W1, b1, W2, b2, w3, b3)
    assert_close(out, mlp64(torch.cat((cand, user), 1).double()))
    # partials
    part = torch.zeros(B, nsplit, E + 4, device=gpu)
    m = torch.randn(B, nsplit, device=gpu, generator=g) * 3
    l = torch.rand(B, nsplit, device=gpu, generator=g) * 5 + 0.1
    O_ = torch.randn(B, nsplit, E, device=gpu, generator=g) * 2
    if nsplit > 1:
        m[::5, 1] = -float("inf")                      # an empty slice
        l[::5, 1] = 0.0
        O_[::5, 1] = 0.0
    m[3] = -float("inf"); l[3] = 0.0; O_[3] = 0.0      # a pair with no rated entry at all: bias alone (the nan_to_num case)
    part[:, :, 0], part[:, :, 1], part[:, :, 4:] = m, l, O_
    ws = part.view(torch.uint8).view(-1)
    M = m.double().max(dim=1, keepdim=True).values
    w = torch.where(torch.isinf(m), torch.zeros_like(m.double()), torch.exp(m.double() - M))
    w = torch.nan_to_num(w, nan=0.0)
    L = (l.double() * w).sum(1, keepdim=True)
    ue = torch.where(L > 0, (O_.double() * w[:, :, None]).sum(1) / L.clamp_min(1e-300), torch.zeros(B, E, dtype=torch.float64, device=gpu)) + ubias.double()
    out2 = native.attn_tail(cand, native.AttnPartials(ws, nsplit, E, B), ubias, W1, b1, W2, b2, w3, b3)
    assert_close(out2, mlp64(torch.cat((cand.double(), ue), 1)))
    assert torch.equal(out2, native.attn_tail(cand, native.AttnPartials(ws, nsplit, E, B), ubias, W1, b1, W2, b2, w3, b3))
    # the hidden layers' weights packed in MFMA operand order (what the model passes): the same values in the same order, bit for bit
    P1, P2 = native.PackedTailWeight(W1), native.PackedTailWeight(W2)
    assert torch.equal(out2, native.attn_tail(cand, native.AttnPartials(ws, nsplit, E, B), ubias, P1, b1, P2, b2, w3, b3))
    assert torch.equal(out, native.attn_tail(cand, user, None, P1, b1, P2, b2, w3, b3))
    with pytest.raises(TypeError):
        native.attn_tail(cand, user, None, P1, b1, W2, b2, w3, b3)


@pytest.mark.parametrize("B,K,N1,N2,users", [(4096, 2094, 64, 128, 64), (100, 2094, 128, 128, 7), (33, 96, 64, 16, 3), (17, 31, 64, 32, 17),
                                               (512, 1000, 128, 256, 1), (1, 65, 64, 64, 1), (2100, 300, 128, 64, 9), (2048, 77, 64, 128, 30),
                                               (700, 2094, 64, 128, 11)])
def test_attn_candidates_both_weight_forms(gpu, B, K, N1, N2, users):
    """ncf_attn_candidates / ncf_attn_candidates_packed (candidate ItemEmbeddings + candidate half of AttentionNet.0 + the listing of
    the pairs by rated set in one launch): both against an fp64 product, bit-identical to each other where the packed form is one
    launch (B > 2048; below that its K range is cut over workgroups), grouping = ncf_group_pairs_rows'.
    Ragged K (2094 = 65 * 32 + 14; 31 < one step), B not a multiple of the 16-row tile, both widths of ItemEmbeddings."""
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(B + K)
    x = ((torch.rand(B, K, generator=g) < 0.1).float() + torch.rand(B, K, generator=g) * 0.1).to(gpu)
    Wi = (torch.randn(N1, K, generator=g) / K ** 0.5).to(gpu)
    bi = (torch.randn(N1, generator=g) * 0.1).to(gpu)
    Wc = (torch.randn(N2, N1, generator=g) / N1 ** 0.5).to(gpu)
    b0 = (torch.randn(N2, generator=g) * 0.1).to(gpu)
    who = torch.randint(0, users, (B,), generator=g).to(gpu)
    ppw = 32
    e64 = x.double() @ Wi.double().t() + bi.double()
    p64 = e64 @ Wc.double().t() + b0.double()
    emb, pc, grp = native.attn_candidates(x, Wi, bi, Wc, b0, who, users, ppw)
    wpk = native.PackedCandidateWeight(Wi)
    wpk.use_packed = True                                       # N1 = 128, large batch: the wrapper prefers the unpacked kernel; test the packed one anyway
    emb2, pc2, grp2 = native.attn_candidates(x, wpk, bi, Wc, b0, who, users, ppw)
    emb3, pc3, none = native.attn_candidates(x, wpk, bi, Wc, b0)
    assert none is None
    assert_close(emb, e64.float())
    assert_close(pc, p64.float())
    assert torch.equal(emb3, emb2) and torch.equal(pc3, pc2)    # with and without the grouping workgroup
    if B > 2048:                                                # one launch: the LDS-staged form's k pairing, slices and order
        assert torch.equal(emb2, emb) and torch.equal(pc2, pc)
    else:                                                       # at most 128 row tiles: K range cut over workgroups, pieces added in order
        assert_close(emb2, e64.float())
        assert_close(pc2, p64.float())
    ref = native.group_pairs(who, users, ppw)
    n_wg = int(ref[2][-1])
    for gg in (grp, grp2):
        assert torch.equal(gg[0], ref[0]) and torch.equal(gg[2], ref[2]) and torch.equal(gg.wg_row[:n_wg], ref.wg_row[:n_wg])
        gp = ref[0].tolist()
        for r in range(users):                                  # the same SET of pairs per row (their order comes from atomics)
            assert sorted(gg[1][gp[r]:gp[r + 1]].tolist()) == sorted(ref[1][gp[r]:gp[r + 1]].tolist())
    native.check_oob(gpu)


def test_dense_user_matrix_forward_replays_as_a_hip_graph(gpu):
    """The reference's call shape (dense user_matrix, on-stream CSR conversion with its hash table) captured into ONE HIP graph and
    replayed back to back: equal to the eager forward every time.  (The table used to be cleared by hipMemsetAsync: as graph nodes the
    memsets and the kernels around them lost their order from the second back-to-back replay on and the probe loop never ended —
    the library now clears with a kernel, ncf_common.h.)"""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    torch.manual_seed(5)
    F, I, B, users = 96, 300, 256, 16
    m = AttentionNCF(item_dim=F, item_emb=64, user_emb=64, att_dense=32, mlp_dense_layers=[256, 128]).eval().to(gpu)
    g = torch.Generator().manual_seed(6)
    rated = (torch.rand(I, F, generator=g) < 0.2).float().to(gpu)
    rows = torch.zeros(users, I)
    mask = torch.rand(users, I, generator=g) < 0.2
    rows[mask] = (torch.randint(1, 11, (users, I), generator=g).float() * 0.5 - 2.9)[mask]
    um = rows[torch.arange(B) // (B // users)].contiguous().to(gpu)
    cand = rated[torch.randint(0, I, (B,), generator=g).to(gpu)].contiguous()
    with torch.no_grad():
        ref = m(cand, rated, um).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(cand, rated, um)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = m(cand, rated, um)
        for _ in range(12):
            gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
        um2 = um.roll(B // users, 0).contiguous()                 # other users' rows into the captured buffer, then replay
        ref2 = m(cand, rated, um2).clone()
        um.copy_(um2)
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ref2)


def test_attn_candidates_random_shapes(gpu):
    """A seeded sweep over batch sizes, feature widths and layer widths through every form of the candidate kernel (LDS-staged, packed
    one-launch, packed with the K range over workgroups): all within 1e-5 of an fp64 product, the pair lists complete."""
    import random
    from deeprecommendation_amd import native
    rnd = random.Random(1234)
    g = torch.Generator().manual_seed(99)
    for case in range(24):
        B = rnd.choice([1, 2, 15, 16, 17, 255, 512, 1000, 2047, 2048, 2049, 3000, 4097])
        K = rnd.choice([1, 3, 31, 32, 33, 64, 100, 257, 1024, 2094])
        N1 = rnd.choice([64, 128])
        N2 = 16 * rnd.randint(1, 16)
        users = rnd.choice([1, 2, 5, 64, max(1, B // 3)])
        x = (torch.randn(B, K, generator=g) * (torch.rand(B, K, generator=g) < 0.3)).to(gpu)
        Wi = (torch.randn(N1, K, generator=g) / max(K, 1) ** 0.5).to(gpu)
        bi = (torch.randn(N1, generator=g) * 0.1).to(gpu)
        Wc = (torch.randn(N2, N1, generator=g) / N1 ** 0.5).to(gpu)
        b0 = (torch.randn(N2, generator=g) * 0.1).to(gpu)
        who = torch.randint(0, users, (B,), generator=g).to(gpu)
        e64 = x.double() @ Wi.double().t() + bi.double()
        p64 = e64 @ Wc.double().t() + b0.double()
        wpk = native.PackedCandidateWeight(Wi)
        wpk.use_packed = True
        for form, w in (("staged", Wi), ("packed", wpk)):
            emb, pc, grp = native.attn_candidates(x, w, bi, Wc, b0, who, users, 32)
            tag = f"case {case} {form}: B={B} K={K} N1={N1} N2={N2} users={users}"
            try:
                assert_close(emb, e64.float(), floor=1.0)          # 1e-5 of the largest output: signed inputs cancel
                assert_close(pc, p64.float(), floor=1.0)
            except AssertionError as exc:
                raise AssertionError(tag) from exc
            ids = grp[1]
            assert ids.numel() == B and torch.equal(torch.sort(ids).values, torch.arange(B, device=gpu)), tag
            assert torch.equal(who[ids], torch.sort(who).values), tag          # listed row by row
    native.check_oob(gpu)
