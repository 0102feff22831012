"""GPU parity tests (MI355X) for K1 gather, K2 MLP and the fused scoring kernel, through the C ABI.

Bars: index gathers bit-exact (torch.equal); fp32 MLP within 1e-5 relative of the CPU oracle — BASELINE.json
north_star tolerance — written as |a-b| <= 1e-5 * |b| + 1e-6 * max|b|: the absolute floor (a tenth of the bar, on the
batch's largest output) is there because scores pass through zero, and the fp32 summation-order difference between
two correct implementations scales with sum|w·x|, not with the (cancelled) result.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, onehot, record_error
from oracle import ncf_oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def assert_close(a, ref, rtol=RTOL, floor=0.1):
    """`floor` x rtol x max|ref| is the absolute part of the bar (default a tenth of the relative bar on the largest output)."""
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert a.shape == ref.shape
    tol = rtol * ref.abs() + floor * rtol * ref.abs().max()
    if ref.numel():
        diff = (a - ref).abs()
        used = diff / tol.clamp_min(1e-300)          # fraction of the bar each element used
        k = int(used.argmax())
        record_error("", float(diff.flatten()[k]), float(tol.flatten()[k]),
                     scale_rel=float(diff.max()) / max(float(ref.abs().max()), 1e-300))
    bad = (a - ref).abs() > tol
    assert not bool(bad.any()), f"max abs err {(a - ref).abs().max().item():.3e}, ref scale {ref.abs().max().item():.3e}"


@pytest.fixture(scope="module")
def native(gpu):
    from deeprecommendation_amd import native as n
    n.load_library()
    return n


# ----------------------------------------------------------------------------- K1
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("EA,EB", [(64, 64), (32, 32), (128, 128), (8, 8), (16, 48), (5, 3), (64, 0), (256, 256)])
@pytest.mark.parametrize("B", [1, 63, 1000])
def test_gather_concat_bit_exact(native, gpu, dtype, EA, EB, B):
    g = torch.Generator().manual_seed(EA * 131 + EB + B)
    ta = torch.randn(997, EA, generator=g).to(dtype).to(gpu)
    tb = torch.randn(211, EB, generator=g).to(dtype).to(gpu) if EB else None
    ia = torch.randint(0, 997, (B,), generator=g).to(gpu)
    ib = torch.randint(0, 211, (B,), generator=g).to(gpu) if EB else None
    out = native.gather_concat(ta, ia, tb, ib)
    ref = torch.cat((ta[ia], tb[ib]), dim=1) if EB else ta[ia]
    assert torch.equal(out, ref)


def test_gather_edge_cases(native, gpu):
    ta = torch.randn(50, 64, device=gpu)
    tb = torch.randn(40, 64, device=gpu)
    # empty batch
    out = native.gather_concat(ta, torch.zeros(0, dtype=torch.long, device=gpu), tb, torch.zeros(0, dtype=torch.long, device=gpu))
    assert out.shape == (0, 128)
    # identity index (dense activations) + strided (non-contiguous rows) table view
    big = torch.randn(50, 96, device=gpu)
    view = big[:, :64]
    out = native.gather_concat(view, None, tb[:50 - 10], None, B=40)
    assert torch.equal(out, torch.cat((view[:40], tb[:40]), 1))
    # maximum / minimum row ids, repeated ids
    ia = torch.tensor([0, 49, 49, 0, 25], device=gpu)
    ib = torch.tensor([39, 0, 39, 0, 1], device=gpu)
    assert torch.equal(native.gather_concat(ta, ia, tb, ib), torch.cat((ta[ia], tb[ib]), 1))
    # out-of-range ids never fault: zero row + sticky flag -> IndexError on check
    bad = torch.tensor([0, 50, -1], device=gpu)
    out = native.gather_concat(ta, bad, tb, torch.tensor([0, 1, 2], device=gpu))
    assert torch.equal(out[0, :64], ta[0]) and float(out[1, :64].abs().sum()) == 0.0 and float(out[2, :64].abs().sum()) == 0.0
    with pytest.raises(IndexError):
        native.check_oob(gpu)
    native.check_oob(gpu)  # flag was cleared


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_dot(native, gpu, dtype):
    g = torch.Generator().manual_seed(3)
    ta = torch.randn(300, 128, generator=g).to(dtype)
    tb = torch.randn(200, 128, generator=g).to(dtype)
    ia = torch.randint(0, 300, (777,), generator=g)
    ib = torch.randint(0, 200, (777,), generator=g)
    out = native.gather_dot(ta.to(gpu), ia.to(gpu), tb.to(gpu), ib.to(gpu))
    ref = (ta[ia].double() * tb[ib].double()).sum(1, keepdim=True).float()
    assert_close(out, ref, rtol=1e-5)


# ----------------------------------------------------------------------------- K2 generic
@pytest.mark.parametrize("M,N,K", [(1, 1, 1), (7, 5, 3), (128, 64, 32), (129, 65, 33), (300, 128, 2094), (1000, 256, 128),
                                   (64, 1, 256), (513, 16, 20), (33, 9, 100)])
def test_linear_generic(native, gpu, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / (K ** 0.5)
    b = torch.randn(N, generator=g)
    out = native.linear(x.to(gpu), w.to(gpu), b.to(gpu))
    assert_close(out, torch.nn.functional.linear(x.double(), w.double(), b.double()).float())
    out = native.linear(x.to(gpu), w.to(gpu), None)
    assert_close(out, torch.nn.functional.linear(x.double(), w.double()).float())


@pytest.mark.parametrize("dims", [[16, 16, 1], [24, 32, 16, 1], [128, 256, 128, 1], [20, 8], [100, 7, 5, 3, 2]])
def test_mlp_generic(native, gpu, dims):
    g = torch.Generator().manual_seed(sum(dims))
    B = 257
    x = torch.randn(B, dims[0], generator=g)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    out = native.mlp_forward(x.to(gpu), [w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    ref = O.mlp_forward(x.double(), [(w.double(), b.double()) for w, b in zip(ws, bs)]).float()
    assert_close(out, ref)


# ----------------------------------------------------------------------------- fused K1+K2
FUSED_SHAPES = [(32, 32, [256, 128]), (32, 32, [256]), (32, 32, [128]), (32, 32, [128, 64]),
                (64, 64, [256, 128]), (64, 64, [256]), (64, 64, [128]), (64, 64, [128, 64]),
                (128, 128, [256, 128]), (128, 128, [256]), (128, 128, [128]), (96, 32, [256, 128]), (8, 120, [256])]


@pytest.mark.parametrize("EA,EB,hidden", FUSED_SHAPES)
@pytest.mark.parametrize("B", [1, 31, 32, 33, 1000])
def test_score_fused_vs_oracle(native, gpu, EA, EB, hidden, B):
    g = torch.Generator().manual_seed(EA + 7 * EB + len(hidden) + B)
    dims = [EA + EB] + hidden + [1]
    assert native.fused_supported(EA, EB, dims)
    ta = torch.randn(500, EA, generator=g) * 0.5
    tb = torch.randn(300, EB, generator=g) * 0.5
    ia = torch.randint(0, 500, (B,), generator=g)
    ib = torch.randint(0, 300, (B,), generator=g)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    out = native.score_fused(ta.to(gpu), ia.to(gpu), tb.to(gpu), ib.to(gpu), packed)
    x = torch.cat((ta[ia], tb[ib]), 1)
    ref = O.mlp_forward(x.double(), [(w.double(), b.double()) for w, b in zip(ws, bs)]).float()
    assert out.shape == (B, 1)
    assert_close(out, ref)
    # fused == unfused (K1 then K2) to the same tolerance
    xg = native.gather_concat(ta.to(gpu), ia.to(gpu), tb.to(gpu), ib.to(gpu))
    assert torch.equal(xg.cpu(), x)
    out2 = native.mlp_forward(xg, [w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    assert_close(out2, ref)


def test_score_fused_asymmetric_weights_catch_layout_swaps(native, gpu):
    """A = I style check with ASYMMETRIC integer data: exact in fp32, so any row/col or k-permutation mix-up
    between the packed weights and the accumulator-as-operand chain shows as a hard mismatch."""
    EA = EB = 32
    dims = [64, 128, 64, 1]
    B = 64
    ta = torch.arange(40 * EA).reshape(40, EA).float() % 7 - 3
    tb = torch.arange(30 * EB).reshape(30, EB).float() % 5 - 2
    ws = [((torch.arange(dims[i + 1] * dims[i]).reshape(dims[i + 1], dims[i]) * 2654435761 % 11) - 5).float() for i in range(3)]
    bs = [(torch.arange(dims[i + 1]) % 3 - 1).float() for i in range(3)]
    ia = torch.arange(B) % 40
    ib = (torch.arange(B) * 7) % 30
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    out = native.score_fused(ta.to(gpu), ia.to(gpu), tb.to(gpu), ib.to(gpu), packed)
    ref = O.mlp_forward(torch.cat((ta[ia], tb[ib]), 1).double(), [(w.double(), b.double()) for w, b in zip(ws, bs)])
    assert float(ref.abs().max()) < 2 ** 24  # integers: fp32 arithmetic is exact
    assert torch.equal(out.cpu().double(), ref)


# ----------------------------------------------------------------------------- models vs golden (reference outputs)
def _model(cls, kw, state, gpu):
    m = cls(**kw)
    m.load_state_dict(state)
    return m.eval().to(gpu)


@pytest.mark.parametrize("name", ["g1_basic_onehot_small", "g1_basic_onehot_e32", "g1_basic_onehot_nodrop",
                                  "g1_basic_onehot_e64", "g1_basic_onehot_e128", "g1_basic_onehot_e64_h256"])
def test_basic_ncf_golden(gpu, name):
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    state, a, kw = load_golden(name)
    m = _model(BasicNCF, kw, state, gpu)
    ref = torch.from_numpy(a["out"])
    with torch.no_grad():
        out_idx = m(torch.as_tensor(a["user_pos"]).to(gpu), torch.as_tensor(a["item_pos"]).to(gpu))
        out_dense = m(onehot(a["user_pos"], kw["user_dim"]).to(gpu), onehot(a["item_pos"], kw["item_dim"]).to(gpu))
    assert_close(out_idx, ref)
    assert_close(out_dense, ref)
    # opt-in variants must reproduce the REFERENCE's outputs too: folded first layer (fp32) where it has an instance
    m.set_fold_first_layer(True)
    with torch.no_grad():
        assert_close(m(torch.as_tensor(a["user_pos"]).to(gpu), torch.as_tensor(a["item_pos"]).to(gpu)), ref)


def test_basic_ncf_dense_profiles_golden(gpu):
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    state, a, kw = load_golden("g1_basic_dense_profiles")
    m = _model(BasicNCF, kw, state, gpu)
    with torch.no_grad():
        out = m(torch.from_numpy(a["X_user"]).to(gpu), torch.from_numpy(a["X_item"]).to(gpu))
    assert_close(out, torch.from_numpy(a["out"]))


def test_mf_golden(gpu):
    from deeprecommendation_amd.neural_collaborative_filtering.models.mf import MF
    state, a, kw = load_golden("g2_mf_onehot")
    m = _model(MF, kw, state, gpu)
    ref = torch.from_numpy(a["out"])
    with torch.no_grad():
        out_idx = m(torch.as_tensor(a["user_pos"]).to(gpu), torch.as_tensor(a["item_pos"]).to(gpu))
        out_dense = m(onehot(a["user_pos"], kw["user_dim"]).to(gpu), onehot(a["item_pos"], kw["item_dim"]).to(gpu))
    assert_close(out_idx, ref)
    assert_close(out_dense, ref)


def test_cfg1_ml1m_golden(gpu):
    """BASELINE configs[0] on the GPU: 20k pairs at 6040 x 3706, emb 32, MLP [256], batch 512."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    state, a, kw = load_golden("cfg1_basic_ml1m")
    m = _model(BasicNCF, kw, state, gpu)
    up = torch.as_tensor(a["user_pos"].astype(np.int64)).to(gpu)
    ip = torch.as_tensor(a["item_pos"].astype(np.int64)).to(gpu)
    with torch.no_grad():
        out = torch.cat([m(up[s:s + 512], ip[s:s + 512]) for s in range(0, len(up), 512)])
    assert_close(out, torch.from_numpy(a["out"]))


def test_full_size_cfg2_properties(native, gpu):
    """BASELINE configs[1] at full size (1M x 100k, emb 64, B = 65536): size-independent properties.
    (a) fused == gather + generic MLP; (b) permuting the batch permutes the output bit-exactly;
    (c) a subsample agrees with the CPU oracle."""
    g = torch.Generator().manual_seed(1234)
    U, I, E, B = 1_000_000, 100_000, 64, 65536
    tu = (torch.randn(U, E, generator=g) * 0.05).to(gpu)
    ti = (torch.randn(I, E, generator=g) * 0.05).to(gpu)
    dims = [128, 256, 128, 1]
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(3)]
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    iu = torch.randint(0, U, (B,), generator=g).to(gpu)
    ii = torch.randint(0, I, (B,), generator=g).to(gpu)
    iu[0], iu[1], ii[0], ii[1] = 0, U - 1, I - 1, 0  # extreme row ids
    out = native.score_fused(tu, iu, ti, ii, packed)
    x = native.gather_concat(tu, iu, ti, ii)
    assert torch.equal(x, torch.cat((tu[iu], ti[ii]), 1))
    out2 = native.mlp_forward(x, [w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    assert_close(out, out2.cpu())
    perm = torch.randperm(B, generator=g).to(gpu)
    outp = native.score_fused(tu, iu[perm].contiguous(), ti, ii[perm].contiguous(), packed)
    assert torch.equal(outp, out[perm])
    sub = torch.arange(0, B, 97)
    ref = O.mlp_forward(x[sub.to(gpu)].cpu().double(), [(w.double(), b.double()) for w, b in zip(ws, bs)]).float()
    assert_close(out[sub.to(gpu)], ref)
    native.check_oob(gpu)


def test_full_size_cfg5_properties(native, gpu, kernel_option):
    """BASELINE configs[4] at one GPU's full size (100 M x 128 and 10 M x 128 bf16 tables = 28 GB, MLP 256-256-128-1):
    size-independent properties of BOTH bf16 kernels at 65 536 and 262 144 pairs — (a) permuting the batch permutes the
    output bit-exactly, (b) extreme row ids (0, last) are read correctly, (c) the two kernels agree within the bf16
    tolerance, (d) a subsample agrees with a float64 evaluation of the same bf16-rounded operands."""
    free, _ = torch.cuda.mem_get_info(gpu)
    if free < 40 << 30:
        pytest.skip("needs 40 GB of free HBM")
    U, I, E = 100_000_000, 10_000_000, 128
    g = torch.Generator(device=gpu).manual_seed(5)

    def table(rows):
        t = torch.empty((rows, E), dtype=torch.bfloat16, device=gpu)
        for s0 in range(0, rows, 5_000_000):
            t[s0:s0 + 5_000_000] = (torch.randn((min(5_000_000, rows - s0), E), device=gpu, generator=g) * 0.05).to(torch.bfloat16)
        return t

    tu, ti = table(U), table(I)
    dims = [2 * E, 256, 128, 1]
    ws = [torch.randn(dims[k + 1], dims[k], device=gpu, generator=g) / dims[k] ** 0.5 for k in range(3)]
    bs = [torch.randn(dims[k + 1], device=gpu, generator=g) * 0.1 for k in range(3)]
    packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
    wr = [w.to(torch.bfloat16).double().cpu() for w in ws[:2]] + [ws[2].double().cpu()]
    for B in (65536, 262144):
        iu = torch.randint(0, U, (B,), device=gpu, generator=g)
        ii = torch.randint(0, I, (B,), device=gpu, generator=g)
        iu[0], iu[1], ii[0], ii[1] = 0, U - 1, I - 1, 0
        perm = torch.randperm(B, device=gpu, generator=g)
        outs = {}
        for kernel in ("stream", "ws", "ws8"):
            kernel_option("bf16_kernel", kernel)
            out = native.score_fused(tu, iu, ti, ii, packed)
            outp = native.score_fused(tu, iu[perm].contiguous(), ti, ii[perm].contiguous(), packed)
            assert torch.equal(outp, out[perm]), kernel
            outs[kernel] = out
        assert_close(outs["ws"], outs["stream"].cpu(), rtol=2e-3)
        assert_close(outs["ws8"], outs["stream"].cpu(), rtol=2e-3)
        sub = torch.cat((torch.arange(0, 4, device=gpu), torch.arange(4, B, 1021, device=gpu)))
        x = torch.cat((tu[iu[sub]], ti[ii[sub]]), 1).double().cpu()
        h = torch.relu(x @ wr[0].t() + bs[0].double().cpu()).float().to(torch.bfloat16).double()
        h = torch.relu(h @ wr[1].t() + bs[1].double().cpu())
        ref = (h @ wr[2].t() + bs[2].double().cpu()).float()
        for kernel in ("stream", "ws", "ws8"):
            assert_close(outs[kernel][sub], ref, rtol=2e-3)
    native.check_oob(gpu)
    del tu, ti
    torch.cuda.empty_cache()


# ----------------------------------------------------------------------------- bf16 fused path (BASELINE config 5 arithmetic)
@pytest.mark.parametrize("E,hidden", [(128, [256, 128]), (128, [256]), (64, [256, 128]), (64, [256])])
@pytest.mark.parametrize("B", [1, 63, 255, 256, 257, 3000, 40000, 100001, 140000])
@pytest.mark.parametrize("kernel", ["stream", "ws", "ws8", "auto"])
def test_score_fused_bf16_vs_oracle(gpu, kernel_option, E, hidden, B, kernel):
    """bf16 tables / weights, fp32 accumulate: gathers are bit-exact on the bf16 table; the MLP is compared with the
    oracle evaluated on the same bf16-rounded operands — tolerance 2e-3 relative (builder-defined: BASELINE pins only
    fp32; the residual is fp32 accumulation order plus bf16 re-rounding of hidden activations that sit on a rounding
    boundary).  All three bf16 kernels (slab-streaming, 4-wave and 8-wave weight-stationary) run every batch size, including
    the ones the library's own dispatch would hand to another kernel: 40 000 / 100 001 / 140 000 pairs give the persistent
    kernels 3-17 tiles (units) per workgroup with ragged tails; a one-hidden-layer MLP sends "ws8" to the 4-wave kernel."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd import native
    if kernel == "auto":
        kernel_option("bf16_kernel", "auto")
        if B not in (3000, 140000):
            pytest.skip("auto dispatch is covered by one batch on each side of the threshold")
    else:
        kernel_option("bf16_kernel", kernel)
    torch.manual_seed(E + len(hidden))
    U, I = 3000, 700
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=hidden).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    u = torch.randint(0, U, (B,), generator=g)
    i = torch.randint(0, I, (B,), generator=g)
    ref = O.basic_ncf_forward_indexed_bf16(state, u, i)
    m.to(gpu).set_scoring_dtype(torch.bfloat16)
    with torch.no_grad():
        out = m(u.to(gpu), i.to(gpu))
    assert m._table("user", m.user_embeddings[0]).dtype == torch.bfloat16
    assert_close(out, ref, rtol=2e-3)
    # the bf16 table itself is the RNE rounding of W^T + b, and gathering it is bit-exact
    tu = m._table("user", m.user_embeddings[0])
    assert torch.equal(tu.cpu(), O.embedding_table(state["user_embeddings.0.weight"], state["user_embeddings.0.bias"]).to(torch.bfloat16))
    assert torch.equal(native.gather_concat(tu, u.to(gpu)), tu[u.to(gpu)])


@pytest.mark.parametrize("units_per_wg", [1, 2, 3, 4, 5, 6, 7, 8, 13])
@pytest.mark.parametrize("ragged", [0, 1, 31, 33])
def test_score_fused_bf16_ws8_pipeline_depths(native, gpu, kernel_option, units_per_wg, ragged):
    """The 8-wave kernel's software pipeline is 4 phases deep (layer 1 -> pack -> layer 2 -> dot -> store) and its row DMAs run 5
    units ahead: every count of 32-pair units per workgroup from 1 up (prologue-only, shorter than the pipeline, steady state),
    with ragged tails, against the 4-wave kernel on the same inputs (same MFMA chains: equal to fp32 summation order)."""
    cus = torch.cuda.get_device_properties(gpu).multi_processor_count
    B = 32 * cus * (units_per_wg - 1) + 32 * (cus // 2 + 1) + ragged
    g = torch.Generator(device=gpu).manual_seed(units_per_wg * 100 + ragged)
    E, U, I = 128, 50000, 20000
    tu = (torch.randn(U, E, device=gpu, generator=g) * 0.05).to(torch.bfloat16)
    ti = (torch.randn(I, E, device=gpu, generator=g) * 0.05).to(torch.bfloat16)
    dims = [2 * E, 256, 128, 1]
    ws = [torch.randn(dims[k + 1], dims[k], device=gpu, generator=g) / dims[k] ** 0.5 for k in range(3)]
    bs = [torch.randn(dims[k + 1], device=gpu, generator=g) * 0.1 for k in range(3)]
    packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
    iu = torch.randint(0, U, (B,), device=gpu, generator=g)
    ii = torch.randint(0, I, (B,), device=gpu, generator=g)
    outs = {}
    for kernel in ("ws", "ws8"):
        kernel_option("bf16_kernel", kernel)
        outs[kernel] = native.score_fused(tu, iu, ti, ii, packed).clone()
    assert_close(outs["ws8"], outs["ws"].cpu(), rtol=1e-5)
    native.check_oob(gpu)


def test_score_fused_bf16_ws8_falls_back_on_other_shapes(native, gpu, kernel_option):
    """Tables of different widths (EA != EB) are not the 8-wave kernel's shape: the option then selects the 4-wave kernel."""
    g = torch.Generator(device=gpu).manual_seed(3)
    EA, EB, B = 192, 64, 5000
    ta = (torch.randn(900, EA, device=gpu, generator=g) * 0.05).to(torch.bfloat16)
    tb = (torch.randn(400, EB, device=gpu, generator=g) * 0.05).to(torch.bfloat16)
    dims = [EA + EB, 256, 128, 1]
    ws = [torch.randn(dims[k + 1], dims[k], device=gpu, generator=g) / dims[k] ** 0.5 for k in range(3)]
    bs = [torch.randn(dims[k + 1], device=gpu, generator=g) * 0.1 for k in range(3)]
    packed = native.PackedMLP(ws, bs, dtype=torch.bfloat16)
    ia = torch.randint(0, 900, (B,), device=gpu, generator=g)
    ib = torch.randint(0, 400, (B,), device=gpu, generator=g)
    outs = {}
    for kernel in ("stream", "ws", "ws8", "auto"):
        kernel_option("bf16_kernel", kernel)
        outs[kernel] = native.score_fused(ta, ia, tb, ib, packed).clone()
    assert torch.equal(outs["ws8"], outs["ws"]) and torch.equal(outs["auto"], outs["ws"])
    assert_close(outs["ws"], outs["stream"].cpu(), rtol=2e-3)


@pytest.mark.parametrize("kernel", ["stream", "ws", "ws8"])
def test_score_fused_bf16_exact_small_integers(native, gpu, kernel_option, kernel):
    """Exact check of the packed layer-2 k permutation: small integer data is exact in bf16 x bf16 -> fp32."""
    kernel_option("bf16_kernel", kernel)
    E = 64
    dims = [128, 256, 128, 1]
    B = 300
    ta = (torch.arange(50 * E).reshape(50, E) % 5 - 2).float()
    tb = (torch.arange(40 * E).reshape(40, E) % 3 - 1).float()
    ws = [((torch.arange(dims[i + 1] * dims[i]).reshape(dims[i + 1], dims[i]) * 2654435761 % 7) - 3).float() for i in range(3)]
    ws[1] = (ws[1] % 2)  # keep layer-2 sums of bf16-exact hidden values exact: hidden stays < 2^8 in magnitude
    bs = [(torch.arange(dims[i + 1]) % 3 - 1).float() for i in range(3)]
    ws[0] = ws[0] * (torch.arange(dims[0]) % 16 == 0).float()  # sparse layer 1 -> |hidden| <= 8*2*3+1, exact in bf16
    ia = torch.arange(B) % 50
    ib = (torch.arange(B) * 7) % 40
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs], dtype=torch.bfloat16)
    out = native.score_fused(ta.to(torch.bfloat16).to(gpu), ia.to(gpu), tb.to(torch.bfloat16).to(gpu), ib.to(gpu), packed)
    ref = O.mlp_forward(torch.cat((ta[ia], tb[ib]), 1).double(), [(w.double(), b.double()) for w, b in zip(ws, bs)])
    assert float(ref.abs().max()) < 2 ** 24
    assert torch.equal(out.cpu().double(), ref)


@pytest.mark.parametrize("kernel", ["stream", "ws", "ws8"])
def test_score_fused_out_of_range_rows_read_as_zeros(native, gpu, kernel_option, kernel):
    """ABI contract: an out-of-range id never faults; that table's part of the row is zeros and the sticky flag is set."""
    kernel_option("bf16_kernel", kernel)
    g = torch.Generator().manual_seed(0)
    E, dims = 64, [128, 256, 128, 1]
    ta = torch.randn(100, E, generator=g)
    tb = torch.randn(50, E, generator=g)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(3)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(3)]
    ia = torch.tensor([3, 100, 5, -1, 7])
    ib = torch.tensor([1, 2, 50, 4, 49])
    for dt, tol in ((torch.float32, 1e-5), (torch.bfloat16, 2e-3)):
        packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs], dtype=dt)
        if dt == torch.bfloat16:
            packed_dims_ok = native.fused_supported(E, E, dims, dtype=dt)
            assert packed_dims_ok
        out = native.score_fused(ta.to(dt).to(gpu), ia.to(gpu), tb.to(dt).to(gpu), ib.to(gpu), packed)
        xa = ta.to(dt).float()[ia.clamp(0, 99)] * ((ia >= 0) & (ia < 100)).float()[:, None]
        xb = tb.to(dt).float()[ib.clamp(0, 49)] * ((ib >= 0) & (ib < 50)).float()[:, None]
        wr = [w.to(dt).float() for w in ws[:2]] + [ws[2]]
        h = torch.relu(torch.cat((xa, xb), 1).double() @ wr[0].double().t() + bs[0].double())
        if dt == torch.bfloat16:
            h = h.float().to(dt).double()
        h = torch.relu(h @ wr[1].double().t() + bs[1].double())
        ref = (h @ wr[2].double().t() + bs[2].double()).float()
        assert_close(out, ref, rtol=tol)
        with pytest.raises(IndexError):
            native.check_oob(gpu)


# ----------------------------------------------------------------------------- folded first layer (opt-in)
@pytest.mark.parametrize("E,hidden", [(64, [256, 128]), (32, [256, 128]), (64, [128, 64]), (16, [256, 64]), (64, [128, 128])])
@pytest.mark.parametrize("B", [1, 255, 256, 257, 2000])
def test_score_folded_vs_oracle(gpu, E, hidden, B):
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    torch.manual_seed(E + hidden[0])
    U, I = 3000, 700
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=hidden).eval()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(B)
    u = torch.randint(0, U, (B,), generator=g)
    i = torch.randint(0, I, (B,), generator=g)
    ref = O.basic_ncf_forward_indexed(state, u, i)
    m.to(gpu)
    with torch.no_grad():
        plain = m(u.to(gpu), i.to(gpu))
        m.set_fold_first_layer(True)
        out = m(u.to(gpu), i.to(gpu))
    assert any(k[0] == "folded" and v is not None for k, v in m._native_cache.items() if isinstance(k, tuple))
    assert_close(out, ref)
    assert_close(plain, ref)


def test_score_folded_out_of_range(native, gpu):
    g = torch.Generator().manual_seed(0)
    PA = torch.randn(100, 256, generator=g)
    PB = torch.randn(50, 256, generator=g)
    w2 = torch.randn(128, 256, generator=g) / 16
    b2 = torch.randn(128, generator=g) * 0.1
    w3 = torch.randn(1, 128, generator=g) / 11
    b3 = torch.randn(1, generator=g)
    tail = native.PackedMLP([w2.to(gpu), w3.to(gpu)], [b2.to(gpu), b3.to(gpu)])
    ia = torch.tensor([3, 100, 5, -1])
    ib = torch.tensor([1, 2, 50, 4])
    out = native.score_folded(PA.to(gpu), ia.to(gpu), PB.to(gpu), ib.to(gpu), tail)
    xa = PA[ia.clamp(0, 99)] * ((ia >= 0) & (ia < 100)).float()[:, None]
    xb = PB[ib.clamp(0, 49)] * ((ib >= 0) & (ib < 50)).float()[:, None]
    h = torch.relu((xa + xb).double())
    ref = (torch.relu(h @ w2.double().t() + b2.double()) @ w3.double().t() + b3.double()).float()
    assert_close(out, ref)
    with pytest.raises(IndexError):
        native.check_oob(gpu)


@pytest.mark.parametrize("E,hidden", [(64, [256, 128]), (32, [256]), (128, [256, 128]), (64, [128, 64])])
def test_score_fused_bit_exact_vs_kernel_order_c_oracle(native, gpu, E, hidden):
    """The fp32 MFMA is an exact k-ordered fmaf chain, so the fused kernel must equal — BIT FOR BIT — a plain C
    restatement of its operation order (oracle/ncf_oracle_c.c, itself held to the reference goldens at 1e-5)."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(E + len(hidden))
    dims = [2 * E] + hidden + [1]
    B = 3000
    ta = torch.randn(700, E, generator=g) * 0.5
    tb = torch.randn(300, E, generator=g) * 0.5
    ia = torch.randint(0, 700, (B,), generator=g)
    ib = torch.randint(0, 300, (B,), generator=g)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    out = native.score_fused(ta.to(gpu), ia.to(gpu), tb.to(gpu), ib.to(gpu), packed)
    ref = c_oracle.score_fused_f32(ta, tb, ia, ib, ws, bs)
    assert torch.equal(out.cpu(), ref), f"max abs diff {(out.cpu() - ref).abs().max().item():.3e}"


@pytest.mark.parametrize("E,hidden", [(64, [256, 128]), (32, [256]), (64, [128, 64]), (128, [256, 128])])
def test_small_batch_kernel_is_bit_identical_to_main_kernel(native, gpu, E, hidden):
    """Below NCF_SMALL_TILES tiles ncf_score_fused runs the 4-waves-per-tile kernel, above it the one-wave-per-tile
    kernel.  Same operation order by construction: the same pairs scored in one big batch and in small batches must
    agree bit for bit, and both equal the kernel-order C oracle."""
    from oracle import c_oracle
    g = torch.Generator().manual_seed(E * 7 + len(hidden))
    dims = [2 * E] + hidden + [1]
    B = 30000  # 938 tiles -> main kernel; chunks of 999 -> small kernel (ragged last tile)
    ta = torch.randn(900, E, generator=g) * 0.5
    tb = torch.randn(400, E, generator=g) * 0.5
    ia = torch.randint(0, 900, (B,), generator=g)
    ib = torch.randint(0, 400, (B,), generator=g)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    tag, tbg, iag, ibg = ta.to(gpu), tb.to(gpu), ia.to(gpu), ib.to(gpu)
    big = native.score_fused(tag, iag, tbg, ibg, packed)
    small = torch.cat([native.score_fused(tag, iag[s:s + 999].contiguous(), tbg, ibg[s:s + 999].contiguous(), packed)
                       for s in range(0, B, 999)])
    assert torch.equal(big, small)
    ref = c_oracle.score_fused_f32(ta, tb, ia, ib, ws, bs)
    assert torch.equal(big.cpu(), ref)


@pytest.mark.parametrize("B", [32768 + 1, 40000, 49152, 65536 + 31, 73729, 32768 + 28000])
def test_ragged_last_round_split_is_bit_identical(native, gpu, B):
    """A batch that does not fill its last round of the one-wave-per-tile kernel is split: full rounds on the main
    kernel, the tail on the 4-waves-per-tile kernel (mlp_fused.hip launch_inst).  The split must not change a bit:
    compare with the same pairs scored in 999-pair batches (small kernel only) and with the kernel-order C oracle,
    and check that an out-of-range id in the tail still raises and reads as a zero row."""
    from oracle import c_oracle
    E, hidden = 64, [256, 128]
    g = torch.Generator().manual_seed(B)
    dims = [2 * E] + hidden + [1]
    ta = torch.randn(900, E, generator=g) * 0.5
    tb = torch.randn(400, E, generator=g) * 0.5
    ia = torch.randint(0, 900, (B,), generator=g)
    ib = torch.randint(0, 400, (B,), generator=g)
    ws = [torch.randn(dims[i + 1], dims[i], generator=g) / dims[i] ** 0.5 for i in range(len(dims) - 1)]
    bs = [torch.randn(dims[i + 1], generator=g) * 0.1 for i in range(len(dims) - 1)]
    packed = native.PackedMLP([w.to(gpu) for w in ws], [b.to(gpu) for b in bs])
    tag, tbg, iag, ibg = ta.to(gpu), tb.to(gpu), ia.to(gpu), ib.to(gpu)
    whole = native.score_fused(tag, iag, tbg, ibg, packed)
    assert whole.shape == (B, 1)
    pieces = torch.cat([native.score_fused(tag, iag[s:s + 999].contiguous(), tbg, ibg[s:s + 999].contiguous(), packed)
                        for s in range(0, B, 999)])
    assert torch.equal(whole, pieces)
    if B <= 49152:
        assert torch.equal(whole.cpu(), c_oracle.score_fused_f32(ta, tb, ia, ib, ws, bs))
    native.check_oob(gpu)
    bad = iag.clone()
    bad[B - 5] = 900
    out = native.score_fused(tag, bad, tbg, ibg, packed)
    with pytest.raises(IndexError):
        native.check_oob(gpu)
    zero_row = native.score_fused(torch.cat([tag, torch.zeros(1, E, device=gpu)]), bad, tbg, ibg, packed)
    native.check_oob(gpu)
    assert torch.equal(out, zero_row)


@pytest.mark.parametrize("M,K,N,relu", [(20000, 256, 128, True), (16500, 64, 128, False), (300, 40, 64, True), (16384, 128, 256, False)])
@pytest.mark.parametrize("kernel", ["rs", "rsp"])
def test_linear_kernels_agree_with_float64(native, gpu, kernel_option, M, K, N, relu, kernel):
    """Both row-streaming GEMM forms (one tile per wave / persistent with the A ring across tiles) against a float64
    product, incl. a ragged last row block and a shape the persistent form declines (K % 64 != 0 falls back by itself)."""
    kernel_option("linear_kernel", kernel)
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    out = native.linear_act(x.to(gpu), w.to(gpu), b.to(gpu), relu)
    ref = x.double() @ w.double().t() + b.double()
    if relu:
        ref = torch.relu(ref)
    assert_close(out, ref.float())


@pytest.mark.parametrize("M,K,N", [(4096, 2094, 64), (1000, 2094, 64), (4100, 1030, 128), (33, 2094, 64), (500, 2048, 64), (700, 96, 64), (64, 40, 32)])
@pytest.mark.parametrize("ks", ["4", "8", None])
def test_skinny_split_k_widths_agree_with_float64(native, gpu, kernel_option, M, K, N, ks):
    """Skinny-deep Linear (AttentionNCF's F = 2094 candidate layer): K split over 4 or 8 waves of a workgroup, partial
    sums added through LDS in slice order; forced either way and by the shape rule, against a float64 product (K % 8 != 0
    tail, ragged last row block; for the 8-slice form, whose operand blocks are staged through LDS: K a whole number of 32-wide
    blocks, fewer blocks than waves, and K below one block)."""
    if ks is None:
        kernel_option("linear_kslices", "auto")
    else:
        kernel_option("linear_kslices", ks)
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    out = native.linear(x.to(gpu), w.to(gpu), b.to(gpu))
    again = native.linear(x.to(gpu), w.to(gpu), b.to(gpu))
    assert torch.equal(out, again)  # slice order is fixed: run-to-run identical
    assert_close(out, (x.double() @ w.double().t() + b.double()).float())
