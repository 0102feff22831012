"""GPU tests of the training step on the HIP autograd blocks: loss and every parameter gradient
against the same model run with plain torch ops on the CPU (the reference's training arithmetic)."""
import copy

import numpy as np

import pytest
import torch

pytestmark = pytest.mark.gpu


def _grads_close(gpu_model, cpu_model, rtol=2e-5, exactly_zero=()):
    for (n, p), (_, q) in zip(gpu_model.named_parameters(), cpu_model.named_parameters()):
        assert p.grad is not None, n
        if n in exactly_zero:      # a gradient that is 0 by construction: the kernels return 0, CPU autograd its rounding noise
            assert float(p.grad.abs().max()) == 0.0 and float(q.grad.abs().max()) < 1e-5, n
            continue
        a, b = p.grad.cpu().double(), q.grad.double()
        scale = float(b.abs().max()) + 1e-30
        err = float((a - b).abs().max())
        assert err <= rtol * scale, f"{n}: max abs err {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("indexed", [True, False])
@pytest.mark.parametrize("hidden", [[256, 128], [256], [48, 20]])
def test_basic_ncf_training_step_gradients(gpu, indexed, hidden):
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    torch.manual_seed(1)
    U, I, E, B = 700, 300, 64, 1000
    m_cpu = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=hidden, dropout_rate=None).train()
    m_gpu = copy.deepcopy(m_cpu).to(gpu).train()
    g = torch.Generator().manual_seed(2)
    u = torch.randint(0, U, (B,), generator=g)
    i = torch.randint(0, I, (B,), generator=g)
    u[:200] = 5  # duplicates: the embedding gradient must accumulate
    y = torch.rand(B, 1, generator=g) * 5
    if indexed:
        xu_c, xi_c = u, i
    else:
        xu_c = torch.nn.functional.one_hot(u, U).float()
        xi_c = torch.nn.functional.one_hot(i, I).float()
    loss_c = torch.nn.functional.mse_loss(m_cpu(xu_c, xi_c), y, reduction="sum")
    loss_c.backward()
    out_g = m_gpu(xu_c.to(gpu), xi_c.to(gpu))
    assert out_g.requires_grad
    loss_g = torch.nn.functional.mse_loss(out_g, y.to(gpu), reduction="sum")
    loss_g.backward()
    assert abs(float(loss_g) - float(loss_c)) <= 1e-5 * abs(float(loss_c))
    _grads_close(m_gpu, m_cpu)


def test_training_with_dropout_runs_and_learns(gpu):
    """Dropout (torch op between the HIP Linear blocks) active: a few Adam steps reduce the loss."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    torch.manual_seed(3)
    m = BasicNCF(item_dim=200, user_dim=300, item_emb=32, user_emb=32, mlp_dense_layers=[256], dropout_rate=0.2).to(gpu).train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(4)
    u = torch.randint(0, 300, (2048,), generator=g).to(gpu)
    i = torch.randint(0, 200, (2048,), generator=g).to(gpu)
    y = ((u % 5).float() + (i % 3).float()).view(-1, 1)
    losses = []
    for _ in range(30):
        opt.zero_grad()
        loss = torch.nn.functional.mse_loss(m(u, i), y, reduction="mean")
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.5 * losses[0]
    # and the eval path (fused kernel) sees the updated weights
    m.eval()
    with torch.no_grad():
        ev = torch.nn.functional.mse_loss(m(u, i), y, reduction="mean")
    assert float(ev) < losses[0]


@pytest.mark.parametrize("M,N1,N2", [(1, 1, 1), (1000, 1, 128), (5000, 256, 128), (777, 20, 48), (4096, 64, 2094), (70000, 128, 256)])
def test_gemm_tn_and_colsum(gpu, M, N1, N2):
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(M + N1)
    A = torch.randn(M, N1, generator=g)
    Bm = torch.randn(M, N2, generator=g)
    out = native.gemm_tn(A.to(gpu), Bm.to(gpu))
    ref = A.double().t() @ Bm.double()
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-6
    out2 = native.gemm_tn(A.to(gpu), Bm.to(gpu))
    assert torch.equal(out, out2)  # ordered split over the batch: bitwise reproducible
    cs = native.colsum(A.to(gpu))
    refc = A.double().sum(0)
    assert float((cs.cpu().double() - refc).abs().max()) <= 1e-5 * float(refc.abs().max()) + 1e-5


def test_scatter_add_rows(gpu):
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(0)
    src = torch.randn(5000, 96, generator=g)
    idx = torch.randint(0, 40, (5000,), generator=g)
    dst = torch.zeros(40, 64)
    ref = dst.clone().double().index_add_(0, idx, src[:, 32:].double())
    out = native.scatter_add_rows(src.to(gpu)[:, 32:], idx.to(gpu), dst.to(gpu))
    assert float((out.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


def test_gather_concat_autograd_block(gpu):
    """GatherConcatFn (row-gather forward, scatter-add backward) for tables kept in table layout."""
    from deeprecommendation_amd.autograd import GatherConcatFn
    g = torch.Generator().manual_seed(0)
    ta = torch.randn(300, 64, generator=g, requires_grad=True)
    tb = torch.randn(100, 32, generator=g, requires_grad=True)
    ia = torch.randint(0, 300, (2000,), generator=g)
    ib = torch.randint(0, 100, (2000,), generator=g)
    wgt = torch.randn(2000, 96, generator=g)
    (torch.cat((ta[ia], tb[ib]), 1) * wgt).sum().backward()
    ga, gb = ta.grad.clone(), tb.grad.clone()
    tag = ta.detach().to(gpu).requires_grad_(True)
    tbg = tb.detach().to(gpu).requires_grad_(True)
    out = GatherConcatFn.apply(tag, ia.to(gpu), tbg, ib.to(gpu))
    assert torch.equal(out.detach().cpu(), torch.cat((ta[ia], tb[ib]), 1).detach())
    (out * wgt.to(gpu)).sum().backward()
    assert float((tag.grad.cpu() - ga).abs().max()) <= 1e-5 * float(ga.abs().max())
    assert float((tbg.grad.cpu() - gb).abs().max()) <= 1e-5 * float(gb.abs().max())


def test_gather_columns_autograd_block(gpu):
    """GatherColumnsFn (Linear-layout embedding parameter W [E, U]) == torch's `W.t()[idx] + b` forward (bit-exact) and
    backward (float atomics: duplicates of an id add in any order -> 4e-6), incl. repeated ids."""
    from deeprecommendation_amd.autograd import GatherColumnsFn
    g = torch.Generator().manual_seed(0)
    E, U, B = 64, 500, 3000
    W0 = torch.randn(E, U, generator=g)
    b0 = torch.randn(E, generator=g)
    idx = torch.randint(0, U, (B,), generator=g)
    idx[:200] = 7                                          # a hot id
    dY = torch.randn(B, E, generator=g)
    W1, b1 = W0.clone().to(gpu).requires_grad_(), b0.clone().to(gpu).requires_grad_()
    y1 = GatherColumnsFn.apply(W1, b1, idx.to(gpu))
    y1.backward(dY.to(gpu))
    W2, b2 = W0.clone().requires_grad_(), b0.clone().requires_grad_()
    y2 = W2.t()[idx] + b2
    y2.backward(dY)
    assert torch.equal(y1.detach().cpu(), y2.detach())
    assert float((W1.grad.cpu() - W2.grad).abs().max()) <= 4e-6 * float(W2.grad.abs().max()) + 1e-6   # 200 duplicates, any order
    assert float((b1.grad.cpu() - b2.grad).abs().max()) <= 1e-5 * float(b2.grad.abs().max())


def test_linear_relu_dropout_block_matches_torch_ops_with_the_same_seed(gpu):
    """LinearReluDropoutFn draws torch's own dropout mask (same seed, same shape -> same Philox stream), so output and
    all three gradients can be compared with relu -> dropout in plain torch ops on the GPU: 1e-5."""
    from deeprecommendation_amd.autograd import LinearReluDropoutFn, mlp_train
    g = torch.Generator().manual_seed(9)
    B, K, N, p = 4096, 128, 256, 0.3
    x0, w0, b0 = torch.randn(B, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1
    dZ = torch.randn(B, N, generator=g).to(gpu)
    leaves = lambda: [t.clone().to(gpu).requires_grad_() for t in (x0, w0, b0)]
    x1, w1, b1 = leaves()
    torch.manual_seed(123)
    z1 = LinearReluDropoutFn.apply(x1, w1, b1, p)
    z1.backward(dZ)
    x2, w2, b2 = leaves()
    torch.manual_seed(123)
    z2 = torch.nn.functional.dropout(torch.relu(torch.nn.functional.linear(x2, w2, b2)), p, True)
    z2.backward(dZ)
    from test_gpu_basic import assert_close
    assert float((z1 == 0).float().mean()) > p          # dropped + inactive
    assert torch.equal(z1 == 0, z2 == 0)
    assert_close(z1, z2)
    for a, b in ((x1.grad, x2.grad), (w1.grad, w2.grad), (b1.grad, b2.grad)):
        assert_close(a, b, rtol=2e-5)
    # mlp_train picks the fused node for Linear -> ReLU -> Dropout(train) and the plain one in eval mode / p = 0
    seq = torch.nn.Sequential(torch.nn.Linear(K, N), torch.nn.ReLU(), torch.nn.Dropout(p), torch.nn.Linear(N, 1)).to(gpu)
    xin = x0.to(gpu).requires_grad_()
    out = mlp_train(seq, xin)
    assert "LinearFn" in type(out.grad_fn).__name__ and "LinearReluDropoutFn" in type(out.grad_fn.next_functions[0][0]).__name__
    seq.eval()
    ev = mlp_train(seq, xin)
    assert "LinearReluDropoutFn" not in type(ev.grad_fn.next_functions[0][0]).__name__
    out.sum().backward()
    assert xin.grad is not None and torch.isfinite(xin.grad).all()


@pytest.mark.parametrize("M,N", [(1000, 256), (77, 30), (1, 4), (513, 129)])
def test_relu_backward_out_of_place(gpu, M, N):
    """ncf_relu_backward_out: dY masked by Y > 0 into a new buffer (16-byte path and scalar path, strided dY), equal to the
    in-place entry and to torch; the incoming gradient is left untouched."""
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(M + N)
    wide = torch.randn(M, N + 8, generator=g).to(gpu)
    dY = wide[:, :N] if N % 4 == 0 else wide[:, 1:N + 1]       # row stride != N; second form is not 16-byte aligned
    Y = torch.relu(torch.randn(M, N, generator=g)).to(gpu)
    keep = dY.clone()
    out = native.relu_backward(dY, Y)
    assert torch.equal(dY, keep)
    assert torch.equal(out, torch.where(Y > 0, dY, torch.zeros_like(dY)))
    assert torch.equal(out, native.relu_backward_(dY.clone(), Y))
    assert torch.equal(native.relu_backward(dY, Y, 1.25), torch.where(Y > 0, dY * 1.25, torch.zeros_like(dY)))
    with pytest.raises(ValueError):
        native.relu_backward(dY, Y[:, :-1] if N > 1 else Y.new_zeros(M, 2))


@pytest.mark.parametrize("biases", [(True, True), (True, False), (False, False)])
def test_gather_columns_concat_autograd_block(gpu, biases):
    """GatherColumnsConcatFn (both id-major embedding parameters + the concat as one gather) == torch's
    cat(Wa.t()[ia] + ba, Wb.t()[ib] + bb): forward bit-exact, weight gradients 4e-6 (float atomics), bias gradients 1e-5;
    different widths, hot ids, optional biases."""
    from deeprecommendation_amd.autograd import GatherColumnsConcatFn
    g = torch.Generator().manual_seed(2)
    Ea, Eb, Ua, Ub, B = 64, 32, 400, 150, 2500
    Wa0, Wb0 = torch.randn(Ea, Ua, generator=g), torch.randn(Eb, Ub, generator=g)
    ba0 = torch.randn(Ea, generator=g) if biases[0] else None
    bb0 = torch.randn(Eb, generator=g) if biases[1] else None
    ia, ib = torch.randint(0, Ua, (B,), generator=g), torch.randint(0, Ub, (B,), generator=g)
    ia[:300], ib[100:500] = 3, 149
    dY = torch.randn(B, Ea + Eb, generator=g)
    # id-major storage ([U, E] buffers), nn.Linear shape [E, U] as transpose views
    Wa1, Wb1 = Wa0.t().contiguous().to(gpu).t().requires_grad_(), Wb0.t().contiguous().to(gpu).t().requires_grad_()
    ba1 = ba0.clone().to(gpu).requires_grad_() if biases[0] else None
    bb1 = bb0.clone().to(gpu).requires_grad_() if biases[1] else None
    y1 = GatherColumnsConcatFn.apply(Wa1, ba1, ia.to(gpu), Wb1, bb1, ib.to(gpu))
    y1.backward(dY.to(gpu))
    Wa2, Wb2 = Wa0.clone().requires_grad_(), Wb0.clone().requires_grad_()
    ba2 = ba0.clone().requires_grad_() if biases[0] else None
    bb2 = bb0.clone().requires_grad_() if biases[1] else None
    ya, yb = Wa2.t()[ia], Wb2.t()[ib]
    y2 = torch.cat((ya + ba2 if biases[0] else ya, yb + bb2 if biases[1] else yb), dim=1)
    y2.backward(dY)
    assert torch.equal(y1.detach().cpu(), y2.detach())
    for got, ref in ((Wa1.grad, Wa2.grad), (Wb1.grad, Wb2.grad)):
        assert got.shape == ref.shape
        # up to 400 duplicates of one id are added by float atomics in any order: ~sqrt(400) * 2^-24 of the sum
        assert float((got.cpu() - ref).abs().max()) <= 4e-6 * float(ref.abs().max()) + 1e-6
    for got, ref, has in ((ba1, ba2, biases[0]), (bb1, bb2, biases[1])):
        if has:
            assert float((got.grad.cpu() - ref.grad).abs().max()) <= 1e-5 * float(ref.grad.abs().max())


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_fused_adam_matches_torch_adam(gpu, wd):
    """deeprecommendation_amd.optim.FusedAdam (one kernel per tensor) follows torch.optim.Adam (train.py:55) step for
    step: 6 updates on tensors of odd sizes (vector body + scalar tail), with and without weight decay."""
    from deeprecommendation_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(1)
    shapes = [(64, 1001), (7,), (256, 128), (1,)]
    p_ref = [torch.randn(s, generator=g).requires_grad_() for s in shapes]
    p_hip = [p.detach().clone().to(gpu).requires_grad_() for p in p_ref]
    o_ref = torch.optim.Adam(p_ref, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    o_hip = FusedAdam(p_hip, lr=3e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd)
    for step in range(6):
        for pr, ph in zip(p_ref, p_hip):
            gr = torch.randn(pr.shape, generator=g) * (0.0 if step == 3 else 1.0)   # a step with zero gradient still moves p
            pr.grad = gr.clone()
            ph.grad = gr.clone().to(gpu)
        o_ref.step()
        o_hip.step()
        for pr, ph in zip(p_ref, p_hip):
            assert float((ph.detach().cpu() - pr.detach()).abs().max()) <= 2e-6 * float(pr.detach().abs().max()) + 1e-7


def test_training_steps_with_fused_adam_track_torch(gpu):
    """Five BasicNCF training steps (no dropout) on the HIP blocks + FusedAdam follow the same steps on torch ops +
    torch.optim.Adam: losses within 1e-5 relative."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.optim import FusedAdam
    torch.manual_seed(5)
    U, I, B = 900, 400, 2048
    base = BasicNCF(item_dim=I, user_dim=U, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128], dropout_rate=None)
    state = {k: v.clone() for k, v in base.state_dict().items()}
    g = torch.Generator().manual_seed(6)
    batches = [(torch.randint(0, U, (B,), generator=g), torch.randint(0, I, (B,), generator=g), torch.rand(B, 1, generator=g) * 5)
               for _ in range(5)]
    losses = {}
    for mode in ("hip", "torch"):
        m = BasicNCF(item_dim=I, user_dim=U, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128], dropout_rate=None)
        m.load_state_dict(state)
        m.to(gpu).train()
        m.train_with_torch_ops = mode == "torch"
        opt = FusedAdam(m.parameters(), lr=1e-3) if mode == "hip" else torch.optim.Adam(m.parameters(), lr=1e-3)
        out = []
        for u, i, y in batches:
            opt.zero_grad(set_to_none=True)
            loss = torch.nn.functional.mse_loss(m(u.to(gpu), i.to(gpu)), y.to(gpu), reduction="sum")
            loss.backward()
            opt.step()
            out.append(float(loss))
        losses[mode] = out
    for a, b in zip(losses["hip"], losses["torch"]):
        assert abs(a - b) <= 1e-5 * abs(b)
    assert losses["hip"][-1] < losses["hip"][0]


def _toy_files(n=6000, U=300, I=120, seed=0):
    import pandas as pd
    rng = np.random.default_rng(seed)
    u, i = rng.integers(1, U + 1, n), rng.integers(1, I + 1, n)
    r = np.clip(np.round(((u % 5) + (i % 3)) * 0.5 + 1 + rng.normal(0, 0.2, n), 1), 0.5, 5.0)
    frame = pd.DataFrame({"userId": u, "movieId": i, "rating": r})
    cut = n * 4 // 5
    return frame.iloc[:cut].reset_index(drop=True), frame.iloc[cut:].reset_index(drop=True), U, I


def test_train_model_resident_loop_equals_dataloader_loop(gpu, tmp_path):
    """train_model (reference train.py:21-227 signature): with shuffling off, the device-resident epoch (ids uploaded once,
    batches gathered on the GPU, loss kept on the device) and the DataLoader epoch see the same batches -> the same
    training / validation curves; FusedAdam vs torch Adam and HIP autograd blocks on both sides."""
    from deeprecommendation_amd.content_providers.index_providers import IndexProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.neural_collaborative_filtering.train import train_model
    tr, va, U, I = _toy_files()
    prov = IndexProvider(np.arange(1, U + 1), np.arange(1, I + 1))
    curves = {}
    for resident in (True, False):
        torch.manual_seed(0)
        m = BasicNCF(item_dim=I, user_dim=U, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128], dropout_rate=0.0)
        curves[resident] = train_model(m, FixedPointwiseDataset(tr, prov), FixedPointwiseDataset(va, prov), lr=2e-3, weight_decay=1e-5,
                                       batch_size=512, val_batch_size=1024, early_stop=True, final_model_path=str(tmp_path / "f.pt"),
                                       checkpoint_model_path=str(tmp_path / "c.pt"), max_epochs=4, device=gpu, resident=resident,
                                       shuffle=False, verbose=False)
    a, b = curves[True], curves[False]
    assert len(a["train_loss"]) == len(b["train_loss"]) >= 2
    np.testing.assert_allclose(a["train_loss"], b["train_loss"], rtol=2e-4)
    np.testing.assert_allclose(a["val_loss"], b["val_loss"], rtol=2e-4)
    assert a["train_loss"][-1] < a["train_loss"][0]


def test_train_model_learns_restores_best_checkpoint_and_saves(gpu, tmp_path):
    """Shuffled resident epochs with dropout: the loss falls; early stopping restores the best epoch's weights (the final
    file holds the checkpoint's state, as train.py:196-199 + :222 leave it) and the checkpoint format loads back."""
    from deeprecommendation_amd.content_providers.index_providers import IndexProvider
    from deeprecommendation_amd.neural_collaborative_filtering.datasets.fixed_datasets import FixedPointwiseDataset
    from deeprecommendation_amd.neural_collaborative_filtering.eval import eval_model
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.neural_collaborative_filtering.train import train_model
    from deeprecommendation_amd.neural_collaborative_filtering.util import load_model
    tr, va, U, I = _toy_files(seed=1)
    prov = IndexProvider(np.arange(1, U + 1), np.arange(1, I + 1))
    torch.manual_seed(1)
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=64, user_emb=64, mlp_dense_layers=[256, 128], dropout_rate=0.2)
    vds = FixedPointwiseDataset(va, prov)
    # lr large enough to overfit / oscillate within a few epochs, patience 0: stops early
    mm = train_model(m, FixedPointwiseDataset(tr, prov), vds, lr=2e-2, weight_decay=0.0, batch_size=256, val_batch_size=1024,
                     early_stop=True, final_model_path=str(tmp_path / "final.pt"), checkpoint_model_path=str(tmp_path / "ckpt.pt"),
                     max_epochs=25, patience=0, max_patience=3, device=gpu, verbose=False)
    assert min(mm["train_loss"]) < mm["train_loss"][0]
    assert len(mm["val_loss"]) == len(mm["train_loss"]) == len(mm["val_ndcg"]) <= 25
    best = int(np.argmin(mm["val_loss"]))
    reloaded = load_model(str(tmp_path / "final.pt"), BasicNCF).to(gpu)
    res = eval_model(reloaded, vds, 1024, device=gpu)
    assert abs(res["mse"] - mm["val_loss"][best]) <= 1e-4 * mm["val_loss"][best]
    with pytest.raises(NotImplementedError):
        train_model(m, object(), vds, 1e-3, 0, 8, 8, False, device=gpu)


def test_fused_adam_invalidates_the_scoring_caches(gpu):
    """FusedAdam writes parameters through raw pointers; the eval path keeps tables / packed weights per parameter
    VERSION.  After every optimiser step the eval-mode HIP scores must be those of the CURRENT weights (checked against
    the CPU oracle on a copy of the weights), and must differ from the scores before the step."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.basic_ncf import BasicNCF
    from deeprecommendation_amd.optim import FusedAdam
    from oracle import ncf_oracle as O
    from test_gpu_basic import assert_close
    torch.manual_seed(11)
    U, I, E, B = 400, 150, 64, 512
    m = BasicNCF(item_dim=I, user_dim=U, item_emb=E, user_emb=E, mlp_dense_layers=[256, 128], dropout_rate=None).to(gpu)
    opt = FusedAdam(m.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(12)
    u = torch.randint(0, U, (B,), generator=g).to(gpu)
    i = torch.randint(0, I, (B,), generator=g).to(gpu)
    y = (torch.rand(B, 1, generator=g) * 5).to(gpu)
    prev = None
    for step in range(3):
        m.eval()
        with torch.no_grad():
            scores = m(u, i)                       # builds / refreshes the cached tables and packed MLP
        state = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
        assert_close(scores, O.basic_ncf_forward_indexed(state, u.cpu(), i.cpu()))
        if prev is not None:
            assert not torch.equal(scores, prev), "scores did not move after an optimiser step: stale caches"
        prev = scores.clone()
        m.train()
        opt.zero_grad(set_to_none=True)
        torch.nn.functional.mse_loss(m(u, i), y, reduction="sum").backward()
        opt.step()


# ----------------------------------------------------------------------------- GraphNCF training step on the HIP blocks
def _train_graph(n_items, n_users, n_inter, seed, binary=False):
    g = torch.Generator().manual_seed(seed)
    key = torch.unique(torch.randint(0, n_users, (n_inter,), generator=g) * n_items + (torch.rand(n_inter, generator=g) ** 2 * n_items).long())
    u, i = key // n_items + n_items, key % n_items
    a = None if binary else torch.randn(u.numel(), generator=g)
    return torch.stack([u, i]), torch.stack([i, u]), a


@pytest.mark.parametrize("hetero", [True, False])
@pytest.mark.parametrize("binary,concat,dot", [(False, False, False), (True, True, False), (False, False, True)])
@pytest.mark.parametrize("mask_targets", [True, False])
def test_graph_ncf_training_step_gradients(gpu, hetero, binary, concat, dot, mask_targets):
    """GraphNCF training step (gnn_ncf.py:298-367 under autograd, train.py:95-110) on the HIP blocks — hoisted Linear,
    SpMM forward, SpMM^T backward on the CSR by source, row gather / scatter, MLP — against the same model run with torch ops
    on the CPU: loss and every parameter gradient, incl. the batch's target edges masked out of both directions with the
    degrees recomputed.  Dropout off (the masks are RNG-dependent: see the adjoint test below for the fused message dropout)."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, GraphNCF
    n_items, n_users, D, B = 40, 300, 64, 256
    u2i, i2u, a = _train_graph(n_items, n_users, 4000, seed=3, binary=binary)
    torch.manual_seed(5)
    m_cpu = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=2, hetero=hetero, node_emb=D, mlp_dense_layers=[128],
                     dropout_rate=0.0, concat=concat, use_dot_product=dot).train()
    m_gpu = copy.deepcopy(m_cpu).to(gpu).train()
    g = torch.Generator().manual_seed(6)
    pick = torch.randint(0, u2i.shape[1], (B,), generator=g)         # batch pairs that ARE edges: the masking has work to do
    users, items = u2i[0][pick], u2i[1][pick]
    y = torch.rand(B, 1, generator=g) * 5

    def graph(dev):
        return GraphData(user2item_edge_index=u2i.to(dev), item2user_edge_index=i2u.to(dev), user2item_edge_attr=None if a is None else a.to(dev),
                         item2user_edge_attr=None if a is None else a.clone().to(dev), num_items=n_items, num_users=n_users)

    out_c = m_cpu(graph("cpu"), users, items, "cpu", mask_targets)
    loss_c = torch.nn.functional.mse_loss(out_c, y, reduction="sum")
    loss_c.backward()
    out_g = m_gpu(graph(gpu), users.to(gpu), items.to(gpu), gpu, mask_targets)
    assert out_g.requires_grad
    loss_g = torch.nn.functional.mse_loss(out_g, y.to(gpu), reduction="sum")
    loss_g.backward()
    assert abs(float(loss_g) - float(loss_c)) <= 2e-5 * abs(float(loss_c))
    _grads_close(m_gpu, m_cpu, rtol=5e-5)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_spmm_message_dropout_is_adjoint_and_unbiased(gpu, p):
    """Fused per-(edge, feature) message dropout (ncf_spmm_csr_dropout): the forward on the CSR by destination and the
    backward on the CSR by source regenerate the SAME mask — <A_mask z, g> == <z, A_mask^T g> to fp32 rounding — the kept
    fraction is 1 - p, kept values are scaled by 1 / (1 - p'), another seed gives another mask, p = 0 is the plain SpMM."""
    from deeprecommendation_amd.autograd import SpmmFn
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, PreparedGraph
    n_items, n_users, D = 50, 2000, 128
    u2i, i2u, a = _train_graph(n_items, n_users, 60000, seed=8)     # hub items: split rows + partial-sum tree in both directions
    graph = GraphData(user2item_edge_index=u2i.to(gpu), item2user_edge_index=i2u.to(gpu), user2item_edge_attr=a.to(gpu),
                      item2user_edge_attr=a.clone().to(gpu), num_items=n_items, num_users=n_users)
    prep = PreparedGraph(graph, hetero=False, seg_len=64)
    N = n_items + n_users
    gen = torch.Generator(device=gpu).manual_seed(1)
    z = torch.randn(N, D, device=gpu, generator=gen).requires_grad_(True)
    gy = torch.randn(N, D, device=gpu, generator=gen)
    y = SpmmFn.apply(z, prep, prep.coef, (p, 1234))
    (dz,) = torch.autograd.grad(y, z, gy)
    lhs, rhs = float((y.double() * gy.double()).sum()), float((z.detach().double() * dz.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(abs(lhs), abs(rhs), 1.0)
    again = SpmmFn.apply(z, prep, prep.coef, (p, 1234))
    assert torch.equal(y, again)                                     # same seed, same mask, same order: bitwise
    other = SpmmFn.apply(z, prep, prep.coef, (p, 99))
    assert not torch.equal(y, other)
    # unbiased: with z = 1 and coef = 1, y[n, f] / deg[n] is the kept fraction / (1 - p') -> 1
    ones = torch.ones(N, D, device=gpu)
    cnt = SpmmFn.apply(ones, prep, torch.ones_like(prep.coef), (p, 7)).sum() / (prep.col.numel() * D)
    assert abs(float(cnt) - 1.0) < 5e-3
    kept = SpmmFn.apply(ones, prep, torch.ones_like(prep.coef), (p, 7))
    thr = round(p * 65536)
    scale = 65536.0 / (65536 - thr)
    assert bool(((kept / scale - (kept / scale).round()).abs() < 1e-3).all())    # every destination sums whole kept entries
    plain = prep.csr.spmm(z.detach())
    assert torch.equal(SpmmFn.apply(z.detach(), prep, prep.coef, None), plain)


def test_graph_ncf_trains_with_message_dropout_on_hip_blocks(gpu):
    """Default hyper-parameters (dropout_rate 0.2 -> per-edge message dropout 0.1 inside the conv, gnn_ncf.py:218) on the HIP
    blocks + FusedAdam: the loss goes down, and the eval-mode HIP scoring follows the updated weights."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.gnn_ncf import GraphData, GraphNCF
    from deeprecommendation_amd.optim import FusedAdam
    n_items, n_users, D = 40, 400, 64
    u2i, i2u, a = _train_graph(n_items, n_users, 6000, seed=11)
    graph = GraphData(user2item_edge_index=u2i.to(gpu), item2user_edge_index=i2u.to(gpu), user2item_edge_attr=a.to(gpu),
                      item2user_edge_attr=a.clone().to(gpu), num_items=n_items, num_users=n_users)
    torch.manual_seed(0)
    m = GraphNCF(item_dim=n_items, user_dim=n_users, num_gnn_layers=2, hetero=True, node_emb=D, mlp_dense_layers=[128]).to(gpu).train()
    opt = FusedAdam(m.parameters(), lr=3e-3)
    users, items = u2i[0].to(gpu), u2i[1].to(gpu)
    y = (a.to(gpu) + 2.5).view(-1, 1)
    losses = []
    for step in range(30):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(graph, users, items, gpu, True), y)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.7 * losses[0], losses[::5]
    m.eval()
    with torch.no_grad():
        s1 = m(graph, users[:64], items[:64], gpu)
    assert torch.isfinite(s1).all()


# ----------------------------------------------------------------------------- AttentionNCF training step on the HIP blocks
def _attention_batch(B, I, Fdim, seed):
    g = torch.Generator().manual_seed(seed)
    rated = (torch.rand(I, Fdim, generator=g) < 0.3).float() * torch.rand(I, Fdim, generator=g)
    cand = (torch.rand(B, Fdim, generator=g) < 0.3).float() * torch.rand(B, Fdim, generator=g)
    cand[1] = rated[3]                                   # a candidate that is one of the rated items: masked while training
    um = torch.where(torch.rand(B, I, generator=g) < 0.5, torch.randint(1, 11, (B, I), generator=g).float() * 0.5 - 2.9, torch.zeros(B, I))
    um[1, 3] = 1.5                                       # ... and the user did rate it
    um[2] = 0                                            # a user without rated items: softmax NaN -> 0 row
    y = torch.rand(B, 1, generator=g) * 5
    return cand, rated, um, y


@pytest.mark.parametrize("cosine,att_dense", [(False, 32), (True, None), (False, 128)])
@pytest.mark.parametrize("training", [True, False])
def test_attention_ncf_training_step_gradients(gpu, cosine, att_dense, training):
    """AttentionNCF training step (attention_ncf.py:136-224 under autograd) on the HIP blocks — Linear layers, the attention
    kernel and ITS BACKWARD kernel (softmax, rating-weighted sum, AttentionNet), MLP — against the same model with torch ops
    on the CPU: loss, attention weights and every parameter gradient; with the train-only self-attention mask, an all-unrated
    user and dropout off (masks are RNG-dependent; the fused hidden dropout has its own test)."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    B, I, Fdim = 48, 40, 36
    cand, rated, um, y = _attention_batch(B, I, Fdim, seed=21)
    torch.manual_seed(4)
    m_cpu = AttentionNCF(item_dim=Fdim, item_emb=64, user_emb=64, att_dense=att_dense, mlp_dense_layers=[128],
                         use_cos_sim_instead=cosine, dropout_rate=0.0)
    m_cpu.train(training)
    m_gpu = copy.deepcopy(m_cpu).to(gpu)
    m_gpu.train(training)
    out_c, w_c = m_cpu(cand, rated, um, return_attention_weights=True)
    loss_c = torch.nn.functional.mse_loss(out_c, y, reduction="sum")
    loss_c.backward()
    out_g, w_g = m_gpu(cand.to(gpu), rated.to(gpu), um.to(gpu), return_attention_weights=True)
    assert out_g.requires_grad
    loss_g = torch.nn.functional.mse_loss(out_g, y.to(gpu), reduction="sum")
    loss_g.backward()
    assert abs(float(loss_g.detach()) - float(loss_c.detach())) <= 2e-5 * abs(float(loss_c.detach()))
    assert float((w_g.cpu() - w_c).abs().max()) <= 1e-5
    if training:
        assert float(w_c[1, 3]) == 0.0 and float(w_g[1, 3]) == 0.0      # the candidate's own rated entry is masked
    assert float(w_g[2].abs().sum()) == 0.0
    # AttentionNet's output bias shifts every score of a row alike: it cancels in the softmax, its gradient is exactly zero
    _grads_close(m_gpu, m_cpu, rtol=1e-4, exactly_zero=("AttentionNet.3.bias",))


def test_attention_hidden_dropout_forward_and_backward_agree(gpu):
    """AttentionNet's hidden dropout on the HIP kernels (ncf_attn_forward_dropout / ncf_attn_backward with the same seed): the
    backward is the gradient of THAT forward — central differences through the seeded kernel on a handful of pc / pr / w1
    elements — a different seed gives different attention weights, p = 0 equals the plain kernel bit for bit."""
    from deeprecommendation_amd import native
    g = torch.Generator().manual_seed(3)
    B, I, A, Fd, nnz = 6, 50, 32, 16, 20
    pc = (torch.randn(B, A, generator=g) * 0.5).double()
    pr = (torch.randn(I, A, generator=g) * 0.5).double()
    w1 = torch.randn(A, generator=g).double()
    feat = torch.randn(I, Fd, generator=g).to(gpu)
    col = torch.stack([torch.randperm(I, generator=g)[:nnz].sort().values for _ in range(B)]).reshape(-1).to(torch.int32).to(gpu)
    val = (torch.randint(1, 11, (B * nnz,), generator=g).float() * 0.5 - 2.9).to(gpu)
    rowptr = torch.arange(0, (B + 1) * nnz, nnz, dtype=torch.int64, device=gpu)
    gout = torch.randn(B, Fd, generator=g).to(gpu)
    drop = (0.3, 4242)

    def loss(pc_, pr_, w1_, d=drop):
        out, wts = native.attn_forward(native.ATT_MLP, pc_.float().to(gpu).contiguous(), pr_.float().to(gpu).contiguous(),
                                       w1_.float().to(gpu).contiguous(), 0.2, rowptr, col, val, feat, dropout=d)
        return float((out.double() * gout.double()).sum()), wts

    l0, wts = loss(pc, pr, w1)
    d_pc, d_pr, d_w1, d_feat = native.attn_backward(native.ATT_MLP, pc.float().to(gpu), pr.float().to(gpu), w1.float().to(gpu), rowptr, col, val,
                                                    feat, wts, gout, dropout=drop)
    eps = 1e-2
    for (name, t, grad, idx) in (("pc", pc, d_pc, (2, 5)), ("pc", pc, d_pc, (4, 31)), ("pr", pr, d_pr, (int(col[3]), 7)),
                                 ("pr", pr, d_pr, (int(col[45]), 0)), ("w1", w1, d_w1, (9,)), ("w1", w1, d_w1, (30,))):
        tp, tm = t.clone(), t.clone()
        tp[idx] += eps
        tm[idx] -= eps
        args = {"pc": (tp, pr, w1), "pr": (pc, tp, w1), "w1": (pc, pr, tp)}[name], {"pc": (tm, pr, w1), "pr": (pc, tm, w1), "w1": (pc, pr, tm)}[name]
        num = (loss(*args[0])[0] - loss(*args[1])[0]) / (2 * eps)
        ana = float(grad[idx])
        assert abs(num - ana) <= 3e-2 * max(abs(num), abs(ana)) + 2e-3, (name, idx, num, ana)
    _, wts_other = loss(pc, pr, w1, (0.3, 7))
    assert not torch.equal(wts, wts_other)
    out_p0, w_p0 = native.attn_forward(native.ATT_MLP, pc.float().to(gpu), pr.float().to(gpu), w1.float().to(gpu), 0.2, rowptr, col, val, feat, dropout=(0.0, 1))
    out_pl, w_pl = native.attn_forward(native.ATT_MLP, pc.float().to(gpu), pr.float().to(gpu), w1.float().to(gpu), 0.2, rowptr, col, val, feat)
    assert torch.equal(out_p0, out_pl) and torch.equal(w_p0, w_pl)


def test_attention_ncf_trains_with_default_dropout_on_hip_blocks(gpu):
    """Default hyper-parameters (dropout_rate 0.2: AttentionNet's hidden dropout and the MLP's) on the HIP blocks + FusedAdam:
    the loss goes down and the eval-mode HIP scoring (grouped kernel, scaled operands) follows the trained weights."""
    from deeprecommendation_amd.neural_collaborative_filtering.models.attention_ncf import AttentionNCF
    from deeprecommendation_amd.optim import FusedAdam
    from oracle import ncf_oracle as O
    from test_gpu_basic import assert_close
    B, I, Fdim = 256, 60, 48
    cand, rated, um, _ = _attention_batch(B, I, Fdim, seed=5)
    y = (cand[:, :4].sum(1, keepdim=True) + 1.0)
    torch.manual_seed(1)
    m = AttentionNCF(item_dim=Fdim, item_emb=64, user_emb=64, att_dense=64, mlp_dense_layers=[128]).to(gpu).train()
    opt = FusedAdam(m.parameters(), lr=3e-3)
    c, r, u, yy = cand.to(gpu), rated.to(gpu), um.to(gpu), y.to(gpu)
    losses = []
    for step in range(40):
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(m(c, r, u), yy)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.6 * losses[0], losses[::8]
    m.eval()
    with torch.no_grad():
        out = m(c, r, u)
    state = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    assert_close(out, O.attention_ncf_forward(state, cand, rated, um))
