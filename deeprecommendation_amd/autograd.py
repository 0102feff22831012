"""torch.autograd wrappers that run a TRAINING step's forward and backward on the HIP kernels.

Only the differentiable building blocks of the scoring path are wrapped — the embedding gather and the Linear(+ReLU)
layers; loss, dropout masks and the optimizer stay torch ops (the reference's train loop, train.py:95-110, drives them).
  GatherConcatFn : forward ncf_gather_concat;  backward ncf_scatter_add_rows into dense table gradients
  LinearFn       : forward ncf_linear_forward (ReLU fused in the epilogue);
                   backward dX = dY . W (row-streaming GEMM with W^T), dW = dY^T . X (ncf_gemm_tn, ordered split over
                   the batch), db = ncf_colsum, ReLU mask = ncf_relu_backward_out
  SpmmFn         : LightGCN propagation y = A z (ncf_spmm_csr on the CSR by destination); backward dz = A^T dy = the SAME
                   kernel on the CSR by source (gnn_ncf.py:74-94 through autograd in the reference); optional per-edge
                   message dropout regenerated from a hash in both directions (ncf_spmm_csr_dropout)
  AttnFn         : grouped item-item attention (ncf_attn_forward_grouped) with its backward (ncf_attn_backward_grouped)
"""
import torch

from . import native


class GatherConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tabA, idxA, tabB, idxB):
        ctx.save_for_backward(idxA, idxB)
        ctx.shapes = (tuple(tabA.shape), tuple(tabB.shape))
        return native.gather_concat(tabA.contiguous(), idxA, tabB.contiguous(), idxB)

    @staticmethod
    def backward(ctx, dX):
        idxA, idxB = ctx.saved_tensors
        (ra, ea), (rb, eb) = ctx.shapes
        dX = dX.contiguous()
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.zeros((ra, ea), dtype=torch.float32, device=dX.device)
            native.scatter_add_rows(dX[:, :ea], idxA, dA)
        if ctx.needs_input_grad[2]:
            dB = torch.zeros((rb, eb), dtype=torch.float32, device=dX.device)
            native.scatter_add_rows(dX[:, ea:], idxB, dB)
        return dA, None, dB, None


class GatherColumnsFn(torch.autograd.Function):
    """x[p, :] = W[:, idx[p]] + b for the nn.Linear-layout embedding parameter W [E, U] (basic_ncf.py:25-33): what
    ``Linear(onehot(idx))`` computes.

    When W is stored id-major (util.row_major_embedding_: the transpose view of a contiguous [U, E] buffer — what
    BasicNCF / MF construct) this is a ROW gather, and the backward scatters 4·E-byte gradient rows (full-rate atomics)
    into a zeroed [U, E] buffer whose transpose VIEW is returned: the gradient has the parameter's own strides, so Adam
    runs elementwise over the raw buffers.  For a plain contiguous [E, U] weight the column kernels are used.  torch's
    ``W.t()[idx]`` backward goes through a sort and two table-sized copies either way."""

    @staticmethod
    def forward(ctx, weight, bias, idx):
        ctx.save_for_backward(idx)
        ctx.wshape = tuple(weight.shape)
        ctx.has_bias = bias is not None
        ctx.row_major = weight.dim() == 2 and weight.t().is_contiguous() and not weight.is_contiguous()
        if ctx.row_major:
            x = native.gather_concat(weight.t(), idx)
            return x if bias is None else x.add_(bias)
        return native.gather_cols(weight.contiguous(), None if bias is None else bias.contiguous(), idx)

    @staticmethod
    def backward(ctx, dX):
        (idx,) = ctx.saved_tensors
        dX = dX.contiguous()
        dW = db = None
        if ctx.needs_input_grad[0]:
            E, U = ctx.wshape
            if ctx.row_major:
                dT = torch.zeros((U, E), dtype=torch.float32, device=dX.device)
                native.scatter_add_rows(dX, idx, dT)
                dW = dT.t()
            else:
                dW = torch.zeros((E, U), dtype=torch.float32, device=dX.device)
                native.scatter_add_cols(dX, idx, dW)
        if ctx.has_bias and ctx.needs_input_grad[1]:
            db = native.colsum(dX)
        return dW, db, None


class GatherColumnsConcatFn(torch.autograd.Function):
    """cat(W_a[:, idx_a] + b_a, W_b[:, idx_b] + b_b) for two id-major embedding parameters in ONE gather into the (B, Ea+Eb)
    activation (``basic_ncf.py:37-40``: both embedding Linears, then the concat).  Against two GatherColumnsFn + torch.cat
    it saves the concat copy forward and, backward, the two column-slice copies and one of the two bias reductions: the
    gradient rows are scattered straight from the column halves of dX (the scatter kernel takes a row stride).  Both
    weights must be id-major (util.row_major_embedding_); the caller falls back to GatherColumnsFn + cat otherwise."""

    @staticmethod
    def forward(ctx, wa, ba, idx_a, wb, bb, idx_b):
        ctx.save_for_backward(idx_a, idx_b)
        ctx.shapes = (tuple(wa.shape), tuple(wb.shape))
        ctx.has_bias = (ba is not None, bb is not None)
        x = native.gather_concat(wa.t(), idx_a, wb.t(), idx_b)
        if ba is not None or bb is not None:
            ea, eb = wa.shape[0], wb.shape[0]
            bias = torch.cat((ba if ba is not None else x.new_zeros(ea), bb if bb is not None else x.new_zeros(eb)))
            x.add_(bias)
        return x

    @staticmethod
    def backward(ctx, dX):
        idx_a, idx_b = ctx.saved_tensors
        (ea, ua), (eb, ub) = ctx.shapes
        dX = dX.contiguous()
        d_wa = d_wb = d_ba = d_bb = None
        if ctx.needs_input_grad[0]:
            ta = torch.zeros((ua, ea), dtype=torch.float32, device=dX.device)
            native.scatter_add_rows(dX[:, :ea], idx_a, ta)
            d_wa = ta.t()
        if ctx.needs_input_grad[3]:
            tb = torch.zeros((ub, eb), dtype=torch.float32, device=dX.device)
            native.scatter_add_rows(dX[:, ea:], idx_b, tb)
            d_wb = tb.t()
        if (ctx.has_bias[0] and ctx.needs_input_grad[1]) or (ctx.has_bias[1] and ctx.needs_input_grad[4]):
            db = native.colsum(dX)
            d_ba = db[:ea] if ctx.has_bias[0] and ctx.needs_input_grad[1] else None
            d_bb = db[ea:] if ctx.has_bias[1] and ctx.needs_input_grad[4] else None
        return d_wa, d_ba, None, d_wb, d_bb, None


class SpmmFn(torch.autograd.Function):
    """y (N, D) = sum over the edges into each destination of coef_e * [dropout mask / (1 - p)] * z[src_e]  —  one LightGCN
    aggregation (gnn_ncf.py:52-70, 74-94).  ``prep`` is the PreparedGraph (CSR by destination + its transpose, built once per
    graph); ``coef`` the per-edge coefficients in CSR order (a training step recomputes them when it masks the batch's target
    edges); ``dropout`` = (p, seed) or None.  The coefficients are data, not parameters: no gradient flows to them."""

    @staticmethod
    def forward(ctx, z, prep, coef, dropout):
        ctx.prep, ctx.dropout = prep, dropout
        ctx.save_for_backward(coef)
        drop = None if dropout is None or dropout[0] <= 0 else (dropout[0], dropout[1], None)
        return prep.csr.spmm(z.contiguous(), coef=coef, dropout=drop)

    @staticmethod
    def backward(ctx, dY):
        (coef,) = ctx.saved_tensors
        prep = ctx.prep
        csr_t, eid = prep.transposed()
        drop = None if ctx.dropout is None or ctx.dropout[0] <= 0 else (ctx.dropout[0], ctx.dropout[1], eid)
        dZ = csr_t.spmm(dY.contiguous(), coef=coef[eid.long()], dropout=drop)   # same mask: entry e of A^T carries edge id eid[e]
        return dZ, None, None, None


class AttnFn(torch.autograd.Function):
    """user_emb (B, UE) = bias + sum_e softmax_e(score(b, e)) * val_e * proj[col_e, :]  —  attention_ncf.py:176-216 for one CSR row
    of rated entries per pair, forward (ncf_attn_forward / _dropout) and backward (ncf_attn_backward) on the HIP kernels.
    ``b1`` (AttentionNet's output bias) is taken only to hand it its gradient, which is exactly zero: a shift of every score
    cancels in the softmax.  ``dropout`` = (p, seed) or None: AttentionNet's hidden dropout, regenerated in both kernels."""

    @staticmethod
    def forward(ctx, pc, pr, w1, b1, proj, bias, rowptr, col, val, mode, dropout):
        pc, pr, proj = pc.contiguous(), pr.contiguous(), proj.contiguous()
        w1c = None if w1 is None else w1.contiguous()
        out, wts = native.attn_forward(mode, pc, pr, w1c, 0.0, rowptr, col, val, proj, out_bias=bias, dropout=dropout)
        ctx.mode, ctx.dropout = mode, dropout
        ctx.has = (w1 is not None, b1 is not None, bias is not None)
        ctx.save_for_backward(pc, pr, w1c, proj, rowptr, col, val, wts)
        ctx.mark_non_differentiable(wts)
        return out, wts

    @staticmethod
    def backward(ctx, dout, _dwts):
        pc, pr, w1c, proj, rowptr, col, val, wts = ctx.saved_tensors
        d_pc, d_pr, d_w1, d_proj = native.attn_backward(ctx.mode, pc, pr, w1c, rowptr, col, val, proj, wts, dout, dropout=ctx.dropout)
        has_w1, has_b1, has_bias = ctx.has
        d_b1 = torch.zeros(1, dtype=torch.float32, device=dout.device) if has_b1 else None
        d_bias = native.colsum(dout.contiguous()) if has_bias else None
        return d_pc, d_pr, (d_w1 if has_w1 else None), d_b1, d_proj, d_bias, None, None, None, None, None


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = x.contiguous()
        w = weight.contiguous()
        y = native.linear_act(x, w, None if bias is None else bias.contiguous(), relu)
        ctx.relu = bool(relu)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dY):
        x, w, y = ctx.saved_tensors
        dY = dY.contiguous()
        if ctx.relu:
            dY = native.relu_backward(dY, y)   # one pass into a new buffer: the incoming gradient is not ours to modify
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            if w.shape[0] == 1:
                dX = dY * w          # 1-wide output layer: dX is the outer product (one exact product per element, no GEMM)
            else:
                dX = native.linear_act(dY, w.t().contiguous(), None, False)   # dX = dY . W
        if ctx.needs_input_grad[1]:
            dW = native.gemm_tn(dY, x)                                   # dW = dY^T . X
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = native.colsum(dY)
        return dX, dW, db, None


class LinearReluDropoutFn(torch.autograd.Function):
    """dropout(relu(x W^T + b), p) — the reference's hidden-layer block in training mode (util.py:12-17) — as ONE autograd
    node.  Forward: the HIP Linear with the ReLU in its epilogue, then torch's dropout (its Philox stream, so a seeded run
    draws the masks it would draw with plain torch ops).  Only the OUTPUT z is kept: z > 0 exactly where the ReLU was
    active and the element survived, so the backward of both is ``dz * 1/(1-p)`` masked by ``z > 0`` — one pass of
    ncf_relu_backward_out instead of torch's masked-scale kernel plus the ReLU mask, and neither the pre-dropout
    activation nor the mask is stored."""

    @staticmethod
    def forward(ctx, x, weight, bias, p):
        x = x.contiguous()
        w = weight.contiguous()
        z = torch.nn.functional.dropout(native.linear_act(x, w, None if bias is None else bias.contiguous(), True), p, True)
        ctx.scale = 1.0 / (1.0 - p)
        ctx.has_bias = bias is not None
        ctx.save_for_backward(x, w, z)
        return z

    @staticmethod
    def backward(ctx, dZ):
        x, w, z = ctx.saved_tensors
        dY = native.relu_backward(dZ.contiguous(), z, ctx.scale)
        dX = dW = db = None
        if ctx.needs_input_grad[0]:
            dX = native.linear_act(dY, w.t().contiguous(), None, False)
        if ctx.needs_input_grad[1]:
            dW = native.gemm_tn(dY, x)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = native.colsum(dY)
        return dX, dW, db, None


def mlp_train(seq: torch.nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """A build_MLP_layers Sequential (Linear, [ReLU, Dropout?, Linear]*) applied with the HIP autograd blocks: every
    Linear that is followed by a ReLU fuses it, and a training-mode Dropout behind that joins the same node
    (LinearReluDropoutFn); any other module runs as the torch op it is."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, torch.nn.Linear):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], torch.nn.ReLU)
            drop = mods[i + 2] if fuse and i + 2 < len(mods) and isinstance(mods[i + 2], torch.nn.Dropout) else None
            if drop is not None and drop.training and 0.0 < drop.p < 1.0 and not drop.inplace:
                x = LinearReluDropoutFn.apply(x, m.weight, m.bias, float(drop.p))
                i += 3
                continue
            x = LinearFn.apply(x, m.weight, m.bias, fuse)
            i += 2 if fuse else 1
        else:
            x = m(x)
            i += 1
    return x
