"""Fused Adam on the HIP path: ``torch.optim.Adam``'s update (the optimiser of the reference's train.py:55) as ONE kernel
per parameter tensor (libncf_hip.so ``ncf_adam_step``) instead of torch's ~9 foreach kernels.  At BASELINE config 2 the
dense Adam over the 71 M embedding parameters is the largest part of a training step (the Linear-layout embeddings make
every gradient dense, and Adam moves every moment every step even where the gradient is zero)."""
from __future__ import annotations

import torch

from . import native


class FusedAdam(torch.optim.Optimizer):
    """Drop-in for ``torch.optim.Adam(params, lr, betas, eps, weight_decay)`` (amsgrad / maximize / capturable are
    not supported) on fp32 CUDA parameters; any other parameter raises (there is no CPU path in this package)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                    raise RuntimeError("FusedAdam needs dense fp32 parameters on the GPU")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)          # preserve_format: the parameter's own strides
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                m, v, g = st["exp_avg"], st["exp_avg_sq"], p.grad
                # the update is elementwise: any layout works as long as all four tensors share it.  Embedding weights are
                # stored id-major (transpose views, util.row_major_embedding_): run on the contiguous transposes.
                if p.dim() == 2 and not p.is_contiguous() and p.t().is_contiguous():
                    pv, mv, vv = p.t(), m.t(), v.t()
                    gv = g.t() if g.t().is_contiguous() else g.t().contiguous()
                elif p.is_contiguous():
                    pv, mv, vv = p, m, v
                    gv = g if g.is_contiguous() else g.contiguous()
                else:
                    raise RuntimeError("FusedAdam needs contiguous (or transposed-contiguous) parameters")
                native.adam_step_(pv, gv, mv, vv, group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"])
                # the kernel writes through raw pointers: tell autograd / the models' derived-tensor caches (tables, packed
                # MLP blobs, propagated graph tables are keyed on (data_ptr, _version)) that the parameter changed
                torch.autograd.graph.increment_version(p)
        return loss
