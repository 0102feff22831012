"""HIP-graph replay of a launch-bound forward.

An evaluation / serving step of AttentionNCF is a dozen short kernels (27 + 57 + 23 + ... us at BASELINE config 3):
enqueueing them from Python costs as much as running them (measured: 145 us of host time for 145 us of GPU time per
step, tools/step_breakdown_cfg3.py).  ``GraphedForward`` captures one call of ``fn`` on static input buffers into a HIP
graph (torch.cuda.CUDAGraph; the library's launches go to torch's current stream, so they are captured like torch's
own) and replays it: per step the host copies the new batch into the static buffers and issues ONE graph launch.

Inputs may be tensors or ``SparseRatings``; a replayed batch must have the example's tensor shapes, except that a
``SparseRatings`` may have fewer rows / entries than the example's capacity (missing rows are made empty).  Anything the
captured code decides on the host (kernel choice, grid sizes) is frozen at capture time — every kernel of this library
sizes its grid from tensor shapes only, never from device data.
"""
from __future__ import annotations

import torch

from .neural_collaborative_filtering.models.attention_ncf import SparseRatings


def _clone(x):
    if isinstance(x, SparseRatings):
        return SparseRatings(x.rowptr.clone(), x.col.clone(), x.val.clone(), x.num_items,
                             None if x.pair_row is None else x.pair_row.clone())
    if isinstance(x, torch.Tensor):
        return x.clone()
    return x


def _copy_into(dst, src):
    if isinstance(dst, SparseRatings):
        if not isinstance(src, SparseRatings) or (dst.pair_row is None) != (src.pair_row is None) or dst.num_items != src.num_items:
            raise ValueError("replayed SparseRatings does not match the captured one")
        R, nnz = src.rowptr.numel() - 1, src.col.numel()
        if R + 1 > dst.rowptr.numel() or nnz > dst.col.numel():
            raise ValueError("replayed SparseRatings exceeds the captured capacity")
        dst.rowptr[:R + 1].copy_(src.rowptr, non_blocking=True)
        if R + 1 < dst.rowptr.numel():
            dst.rowptr[R + 1:] = src.rowptr[-1]            # rows beyond the batch's are empty
        dst.col[:nnz].copy_(src.col, non_blocking=True)
        dst.val[:nnz].copy_(src.val, non_blocking=True)
        if dst.pair_row is not None:
            if dst.pair_row.numel() != src.pair_row.numel():
                raise ValueError("replayed batch has a different number of pairs")
            dst.pair_row.copy_(src.pair_row, non_blocking=True)
    elif isinstance(dst, torch.Tensor):
        if dst.shape != src.shape or dst.dtype != src.dtype:
            raise ValueError(f"replayed input {tuple(src.shape)}/{src.dtype} does not match the captured {tuple(dst.shape)}/{dst.dtype}")
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src, non_blocking=True)
    elif dst != src:
        raise ValueError("non-tensor arguments are frozen at capture time")


class GraphedForward:
    def __init__(self, fn, example_inputs, warmup: int = 3, static_inputs: bool = False):
        """``static_inputs=True`` uses the example tensors themselves as the static buffers (no clone): the caller
        then updates them in place (e.g. a catalogue that never changes)."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedForward needs a GPU; deeprecommendation_amd has no CPU path")
        self.fn = fn
        self.static_in = list(example_inputs) if static_inputs else [_clone(x) for x in example_inputs]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):                # caches (packed weights, catalogue projections) fill up here
                fn(*self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        if len(inputs) != len(self.static_in):
            raise ValueError("wrong number of inputs")
        for dst, src in zip(self.static_in, inputs):
            _copy_into(dst, src)
        self.graph.replay()
        return self.static_out
