"""Multi-GPU forms of the scoring path, one process per GPU over torch.distributed (RCCL on ROCm).

* ``RowShardedTable`` / ``ShardedBasicNCF`` — BASELINE config 5: embedding tables sharded row-wise over the ranks
  (contiguous row ranges), batch data-parallel.  Per step and per table: unique the requested ids (sorted, so they
  fall into contiguous owner buckets), all-to-all the bucket sizes and the ids, owners gather their rows with the HIP
  K1 kernel, all-to-all the rows back, and the fused scoring kernel reads the received unique rows through the
  inverse map (the (B, E) batch is never materialised).  Levers from SURVEY §7: de-duplication before the exchange,
  optional replication of the small table.  xGMI is point-to-point: all-to-all uses all 7 links at once.
* ``PartitionedLightGCN`` — BASELINE config 4: (a) edge-partitioned + all-reduce of the (N, D) partial sums, the form
  BASELINE names; (b) destination-partitioned (edge-balanced contiguous row blocks) + all-gather of the blocks, which
  moves 1/W of the bytes per rank and reproduces the single-GPU result bit for bit.

Replica scaling of BasicNCF / MF / AttentionNCF needs no code here: every rank holds the whole model and scores its
slice of the batch (bench.py --gpus N).

The local compute is always the HIP library.  ``local_ops`` exists only so the exchange logic can be exercised by the
world-size-2 gloo tests on a CPU-only host; the default (None) is the HIP path and nothing falls back to it silently.
"""
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from . import native


class _HipOps:
    """Local compute used by the sharded paths — thin names over native.*"""

    @staticmethod
    def gather_rows(table, idx):
        return native.gather_concat(table, idx)

    @staticmethod
    def score(tabA, idxA, tabB, idxB, packed, weights, biases):
        if packed is not None and packed.supports(tabA.shape[1], tabB.shape[1]) and packed.dtype == tabA.dtype:
            return native.score_fused(tabA, idxA, tabB, idxB, packed)
        x = native.gather_concat(tabA, idxA, tabB, idxB)
        return native.mlp_forward(x.float() if x.dtype != torch.float32 else x, weights, biases)


def _a2a(out, inp, out_splits, in_splits, group):
    dist.all_to_all_single(out, inp, out_splits, in_splits, group=group)


class RowShardedTable:
    """Rows ``[rank*rpr, min((rank+1)*rpr, total))`` of a (total_rows, E) table live on this rank."""

    def __init__(self, local_rows: torch.Tensor, total_rows: int, group=None, local_ops=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_rows = int(total_rows)
        self.rows_per_rank = (self.total_rows + self.world - 1) // self.world
        lo = self.rank * self.rows_per_rank
        hi = min(lo + self.rows_per_rank, self.total_rows)
        if local_rows.shape[0] != max(hi - lo, 0):
            raise ValueError(f"rank {self.rank} must hold rows [{lo}, {hi}) = {hi - lo} rows, got {local_rows.shape[0]}")
        self.local = local_rows.contiguous()
        self.ops = local_ops or _HipOps
        self.last_stats = {}

    @staticmethod
    def shard_bounds(total_rows: int, world: int, rank: int):
        rpr = (total_rows + world - 1) // world
        return rank * rpr, min((rank + 1) * rpr, total_rows)

    def lookup_unique(self, idx: torch.Tensor):
        """Returns (rows_of_unique_ids (n_unique, E), inverse (B,) int64) with rows[inverse[p]] == table[idx[p]]."""
        if self.world == 1:
            return self.local, idx.contiguous()  # nothing to exchange: the scoring kernel gathers from the table itself
        uniq, inverse = torch.unique(idx, return_inverse=True)  # sorted ascending -> contiguous owner buckets
        dev = idx.device
        owner = torch.div(uniq, self.rows_per_rank, rounding_mode="floor")
        send_counts = torch.bincount(owner, minlength=self.world)
        recv_counts = torch.empty_like(send_counts)
        _a2a(recv_counts, send_counts, None, None, self.group)
        sc, rc = send_counts.tolist(), recv_counts.tolist()  # split sizes must be host ints (one sync per lookup)
        local_ids = (uniq - owner * self.rows_per_rank).contiguous()
        recv_ids = torch.empty(sum(rc), dtype=torch.int64, device=dev)
        _a2a(recv_ids, local_ids, rc, sc, self.group)                      # exchange #1: row ids to their owners
        rows_out = self.ops.gather_rows(self.local, recv_ids)              # owners gather (HIP K1)
        E = self.local.shape[1]
        rows_in = torch.empty((uniq.numel(), E), dtype=self.local.dtype, device=dev)
        _a2a(rows_in, rows_out, sc, rc, self.group)                        # exchange #2: rows back, same bucket order
        self.last_stats = {"requested": int(idx.numel()), "unique": int(uniq.numel()),
                           "remote_rows": int(uniq.numel() - sc[self.rank]),
                           "bytes_received": int((uniq.numel() - sc[self.rank]) * E * self.local.element_size())}
        return rows_in, inverse.contiguous()

    def lookup(self, idx: torch.Tensor) -> torch.Tensor:
        rows, inverse = self.lookup_unique(idx)
        return self.ops.gather_rows(rows, inverse)


class ShardedBasicNCF:
    """BasicNCF scoring with row-sharded embedding tables (tables built from a BasicNCF's weights or given directly).

    ``user_table`` / ``item_table`` are this rank's shards of ``T = W^T + b``; ``item_table`` may instead be the full
    table with ``replicate_items=True`` (the 2.56 GB item table of config 5 fits every GPU; only users are exchanged).
    """

    def __init__(self, user_table, num_users, item_table, num_items, mlp_weights: Sequence[torch.Tensor],
                 mlp_biases: Sequence[Optional[torch.Tensor]], replicate_items=False, group=None, local_ops=None,
                 dtype=None):
        self.ops = local_ops or _HipOps
        self.users = RowShardedTable(user_table, num_users, group, local_ops)
        self.replicate_items = replicate_items
        self.items_full = item_table.contiguous() if replicate_items else None
        self.items = None if replicate_items else RowShardedTable(item_table, num_items, group, local_ops)
        self.weights = [w.detach().float().contiguous() for w in mlp_weights]
        self.biases = [None if b is None else b.detach().float().contiguous() for b in mlp_biases]
        self.packed = None
        if local_ops is None:
            try:
                self.packed = native.PackedMLP(self.weights, self.biases, dtype=dtype or user_table.dtype)
            except native.NativeError as e:
                if e.code != native.NCF_EUNSUPPORTED:
                    raise

    def forward(self, user_pos: torch.Tensor, item_pos: torch.Tensor) -> torch.Tensor:
        """Global positions of this rank's local batch -> (B, 1) scores (cat(user, item) -> MLP, basic_ncf.py:40-41)."""
        urows, uinv = self.users.lookup_unique(user_pos)
        if self.replicate_items:
            irows, iinv = self.items_full, item_pos.contiguous()
        else:
            irows, iinv = self.items.lookup_unique(item_pos)
        return self.ops.score(urows, uinv, irows, iinv, self.packed, self.weights, self.biases)

    __call__ = forward


# ---------------------------------------------------------------------------------------------------------------
class PartitionedLightGCN:
    """Multi-GPU LightGCN propagation for GraphNCF scoring (hetero or not, mean readout).

    mode "dst"  : rank r owns the contiguous destination rows [bounds[r], bounds[r+1]) chosen so that every rank
                  has ~E/W edges (prefix sums of the in-degrees); per layer: hoisted Linear on the local block,
                  all-gather of the (padded) blocks into the full Z, local SpMM.  Per-row sums are identical to the
                  single-GPU kernel's -> bitwise equal results.
    mode "edge" : rank r owns edges r::W of every destination row and the full node table; per layer: hoisted
                  Linear (replicated), local SpMM -> partial (N, D), all-reduce(sum).  BASELINE's named form.
    ``spmm`` / ``linear`` default to the HIP library; the gloo tests pass torch stand-ins.
    """

    def __init__(self, model, graph, mode="dst", group=None, local_ops=None):
        if mode not in ("dst", "edge"):
            raise ValueError("mode must be 'dst' or 'edge'")
        self.model, self.mode, self.group = model, mode, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ops = local_ops
        self.N = graph.num_items + graph.num_users
        self.I = graph.num_items
        self._build(graph)

    # -- graph partition (torch index plumbing; coefficients come from the HIP kernels or the test stand-in)
    def _coef(self, graph):
        u2i, i2u = graph.user2item_edge_index, graph.item2user_edge_index
        N = self.N
        if self.ops is None:
            deg = torch.zeros(N, dtype=torch.float32, device=u2i.device)
            native.degree_accumulate(u2i[1].contiguous(), N, deg)
            native.degree_accumulate(i2u[1].contiguous(), N, deg)
            c1 = native.edge_coef(u2i[0].contiguous(), u2i[1].contiguous(), graph.user2item_edge_attr, deg)
            c2 = native.edge_coef(i2u[0].contiguous(), i2u[1].contiguous(), graph.item2user_edge_attr, deg)
            return c1, c2
        return self.ops.coef(graph, N)

    def _build(self, graph):
        hetero = self.model.hetero
        u2i, i2u = graph.user2item_edge_index, graph.item2user_edge_index
        N, W, r = self.N, self.world, self.rank
        c1, c2 = self._coef(graph)
        self.stacked = False
        src2 = i2u[0]
        if hetero:
            bip = (u2i.shape[1] == 0 or int(u2i[0].min()) >= self.I) and (i2u.shape[1] == 0 or int(i2u[0].max()) < self.I)
            if not bip:
                self.stacked = True
                src2 = i2u[0] + N
        src = torch.cat([u2i[0], src2])
        dst = torch.cat([u2i[1], i2u[1]])
        coef = torch.cat([c1, c2])
        order = torch.argsort(dst, stable=True)
        src, dst, coef = src[order], dst[order], coef[order]
        counts = torch.bincount(dst, minlength=N)
        rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dst.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        E = int(rowptr[-1])
        if self.mode == "dst":
            targets = torch.arange(1, W, device=dst.device, dtype=torch.int64) * E // W
            inner = torch.searchsorted(rowptr, targets)  # first row whose prefix reaches the target
            bounds = torch.cat([torch.zeros(1, dtype=torch.int64, device=dst.device), inner.clamp(max=N),
                                torch.full((1,), N, dtype=torch.int64, device=dst.device)])
            bounds = torch.cummax(bounds, 0).values
            self.bounds = bounds.tolist()
            lo, hi = self.bounds[r], self.bounds[r + 1]
            e0, e1 = int(rowptr[lo]), int(rowptr[hi])
            self.lo, self.hi = lo, hi
            self.col = src[e0:e1].to(torch.int32).contiguous()
            self.coef = coef[e0:e1].contiguous()
            self.rowptr = (rowptr[lo:hi + 1] - e0).contiguous()
            self.max_block = max(self.bounds[i + 1] - self.bounds[i] for i in range(W))
        else:
            keep = (torch.arange(E, device=dst.device) % W) == r
            self.lo, self.hi = 0, N
            self.col = src[keep].to(torch.int32).contiguous()
            self.coef = coef[keep].contiguous()
            cnt = torch.bincount(dst[keep], minlength=N)
            self.rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dst.device)
            self.rowptr[1:] = torch.cumsum(cnt, 0)
        self.csr = native.SegmentedCSR(self.rowptr, self.col, self.coef) if self.ops is None else None

    # -- local compute
    def _linear(self, x, lin):
        if self.ops is None:
            return native.linear(x.contiguous(), lin.weight.detach(), lin.bias.detach())
        return self.ops.linear(x, lin.weight.detach(), lin.bias.detach())

    def _spmm(self, z, n_rows, acc=None):
        """y = SpMM(z) over this rank's edges; acc += y fused into the kernel epilogue when given (layer-mean sum)."""
        if self.ops is None:
            return self.csr.spmm(z, acc_sum=acc)
        y = self.ops.spmm(self.rowptr, self.col, self.coef, z, n_rows)
        if acc is not None:
            acc += y
        return y

    def _mean(self, acc, n_layers):
        if self.ops is None:
            return native.scale_rows(acc, float(n_layers + 1))
        return acc / float(n_layers + 1)

    def _hoist_rows(self, x_rows, lo, hi):
        """Z rows for global rows [lo, hi) given x rows of the same range (per-node Linear, gnn_ncf.py:91-93 hoisted)."""
        conv = self.model.gnn_convs[0]
        D = x_rows.shape[1]
        if not conv.hetero:
            return self._linear(x_rows, conv.W[0])
        if self.stacked:
            raise NotImplementedError("non-bipartite hetero graphs are single-GPU only")
        z = torch.empty((hi - lo, D), dtype=torch.float32, device=x_rows.device)
        split = min(max(self.I - lo, 0), hi - lo)  # rows below I are items -> item2user_W; the rest users -> user2item_W
        if split > 0:
            z[:split] = self._linear(x_rows[:split], conv.item2user_W[0])
        if split < hi - lo:
            z[split:] = self._linear(x_rows[split:], conv.user2item_W[0])
        return z

    def _all_gather_blocks(self, block):
        """Blocks have different row counts (edge-balanced): pad to the largest, all-gather, stitch."""
        if self.world == 1:
            return block
        D = block.shape[1]
        padded = torch.zeros((self.max_block, D), dtype=block.dtype, device=block.device)
        padded[: block.shape[0]] = block
        out = torch.empty((self.world * self.max_block, D), dtype=block.dtype, device=block.device)
        dist.all_gather_into_tensor(out, padded, group=self.group)
        return torch.cat([out[i * self.max_block: i * self.max_block + (self.bounds[i + 1] - self.bounds[i])]
                          for i in range(self.world)], dim=0)

    def propagate(self, x0: torch.Tensor) -> torch.Tensor:
        """x0: full (N, D) initial node table (replicated).  Returns the full mean-combined table (gnn_ncf.py:351)."""
        L = len(self.model.gnn_convs)
        if self.mode == "dst":
            lo, hi = self.lo, self.hi
            x_blk = x0[lo:hi]
            acc = x_blk.clone()
            for _ in range(L):
                z_full = self._all_gather_blocks(self._hoist_rows(x_blk, lo, hi))
                x_blk = self._spmm(z_full, hi - lo, acc)
            return self._all_gather_blocks(self._mean(acc, L))
        x = x0
        acc = x0.clone()
        for _ in range(L):
            z = self._hoist_rows(x, 0, self.N)
            y = self._spmm(z, self.N)
            if self.world > 1:
                dist.all_reduce(y, op=dist.ReduceOp.SUM, group=self.group)  # RCCL ring: 2(W-1)/W x N x D x 4 bytes per rank
            x = y
            acc += y
        return self._mean(acc, L)
