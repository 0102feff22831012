"""Multi-GPU forms of the scoring path, one process per GPU over torch.distributed (RCCL on ROCm).

* ``RowShardedTable`` / ``ShardedBasicNCF`` — BASELINE config 5: embedding tables sharded row-wise over the ranks
  (contiguous row ranges), batch data-parallel.  Two exchange forms:
    - ``bounded`` (default): owner bucketing on the device into a FIXED-capacity buffer, so that the id all-to-all and the
      row all-to-all use equal splits and no size ever travels to the host: zero host synchronisations per step.  Round 3:
      the bucketing DE-DUPLICATES on the device (ncf_bucket_ids_dedup: a hash set in device memory, no sort — every
      distinct id of the batch travels once, all its pairs share the row that comes back) and every bucket carries its
      count, so the owner gathers only the rows that were asked for (ncf_gather_buckets) and the statistics know how
      many of the shipped slots were padding.  The capacity is agreed once (largest bucket of DISTINCT ids of a sample
      batch over all ranks + a margin of a few standard deviations); a bucket that outgrows it sets a sticky device flag
      that ``check()`` — a COLLECTIVE: every rank raises the same exception — turns into ``ExchangeOverflow``; every rank
      then calls ``grow_capacity()`` and repeats the pass.
      ``submit()`` runs the exchange of step t+1 on a second stream under step t's MLP; ``score()`` waits on its event.
    - ``unique``: torch.unique of the ids (sorted -> contiguous owner buckets, duplicates travel once), exact split
      sizes (one host read per lookup).  For traffic with many repeated ids per batch (Zipf), where de-duplication
      saves more xGMI bytes than the synchronisation costs.
  Levers from SURVEY §7: optional replication of the small table; rows stay bf16 on the wire.  xGMI is point-to-point:
  an all-to-all uses all 7 links of a GPU at once.
* ``PartitionedLightGCN`` — BASELINE config 4: (a) edge-partitioned + all-reduce of the (N, D) partial sums, the form
  BASELINE names; (b) destination-partitioned (edge-balanced contiguous row blocks) + direct exchange of the blocks
  (every rank sends its block to every peer: an all-gather over all links, no padding, no re-stitch), which moves
  1/W of the bytes per rank and reproduces the single-GPU result bit for bit.

Replica scaling of BasicNCF / MF / AttentionNCF needs no code here: every rank holds the whole model and scores its
slice of the batch (bench.py --gpus N).

The local compute is always the HIP library.  ``local_ops`` exists only so the exchange logic can be exercised by the
world-size-2 gloo tests on a CPU-only host; the default (None) is the HIP path and nothing falls back to it silently.
With the gloo backend, GPU tensors travel through host memory (rehearsals of the multi-process flow on a one-GPU box).
"""
from typing import Optional, Sequence

import torch
import torch.distributed as dist

from . import native


class ExchangeOverflow(RuntimeError):
    """Some rank's bucket for one owner outgrew the agreed capacity: the affected pairs were scored as out-of-range rows.
    Raised by ``check()`` on EVERY rank of the group (the flags are all-reduced), so that every rank takes the same
    recovery: ``grow_capacity()`` (or ``negotiate_capacity`` / ``set_capacity`` with the same value everywhere) and repeat the pass."""


class _HipOps:
    """Local compute used by the sharded paths — thin names over native.*"""

    @staticmethod
    def gather_rows(table, idx, out=None):
        return native.gather_concat(table, idx, out=out)

    @staticmethod
    def bucket_ids(idx, rows_per_rank, total_rows, world, cap, send, slot, counts, overflow):
        return native.bucket_ids(idx, rows_per_rank, total_rows, world, cap, send, slot, counts, overflow)

    @staticmethod
    def bucket_ids_dedup(idx, rows_per_rank, total_rows, world, cap, send, slot, counts, overflow, hkeys, hvals):
        return native.bucket_ids_dedup(idx, rows_per_rank, total_rows, world, cap, send, slot, counts, overflow, hkeys, hvals)

    @staticmethod
    def gather_buckets(table, recv, world, cap, out):
        return native.gather_buckets(table, recv, world, cap, out)

    @staticmethod
    def device_flags(device):
        """The library's sticky out-of-range flag on this device, as a (1,) int32 tensor (all-reduced by check())."""
        return native._oob_flag(device)

    @staticmethod
    def score(tabA, idxA, tabB, idxB, packed, weights, biases):
        if packed is not None and packed.supports(tabA.shape[1], tabB.shape[1]) and packed.dtype == tabA.dtype:
            return native.score_fused(tabA, idxA, tabB, idxB, packed)
        x = native.gather_concat(tabA, idxA, tabB, idxB)
        return native.mlp_forward(x.float() if x.dtype != torch.float32 else x, weights, biases)


class Comm:
    """The collectives of this module on one process group.  nccl (= RCCL): device tensors straight through; gloo:
    device tensors are staged through host memory (a rehearsal transport — tests and one-GPU boxes)."""

    def __init__(self, group=None):
        self.group = group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0
        self.backend = dist.get_backend(group) if self.on else None

    def _staged(self, t):
        return self.backend != "nccl" and t.is_cuda

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if self._staged(out):
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(o)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def all_reduce(self, t, op=None):
        op = dist.ReduceOp.SUM if op is None else op
        if self._staged(t):
            h = t.cpu()
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=op, group=self.group)
        return t

    def exchange_blocks(self, full: torch.Tensor, bounds: Sequence[int]):
        """Rows [bounds[r], bounds[r+1]) of ``full`` are valid on rank r; fills every other block from its owner.
        One send per peer and one receive per peer, straight into place: a direct all-gather of UNEVEN blocks (no
        padding to the largest block, no re-stitch), all peers at once — on xGMI every link of the GPU carries one."""
        if self.world == 1:
            return full
        staged = self._staged(full)
        buf = full.cpu() if staged else full
        mine = buf[bounds[self.rank]:bounds[self.rank + 1]]
        ops = []
        for d in range(1, self.world):  # staggered peers: rank r talks to r+d / r-d in step d
            to, frm = (self.rank + d) % self.world, (self.rank - d) % self.world
            if mine.numel():
                ops.append(dist.P2POp(dist.isend, mine, to, self.group))
            blk = buf[bounds[frm]:bounds[frm + 1]]
            if blk.numel():
                ops.append(dist.P2POp(dist.irecv, blk, frm, self.group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if staged:
            full.copy_(buf)
        return full


class _ExchangeBuffers:
    """Device buffers of one in-flight bounded lookup (allocated once, reused every ``depth`` steps)."""

    def __init__(self, world, cap, E, dtype, device, batch, dedup=True, table_slots=0):
        n = world * cap
        self.cap = cap
        self.dedup = dedup
        hdr = 1 if dedup else 0                     # de-duplicating buckets carry their count in front: [count, ids...]
        self.send = torch.empty(world * (cap + hdr), dtype=torch.int64, device=device)
        self.recv_ids = torch.empty(world * (cap + hdr), dtype=torch.int64, device=device)
        self.rows_out = torch.empty((n, E), dtype=dtype, device=device)
        self.rows_in = torch.empty((n, E), dtype=dtype, device=device)
        self.slot = torch.empty(max(batch, 1), dtype=torch.int64, device=device)
        self.counts = torch.empty(world, dtype=torch.int32, device=device)
        if dedup:                                   # hash set of the batch's ids (keys / bucket slots), caller-owned scratch
            self.hkeys = torch.empty(max(table_slots, 1), dtype=torch.int64, device=device)
            self.hvals = torch.empty(max(table_slots, 1), dtype=torch.int64, device=device)


class RowShardedTable:
    """Rows ``[rank*rpr, min((rank+1)*rpr, total))`` of a (total_rows, E) table live on this rank."""

    CAP_SIGMAS = 8.0      # capacity = largest bucket seen at negotiation + CAP_SIGMAS * sqrt(that) + CAP_ROUND, rounded up to CAP_ROUND:
    CAP_ROUND = 64        # a bucket of n ids fluctuates by ~sqrt(n) from batch to batch (8192 ids: +9 % instead of round 2's +25 %)
    GROW = 1.5            # grow_capacity() factor

    def __init__(self, local_rows: torch.Tensor, total_rows: int, group=None, local_ops=None, comm=None, dedup=True):
        self.comm = comm if comm is not None else Comm(group)   # `comm`: an object with Comm's interface (tests: a loopback)
        self.group = group
        self.world, self.rank = self.comm.world, self.comm.rank
        self.total_rows = int(total_rows)
        self.rows_per_rank = max((self.total_rows + self.world - 1) // self.world, 1)
        lo = min(self.rank * self.rows_per_rank, self.total_rows)
        hi = min(lo + self.rows_per_rank, self.total_rows)
        if local_rows.shape[0] != max(hi - lo, 0):
            raise ValueError(f"rank {self.rank} must hold rows [{lo}, {hi}) = {hi - lo} rows, got {local_rows.shape[0]}")
        self.local = local_rows.contiguous()
        self.ops = local_ops or _HipOps
        self.last_stats = {}
        self.last_oob = False
        self.cap = None
        self.dedup = bool(dedup)
        self.overflow = torch.zeros(1, dtype=torch.int32, device=self.local.device)
        # statistics of the bounded exchange since the last check(), kept on the device (read by check()): ids asked for,
        # distinct ids listed (= rows that had to travel or be gathered locally), of which for OTHER ranks, lookups
        # statistics of the bounded lookups: ids listed per owner accumulate on the device (ONE small add per lookup, no read);
        # what the host knows anyway (ids asked, lookups) is counted on the host
        self._listed = torch.zeros(max(self.world, 1), dtype=torch.int64, device=self.local.device)
        self._asked = 0
        self._lookups = 0

    @staticmethod
    def shard_bounds(total_rows: int, world: int, rank: int):
        rpr = max((total_rows + world - 1) // world, 1)
        return min(rank * rpr, total_rows), min((rank + 1) * rpr, total_rows)

    # ------------------------------------------------------------------------------------------ unique (exact splits)
    def lookup_unique(self, idx: torch.Tensor, raise_oob: bool = True):
        """Returns (rows_of_unique_ids (n_unique, E), inverse (B,) int64) with rows[inverse[p]] == table[idx[p]].
        An id outside [0, total_rows) raises IndexError AFTER the exchange, so the ranks' collectives stay matched
        (``raise_oob=False``: only ``self.last_oob`` is set — a caller with further collectives to run raises later)."""
        self.last_oob = False
        if self.world == 1:
            return self.local, idx.contiguous()  # nothing to exchange: the scoring kernel gathers from the table itself
        dev = idx.device
        bad = ((idx < 0) | (idx >= self.total_rows)).any().view(1).to(torch.int64)
        uniq, inverse = torch.unique(idx.clamp(0, max(self.total_rows - 1, 0)), return_inverse=True)  # sorted -> contiguous owner buckets
        owner = torch.div(uniq, self.rows_per_rank, rounding_mode="floor")
        send_counts = torch.bincount(owner, minlength=self.world)
        recv_counts = torch.empty_like(send_counts)
        self.comm.all_to_all(recv_counts, send_counts)
        sizes = torch.cat([send_counts, recv_counts, bad]).tolist()  # split sizes must be host ints (one sync per lookup)
        sc, rc, is_bad = sizes[:self.world], sizes[self.world:2 * self.world], sizes[-1]
        local_ids = (uniq - owner * self.rows_per_rank).contiguous()
        recv_ids = torch.empty(sum(rc), dtype=torch.int64, device=dev)
        self.comm.all_to_all(recv_ids, local_ids, rc, sc)                  # exchange #1: row ids to their owners
        E = self.local.shape[1]
        if recv_ids.numel():
            rows_out = self.ops.gather_rows(self.local, recv_ids)          # owners gather (HIP K1)
        else:
            rows_out = torch.empty((0, E), dtype=self.local.dtype, device=dev)
        rows_in = torch.empty((uniq.numel(), E), dtype=self.local.dtype, device=dev)
        self.comm.all_to_all(rows_in, rows_out, sc, rc)                    # exchange #2: rows back, same bucket order
        self.last_stats = {"requested": int(idx.numel()), "unique": int(uniq.numel()),
                           "remote_rows": int(uniq.numel() - sc[self.rank]),
                           "bytes_received": int((uniq.numel() - sc[self.rank]) * E * self.local.element_size())}
        self.last_oob = bool(is_bad)
        if is_bad and raise_oob:
            raise IndexError(f"index out of range for a sharded table of {self.total_rows} rows")
        return rows_in, inverse.contiguous()

    def lookup(self, idx: torch.Tensor) -> torch.Tensor:
        rows, inverse = self.lookup_unique(idx)
        return self.ops.gather_rows(rows, inverse)

    # ------------------------------------------------------------------------------------------ bounded (no host sync)
    def set_capacity(self, cap: int):
        self.cap = int(cap)

    def negotiate_capacity(self, idx: torch.Tensor, slack: Optional[float] = None) -> int:
        """Agree ONCE on the per-owner capacity of the exchange buffers: the largest bucket of this sample batch over all
        ranks (distinct ids when the exchange de-duplicates; one all-reduce MAX, one host read) plus a margin.  Every rank must
        call it with its own batch.  ``slack``: a factor on the largest bucket instead of the sigma margin."""
        ids = idx.clamp(0, max(self.total_rows - 1, 0))
        if self.dedup:
            ids = torch.unique(ids)
        owner = torch.div(ids, self.rows_per_rank, rounding_mode="floor")
        mx = torch.bincount(owner, minlength=self.world).max().view(1)
        if self.world > 1:
            self.comm.all_reduce(mx, dist.ReduceOp.MAX)
        m = int(mx.item())
        need = int(m * slack) + 1 if slack is not None else int(m + self.CAP_SIGMAS * m ** 0.5) + self.CAP_ROUND
        self.cap = (need + self.CAP_ROUND - 1) // self.CAP_ROUND * self.CAP_ROUND
        return self.cap

    def grow_capacity(self, factor: Optional[float] = None) -> int:
        """After ExchangeOverflow: every rank calls this (check() raised on all of them) and gets the same larger capacity."""
        if self.cap is None:
            raise RuntimeError("capacity not set")
        need = int(self.cap * (self.GROW if factor is None else factor)) + 1
        self.cap = (need + self.CAP_ROUND - 1) // self.CAP_ROUND * self.CAP_ROUND
        return self.cap

    def new_buffers(self, batch: int) -> _ExchangeBuffers:
        if self.cap is None:
            raise RuntimeError("capacity not set: call negotiate_capacity(sample_ids) or set_capacity(cap) first")
        slots = 0
        if self.dedup:
            slots = 1024
            while slots < 2 * max(batch, 1):
                slots <<= 1
        return _ExchangeBuffers(self.world, self.cap, self.local.shape[1], self.local.dtype, self.local.device, batch,
                                dedup=self.dedup, table_slots=slots)

    def lookup_bounded(self, idx: torch.Tensor, buf: _ExchangeBuffers):
        """(rows, slot) with rows[slot[p]] == table[idx[p]], through ``buf``; everything is enqueued on the current
        stream and nothing is read back: bucket kernel -> all-to-all of the id buckets (equal splits) -> the owners'
        gather -> all-to-all of the rows back (equal splits).  Dropped pairs (out of range / over capacity) have
        slot -1 and read as out-of-range rows downstream; see check()."""
        if self.world == 1:
            return self.local, idx.contiguous()
        if buf.cap != self.cap or idx.numel() > buf.slot.numel() or buf.dedup != self.dedup:
            raise ValueError("exchange buffers were built for another capacity / a smaller batch / the other bucketing")
        W, cap = self.world, self.cap
        if self.dedup:
            self.ops.bucket_ids_dedup(idx.contiguous(), self.rows_per_rank, self.total_rows, W, cap, buf.send, buf.slot,
                                      buf.counts, self.overflow, buf.hkeys, buf.hvals)
            self.comm.all_to_all(buf.recv_ids, buf.send)    # exchange #1: W buckets of [count, ids...], equal splits of cap + 1
            if self.local.shape[0]:
                self.ops.gather_buckets(self.local, buf.recv_ids, W, cap, buf.rows_out)   # count rows per bucket; padding untouched
        else:
            self.ops.bucket_ids(idx.contiguous(), self.rows_per_rank, self.total_rows, W, cap, buf.send, buf.slot,
                                buf.counts, self.overflow)
            self.comm.all_to_all(buf.recv_ids, buf.send)    # exchange #1: W x cap local row ids, equal splits
            if self.local.shape[0]:
                self.ops.gather_rows(self.local, buf.recv_ids, out=buf.rows_out)   # owners gather (HIP K1); padding = row 0
            else:
                buf.rows_out.zero_()
        self.comm.all_to_all(buf.rows_in, buf.rows_out)     # exchange #2: W x cap rows back, equal splits
        self._listed.add_(buf.counts[:W].clamp(max=cap))   # per owner, on the device (no read); eight tiny launches did this before
        self._asked += idx.numel()
        self._lookups += 1
        return buf.rows_in, buf.slot[:idx.numel()]

    def _flags(self):
        """(overflow, out-of-range) of this rank as a (2,) int32 tensor on the table's device (no read)."""
        oob = self.ops.device_flags(self.local.device) if hasattr(self.ops, "device_flags") else None
        if oob is None:
            oob = torch.zeros(1, dtype=torch.int32, device=self.local.device)
        return torch.cat([self.overflow.view(1), oob.view(1).to(self.overflow.dtype)])

    def _clear_flags(self):
        self.overflow.zero_()
        oob = self.ops.device_flags(self.local.device) if hasattr(self.ops, "device_flags") else None
        if oob is not None:
            oob.zero_()

    def wire_stats(self, reset: bool = True) -> dict:
        """Traffic of the bounded lookups since the last call (one host read): what travelled and how much of it was padding."""
        per_owner = self._listed.tolist()
        asked, lookups, listed = self._asked, self._lookups, int(sum(per_owner))
        remote = listed - int(per_owner[self.rank]) if self.world > 1 else 0
        if reset:
            self._listed.zero_()
            self._asked = self._lookups = 0
        E, elt = self.local.shape[1], self.local.element_size()
        cap = self.cap or 0
        shipped_rows = lookups * (self.world - 1) * cap
        hdr = 1 if self.dedup else 0
        return {"lookups": lookups, "ids_asked": asked, "ids_listed": listed, "remote_rows_needed": remote,
                "rows_shipped_incl_padding": shipped_rows,
                "row_bytes_on_wire": shipped_rows * E * elt, "row_bytes_needed": remote * E * elt,
                "id_bytes_on_wire": lookups * (self.world - 1) * (cap + hdr) * 8,
                "padding_fraction": (1.0 - remote / shipped_rows) if shipped_rows else 0.0,
                "duplicates_removed_fraction": (1.0 - listed / asked) if asked else 0.0}

    def check(self):
        """End of a pass: synchronising, COLLECTIVE check of the sticky flags — every rank of the group must call it and every
        rank raises the same exception (ExchangeOverflow before IndexError); all flags are cleared first, so a repeated pass
        starts clean."""
        _raise_flags(self.comm, [self])


def _raise_flags(comm, tables):
    flags = torch.cat([t._flags() for t in tables])            # per table: overflow, out-of-range (the library's flag is per device)
    if comm.world > 1:
        comm.all_reduce(flags, dist.ReduceOp.MAX)
    vals = flags.tolist()
    for t in tables:
        t._clear_flags()
    if any(vals[0::2]):
        raise ExchangeOverflow("a bucket outgrew the exchange capacity on some rank (capacities: "
                               + ", ".join(str(t.cap) for t in tables) + " ids per owner): grow_capacity() on every rank and repeat the pass")
    if any(vals[1::2]):
        raise IndexError("index out of range for a sharded embedding table (on some rank)")


class _Ticket:
    __slots__ = ("urows", "uslot", "irows", "islot", "ready", "bufs", "keep")


class ShardedBasicNCF:
    """BasicNCF scoring with row-sharded embedding tables (tables built from a BasicNCF's weights or given directly).

    ``user_table`` / ``item_table`` are this rank's shards of ``T = W^T + b``; ``item_table`` may instead be the full
    table with ``replicate_items=True`` (the 2.56 GB item table of config 5 fits every GPU; only users are exchanged).
    ``exchange``: "bounded" (device bucketing, equal-split all-to-alls, no host sync; default) or "unique" (sorted
    de-duplicated ids, exact splits, one host read per lookup).

    Pipelined use (step t+1's exchange under step t's MLP):
        t = m.submit(u0, i0)
        for k in range(n):
            nxt = m.submit(u[k+1], i[k+1]) if k + 1 < n else None     # enqueued on the exchange stream
            out = m.score(t)                                          # current stream; waits on t's event only
            t = nxt
    ``forward(u, i)`` = ``score(submit(u, i))``.
    """

    def __init__(self, user_table, num_users, item_table, num_items, mlp_weights: Sequence[torch.Tensor],
                 mlp_biases: Sequence[Optional[torch.Tensor]], replicate_items=False, group=None, local_ops=None,
                 dtype=None, exchange="bounded", depth=2, comm=None, dedup=True):
        if exchange not in ("bounded", "unique"):
            raise ValueError("exchange must be 'bounded' or 'unique'")
        self.ops = local_ops or _HipOps
        self.exchange = exchange
        self.users = RowShardedTable(user_table, num_users, group, local_ops, comm, dedup=dedup)
        self.replicate_items = replicate_items
        self.items_full = item_table.contiguous() if replicate_items else None
        self.items = None if replicate_items else RowShardedTable(item_table, num_items, group, local_ops, comm, dedup=dedup)
        self.weights = [w.detach().float().contiguous() for w in mlp_weights]
        self.biases = [None if b is None else b.detach().float().contiguous() for b in mlp_biases]
        self.packed = None
        if local_ops is None:
            try:
                self.packed = native.PackedMLP(self.weights, self.biases, dtype=dtype or user_table.dtype)
            except native.NativeError as e:
                if e.code != native.NCF_EUNSUPPORTED:
                    raise
        self.depth = int(depth)
        self._ring, self._consumed, self._n = [None] * self.depth, [None] * self.depth, 0
        self._xstream = torch.cuda.Stream(device=user_table.device) if (user_table.is_cuda and self.users.world > 1) else None

    # -- capacity
    def negotiate_capacity(self, user_pos, item_pos=None):
        caps = {"users": self.users.negotiate_capacity(user_pos)}
        if self.items is not None:
            caps["items"] = self.items.negotiate_capacity(item_pos)
        self._ring = [None] * self.depth    # buffers are sized by the capacity
        return caps

    def _buffers(self, B):
        k = self._n % self.depth
        have = self._ring[k]
        if (have is None or have[0].slot.numel() < B or have[0].cap != self.users.cap
                or (self.items is not None and have[1].cap != self.items.cap)):
            if have is not None and self._xstream is not None:   # rare (capacity / batch growth): nothing may still use the old set
                self._xstream.synchronize()
                torch.cuda.current_stream(self.users.local.device).synchronize()
            self._ring[k] = (self.users.new_buffers(B), None if self.items is None else self.items.new_buffers(B))
            if self._xstream is not None and self._consumed[k] is None:
                self._consumed[k] = torch.cuda.Event()
                self._consumed[k].record(torch.cuda.current_stream(self.users.local.device))
        return k

    # -- the two halves of a step
    def submit(self, user_pos: torch.Tensor, item_pos: torch.Tensor) -> _Ticket:
        t = _Ticket()
        t.ready, t.bufs, t.keep = None, None, (user_pos, item_pos)
        if self.users.world == 1 or self.exchange == "unique":
            t.urows, t.uslot = self.users.lookup_unique(user_pos, raise_oob=False)
            if self.replicate_items:
                t.irows, t.islot = self.items_full, item_pos.contiguous()
            else:
                t.irows, t.islot = self.items.lookup_unique(item_pos, raise_oob=False)
            if self.users.last_oob or (self.items is not None and self.items.last_oob):   # after BOTH tables' collectives
                raise IndexError("index out of range for a sharded embedding table")
            return t
        if self.users.cap is None:
            self.negotiate_capacity(user_pos, item_pos)   # first batch: agree on the capacity (the ONE host read)
        k = self._buffers(user_pos.numel())
        self._n += 1
        ubuf, ibuf = self._ring[k]
        t.bufs = k
        xs = self._xstream
        if xs is None:   # CPU tensors (gloo tests): same logic, no streams
            t.urows, t.uslot = self.users.lookup_bounded(user_pos, ubuf)
            if self.replicate_items:
                t.irows, t.islot = self.items_full, item_pos.contiguous()
            else:
                t.irows, t.islot = self.items.lookup_bounded(item_pos, ibuf)
            return t
        cur = torch.cuda.current_stream(user_pos.device)
        xs.wait_stream(cur)                       # the ids are produced on the caller's stream
        xs.wait_event(self._consumed[k])          # the step that last read this slot's rows has been scored
        user_pos.record_stream(xs)
        item_pos.record_stream(xs)
        with torch.cuda.stream(xs):
            t.urows, t.uslot = self.users.lookup_bounded(user_pos, ubuf)
            if self.replicate_items:
                t.irows, t.islot = self.items_full, item_pos.contiguous()
            else:
                t.irows, t.islot = self.items.lookup_bounded(item_pos, ibuf)
            t.ready = torch.cuda.Event()
            t.ready.record(xs)
        return t

    def score(self, t: _Ticket) -> torch.Tensor:
        if t.ready is not None:
            torch.cuda.current_stream(t.urows.device).wait_event(t.ready)
        out = self.ops.score(t.urows, t.uslot, t.irows, t.islot, self.packed, self.weights, self.biases)
        if t.ready is not None:
            self._consumed[t.bufs].record(torch.cuda.current_stream(t.urows.device))
        return out

    def forward(self, user_pos: torch.Tensor, item_pos: torch.Tensor) -> torch.Tensor:
        """Global positions of this rank's local batch -> (B, 1) scores (cat(user, item) -> MLP, basic_ncf.py:40-41)."""
        return self.score(self.submit(user_pos, item_pos))

    __call__ = forward

    def check(self):
        """End of a pass (COLLECTIVE: every rank calls it, every rank raises the same exception): ExchangeOverflow / IndexError if
        any step on any rank since the last check dropped a pair.  Both tables' flags and the device's out-of-range flag are read
        in ONE all-reduce and cleared before raising, so the repeated pass starts clean."""
        _raise_flags(self.users.comm, [self.users] + ([self.items] if self.items is not None else []))

    def grow_capacity(self, factor=None):
        """After ExchangeOverflow, on every rank: larger buffers for both tables (the same value everywhere)."""
        caps = {"users": self.users.grow_capacity(factor)}
        if self.items is not None:
            caps["items"] = self.items.grow_capacity(factor)
        return caps

    def wire_stats(self, reset=True):
        out = {"users": self.users.wire_stats(reset)}
        if self.items is not None:
            out["items"] = self.items.wire_stats(reset)
        return out


# ---------------------------------------------------------------------------------------------------------------
class PartitionedLightGCN:
    """Multi-GPU LightGCN propagation for GraphNCF scoring (hetero or not, mean readout).

    mode "dst"  : rank r owns the contiguous destination rows [bounds[r], bounds[r+1]) chosen so that every rank
                  has ~E/W edges (prefix sums of the in-degrees); per layer: hoisted Linear on the local block written
                  in place into the full Z, direct exchange of the blocks (Comm.exchange_blocks), local SpMM.  Per-row
                  sums are identical to the single-GPU kernel's -> bitwise equal results.
    mode "edge" : rank r owns edges r::W of every destination row and the full node table; per layer: hoisted
                  Linear (replicated), local SpMM -> partial (N, D), all-reduce(sum).  BASELINE's named form.
    LightGAT (gnn_ncf.py:97-177; mode "dst" only: the per-destination softmax needs all in-edges of a row on one rank): the
    source scores s[n] = w_j . x[n] travel WITH the Z blocks — the exchanged buffer is (N, D + 4), Z in columns [0, D) and s in
    column D — and every rank turns its rows' scores into edge coefficients (ncf_edge_softmax_csr) before its SpMM.
    ``spmm`` / ``linear`` default to the HIP library; the gloo tests pass torch stand-ins.
    """

    def __init__(self, model, graph, mode="dst", group=None, local_ops=None):
        if mode not in ("dst", "edge"):
            raise ValueError("mode must be 'dst' or 'edge'")
        self.gat = getattr(model, "convType", "LightGCN") == "LightGAT"
        if getattr(model, "convType", "LightGCN") not in ("LightGCN", "LightGAT") or getattr(model, "concat", False):
            raise NotImplementedError("PartitionedLightGCN propagates LightGCN / LightGAT layers with the mean readout (concat=False)")
        if self.gat and mode != "dst":
            raise NotImplementedError("LightGAT needs all in-edges of a destination on one rank (its softmax runs over them): mode='dst' only")
        self.model, self.mode, self.group = model, mode, group
        self.comm = Comm(group)
        self.world, self.rank = self.comm.world, self.comm.rank
        self.ops = local_ops
        self.N = graph.num_items + graph.num_users
        self.I = graph.num_items
        self._build(graph)

    # -- graph partition (torch index plumbing; coefficients come from the HIP kernels or the test stand-in)
    def _coef(self, graph, deg):
        u2i, i2u = graph.user2item_edge_index, graph.item2user_edge_index
        if self.ops is None:
            c1 = native.edge_coef(u2i[0].contiguous(), u2i[1].contiguous(), graph.user2item_edge_attr, deg)
            c2 = native.edge_coef(i2u[0].contiguous(), i2u[1].contiguous(), graph.item2user_edge_attr, deg)
            return c1, c2
        return self.ops.coef(graph, self.N)

    def _build(self, graph):
        hetero = self.model.hetero
        u2i, i2u = graph.user2item_edge_index, graph.item2user_edge_index
        N, W, r = self.N, self.world, self.rank
        dst = torch.cat([u2i[1], i2u[1]])
        src_all = torch.cat([u2i[0], i2u[0]])
        if dst.numel() and (int(dst.min()) < 0 or int(dst.max()) >= N or int(src_all.min()) < 0 or int(src_all.max()) >= N):
            raise IndexError(f"edge endpoint out of range for a graph of {N} nodes")
        del src_all
        counts = torch.bincount(dst, minlength=N)
        c1, c2 = self._coef(graph, counts.to(torch.float32))   # degree = in-edge count over both edge lists (gnn_ncf.py:41,48)
        self.stacked = False
        src2 = i2u[0]
        if hetero:
            bip = (u2i.shape[1] == 0 or int(u2i[0].min()) >= self.I) and (i2u.shape[1] == 0 or int(i2u[0].max()) < self.I)
            if not bip:
                self.stacked = True
                src2 = i2u[0] + N
        src = torch.cat([u2i[0], src2])
        coef = torch.cat([c1, c2])
        a1, a2 = graph.user2item_edge_attr, graph.item2user_edge_attr
        attr = None                                   # LightGAT: the raw edge weights (weight * alpha, no degree norm, gnn_ncf.py:173-176)
        if self.gat and a1 is not None and a2 is not None:
            attr = torch.cat([a1, a2]).to(torch.float32)
        if self.gat and hetero and self.stacked:
            raise NotImplementedError("LightGAT on a hetero graph whose destinations mix edge types is single-GPU / torch only")
        order = torch.argsort(dst, stable=True)
        src, dst, coef = src[order], dst[order], coef[order]
        if attr is not None:
            attr = attr[order]
        del order
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
            self.attr = None if attr is None else attr[e0:e1].contiguous()
            self.rowptr = (rowptr[lo:hi + 1] - e0).contiguous()
        else:
            keep = (torch.arange(E, device=dst.device) % W) == r
            self.bounds = None
            self.lo, self.hi = 0, N
            self.col = src[keep].to(torch.int32).contiguous()
            self.coef = coef[keep].contiguous()
            cnt = torch.bincount(dst[keep], minlength=N)
            self.rowptr = torch.zeros(N + 1, dtype=torch.int64, device=dst.device)
            self.rowptr[1:] = torch.cumsum(cnt, 0)
        self.local_edges = int(self.col.numel())
        self.csr = native.SegmentedCSR(self.rowptr, self.col, self.coef) if self.ops is None else None

    # -- local compute
    def _linear(self, x, lin, out=None):
        if self.ops is None:
            return native.linear(x.contiguous(), lin.weight.detach(), lin.bias.detach(), out=out)
        y = self.ops.linear(x, lin.weight.detach(), lin.bias.detach())
        if out is not None:
            out.copy_(y)
            return out
        return y

    def _spmm(self, z, n_rows, acc=None):
        """y = SpMM(z) over this rank's edges; acc += y fused into the kernel epilogue when given (layer-mean sum)."""
        if self.ops is None:
            return self.csr.spmm(z, acc_sum=acc)
        y = self.ops.spmm(self.rowptr, self.col, self.coef, z, n_rows)
        if acc is not None:
            acc += y
        return y

    def _mean(self, acc, n_layers, out=None):
        if self.ops is None:
            return native.scale_rows(acc, float(n_layers + 1), out=out)
        y = acc / float(n_layers + 1)
        if out is not None:
            out.copy_(y)
            return out
        return y

    def _hoist_rows(self, x_rows, lo, hi, out):
        """Z rows for global rows [lo, hi) given x rows of the same range (per-node Linear, gnn_ncf.py:91-93 hoisted),
        written into ``out`` (a row range of the full Z)."""
        conv = self.model.gnn_convs[0]
        if hi <= lo:
            return out
        if not conv.hetero:
            return self._linear(x_rows, conv.W[0], out=out)
        if self.stacked:
            raise NotImplementedError("non-bipartite hetero graphs are single-GPU only")
        split = min(max(self.I - lo, 0), hi - lo)  # rows below I are items -> item2user_W; the rest users -> user2item_W
        if split > 0:
            self._linear(x_rows[:split], conv.item2user_W[0], out=out[:split])
        if split < hi - lo:
            self._linear(x_rows[split:], conv.user2item_W[0], out=out[split:])
        return out

    def _score_rows(self, x_rows, lo, hi, out):
        """LightGAT source scores s[n] = w_j . x[n] for global rows [lo, hi) (the destination half of AttNet and its bias are
        constant inside a destination's softmax and cancel, models/gnn_ncf.py LightGATConv) into ``out`` ((hi - lo, 1), strided)."""
        conv = self.model.gnn_convs[0]
        D = x_rows.shape[1]
        if hi <= lo:
            return out

        def gemv(x, att, o):
            w = att[0].weight.detach()[:, :D].contiguous()
            if self.ops is None:
                return native.linear(x.contiguous(), w, None, out=o)
            o.copy_(self.ops.linear(x, w, None))
            return o
        if not conv.hetero:
            return gemv(x_rows, conv.AttNet, out)
        split = min(max(self.I - lo, 0), hi - lo)          # items are the sources of item->user edges, users of user->item edges
        if split > 0:
            gemv(x_rows[:split], conv.item2user_AttNet, out[:split])
        if split < hi - lo:
            gemv(x_rows[split:], conv.user2item_AttNet, out[split:])
        return out

    def _edge_softmax(self, s):
        if self.ops is None:
            if getattr(self, "_softmax_seg", None) is None:       # the local CSR's segments (hub rows are split: no wave walks millions of edges)
                segptr, row_of, _ = self.csr.levels[0]
                self._softmax_seg = False if row_of is None else (
                    segptr, row_of, torch.searchsorted(row_of.to(torch.int64), torch.arange(self.rowptr.numel(), device=row_of.device)).contiguous())
            return native.edge_softmax_csr(self.rowptr, self.col, self.attr, s, segments=self._softmax_seg or None)
        return self.ops.edge_softmax(self.rowptr, self.col, self.attr, s)

    def _propagate_gat(self, x0):
        L = len(self.model.gnn_convs)
        D = x0.shape[1]
        lo, hi = self.lo, self.hi
        x_blk = x0[lo:hi]
        acc = x_blk.clone()
        zs = torch.zeros((self.N, D + 4), dtype=torch.float32, device=x0.device)   # Z | s | 3 pad columns (rows stay 16-byte multiples)
        for _ in range(L):
            self._hoist_rows(x_blk, lo, hi, zs[lo:hi, :D])
            self._score_rows(x_blk, lo, hi, zs[lo:hi, D:D + 1])
            self.comm.exchange_blocks(zs, self.bounds)         # the scores travel with the Z blocks: one exchange per layer
            coef = self._edge_softmax(zs[:, D].contiguous())
            if self.ops is None:
                x_blk = self.csr.spmm(zs[:, :D], acc_sum=acc, coef=coef)
            else:
                x_blk = self.ops.spmm(self.rowptr, self.col, coef, zs[:, :D], hi - lo)
                acc += x_blk
        out = torch.empty((self.N, D), dtype=torch.float32, device=x0.device)
        self._mean(acc, L, out=out[lo:hi])
        return self.comm.exchange_blocks(out, self.bounds)

    def propagate(self, x0: torch.Tensor) -> torch.Tensor:
        """x0: full (N, D) initial node table (replicated).  Returns the full mean-combined table (gnn_ncf.py:351)."""
        L = len(self.model.gnn_convs)
        D = x0.shape[1]
        if self.gat:
            return self._propagate_gat(x0)
        if self.mode == "dst":
            lo, hi = self.lo, self.hi
            x_blk = x0[lo:hi]
            acc = x_blk.clone()
            z_full = torch.empty((self.N, D), dtype=torch.float32, device=x0.device)
            for _ in range(L):
                self._hoist_rows(x_blk, lo, hi, z_full[lo:hi])
                self.comm.exchange_blocks(z_full, self.bounds)
                x_blk = self._spmm(z_full, hi - lo, acc)
            self._mean(acc, L, out=z_full[lo:hi])          # z_full is free again: it becomes the combined table
            return self.comm.exchange_blocks(z_full, self.bounds)
        x = x0
        acc = x0.clone()
        z = torch.empty((self.N, D), dtype=torch.float32, device=x0.device)
        for _ in range(L):
            self._hoist_rows(x, 0, self.N, z)
            y = self._spmm(z, self.N)
            if self.world > 1:
                self.comm.all_reduce(y)  # RCCL ring: 2(W-1)/W x N x D x 4 bytes per rank
            x = y
            acc += y
        return self._mean(acc, L)
