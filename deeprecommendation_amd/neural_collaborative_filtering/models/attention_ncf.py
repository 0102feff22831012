"""AttentionNCF — drop-in for reference models/attention_ncf.py:64-224 (same kwargs, state_dict keys, forward).

HIP scoring path (eval / no-grad), all arithmetic in libncf_hip.so:
  1. ItemEmbeddings on candidates and rated items                         (attention_ncf.py:150-151)  ncf_mlp_forward
  2. AttentionNet's first Linear is split at the concat boundary of :176,  W0 = [Wc | Wr]:
        pc = cand_emb @ Wc^T + b0   (B, A),   pr = rated_emb @ Wr^T   (I, A)
     so the score of a pair is  b1 + w1 . relu(pc[b] + pr[i])  — no (B·nnz, 2·IE) pair matrix, no per-pair GEMM.
  3. scores -> masked softmax -> x ratings -> weighted sum              (:182-213)                  ncf_attn_forward
     over the CSR of `user_matrix != 0` (:158).  The weighted sum uses the linearity of UserEmbeddings (:216):
        UserEmbeddings(sum_i a_i f_i) = sum_i a_i (f_i @ Wu^T) + bu
     so the kernel aggregates the projected rows P = rated_items @ Wu^T (I, UE) instead of F-wide feature rows.
  4. cat(candidate_emb, user_emb) -> MLP                                 (:219-222)                  ncf_score_fused
Both reformulations change only the fp32 summation order (tests hold them to 1e-5 relative of the reference).
A training step keeps to differentiable torch ops with the reference's train-only message dropout and target mask.
"""
import sys

import torch
import torch.nn.functional as F
from torch import nn

from ... import native
from ..util import build_MLP_layers, require_gpu, use_native
from .base import NCF
from .basic_ncf import _ScoringMixin


class RowsOf:
    """``table[index]`` left unevaluated: a (B, F) candidate batch that is a selection of rows of a resident feature
    table.  When the table is the very tensor passed as ``rated_items`` the model takes the candidates' embeddings from the
    catalogue embeddings it has already computed (``rated_emb[index]``: a 1 MB gather) instead of gathering B × F
    features and running the F -> IE Linear again (34 MB + 1.1 GFLOP per 4096-pair batch at config 3)."""

    def __init__(self, table: torch.Tensor, index: torch.Tensor):
        self.table, self.index = table, index

    @property
    def is_cuda(self):
        return self.table.is_cuda and self.index.is_cuda

    @property
    def device(self):
        return self.table.device

    @property
    def shape(self):
        return (self.index.shape[0], self.table.shape[1])

    def float(self):
        return self if self.table.dtype == torch.float32 else RowsOf(self.table.float(), self.index)

    def to(self, device):
        return self if self.table.device == torch.device(device) and self.index.device == torch.device(device) else RowsOf(self.table.to(device), self.index.to(device))

    def materialise(self) -> torch.Tensor:
        return self.table.index_select(0, self.index)


class SparseRatings:
    """CSR form of the (B, I) ``user_matrix``: row b owns entries [rowptr[b], rowptr[b+1]) with ``col`` = position in
    the rated-item list and ``val`` = the non-zero normalised rating.  Providers can emit this directly instead of
    the dense matrix (which is 2·B·I floats of mostly zeros).

    ``pair_row`` (optional, (B,) int64): the CSR then has one row per DISTINCT user of the batch and pair b uses row
    ``pair_row[b]`` — the dense matrix repeats a user's row for each of their pairs (dynamic_datasets.py:24-40); sharing
    it is what lets the LDS-tiled attention kernel stage a user's rated rows once for all of their pairs."""

    # below this average the per-pair kernel is used.  Measured at config-3 shapes (tools/ab_attn_grouped.py threshold),
    # grouped vs per-pair: 1 pair per set 335 vs 154 us, 2: 216 vs 153, 4: 128 vs 152, 8: 77 vs 151, 16+: 57 vs 147
    GROUPED_MIN_PAIRS_PER_ROW = 4

    def __init__(self, rowptr, col, val, num_items, pair_row=None, pairs_per_row_hint=None):
        self.rowptr, self.col, self.val, self.num_items = rowptr, col, val, int(num_items)
        self.pair_row = pair_row
        # average number of pairs per row IN USE, when the CSR holds more rows than the batch touches (a whole user
        # base kept on the GPU, pair_row = user position): the row count then says nothing about sharing
        self.pairs_per_row_hint = pairs_per_row_hint

    @property
    def pairs_per_row(self) -> float:
        if self.pairs_per_row_hint is not None:
            return float(self.pairs_per_row_hint)
        return self.num_pairs / max(1, int(self.rowptr.numel() - 1))

    @property
    def num_pairs(self) -> int:
        return int(self.pair_row.numel()) if self.pair_row is not None else int(self.rowptr.numel() - 1)

    def expanded(self) -> "SparseRatings":
        """Per-pair CSR (row b = pair b) of a shared-row instance; self if rows are not shared."""
        if self.pair_row is None:
            return self
        pr = self.pair_row.to(torch.int64)
        counts = (self.rowptr[1:] - self.rowptr[:-1])[pr]
        rowptr = torch.zeros(pr.numel() + 1, dtype=torch.int64, device=self.rowptr.device)
        rowptr[1:] = torch.cumsum(counts, 0)
        # entry k of pair b is entry rowptr_shared[pair_row[b]] + k of the shared arrays
        owner = torch.repeat_interleave(torch.arange(pr.numel(), device=pr.device), counts)
        src = self.rowptr[:-1][pr][owner] + (torch.arange(owner.numel(), device=pr.device) - rowptr[:-1][owner])
        ex = SparseRatings(rowptr, self.col[src].contiguous(), self.val[src].contiguous(), self.num_items)
        ex.shared_entry = src   # entry of the shared arrays each expanded entry came from
        return ex

    @staticmethod
    def from_dense(user_matrix: torch.Tensor, share_identical_rows: bool = True) -> "SparseRatings":
        """CSR of a dense (B, I) user matrix.  The reference's datasets repeat a user's row for each of their samples
        (dynamic_datasets.py:24-40) and its web backend passes ONE row repeated for every candidate
        (webapp/backend.py:78-121): with ``share_identical_rows`` rows that are exactly equal share one CSR row
        (``pair_row`` maps pairs to rows), which is what lets the LDS-tiled attention kernel stage a rated set once
        for all of its pairs.  Equal rows are found by a random projection (float64) and then VERIFIED element by
        element; a batch of mostly distinct rows skips the verification and keeps one row per pair.  (This function
        reads sizes back to the host in any case.)"""
        B = user_matrix.shape[0]
        if share_identical_rows and B > 1 and user_matrix.shape[1] > 0:
            gen = torch.Generator(device=user_matrix.device).manual_seed(0x5EED)
            proj = torch.rand(user_matrix.shape[1], dtype=torch.float64, device=user_matrix.device, generator=gen) + 0.5
            key = user_matrix.double() @ proj
            uniq, inv = torch.unique(key, return_inverse=True)
            R = int(uniq.numel())
            if R * SparseRatings.GROUPED_MIN_PAIRS_PER_ROW <= B:
                rep = torch.full((R,), B, dtype=torch.int64, device=user_matrix.device)
                rep.scatter_reduce_(0, inv, torch.arange(B, device=user_matrix.device), reduce="amin")   # first pair of each row
                if bool((user_matrix == user_matrix[rep][inv]).all()):
                    shared = SparseRatings.from_dense(user_matrix[rep], share_identical_rows=False)
                    shared.pair_row = inv.contiguous()
                    return shared
        mask = user_matrix != 0  # attention_ncf.py:158 — an entry that is exactly 0 counts as unrated
        rowptr = torch.zeros(user_matrix.shape[0] + 1, dtype=torch.int64, device=user_matrix.device)
        rowptr[1:] = torch.cumsum(mask.sum(dim=1), 0)
        nz = mask.nonzero()  # row-major: sorted by row, then by column
        return SparseRatings(rowptr, nz[:, 1].to(torch.int32).contiguous(), user_matrix[mask].float().contiguous(),
                             user_matrix.shape[1])

    @staticmethod
    def from_dense_on_stream(user_matrix: torch.Tensor, share_identical_rows: bool = True) -> "SparseRatings":
        """``from_dense`` without a single host read (libncf_hip.so: ncf_dense_csr_rows + a cumulative sum + ncf_dense_csr_fill, all
        on the current stream): rows that are exactly equal share one CSR row — found by a row hash, VERIFIED element by element on
        the device.  The CSR keeps B rows (a row that shares another's is empty) and ``col`` / ``val`` are sized for the worst case
        B * I, of which rowptr[B] entries are valid: sizes the host never learns.  What the host cannot know either is how much
        sharing there is; the reference's callers repeat rows (datasets/dynamic_datasets.py:24-40, webapp/backend.py:78-121), so the
        result carries ``pairs_per_row_hint`` = the grouped kernels' threshold: they are correct for any amount of sharing."""
        rowptr, col, val, pair_row = native.dense_to_csr(user_matrix.contiguous(), share_identical_rows)
        if not share_identical_rows:             # one CSR row per pair (the per-pair kernel's input): no indirection
            return SparseRatings(rowptr, col, val, user_matrix.shape[1])
        r = SparseRatings(rowptr, col, val, user_matrix.shape[1], pair_row=pair_row,
                          pairs_per_row_hint=SparseRatings.GROUPED_MIN_PAIRS_PER_ROW)
        r.nnz_hint = max(1, user_matrix.shape[1] // 4) * user_matrix.shape[0]     # for launch geometry only (slices per rated set)
        return r

    def to_dense(self, values: torch.Tensor) -> torch.Tensor:
        if self.pair_row is not None:
            ex = self.expanded()   # `values` may be aligned with the shared entries (e.g. self.val) or the expanded ones
            return ex.to_dense(values[ex.shared_entry] if values.numel() == self.col.numel() else values)
        B = self.rowptr.numel() - 1
        rows = torch.repeat_interleave(torch.arange(B, device=values.device), self.rowptr[1:] - self.rowptr[:-1])
        out = torch.zeros((B, self.num_items), dtype=values.dtype, device=values.device)
        out[rows, self.col.long()] = values
        return out


class AttentionNCF(_ScoringMixin, NCF):
    compatible_datasets = ("DynamicPointwiseDataset", "DynamicRankingDataset")
    # a dense (B, I) user_matrix is converted to shared-row CSR on the stream (no host read; the grouped kernels are used whatever the
    # sharing turns out to be).  False: round 2's conversion — two host reads, exact choice between the grouped and the per-pair kernel.
    dense_user_matrix_on_stream = True

    def __init__(self, item_dim, item_emb=128, user_emb=128, att_dense=None, mlp_dense_layers=None,
                 use_cos_sim_instead=False, dropout_rate=0.2, message_dropout=None):
        super().__init__()
        if mlp_dense_layers is None:
            mlp_dense_layers = [256, 128]
        self.kwargs = {'item_dim': item_dim, 'item_emb': item_emb, 'user_emb': user_emb, 'att_dense': att_dense,
                       'mlp_dense_layers': mlp_dense_layers, 'dropout_rate': dropout_rate,
                       'use_cos_sim_instead': use_cos_sim_instead, 'message_dropout': message_dropout}
        self.use_cos_sim_instead = use_cos_sim_instead
        self.message_dropout = message_dropout
        self.ItemEmbeddings = nn.Sequential(nn.Linear(item_dim, item_emb))
        self.UserEmbeddings = nn.Sequential(nn.Linear(item_dim, user_emb))
        if not use_cos_sim_instead:
            if att_dense is not None:
                self.att_dense = att_dense
                self.AttentionNet = nn.Sequential(nn.Linear(2 * item_emb, att_dense), nn.ReLU(), nn.Dropout(dropout_rate),
                                                  nn.Linear(att_dense, 1))
            else:
                self.att_dense = 0
                self.AttentionNet = nn.Sequential(nn.Linear(2 * item_emb, 1))
        self.MLP = build_MLP_layers(item_emb + user_emb, mlp_dense_layers, dropout_rate=dropout_rate)

    def get_model_parameters(self) -> dict:
        return self.kwargs

    def important_hypeparams(self) -> str:
        return '_cosine' if self.use_cos_sim_instead else f'_attNet{self.att_dense}'

    # ------------------------------------------------------------------------------------------ HIP scoring
    def _att_split(self, cache=None):
        """Contiguous halves of AttentionNet.0 split at the cat(candidate, rated) boundary (:176).  With a hidden attention
        layer (att_dense) both halves and the bias carry the factor 2^-64 and AttentionNet's output weight 2^64
        (native.ATT_MLP_SCALED: w1 relu(pc + pr) is unchanged bit for bit — powers of two scale exactly — and in scaled
        units relu is the [0, 1] clamp of the kernel's packed add)."""
        cache = self._refresh() if cache is None else cache
        if "att_split" not in cache:
            l0 = self.AttentionNet[0]
            IE = self.ItemEmbeddings[0].out_features
            w = l0.weight.detach()
            f = 2.0 ** -native.ATT_SCALE_LOG2 if self.att_dense else 1.0
            cache["att_split"] = ((w[:, :IE] * f).contiguous(), (w[:, IE:] * f).contiguous(), (l0.bias.detach() * f).contiguous())
        return cache["att_split"]

    def _tail_mlp(self, cache):
        """(W1, b1, W2, b2, w3, b3) for ncf_attn_tail when the MLP has its shape (two hidden layers [256, 128] over item_emb = user_emb
        in {64, 128}: the class default and the shipped checkpoints), else None.  Cached per weight version (b3 is read once)."""
        if "tail_mlp" not in cache:
            from ..util import mlp_linears
            lins = mlp_linears(self.MLP)
            IE, UE = self.ItemEmbeddings[0].out_features, self.UserEmbeddings[0].out_features
            ok = (len(lins) == 3 and lins[2].out_features == 1 and not self.fold_first_layer
                  and native.attn_tail_supported(IE, UE, lins[0].out_features, lins[1].out_features))
            cache["tail_mlp"] = None
            if ok:
                # the two hidden layers' weights packed once per weight version in MFMA operand order (coalesced weight loads in the kernel)
                cache["tail_mlp"] = (native.PackedTailWeight(lins[0].weight.detach().contiguous()), lins[0].bias.detach().contiguous(),
                                     native.PackedTailWeight(lins[1].weight.detach().contiguous()), lins[1].bias.detach().contiguous(),
                                     lins[2].weight.detach().reshape(-1).contiguous(), float(lins[2].bias.detach().item()))
        return cache["tail_mlp"]

    def precompute_catalog(self, rated_items: torch.Tensor, cache=None):
        """Everything that depends only on the rated-item list: its embeddings, the rated half of the attention
        projection and the UserEmbeddings projection of the raw features.  Cached for the last list seen (serving
        scores one user against the whole catalogue, reference webapp/backend.py:78-121)."""
        cache = self._refresh() if cache is None else cache
        key = (rated_items.data_ptr(), tuple(rated_items.shape), rated_items._version)
        hit = cache.get("catalog")
        if hit is not None and hit[0] == key:
            return hit[1]
        x = rated_items.float().contiguous()
        li, lu = self.ItemEmbeddings[0], self.UserEmbeddings[0]
        rated_emb = native.linear(x, li.weight.detach(), li.bias.detach())
        if self.use_cos_sim_instead:
            pr = native.l2_normalize_rows(rated_emb)
        else:
            _, wr, _ = self._att_split(cache)
            pr = native.linear(rated_emb, wr, None)
        proj = native.linear(x, lu.weight.detach(), None)  # P = rated_items @ Wu^T ; bias added once per user row
        val = (rated_emb, pr, proj)
        cache["catalog"] = (key, val, rated_items)
        return val

    def forward(self, candidate_items, rated_items, user_matrix, return_attention_weights=False):
        if not use_native(self):
            if isinstance(candidate_items, RowsOf):
                candidate_items = candidate_items.materialise()
            return self._forward_train(candidate_items, rated_items, user_matrix, return_attention_weights)
        require_gpu(candidate_items, rated_items)
        cache = self._refresh()  # ONE parameter fingerprint per forward (it walks the module tree: ~12 us of host time)
        li, lu = self.ItemEmbeddings[0], self.UserEmbeddings[0]
        rated_emb, pr, proj = self.precompute_catalog(rated_items, cache)
        pc_kept = None
        att_dense = 0 if self.use_cos_sim_instead else int(self.att_dense or 0)
        A_att = li.out_features if self.use_cos_sim_instead else (att_dense or 1)
        mode_att = native.ATT_COS if self.use_cos_sim_instead else (native.ATT_MLP_SCALED if att_dense else native.ATT_LINEAR)
        can_group = native.attn_grouped_supported(mode_att, A_att, lu.out_features)
        if isinstance(user_matrix, SparseRatings):
            ratings = user_matrix
        elif (user_matrix.is_cuda and user_matrix.dtype == torch.float32 and not return_attention_weights
              and 0 < user_matrix.numel() <= native.DENSE_CSR_MAX_ENTRIES and self.dense_user_matrix_on_stream):
            # the reference's call shape, no host round trip; rows are shared only where a grouped kernel can use the sharing
            ratings = SparseRatings.from_dense_on_stream(user_matrix, share_identical_rows=can_group)
        else:
            ratings = SparseRatings.from_dense(user_matrix)
        shared = ratings.pair_row is not None
        grouped = shared and can_group and ratings.pairs_per_row >= SparseRatings.GROUPED_MIN_PAIRS_PER_ROW
        grouping = None
        pc = None
        # never inside a HIP-graph capture: a hit would leave the candidate projections out of the captured graph
        keep = not isinstance(candidate_items, RowsOf) and not torch.cuda.is_current_stream_capturing()
        if isinstance(candidate_items, RowsOf):
            if candidate_items.table is rated_items:
                cand_emb = rated_emb.index_select(0, candidate_items.index)   # same Linear, already applied to every row
            else:
                cand_emb = native.linear(candidate_items.materialise().float().contiguous(), li.weight.detach(), li.bias.detach())
        else:
            # The candidate projections depend only on the candidate tensor and the weights: a serving loop scores every user
            # against the SAME catalogue tensor (webapp/backend.py:78-121), so the last one seen is kept (like the rated-item
            # catalogue above).  The key holds address, shape and version; the tensor itself is kept alive with the entry, so
            # the address cannot be recycled under it, and an in-place edit bumps the version.
            ckey = (candidate_items.data_ptr(), tuple(candidate_items.shape), candidate_items.dtype, candidate_items._version)
            kept = cache.get("candidates") if keep else None
            if kept is not None and kept[0] == ckey:
                cand_emb, pc_kept = kept[1], kept[2]
            elif att_dense and native.attn_candidates_supported(li.in_features, li.out_features, att_dense):
                # ONE launch: ItemEmbeddings on the candidates, their half of AttentionNet.0, and (in a spare workgroup) the
                # batch's pairs listed by rated set for the grouped attention kernel
                wc, _, b0 = self._att_split(cache)
                B = candidate_items.shape[0]
                R = ratings.rowptr.numel() - 1
                fuse_grouping = grouped and not return_attention_weights and B <= 32768 and R <= 32768
                ppw = native.default_pairs_per_wg(B)
                if "cand_packed" not in cache:       # ItemEmbeddings' weight in MFMA operand order: once per weight version
                    cache["cand_packed"] = native.PackedCandidateWeight(li.weight.detach())
                cand_emb, pc, grouping = native.attn_candidates(
                    candidate_items.float().contiguous(), cache["cand_packed"], li.bias.detach(), wc, b0,
                    ratings.pair_row.to(torch.int64).contiguous() if fuse_grouping else None, R, ppw)
                if grouping is not None:
                    grouping = (grouping, ppw)
            else:
                cand_emb = native.linear(candidate_items.float().contiguous(), li.weight.detach(), li.bias.detach())
        if pc is not None:
            pass
        elif pc_kept is not None:
            pc = pc_kept
        elif self.use_cos_sim_instead:
            pc = native.l2_normalize_rows(cand_emb)
        else:
            wc, _, b0 = self._att_split(cache)
            pc = native.linear(cand_emb, wc, b0)
        if keep and pc_kept is None:
            cache["candidates"] = (ckey, cand_emb, pc, candidate_items)
        if self.use_cos_sim_instead:
            mode, w1, b1 = native.ATT_COS, None, 0.0
        else:
            if self.att_dense:
                l1 = self.AttentionNet[-1]
                if "att_out" not in cache:
                    cache["att_out"] = ((l1.weight.detach().reshape(-1) * 2.0 ** native.ATT_SCALE_LOG2).contiguous(),
                                        float(l1.bias.detach().item()))
                w1, b1 = cache["att_out"]
                mode = native.ATT_MLP_SCALED
            else:
                mode, w1, b1 = native.ATT_LINEAR, None, 0.0
        if grouped:
            # several pairs per rated set: stage each set once per workgroup in LDS (K3 grouped form)
            tail = None if return_attention_weights else self._tail_mlp(cache)
            res = native.attn_forward_grouped(mode, pc, pr, w1, b1, ratings.rowptr, ratings.col, ratings.val,
                                              ratings.pair_row, proj, out_bias=lu.bias.detach(),
                                              return_weights=return_attention_weights, grouping=grouping,
                                              nnz_hint=getattr(ratings, "nnz_hint", None), leave_partials=tail is not None)
            if tail is not None:
                # ONE launch: merge of the attention's partials (+ UserEmbeddings' bias), cat(candidate_emb, user_emb), MLP (:208-222)
                parts = isinstance(res, native.AttnPartials)
                return native.attn_tail(cand_emb, res, lu.bias.detach() if parts else None, *tail)
            if return_attention_weights:
                out = self._score(cand_emb, None, res[0], None, cache=cache)
                return out, ratings.expanded().to_dense(res[1])
            return self._score(cand_emb, None, res, None, cache=cache)
        if shared:
            ratings = ratings.expanded()
        user_emb, wts = native.attn_forward(mode, pc, pr, w1, b1, ratings.rowptr, ratings.col, ratings.val, proj,
                                            out_bias=lu.bias.detach())
        out = self._score(cand_emb, None, user_emb, None, cache=cache)  # cat(candidate_emb, user_emb): :219
        if return_attention_weights:
            return out, ratings.to_dense(wts)
        return out

    # ------------------------------------------------------------------------------------------ torch training path
    def _forward_train_hip(self, candidate_items, rated_items, user_matrix, return_attention_weights):
        """attention_ncf.py:136-224 with autograd recording, on the HIP blocks: the Linear layers (LinearFn), the attention with its
        softmax and weighted sum (AttnFn: forward and backward kernels; AttentionNet's hidden dropout regenerated from a hash in
        both), the MLP (mlp_train).  The train-only target masking (:195-205, a candidate must not attend to itself) is decided
        per rated ENTRY with the reference's own test (isclose of the two embedding rows) and applied by dropping the entry."""
        from ...autograd import AttnFn, LinearFn, mlp_train
        li, lu = self.ItemEmbeddings[0], self.UserEmbeddings[0]
        cand = candidate_items.float().contiguous()
        rated = rated_items.float().contiguous()
        cand_emb = LinearFn.apply(cand, li.weight, li.bias, False)
        rated_emb = LinearFn.apply(rated, li.weight, li.bias, False)
        ratings = user_matrix if isinstance(user_matrix, SparseRatings) else SparseRatings.from_dense(user_matrix, share_identical_rows=False)
        if ratings.pair_row is not None:
            ratings = ratings.expanded()
        rowptr, col, val = ratings.rowptr, ratings.col, ratings.val
        if self.training and col.numel():
            b_of = torch.repeat_interleave(torch.arange(rowptr.numel() - 1, device=col.device), rowptr[1:] - rowptr[:-1])
            same = torch.isclose(cand_emb.detach()[b_of], rated_emb.detach()[col.long()], atol=1e-5).all(dim=1)
            col = torch.where(same, torch.full_like(col, -1), col)        # a dropped entry: score -inf, weight 0 (:192-205)
        dropout = None
        if self.use_cos_sim_instead:
            mode, w1, b1 = native.ATT_COS, None, None
            pc, pr = F.normalize(cand_emb, p=2, dim=1), F.normalize(rated_emb, p=2, dim=1)
        else:
            l0, l1 = self.AttentionNet[0], self.AttentionNet[-1]
            IE = li.out_features
            pc = LinearFn.apply(cand_emb, l0.weight[:, :IE], l0.bias, False)      # AttentionNet.0 split at the cat boundary (:176)
            pr = LinearFn.apply(rated_emb, l0.weight[:, IE:], None, False)
            mode, w1, b1 = native.ATT_MLP, l1.weight.view(-1), l1.bias
            drop = self.AttentionNet[2]
            if self.training and isinstance(drop, nn.Dropout) and drop.p > 0:
                dropout = (float(drop.p), int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))    # host generator: no device sync
        proj = LinearFn.apply(rated, lu.weight, None, False)        # UserEmbeddings is linear over the weighted sum (:212-216)
        user_emb, wts = AttnFn.apply(pc, pr, w1, b1, proj, lu.bias, rowptr, col, val, mode, dropout)
        out = mlp_train(self.MLP, torch.cat((cand_emb, user_emb), dim=1))
        if return_attention_weights:
            dense = SparseRatings(rowptr, ratings.col, val, ratings.num_items).to_dense(wts.detach())
            return out, dense
        return out

    def _forward_train(self, candidate_items, rated_items, user_matrix, return_attention_weights):
        hip = (candidate_items.is_cuda and not getattr(self, "train_with_torch_ops", False) and not (self.training and self.message_dropout)
               and (self.use_cos_sim_instead or self.att_dense) and native.attn_backward_supported(
                   native.ATT_COS if self.use_cos_sim_instead else native.ATT_MLP,
                   self.ItemEmbeddings[0].out_features if self.use_cos_sim_instead else int(self.att_dense), self.UserEmbeddings[0].out_features))
        if hip:
            return self._forward_train_hip(candidate_items, rated_items, user_matrix, return_attention_weights)
        if isinstance(user_matrix, SparseRatings):
            user_matrix = user_matrix.expanded()
            user_matrix = user_matrix.to_dense(user_matrix.val)
        B, I = candidate_items.shape[0], rated_items.shape[0]
        cand_emb = self.ItemEmbeddings(candidate_items)
        rated_emb = self.ItemEmbeddings(rated_items)
        valid = user_matrix != 0
        pairs = valid.nonzero()
        c, r = cand_emb[pairs[:, 0]], rated_emb[pairs[:, 1]]
        if self.use_cos_sim_instead:
            att = (F.normalize(c, p=2, dim=1) * F.normalize(r, p=2, dim=1)).sum(dim=1)
        else:
            att = self.AttentionNet(torch.cat((c, r), dim=1)).view(-1)
        if self.training and self.message_dropout is not None:
            att = F.dropout(att, p=self.message_dropout, training=True)
            att = torch.where(att == 0, torch.full_like(att, -float('inf')), att)
        scores = torch.full((B, I), -float('inf'), dtype=torch.float32, device=att.device)
        scores[valid] = att
        if self.training:
            # the candidate itself must not attend to itself while fitting its rating (:195-205)
            try:
                same = torch.isclose(cand_emb.unsqueeze(1), rated_emb.unsqueeze(0), atol=1e-5).all(dim=2)
                scores = scores.masked_fill(same, -float('inf'))
            except Exception:  # the reference only warns here
                print("Warning: Could not calculate training mask. Ignoring masking this time.", file=sys.stderr)
        scores = F.softmax(scores, dim=1).nan_to_num(nan=0.0, posinf=0.0, neginf=0.0)
        user_feat = torch.matmul(scores * user_matrix, rated_items)
        user_emb = self.UserEmbeddings(user_feat)
        out = self.MLP(torch.cat((cand_emb, user_emb), dim=1))
        return (out, scores.detach()) if return_attention_weights else out
