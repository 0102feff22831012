/*
 * ncf_abi.h — C ABI of libncf_hip.so, the MI355X (gfx950) hot path of the NCF scoring forward.
 *
 * The reference (michaelbzms/DeepRecommendation) is 100 % Python on PyTorch; it has no FFI, no custom
 * operator and no native code.  The "interface each entry point replaces" is therefore the sequence of
 * ATen ops a reference forward() issues; each declaration below cites it as
 * src/neural_collaborative_filtering/<file>:<line> of the reference.
 *
 * Conventions (all entry points):
 *   - extern "C", plain pointers and sizes, no torch / C++ types.
 *   - return 0 (NCF_OK) or a negative ncf_status; never throw.  ncf_last_error() returns a thread-local
 *     human-readable message for the last failure on the calling thread.
 *   - every pointer named dev_* / documented "device" is a HIP device pointer owned by the caller; the
 *     library never allocates persistent device memory, never synchronises the stream, never copies to the
 *     host.  Work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 *   - index arrays are int64 at the ABI (the reference uses torch.long: datasets/gnn_datasets.py:28);
 *     CSR column ids are int32 (node counts < 2^31).
 *   - leading dimensions (ld*) are in ELEMENTS of the tensor's dtype.
 *   - re-entrant and thread-safe.  State held by the library, all of it listed here: the thread-local error
 *     string; the process-wide option table written only by ncf_set_option (atomic ints, every option defaults to
 *     0 = "choose by shape and size"; no entry point reads the environment); a per-device cache of the CU count;
 *     and the dynamic-LDS limit of the attention kernels, raised once per kernel instantiation and device.
 *   - out-of-range indices never fault: the offending row is read as zeros and, when `dev_oob_flag` is
 *     non-NULL, *dev_oob_flag is set to 1 (the host wrapper turns that into IndexError on request, which is
 *     what torch indexing raises in the reference).
 */
#ifndef NCF_ABI_H
#define NCF_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCF_ABI_VERSION 1

typedef void* ncf_stream_t; /* hipStream_t */

enum ncf_dtype { NCF_F32 = 0, NCF_BF16 = 1 };

enum ncf_status {
    NCF_OK = 0,
    NCF_EINVAL = -1,       /* bad argument (null pointer, negative size, misaligned, inconsistent dims) */
    NCF_EUNSUPPORTED = -2, /* shape/dtype has no specialised kernel: caller should take the unfused entry points */
    NCF_ELAUNCH = -3,      /* HIP reported an error at launch */
    NCF_EWORKSPACE = -4    /* workspace too small */
};

int ncf_version(void);
const char* ncf_last_error(void);
/* Architecture the device code was compiled for ("gfx950"). */
const char* ncf_build_arch(void);
/* Hash (16 hex digits) of the sources, headers and compiler flags this library was built from — csrc/build.py:source_id().  The
 * Python binding and __graft_entry__.build() refuse a library whose id differs from the sources next to it (a stale .so). */
const char* ncf_build_id(void);

/* Process-wide kernel-selection overrides for A/B measurements and for tests that must drive every kernel variant
 * (no counterpart upstream).  0 always means "choose by shape and size" (the default).  Options:
 *   "bf16_kernel"          1 = 4-wave weight-stationary kernel, 2 = slab-streaming kernel, 3 = 8-wave weight-stationary kernel
 *                          (ncf_score_fused, NCF_BF16; 3 falls back to 1 on shapes it does not take)
 *   "linear_kernel"        1 = one row tile per wave, 2 = persistent row-streaming form          (ncf_mlp_forward / ncf_linear_forward)
 *   "linear_kslices"       4 | 8 = K-slices of the skinny-deep Linear form
 *   "attn_grouped_kernel"  1 = LDS-broadcast form, 2 = scalar-operand form                       (ncf_attn_forward_grouped)
 *   "gather_kernel"        1 = one step per wave, 2 = persistent prefetching waves                (ncf_gather_concat)
 * Every variant of one entry computes the same function (bit-identical where the entry promises it).
 * NCF_EINVAL for an unknown name or a value outside the option's set. */
int ncf_set_option(const char* name, int value);
int ncf_get_option(const char* name, int* value);

/* ------------------------------------------------------------------------------------------------
 * K1  embedding gather (+ fused concat)
 * Replaces: nn.Linear applied to one-hot rows + torch.cat — models/basic_ncf.py:38-40, models/mf.py:29-30;
 *           combined_graph_emb[itemIds] / [userIds] + cat — models/gnn_ncf.py:354-361.
 * out[p, 0:EA] = tabA[idxA[p], 0:EA];  out[p, EA:EA+EB] = tabB[idxB[p], 0:EB]     (bit-exact copy)
 * tabB/idxB may be NULL with EB = 0 (single-table gather).  idxA / idxB may be NULL = identity (row p).
 * ------------------------------------------------------------------------------------------------ */
int ncf_gather_concat(int dtype,
                      const void* dev_tabA, int64_t rowsA, int64_t ldA,
                      const void* dev_tabB, int64_t rowsB, int64_t ldB,
                      const int64_t* dev_idxA, const int64_t* dev_idxB,
                      int64_t B, int EA, int EB,
                      void* dev_out, int64_t ldOut,
                      int32_t* dev_oob_flag, ncf_stream_t stream);

/* Row-wise dot product of two gathered rows, out[p] = sum_e tabA[idxA[p], e] * tabB[idxB[p], e]  (fp32
 * accumulate, fp32 out).  Replaces: torch.bmm(user_emb.unsqueeze(1), item_emb.unsqueeze(2)) —
 * models/mf.py:31, models/gnn_ncf.py:365. */
int ncf_gather_dot(int dtype,
                   const void* dev_tabA, int64_t rowsA, int64_t ldA,
                   const void* dev_tabB, int64_t rowsB, int64_t ldB,
                   const int64_t* dev_idxA, const int64_t* dev_idxB,
                   int64_t B, int E, float* dev_out,
                   int32_t* dev_oob_flag, ncf_stream_t stream);

/* Exchange preparation for ROW-SHARDED tables (BASELINE config 5: the table outgrows one GPU; the reference is single
 * device — its lookup is the nn.Linear on one-hot rows of models/basic_ncf.py:38-39 — so this has no upstream counterpart).
 * Rank o of `world` owns rows [o * rows_per_rank, (o+1) * rows_per_rank) of a table of total_rows rows.  Lists the batch's
 * ids by owner in a FIXED-capacity send buffer so that the all-to-all exchanges that follow use equal splits and need
 * no size on the host:
 *   dev_send[o * cap + k] = id - o * rows_per_rank   for the k-th id of owner o (k < cap; order inside a bucket unspecified);
 *                           unused slots are 0 (a valid local row wherever the shard is not empty)
 *   dev_slot[p]           = o * cap + k  — the row of pair p in the (world * cap)-row buffer the row exchange returns;
 *                           -1 for an id outside [0, total_rows) (sets *dev_oob_flag) or beyond its bucket's capacity
 *                           (sets *dev_overflow_flag): such a pair then reads as an out-of-range row downstream
 *   dev_counts[o]         = number of ids owned by o (may exceed cap)
 * dev_send (world * cap) and dev_counts (world) are zero-filled by this call.  world <= 1024. */
int ncf_bucket_ids(const int64_t* dev_idx, int64_t B, int64_t rows_per_rank, int64_t total_rows, int world, int64_t cap,
                   int64_t* dev_send, int64_t* dev_slot, int32_t* dev_counts,
                   int32_t* dev_oob_flag, int32_t* dev_overflow_flag, ncf_stream_t stream);

/* De-duplicating form of ncf_bucket_ids ("dedup duplicate ids per destination before sending"): every DISTINCT id
 * of the batch takes one slot of its owner's bucket and all of its pairs share it (dev_slot), through a hash set in device memory
 * — no sort, no size on the host.  Buckets carry their count: dev_send holds world buckets of (cap + 1) int64,
 * [count, id_0 .. id_{cap-1}], so the counts travel with the id all-to-all and the owner gathers count rows per bucket
 * (ncf_gather_buckets); padding is never initialised.  dev_slot[p] indexes the (world * cap) rows that come back.
 * dev_hkeys / dev_hvals: table_slots >= ncf_bucket_dedup_table_slots(B) int64 each (a power of two >= 2 B), caller-owned scratch.
 * Flags as ncf_bucket_ids.  The reference has no counterpart (single device). */
size_t ncf_bucket_dedup_table_slots(int64_t B);
int ncf_bucket_ids_dedup(const int64_t* dev_idx, int64_t B, int64_t rows_per_rank, int64_t total_rows, int world, int64_t cap,
                         int64_t* dev_hkeys, int64_t* dev_hvals, int64_t table_slots,
                         int64_t* dev_send, int64_t* dev_slot, int32_t* dev_counts,
                         int32_t* dev_oob_flag, int32_t* dev_overflow_flag, ncf_stream_t stream);

/* The owner's gather for the buckets of ncf_bucket_ids_dedup: dev_recv = world buckets of [count, ids ...] as received;
 * out row (r * cap + k) = table[recv[r][1 + k]] for k < count_r; rows beyond a bucket's count are neither read nor written.
 * Rows must be multiples of 16 bytes.  An id outside [0, rows) gives a zero row and sets *dev_oob_flag. */
int ncf_gather_buckets(int dtype, const void* dev_table, int64_t rows, int64_t ld, const int64_t* dev_recv, int world, int64_t cap, int E,
                       void* dev_out, int64_t ldout, int32_t* dev_oob_flag, ncf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K2  MLP forward, generic layer-by-layer path (any dims)
 * Replaces: nn.Sequential(Linear, [ReLU, Dropout(eval = identity), Linear]*) — util.py:5-18, called at
 *           models/basic_ncf.py:41, models/attention_ncf.py:179,222, models/gnn_ncf.py:362; and single
 *           nn.Linear layers (n_layers = 1): models/attention_ncf.py:150-151,216, models/gnn_ncf.py:300-301,
 *           the hoisted per-node form of gnn_ncf.py:91-93.
 * dims[0] = input width, dims[i] = out_features of layer i (i = 1..n_layers).  W[i] is device [dims[i+1]][dims[i]]
 * row-major (torch nn.Linear.weight layout), b[i] is device [dims[i+1]] (may be NULL = no bias).
 * ReLU is applied between layers, never after the last one.  Intermediates live in `dev_workspace`
 * (ncf_mlp_workspace_bytes).  fp32 only on this entry (NCF_F32); accumulation is an exact fp32 FMA chain
 * (v_mfma_f32_32x32x2_f32).
 * `host_W` / `host_b` are HOST arrays of DEVICE pointers.
 * ------------------------------------------------------------------------------------------------ */
size_t ncf_mlp_workspace_bytes(int dtype, int64_t B, int n_layers, const int* dims);
int ncf_mlp_forward(int dtype, const void* dev_x, int64_t B, int64_t ldx,
                    int n_layers, const int* dims,
                    const void* const* host_W, const void* const* host_b,
                    void* dev_workspace, size_t workspace_bytes,
                    void* dev_out, int64_t ldOut, ncf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K1+K2 fused: gather -> concat -> MLP -> (B,1) without the (B, EA+EB) round trip through HBM.
 * Replaces the whole of models/basic_ncf.py:37-42 (table form) and models/gnn_ncf.py:354-362.
 *
 * The MLP must end in a 1-wide layer (util.py:10 output_size = 1): dims = {EA+EB, h1, [h2,] 1}.
 * Weights are passed PRE-PACKED (ncf_mlp_pack) in the lane order the MFMA operands are consumed in, so a
 * wave streams them with fully coalesced 16-byte loads; pack once per model, reuse for every batch.
 * Returns NCF_EUNSUPPORTED when (dtype, EA, EB, dims) has no specialised instance — use
 * ncf_gather_concat + ncf_mlp_forward then.  ncf_score_fused_supported() answers without launching.
 * idxA / idxB NULL = identity, so a dense activation matrix x (B, K0) can be scored with tabA = x, EB = 0.
 * ------------------------------------------------------------------------------------------------ */
int ncf_score_fused_supported(int dtype, int EA, int EB, int n_layers, const int* dims);
size_t ncf_mlp_packed_bytes(int dtype, int n_layers, const int* dims);
int ncf_mlp_pack(int dtype, int n_layers, const int* dims,
                 const void* const* host_W, const void* const* host_b,
                 void* dev_packed, size_t packed_bytes, ncf_stream_t stream);
int ncf_score_fused(int dtype,
                    const void* dev_tabA, int64_t rowsA, int64_t ldA,
                    const void* dev_tabB, int64_t rowsB, int64_t ldB,
                    const int64_t* dev_idxA, const int64_t* dev_idxB,
                    int64_t B, int EA, int EB,
                    int n_layers, const int* dims, const void* dev_packed,
                    float* dev_out, int32_t* dev_oob_flag, ncf_stream_t stream);

/* Opt-in inference-time folding of the first MLP layer into the tables (frozen weights):
 *   relu(W1 . cat(a, b) + b1) == relu(PA[ia] + PB[ib])  with  PA = TA . W1[:, :EA]^T + b1  (rowsA, N1)  and
 *   PB = TB . W1[:, EA:]^T  (rowsB, N1), both built once by the caller (ncf_mlp_forward with n_layers = 1).
 * out[p] = tail( relu(PA[idxA[p]] + PB[idxB[p]]) ),  tail = the remaining MLP [N1 -> N2 -> 1] packed by
 * ncf_mlp_pack(dims = {N1, N2, 1}).  Same function as ncf_score_fused up to fp32 summation order; layer 1 costs no
 * matrix work, the gathered rows are N1 wide (4x the bytes at E = 64, N1 = 256).  Replaces basic_ncf.py:38-41 /
 * gnn_ncf.py:354-362 like ncf_score_fused.  NCF_EUNSUPPORTED when (N1, N2) has no instance. */
int ncf_score_folded_supported(int dtype, int N1, int N2);
int ncf_score_folded(int dtype,
                     const void* dev_PA, int64_t rowsA, int64_t ldPA,
                     const void* dev_PB, int64_t rowsB, int64_t ldPB,
                     const int64_t* dev_idxA, const int64_t* dev_idxB,
                     int64_t B, int N1, int N2, const void* dev_packed_tail,
                     float* dev_out, int32_t* dev_oob_flag, ncf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K4/K5  LightGCN propagation as CSR-by-destination SpMM with a wavefront segmented reduction
 * Replaces: PyG MessagePassing.propagate(aggr='add') + message() — models/gnn_ncf.py:52-70,74-94
 *           (per-edge Linear hoisted to a per-node Linear, which is ncf_mlp_forward with n_layers = 1),
 *           torch_geometric.utils.degree + deg.pow(-0.5) — gnn_ncf.py:47-50,
 *           torch.mean(torch.stack(hs)) — gnn_ncf.py:351 (running sum fused into the SpMM epilogue).
 *
 * Segment s owns edges [segptr[s], segptr[s+1]) and belongs to destination row row_of[s] (row_of == NULL:
 * row s, i.e. segptr is a plain CSR rowptr).  Long rows may be split by the caller into several CONSECUTIVE
 * segments (load balance under Zipf-skewed degrees); their partial sums are added in segment order, so
 *   y[n, :] = sum over the segments of row n, in order, of  sum_{e in segment} coef[e] * z[col[e], :]
 * is bitwise reproducible (no float atomics).  Every row must own at least one (possibly empty) segment.
 * if dev_sum != NULL:  dev_sum[n, :] += y[n, :]   (layer-sum accumulator for the final mean)
 * z is (Nz, D) with leading dimension ldz; y and sum are (N, D).  D % 4 == 0, D <= 256.
 * coef may be NULL (all ones).  dev_partial: (n_seg, D) floats scratch, required when dev_row_of != NULL.
 * fixup = 1: rows split over several segments are completed inside this call (one lane group per row walks its
 *   partials in order — simple, but serial per row).
 * fixup = 0: such rows are NOT written; their per-segment sums stay in dev_partial[s, :] and the caller finishes them
 *   with further calls of this same entry (z = dev_partial, col = segment ids, coef = NULL, segments of the partial
 *   list) until every row is down to one segment — a log-depth ordered tree, used for hub rows with millions of edges.
 * ------------------------------------------------------------------------------------------------ */
int ncf_spmm_csr(int dtype, const int64_t* dev_segptr, const int32_t* dev_row_of, int64_t n_seg,
                 const int32_t* dev_col, const float* dev_coef,
                 const void* dev_z, int64_t Nz, int64_t ldz, int D,
                 void* dev_y, int64_t ldy, float* dev_sum, int64_t ldsum,
                 float* dev_partial, int fixup, ncf_stream_t stream);

/* Training-step form of ncf_spmm_csr: the reference applies Dropout INSIDE the per-edge Linear of a LightGCN layer
 * (gnn_ncf.py:22-31, 91-93: message_e = coef_e * dropout_p(W(x[src_e])), one mask element per (edge, feature), kept values
 * scaled by 1 / (1 - p)), which a per-node hoisted Linear cannot express.  Here the mask is regenerated inside the kernel from a
 * counter-based hash of (seed, edge id, feature): element (e, f) of the mask is a pure function of the three, so
 *   forward   y  = sum_e coef_e * mask(id_e, :) / (1 - p') * z[col_e]        (CSR by destination, dev_edge_id = NULL: id = position)
 *   backward  dz = the same call on the CSR by SOURCE with dev_edge_id[.] = each entry's position in the forward CSR
 * use the same mask without storing it.  p' = round(p * 65536) / 65536 (the keep test is on 16 hash bits).  The masks are
 * NOT torch's Philox stream: a seeded run differs from the reference's in the (random) masks drawn, not in their law.
 * Only edge-level calls take a mask (the partial-sum tree levels of split rows go through ncf_spmm_csr).  p = 0: ncf_spmm_csr. */
int ncf_spmm_csr_dropout(int dtype, const int64_t* dev_segptr, const int32_t* dev_row_of, int64_t n_seg,
                         const int32_t* dev_col, const float* dev_coef,
                         const void* dev_z, int64_t Nz, int64_t ldz, int D,
                         void* dev_y, int64_t ldy, float* dev_sum, int64_t ldsum,
                         float* dev_partial, int fixup, const int32_t* dev_edge_id, uint32_t seed, float p, ncf_stream_t stream);

/* deg[n] = number of edges whose destination is n (float, exact below 2^24) — PyG degree(), gnn_ncf.py:48.
 * dev_deg must be zero-filled by the caller; two calls (u2i, i2u) accumulate into the same array, which is
 * the cat at gnn_ncf.py:41. */
int ncf_degree_accumulate(const int64_t* dev_dst, int64_t E, int64_t N, float* dev_deg,
                          int32_t* dev_oob_flag, ncf_stream_t stream);
/* coef[e] = (attr ? attr[e] : 1) * (dis[src[e]] * dis[dst[e]]),  dis = deg^-1/2 with inf -> 0
 * (gnn_ncf.py:49-50,54,58,66 and the product order of :91 / :93). */
int ncf_edge_coef(const int64_t* dev_src, const int64_t* dev_dst, const float* dev_attr,
                  const float* dev_deg, int64_t E, int64_t N, float* dev_coef, ncf_stream_t stream);
/* LightGAT edge attention over a CSR-by-destination graph (models/gnn_ncf.py:151-177; PyG softmax(src, index) =
 * exp(src - max_group) / (sum_group + 1e-16)):
 *   dev_out[e] = (attr ? attr[e] : 1) * softmax over the edges of e's destination row of s[col[.]]
 * s[n] = w_j . x[n] is the source half of AttNet(cat(x_j, x_i)) (the destination half + bias is constant inside a
 * group and cancels in the softmax).  The result is the per-edge coefficient for ncf_spmm_csr. */
int ncf_edge_softmax_csr(const int64_t* dev_rowptr, const int32_t* dev_col, const float* dev_attr, const float* dev_s,
                         int64_t n_rows, int64_t Ns, float* dev_out, ncf_stream_t stream);
/* The same function over the SEGMENTS of the SpMM's split rows (dev_segptr / dev_row_of of ncf_spmm_csr, dev_seg_first[r] = first
 * segment of row r, n_rows + 1 entries): three passes parallel over segments / rows instead of one wave per destination — a hub
 * destination with millions of in-edges no longer serialises.  Fixed order, no atomics; equal to ncf_edge_softmax_csr up to the
 * fp32 summation order.  workspace: ncf_edge_softmax_segmented_workspace_bytes(n_seg, n_rows). */
size_t ncf_edge_softmax_segmented_workspace_bytes(int64_t n_seg, int64_t n_rows);
int ncf_edge_softmax_segmented(const int64_t* dev_segptr, const int32_t* dev_row_of, int64_t n_seg, const int64_t* dev_seg_first,
                               int64_t n_rows, const int32_t* dev_col, const float* dev_attr, const float* dev_s, int64_t n_src,
                               float* dev_out, void* dev_workspace, size_t workspace_bytes, ncf_stream_t stream);

/* out[n, :] = in[n, :] / divisor  (the division of torch.mean over the L+1 stacked layers, gnn_ncf.py:351). */
int ncf_scale_rows(const float* dev_in, int64_t ldin, int64_t N, int D, float divisor,
                   float* dev_out, int64_t ldout, ncf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * K3  AttentionNCF item-item attention: scores -> masked row softmax -> rating-weighted aggregation
 * Replaces models/attention_ncf.py:154-213 (eval mode).  Inputs are the *projected* operands:
 *   att_dense > 0 (AttentionNet = Linear(2·IE -> A), ReLU, Linear(A -> 1), :112-117):
 *       dev_pc (B, A)   = cand_emb  @ W0[:, :IE]^T + b0        (candidate half of AttentionNet.0, bias folded)
 *       dev_pr (I, A)   = rated_emb @ W0[:, IE:]^T            (rated half)
 *       score(b, i)     = b1 + sum_a w1[a] * relu(pc[b, a] + pr[i, a])
 *     which is AttentionNet(cat(cand, rated)) with the first Linear split at the concat boundary (:176).
 *   att_dense == 0 (AttentionNet = Linear(2·IE -> 1), :120-122): A = 1, no ReLU:
 *       score(b, i) = pc[b, 0] + pr[i, 0]
 *   cosine (:162-173): dev_pc / dev_pr are the L2-normalised embeddings (B, IE) / (I, IE), mode = NCF_ATT_COS,
 *       score = dot.
 * The user's rated set is CSR: row b owns entries [rowptr[b], rowptr[b+1]) with col = position in the rated
 * list (0..I-1) and val = the non-zero user_matrix entry (:158 `user_matrix != 0`).
 *   w = softmax over the row's entries (empty row -> all zeros, :208-209)
 *   dev_out_feat[b, :] = bias + sum_i  w[b,i] * val[b,i] * feat[col_i, :]      (:212-213), feat is (I, Fdim)
 *     feat may be the raw item features (Fdim = F; bias NULL; UserEmbeddings applied afterwards, :216) or, using
 *     the linearity of UserEmbeddings over the weighted sum, the pre-projected rows feat = rated_items @ Wu^T
 *     (Fdim = UE) with dev_out_bias = bu — then dev_out_feat IS user_embeddings of :216.
 *   dev_weights ((nnz,) floats aligned with the CSR entries, REQUIRED: it doubles as the score scratch) receives
 *     the attention weights w (what return_attention_weights exposes, :224).
 * ------------------------------------------------------------------------------------------------ */
/* NCF_ATT_MLP_SCALED: NCF_ATT_MLP whose caller pre-multiplied dev_pc and dev_pr by 2^-NCF_ATT_SCALE_LOG2 and dev_w1 by
 * 2^NCF_ATT_SCALE_LOG2 (fold the factors into the weights that produce them: exact, powers of two).  The function is
 * unchanged bit for bit — w' relu(p' + q') == w relu(p + q) — but in scaled units every sum lies in (-1, 1), so the
 * grouped kernel may evaluate relu as the [0, 1] CLAMP output modifier of its packed add (gfx950 has no packed fp32 max):
 * half the vector instructions of the score loop.  Differs from relu only where |pc + pr| >= 2^64 in true units (scores
 * that are numerically meaningless in fp32 anyway: they saturate instead of overflowing); a NaN sum gives 0 as with max. */
enum ncf_att_mode { NCF_ATT_MLP = 0, NCF_ATT_LINEAR = 1, NCF_ATT_COS = 2, NCF_ATT_MLP_SCALED = 3 };
#define NCF_ATT_SCALE_LOG2 64
int ncf_attn_forward(int mode,
                     const float* dev_pc, int64_t ldpc, const float* dev_pr, int64_t ldpr, int A,
                     const float* dev_w1, float b1,
                     const int64_t* dev_rowptr, const int32_t* dev_col, const float* dev_val,
                     int64_t B, int64_t I,
                     const float* dev_feat, int64_t ldfeat, int Fdim,
                     const float* dev_out_bias,
                     float* dev_out_feat, int64_t ldout,
                     float* dev_weights,
                     ncf_stream_t stream);

/* Training-step forms of ncf_attn_forward (the reference differentiates attention_ncf.py:176-216 through autograd):
 *   ncf_attn_forward_dropout  NCF_ATT_MLP with AttentionNet's hidden Dropout(p) active (attention_ncf.py:112-117: one mask element
 *       per (pair, rated entry, hidden unit), kept values scaled by 1 / (1 - p')): the mask is regenerated in the kernel from a
 *       counter-based hash of (seed, CSR entry, unit) — NOT torch's Philox stream, same law.  p' = round(p * 65536) / 65536.
 *   ncf_attn_backward         gradients of ncf_attn_forward[_dropout] given dev_weights (the attention weights the forward wrote)
 *       and dev_dout (B, Fdim) = d loss / d dev_out_feat:
 *         dev_d_pc (B, A)            written
 *         dev_d_pr (I, A)            ADDED to (float atomics): zero it first          dev_d_feat (I, Fdim)  likewise
 *         dev_d_w1_part (B, A)       per-pair partial rows of d w1 (NCF_ATT_MLP; column-sum them with ncf_colsum)
 *       d b1 is exactly 0 (a shift of all scores cancels in the softmax); d out_bias = column sums of dev_dout (ncf_colsum).
 *       dev_scratch: nnz floats.  seed / p as in the forward call (p = 0: no dropout).  NCF_ATT_MLP(_SCALED) and NCF_ATT_COS;
 *       A, Fdim % 4 == 0 and <= 256. */
int ncf_attn_forward_dropout(int mode,
                             const float* dev_pc, int64_t ldpc, const float* dev_pr, int64_t ldpr, int A,
                             const float* dev_w1, float b1,
                             const int64_t* dev_rowptr, const int32_t* dev_col, const float* dev_val,
                             int64_t B, int64_t I,
                             const float* dev_feat, int64_t ldfeat, int Fdim, const float* dev_out_bias,
                             float* dev_out_feat, int64_t ldout, float* dev_weights,
                             uint32_t seed, float p, ncf_stream_t stream);
int ncf_attn_backward(int mode,
                      const float* dev_pc, int64_t ldpc, const float* dev_pr, int64_t ldpr, int A, const float* dev_w1,
                      const int64_t* dev_rowptr, const int32_t* dev_col, const float* dev_val, int64_t B, int64_t I,
                      const float* dev_feat, int64_t ldfeat, int Fdim,
                      const float* dev_weights, const float* dev_dout, int64_t lddout,
                      float* dev_d_pc, int64_t ldd_pc, float* dev_d_pr, int64_t ldd_pr, float* dev_d_w1_part,
                      float* dev_d_feat, int64_t ldd_feat, float* dev_scratch,
                      uint32_t seed, float p, ncf_stream_t stream);

/* The same computation for batches in which several pairs share one rated set (evaluation batches grouped by user;
 * serving one user against the whole catalogue, reference webapp/backend.py:78-121), LDS-tiled: the CSR has one row
 * per USER (n_rows rows), the B pairs are listed user by user —
 *   dev_grp_ptr (n_rows+1):  pairs of row r are dev_pair_ids[grp_ptr[r] .. grp_ptr[r+1])  (indices into pc / out rows)
 *   dev_wg_ptr  (n_rows+1):  exclusive prefix sum of ceil(pairs_of_row / pairs_per_wg)     (workgroups per row)
 * — and a workgroup stages a user's rated rows once per 64 entries into LDS for up to pairs_per_wg (1..32) pairs, with
 * an online softmax over the tiles.  Same result as ncf_attn_forward on the expanded per-pair CSR up to fp32 summation
 * order.  dev_weights (optional) receives the attention weights (models/attention_ncf.py:224) in the layout of that
 * expanded CSR: pair b's weights start at dev_weights[dev_weights_off[b]] and follow its row's entries
 * (dev_weights_off (B,) = exclusive prefix sum of the row lengths of pair_row[b]).
 * pairs_per_wg <= 16 with Fdim % 4 == 0 takes the scalar-operand form (256-thread workgroups: pc rows and w1 as scalar
 * operands through the scalar cache, tiles by LDS-DMA, aggregation on the matrix cores); 17..32 the first form.
 * NCF_EUNSUPPORTED (fall back to ncf_attn_forward): mode NCF_ATT_LINEAR, A % 4 != 0, A > 256, Fdim > 256. */
int ncf_attn_forward_grouped(int mode,
                             const float* dev_pc, int64_t ldpc, const float* dev_pr, int64_t ldpr, int A,
                             const float* dev_w1, float b1,
                             const int64_t* dev_rowptr, const int32_t* dev_col, const float* dev_val,
                             int64_t n_rows, int64_t n_rated,
                             const int64_t* dev_grp_ptr, const int64_t* dev_pair_ids, const int64_t* dev_wg_ptr,
                             int64_t B, int pairs_per_wg,
                             const float* dev_feat, int64_t ldfeat, int Fdim, const float* dev_out_bias,
                             float* dev_out_feat, int64_t ldout,
                             float* dev_weights, const int64_t* dev_weights_off, ncf_stream_t stream);

/* Builds dev_grp_ptr / dev_pair_ids / dev_wg_ptr of ncf_attn_forward_grouped from dev_pair_row (B,) = the CSR row of
 * each pair (what the dense user_matrix expresses by repeating a user's row, dynamic_datasets.py:24-40): a counting
 * sort on the stream, no host synchronisation.  The order of a row's pairs inside dev_pair_ids is unspecified (each
 * pair's output is independent of it).  A pair_row outside [0, n_rows) sets *dev_oob_flag (sticky) and drops the pair.
 * workspace: ncf_group_pairs_workspace_bytes(n_rows) bytes of device memory. */
size_t ncf_group_pairs_workspace_bytes(int64_t n_rows);
int ncf_group_pairs(const int64_t* dev_pair_row, int64_t B, int64_t n_rows, int pairs_per_wg,
                    int64_t* dev_grp_ptr, int64_t* dev_pair_ids, int64_t* dev_wg_ptr,
                    void* dev_workspace, size_t workspace_bytes, int32_t* dev_oob_flag, ncf_stream_t stream);

/* ncf_group_pairs that also writes dev_wg_row[w] = CSR row of workgroup w (int32, ceil(B / pairs_per_wg) + min(n_rows, B) slots —
 * the grid bound of the grouped kernels): ncf_attn_forward_split reads it instead of searching dev_wg_ptr. */
int ncf_group_pairs_rows(const int64_t* dev_pair_row, int64_t B, int64_t n_rows, int pairs_per_wg,
                         int64_t* dev_grp_ptr, int64_t* dev_pair_ids, int64_t* dev_wg_ptr, int32_t* dev_wg_row,
                         void* dev_workspace, size_t workspace_bytes, int32_t* dev_oob_flag, ncf_stream_t stream);

/* Entry-split form of ncf_attn_forward_grouped (same operands and result; attention weights are not available here):
 * the rated set of a row is cut into `nsplit` slices of whole 64-entry tiles, one workgroup per (group of pairs, slice)
 * leaves a softmax partial (running maximum, sum, un-normalised aggregate) in dev_workspace and a second kernel of the same
 * call merges them in slice order (deterministic; merge = 0 leaves them for ncf_attn_tail).  Replaces the same reference lines as ncf_attn_forward_grouped
 * (models/attention_ncf.py:154-216); it exists because a batch of a few thousand pairs is too few workgroups of too long
 * a chain for the one-workgroup-per-group form.  nsplit = 1 needs no workspace and launches one kernel.
 * Shapes: ncf_attn_split_supported() (A % 32 == 0, A <= 256, Fdim 64 or 128, MLP / cosine modes); else NCF_EUNSUPPORTED.
 * dev_wg_row may be NULL (binary search over dev_wg_ptr). */
int ncf_attn_split_supported(int mode, int A, int Fdim, int pairs_per_wg);
size_t ncf_attn_split_workspace_bytes(int64_t B, int Fdim, int nsplit);
int ncf_attn_forward_split(int mode,
                           const float* dev_pc, int64_t ldpc, const float* dev_pr, int64_t ldpr, int A,
                           const float* dev_w1, float b1,
                           const int64_t* dev_rowptr, const int32_t* dev_col, const float* dev_val,
                           int64_t n_rows, int64_t n_rated,
                           const int64_t* dev_grp_ptr, const int64_t* dev_pair_ids, const int64_t* dev_wg_ptr,
                           const int32_t* dev_wg_row, int64_t B, int pairs_per_wg,
                           const float* dev_feat, int64_t ldfeat, int Fdim, const float* dev_out_bias,
                           float* dev_out_feat, int64_t ldout,
                           int nsplit, int merge, void* dev_workspace, size_t workspace_bytes, ncf_stream_t stream);

/* The end of AttentionNCF.forward in one launch (models/attention_ncf.py:208-222): merge of the entry-split attention's partials
 * (ncf_attn_forward_split with merge = 0: dev_part = its workspace, (B, nsplit, UE + 4) floats — [m, l, -, -, O...]) into the user
 * embeddings (+ dev_ubias = UserEmbeddings' bias), cat(candidate_emb, user_emb) (:219), then the MLP of util.py:5-18
 * (EA + UE -> N1 -> N2 -> 1, ReLU between, none after the last).  dev_part == NULL: dev_user holds finished (B, UE) rows.
 * Weights: W1 (N1, EA + UE), W2 (N2, N1), w3 (N2) — weights_packed = 0: W1 / W2 in the reference's own row-major layout;
 * weights_packed = 1: each packed once per weight version by ncf_attn_tail_pack_weight (MFMA operand order: a wave's weight load is
 * one contiguous 1 KB run instead of 16 rows x 64 bytes — 12.8 -> 9.8 us at BASELINE config 3; bit-identical results).  Replaces
 * attn_combine + the fused scoring kernel for evaluation batches (16 pairs per workgroup instead of 32: twice the workgroups).
 * Shapes: ncf_attn_tail_supported (EA = UE in {64, 128}, N1 = 256, N2 = 128); else NCF_EUNSUPPORTED (use ncf_score_fused). */
int ncf_attn_tail_supported(int EA, int UE, int N1, int N2);
int ncf_attn_tail_pack_weight(const float* dev_W, int N, int K, float* dev_packed, ncf_stream_t stream);
int ncf_attn_tail(const float* dev_cand_emb, int64_t ldcand, int EA,
                  const float* dev_part, int nsplit, const float* dev_user, int64_t lduser, int UE, const float* dev_ubias,
                  const float* dev_W1, const float* dev_b1, int N1, const float* dev_W2, const float* dev_b2, int N2,
                  const float* dev_w3, float b3, int weights_packed, float* dev_out, int64_t B, ncf_stream_t stream);

/* Dense user_matrix -> CSR with shared rows, on the stream (no size is read by the host).  The reference passes AttentionNCF.forward a
 * dense (B, I) user_matrix in which a user's row is repeated for each of their samples (datasets/dynamic_datasets.py:24-40,
 * content_providers/dynamic_profiles_provider.py:55-71; one row for every candidate in webapp/backend.py:78-121) and keeps the
 * entries != 0 (models/attention_ncf.py:158).  Two calls around one cumulative sum the caller runs on the stream:
 *   ncf_dense_csr_rows  -> dev_pair_row[b] = the row pair b uses (the smallest pair index with an IDENTICAL row when share_rows != 0
 *                          — found by a row hash and verified element by element —, b itself otherwise) and dev_rowptr (B + 1) =
 *                          [0, entries row 0 will list, entries row 1 will list, ...] (0 for a row that shares another's)
 *   an inclusive cumulative sum over dev_rowptr, IN PLACE, makes it the CSR's rowptr
 *   ncf_dense_csr_fill  -> the representatives' (col, val) in column order at rowptr[b]; dev_col / dev_val must hold rowptr[B]
 *                          entries (B * I is always enough).
 * -0.0 counts as 0 (unrated); a row containing NaN never shares.  workspace: ncf_dense_csr_workspace_bytes(B), 16-byte aligned. */
size_t ncf_dense_csr_workspace_bytes(int64_t B);
int ncf_dense_csr_rows(const float* dev_user_matrix, int64_t ld, int64_t B, int64_t I, int share_rows,
                       int64_t* dev_pair_row, int64_t* dev_rowptr, void* dev_workspace, size_t workspace_bytes, ncf_stream_t stream);
int ncf_dense_csr_fill(const float* dev_user_matrix, int64_t ld, int64_t B, int64_t I, const int64_t* dev_rowptr,
                       const int64_t* dev_pair_row, int32_t* dev_col, float* dev_val, ncf_stream_t stream);

/* The candidate side of AttentionNCF.forward in one launch — models/attention_ncf.py:150 (ItemEmbeddings on the candidates) and
 * the candidate half of AttentionNet's first Linear (:176-179 with AttentionNet.0.weight split at the cat boundary):
 *     emb = x . Wi^T + bi   (B, N1);     pc = emb . Wc^T + b0   (B, N2)
 * x (B, K) rows of ldx floats (4-byte aligned rows are fine: K = 2094 in the reference's data), Wi (N1, K) rows of ldw floats,
 * Wc contiguous (N2, N1).  With dev_pair_row != NULL a spare workgroup of the same launch also runs ncf_group_pairs_rows on the
 * batch (B, n_rows <= 32768; same outputs, same workspace size: ncf_attn_candidates_workspace_bytes).
 * Shapes: N1 in {64, 128}, N2 % 16 == 0, N2 <= 256 (ncf_attn_candidates_supported); else NCF_EUNSUPPORTED (use ncf_linear_forward). */
int ncf_attn_candidates_supported(int K, int N1, int N2);
size_t ncf_attn_candidates_workspace_bytes(int64_t n_rows);
int ncf_attn_candidates(const float* dev_x, int64_t B, int64_t ldx, int K,
                        const float* dev_Wi, int64_t ldw, const float* dev_bi, int N1,
                        const float* dev_Wc, const float* dev_b0, int N2,
                        float* dev_emb, int64_t ldemb, float* dev_pc, int64_t ldpc,
                        const int64_t* dev_pair_row, int64_t n_rows, int pairs_per_wg,
                        int64_t* dev_grp_ptr, int64_t* dev_pair_ids, int64_t* dev_wg_ptr, int32_t* dev_wg_row,
                        void* dev_workspace, size_t workspace_bytes, int32_t* dev_oob_flag, ncf_stream_t stream);

/* The same launch with ItemEmbeddings' weight PRE-PACKED in MFMA operand order (once per weight version: ncf_attn_candidates_pack
 * into ncf_attn_candidates_pack_floats(K, N1) floats, 16-byte aligned).  Every weight load is then one contiguous 1 KB run straight
 * into operand registers: the kernel fits two workgroups per CU, so the grouping workgroup runs beside a row tile instead of in front
 * of one.  Same arguments otherwise (no ldw), bit-identical emb / pc — except for batches of at most 128 row tiles (B <= 2048), where a
 * row tile's K range is cut over up to 8 workgroups (every CU ingests a piece of Wi instead of a few ingesting all of it) and a short
 * second launch adds the pieces in order: deterministic, another summation order.  workspace (16-byte aligned):
 * ncf_attn_candidates_packed_workspace_bytes(B, N1, n_rows or -1 without grouping). */
size_t ncf_attn_candidates_pack_floats(int K, int N1);
size_t ncf_attn_candidates_packed_workspace_bytes(int64_t B, int N1, int64_t n_rows);
int ncf_attn_candidates_pack(const float* dev_Wi, int64_t ldw, int K, int N1, float* dev_packed, ncf_stream_t stream);
int ncf_attn_candidates_packed(const float* dev_x, int64_t B, int64_t ldx, int K,
                               const float* dev_Wi_packed, const float* dev_bi, int N1,
                               const float* dev_Wc, const float* dev_b0, int N2,
                               float* dev_emb, int64_t ldemb, float* dev_pc, int64_t ldpc,
                               const int64_t* dev_pair_row, int64_t n_rows, int pairs_per_wg,
                               int64_t* dev_grp_ptr, int64_t* dev_pair_ids, int64_t* dev_wg_ptr, int32_t* dev_wg_row,
                               void* dev_workspace, size_t workspace_bytes, int32_t* dev_oob_flag, ncf_stream_t stream);

/* out[r, :] = x[r, :] / max(||x[r, :]||_2, 1e-12) — torch.nn.functional.normalize(p=2, dim=1) of the cosine
 * variant, models/attention_ncf.py:167-168. */
int ncf_l2_normalize_rows(const float* dev_x, int64_t ldx, int64_t R, int E, float* dev_out, int64_t ldout,
                          ncf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backward (training step).  The reference trains through torch autograd (train.py:95-110); these
 * are the gradient kernels of K1 / K2.  fp32.
 *   dX = dY . W            : ncf_mlp_forward(n_layers = 1) with the transposed weight (no dedicated entry)
 *   dW = dY^T . X          : ncf_gemm_tn(A = dY (M, N1), B = X (M, N2)) -> (N1, N2); ordered split over M, deterministic
 *   db = column sums of dY : ncf_colsum
 *   ReLU                   : ncf_relu_backward zeroes dY where the forward output was <= 0 (ncf_relu_backward_out: into a new buffer)
 *   embedding rows         : ncf_scatter_add_rows  dst[idx[p], :] += src[p, :]  (float atomics: order-dependent last bits)
 * ------------------------------------------------------------------------------------------------ */
/* out = act(x . W^T + b) for ONE layer with the activation selectable (relu != 0 -> ReLU): the single-layer form of
 * ncf_mlp_forward used by the autograd wrappers, which need every layer's output (util.py:12-17 one Linear at a time). */
int ncf_linear_forward(int dtype, const void* dev_x, int64_t M, int64_t ldx, const void* dev_W, const void* dev_b, int K, int N,
                       int relu, void* dev_out, int64_t ldo, ncf_stream_t stream);
size_t ncf_gemm_tn_workspace_bytes(int64_t M, int N1, int N2);
int ncf_gemm_tn(const float* dev_A, int64_t lda, const float* dev_B, int64_t ldb, int64_t M, int N1, int N2,
                float* dev_out, int64_t ldo, void* dev_workspace, size_t workspace_bytes, ncf_stream_t stream);
size_t ncf_colsum_workspace_bytes(int64_t M, int N);
int ncf_colsum(const float* dev_X, int64_t ldx, int64_t M, int N, float* dev_out, void* dev_workspace, size_t workspace_bytes,
               ncf_stream_t stream);
int ncf_relu_backward(float* dev_dY, int64_t ld_dY, const float* dev_Y, int64_t ld_Y, int64_t M, int N, ncf_stream_t stream);
/* out = scale * dY where Y > 0, else 0; dY is not written (the gradient autograd hands over may be shared).  scale = 1 is the
 * ReLU backward; with Y = dropout(relu(.)) and scale = 1 / (1 - p) it is the backward of the ReLU and the dropout together. */
int ncf_relu_backward_out(const float* dev_dY, int64_t ld_dY, const float* dev_Y, int64_t ld_Y, float* dev_out, int64_t ld_out, int64_t M,
                          int N, float scale, ncf_stream_t stream);
int ncf_scatter_add_rows(const float* dev_src, int64_t ld_src, const int64_t* dev_idx, int64_t B, int E,
                         float* dev_dst, int64_t ld_dst, int64_t rows, int32_t* dev_oob_flag, ncf_stream_t stream);

/* nn.Linear-layout embeddings (models/basic_ncf.py:25-33: the parameter is W [E, U], Linear(onehot(i)) = W[:, i] + b):
 *   ncf_gather_cols       out[p, e]      = W[e, idx[p]] + b[e]          (forward of a training step: no W^T + b table)
 *   ncf_scatter_add_cols  dW[e, idx[p]] += src[p, e]                    (its gradient, straight into the [E, U] layout)
 * An out-of-range idx reads zeros / is dropped and sets the sticky flag. */
int ncf_gather_cols(const float* dev_W, int64_t ldw, const float* dev_bias, const int64_t* dev_idx, int64_t B, int E,
                    int64_t n_cols, float* dev_out, int64_t ldo, int32_t* dev_oob_flag, ncf_stream_t stream);
int ncf_scatter_add_cols(const float* dev_src, int64_t ld_src, const int64_t* dev_idx, int64_t B, int E,
                         float* dev_dst, int64_t ld_dst, int64_t n_cols, int32_t* dev_oob_flag, ncf_stream_t stream);

/* One torch.optim.Adam update (train.py:55; amsgrad and maximize off) of ONE tensor in a single pass:
 *   g' = g + weight_decay * p;  m += (1 - beta1)(g' - m);  v = beta2 v + (1 - beta2) g'^2;
 *   p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)          (step counts from 1) */
int ncf_adam_step(float* dev_p, const float* dev_g, float* dev_m, float* dev_v, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, int64_t step, ncf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Calibration probe — the bf16 MFMA rate the chip SUSTAINS on random operands (it lowers its clock under matrix load), so that
 * bench.py can state a kernel's fraction of it next to the fraction of the 2.5 PFLOP/s datasheet figure.  Not part of scoring.
 * Launches `blocks` 512-thread workgroups; every wave issues iters * 32 v_mfma_f32_16x16x32_bf16 (16 384 flop each) on register
 * operands taken from dev_rnd (64 KiB of random bf16 bit patterns, finite values); with_lds != 0 adds the scoring kernels' LDS
 * operand reads.  dev_sink: blocks * 512 floats (written, never meaningful); dev_clk: blocks * 8 pairs of u64
 * {shader cycles, 100 MHz ticks} per wave.  flop per launch = blocks * 8 * iters * 32 * 16384.
 * ------------------------------------------------------------------------------------------------ */
int ncf_probe_mfma_bf16(const void* dev_rnd, int iters, int blocks, int with_lds, float* dev_sink, unsigned long long* dev_clk,
                        ncf_stream_t stream);
/* Memory-side probes for the standalone gather's roofline: a streaming 16-byte copy of `bytes` (the chip's copy ceiling), and
 * random whole-row READS (rows of row_bytes = 16 .. 1024, a power of two; ids from dev_idx, n of them; `inflight` = 1 / 2 / 4 / 8
 * independent rows per lane group; nothing is written but one word per thread into dev_sink (blocks * 256 words)). */
int ncf_probe_copy(const void* dev_src, void* dev_dst, int64_t bytes, ncf_stream_t stream);
int ncf_probe_gather_read(const void* dev_table, int64_t rows, int64_t ld_bytes, int row_bytes, const int64_t* dev_idx, int64_t n,
                          int inflight, int blocks, uint32_t* dev_sink, ncf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NCF_ABI_H */
