// Dense user_matrix -> shared-row CSR on the stream (round 3), gfx950.
//
// The reference hands AttentionNCF.forward a DENSE (B, I) user_matrix (content_providers/dynamic_profiles_provider.py:55-71 builds
// it; models/attention_ncf.py:158 keeps the entries that are != 0), with a user's row repeated for each of their samples
// (neural_collaborative_filtering/datasets/dynamic_datasets.py:24-40) — the web backend repeats ONE row for every candidate
// (webapp/backend.py:78-121).  The attention kernels want CSR, and the grouped kernels want equal rows to SHARE one CSR row.
// Round 2 did that with torch ops and two host reads (a float64 projection of the matrix, torch.unique, int(), bool()).
// Here: four small kernels and one cumulative sum (in place), no size ever read by the host.
//   0. row_rep_clear       the hash table's empty state (a kernel, not hipMemsetAsync: ncf_common.h)
//   1. dense_row_scan      one wave per row: number of non-zero entries + a 64-bit hash of the row's (column, value) set, and (same
//                          launch) hash -> slot of an open-addressing table (atomicCAS on the key); payload = atomicMin of the pair
//                          index: the SMALLEST pair index with that hash is the candidate representative (deterministic)
//   3. row_rep_verify      one wave per row: the row is compared with its candidate's, element by element; equal -> it shares the
//                          candidate's CSR row and lists no entries itself; different (a hash collision) -> it represents itself.
//                          Correctness never rests on the hash.
//      (cumulative sum of the kept counts -> rowptr: torch.cumsum on the stream)
//   4. dense_row_compact   one wave per representative row: its (col, val) entries in column order at rowptr[b]
// Result: a CSR with B rows (rows of non-representatives are empty) and pair_row[b] = the row pair b uses.  col / val are sized by
// the caller for the worst case (B * I entries); only rowptr[B] of them are written.  Byte work, three reads of the matrix.
#include "ncf_common.h"

namespace ncf {

__device__ __forceinline__ uint64_t dmix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t lo = __shfl_xor((uint32_t)v, off), hi = __shfl_xor((uint32_t)(v >> 32), off);
        v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

__global__ __launch_bounds__(256) void dense_row_scan_kernel(const float* __restrict__ um, int64_t ld, int64_t B, int64_t I,
                                                             int32_t* __restrict__ cnt, unsigned long long* __restrict__ hkeys,
                                                             int* __restrict__ hrep, int64_t hmask, int64_t* __restrict__ pos) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* row = um + b * ld;
    int n = 0;
    uint64_t h = 0;
    constexpr int U = 8;                                      // U loads in flight per lane: a plain loop over a run-time I is a chain of round trips
    for (int64_t c0 = lane; c0 < I; c0 += 64 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = c0 + 64 * u < I ? row[c0 + 64 * u] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (v[u] != 0.f) {                                // attention_ncf.py:158 — an entry that is exactly 0 (or -0) is unrated
                ++n;
                h += dmix64(((uint64_t)(c0 + 64 * u) << 32) | (uint64_t)__float_as_uint(v[u]));   // commutative: a set hash of (column, value)
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off);
    h = wave_sum_u64(h);
    if (lane == 0) {
        cnt[b] = n;
        if (hkeys) {                                          // step 2 in the same launch: the row's hash into the table (cleared by the launch before)
            const unsigned long long key = h == ~0ull ? 0ull : h;     // ~0 is the table's empty key
            int64_t slot = (int64_t)(dmix64(key) & (uint64_t)hmask);
            for (;;) {
                const unsigned long long old = atomicCAS(&hkeys[slot], ~0ull, key);
                if (old == ~0ull || old == key) break;
                slot = (slot + 1) & hmask;
            }
            atomicMin(&hrep[slot], (int)b);
            pos[b] = slot;
        }
    }
}

__global__ __launch_bounds__(256) void row_rep_clear_kernel(unsigned long long* __restrict__ hkeys, int* __restrict__ hrep, int64_t H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H) { hkeys[i] = ~0ull; hrep[i] = 0x7F7F7F7F; }
}

__global__ __launch_bounds__(256) void row_rep_verify_kernel(const float* __restrict__ um, int64_t ld, int64_t B, int64_t I,
                                                             const int* __restrict__ hrep, const int64_t* __restrict__ pos,
                                                             int32_t* __restrict__ cnt, int64_t* __restrict__ pair_row,
                                                             int64_t* __restrict__ keep_cnt) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    if (b == 0 && lane == 0) keep_cnt[-1] = 0;                 // keep_cnt = rowptr + 1: the cumulative sum runs in place, rowptr[0] = 0
    int64_t rep = hrep[pos[b]];
    if (rep != b) {
        const float* r0 = um + b * ld;
        const float* r1 = um + rep * ld;
        bool same = true;
        constexpr int U = 4;
        for (int64_t c0 = lane; c0 < I; c0 += 64 * U) {
            float x[U], y[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool in = c0 + 64 * u < I;
                x[u] = in ? r0[c0 + 64 * u] : 0.f;
                y[u] = in ? r1[c0 + 64 * u] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) same &= (x[u] == y[u]);    // -0 == 0; a NaN is never equal: such a row represents itself
        }
        if (!__all(same)) rep = b;                             // a hash collision between different rows: no sharing for this one
    }
    if (lane == 0) {
        pair_row[b] = rep;
        keep_cnt[b] = rep == b ? cnt[b] : 0;                   // only representatives list entries
    }
}

__global__ __launch_bounds__(256) void rows_self_kernel(int64_t B, const int32_t* __restrict__ cnt, int64_t* __restrict__ pair_row,
                                                        int64_t* __restrict__ keep_cnt) {   // no sharing: every row represents itself
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) keep_cnt[-1] = 0;
    if (b < B) { pair_row[b] = b; keep_cnt[b] = cnt[b]; }
}

__global__ __launch_bounds__(256) void dense_row_compact_kernel(const float* __restrict__ um, int64_t ld, int64_t B, int64_t I,
                                                                const int64_t* __restrict__ rowptr, const int64_t* __restrict__ pair_row,
                                                                int32_t* __restrict__ col, float* __restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B || pair_row[b] != b) return;                   // wave-uniform
    const float* row = um + b * ld;
    int64_t w = rowptr[b];
    constexpr int U = 8;
    for (int64_t cb = 0; cb < I; cb += 64 * U) {               // U blocks of 64 columns loaded at once, then ranked in column order
        float vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) vv[u] = cb + 64 * u + lane < I ? row[cb + 64 * u + lane] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {                          // a ballot ranks the wave's non-zero entries
            const int64_t c = cb + 64 * u + lane;
            const float v = vv[u];
            const bool nz = v != 0.f;
            const unsigned long long m = __ballot(nz);
            if (nz) {
                const int64_t k = w + __popcll(m & ((1ull << lane) - 1ull));
                col[k] = (int32_t)c;
                val[k] = v;
            }
            w += __popcll(m);
        }
    }
}

}  // namespace ncf

using namespace ncf;

static size_t ncf_dense_csr_table_slots(int64_t B) {       // slots of the row-hash table: a power of two >= 2 B
    size_t h = 1024;
    while ((int64_t)h < 2 * B) h <<= 1;
    return h;
}

/* workspace: cnt int32[B] | (unused) u64[B] | pos i64[B] | hkeys u64[H] | hrep int32[H]   (H = ncf_dense_csr_table_slots(B)) */
extern "C" size_t ncf_dense_csr_workspace_bytes(int64_t B) {
    const size_t b = (size_t)(B > 0 ? B : 0), h = ncf_dense_csr_table_slots(B);
    return ((b * 4 + 15) / 16) * 16 + b * 8 + b * 8 + h * 8 + h * 4;
}

namespace {
struct DenseWs {
    int32_t* cnt; uint64_t* hash; int64_t* pos; unsigned long long* hkeys; int* hrep; size_t H;
};
DenseWs carve(void* ws, int64_t B) {
    DenseWs w;
    char* p = (char*)ws;
    const size_t b = (size_t)B;
    w.H = ncf_dense_csr_table_slots(B);
    w.cnt = (int32_t*)p; p += ((b * 4 + 15) / 16) * 16;
    w.hash = (uint64_t*)p; p += b * 8;
    w.pos = (int64_t*)p; p += b * 8;
    w.hkeys = (unsigned long long*)p; p += w.H * 8;
    w.hrep = (int*)p;
    return w;
}
}  // namespace

/* Phase 1 (everything before the cumulative sum): pair_row (B) and rowptr (B + 1) = [0, entries row 0 will list, entries row 1 will
 * list, ...]: the caller's inclusive cumulative sum over rowptr, IN PLACE, turns it into the CSR's rowptr. */
extern "C" int ncf_dense_csr_rows(const float* um, int64_t ld, int64_t B, int64_t I, int share_rows, int64_t* pair_row, int64_t* rowptr,
                                  void* workspace, size_t workspace_bytes, ncf_stream_t stream) {
    int64_t* const keep_cnt = rowptr ? rowptr + 1 : nullptr;
    if (B < 0 || I < 0 || ld < I) return fail(NCF_EINVAL, "ncf_dense_csr_rows: bad sizes");
    if (B == 0) {
        if (rowptr) fill_u32_async(rowptr, 0u, sizeof(int64_t), (hipStream_t)stream);
        return check_launch("ncf_dense_csr_rows (empty)");
    }
    if (B >= (1ll << 31) || I >= (1ll << 31)) return fail(NCF_EUNSUPPORTED, "ncf_dense_csr_rows: more than 2^31 rows or columns");
    if (!um || !pair_row || !keep_cnt || !workspace) return fail(NCF_EINVAL, "ncf_dense_csr_rows: null pointer");
    if (workspace_bytes < ncf_dense_csr_workspace_bytes(B) || !aligned16(workspace))
        return fail(NCF_EWORKSPACE, "ncf_dense_csr_rows: workspace too small (ncf_dense_csr_workspace_bytes) or misaligned");
    hipStream_t s = (hipStream_t)stream;
    DenseWs w = carve(workspace, B);
    const unsigned wblocks = (unsigned)((B + 3) / 4);
    if (share_rows) {
        hipLaunchKernelGGL(row_rep_clear_kernel, dim3((unsigned)((w.H + 255) / 256)), dim3(256), 0, s, w.hkeys, w.hrep, (int64_t)w.H);   // a kernel, not memsets: ncf_common.h
        hipLaunchKernelGGL(dense_row_scan_kernel, dim3(wblocks), dim3(256), 0, s, um, ld, B, I, w.cnt, w.hkeys, w.hrep, (int64_t)w.H - 1, w.pos);
        hipLaunchKernelGGL(row_rep_verify_kernel, dim3(wblocks), dim3(256), 0, s, um, ld, B, I, w.hrep, w.pos, w.cnt, pair_row, keep_cnt);
    } else {
        hipLaunchKernelGGL(dense_row_scan_kernel, dim3(wblocks), dim3(256), 0, s, um, ld, B, I, w.cnt, (unsigned long long*)nullptr, (int*)nullptr, (int64_t)0, (int64_t*)nullptr);
        hipLaunchKernelGGL(rows_self_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, B, w.cnt, pair_row, keep_cnt);
    }
    return check_launch("ncf_dense_csr_rows");
}

/* Phase 2 (after rowptr = [0, cumsum(keep_cnt)]): the representatives' entries. */
extern "C" int ncf_dense_csr_fill(const float* um, int64_t ld, int64_t B, int64_t I, const int64_t* rowptr, const int64_t* pair_row,
                                  int32_t* col, float* val, ncf_stream_t stream) {
    if (B < 0 || I < 0 || ld < I) return fail(NCF_EINVAL, "ncf_dense_csr_fill: bad sizes");
    if (B == 0 || I == 0) return NCF_OK;
    if (!um || !rowptr || !pair_row || !col || !val) return fail(NCF_EINVAL, "ncf_dense_csr_fill: null pointer");
    hipLaunchKernelGGL(dense_row_compact_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, um, ld, B, I, rowptr, pair_row, col, val);
    return check_launch("ncf_dense_csr_fill");
}
