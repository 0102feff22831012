// Exchange preparation for row-sharded tables (BASELINE config 5; the reference is single-device).
//
// ncf_bucket_ids lists a batch's ids by owner rank into a FIXED-CAPACITY send buffer (cap slots per owner), so that
// the id all-to-all and the row all-to-all that follow both use equal splits and need no size on the host: no
// torch.unique sort, no .tolist() synchronisation per step.  One pass over the ids: a 256-thread workgroup takes
// 2048 ids, ranks them inside the workgroup with returning LDS atomics on `world` counters, reserves its range of
// every owner's bucket with ONE global atomic per (workgroup, owner), then writes local row ids and slots.
// Integer byte work, 24 B per id — nothing here for the matrix cores.
#include "ncf_common.h"

namespace ncf {

constexpr int kBucketThreads = 256;
constexpr int kBucketPerThread = 8;
constexpr int kBucketMaxWorld = 1024;

__global__ __launch_bounds__(kBucketThreads) void bucket_ids_kernel(
    const int64_t* __restrict__ idx, int64_t B, int64_t rows_per_rank, int64_t total_rows, int world, int64_t cap,
    int64_t* __restrict__ send, int64_t* __restrict__ slot, int32_t* __restrict__ counts, int32_t* oob, int32_t* overflow) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    int* hist = smem_i;          // [world] ids of this workgroup per owner
    int* base = smem_i + world;  // [world] start of this workgroup's range in the owner's bucket
    for (int o = threadIdx.x; o < world; o += blockDim.x) hist[o] = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * (kBucketThreads * kBucketPerThread);
    int own[kBucketPerThread], rnk[kBucketPerThread];
    int64_t loc[kBucketPerThread];
#pragma unroll
    for (int j = 0; j < kBucketPerThread; ++j) {
        const int64_t p = first + (int64_t)j * kBucketThreads + threadIdx.x;  // consecutive lanes on consecutive ids
        own[j] = -1;
        rnk[j] = 0;
        loc[j] = 0;
        if (p < B) {
            const int64_t id = idx[p];
            if (id >= 0 && id < total_rows) {
                const int64_t o = id / rows_per_rank;
                own[j] = (int)o;
                loc[j] = id - o * rows_per_rank;
                rnk[j] = atomicAdd(&hist[o], 1);
            } else if (oob) {
                *oob = 1;
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < world; o += blockDim.x) base[o] = hist[o] ? atomicAdd(&counts[o], hist[o]) : 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBucketPerThread; ++j) {
        const int64_t p = first + (int64_t)j * kBucketThreads + threadIdx.x;
        if (p >= B) continue;
        int64_t s = -1;
        if (own[j] >= 0) {
            const int64_t k = (int64_t)base[own[j]] + rnk[j];
            if (k < cap) {
                s = (int64_t)own[j] * cap + k;
                send[s] = loc[j];
            } else if (overflow) {
                *overflow = 1;
            }
        }
        slot[p] = s;
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// De-duplicating form (round 3).  A Zipf batch names its hot rows many times; ncf_bucket_ids ships every repeat.  Here every
// DISTINCT id of the batch takes one slot of its owner's bucket and all its pairs share it — without a sort and without a size on
// the host:
//   pass A  every pair inserts its id into an open-addressing hash set in device memory (atomicCAS on the key word; H >= 2 B slots,
//           linear probing).  The pair whose CAS installs the key is the id's FIRST occurrence: it is ranked inside its workgroup
//           per owner (LDS atomics), the workgroup reserves its ranges with one global atomic per owner, and the winner writes the
//           local row id into the bucket and the bucket slot into the table's value word.  Every pair remembers its table position.
//   pass B  every pair reads its slot from the table (the launch boundary orders it behind pass A); one thread per owner writes the
//           bucket's HEADER: the number of ids in it.
// Bucket layout: (cap + 1) int64 per owner — [count, id_0 .. id_{cap-1}] — so the counts travel WITH the id all-to-all (equal
// splits of cap + 1) and the owner gathers only count rows per bucket (ncf_gather_buckets); padding is never initialised, gathered
// or read.  Integer byte work: 8 B read + 8 B written per pair, ~2 random 8-byte probes per pair in a table that fits L2.
constexpr int64_t kEmptyKey = -1;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {   // splitmix64 finaliser
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
    x ^= x >> 27; x *= 0x94d049bb133111ebull;
    x ^= x >> 31;
    return x;
}

__global__ __launch_bounds__(kBucketThreads) void bucket_dedup_insert_kernel(
    const int64_t* __restrict__ idx, int64_t B, int64_t rows_per_rank, int64_t total_rows, int world, int64_t cap,
    unsigned long long* __restrict__ hkeys, int64_t* __restrict__ hvals, int64_t hmask, int64_t* __restrict__ send,
    int64_t* __restrict__ slot, int32_t* __restrict__ counts, int32_t* oob, int32_t* overflow) {
    extern __shared__ __attribute__((aligned(16))) int smem_i[];
    int* hist = smem_i;          // [world] FIRST occurrences of this workgroup per owner
    int* base = smem_i + world;  // [world] start of this workgroup's range in the owner's bucket
    for (int o = threadIdx.x; o < world; o += blockDim.x) hist[o] = 0;
    __syncthreads();
    const int64_t first = (int64_t)blockIdx.x * (kBucketThreads * kBucketPerThread);
    int own[kBucketPerThread], rnk[kBucketPerThread];
    int64_t loc[kBucketPerThread], pos[kBucketPerThread];
#pragma unroll
    for (int j = 0; j < kBucketPerThread; ++j) {
        const int64_t p = first + (int64_t)j * kBucketThreads + threadIdx.x;  // consecutive lanes on consecutive ids
        own[j] = -1;                                          // -1: not a first occurrence (or not a valid id)
        rnk[j] = 0;
        loc[j] = 0;
        pos[j] = -1;
        if (p < B) {
            const int64_t id = idx[p];
            if (id >= 0 && id < total_rows) {
                int64_t h = (int64_t)(mix64((uint64_t)id) & (uint64_t)hmask);
                for (;;) {                                    // H >= 2 B: an empty slot always exists
                    const unsigned long long old = atomicCAS(&hkeys[h], (unsigned long long)kEmptyKey, (unsigned long long)id);
                    if (old == (unsigned long long)kEmptyKey) {   // installed: this pair is the id's first occurrence
                        const int64_t o = id / rows_per_rank;
                        own[j] = (int)o;
                        loc[j] = id - o * rows_per_rank;
                        rnk[j] = atomicAdd(&hist[o], 1);
                        break;
                    }
                    if (old == (unsigned long long)id) break;
                    h = (h + 1) & hmask;
                }
                pos[j] = h;
            } else if (oob) {
                *oob = 1;
            }
        }
    }
    __syncthreads();
    for (int o = threadIdx.x; o < world; o += blockDim.x) base[o] = hist[o] ? atomicAdd(&counts[o], hist[o]) : 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBucketPerThread; ++j) {
        const int64_t p = first + (int64_t)j * kBucketThreads + threadIdx.x;
        if (p >= B) continue;
        if (own[j] >= 0) {
            const int64_t k = (int64_t)base[own[j]] + rnk[j];
            int64_t s = -1;
            if (k < cap) {
                s = (int64_t)own[j] * cap + k;                // row of the exchanged buffer (no header there)
                send[(int64_t)own[j] * (cap + 1) + 1 + k] = loc[j];
            } else if (overflow) {
                *overflow = 1;
            }
            hvals[pos[j]] = s;
        }
        slot[p] = pos[j];                                     // table position; pass B turns it into the slot
    }
}

__global__ __launch_bounds__(256) void bucket_dedup_resolve_kernel(int64_t B, int world, int64_t cap, const int64_t* __restrict__ hvals,
                                                                   const int32_t* __restrict__ counts, int64_t* __restrict__ send,
                                                                   int64_t* __restrict__ slot) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p < world) {
        const int64_t c = counts[p];
        send[p * (cap + 1)] = c < cap ? c : cap;              // bucket header: ids in this bucket
    }
    if (p < B) {
        const int64_t h = slot[p];
        slot[p] = h >= 0 ? hvals[h] : -1;
    }
}

// The owner's side: rows of the ids it was sent, bucket by bucket, padding skipped.  recv = world buckets of [count, ids...];
// out row (r * cap + k) = table[recv[r][1 + k]] for k < count_r.  LPR lanes per row, 16 bytes per lane.
template <int LPR>
__global__ __launch_bounds__(256) void gather_buckets_kernel(const char* __restrict__ tab, int64_t rows, int64_t ld_bytes,
                                                             const int64_t* __restrict__ recv, int world, int64_t cap, int chunks,
                                                             char* __restrict__ out, int64_t ldo_bytes, int32_t* oob) {
    constexpr int RPW = kWave / LPR;
    const int lane = threadIdx.x & 63, sub = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t total = (int64_t)world * cap;
    for (int64_t base = wave * RPW; base < total; base += nwaves * RPW) {
        const int64_t s = base + lane / LPR;
        if (s >= total) continue;
        const int64_t r = s / cap, k = s - r * cap;
        const int64_t* bucket = recv + r * (cap + 1);
        if (k >= bucket[0]) continue;                          // padding: not gathered, not written
        const int64_t id = bucket[1 + k];
        const bool ok = id >= 0 && id < rows;
        if (!ok && oob && sub == 0) *oob = 1;
        const char* src = tab + id * ld_bytes;
        char* dst = out + s * ldo_bytes;
        for (int c = sub; c < chunks; c += LPR) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(src + (int64_t)c * 16);
            *reinterpret_cast<u32x4*>(dst + (int64_t)c * 16) = v;
        }
    }
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_bucket_ids(const int64_t* idx, int64_t B, int64_t rows_per_rank, int64_t total_rows, int world, int64_t cap,
                              int64_t* send, int64_t* slot, int32_t* counts, int32_t* oob, int32_t* overflow, ncf_stream_t stream) {
    if (B < 0 || rows_per_rank <= 0 || total_rows < 0 || world < 1 || world > kBucketMaxWorld || cap < 1)
        return fail(NCF_EINVAL, "ncf_bucket_ids: bad sizes B=%lld rows_per_rank=%lld world=%d cap=%lld", (long long)B,
                    (long long)rows_per_rank, world, (long long)cap);
    if (rows_per_rank * (int64_t)world < total_rows) return fail(NCF_EINVAL, "ncf_bucket_ids: world * rows_per_rank < total_rows");
    if (!send || !counts || (B > 0 && (!idx || !slot))) return fail(NCF_EINVAL, "ncf_bucket_ids: null pointer");
    hipStream_t s = (hipStream_t)stream;
    // padding slots name local row 0 (a valid row wherever the shard is not empty): the owner gathers them like any other
    fill_u32_async(send, 0u, sizeof(int64_t) * (size_t)world * (size_t)cap, s);
    fill_u32_async(counts, 0u, sizeof(int32_t) * (size_t)world, s);
    if (B == 0) return NCF_OK;
    const int64_t per_block = kBucketThreads * kBucketPerThread;
    const int64_t blocks = (B + per_block - 1) / per_block;
    hipLaunchKernelGGL(bucket_ids_kernel, dim3((unsigned)blocks), dim3(kBucketThreads), 2 * world * sizeof(int), s, idx, B,
                       rows_per_rank, total_rows, world, cap, send, slot, counts, oob, overflow);
    return check_launch("ncf_bucket_ids");
}

extern "C" size_t ncf_bucket_dedup_table_slots(int64_t B) {
    size_t h = 1024;
    while ((int64_t)h < 2 * B) h <<= 1;
    return h;
}

extern "C" int ncf_bucket_ids_dedup(const int64_t* idx, int64_t B, int64_t rows_per_rank, int64_t total_rows, int world, int64_t cap,
                                    int64_t* hkeys, int64_t* hvals, int64_t table_slots, int64_t* send, int64_t* slot, int32_t* counts,
                                    int32_t* oob, int32_t* overflow, ncf_stream_t stream) {
    if (B < 0 || rows_per_rank <= 0 || total_rows < 0 || world < 1 || world > kBucketMaxWorld || cap < 1)
        return fail(NCF_EINVAL, "ncf_bucket_ids_dedup: bad sizes B=%lld rows_per_rank=%lld world=%d cap=%lld", (long long)B,
                    (long long)rows_per_rank, world, (long long)cap);
    if (rows_per_rank * (int64_t)world < total_rows) return fail(NCF_EINVAL, "ncf_bucket_ids_dedup: world * rows_per_rank < total_rows");
    if (!send || !counts || !hkeys || !hvals || (B > 0 && (!idx || !slot))) return fail(NCF_EINVAL, "ncf_bucket_ids_dedup: null pointer");
    if (table_slots < (int64_t)ncf_bucket_dedup_table_slots(B) || (table_slots & (table_slots - 1)))
        return fail(NCF_EWORKSPACE, "ncf_bucket_ids_dedup: the hash table needs a power of two of at least ncf_bucket_dedup_table_slots(B) slots");
    hipStream_t s = (hipStream_t)stream;
    const int64_t used = (int64_t)ncf_bucket_dedup_table_slots(B);   // only this prefix of the table is touched
    fill_u32_async(hkeys, 0xFFFFFFFFu, sizeof(int64_t) * (size_t)used, s);
    fill_u32_async(counts, 0u, sizeof(int32_t) * (size_t)world, s);
    const int64_t per_block = kBucketThreads * kBucketPerThread;
    if (B > 0) {
        const int64_t blocks = (B + per_block - 1) / per_block;
        hipLaunchKernelGGL(bucket_dedup_insert_kernel, dim3((unsigned)blocks), dim3(kBucketThreads), 2 * world * sizeof(int), s, idx, B,
                           rows_per_rank, total_rows, world, cap, (unsigned long long*)hkeys, hvals, used - 1, send, slot, counts, oob, overflow);
    }
    const int64_t n = B > world ? B : world;
    hipLaunchKernelGGL(bucket_dedup_resolve_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, B, world, cap, hvals, counts, send, slot);
    return check_launch("ncf_bucket_ids_dedup");
}

extern "C" int ncf_gather_buckets(int dtype, const void* tab, int64_t rows, int64_t ld, const int64_t* recv, int world, int64_t cap, int E,
                                  void* out, int64_t ldo, int32_t* oob, ncf_stream_t stream) {
    if (dtype != NCF_F32 && dtype != NCF_BF16) return fail(NCF_EINVAL, "ncf_gather_buckets: bad dtype %d", dtype);
    if (world < 1 || cap < 1 || E <= 0 || rows < 0 || ld < E || ldo < E) return fail(NCF_EINVAL, "ncf_gather_buckets: bad sizes");
    if (!recv || !out || (rows > 0 && !tab)) return fail(NCF_EINVAL, "ncf_gather_buckets: null pointer");
    const int elt = dtype == NCF_F32 ? 4 : 2;
    if ((E * elt) % 16 || (ld * elt) % 16 || (ldo * elt) % 16 || !aligned16(tab) || !aligned16(out))
        return fail(NCF_EUNSUPPORTED, "ncf_gather_buckets: rows must be multiples of 16 bytes, 16-byte aligned");
    const int chunks = E * elt / 16;
    hipStream_t s = (hipStream_t)stream;
    const int64_t total = (int64_t)world * cap;
#define LAUNCH_GB(L)                                                                                                       \
    do {                                                                                                                   \
        int64_t blocks = (total + (256 / L) - 1) / (256 / L);                                                              \
        if (blocks > 256 * 64) blocks = 256 * 64;                                                                          \
        hipLaunchKernelGGL(gather_buckets_kernel<L>, dim3((unsigned)blocks), dim3(256), 0, s, (const char*)tab, rows, ld * elt, recv, world, \
                           cap, chunks, (char*)out, ldo * elt, oob);                                                      \
    } while (0)
    if (chunks >= 16) LAUNCH_GB(16);
    else if (chunks >= 8) LAUNCH_GB(8);
    else if (chunks >= 4) LAUNCH_GB(4);
    else LAUNCH_GB(1);
#undef LAUNCH_GB
    return check_launch("ncf_gather_buckets");
}
