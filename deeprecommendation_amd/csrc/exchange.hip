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
    if (hipMemsetAsync(send, 0, sizeof(int64_t) * (size_t)world * (size_t)cap, s) != hipSuccess ||
        hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)world, s) != hipSuccess)
        return check_launch("ncf_bucket_ids (memset)");
    if (B == 0) return NCF_OK;
    const int64_t per_block = kBucketThreads * kBucketPerThread;
    const int64_t blocks = (B + per_block - 1) / per_block;
    hipLaunchKernelGGL(bucket_ids_kernel, dim3((unsigned)blocks), dim3(kBucketThreads), 2 * world * sizeof(int), s, idx, B,
                       rows_per_rank, total_rows, world, cap, send, slot, counts, oob, overflow);
    return check_launch("ncf_bucket_ids");
}
