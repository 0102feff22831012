// Calibration probe (round 3): the bf16 MFMA rate this chip SUSTAINS on random operands, measured in the same process and minutes
// as the kernels that are priced against it.  gfx950 lowers its clock under MFMA load (1.5-2.0 GHz on random data against 2.4 GHz
// on zeros: MI355X_MICROARCH.md, "DVFS give-back"), so "2.5 PFLOP/s" is a datasheet figure no kernel can reach on real data;
// bench.py reports a kernel's fraction of the spec figure AND of this probe's figure.
// The kernel: 2 waves per SIMD (512 threads, one workgroup per CU), every wave issues v_mfma_f32_16x16x32_bf16 back to back on 8
// independent accumulators with operands held in registers (loaded once from a caller-supplied random buffer) — no memory traffic,
// no LDS: the matrix pipe alone.  `with_lds` != 0 adds one 1-KiB ds_read_b128 per 4 MFMAs (the scoring kernel's operand traffic) to
// see what the LDS reads cost in clock.  Clock = delta s_memtime / delta s_memrealtime (100 MHz), per wave, in the out buffer.
#include "ncf_common.h"

namespace ncf {

typedef short s16x8 __attribute__((ext_vector_type(8)));

template <bool LDS>
__global__ __launch_bounds__(512, 2) void probe_mfma_bf16_kernel(const uint32_t* __restrict__ rnd, int iters, float* __restrict__ sink,
                                                                 unsigned long long* __restrict__ clk) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[LDS ? 16384 : 4];
    const int lane = threadIdx.x & 63;
    union { u32x4 u; s16x8 s; } a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i].u = *reinterpret_cast<const u32x4*>(rnd + ((threadIdx.x * 8 + i) & 4095) * 4);
        b[i].u = *reinterpret_cast<const u32x4*>(rnd + ((threadIdx.x * 8 + 4 + i) & 4095) * 4);
    }
    if (LDS) {
        for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = rnd[i & 16383];
        __syncthreads();
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (LDS) {                                     // a fresh B fragment from LDS for this group of 8 MFMAs (1 KiB per 4 of them, twice)
                b[r].u = *reinterpret_cast<const u32x4*>(lds + ((it * 4 + r) & 15) * 1024 + lane * 4);
                b[(r + 2) & 3].u = *reinterpret_cast<const u32x4*>(lds + ((it * 4 + r + 7) & 15) * 1024 + lane * 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + r) & 3].s, b[i & 3].s, acc[i], 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 512 + threadIdx.x] = s;              // keeps the chains alive; never read
    if (lane == 0) {
        const int wv = blockIdx.x * 8 + (threadIdx.x >> 6);
        clk[2 * wv] = t1 - t0;
        clk[2 * wv + 1] = r1 - r0;
    }
}

// ---- memory-side probes for the standalone gather (K1): what this chip delivers, now, for (a) a streaming 16-byte copy of the same
// volume and (b) random whole-row reads of `row_bytes` (256 B at cfg 2) with U rows in flight per lane group — the two ceilings the
// gather kernel's 8 TB/s-relative fraction is read against (the guide's 5.5-5.8 TB/s random-row figure is for rows of 1 KB and more).
__global__ __launch_bounds__(256) void probe_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t n16) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) __builtin_nontemporal_store(src[i], dst + i);
}

template <int U>
__global__ __launch_bounds__(256) void probe_gather_read_kernel(const char* __restrict__ tab, int64_t ld_bytes, int lpr, const int64_t* __restrict__ idx,
                                                                int64_t n, uint32_t* __restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int rpw = 64 / lpr, sub = lane % lpr;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * 4;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int64_t base = wave * rpw * U; base < n; base += nwaves * rpw * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {                       // U independent (id -> row) chains in flight per lane
            const int64_t p = base + u * rpw + lane / lpr;
            const int64_t r = idx[p < n ? p : n - 1];
            v[u] = *reinterpret_cast<const u32x4*>(tab + r * ld_bytes + (int64_t)sub * 16);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    sink[(int64_t)blockIdx.x * 256 + threadIdx.x] = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_probe_copy(const void* src, void* dst, int64_t bytes, ncf_stream_t stream) {
    if (!src || !dst || bytes < 16 || bytes % 16 || !aligned16(src) || !aligned16(dst)) return fail(NCF_EINVAL, "ncf_probe_copy: bad argument");
    const int64_t n16 = bytes / 16;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(probe_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, (u32x4*)dst, n16);
    return check_launch("ncf_probe_copy");
}

extern "C" int ncf_probe_gather_read(const void* table, int64_t rows, int64_t ld_bytes, int row_bytes, const int64_t* idx, int64_t n,
                                     int inflight, int blocks, uint32_t* sink, ncf_stream_t stream) {
    if (!table || !idx || !sink || rows < 1 || n < 1 || blocks < 1 || blocks > 65536) return fail(NCF_EINVAL, "ncf_probe_gather_read: bad argument");
    if (row_bytes < 16 || row_bytes > 1024 || (row_bytes & (row_bytes - 1)) || ld_bytes < row_bytes || ld_bytes % 16 || !aligned16(table))
        return fail(NCF_EUNSUPPORTED, "ncf_probe_gather_read: rows of 16 .. 1024 bytes (a power of two), 16-byte aligned");
    const int lpr = row_bytes / 16;
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_PG(UU) hipLaunchKernelGGL(probe_gather_read_kernel<UU>, dim3((unsigned)blocks), dim3(256), 0, s, (const char*)table, ld_bytes, lpr, idx, n, sink)
    if (inflight <= 1) LAUNCH_PG(1);
    else if (inflight <= 2) LAUNCH_PG(2);
    else if (inflight <= 4) LAUNCH_PG(4);
    else LAUNCH_PG(8);
#undef LAUNCH_PG
    return check_launch("ncf_probe_gather_read");
}


/* flop of one launch = blocks * 8 waves * iters * 32 MFMAs * 16384 */
extern "C" int ncf_probe_mfma_bf16(const void* rnd64k, int iters, int blocks, int with_lds, float* sink, unsigned long long* clk,
                                   ncf_stream_t stream) {
    if (!rnd64k || !sink || !clk || iters < 1 || blocks < 1 || blocks > 4096) return fail(NCF_EINVAL, "ncf_probe_mfma_bf16: bad argument");
    if (!aligned16(rnd64k)) return fail(NCF_EINVAL, "ncf_probe_mfma_bf16: the random buffer must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (with_lds)
        hipLaunchKernelGGL(probe_mfma_bf16_kernel<true>, dim3(blocks), dim3(512), 0, s, (const uint32_t*)rnd64k, iters, sink, clk);
    else
        hipLaunchKernelGGL(probe_mfma_bf16_kernel<false>, dim3(blocks), dim3(512), 0, s, (const uint32_t*)rnd64k, iters, sink, clk);
    return check_launch("ncf_probe_mfma_bf16");
}
