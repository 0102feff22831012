// K4/K5 — LightGCN propagation: CSR-by-destination SpMM as a wavefront segmented reduction (fp32, gfx950).
//
// HBM-bound row gather: per directed edge D*4 bytes of source row + 4 (col) + 4 (coef).  One wave owns one
// SEGMENT (a contiguous edge range of one destination row).  LPR = D/4 lanes (rounded up to a power of two)
// cover one source row with 16-byte loads, so a wave-instruction reads 64/LPR whole contiguous rows; 4 such
// instructions are kept in flight per lane.  The 64/LPR partial sums are combined with cross-lane shuffles.
// Load balance: destination rows longer than the caller's segment length are split into several segments
// (`row_of` maps segment -> destination); their partial sums go to `partial` and a second kernel adds them in
// segment order.  No float atomics anywhere: the result is bitwise reproducible run to run.
// The layer-mean accumulator (gnn_ncf.py:351) is fused into the epilogue: sum[row] += y[row].
#include "ncf_common.h"
#include <math.h>

namespace ncf {

template <int LPR>
__device__ __forceinline__ f32x4 reduce_groups(f32x4 v) {
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        v[0] += __shfl_xor(v[0], off);
        v[1] += __shfl_xor(v[1], off);
        v[2] += __shfl_xor(v[2], off);
        v[3] += __shfl_xor(v[3], off);
    }
    return v;
}

// Training-time dropout of the per-EDGE messages (the reference applies Dropout INSIDE the per-edge Linear, gnn_ncf.py:22-31,
// 91-93: message_e = coef_e * dropout(W(x[src_e])), one mask element per (edge, feature)).  The mask is never stored: element
// (edge id, feature) keeps iff a 16-bit slice of a counter-based hash of (seed, edge id, 16-byte chunk) reaches the threshold —
// the forward pass (CSR by destination) and its transpose (CSR by source, edge ids through `eid`) regenerate the same mask.
struct DropArgs {
    const int32_t* eid;   // edge id of each CSR entry (NULL: the entry's own position)
    uint32_t seed, thr;   // keep iff hash16 >= thr, thr = round(p * 65536)
    float scale;          // 1 / (1 - thr / 65536)
};
__device__ __forceinline__ uint32_t mix32(uint32_t x) {   // lowbias32 (bijective avalanche mix)
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ f32x4 drop4(f32x4 v, uint32_t edge, int chunk, const DropArgs& d) {
    const uint32_t h0 = mix32(edge * 0x9E3779B1U ^ d.seed ^ (uint32_t)chunk * 0x85EBCA77U);
    const uint32_t h1 = mix32(h0 ^ 0x68E31DA4U);
    v[0] = (h0 & 0xFFFFU) >= d.thr ? v[0] * d.scale : 0.f;
    v[1] = (h0 >> 16) >= d.thr ? v[1] * d.scale : 0.f;
    v[2] = (h1 & 0xFFFFU) >= d.thr ? v[2] * d.scale : 0.f;
    v[3] = (h1 >> 16) >= d.thr ? v[3] * d.scale : 0.f;
    return v;
}

template <int LPR, bool DROP = false>
__global__ __launch_bounds__(256) void spmm_seg_kernel(const int64_t* __restrict__ segptr, const int32_t* __restrict__ row_of,
                                                       int64_t n_seg, const int32_t* __restrict__ col,
                                                       const float* __restrict__ coef, const float* __restrict__ z,
                                                       int64_t Nz, int64_t ldz, int chunks, float* __restrict__ y,
                                                       int64_t ldy, float* __restrict__ sum, int64_t ldsum,
                                                       float* __restrict__ partial, DropArgs drop = DropArgs{}) {
    constexpr int EPI = 64 / LPR;  // edges per wave-instruction
    constexpr int UNROLL = 4;
    const int lane = threadIdx.x & 63;
    const int c = lane % LPR;       // 16-byte chunk of the row this lane owns
    const int eg = lane / LPR;      // which of the EPI concurrent edges
    const bool active = c < chunks;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t s = wave0; s < n_seg; s += nwaves) {
        const int64_t beg = segptr[s], end = segptr[s + 1];
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int64_t e0 = beg; e0 < end; e0 += EPI * UNROLL) {
            f32x4 v[UNROLL];
            float w[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                const int64_t e = e0 + u * EPI + eg;
                v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                w[u] = 0.f;
                if (e < end && active) {
                    const int64_t src = col[e];
                    if (src >= 0 && src < Nz) {
                        w[u] = coef ? coef[e] : 1.f;
                        v[u] = *reinterpret_cast<const f32x4*>(z + src * ldz + 4 * c);
                        if (DROP) v[u] = drop4(v[u], (uint32_t)(drop.eid ? drop.eid[e] : e), c, drop);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                acc[0] = fmaf(w[u], v[u][0], acc[0]);
                acc[1] = fmaf(w[u], v[u][1], acc[1]);
                acc[2] = fmaf(w[u], v[u][2], acc[2]);
                acc[3] = fmaf(w[u], v[u][3], acc[3]);
            }
        }
        acc = reduce_groups<LPR>(acc);
        if (eg == 0 && active) {
            const int64_t row = row_of ? row_of[s] : s;
            const bool first = !row_of || s == 0 || row_of[s - 1] != row;
            const bool last = !row_of || s == n_seg - 1 || row_of[s + 1] != row;
            if (first && last) {
                *reinterpret_cast<f32x4*>(y + row * ldy + 4 * c) = acc;
                if (sum) {
                    f32x4* sp = reinterpret_cast<f32x4*>(sum + row * ldsum + 4 * c);
                    *sp = *sp + acc;
                }
            } else if (partial) {
                *reinterpret_cast<f32x4*>(partial + s * (int64_t)(4 * chunks) + 4 * c) = acc;
            }
        }
    }
}

// Adds the partial sums of rows that were split over several segments, in segment order.
template <int LPR>
__global__ __launch_bounds__(256) void spmm_fix_kernel(const int32_t* __restrict__ row_of, int64_t n_seg, int chunks,
                                                       const float* __restrict__ partial, float* __restrict__ y,
                                                       int64_t ldy, float* __restrict__ sum, int64_t ldsum) {
    const int c = threadIdx.x % LPR;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) / LPR;
    for (int64_t s = grp; s < n_seg; s += ngrp) {
        const int32_t row = row_of[s];
        const bool first = s == 0 || row_of[s - 1] != row;
        const bool last = s == n_seg - 1 || row_of[s + 1] != row;
        if (!first || last || c >= chunks) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int64_t t = s; t < n_seg && row_of[t] == row; ++t)
            acc = acc + *reinterpret_cast<const f32x4*>(partial + t * (int64_t)(4 * chunks) + 4 * c);
        *reinterpret_cast<f32x4*>(y + (int64_t)row * ldy + 4 * c) = acc;
        if (sum) {
            f32x4* sp = reinterpret_cast<f32x4*>(sum + (int64_t)row * ldsum + 4 * c);
            *sp = *sp + acc;
        }
    }
}

__global__ void degree_kernel(const int64_t* __restrict__ dst, int64_t E, int64_t N, float* __restrict__ deg, int32_t* oob) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = dst[e];
        if (d >= 0 && d < N) atomicAdd(deg + d, 1.0f);  // exact: counts stay below 2^24
        else if (oob) *oob = 1;
    }
}

__device__ __forceinline__ float inv_sqrt_or_zero(float d) {
    // deg.pow(-0.5) with inf -> 0 (gnn_ncf.py:49-50); torch's pow(-0.5) is 1 / sqrt(x)
    return d > 0.f ? 1.0f / sqrtf(d) : 0.f;
}

__global__ void edge_coef_kernel(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                                 const float* __restrict__ attr, const float* __restrict__ deg, int64_t E, int64_t N,
                                 float* __restrict__ coef) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t s = src[e], d = dst[e];
        float norm = 0.f;
        if (s >= 0 && s < N && d >= 0 && d < N) norm = inv_sqrt_or_zero(deg[s]) * inv_sqrt_or_zero(deg[d]);  // :54/:58/:66
        coef[e] = attr ? attr[e] * norm : norm;  // (weight * norm) * W(x_j), :91 / :93
    }
}

__global__ void scale_rows_kernel(const float* __restrict__ in, int64_t ldin, int64_t N, int D, float divisor,
                                  float* __restrict__ out, int64_t ldout) {
    const int64_t total = N * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / D;
        const int d = (int)(i - n * D);
        out[n * ldout + d] = in[n * ldin + d] / divisor;  // torch.mean = sum / n
    }
}

// LightGAT edge attention (gnn_ncf.py:151-177 with PyG softmax): for destination row r with incoming edges e,
//   out[e] = (attr ? attr[e] : 1) * exp(s[col[e]] - max_r) / (sum_r exp(s[col[.]] - max_r) + 1e-16)
// where s[n] = w_j . x[n] is the SOURCE half of AttNet(cat(x_j, x_i)); the destination half w_i . x_i + b is constant
// inside a softmax group and cancels.  One wave per row, two passes over 4-byte scalars (L2-resident).
__global__ __launch_bounds__(256) void edge_softmax_kernel(const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const float* __restrict__ attr, const float* __restrict__ s,
                                                           int64_t n_rows, int64_t Ns, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave0; r < n_rows; r += nwaves) {
        const int64_t beg = rowptr[r], end = rowptr[r + 1];
        float mx = -INFINITY;
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t c = col[e];
            if (c >= 0 && c < Ns) mx = fmaxf(mx, s[c]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sm = 0.f;
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t c = col[e];
            if (c >= 0 && c < Ns) sm += expf(s[c] - mx);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sm += __shfl_xor(sm, off);
        const float den = sm + 1e-16f;
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t c = col[e];
            const float a = (c >= 0 && c < Ns) ? expf(s[c] - mx) / den : 0.f;
            out[e] = attr ? attr[e] * a : a;  // weight * a_scores * W(x_j), :174 / :176
        }
    }
}


// The same softmax over the SEGMENTS of the SpMM (rows longer than 512 edges are split; a hub item of BASELINE config 4 has 4 M
// in-edges: one wave walking them three times is ~0.2 s, the whole 100 M-edge graph otherwise ~1 ms).  Three passes, every one
// parallel over segments or rows, no atomics, fixed order (bitwise repeatable):
//   1. per segment (one wave):  m_seg = max s[col[e]],  l_seg = sum exp(s[col[e]] - m_seg)
//   2. per row (one wave over the row's segments [seg_first[r], seg_first[r+1])):  M = max m_seg,  L = sum l_seg exp(m_seg - M)
//   3. per segment:  out[e] = (attr ? attr[e] : 1) * exp(s[col[e]] - M_row) / (L_row + 1e-16)
__global__ __launch_bounds__(256) void edge_softmax_seg_stats_kernel(const int64_t* __restrict__ segptr, int64_t n_seg, const int32_t* __restrict__ col,
                                                                     const float* __restrict__ s, int64_t Ns, float* __restrict__ segm,
                                                                     float* __restrict__ segl) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t g = wave0; g < n_seg; g += nwaves) {
        const int64_t beg = segptr[g], end = segptr[g + 1];
        float mx = -INFINITY;
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t c = col[e];
            if (c >= 0 && c < Ns) mx = fmaxf(mx, s[c]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sm = 0.f;
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t c = col[e];
            if (c >= 0 && c < Ns) sm += expf(s[c] - mx);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sm += __shfl_xor(sm, off);
        if (lane == 0) { segm[g] = mx; segl[g] = sm; }
    }
}

__global__ __launch_bounds__(256) void edge_softmax_row_stats_kernel(const int64_t* __restrict__ seg_first, int64_t n_rows, const float* __restrict__ segm,
                                                                     const float* __restrict__ segl, float* __restrict__ rowm, float* __restrict__ rowden) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t r = wave0; r < n_rows; r += nwaves) {
        const int64_t beg = seg_first[r], end = seg_first[r + 1];
        float mx = -INFINITY;
        for (int64_t g = beg + lane; g < end; g += 64) mx = fmaxf(mx, segm[g]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
        float sm = 0.f;
        for (int64_t g = beg + lane; g < end; g += 64) {
            const float m = segm[g];
            if (m != -INFINITY) sm += segl[g] * expf(m - mx);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sm += __shfl_xor(sm, off);
        if (lane == 0) { rowm[r] = mx; rowden[r] = sm + 1e-16f; }
    }
}

__global__ __launch_bounds__(256) void edge_softmax_seg_apply_kernel(const int64_t* __restrict__ segptr, const int32_t* __restrict__ row_of, int64_t n_seg,
                                                                     const int32_t* __restrict__ col, const float* __restrict__ attr,
                                                                     const float* __restrict__ s, int64_t Ns, const float* __restrict__ rowm,
                                                                     const float* __restrict__ rowden, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    for (int64_t g = wave0; g < n_seg; g += nwaves) {
        const int64_t beg = segptr[g], end = segptr[g + 1];
        const int64_t r = row_of[g];
        const float mx = rowm[r], den = rowden[r];
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t c = col[e];
            const float a = (c >= 0 && c < Ns) ? expf(s[c] - mx) / den : 0.f;
            out[e] = attr ? attr[e] * a : a;
        }
    }
}

template <int LPR>
static void launch_spmm(const int64_t* segptr, const int32_t* row_of, int64_t n_seg, const int32_t* col, const float* coef,
                        const float* z, int64_t Nz, int64_t ldz, int chunks, float* y, int64_t ldy, float* sum, int64_t ldsum,
                        float* partial, bool fixup, hipStream_t s, const DropArgs* drop = nullptr) {
    int64_t blocks = (n_seg + 3) / 4;
    if (blocks > 256 * 64) blocks = 256 * 64;
    if (drop)
        hipLaunchKernelGGL((spmm_seg_kernel<LPR, true>), dim3((unsigned)blocks), dim3(256), 0, s, segptr, row_of, n_seg, col, coef, z, Nz,
                           ldz, chunks, y, ldy, sum, ldsum, partial, *drop);
    else
        hipLaunchKernelGGL((spmm_seg_kernel<LPR, false>), dim3((unsigned)blocks), dim3(256), 0, s, segptr, row_of, n_seg, col, coef, z, Nz,
                           ldz, chunks, y, ldy, sum, ldsum, partial, DropArgs{});
    if (row_of && fixup) {
        int64_t fb = (n_seg * LPR + 255) / 256;
        if (fb > 256 * 64) fb = 256 * 64;
        hipLaunchKernelGGL((spmm_fix_kernel<LPR>), dim3((unsigned)fb), dim3(256), 0, s, row_of, n_seg, chunks, partial, y, ldy, sum, ldsum);
    }
}

}  // namespace ncf

using namespace ncf;

static int spmm_impl(const char* who, int dtype, const int64_t* segptr, const int32_t* row_of, int64_t n_seg, const int32_t* col,
                     const float* coef, const void* z, int64_t Nz, int64_t ldz, int D, void* y, int64_t ldy, float* sum, int64_t ldsum,
                     float* partial, int fixup, const DropArgs* drop, ncf_stream_t stream) {
    if (dtype != NCF_F32) return fail(NCF_EUNSUPPORTED, "%s: fp32 only", who);
    if (!segptr || !z || !y || n_seg < 0 || D <= 0) return fail(NCF_EINVAL, "%s: bad argument", who);
    if (D % 4 || D > 256) return fail(NCF_EUNSUPPORTED, "%s: D = %d (need D %% 4 == 0 and D <= 256)", who, D);
    if (ldz % 4 || ldy % 4 || (sum && ldsum % 4) || !aligned16(z) || !aligned16(y) || (sum && !aligned16(sum)) || (partial && !aligned16(partial)))
        return fail(NCF_EINVAL, "%s: rows must be 16-byte aligned (ld %% 4 == 0)", who);
    if (row_of && !partial && fixup) return fail(NCF_EINVAL, "%s: split rows need a partial buffer", who);
    if (n_seg == 0) return NCF_OK;
    // col may be null for an edgeless graph (every segment empty): the kernel dereferences it only inside a segment
    hipStream_t s = (hipStream_t)stream;
    const int chunks = D / 4;
    const float* zf = (const float*)z;
    float* yf = (float*)y;
    if (chunks <= 8) launch_spmm<8>(segptr, row_of, n_seg, col, coef, zf, Nz, ldz, chunks, yf, ldy, sum, ldsum, partial, fixup != 0, s, drop);
    else if (chunks <= 16) launch_spmm<16>(segptr, row_of, n_seg, col, coef, zf, Nz, ldz, chunks, yf, ldy, sum, ldsum, partial, fixup != 0, s, drop);
    else if (chunks <= 32) launch_spmm<32>(segptr, row_of, n_seg, col, coef, zf, Nz, ldz, chunks, yf, ldy, sum, ldsum, partial, fixup != 0, s, drop);
    else launch_spmm<64>(segptr, row_of, n_seg, col, coef, zf, Nz, ldz, chunks, yf, ldy, sum, ldsum, partial, fixup != 0, s, drop);
    return check_launch(who);
}

extern "C" int ncf_spmm_csr(int dtype, const int64_t* segptr, const int32_t* row_of, int64_t n_seg, const int32_t* col,
                            const float* coef, const void* z, int64_t Nz, int64_t ldz, int D, void* y, int64_t ldy,
                            float* sum, int64_t ldsum, float* partial, int fixup, ncf_stream_t stream) {
    return spmm_impl("ncf_spmm_csr", dtype, segptr, row_of, n_seg, col, coef, z, Nz, ldz, D, y, ldy, sum, ldsum, partial, fixup, nullptr, stream);
}

extern "C" int ncf_spmm_csr_dropout(int dtype, const int64_t* segptr, const int32_t* row_of, int64_t n_seg, const int32_t* col,
                                    const float* coef, const void* z, int64_t Nz, int64_t ldz, int D, void* y, int64_t ldy,
                                    float* sum, int64_t ldsum, float* partial, int fixup, const int32_t* edge_id, uint32_t seed,
                                    float p, ncf_stream_t stream) {
    if (!(p >= 0.f && p < 1.f)) return fail(NCF_EINVAL, "ncf_spmm_csr_dropout: p = %g is not in [0, 1)", (double)p);
    DropArgs d;
    d.eid = edge_id;
    d.seed = seed;
    d.thr = (uint32_t)(p * 65536.f + 0.5f);
    if (d.thr > 65535u) d.thr = 65535u;
    d.scale = 65536.f / (float)(65536u - d.thr);
    return spmm_impl("ncf_spmm_csr_dropout", dtype, segptr, row_of, n_seg, col, coef, z, Nz, ldz, D, y, ldy, sum, ldsum, partial, fixup,
                     d.thr ? &d : nullptr, stream);
}

extern "C" int ncf_degree_accumulate(const int64_t* dst, int64_t E, int64_t N, float* deg, int32_t* oob, ncf_stream_t stream) {
    if (E < 0 || N < 0 || (E > 0 && (!dst || !deg))) return fail(NCF_EINVAL, "ncf_degree_accumulate: bad argument");
    if (E == 0) return NCF_OK;
    int64_t blocks = (E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(degree_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dst, E, N, deg, oob);
    return check_launch("ncf_degree_accumulate");
}

extern "C" int ncf_edge_coef(const int64_t* src, const int64_t* dst, const float* attr, const float* deg, int64_t E, int64_t N,
                             float* coef, ncf_stream_t stream) {
    if (E < 0 || N < 0 || (E > 0 && (!src || !dst || !deg || !coef))) return fail(NCF_EINVAL, "ncf_edge_coef: bad argument");
    if (E == 0) return NCF_OK;
    int64_t blocks = (E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(edge_coef_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, dst, attr, deg, E, N, coef);
    return check_launch("ncf_edge_coef");
}

extern "C" int ncf_scale_rows(const float* in, int64_t ldin, int64_t N, int D, float divisor, float* out, int64_t ldout,
                              ncf_stream_t stream) {
    if (N < 0 || D <= 0 || (N > 0 && (!in || !out))) return fail(NCF_EINVAL, "ncf_scale_rows: bad argument");
    if (N == 0) return NCF_OK;
    int64_t blocks = (N * D + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, ldin, N, D, divisor, out, ldout);
    return check_launch("ncf_scale_rows");
}

extern "C" int ncf_edge_softmax_csr(const int64_t* rowptr, const int32_t* col, const float* attr, const float* s, int64_t n_rows,
                                    int64_t Ns, float* out, ncf_stream_t stream) {
    if (n_rows < 0 || Ns < 0) return fail(NCF_EINVAL, "ncf_edge_softmax_csr: bad sizes");
    if (n_rows == 0) return NCF_OK;
    if (!rowptr || !s) return fail(NCF_EINVAL, "ncf_edge_softmax_csr: null pointer");
    int64_t blocks = (n_rows + 3) / 4;
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(edge_softmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, rowptr, col, attr, s, n_rows, Ns, out);
    return check_launch("ncf_edge_softmax_csr");
}

extern "C" size_t ncf_edge_softmax_segmented_workspace_bytes(int64_t n_seg, int64_t n_rows) {
    return (size_t)(2 * (n_seg > 0 ? n_seg : 0) + 2 * (n_rows > 0 ? n_rows : 0)) * sizeof(float);
}

extern "C" int ncf_edge_softmax_segmented(const int64_t* segptr, const int32_t* row_of, int64_t n_seg, const int64_t* seg_first, int64_t n_rows,
                                          const int32_t* col, const float* attr, const float* s, int64_t Ns, float* out, void* workspace,
                                          size_t workspace_bytes, ncf_stream_t stream) {
    if (n_rows < 0 || n_seg < 0 || Ns < 0) return fail(NCF_EINVAL, "ncf_edge_softmax_segmented: bad sizes");
    if (n_rows == 0 || n_seg == 0) return NCF_OK;
    if (!segptr || !row_of || !seg_first || !s || !workspace) return fail(NCF_EINVAL, "ncf_edge_softmax_segmented: null pointer");
    if (workspace_bytes < ncf_edge_softmax_segmented_workspace_bytes(n_seg, n_rows)) return fail(NCF_EWORKSPACE, "ncf_edge_softmax_segmented: workspace too small");
    float* segm = (float*)workspace;
    float* segl = segm + n_seg;
    float* rowm = segl + n_seg;
    float* rowden = rowm + n_rows;
    hipStream_t st = (hipStream_t)stream;
    int64_t bs = (n_seg + 3) / 4, br = (n_rows + 3) / 4;
    if (bs > 256 * 64) bs = 256 * 64;
    if (br > 256 * 64) br = 256 * 64;
    hipLaunchKernelGGL(edge_softmax_seg_stats_kernel, dim3((unsigned)bs), dim3(256), 0, st, segptr, n_seg, col, s, Ns, segm, segl);
    hipLaunchKernelGGL(edge_softmax_row_stats_kernel, dim3((unsigned)br), dim3(256), 0, st, seg_first, n_rows, segm, segl, rowm, rowden);
    hipLaunchKernelGGL(edge_softmax_seg_apply_kernel, dim3((unsigned)bs), dim3(256), 0, st, segptr, row_of, n_seg, col, attr, s, Ns, rowm, rowden, out);
    return check_launch("ncf_edge_softmax_segmented");
}
