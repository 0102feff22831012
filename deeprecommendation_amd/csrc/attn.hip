// K3 — AttentionNCF item-item attention (models/attention_ncf.py:154-213, eval mode), fp32, gfx950.
//
// One wave per (user, candidate) pair b; its rated set is the CSR row [rowptr[b], rowptr[b+1]).
//   phase 1  score_e = b1 + sum_a w1[a] * relu(pc[b,a] + pr[col_e,a])      (AttentionNet with the first Linear
//            split at the concat boundary; the per-pair (2·IE -> A) GEMM of the reference collapses to an add +
//            ReLU + dot over A once cand·Wc^T and rated·Wr^T are precomputed per item)
//            LPA = A/4 lanes read one projected row pr[col_e,:] as 16-byte pieces (coalesced whole rows),
//            64/LPA entries per wave-instruction; a shuffle tree finishes each dot.
//   phase 2  masked row softmax over the entries only (== softmax over a (B,I) row filled with -inf, :182-209;
//            an empty row gives all zeros, the nan_to_num case)
//   phase 3  out[b,:] = bias + sum_e (w_e * val_e) * feat[col_e,:]           (:212-213), entries in col order
// The tables (pr, feat) are small (catalogue x 64..128 floats) and stay L2 / Infinity-Cache resident; the
// kernel is bound by gathered cache bandwidth, not by arithmetic.
#include "ncf_common.h"
#include <math.h>

#ifndef ATT_UNROLL
#define ATT_UNROLL 8   // measured (cfg 3, A/B): 2 -> 141.7 us, 4 -> 138.9, 8 -> 133.9, 16 -> 256 (registers)
#endif

namespace ncf {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// MODE 0: MLP (relu + w1 dot), 1: linear (A == 1, pc + pr), 2: cosine (dot of normalised rows)
template <int MODE>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ pc, int64_t ldpc, const float* __restrict__ pr,
                                                   int64_t ldpr, int A, const float* __restrict__ w1, float b1,
                                                   const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const float* __restrict__ val, int64_t B, int64_t I,
                                                   const float* __restrict__ feat, int64_t ldfeat, int Fdim,
                                                   const float* __restrict__ out_bias, float* __restrict__ out,
                                                   int64_t ldout, float* __restrict__ wts) {
    const int lane = threadIdx.x & 63;
    // XCD-aware block order: hardware deals blocks round-robin over the 8 XCDs (b and b+8 share one), so logical block
    // L = (b % 8) * ceil(n/8) + b / 8 (bijective form) puts CONSECUTIVE pair groups on the same XCD.  Batches are
    // usually grouped by user (many candidates against one rated set): the set's projected rows are then served by
    // that XCD's L2 instead of the Infinity Cache.  Placement only changes speed, never results.
    const unsigned nblk = gridDim.x, q8 = nblk / 8, r8 = nblk % 8, xcd = blockIdx.x % 8;
    const unsigned lblk = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + blockIdx.x / 8;
    const int64_t b = (int64_t)lblk * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;  // wave-uniform
    const int64_t beg = rowptr[b], end = rowptr[b + 1];
    const int64_t nnz = end - beg;

    // ---------------- phase 1: raw scores -> wts[beg..end) ----------------
    if (MODE == 1) {
        const float c0 = pc[b * ldpc];
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t i = col[e];
            wts[e] = (i >= 0 && i < I) ? c0 + pr[i * ldpr] : -INFINITY;
        }
    } else {
        const bool vec = (A % 4 == 0) && (ldpr % 4 == 0) && (ldpc % 4 == 0) && A <= 256;
        if (vec) {
            const int chunks = A / 4;
            int LPA = 8;
            while (LPA < chunks) LPA <<= 1;  // 8, 16, 32, 64 lanes per entry
            const int c = lane % LPA, eg = lane / LPA, EPI = 64 / LPA;
            const bool active = c < chunks;
            f32x4 pcv = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
            if (active) {
                pcv = *reinterpret_cast<const f32x4*>(pc + b * ldpc + 4 * c);
                if (MODE == 0) wv = *reinterpret_cast<const f32x4*>(w1 + 4 * c);
            }
            // ATT_UNROLL entries per lane group are in flight at once: with one dependent (col -> row) load per
            // iteration the loop ran at one L2/Infinity-Cache round trip per 64/LPA entries (latency-bound, 178 us at
            // cfg 3); the loads of the unrolled steps are independent and overlap.
            constexpr int U = ATT_UNROLL;
            for (int64_t e0 = beg; e0 < end; e0 += (int64_t)EPI * U) {
                f32x4 r[U];
                bool ok[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t e = e0 + u * EPI + eg;
                    ok[u] = false;
                    r[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (e < end && active) {
                        const int64_t i = col[e];
                        ok[u] = (i >= 0 && i < I);
                        if (ok[u]) r[u] = *reinterpret_cast<const f32x4*>(pr + i * ldpr + 4 * c);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t e = e0 + u * EPI + eg;
                    float part = 0.f;
                    if (MODE == 0) {
                        part = fmaf(wv[0], fmaxf(pcv[0] + r[u][0], 0.f), part);
                        part = fmaf(wv[1], fmaxf(pcv[1] + r[u][1], 0.f), part);
                        part = fmaf(wv[2], fmaxf(pcv[2] + r[u][2], 0.f), part);
                        part = fmaf(wv[3], fmaxf(pcv[3] + r[u][3], 0.f), part);
                    } else {
                        part = fmaf(pcv[0], r[u][0], part);
                        part = fmaf(pcv[1], r[u][1], part);
                        part = fmaf(pcv[2], r[u][2], part);
                        part = fmaf(pcv[3], r[u][3], part);
                    }
                    for (int off = 1; off < LPA; off <<= 1) part += __shfl_xor(part, off);
                    if (e < end && c == 0) wts[e] = ok[u] ? part + (MODE == 0 ? b1 : 0.f) : -INFINITY;
                }
            }
        } else {
            // generic: one lane per entry, sequential over A
            for (int64_t e = beg + lane; e < end; e += 64) {
                const int64_t i = col[e];
                float s = -INFINITY;
                if (i >= 0 && i < I) {
                    float acc = 0.f;
                    for (int a = 0; a < A; ++a) {
                        const float r = pr[i * ldpr + a], cv = pc[b * ldpc + a];
                        acc = MODE == 0 ? fmaf(w1[a], fmaxf(cv + r, 0.f), acc) : fmaf(cv, r, acc);
                    }
                    s = acc + (MODE == 0 ? b1 : 0.f);
                }
                wts[e] = s;
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // other lanes of this wave read what this lane stored
    __builtin_amdgcn_wave_barrier();

    // ---------------- phase 2: softmax over the row's entries ----------------
    float mx = -INFINITY;
    for (int64_t e = beg + lane; e < end; e += 64) mx = fmaxf(mx, wts[e]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int64_t e = beg + lane; e < end; e += 64) {
        const float ex = (mx == -INFINITY) ? 0.f : expf(wts[e] - mx);
        wts[e] = ex;
        sm += ex;
    }
    sm = wave_sum(sm);
    const float inv = sm > 0.f ? 1.0f / sm : 0.f;  // empty / all -inf row -> zeros (nan_to_num, :209)
    for (int64_t e = beg + lane; e < end; e += 64) wts[e] = wts[e] * inv;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // other lanes of this wave read what this lane stored
    __builtin_amdgcn_wave_barrier();

    // ---------------- phase 3: out[b,:] = bias + sum_e (w_e * val_e) * feat[col_e,:] ----------------
    const bool fvec = (Fdim % 4 == 0) && (ldfeat % 4 == 0) && (ldout % 4 == 0) && Fdim <= 256;
    if (fvec) {
        const int chunks = Fdim / 4;
        int LPF = 8;
        while (LPF < chunks) LPF <<= 1;
        const int c = lane % LPF, eg = lane / LPF, EPI = 64 / LPF;
        const bool active = c < chunks;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        constexpr int U = ATT_UNROLL;
        for (int64_t e0 = beg; e0 < end; e0 += (int64_t)EPI * U) {
            f32x4 f[U];
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t e = e0 + u * EPI + eg;
                f[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                av[u] = 0.f;
                if (e < end && active) {
                    const int64_t i = col[e];
                    if (i >= 0 && i < I) {
                        av[u] = wts[e] * val[e];  // attended_user_matrix entry (:212)
                        f[u] = *reinterpret_cast<const f32x4*>(feat + i * ldfeat + 4 * c);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc[0] = fmaf(av[u], f[u][0], acc[0]);
                acc[1] = fmaf(av[u], f[u][1], acc[1]);
                acc[2] = fmaf(av[u], f[u][2], acc[2]);
                acc[3] = fmaf(av[u], f[u][3], acc[3]);
            }
        }
        for (int off = LPF; off < 64; off <<= 1) {
            acc[0] += __shfl_xor(acc[0], off);
            acc[1] += __shfl_xor(acc[1], off);
            acc[2] += __shfl_xor(acc[2], off);
            acc[3] += __shfl_xor(acc[3], off);
        }
        if (eg == 0 && active) {
            if (out_bias) acc = acc + *reinterpret_cast<const f32x4*>(out_bias + 4 * c);
            *reinterpret_cast<f32x4*>(out + b * ldout + 4 * c) = acc;
        }
    } else {
        // generic: lanes across features, sequential over entries (feature rows read coalesced)
        for (int f0 = 0; f0 < Fdim; f0 += 64) {
            const int f = f0 + lane;
            float acc = 0.f;
            if (f < Fdim) {
                for (int64_t e = beg; e < end; ++e) {
                    const int64_t i = col[e];
                    if (i >= 0 && i < I) acc = fmaf(wts[e] * val[e], feat[i * ldfeat + f], acc);
                }
                out[b * ldout + f] = acc + (out_bias ? out_bias[f] : 0.f);
            }
        }
    }
    (void)nnz;
}

// out[r,:] = x[r,:] / max(||x[r,:]||_2, 1e-12)   (torch.nn.functional.normalize(p=2, dim=1), attention_ncf.py:167-168)
__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(const float* __restrict__ x, int64_t ldx, int64_t R, int E,
                                                                float* __restrict__ out, int64_t ldo) {
    const int sub = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t Rpad = (R + 3) & ~int64_t(3);
    for (int64_t r = grp; r < Rpad; r += ngrp) {
        float ss = 0.f;
        if (r < R)
            for (int e = sub; e < E; e += 16) ss = fmaf(x[r * ldx + e], x[r * ldx + e], ss);
        ss += __shfl_xor(ss, 8);
        ss += __shfl_xor(ss, 4);
        ss += __shfl_xor(ss, 2);
        ss += __shfl_xor(ss, 1);
        if (r < R) {
            const float d = fmaxf(sqrtf(ss), 1e-12f);
            for (int e = sub; e < E; e += 16) out[r * ldo + e] = x[r * ldx + e] / d;
        }
    }
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_l2_normalize_rows(const float* x, int64_t ldx, int64_t R, int E, float* out, int64_t ldo, ncf_stream_t stream) {
    if (R == 0) return NCF_OK;
    if (R < 0 || E <= 0 || !x || !out || ldx < E || ldo < E) return fail(NCF_EINVAL, "ncf_l2_normalize_rows: bad argument");
    int64_t blocks = (R + 15) / 16;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, R, E, out, ldo);
    return check_launch("ncf_l2_normalize_rows");
}

extern "C" int ncf_attn_forward(int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A,
                                const float* w1, float b1, const int64_t* rowptr, const int32_t* col, const float* val,
                                int64_t B, int64_t I, const float* feat, int64_t ldfeat, int Fdim, const float* out_bias,
                                float* out, int64_t ldout, float* wts, ncf_stream_t stream) {
    if (mode < 0 || mode > 2) return fail(NCF_EINVAL, "ncf_attn_forward: bad mode %d", mode);
    if (B < 0 || I < 0 || A <= 0 || Fdim <= 0) return fail(NCF_EINVAL, "ncf_attn_forward: bad sizes");
    if (B == 0) return NCF_OK;
    if (!pc || !pr || !rowptr || !feat || !out || !wts) return fail(NCF_EINVAL, "ncf_attn_forward: null pointer");
    if (mode == NCF_ATT_MLP && !w1) return fail(NCF_EINVAL, "ncf_attn_forward: w1 is null");
    if (mode == NCF_ATT_LINEAR && A != 1) return fail(NCF_EINVAL, "ncf_attn_forward: linear mode needs A == 1");
    if (ldpc < A || ldpr < A || ldfeat < Fdim || ldout < Fdim) return fail(NCF_EINVAL, "ncf_attn_forward: leading dimension smaller than row");
    if ((A % 4 == 0 && (!aligned16(pc) || !aligned16(pr) || (w1 && !aligned16(w1)))) ||
        (Fdim % 4 == 0 && (!aligned16(feat) || !aligned16(out) || (out_bias && !aligned16(out_bias)))))
        return fail(NCF_EINVAL, "ncf_attn_forward: operands must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((B + 3) / 4);
#define LAUNCH(M) hipLaunchKernelGGL(attn_kernel<M>, dim3(blocks), dim3(256), 0, s, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, val, B, I, feat, ldfeat, Fdim, out_bias, out, ldout, wts)
    if (mode == 0) LAUNCH(0);
    else if (mode == 1) LAUNCH(1);
    else LAUNCH(2);
#undef LAUNCH
    return check_launch("ncf_attn_forward");
}
