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
#include "attn_util.h"
#include "group_pairs.h"
#include <math.h>
#include <atomic>
#include <type_traits>

#ifndef ATT_UNROLL
#define ATT_UNROLL 8   // measured (cfg 3, A/B): 2 -> 141.7 us, 4 -> 138.9, 8 -> 133.9, 16 -> 256 (registers)
#endif

namespace ncf {

// Training-time dropout of AttentionNet's hidden layer (attention_ncf.py:112-117: Linear, ReLU, Dropout, Linear — one mask element
// per (pair, rated entry, hidden unit)).  Never stored: element (entry e, units 4c..4c+3) keeps iff a 16-bit slice of a
// counter-based hash of (seed, e, c) reaches the threshold; the backward kernel regenerates the same mask.
struct AttDrop {
    uint32_t seed, thr;   // keep iff hash16 >= thr, thr = round(p * 65536)
    float scale;          // 1 / (1 - thr / 65536)
};
__device__ __forceinline__ uint32_t att_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ f32x4 att_mask4(uint32_t entry, int chunk, const AttDrop& d) {   // 0 or 1/(1-p') per element
    const uint32_t h0 = att_mix32(entry * 0x9E3779B1U ^ d.seed ^ (uint32_t)chunk * 0x85EBCA77U);
    const uint32_t h1 = att_mix32(h0 ^ 0x68E31DA4U);
    return f32x4{(h0 & 0xFFFFU) >= d.thr ? d.scale : 0.f, (h0 >> 16) >= d.thr ? d.scale : 0.f,
                 (h1 & 0xFFFFU) >= d.thr ? d.scale : 0.f, (h1 >> 16) >= d.thr ? d.scale : 0.f};
}

// MODE 0: MLP (relu + w1 dot), 1: linear (A == 1, pc + pr), 2: cosine (dot of normalised rows)
template <int MODE, bool DROP = false>
__global__ __launch_bounds__(256) void attn_kernel(const float* __restrict__ pc, int64_t ldpc, const float* __restrict__ pr,
                                                   int64_t ldpr, int A, const float* __restrict__ w1, float b1,
                                                   const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                   const float* __restrict__ val, int64_t B, int64_t I,
                                                   const float* __restrict__ feat, int64_t ldfeat, int Fdim,
                                                   const float* __restrict__ out_bias, float* __restrict__ out,
                                                   int64_t ldout, float* __restrict__ wts, AttDrop drop = AttDrop{}) {
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
                    if (MODE == 0 && DROP) {
                        const f32x4 mk = att_mask4((uint32_t)e, c, drop);
                        part = fmaf(wv[0], fmaxf(pcv[0] + r[u][0], 0.f) * mk[0], part);
                        part = fmaf(wv[1], fmaxf(pcv[1] + r[u][1], 0.f) * mk[1], part);
                        part = fmaf(wv[2], fmaxf(pcv[2] + r[u][2], 0.f) * mk[2], part);
                        part = fmaf(wv[3], fmaxf(pcv[3] + r[u][3], 0.f) * mk[3], part);
                    } else if (MODE == 0) {
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

// ---------------------------------------------------------------------------------------------------------------
// K3 backward (training step; the reference differentiates attention_ncf.py:176-216 through autograd).  One wave per pair b,
// given the forward's attention weights p_e (`wts`) and dOut[b, :]:
//   dpv_e   = val_e * <dOut[b,:], feat[i_e,:]>                      d feat[i_e,:] += p_e val_e dOut[b,:]        (float atomics)
//   ds_e    = p_e (dpv_e - sum_e' p_e' dpv_e')                      (softmax backward; sum_e ds_e = 0, so d b1 = 0 exactly)
//   MLP:    u = pc[b,:] + pr[i_e,:],  h = relu(u) * mask            d w1 += ds_e h   (per-pair partial rows, summed by ncf_colsum)
//           g = ds_e * w1 * [u > 0] * mask                          d pc[b,:] += g (registers),  d pr[i_e,:] += g (float atomics)
//   cosine: d pc[b,:] += ds_e pr[i_e,:],  d pr[i_e,:] += ds_e pc[b,:]
// A % 4 == 0, Fdim % 4 == 0, both <= 256.  `ds` (nnz floats) is scratch.  Atomic adds make d pr / d feat order-dependent in the
// last bits, like ncf_scatter_add_rows.
template <int MODE, bool DROP>
__global__ __launch_bounds__(256) void attn_backward_kernel(const float* __restrict__ pc, int64_t ldpc, const float* __restrict__ pr,
                                                            int64_t ldpr, int A, const float* __restrict__ w1,
                                                            const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                            const float* __restrict__ val, int64_t B, int64_t I,
                                                            const float* __restrict__ feat, int64_t ldfeat, int Fdim,
                                                            const float* __restrict__ wts, const float* __restrict__ dout, int64_t lddout,
                                                            float* __restrict__ d_pc, int64_t ldd_pc, float* __restrict__ d_pr, int64_t ldd_pr,
                                                            float* __restrict__ d_w1_part, float* __restrict__ d_feat, int64_t ldd_feat,
                                                            float* __restrict__ ds, AttDrop drop) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;  // wave-uniform
    const int64_t beg = rowptr[b], end = rowptr[b + 1];
    // ---- phase 1: dpv_e, the softmax-weighted mean t, and the feature-table gradient ----
    {
        const int chunks = Fdim / 4;
        int LPF = 8;
        while (LPF < chunks) LPF <<= 1;
        const int c = lane % LPF, eg = lane / LPF, EPI = 64 / LPF;
        const bool active = c < chunks;
        f32x4 dv = {0.f, 0.f, 0.f, 0.f};
        if (active) dv = *reinterpret_cast<const f32x4*>(dout + b * lddout + 4 * c);
        float tsum = 0.f;
        for (int64_t e0 = beg; e0 < end; e0 += EPI) {
            const int64_t e = e0 + eg;
            float part = 0.f, pe = 0.f, ve = 0.f;
            int64_t i = -1;
            if (e < end) {
                i = col[e];
                if (i >= 0 && i < I) { pe = wts[e]; ve = val[e]; } else i = -1;
            }
            if (i >= 0 && active) {
                const f32x4 f = *reinterpret_cast<const f32x4*>(feat + i * ldfeat + 4 * c);
                part = fmaf(dv[0], f[0], fmaf(dv[1], f[1], fmaf(dv[2], f[2], dv[3] * f[3])));
                const float a = pe * ve;
                float* g = d_feat + i * ldd_feat + 4 * c;
                atomicAdd(g + 0, a * dv[0]); atomicAdd(g + 1, a * dv[1]); atomicAdd(g + 2, a * dv[2]); atomicAdd(g + 3, a * dv[3]);
            }
            for (int off = 1; off < LPF; off <<= 1) part += __shfl_xor(part, off);
            const float dpv = part * ve;
            if (e < end && c == 0) { ds[e] = dpv; tsum += pe * dpv; }
        }
        tsum = wave_sum(tsum);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // other lanes of this wave read what this lane stored
        __builtin_amdgcn_wave_barrier();
        for (int64_t e = beg + lane; e < end; e += 64) {
            const int64_t i = col[e];
            ds[e] = (i >= 0 && i < I) ? wts[e] * (ds[e] - tsum) : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // ---- phase 2: back through the score ----
    {
        const int chunks = A / 4;
        int LPA = 8;
        while (LPA < chunks) LPA <<= 1;
        const int c = lane % LPA, eg = lane / LPA, EPI = 64 / LPA;
        const bool active = c < chunks;
        f32x4 pcv = {0.f, 0.f, 0.f, 0.f}, wv = {0.f, 0.f, 0.f, 0.f};
        if (active) {
            pcv = *reinterpret_cast<const f32x4*>(pc + b * ldpc + 4 * c);
            if (MODE == 0) wv = *reinterpret_cast<const f32x4*>(w1 + 4 * c);
        }
        f32x4 dpc = {0.f, 0.f, 0.f, 0.f}, dw = {0.f, 0.f, 0.f, 0.f};
        for (int64_t e0 = beg; e0 < end; e0 += EPI) {
            const int64_t e = e0 + eg;
            if (e < end && active) {
                const int64_t i = col[e];
                if (i >= 0 && i < I) {
                    const float dse = ds[e];
                    const f32x4 r = *reinterpret_cast<const f32x4*>(pr + i * ldpr + 4 * c);
                    f32x4 g;
                    if (MODE == 0) {
                        f32x4 mk = {1.f, 1.f, 1.f, 1.f};
                        if (DROP) mk = att_mask4((uint32_t)e, c, drop);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float u = pcv[j] + r[j];
                            dw[j] = fmaf(dse, fmaxf(u, 0.f) * mk[j], dw[j]);
                            g[j] = u > 0.f ? dse * wv[j] * mk[j] : 0.f;
                            dpc[j] += g[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            dpc[j] = fmaf(dse, r[j], dpc[j]);
                            g[j] = dse * pcv[j];
                        }
                    }
                    float* gp = d_pr + i * ldd_pr + 4 * c;
                    atomicAdd(gp + 0, g[0]); atomicAdd(gp + 1, g[1]); atomicAdd(gp + 2, g[2]); atomicAdd(gp + 3, g[3]);
                }
            }
        }
        for (int off = LPA; off < 64; off <<= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dpc[j] += __shfl_xor(dpc[j], off);
                dw[j] += __shfl_xor(dw[j], off);
            }
        }
        if (eg == 0 && active) {
            *reinterpret_cast<f32x4*>(d_pc + b * ldd_pc + 4 * c) = dpc;
            if (MODE == 0) *reinterpret_cast<f32x4*>(d_w1_part + b * (int64_t)A + 4 * c) = dw;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K3, LDS-tiled form for batches in which several pairs share one rated set (evaluation batches grouped by user,
// serving one user against the whole catalogue — reference webapp/backend.py:78-121).
//
// attn_kernel above gathers every pair's rated rows from L2 / the Infinity Cache: nnz·(A + Fdim)·4 bytes per PAIR
// (197 KB at config 3).  Here the CSR has one row per USER, `grp_ptr`/`pair_ids` list each user's pairs, and a
// 256-thread workgroup takes up to `ppw` pairs of one user: it stages the user's rated rows once, 64 entries at a
// time, into LDS — the attention-projection rows pr[col_e, :] (row stride A+4 floats: a lane's 16-byte reads of ITS
// entry's row are conflict-free) and the aggregation rows feat[col_e, :] — and every pair of the group is scored
// against the staged tile.  Gathered bytes drop by the group size; the score loop runs out of LDS:
//   lane = entry of the tile;  s_e = b1 + Σ_a w1[a]·relu(pc[b,a] + pr[e,a])   (pc row and w1 are LDS broadcasts)
// Softmax is ONLINE over the tiles (running max m, running sum l, output rescaled by exp(m - m')), so a user with any
// number of rated items needs no per-pair score storage:
//   out[b,:] = bias + (Σ_e exp(s_e - m)·val_e·feat[col_e,:]) / l
// which is the same weighted sum as attn_kernel's up to fp32 summation order (tests hold both to the oracle at 1e-5).
// An empty or fully masked row gives l = 0 -> zeros + bias (the nan_to_num case, :208-209).
typedef float f32x2 __attribute__((ext_vector_type(2)));

// FO = output registers per lane = ceil(Fdim / 64); NPF = 16-byte pieces of a tile each thread stages = (A + Fdim) / 32 rounded up to 4 / 8 / 16
#ifndef ATT_G_WAVES_PER_SIMD
#define ATT_G_WAVES_PER_SIMD 2   // one 512-thread workgroup per CU at 198 VGPRs; 4 (two workgroups, 128 VGPRs) spills 74+ registers
#endif
template <int MODE, int FO, int NPF>
__global__ __launch_bounds__(512, ATT_G_WAVES_PER_SIMD) void attn_grouped_kernel(const float* __restrict__ pc, int64_t ldpc, const float* __restrict__ pr,
                                                           int64_t ldpr, int A, const float* __restrict__ w1, float b1,
                                                           const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const float* __restrict__ val, int64_t R, int64_t I,
                                                           const int64_t* __restrict__ grp_ptr, const int64_t* __restrict__ pair_ids,
                                                           const int64_t* __restrict__ wg_ptr, int ppw,
                                                           const float* __restrict__ feat, int64_t ldfeat, int Fdim,
                                                           const float* __restrict__ out_bias, float* __restrict__ out, int64_t ldout,
                                                           float* __restrict__ wts, const int64_t* __restrict__ wts_off) {
    constexpr int EC = 64;     // entries per tile = lanes
    constexpr int MAXP = 4;    // pairs per wave (ppw <= 32)
    constexpr int NT = 512, NW = NT / 64;   // 8 waves share a tile: 2 workgroups per CU give 4 waves per SIMD
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int AS = A + 4;
    float* prc = reinterpret_cast<float*>(smem);           // [EC][AS]
    float* fc = prc + EC * AS;                             // [EC][Fdim]
    float* pcs = fc + EC * Fdim;                           // [ppw][A]
    float* w1s = pcs + ppw * A;                            // [A]
    float* vals = w1s + A;                                 // [EC]  val_e, 0 for a masked entry
    int* oks = reinterpret_cast<int*>(vals + EC);          // [EC]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t g = blockIdx.x;
    if (g >= wg_ptr[R]) return;                            // the grid is an upper bound (no host sync for its size)
    int64_t lo = 0, hi = R;                                // row r with wg_ptr[r] <= g < wg_ptr[r+1]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (wg_ptr[mid] <= g) lo = mid; else hi = mid;
    }
    const int64_t r = lo;
    const int64_t start = grp_ptr[r] + (g - wg_ptr[r]) * ppw;
    const int64_t left = grp_ptr[r + 1] - start;
    const int cnt = (int)(left < ppw ? left : ppw);
    const int64_t beg = rowptr[r], end = rowptr[r + 1];
    const int A4 = A / 4;
    const bool fvec = (Fdim % 4 == 0) && (ldfeat % 4 == 0);
    const int F4 = fvec ? Fdim / 4 : 0;
    const int PIECES = EC * (A4 + F4);                     // float4 pieces of one tile (pr part, then feat part)

    for (int idx = tid; idx < cnt * A4; idx += NT) {
        const int j = idx / A4, c = idx % A4;
        const int64_t b = pair_ids[start + j];
        *reinterpret_cast<f32x4*>(pcs + j * A + 4 * c) = *reinterpret_cast<const f32x4*>(pc + b * ldpc + 4 * c);
    }
    if (MODE == 0)
        for (int c = tid; c < A4; c += NT) *reinterpret_cast<f32x4*>(w1s + 4 * c) = *reinterpret_cast<const f32x4*>(w1 + 4 * c);

    // A tile's rows travel global -> registers -> LDS; the loads of tile t+1 are issued before tile t is scored and
    // land under its arithmetic (NPF pieces per thread in flight).
    f32x4 stage[NPF];
    float sval = 0.f;
    int sok = 0;
    auto fetch = [&](int64_t e0) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int idx = tid + NT * k;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < PIECES) {
                const bool isf = idx >= EC * A4;
                const int q = isf ? idx - EC * A4 : idx;
                const int per = isf ? F4 : A4;
                const int e = q / per, c = q % per;
                const int64_t ee = e0 + e;
                const int64_t i = ee < end ? (int64_t)col[ee] : -1;
                if (i >= 0 && i < I) v = *reinterpret_cast<const f32x4*>((isf ? feat + i * ldfeat : pr + i * ldpr) + 4 * c);
            }
            stage[k] = v;
        }
        if (tid < EC) {
            const int64_t e = e0 + tid;
            const int64_t i = e < end ? (int64_t)col[e] : -1;
            sok = i >= 0 && i < I;
            sval = sok ? val[e] : 0.f;
        }
    };
    auto commit = [&](int64_t e0) {
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int idx = tid + NT * k;
            if (idx < PIECES) {
                const bool isf = idx >= EC * A4;
                const int q = isf ? idx - EC * A4 : idx;
                const int per = isf ? F4 : A4;
                const int e = q / per, c = q % per;
                *reinterpret_cast<f32x4*>((isf ? fc + e * Fdim : prc + e * AS) + 4 * c) = stage[k];
            }
        }
        if (tid < EC) {
            oks[tid] = sok;
            vals[tid] = sval;
        }
        if (!fvec) {   // feature rows that cannot be read as 16-byte pieces: staged directly (rare shapes)
            for (int idx = tid; idx < EC * Fdim; idx += NT) {
                const int e = idx / Fdim, f = idx % Fdim;
                const int64_t ee = e0 + e;
                const int64_t i = ee < end ? (int64_t)col[ee] : -1;
                fc[e * Fdim + f] = (i >= 0 && i < I) ? feat[i * ldfeat + f] : 0.f;
            }
        }
    };

    float m[MAXP], l[MAXP], o[MAXP][FO];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        m[k] = -INFINITY;
        l[k] = 0.f;
#pragma unroll
        for (int f = 0; f < FO; ++f) o[k][f] = 0.f;
    }

    if (beg < end) fetch(beg);
    for (int64_t e0 = beg; e0 < end; e0 += EC) {
        __syncthreads();                                   // the previous tile has been consumed (and pcs / w1s are written)
        commit(e0);
        __syncthreads();
        if (e0 + EC < end) fetch(e0 + EC);                 // in flight while this tile is scored

        const bool ok = oks[lane] != 0;
        const float vl = vals[lane];
        const float* myrow = prc + lane * AS;
        // The wave's pairs (np of them, j = wave + NW*k) are scored TOGETHER: one read of the entry's row piece and of
        // w1 serves all of them, and their dependency chains interleave (2 waves per SIMD cannot hide LDS latency on
        // a single chain).  1, 2 or 4 slots are computed; slots k >= np work on pair 0's row and are ignored.
        const int np = cnt > wave ? (cnt - wave + NW - 1) / NW : 0;   // wave-uniform
        auto score_tile = [&](auto slots) {
        constexpr int NS = decltype(slots)::value;             // slots computed: 1, 2 or 4 (>= np)
        const float* pcj[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) pcj[k] = pcs + (k < np ? wave + NW * k : wave) * A;
        f32x2 s2[NS], t2[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) { s2[k] = f32x2{0.f, 0.f}; t2[k] = f32x2{0.f, 0.f}; }
#pragma unroll 2
        for (int c = 0; c < A4; ++c) {
            const f32x4 p = *reinterpret_cast<const f32x4*>(myrow + 4 * c);
            const f32x2 p01 = {p[0], p[1]}, p23 = {p[2], p[3]};
            f32x2 w01 = {0.f, 0.f}, w23 = {0.f, 0.f};
            if (MODE == 0) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(w1s + 4 * c);   // same address in every lane: broadcast
                w01 = f32x2{ww[0], ww[1]};
                w23 = f32x2{ww[2], ww[3]};
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(pcj[k] + 4 * c);  // broadcast
                const f32x2 q01 = {q[0], q[1]}, q23 = {q[2], q[3]};
                if (MODE == 0) {
                    f32x2 u = q01 + p01, v = q23 + p23;    // v_pk_add_f32
                    u = __builtin_elementwise_max(u, (f32x2){0.f, 0.f});
                    v = __builtin_elementwise_max(v, (f32x2){0.f, 0.f});
                    s2[k] = w01 * u + s2[k];               // v_pk_fma_f32
                    t2[k] = w23 * v + t2[k];
                } else {
                    s2[k] = q01 * p01 + s2[k];
                    t2[k] = q23 * p23 + t2[k];
                }
            }
        }
        float pv[NS];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            pv[k] = 0.f;
            if (k >= np) continue;                         // wave-uniform
            const float ssum = (s2[k][0] + t2[k][0]) + (s2[k][1] + t2[k][1]);
            const float sc = ok ? ssum + (MODE == 0 ? b1 : 0.f) : -INFINITY;
            if (wts && e0 + lane < end)                    // raw score now, normalised in place after the last tile
                wts[wts_off[pair_ids[start + wave + NW * k]] + (e0 - beg) + lane] = sc;
            const float mnew = fmaxf(m[k], wave_reduce_dpp(sc, [](float x, float y) { return fmaxf(x, y); }));
            if (mnew == -INFINITY) continue;               // nothing valid so far (wave-uniform): m, l, o stay 0
            const float scale = expf(m[k] - mnew);         // exp(-inf) = 0 on the first valid tile
            const float pe = ok ? expf(sc - mnew) : 0.f;
            l[k] = l[k] * scale + wave_reduce_dpp(pe, [](float x, float y) { return x + y; });
            m[k] = mnew;
            pv[k] = pe * vl;                               // attended_user_matrix entry (:212)
#pragma unroll
            for (int f = 0; f < FO; ++f) o[k][f] *= scale;
        }
#pragma unroll 16
        for (int e = 0; e < EC; ++e) {
            float fv[FO];
#pragma unroll
            for (int f = 0; f < FO; ++f) {
                const int ff = f * 64 + lane;
                fv[f] = fc[e * Fdim + (ff < Fdim ? ff : 0)];
            }
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const float a_e = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pv[k]), e));
#pragma unroll
                for (int f = 0; f < FO; ++f) o[k][f] = fmaf(a_e, fv[f], o[k][f]);
            }
        }
            };
        if (np > 2) score_tile(std::integral_constant<int, 4>{});
        else if (np == 2) score_tile(std::integral_constant<int, 2>{});
        else if (np == 1) score_tile(std::integral_constant<int, 1>{});
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int j = wave + NW * k;
        if (j >= cnt) break;
        const int64_t b = pair_ids[start + j];
        const float inv = l[k] > 0.f ? 1.0f / l[k] : 0.f;
#pragma unroll
        for (int f = 0; f < FO; ++f) {
            const int ff = f * 64 + lane;
            if (ff < Fdim) out[b * ldout + ff] = o[k][f] * inv + (out_bias ? out_bias[ff] : 0.f);
        }
        if (wts) {                                         // attention weights (:224): softmax of the stored raw scores
            float* wrow = wts + wts_off[b];                // written by this same lane pattern above
            for (int64_t e = lane; e < end - beg; e += 64) {
                const float sraw = wrow[e];
                wrow[e] = (l[k] > 0.f && sraw != -INFINITY) ? expf(sraw - m[k]) * inv : 0.f;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K3 grouped, scalar-operand form (round 2).  Same contract as attn_grouped_kernel; what changed is where the operands
// of the (pair, entry, a) loop live and which unit does what.  In the first form every lane re-read pc[b, 4c..] of its
// wave's pairs and w1[4c..] from LDS on every step: 5 of 6 `ds_read_b128` per step were wave-uniform broadcasts and the
// CU's LDS port, not the VALU, was saturated.  Here
//   * pc rows and w1 are WAVE-UNIFORM: they are read straight from global memory through the scalar cache
//     (`s_load_dwordx8` -> SGPR pairs) and enter the packed VALU ops as scalar operands — no LDS, no VGPRs;
//   * a lane (= one entry of the 64-entry tile) reads ITS row of the staged pr tile into registers ONCE per tile (and
//     a-block of CPB 16-byte chunks) and scores all of its wave's pairs against it: the inner loop is VALU only;
//   * the tile is staged by LDS-DMA (`global_load_lds_dwordx4`, per-lane source address, full 1 KiB pieces; rows of
//     A % 64 == 0 floats are XOR-swizzled on the SOURCE chunk so that the row reads are conflict-free).  Because the rows
//     move to registers first, the pr image is free again as soon as every wave has read it: the next tile's DMA is issued
//     THEN and lands under this tile's arithmetic — one image of each kind, no staging VGPRs, no ds_write pass;
//   * the rating-weighted aggregation  O[pair, f] += sum_e (p_e val_e) feat[e, f]  is a real contraction over the
//     entries: it runs on the matrix cores (`v_mfma_f32_16x16x4_f32`, exact fp32 fmaf chains; M = the workgroup's <= 16
//     pairs, N = 16 features per tile and wave, K = the tile's 64 entries), fed from LDS (P written once per tile, the feat
//     tile from its DMA image) — the first form spent 64 x (readlane + fma) per pair and tile on it;
//   * online softmax as before (running max m, per-LANE partial sums l reduced once at the end, O rescaled per tile).
// 256-thread workgroups of up to 16 pairs (4 per wave); LDS 64 x (A + Fdim) floats + P (58 KB at config 3): two
// workgroups per CU, each with its own barriers.
#ifdef ATT_SC_STAMP   // diagnostic build only: per-phase shader-clock sums of wave 0 of every workgroup -> the buffer passed as wts_off
#define ATT_STAMP(slot)                                                                                 \
    do {                                                                                                \
        unsigned long long t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                     \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        stamp_sum[slot] += t_ - stamp_last;                                                             \
        stamp_last = t_;                                                                                \
    } while (0)
#else
#define ATT_STAMP(slot) do {} while (0)
#endif

template <int MODE, int CPB, int NW>
__global__ __launch_bounds__(64 * NW, 2) void attn_grouped_sc_kernel(const float* __restrict__ pc, int64_t ldpc, const float* __restrict__ pr,
                                                                 int64_t ldpr, int A, const float* __restrict__ w1, float b1,
                                                                 const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                                 const float* __restrict__ val, int64_t R, int64_t I,
                                                                 const int64_t* __restrict__ grp_ptr, const int64_t* __restrict__ pair_ids,
                                                                 const int64_t* __restrict__ wg_ptr, int ppw,
                                                                 const float* __restrict__ feat, int64_t ldfeat, int Fdim,
                                                                 const float* __restrict__ out_bias, float* __restrict__ out, int64_t ldout,
                                                                 float* __restrict__ wts, const int64_t* __restrict__ wts_off) {
    constexpr int EC = 64, MAXP = 4, PP = MAXP * NW, MT = PP / 16, PS = 66, MAXNT = 4;   // NW = 4 or 8 waves, 4 pairs per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int A4 = A / 4, F4 = Fdim / 4;
    float* prt = reinterpret_cast<float*>(smem);           // [EC][A]     pr tile image (DMA)
    float* fct = prt + EC * A;                              // [EC][Fdim]  feat tile image (DMA)
    float* Pm = fct + EC * Fdim;                            // [PP][PS]    p_e * val_e of the current tile, pair-major
    float* scl = Pm + PP * PS;                              // [PP]        this tile's rescale factor per pair; at the end 1 / l
    int64_t* pid = reinterpret_cast<int64_t*>(scl + PP);    // [PP]        output row of each pair of the group
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g = blockIdx.x;
    if (g >= wg_ptr[R]) return;                            // the grid is an upper bound (no host sync for its size)
    int64_t lo = 0, hi = R;                                // row r with wg_ptr[r] <= g < wg_ptr[r+1]
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (wg_ptr[mid] <= g) lo = mid; else hi = mid;
    }
    const int64_t r = lo;
    const int64_t start = grp_ptr[r] + (g - wg_ptr[r]) * ppw;
    const int64_t left = grp_ptr[r + 1] - start;
    const int cnt = (int)(left < ppw ? left : ppw);
    const int64_t beg = rowptr[r], end = rowptr[r + 1];

    // LDS starts as zeros: slots of pairs / entries that nothing writes then hold finite values (0 x garbage must not
    // make a NaN in the MFMA of a VALID pair; rows of unused pairs are never stored)
    {
        const int total4 = (EC * (A + Fdim) + PP * PS + PP + 2 * PP) / 4;
        for (int i = tid; i < total4; i += (int)blockDim.x) reinterpret_cast<f32x4*>(smem)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // DMA pieces of one tile: the pr image is EC*A4 chunks = A4 wave-instructions, the feat image F4; wave w issues pieces
    // w, w+4, ...  A piece's lane writes chunk `j` of entry `e` of the image and fetches SOURCE chunk j ^ (e & 15) of
    // that row when rows are whole 256-byte bank rows (A % 64 == 0): the reader applies the same XOR.
    const bool swz = (A4 % 16) == 0;
    const int shA = (A4 & (A4 - 1)) == 0 ? __builtin_ctz(A4) : -1, shF = (F4 & (F4 - 1)) == 0 ? __builtin_ctz(F4) : -1;
    // the tile's own entry of this lane: column and rating, loaded two tiles ahead of their use — unconditionally (index
    // clamped into the row) and back to back: a dependent load would need a vmcnt wait, and every vmcnt wait also drains the
    // tile DMAs in flight.  Validity (inside the row, column inside the catalogue) is decided when the values are used.
    auto load_cv = [&](int64_t e0, int& c, float& v) {
        const int64_t e = e0 + lane < end ? e0 + lane : (end > beg ? end - 1 : beg);
        c = end > beg ? col[e] : -1;
        v = end > beg ? val[e] : 0.f;
    };
    auto valid_c = [&](int64_t e0, int c) { return (e0 + lane < end && c >= 0 && c < I) ? c : -1; };
    auto issue = [&](const float* tab, int64_t ld, int X4, int shX, bool xorj, unsigned lds_base, int cols) {
        // `cols` = this lane's col of the tile being fetched (or -1).  A piece (one wave-instruction, 64 chunks) covers
        // RPP = 64 / X4 whole rows when rows are a power of two of chunks: the row's col then comes from 1 / 2 / 4 readlanes
        // (no LDS round trip, nothing to wait for) and every piece costs a dozen instructions.
        const int rpp = shX >= 0 ? 64 >> shX : 0;
        if (rpp >= 1 && rpp <= 4) {
            const int sub = lane >> shX, j = lane & (X4 - 1);
            for (int piece = wave; piece < X4; piece += NW) {   // wave-uniform trip count
                const int e0p = piece * rpp;
                int ci = __builtin_amdgcn_readlane(cols, e0p);
                if (rpp >= 2) { const int c1 = __builtin_amdgcn_readlane(cols, e0p + 1); ci = sub == 1 ? c1 : ci; }
                if (rpp == 4) {
                    const int c2 = __builtin_amdgcn_readlane(cols, e0p + 2), c3 = __builtin_amdgcn_readlane(cols, e0p + 3);
                    ci = sub == 2 ? c2 : (sub == 3 ? c3 : ci);
                }
                const int jj = xorj ? (j ^ ((e0p + sub) & 15)) : j;
                if (ci >= 0) dma16(tab + (int64_t)ci * ld + 4 * jj, lds_base + (unsigned)piece * 1024u);
            }
            return;
        }
        for (int piece = wave; piece < X4; piece += NW) {  // general row widths: (entry, chunk) by division, col by a shuffle
            const int gi = piece * 64 + lane;
            const int e = shX >= 0 ? gi >> shX : gi / X4;
            const int j = shX >= 0 ? gi & (X4 - 1) : gi - e * X4;
            const int ci = __shfl(cols, e);
            if (ci >= 0) dma16(tab + (int64_t)ci * ld + 4 * (xorj ? (j ^ (e & 15)) : j), lds_base + (unsigned)piece * 1024u);
        }
    };
    auto issue_pr = [&](int cols) { issue(pr, ldpr, A4, shA, swz, lds0, cols); };
    auto issue_feat = [&](int cols) { issue(feat, ldfeat, F4, shF, false, lds0 + (unsigned)(EC * A * 4), cols); };

    // the wave's pairs: slots j = wave + NW*k; their pc rows are wave-uniform pointers (scalar loads)
    const int np = cnt > wave ? (cnt - wave + NW - 1) / NW : 0;
    const float* pcrow[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int j = k < np ? wave + NW * k : 0;
        const int64_t b = pair_ids[start + j];
        const int blo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffff)), bhi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
        pcrow[k] = pc + (((int64_t)bhi << 32) | (unsigned)blo) * ldpc;
#ifdef ATT_SC_SAMEROW      // diagnostic build (wrong results): every slot reads pair 0's row -> the scalar working set shrinks 4x
        pcrow[k] = pcrow[0];
#endif
    }
    if (tid < PP) pid[tid] = tid < cnt ? pair_ids[start + tid] : -1;

    // MFMA roles: a job = one 16-pair x 16-feature tile of O over all 16 k-steps (4 entries each) of a tile; MT * NTILES jobs
    const int NTILES = (Fdim + 15) / 16;
    f32x4 acc[MAXNT];
#pragma unroll
    for (int i = 0; i < MAXNT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    float m[MAXP], l[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { m[k] = -INFINITY; l[k] = 0.f; }

    const int64_t ntiles = (end - beg + EC - 1) / EC;
    int c_cur = -1, c_nxt = -1, c_nn = -1;
    float v_cur = 0.f, v_nxt = 0.f, v_nn = 0.f;
    __syncthreads();                                       // LDS zeroed before the first DMA lands
    if (ntiles > 0) {
        load_cv(beg, c_cur, v_cur);
        load_cv(beg + EC, c_nxt, v_nxt);
        c_cur = valid_c(beg, c_cur);
        c_nxt = valid_c(beg + EC, c_nxt);
        issue_pr(c_cur);
    }
    const int sw = swz ? (lane & 15) : 0;
    const int g4 = lane >> 4, i16 = lane & 15;
#ifdef ATT_SC_STAMP
    unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_last)::"memory");
#endif
    for (int64_t t = 0; t < ntiles; ++t) {
        const int64_t e0 = beg + t * EC;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of pr(t) have landed (and its col / val loads) ...
        __syncthreads();                                   // ... and everybody's; the MFMAs of tile t-1 have read fct / Pm / scl
        if (t > 0) {                                       // rotate the look-ahead registers HERE: nothing is in flight, no wait
            c_cur = c_nxt; v_cur = v_nxt;
            c_nxt = valid_c(e0 + EC, c_nn); v_nxt = v_nn;
        }
        ATT_STAMP(0);
        issue_feat(c_cur);                                 // lands under the scoring below (read after the third barrier)
        load_cv(e0 + 2 * EC, c_nn, v_nn);
        ATT_STAMP(1);

        const bool ok = c_cur >= 0;
        const float vl = ok ? v_cur : 0.f;
        const float* myrow = prt + lane * A;
        f32x2 s2[MAXP], t2[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) { s2[k] = f32x2{0.f, 0.f}; t2[k] = f32x2{0.f, 0.f}; }
        for (int blk = 0; blk < A4; blk += CPB) {
            f32x4 row[CPB];
#pragma unroll
            for (int c = 0; c < CPB; ++c) row[c] = *reinterpret_cast<const f32x4*>(myrow + 4 * ((blk + c) ^ sw));
            ATT_STAMP(2);
            if (blk + CPB >= A4) {                         // last a-block: once every wave holds its rows the pr image is free
                // raw barrier: __syncthreads() would wait vmcnt(0) first, i.e. for the feat DMA issued a moment ago
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's row reads have returned
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < ntiles) issue_pr(c_nxt);       // tile t+1 lands under this tile's arithmetic
            }
            ATT_STAMP(3);
            // Scalar operands arrive G chunks at a time, one step ahead: a wait on scalar loads is always lgkmcnt(0)
            // (they return out of order), so the order inside a step is  wait(step s)  ->  issue loads(step s+1)  ->
            // VALU(step s): the next step's loads fly under this step's arithmetic, and the live scalars stay at two
            // steps' worth (all of them at once would be 1024 SGPRs: spilled through v_writelane / v_readlane).
            auto score_block = [&](auto slots) {
                constexpr int NS = decltype(slots)::value;     // slots computed: 1, 2 or 4 (>= np)
                constexpr int G = CPB >= 2 ? 2 : 1, NST = CPB / G;
                f32x4 qs[2][NS][G], ws[2][G];
                auto load_step = [&](int st, int slot) {
#pragma unroll
                    for (int gk = 0; gk < G; ++gk) {
                        if (MODE != 2) ws[slot][gk] = *reinterpret_cast<const f32x4*>(w1 + 4 * (blk + st * G + gk));
#pragma unroll
                        for (int k = 0; k < NS; ++k) qs[slot][k][gk] = *reinterpret_cast<const f32x4*>(pcrow[k] + 4 * (blk + st * G + gk));
                    }
                };
                load_step(0, 0);
#pragma unroll
                for (int st = 0; st < NST; ++st) {
                    const int cur = st & 1;
                    asm volatile("" ::"s"(qs[cur][0][0][0]));   // the compiler's wait for step st sits HERE, before the next issue
                    __builtin_amdgcn_sched_barrier(0);
                    if (st + 1 < NST) load_step(st + 1, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int gk = 0; gk < G; ++gk) {
                        const int c = st * G + gk;
                        const f32x2 p01 = {row[c][0], row[c][1]}, p23 = {row[c][2], row[c][3]};
#pragma unroll
                        for (int k = 0; k < NS; ++k) {
                            const f32x4 q = qs[cur][k][gk];
                            const f32x2 q01 = {q[0], q[1]}, q23 = {q[2], q[3]};
                            if (MODE == 0 || MODE == 3) {
                                const f32x4 ww = ws[cur][gk];
                                const f32x2 w01 = {ww[0], ww[1]}, w23 = {ww[2], ww[3]};
                                f32x2 u, v;
                                if (MODE == 3) {
                                    // operands scaled by 2^-64 (w1 by 2^64): relu(p + q) IS the [0, 1] clamp of the sum, an output
                                    // modifier of the packed add — one instruction for two elements where gfx950 has no packed fp32
                                    // max (2 x v_max_f32 otherwise); powers of two scale exactly: same bits as the unscaled relu
                                    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(u) : "v"(p01), "s"(q01));
                                    asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(v) : "v"(p23), "s"(q23));
                                } else {
                                    u = p01 + q01, v = p23 + q23;               // v_pk_add_f32, scalar operand
                                    u = __builtin_elementwise_max(u, (f32x2){0.f, 0.f});   // 2 x v_max_f32
                                    v = __builtin_elementwise_max(v, (f32x2){0.f, 0.f});
                                }
                                s2[k] = u * w01 + s2[k];                        // v_pk_fma_f32, scalar operand
                                t2[k] = v * w23 + t2[k];
                            } else {
                                s2[k] = p01 * q01 + s2[k];
                                t2[k] = p23 * q23 + t2[k];
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (np > 2) score_block(std::integral_constant<int, 4>{});
            else if (np == 2) score_block(std::integral_constant<int, 2>{});
            else if (np == 1) score_block(std::integral_constant<int, 1>{});
            ATT_STAMP(4);
        }
        // Epilogue of the tile, the wave's pairs side by side (branch-free, so that the four dependency chains of DPP steps
        // and exponentials interleave): tile maximum, running maximum, rescale factor, p_e.
        float sc[MAXP], mx[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const float ssum = (s2[k][0] + t2[k][0]) + (s2[k][1] + t2[k][1]);
            sc[k] = ok ? ssum + (MODE != 2 ? b1 : 0.f) : -INFINITY;
            mx[k] = sc[k];
        }
        wave_reduce_dpp_n<MAXP>(mx, [](float x, float y) { return fmaxf(x, y); });
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const float mnew = fmaxf(m[k], mx[k]);
            const bool any = mnew != -INFINITY;            // wave-uniform; false: nothing valid so far, m, l, O stay 0
            const float scale = any ? exp_le0(m[k] - mnew) : 1.f;       // exp(-inf) = 0 on the first valid tile
            const float pe = (any && ok) ? exp_le0(sc[k] - mnew) : 0.f;
            l[k] = l[k] * scale + pe;                      // per-lane partial sum; lanes are added once, after the last tile
            m[k] = mnew;
            if (k < np) {                                  // wave-uniform
                const int j = wave + NW * k;
                if (wts && e0 + lane < end)                // raw score now, normalised in place after the last tile
                    wts[wts_off[pid[j]] + (e0 - beg) + lane] = sc[k];
                Pm[j * PS + lane] = pe * vl;               // attended_user_matrix entry (:212)
                if (lane == 0) scl[j] = scale;
            }
        }
        ATT_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's pieces of feat(t) (and of pr(t+1)) have landed; P written
        __builtin_amdgcn_s_barrier();                      // P, scl and the feat image of this tile are complete
        ATT_STAMP(6);
        {
            // jobs: (16-pair row tile mt, 16-feature column tile nt) = (j % MT, j / MT) for j = wave, wave + NW, ...
#pragma unroll
            for (int n = 0; n < MAXNT; ++n) {
                const int job = wave + NW * n;
                const int mt = job % MT, nt = job / MT;
                if (nt >= NTILES) break;                   // wave-uniform
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(scl + 16 * mt + 4 * g4);   // accumulator register i holds pair row 16*mt + 4*g4 + i
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[n][i] *= s4[i];
                const float* fb = fct + 16 * nt + i16;
                const float* pa = Pm + (16 * mt + i16) * PS + g4;
#pragma unroll
                for (int q0 = 0; q0 < EC / 4; q0 += 8) {    // 8 k-steps' operands first (16 LDS reads in flight), then their chain
                    float av[8], bv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {           // A[i = pair][k = 4q + g4], B[k][j = feature]
                        av[q] = pa[4 * (q0 + q)];
                        bv[q] = fb[(4 * (q0 + q) + g4) * Fdim];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], bv[q], acc[n], 0, 0, 0);
                }
            }
        }
        ATT_STAMP(7);
    }
#ifdef ATT_SC_STAMP
    if (lane == 0 && wave == 0 && !wts && wts_off) {
        unsigned long long* dbg = (unsigned long long*)wts_off;
        for (int i = 0; i < 8; ++i) dbg[blockIdx.x * 8 + i] = stamp_sum[i];
    }
#endif
    float linv[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        linv[k] = 0.f;
        if (k >= np) continue;
        const float lt = wave_reduce_dpp(l[k], [](float x, float y) { return x + y; });
        linv[k] = lt > 0.f ? 1.0f / lt : 0.f;
    }
    __syncthreads();                                       // the last tile's MFMAs have read scl
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
        if (k < np && lane == 0) scl[wave + NW * k] = linv[k];
    __syncthreads();
    {
#pragma unroll
        for (int n = 0; n < MAXNT; ++n) {
            const int job = wave + NW * n;
            const int mt = job % MT, nt = job / MT;
            if (nt >= NTILES) break;
            const f32x4 s4 = *reinterpret_cast<const f32x4*>(scl + 16 * mt + 4 * g4);
            const int f = 16 * nt + i16;
            const float bias = (out_bias && f < Fdim) ? out_bias[f] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * mt + 4 * g4 + i;
                if (row < cnt && f < Fdim) out[pid[row] * ldout + f] = acc[n][i] * s4[i] + bias;
            }
        }
    }
    if (wts) {                                             // attention weights (:224): softmax of the stored raw scores
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            if (k >= np) continue;
            float* wrow = wts + wts_off[pid[wave + NW * k]];    // written by this same lane pattern above
            for (int64_t e = lane; e < end - beg; e += 64) {
                const float sraw = wrow[e];
                wrow[e] = (linv[k] > 0.f && sraw != -INFINITY) ? expf(sraw - m[k]) * linv[k] : 0.f;
            }
        }
    }
}

// ---- pairs listed row by row for the grouped kernel (a counting sort; order inside a row is irrelevant: every pair's
// output is computed independently) ----
__global__ void group_count_kernel(const int64_t* __restrict__ pair_row, int64_t B, int64_t R, int* __restrict__ counts,
                                   int* __restrict__ bad) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = b < B ? pair_row[b] : -1;
    const bool ok = b < B && r >= 0 && r < R;
    if (b < B && !ok) *bad = 1;
    (void)wave_aggregated_inc(counts, r, ok);
}
// one workgroup: exclusive scans of counts -> grp_ptr and of ceil(counts / ppw) -> wg_ptr; cursor[r] = grp_ptr[r]
__global__ __launch_bounds__(1024) void group_scan_kernel(const int* __restrict__ counts, int64_t R, int ppw,
                                                          int64_t* __restrict__ grp_ptr, int64_t* __restrict__ wg_ptr,
                                                          int* __restrict__ cursor, int32_t* __restrict__ wg_row) {
    __shared__ int64_t sa[1024], sb[1024];
    __shared__ int64_t carry_a, carry_b;
    if (threadIdx.x == 0) { carry_a = 0; carry_b = 0; }
    __syncthreads();
    for (int64_t base = 0; base < R; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const int64_t c = i < R ? counts[i] : 0;
        const int64_t wgs = (c + ppw - 1) / ppw;
        sa[threadIdx.x] = c;
        sb[threadIdx.x] = wgs;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {         // Hillis-Steele inclusive scan
            const int64_t va = threadIdx.x >= off ? sa[threadIdx.x - off] : 0;
            const int64_t vb = threadIdx.x >= off ? sb[threadIdx.x - off] : 0;
            __syncthreads();
            sa[threadIdx.x] += va;
            sb[threadIdx.x] += vb;
            __syncthreads();
        }
        if (i < R) {
            const int64_t ea = carry_a + sa[threadIdx.x] - c, eb = carry_b + sb[threadIdx.x] - wgs;
            grp_ptr[i] = ea;
            wg_ptr[i] = eb;
            cursor[i] = (int)ea;
            if (wg_row)
                for (int64_t w = 0; w < wgs; ++w) wg_row[eb + w] = (int32_t)i;   // workgroup -> row, read instead of a binary search
        }
        __syncthreads();
        if (threadIdx.x == 1023) { carry_a += sa[1023]; carry_b += sb[1023]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { grp_ptr[R] = carry_a; wg_ptr[R] = carry_b; }
}
template <bool LDS_COUNTERS>
__global__ __launch_bounds__(1024) void group_small_kernel(const int64_t* __restrict__ pair_row, int64_t B, int64_t R, int ppw,
                                                           int* __restrict__ gcounts, int* __restrict__ gcursor, int* __restrict__ bad,
                                                           int64_t* __restrict__ grp_ptr, int64_t* __restrict__ wg_ptr,
                                                           int64_t* __restrict__ pair_ids, int32_t* __restrict__ wg_row) {
    __shared__ int lds[group_small_lds_ints<LDS_COUNTERS>(1024)];
    group_small_body<LDS_COUNTERS, 1024>(pair_row, B, R, ppw, gcounts, gcursor, bad, grp_ptr, wg_ptr, pair_ids, wg_row, lds);
}
__global__ void group_scatter_kernel(const int64_t* __restrict__ pair_row, int64_t B, int64_t R, int* __restrict__ cursor,
                                     int64_t* __restrict__ pair_ids) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = b < B ? pair_row[b] : -1;
    const bool ok = b < B && r >= 0 && r < R;
    const int slot = wave_aggregated_inc(cursor, r, ok);
    if (ok) pair_ids[slot] = b;
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

static int attn_forward_impl(const char* who, int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A,
                             const float* w1, float b1, const int64_t* rowptr, const int32_t* col, const float* val,
                             int64_t B, int64_t I, const float* feat, int64_t ldfeat, int Fdim, const float* out_bias,
                             float* out, int64_t ldout, float* wts, const AttDrop* drop, ncf_stream_t stream) {
    if (mode < 0 || mode > 3) return fail(NCF_EINVAL, "%s: bad mode %d", who, mode);
    if (mode == NCF_ATT_MLP_SCALED) mode = NCF_ATT_MLP;   // w' relu(p' + q') with exact power-of-two scales IS w relu(p + q): same kernel
    if (B < 0 || I < 0 || A <= 0 || Fdim <= 0) return fail(NCF_EINVAL, "%s: bad sizes", who);
    if (B == 0) return NCF_OK;
    if (!pc || !pr || !rowptr || !feat || !out || !wts) return fail(NCF_EINVAL, "%s: null pointer", who);
    if (mode == NCF_ATT_MLP && !w1) return fail(NCF_EINVAL, "%s: w1 is null", who);
    if (mode == NCF_ATT_LINEAR && A != 1) return fail(NCF_EINVAL, "%s: linear mode needs A == 1", who);
    if (ldpc < A || ldpr < A || ldfeat < Fdim || ldout < Fdim) return fail(NCF_EINVAL, "%s: leading dimension smaller than row", who);
    if ((A % 4 == 0 && (!aligned16(pc) || !aligned16(pr) || (w1 && !aligned16(w1)))) ||
        (Fdim % 4 == 0 && (!aligned16(feat) || !aligned16(out) || (out_bias && !aligned16(out_bias)))))
        return fail(NCF_EINVAL, "%s: operands must be 16-byte aligned", who);
    if (drop && (mode != NCF_ATT_MLP || A % 4 || A > 256 || ldpr % 4 || ldpc % 4))
        return fail(NCF_EUNSUPPORTED, "%s: hidden-layer dropout needs the MLP mode with A %% 4 == 0, A <= 256", who);
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((B + 3) / 4);
#define LAUNCH(M) hipLaunchKernelGGL((attn_kernel<M, false>), dim3(blocks), dim3(256), 0, s, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, val, B, I, feat, ldfeat, Fdim, out_bias, out, ldout, wts, AttDrop{})
    if (drop) hipLaunchKernelGGL((attn_kernel<0, true>), dim3(blocks), dim3(256), 0, s, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, val, B, I, feat, ldfeat, Fdim, out_bias, out, ldout, wts, *drop);
    else if (mode == 0) LAUNCH(0);
    else if (mode == 1) LAUNCH(1);
    else LAUNCH(2);
#undef LAUNCH
    return check_launch(who);
}

static bool att_drop_args(float p, uint32_t seed, AttDrop& d) {
    d.seed = seed;
    d.thr = (uint32_t)(p * 65536.f + 0.5f);
    if (d.thr > 65535u) d.thr = 65535u;
    d.scale = 65536.f / (float)(65536u - d.thr);
    return d.thr != 0;
}

extern "C" int ncf_attn_forward(int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A,
                                const float* w1, float b1, const int64_t* rowptr, const int32_t* col, const float* val,
                                int64_t B, int64_t I, const float* feat, int64_t ldfeat, int Fdim, const float* out_bias,
                                float* out, int64_t ldout, float* wts, ncf_stream_t stream) {
    return attn_forward_impl("ncf_attn_forward", mode, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, val, B, I, feat, ldfeat, Fdim, out_bias,
                             out, ldout, wts, nullptr, stream);
}

extern "C" int ncf_attn_forward_dropout(int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A,
                                        const float* w1, float b1, const int64_t* rowptr, const int32_t* col, const float* val,
                                        int64_t B, int64_t I, const float* feat, int64_t ldfeat, int Fdim, const float* out_bias,
                                        float* out, int64_t ldout, float* wts, uint32_t seed, float p, ncf_stream_t stream) {
    if (!(p >= 0.f && p < 1.f)) return fail(NCF_EINVAL, "ncf_attn_forward_dropout: p = %g is not in [0, 1)", (double)p);
    AttDrop d;
    const bool on = att_drop_args(p, seed, d);
    return attn_forward_impl("ncf_attn_forward_dropout", mode, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, val, B, I, feat, ldfeat, Fdim,
                             out_bias, out, ldout, wts, on ? &d : nullptr, stream);
}

extern "C" int ncf_attn_backward(int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A, const float* w1,
                                 const int64_t* rowptr, const int32_t* col, const float* val, int64_t B, int64_t I,
                                 const float* feat, int64_t ldfeat, int Fdim, const float* wts, const float* dout, int64_t lddout,
                                 float* d_pc, int64_t ldd_pc, float* d_pr, int64_t ldd_pr, float* d_w1_part, float* d_feat,
                                 int64_t ldd_feat, float* ds, uint32_t seed, float p, ncf_stream_t stream) {
    if (mode == NCF_ATT_MLP_SCALED) mode = NCF_ATT_MLP;
    if (mode != NCF_ATT_MLP && mode != NCF_ATT_COS) return fail(NCF_EUNSUPPORTED, "ncf_attn_backward: mode %d (MLP and cosine only)", mode);
    if (B < 0 || I < 0 || A <= 0 || Fdim <= 0) return fail(NCF_EINVAL, "ncf_attn_backward: bad sizes");
    if (B == 0) return NCF_OK;
    if (!pc || !pr || !rowptr || !feat || !wts || !dout || !d_pc || !d_pr || !d_feat || !ds || (mode == NCF_ATT_MLP && (!w1 || !d_w1_part)))
        return fail(NCF_EINVAL, "ncf_attn_backward: null pointer");
    if (A % 4 || Fdim % 4 || A > 256 || Fdim > 256 || ldpc % 4 || ldpr % 4 || ldfeat % 4 || lddout % 4 || ldd_pc % 4)
        return fail(NCF_EUNSUPPORTED, "ncf_attn_backward: needs A, Fdim and the leading dimensions %% 4 == 0, A, Fdim <= 256");
    if (ldpc < A || ldpr < A || ldfeat < Fdim || lddout < Fdim || ldd_pc < A || ldd_pr < A || ldd_feat < Fdim)
        return fail(NCF_EINVAL, "ncf_attn_backward: leading dimension smaller than row");
    if (!aligned16(pc) || !aligned16(pr) || (w1 && !aligned16(w1)) || !aligned16(feat) || !aligned16(dout) || !aligned16(d_pc) ||
        (d_w1_part && !aligned16(d_w1_part)))
        return fail(NCF_EINVAL, "ncf_attn_backward: operands must be 16-byte aligned");
    if (!(p >= 0.f && p < 1.f)) return fail(NCF_EINVAL, "ncf_attn_backward: p = %g is not in [0, 1)", (double)p);
    AttDrop d;
    const bool on = att_drop_args(p, seed, d) && mode == NCF_ATT_MLP;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((B + 3) / 4);
#define LAUNCHB(M, D) hipLaunchKernelGGL((attn_backward_kernel<M, D>), dim3(blocks), dim3(256), 0, s, pc, ldpc, pr, ldpr, A, w1, rowptr, col, val, B, I, \
                                         feat, ldfeat, Fdim, wts, dout, lddout, d_pc, ldd_pc, d_pr, ldd_pr, d_w1_part, d_feat, ldd_feat, ds, d)
    if (mode == NCF_ATT_MLP && on) LAUNCHB(0, true);
    else if (mode == NCF_ATT_MLP) LAUNCHB(0, false);
    else LAUNCHB(2, false);
#undef LAUNCHB
    return check_launch("ncf_attn_backward");
}

extern "C" int ncf_attn_forward_grouped(int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A,
                                        const float* w1, float b1, const int64_t* rowptr, const int32_t* col, const float* val,
                                        int64_t R, int64_t I, const int64_t* grp_ptr, const int64_t* pair_ids,
                                        const int64_t* wg_ptr, int64_t B, int pairs_per_wg, const float* feat, int64_t ldfeat,
                                        int Fdim, const float* out_bias, float* out, int64_t ldout, float* wts,
                                        const int64_t* wts_off, ncf_stream_t stream) {
    if (mode != NCF_ATT_MLP && mode != NCF_ATT_COS && mode != NCF_ATT_MLP_SCALED)
        return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_grouped: mode %d has no LDS-tiled form (use ncf_attn_forward)", mode);
    const bool scaled = mode == NCF_ATT_MLP_SCALED;       // only the scalar-operand form evaluates relu as a clamp; elsewhere
    if (scaled) mode = NCF_ATT_MLP;                       // the scaled operands give the same bits through max
    if (B < 0 || R < 0 || I < 0 || A <= 0 || Fdim <= 0) return fail(NCF_EINVAL, "ncf_attn_forward_grouped: bad sizes");
    if (B == 0 || R == 0) return NCF_OK;
    if (!pc || !pr || !rowptr || !grp_ptr || !pair_ids || !wg_ptr || !feat || !out)
        return fail(NCF_EINVAL, "ncf_attn_forward_grouped: null pointer");
    if (mode == NCF_ATT_MLP && !w1) return fail(NCF_EINVAL, "ncf_attn_forward_grouped: w1 is null");
    if (wts && !wts_off) return fail(NCF_EINVAL, "ncf_attn_forward_grouped: weights requested without their per-pair offsets");
    if (ldpc < A || ldpr < A || ldfeat < Fdim || ldout < Fdim)
        return fail(NCF_EINVAL, "ncf_attn_forward_grouped: leading dimension smaller than row");
    if (pairs_per_wg < 1 || pairs_per_wg > 32) return fail(NCF_EINVAL, "ncf_attn_forward_grouped: pairs_per_wg must be 1..32");
    if (A % 4 || ldpc % 4 || ldpr % 4 || A > 256 || Fdim > 256)
        return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_grouped: needs A %% 4 == 0, A <= 256, Fdim <= 256 (A = %d, Fdim = %d)", A, Fdim);
    if (!aligned16(pc) || !aligned16(pr) || (w1 && !aligned16(w1)) || (Fdim % 4 == 0 && ldfeat % 4 == 0 && !aligned16(feat)))
        return fail(NCF_EINVAL, "ncf_attn_forward_grouped: operands must be 16-byte aligned");
    const bool fvec = Fdim % 4 == 0 && ldfeat % 4 == 0;
    hipStream_t s = (hipStream_t)stream;
    // upper bound on sum_r ceil(n_r / ppw): every non-empty row adds at most one partly filled workgroup, and at most
    // min(R, B) rows are non-empty (R may be a whole user base with B pairs of a few users: device-resident evaluation)
    const unsigned blocks = (unsigned)((B + pairs_per_wg - 1) / pairs_per_wg + (R < B ? R : B));
    // the dynamic-LDS limit of an instantiation is raised ONCE per device (to the 160 KiB a workgroup may declare)
    auto raise_lds = [](const void* fn, std::atomic<unsigned long long>& done) -> bool {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (done.load(std::memory_order_relaxed) >> dev & 1ull) return true;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        done.fetch_or(1ull << dev, std::memory_order_relaxed);
        return true;
    };
    // ---- scalar-operand form (default): tiles by LDS-DMA, aggregation on the matrix cores; 4 waves for up to 16 pairs per
    // workgroup, 8 waves for 17..32 (half the staged bytes per pair: the choice for batches that fill the chip either way) ----
    {
        const int A4 = A / 4;
        const int pp = pairs_per_wg <= 16 ? 16 : 32;
        const int jobs = (pp / 16) * ((Fdim + 15) / 16);   // 16 x 16 output tiles of the aggregation; at most 4 per wave
        const size_t lds2 = ((size_t)64 * (A + Fdim) + pp * 66 + pp + 2 * pp) * 4;
        const bool sc_ok = fvec && lds2 <= 160 * 1024 && aligned16(feat) && jobs <= 4 * (pp / 4);
        const int force = option(NCF_OPT_ATTN_GROUPED_KERNEL);
        if (sc_ok && force != 1) {
#define LAUNCH_SC(M, C)                                                                                                           \
    do {                                                                                                                       \
        static std::atomic<unsigned long long> done{0}, done2{0};                                                              \
        if (lds2 > 64 * 1024 && !(raise_lds((const void*)attn_grouped_sc_kernel<M, C, 4>, done) &&                            \
                                  raise_lds((const void*)attn_grouped_sc_kernel<M, C, 8>, done2)))                            \
            return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_grouped: cannot reserve %zu bytes of LDS", lds2);                  \
        if (pp == 16)                                                                                                          \
            hipLaunchKernelGGL((attn_grouped_sc_kernel<M, C, 4>), dim3(blocks), dim3(256), lds2, s, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, \
                               val, R, I, grp_ptr, pair_ids, wg_ptr, pairs_per_wg, feat, ldfeat, Fdim, out_bias, out, ldout, wts, wts_off); \
        else                                                                                                                   \
            hipLaunchKernelGGL((attn_grouped_sc_kernel<M, C, 8>), dim3(blocks), dim3(512), lds2, s, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, \
                               val, R, I, grp_ptr, pair_ids, wg_ptr, pairs_per_wg, feat, ldfeat, Fdim, out_bias, out, ldout, wts, wts_off); \
    } while (0)
#define LAUNCH_SC_C(M) do { if (A4 % 32 == 0) LAUNCH_SC(M, 32); else if (A4 % 16 == 0) LAUNCH_SC(M, 16); else if (A4 % 8 == 0) LAUNCH_SC(M, 8); else LAUNCH_SC(M, 1); } while (0)
            if (mode == NCF_ATT_MLP && scaled) LAUNCH_SC_C(3);
            else if (mode == NCF_ATT_MLP) LAUNCH_SC_C(0);
            else LAUNCH_SC_C(2);
#undef LAUNCH_SC_C
#undef LAUNCH_SC
            return check_launch("ncf_attn_forward_grouped");
        }
        if (force == 2) return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_grouped: the scalar-operand form needs Fdim %% 4 == 0 and <= 160 KiB of LDS");
    }
    // ---- first form: tile staged through registers, operands broadcast from LDS ----
    const size_t lds = ((size_t)64 * (A + 4) + (size_t)64 * Fdim + (size_t)pairs_per_wg * A + A + 64 + 64) * 4;
    if (lds > 160 * 1024) return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_grouped: tile needs %zu bytes of LDS", lds);
    const int pieces_per_thread = (64 * (A / 4 + (fvec ? Fdim / 4 : 0)) + 511) / 512;   // <= 16 for A, Fdim <= 256
#define LAUNCH1(M, F, P)                                                                                                          \
    do {                                                                                                                       \
        static std::atomic<unsigned long long> done{0};                                                                        \
        if (lds > 64 * 1024 && !raise_lds((const void*)attn_grouped_kernel<M, F, P>, done))                                    \
            return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_grouped: cannot reserve %zu bytes of LDS", lds);                  \
        hipLaunchKernelGGL((attn_grouped_kernel<M, F, P>), dim3(blocks), dim3(512), lds, s, pc, ldpc, pr, ldpr, A, w1, b1, rowptr, col, val, R, I, \
                           grp_ptr, pair_ids, wg_ptr, pairs_per_wg, feat, ldfeat, Fdim, out_bias, out, ldout, wts, wts_off);   \
    } while (0)
#define LAUNCH_P(M, F) do { if (pieces_per_thread <= 4) LAUNCH1(M, F, 4); else if (pieces_per_thread <= 8) LAUNCH1(M, F, 8); else LAUNCH1(M, F, 16); } while (0)
#define LAUNCH_F(M) do { if (Fdim <= 64) LAUNCH_P(M, 1); else if (Fdim <= 128) LAUNCH_P(M, 2); else LAUNCH_P(M, 4); } while (0)
    if (mode == NCF_ATT_MLP) LAUNCH_F(0);
    else LAUNCH_F(2);
#undef LAUNCH_F
#undef LAUNCH_P
#undef LAUNCH1
    return check_launch("ncf_attn_forward_grouped");
}

extern "C" size_t ncf_group_pairs_workspace_bytes(int64_t n_rows) { return (size_t)(2 * (n_rows > 0 ? n_rows : 0) + 1) * sizeof(int); }

extern "C" int ncf_group_pairs(const int64_t* pair_row, int64_t B, int64_t R, int pairs_per_wg, int64_t* grp_ptr, int64_t* pair_ids,
                               int64_t* wg_ptr, void* workspace, size_t workspace_bytes, int32_t* oob, ncf_stream_t stream) {
    return ncf_group_pairs_rows(pair_row, B, R, pairs_per_wg, grp_ptr, pair_ids, wg_ptr, nullptr, workspace, workspace_bytes, oob, stream);
}

extern "C" int ncf_group_pairs_rows(const int64_t* pair_row, int64_t B, int64_t R, int pairs_per_wg, int64_t* grp_ptr, int64_t* pair_ids,
                                    int64_t* wg_ptr, int32_t* wg_row, void* workspace, size_t workspace_bytes, int32_t* oob,
                                    ncf_stream_t stream) {
    if (B < 0 || R < 0 || pairs_per_wg < 1) return fail(NCF_EINVAL, "ncf_group_pairs: bad sizes");
    if (R >= (int64_t)1 << 31) return fail(NCF_EUNSUPPORTED, "ncf_group_pairs: more than 2^31 rows");
    if (!grp_ptr || !wg_ptr || (B > 0 && (!pair_row || !pair_ids)) || !workspace) return fail(NCF_EINVAL, "ncf_group_pairs: null pointer");
    if (B >= (int64_t)1 << 31) return fail(NCF_EUNSUPPORTED, "ncf_group_pairs: more than 2^31 pairs");
    if (workspace_bytes < ncf_group_pairs_workspace_bytes(R)) return fail(NCF_EWORKSPACE, "ncf_group_pairs: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    int* counts = (int*)workspace;
    int* cursor = counts + R;
    int* bad = oob ? oob : cursor + R;                     // the spare word of the workspace when the caller passes no flag
    if (B <= 32768 && R <= 32768) {
        if (R <= 4096) hipLaunchKernelGGL(group_small_kernel<true>, dim3(1), dim3(1024), 0, s, pair_row, B, R, pairs_per_wg, counts, cursor, bad, grp_ptr, wg_ptr, pair_ids, wg_row);
        else hipLaunchKernelGGL(group_small_kernel<false>, dim3(1), dim3(1024), 0, s, pair_row, B, R, pairs_per_wg, counts, cursor, bad, grp_ptr, wg_ptr, pair_ids, wg_row);
        return check_launch("ncf_group_pairs");
    }
    fill_u32_async(workspace, 0u, ncf_group_pairs_workspace_bytes(R), s);
    if (B > 0) hipLaunchKernelGGL(group_count_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, pair_row, B, R, counts, bad);
    hipLaunchKernelGGL(group_scan_kernel, dim3(1), dim3(1024), 0, s, counts, R, pairs_per_wg, grp_ptr, wg_ptr, cursor, wg_row);
    if (B > 0) hipLaunchKernelGGL(group_scatter_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, pair_row, B, R, cursor, pair_ids);
    return check_launch("ncf_group_pairs");
}
