// K2c — the candidate side of an AttentionNCF forward in ONE launch (models/attention_ncf.py:150 and the candidate half of
// AttentionNet.0, :176-179), fp32, gfx950:
//     cand_emb = x . Wi^T + bi            (B, K) x (N1, K)^T     K = item_dim (2094 in the reference's data), N1 = item_emb
//     pc       = cand_emb . Wc^T + b0     (B, N1) x (N2, N1)^T   N2 = att_dense, Wc = AttentionNet.0.weight[:, :N1]
// and, in a spare workgroup, the grouping of the batch's pairs by rated set (ncf_group_pairs: it depends on pair_row only).
// Round 2 ran these as three launches — a skinny split-K Linear (21.6 us: one 32-column block per workgroup, so x was read twice and
// every wave waited a full HBM round trip per 32-wide K block), a 12 us Linear for 67 MFLOP, an 8 us single-workgroup counting sort.
//
// One workgroup = 16 rows of x and ALL N1 columns (x is read once; B = 4096 gives 256 workgroups, one per CU), 8 waves = 8 K slices.
// A wave streams its slice in 32-wide blocks: global -> registers (eight lanes per row: whole 128-byte runs; rows of K floats are
// only 4-byte aligned) TWO blocks ahead, registers -> its own LDS staging area (16-byte slot s of row r at s ^ ((r >> 1) & 7)),
// LDS -> MFMA operands (v_mfma_f32_16x16x4_f32: exact fp32 fmaf chains; lane (i, g) holds k = 16h + 4g + j of row i for step (h, j):
// the same k pairing for x and W, so only the summation order inside a block is permuted).  No barrier in the loop: a wave reads
// only what it wrote.  The 8 partial tiles are added through LDS in slice order (deterministic), the 16 x N1 cand_emb tile stays in
// LDS and feeds the second product (N2 / 16 column tiles over the waves, Wc from L2).
#include "ncf_common.h"
#include "group_pairs.h"
#include <atomic>

#ifndef ATT_CAND_DIAG
#define ATT_CAND_DIAG 0   // diagnostic builds (wrong results): 1 = W blocks loaded once (no L2 re-reads), 2 = x blocks loaded once, 3 = no MFMAs,
#endif                    // 4 = no LDS staging (operands = whatever LDS holds)

namespace ncf {

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // rows of x / Wi may be only 4-byte aligned

struct CandArgs {
    const float* x; const float* Wi; const float* bi; const float* Wc; const float* b0;
    float* emb; float* pc;
    int64_t B, ldx, ldw, ldemb, ldpc;
    int K, N2;
    // grouping (pair_row == nullptr: none)
    const int64_t* pair_row; int64_t R; int ppw;
    int* gcounts; int* gcursor; int* bad; int64_t* grp_ptr; int64_t* wg_ptr; int64_t* pair_ids; int32_t* wg_row;
};

template <int N1>
__global__ __launch_bounds__(512, 2) void attn_cand_kernel(const CandArgs a) {
    constexpr int NWV = 8, TM = 16, NT = N1 / 16, WC = N1 / 8;     // column tiles; W load instructions per block
    constexpr int STW = (TM + N1) * 32;                            // floats of one wave's staging area
    constexpr int CES = N1 + 4;                                    // row stride of the cand_emb tile (bank spread)
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool grouping = a.pair_row != nullptr;
    if (grouping && blockIdx.x == 0) {                             // the spare workgroup: dispatched first, done long before the rest
        int* lds = reinterpret_cast<int*>(smem_f);
        __builtin_amdgcn_s_setprio(3);                             // it shares its CU with a tile of the product: let it go first
        if (a.R <= kGroupLdsRows)
            group_small_body<true, 512>(a.pair_row, a.B, a.R, a.ppw, a.gcounts, a.gcursor, a.bad, a.grp_ptr, a.wg_ptr, a.pair_ids, a.wg_row, lds);
        else
            group_small_body<false, 512>(a.pair_row, a.B, a.R, a.ppw, a.gcounts, a.gcursor, a.bad, a.grp_ptr, a.wg_ptr, a.pair_ids, a.wg_row, lds);
        return;
    }
    const int64_t row0 = ((int64_t)blockIdx.x - (grouping ? 1 : 0)) * TM;
    if (row0 >= a.B) return;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int lr = lane >> 3, ls = lane & 7;
    float* const stA = smem_f + (size_t)wave * STW;
    float* const stW = stA + TM * 32;
    const int K = a.K;
    const int NBF = K / 32;                                        // whole 32-wide blocks, split over the waves
    const int blo = (int)((int64_t)wave * NBF / NWV), bhi = (int)((int64_t)(wave + 1) * NBF / NWV);

    const float* ga[2];
    const float* gw[WC];
    unsigned woA[2], woW[WC];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int r = 8 * c + lr;
        const int64_t m = row0 + r < a.B ? row0 + r : a.B - 1;
        ga[c] = a.x + m * a.ldx + 4 * ls;
        woA[c] = r * 32 + ((ls ^ ((r >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int c = 0; c < WC; ++c) {
        const int r = 8 * c + lr;
        gw[c] = a.Wi + (int64_t)r * a.ldw + 4 * ls;
        woW[c] = r * 32 + ((ls ^ ((r >> 1) & 7)) << 2);
    }
    unsigned ro[2];                                                // operand reads: chunk g4 + 4h of row i16 (+ 16 nt)
#pragma unroll
    for (int h = 0; h < 2; ++h) ro[h] = i16 * 32 + ((((g4 + 4 * h) ^ ((i16 >> 1) & 7)) & 7) << 2);

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};

    bool first_load = true;
    auto gload = [&](f32x4 (&va)[2], f32x4 (&vw)[WC], int b) {
        if (ATT_CAND_DIAG != 2 || first_load) {
#pragma unroll
            for (int c = 0; c < 2; ++c) va[c] = *reinterpret_cast<const f32x4u*>(ga[c] + 32 * b);
        }
        if (ATT_CAND_DIAG != 1 || first_load) {
#pragma unroll
            for (int c = 0; c < WC; ++c) vw[c] = *reinterpret_cast<const f32x4u*>(gw[c] + 32 * b);
        }
    };
    auto stage = [&](const f32x4 (&va)[2], const f32x4 (&vw)[WC]) {
        if (ATT_CAND_DIAG == 4) { asm volatile("" ::"v"(va[0]), "v"(va[1]), "v"(vw[0]), "v"(vw[WC - 1])); return; }
#pragma unroll
        for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(stA + woA[c]) = va[c];
#pragma unroll
        for (int c = 0; c < WC; ++c) *reinterpret_cast<f32x4*>(stW + woW[c]) = vw[c];
    };
    auto compute = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(stA + ro[h]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(stW + nt * 16 * 32 + ro[h]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (ATT_CAND_DIAG == 3) acc[nt][j] += av[j] * wv[j];
                    else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc[nt], 0, 0, 0);
                }
            }
        }
    };
    if (bhi > blo) {
        f32x4 va0[2], vw0[WC], va1[2], vw1[WC];
        const int last = bhi - 1;
        gload(va0, vw0, blo);
        gload(va1, vw1, blo + 1 < bhi ? blo + 1 : last);
        first_load = false;
        for (int b = blo; b < bhi; b += 2) {                       // two blocks in flight in registers; a step's LDS traffic is its own
            stage(va0, vw0);
            gload(va0, vw0, b + 2 < bhi ? b + 2 : last);           // past the slice: the last block again (harmless, no branch)
            compute();
            if (b + 1 < bhi) {
                stage(va1, vw1);
                gload(va1, vw1, b + 3 < bhi ? b + 3 : last);
                compute();
            }
        }
    }
    if (wave == 0 && (K & 31)) {                                   // ragged end of K: guarded scalar loads, zero fill (one block, one wave)
        const int k0 = 32 * NBF + 4 * ls;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = k0 + j < K ? ga[c][32 * NBF + j] : 0.f;
            *reinterpret_cast<f32x4*>(stA + woA[c]) = v;
        }
#pragma unroll
        for (int c = 0; c < WC; ++c) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = k0 + j < K ? gw[c][32 * NBF + j] : 0.f;
            *reinterpret_cast<f32x4*>(stW + woW[c]) = v;
        }
        compute();
    }
    // ---- the 8 K-slices added in slice order ----
    __syncthreads();                                               // every wave is done with its staging area
    float* const red = smem_f;                                     // [NWV][TM][N1]
    float* const ce = smem_f + NWV * TM * N1;                      // [TM][CES]  the cand_emb tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * TM + 4 * g4 + i) * N1 + 16 * nt + i16] = acc[nt][i];
    __syncthreads();
    for (int o = tid; o < TM * N1; o += 512) {
        const int r = o / N1, c = o - r * N1;
        float v = red[o];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += red[w * TM * N1 + o];
        v += a.bi ? a.bi[c] : 0.f;
        ce[r * CES + c] = v;
        if (row0 + r < a.B) a.emb[(row0 + r) * a.ldemb + c] = v;
    }
    __syncthreads();
    // ---- pc tile = cand_emb tile . Wc^T + b0 ----
    for (int ct = wave; ct < a.N2 / 16; ct += NWV) {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        const float* wrow = a.Wc + (int64_t)(16 * ct + i16) * N1 + 4 * g4;
#pragma unroll
        for (int kk = 0; kk < N1 / 16; ++kk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ce + i16 * CES + 16 * kk + 4 * g4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc2, 0, 0, 0);
        }
        const float bv = a.b0 ? a.b0[16 * ct + i16] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = row0 + 4 * g4 + i;
            if (m < a.B) a.pc[m * a.ldpc + 16 * ct + i16] = acc2[i] + bv;
        }
    }
}

// ---- Wi pre-packed in MFMA operand order (round 3, second form) ---------------------------------------------------------------
// Why: the kernel above needs 148 VGPRs and 80 KB of LDS — ONE workgroup per CU — so the 257th workgroup of a 4096-row batch (256
// row tiles + the grouping workgroup) cannot start before the grouping workgroup has left its CU: 21.8 us became 24.0-24.7.  With
// Wi packed once per weight version so that every load instruction is one contiguous 1 KB run that lands in the MFMA's B-operand
// registers as it is — Wp[(((s NT + nt) 2 + h) 64 + lane) 4 + j] = Wi[16 nt + (lane & 15)][32 s + 16 h + 4 (lane >> 4) + j], zero past
// K — 80 % of the operand bytes skip the LDS round trip: <= 128 VGPRs, 37 KB (N1 = 64) / 74 KB (N1 = 128) of LDS, two workgroups per
// CU, and the grouping workgroup runs BESIDE a row tile.  x still goes through a per-wave LDS area (coalesced 128-byte runs in,
// operand order out: a direct operand-order load touches 16 rows x 64 B per instruction and measured slower).
// A step = 32 k; a sub-step = (step, 64-column half of N1): D = 2 sub-steps in flight per wave (N1 = 64: two steps; N1 = 128: the two
// halves of one step, x re-read from L1 for the second half).  Same k pairing, slices and summation order as above: bit-identical results.
__global__ __launch_bounds__(256) void cand_pack_kernel(const float* __restrict__ W, int64_t ldw, int K, int N1, int S, float* __restrict__ Wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int NT = N1 / 16;
    if (idx >= (int64_t)S * NT * 512) return;
    const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63), h = (int)((idx >> 8) & 1);
    const int64_t q = idx >> 9;
    const int nt = (int)(q % NT), s = (int)(q / NT);
    const int n = 16 * nt + (lane & 15), k = 32 * s + 16 * h + 4 * (lane >> 4) + j;
    Wp[idx] = k < K ? W[(int64_t)n * ldw + k] : 0.f;
}

template <int N1>
__global__ __launch_bounds__(512, 4) void attn_cand_packed_kernel(const CandArgs a) {
    constexpr int NWV = 8, TM = 16, NT = N1 / 16, NH = N1 / 64, D = 2;
    constexpr int CES = N1 + 4;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool grouping = a.pair_row != nullptr;
    if (grouping && blockIdx.x == 0) {                             // shares its CU with a row tile now (priority 3 changes nothing: 20.2 vs 20.3 us)
        int* lds = reinterpret_cast<int*>(smem_f);
        if (a.R <= kGroupLdsRows)
            group_small_body<true, 512>(a.pair_row, a.B, a.R, a.ppw, a.gcounts, a.gcursor, a.bad, a.grp_ptr, a.wg_ptr, a.pair_ids, a.wg_row, lds);
        else
            group_small_body<false, 512>(a.pair_row, a.B, a.R, a.ppw, a.gcounts, a.gcursor, a.bad, a.grp_ptr, a.wg_ptr, a.pair_ids, a.wg_row, lds);
        return;
    }
    const int64_t row0 = ((int64_t)blockIdx.x - (grouping ? 1 : 0)) * TM;
    if (row0 >= a.B) return;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int lr = lane >> 3, ls = lane & 7;
    const int K = a.K;
    const int SF = K / 32;                                         // whole 32-wide steps, split over the waves exactly as above
    const int s_lo = (int)((int64_t)wave * SF / NWV), e = (int)((int64_t)(wave + 1) * SF / NWV);
    float* const stA = smem_f + (size_t)wave * (D * TM * 32);      // this wave's x staging: D slots of 16 rows x 32 k
    const float* xc[2];
    unsigned woA[2], ro[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int r = 8 * c + lr;
        const int64_t m = row0 + r < a.B ? row0 + r : a.B - 1;
        xc[c] = a.x + m * a.ldx + 4 * ls;
        woA[c] = r * 32 + ((ls ^ ((r >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) ro[h] = i16 * 32 + ((((g4 + 4 * h) ^ ((i16 >> 1) & 7)) & 7) << 2);
    const f32x4* const wa = reinterpret_cast<const f32x4*>(a.Wi) + lane;   // a.Wi = the packed copy

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 va[D][2], vw[D][4][2];
    // slot d holds sub-step q (q - q_lo = d mod D): step q / NH, half q % NH.  q_lo is a multiple of NH = D or 1, so slot d is
    // always half (NH == 2 ? d : 0): the accumulator index below is a compile-time constant.
    auto load = [&](int d, int q) {
        const int s = NH == 2 ? q >> 1 : q;
#pragma unroll
        for (int c = 0; c < 2; ++c) va[d][c] = *reinterpret_cast<const f32x4u*>(xc[c] + 32 * s);
        const int hf = NH == 2 ? (q & 1) : 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) vw[d][t][h] = wa[(((int64_t)s * NT + 4 * hf + t) * 2 + h) * 64];
    };
    auto compute = [&](int d) {
        float* const st = stA + d * (TM * 32);
#pragma unroll
        for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(st + woA[c]) = va[d][c];
        f32x4 av[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) av[h] = *reinterpret_cast<const f32x4*>(st + ro[h]);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int nt = (NH == 2 ? 4 * d : 0) + t;      // d is a literal at every call site
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][j], vw[d][t][h][j], acc[nt], 0, 0, 0);
                }
    };
    const int q_lo = s_lo * NH, q_hi = e * NH;
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (q_lo + d < q_hi) load(d, q_lo + d);
    for (int q = q_lo; q < q_hi; q += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (q + d < q_hi) {                                    // wave-uniform
                compute(d);
                if (q + d + D < q_hi) load(d, q + d + D);
            }
        }
    }
    if (wave == 0 && (K & 31)) {                                   // ragged end of K (wave 0, as above): guarded loads of x; the packed W is zero past K
        const int64_t m = row0 + i16 < a.B ? row0 + i16 : a.B - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 32 * SF + 16 * h + 4 * g4 + j;
                v[j] = k < K ? a.x[m * a.ldx + k] : 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 w = wa[(((int64_t)SF * NT + nt) * 2 + h) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], w[j], acc[nt], 0, 0, 0);
            }
        }
    }
    // ---- the 8 K-slices added in slice order; pc tile = cand_emb tile . Wc^T + b0 (as above) ----
    __syncthreads();
    float* const red = smem_f;                                     // [NWV][TM][N1]
    float* const ce = smem_f + NWV * TM * N1;                      // [TM][CES]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * TM + 4 * g4 + i) * N1 + 16 * nt + i16] = acc[nt][i];
    __syncthreads();
    for (int o = tid; o < TM * N1; o += 512) {
        const int r = o / N1, c = o - r * N1;
        float v = red[o];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += red[w * TM * N1 + o];
        v += a.bi ? a.bi[c] : 0.f;
        ce[r * CES + c] = v;
        if (row0 + r < a.B) a.emb[(row0 + r) * a.ldemb + c] = v;
    }
    __syncthreads();
    for (int ct = wave; ct < a.N2 / 16; ct += NWV) {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        const float* wrow = a.Wc + (int64_t)(16 * ct + i16) * N1 + 4 * g4;
#pragma unroll
        for (int kk = 0; kk < N1 / 16; ++kk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ce + i16 * CES + 16 * kk + 4 * g4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc2, 0, 0, 0);
        }
        const float bv = a.b0 ? a.b0[16 * ct + i16] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = row0 + 4 * g4 + i;
            if (m < a.B) a.pc[m * a.ldpc + 16 * ct + i16] = acc2[i] + bv;
        }
    }
}

// ---- small batches: the K range split over WORKGROUPS (packed weights) ----------------------------------------------------------
// Every workgroup of the kernels above pulls the whole of Wi through its CU's L2 port (N1 = 128: 1 MB at ~70 GB/s = 15 us), however
// few row tiles there are: at the reference's evaluation batch (512 rows = 32 tiles) 32 CUs do that while 224 idle (32.8 us).  Here
// a row tile's K range is cut into KS pieces of 8 wave slices each (KS * tiles <= 256 workgroups): a workgroup ingests 1 / KS of Wi,
// leaves its partial 16 x N1 tile in the workspace, and a second, short launch adds the KS partials in order (+ bias), writes emb,
// and runs the second product (and the grouping workgroup).  Deterministic; another summation order than the one-launch forms.
template <int N1>
__global__ __launch_bounds__(512) void cand_splitk_kernel(const CandArgs a, float* __restrict__ part, int KS) {
    constexpr int NWV = 8, TM = 16, NT = N1 / 16;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TM;
    const int ks = blockIdx.y;
    const int K = a.K, SF = K / 32;
    const int slice = ks * NWV + wave, nsl = KS * NWV;
    const int s_lo = (int)((int64_t)slice * SF / nsl), s_hi = (int)((int64_t)(slice + 1) * SF / nsl);
    const int64_t m = row0 + i16 < a.B ? row0 + i16 : a.B - 1;
    const float* xa = a.x + m * a.ldx + 4 * g4;                    // operand-order loads (16 rows x 64 B per instruction): one or two steps per wave
    const f32x4* const wa = reinterpret_cast<const f32x4*>(a.Wi) + lane;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = s_lo; s < s_hi; ++s) {
        f32x4 av[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) av[h] = *reinterpret_cast<const f32x4u*>(xa + 32 * s + 16 * h);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 w[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) w[h] = wa[(((int64_t)s * NT + nt) * 2 + h) * 64];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][j], w[h][j], acc[nt], 0, 0, 0);
        }
    }
    if (slice == 0 && (K & 31)) {                                  // ragged end of K: guarded loads of x; the packed W is zero past K
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 32 * SF + 16 * h + 4 * g4 + j;
                v[j] = k < K ? a.x[m * a.ldx + k] : 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 w = wa[(((int64_t)SF * NT + nt) * 2 + h) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], w[j], acc[nt], 0, 0, 0);
            }
        }
    }
    float* const red = smem_f;                                     // [NWV][TM][N1]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * TM + 4 * g4 + i) * N1 + 16 * nt + i16] = acc[nt][i];
    __syncthreads();
    float* const dst = part + ((int64_t)ks * gridDim.x + blockIdx.x) * (TM * N1);
    for (int o = tid; o < TM * N1; o += 512) {
        float v = red[o];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += red[w * TM * N1 + o];
        dst[o] = v;
    }
}

template <int N1>
__global__ __launch_bounds__(512) void cand_finish_kernel(const CandArgs a, const float* __restrict__ part, int KS, int tiles) {
    constexpr int NWV = 8, TM = 16, CES = N1 + 4;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool grouping = a.pair_row != nullptr;
    if (grouping && blockIdx.x == 0) {
        int* lds = reinterpret_cast<int*>(smem_f);
        if (a.R <= kGroupLdsRows)
            group_small_body<true, 512>(a.pair_row, a.B, a.R, a.ppw, a.gcounts, a.gcursor, a.bad, a.grp_ptr, a.wg_ptr, a.pair_ids, a.wg_row, lds);
        else
            group_small_body<false, 512>(a.pair_row, a.B, a.R, a.ppw, a.gcounts, a.gcursor, a.bad, a.grp_ptr, a.wg_ptr, a.pair_ids, a.wg_row, lds);
        return;
    }
    const int tile = (int)blockIdx.x - (grouping ? 1 : 0);
    const int64_t row0 = (int64_t)tile * TM;
    if (row0 >= a.B) return;
    const int i16 = lane & 15, g4 = lane >> 4;
    float* const ce = smem_f;                                      // [TM][CES]  the cand_emb tile
    for (int o = tid; o < TM * N1; o += 512) {
        const int r = o / N1, c = o - r * N1;
        float pv[8];                                               // all pieces requested at once (KS <= 8), added in K order
#pragma unroll
        for (int k = 0; k < 8; ++k) pv[k] = part[((int64_t)(k < KS ? k : KS - 1) * tiles + tile) * (TM * N1) + o];
        float v = pv[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) v += k < KS ? pv[k] : 0.f;
        v += a.bi ? a.bi[c] : 0.f;
        ce[r * CES + c] = v;
        if (row0 + r < a.B) a.emb[(row0 + r) * a.ldemb + c] = v;
    }
    __syncthreads();
    for (int ct = wave; ct < a.N2 / 16; ct += NWV) {
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};
        const float* wrow = a.Wc + (int64_t)(16 * ct + i16) * N1 + 4 * g4;
#pragma unroll
        for (int kk = 0; kk < N1 / 16; ++kk) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(ce + i16 * CES + 16 * kk + 4 * g4);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[j], acc2, 0, 0, 0);
        }
        const float bv = a.b0 ? a.b0[16 * ct + i16] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t m = row0 + 4 * g4 + i;
            if (m < a.B) a.pc[m * a.ldpc + 16 * ct + i16] = acc2[i] + bv;
        }
    }
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_attn_candidates_supported(int K, int N1, int N2) {
    return K >= 1 && (N1 == 64 || N1 == 128) && N2 >= 16 && N2 % 16 == 0 && N2 <= 256;
}

extern "C" size_t ncf_attn_candidates_workspace_bytes(int64_t n_rows) { return (size_t)(2 * (n_rows > 0 ? n_rows : 0) + 1) * sizeof(int); }

namespace {
// pieces of the K range a row tile is cut into (packed form): 1 = one launch; 2..8 when the batch has at most 128 row tiles
int cand_splitk_pieces(int64_t B) {
    const int64_t tiles = (B + 15) / 16;
    if (tiles < 1 || tiles > 128) return 1;
    const int64_t ks = 256 / tiles;
    return ks > 8 ? 8 : (int)ks;
}
size_t cand_partial_bytes(int64_t B, int N1) {
    const int ks = cand_splitk_pieces(B);
    return ks > 1 ? (size_t)ks * (size_t)((B + 15) / 16) * 16 * (size_t)N1 * sizeof(float) : 0;
}

int cand_launch(bool packed, const float* x, int64_t B, int64_t ldx, int K, const float* Wi, int64_t ldw, const float* bi, int N1,
                const float* Wc, const float* b0, int N2, float* emb, int64_t ldemb, float* pc, int64_t ldpc,
                const int64_t* pair_row, int64_t R, int pairs_per_wg, int64_t* grp_ptr, int64_t* pair_ids,
                int64_t* wg_ptr, int32_t* wg_row, void* workspace, size_t workspace_bytes, int32_t* oob, ncf_stream_t stream) {
    const char* who = packed ? "ncf_attn_candidates_packed" : "ncf_attn_candidates";
    if (!ncf_attn_candidates_supported(K, N1, N2))
        return fail(NCF_EUNSUPPORTED, "%s: needs item_emb 64 or 128 and att_dense %% 16 == 0, <= 256 (N1 = %d, N2 = %d)", who, N1, N2);
    if (B < 0 || ldx < K || (!packed && ldw < K) || ldemb < N1 || ldpc < N2) return fail(NCF_EINVAL, "%s: bad sizes", who);
    if (B == 0) return NCF_OK;
    if (!x || !Wi || !Wc || !emb || !pc) return fail(NCF_EINVAL, "%s: null pointer", who);
    if (!aligned16(Wc)) return fail(NCF_EINVAL, "%s: Wc must be 16-byte aligned (contiguous (N2, N1))", who);
    if (packed && !aligned16(Wi)) return fail(NCF_EINVAL, "%s: the packed weights must be 16-byte aligned", who);
    // packed form: [partial tiles of the split-K path | grouping ints]
    const int KS = packed ? cand_splitk_pieces(B) : 1;
    const size_t part_bytes = packed ? cand_partial_bytes(B, N1) : 0;
    if (part_bytes && (!workspace || workspace_bytes < part_bytes || !aligned16(workspace)))
        return fail(NCF_EWORKSPACE, "%s: workspace too small (ncf_attn_candidates_packed_workspace_bytes) or misaligned", who);
    float* const part = (float*)workspace;
    if (part_bytes) { workspace = (char*)workspace + part_bytes; workspace_bytes -= part_bytes; }
    CandArgs a{};
    a.x = x; a.Wi = Wi; a.bi = bi; a.Wc = Wc; a.b0 = b0; a.emb = emb; a.pc = pc;
    a.B = B; a.ldx = ldx; a.ldw = ldw; a.ldemb = ldemb; a.ldpc = ldpc; a.K = K; a.N2 = N2;
    if (pair_row) {
        if (R < 0 || pairs_per_wg < 1 || B > 32768 || R > 32768)
            return fail(NCF_EUNSUPPORTED, "%s: the fused grouping takes B, n_rows <= 32768 (use ncf_group_pairs_rows)", who);
        if (!grp_ptr || !pair_ids || !wg_ptr || !workspace) return fail(NCF_EINVAL, "%s: null grouping pointer", who);
        if (workspace_bytes < ncf_attn_candidates_workspace_bytes(R)) return fail(NCF_EWORKSPACE, "%s: workspace too small", who);
        a.pair_row = pair_row; a.R = R; a.ppw = pairs_per_wg;
        a.gcounts = (int*)workspace; a.gcursor = a.gcounts + R;
        a.bad = oob ? oob : a.gcursor + R;
        a.grp_ptr = grp_ptr; a.wg_ptr = wg_ptr; a.pair_ids = pair_ids; a.wg_row = wg_row;
    }
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((B + 15) / 16) + (pair_row ? 1u : 0u);
    // staging (LDS-staged form: 8 waves x (16 + N1) rows x 32 k; packed form: 8 waves x 2 slots x 16 rows x 32 k of x only);
    // the reduction (8 x 16 x N1) + the cand_emb tile reuse the same bytes
    size_t lds = packed ? (size_t)8 * 2 * 16 * 32 * 4 : (size_t)8 * (16 + N1) * 32 * 4;
    const size_t lds_red = ((size_t)8 * 16 * N1 + 16 * (N1 + 4)) * 4;
    const size_t lds_grp = (size_t)group_small_lds_ints<true>(512) * sizeof(int);
    if (lds_red > lds) lds = lds_red;
    if (pair_row && lds_grp > lds) lds = lds_grp;
    auto raise = [&](const void* fn, std::atomic<unsigned long long>& done) -> bool {
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
#define NCF_CAND_LAUNCH(KERNEL)                                                                                                   \
    {                                                                                                                             \
        static std::atomic<unsigned long long> done{0};                                                                           \
        if (!raise((const void*)KERNEL, done)) return fail(NCF_EUNSUPPORTED, "%s: cannot reserve %zu bytes of LDS", who, lds);     \
        hipLaunchKernelGGL(KERNEL, dim3(blocks), dim3(512), lds, s, a);                                                           \
    }
    if (packed && KS > 1) {                                        // few row tiles: K range over workgroups, then a short finishing launch
        const unsigned tiles = (unsigned)((B + 15) / 16);
        const size_t lds_a = (size_t)8 * 16 * N1 * 4;
        size_t lds_b = (size_t)16 * (N1 + 4) * 4;
        if (pair_row && lds_grp > lds_b) lds_b = lds_grp;
#define NCF_CAND_SPLITK(N)                                                                                                        \
    {                                                                                                                             \
        static std::atomic<unsigned long long> done{0};                                                                           \
        if (!raise((const void*)cand_splitk_kernel<N>, done)) return fail(NCF_EUNSUPPORTED, "%s: cannot reserve %zu bytes of LDS", who, lds_a); \
        hipLaunchKernelGGL(cand_splitk_kernel<N>, dim3(tiles, (unsigned)KS), dim3(512), lds_a, s, a, part, KS);                   \
        hipLaunchKernelGGL(cand_finish_kernel<N>, dim3(blocks), dim3(512), lds_b, s, a, (const float*)part, KS, (int)tiles);      \
    }
        if (N1 == 64) NCF_CAND_SPLITK(64) else NCF_CAND_SPLITK(128)
#undef NCF_CAND_SPLITK
    } else if (packed) {
        if (N1 == 64) NCF_CAND_LAUNCH(attn_cand_packed_kernel<64>) else NCF_CAND_LAUNCH(attn_cand_packed_kernel<128>)
    } else {
        if (N1 == 64) NCF_CAND_LAUNCH(attn_cand_kernel<64>) else NCF_CAND_LAUNCH(attn_cand_kernel<128>)
    }
#undef NCF_CAND_LAUNCH
    return check_launch(who);
}
}  // namespace

extern "C" int ncf_attn_candidates(const float* x, int64_t B, int64_t ldx, int K, const float* Wi, int64_t ldw, const float* bi, int N1,
                                   const float* Wc, const float* b0, int N2, float* emb, int64_t ldemb, float* pc, int64_t ldpc,
                                   const int64_t* pair_row, int64_t R, int pairs_per_wg, int64_t* grp_ptr, int64_t* pair_ids,
                                   int64_t* wg_ptr, int32_t* wg_row, void* workspace, size_t workspace_bytes, int32_t* oob,
                                   ncf_stream_t stream) {
    return cand_launch(false, x, B, ldx, K, Wi, ldw, bi, N1, Wc, b0, N2, emb, ldemb, pc, ldpc, pair_row, R, pairs_per_wg, grp_ptr, pair_ids,
                       wg_ptr, wg_row, workspace, workspace_bytes, oob, stream);
}

/* floats of the packed copy of a (N1, K) ItemEmbeddings weight */
extern "C" size_t ncf_attn_candidates_pack_floats(int K, int N1) { return K > 0 && N1 > 0 ? (size_t)((K + 31) / 32) * 32 * (size_t)N1 : 0; }

extern "C" int ncf_attn_candidates_pack(const float* Wi, int64_t ldw, int K, int N1, float* packed, ncf_stream_t stream) {
    if (K < 1 || (N1 != 64 && N1 != 128) || ldw < K) return fail(NCF_EINVAL, "ncf_attn_candidates_pack: bad sizes (N1 = 64 or 128)");
    if (!Wi || !packed || !aligned16(packed)) return fail(NCF_EINVAL, "ncf_attn_candidates_pack: null or misaligned pointer");
    const int S = (K + 31) / 32;
    const int64_t n = (int64_t)S * (N1 / 16) * 512;
    hipLaunchKernelGGL(cand_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, Wi, ldw, K, N1, S, packed);
    return check_launch("ncf_attn_candidates_pack");
}

/* workspace of ncf_attn_candidates_packed: the split-K path's partial tiles (batches of at most 128 row tiles) + the grouping's
 * counters (n_rows < 0: no grouping) */
extern "C" size_t ncf_attn_candidates_packed_workspace_bytes(int64_t B, int N1, int64_t n_rows) {
    return cand_partial_bytes(B > 0 ? B : 0, N1) + (n_rows >= 0 ? ncf_attn_candidates_workspace_bytes(n_rows) : 0);
}

extern "C" int ncf_attn_candidates_packed(const float* x, int64_t B, int64_t ldx, int K, const float* Wi_packed, const float* bi, int N1,
                                          const float* Wc, const float* b0, int N2, float* emb, int64_t ldemb, float* pc, int64_t ldpc,
                                          const int64_t* pair_row, int64_t R, int pairs_per_wg, int64_t* grp_ptr, int64_t* pair_ids,
                                          int64_t* wg_ptr, int32_t* wg_row, void* workspace, size_t workspace_bytes, int32_t* oob,
                                          ncf_stream_t stream) {
    return cand_launch(true, x, B, ldx, K, Wi_packed, 0, bi, N1, Wc, b0, N2, emb, ldemb, pc, ldpc, pair_row, R, pairs_per_wg, grp_ptr, pair_ids,
                       wg_ptr, wg_row, workspace, workspace_bytes, oob, stream);
}
