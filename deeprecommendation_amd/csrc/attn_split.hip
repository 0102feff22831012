// K3 grouped, ENTRY-SPLIT form (round 3): AttentionNCF item-item attention (models/attention_ncf.py:154-216, eval mode) for pairs
// that share rated sets, fp32, gfx950.
//
// The scalar-operand grouped kernel (attn.hip, round 2) gives one workgroup up to 32 pairs of ONE user and walks the user's whole
// rated set tile by tile: a batch of 4096 pairs is 1024 four-pair wave jobs — one wave per SIMD on the chip — each a serial chain
// of 4 tiles x (DMA wait, 3 barriers, ~1400 instructions), at 256 VGPRs with 22 of them spilled to scratch.  Measured 36 us where
// the arithmetic is a few us of the chip: the kernel is bound by its own critical path, not by a unit.
// Here the unit of work is (group of pairs) x (a SLICE of the rated set): gridDim.y = nsplit slices, each workgroup runs the same
// tile pipeline over its tiles only and leaves a softmax PARTIAL per pair — running maximum m, sum l and the un-normalised
// aggregate O (Fdim floats) — which attn_combine_kernel merges:  out = sum_s O_s e^(m_s - M) / sum_s l_s e^(m_s - M) + bias.
// Four times the workgroups at a quarter of the chain; and the kernel is built for <= 128 VGPRs (row blocks of 8 chunks, double
// buffered, instead of the lane's whole 128-float row), so two 512-thread workgroups share a CU — four waves per SIMD cover each
// other's scalar-load, LDS and DMA latencies.  No spills: the arguments travel as one struct (read on demand from the kernarg
// segment), the workgroup -> row map is an array written by the grouping pass (the binary search over wg_ptr was 6 dependent
// global loads before the first useful instruction), and nothing zero-fills LDS (masked entries gather row 0 and weigh 0).
//   * lane = entry of the 64-entry tile; a wave scores its 4 pairs against the lane's row block with pc rows and w1 as SCALAR
//     operands of packed fp32 ops (s_load -> SGPR pairs; relu = the clamp modifier of the packed add on 2^-64-scaled operands);
//   * tiles arrive by LDS-DMA (global_load_lds_dwordx4, source-swizzled for conflict-free row reads);
//   * the aggregation over the tile's entries runs on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 fmaf chains).
// Same arithmetic per (pair, entry) as the round-2 kernel; the split only changes where the softmax partials are merged, i.e. the
// fp32 summation order (held to 1e-5 against the per-pair kernel and the oracle).  Fixed split: bitwise reproducible run to run.
#include "ncf_common.h"
#include "attn_util.h"
#include <math.h>
#include <atomic>
#include <type_traits>

#ifndef ATT_SPLIT_DIAG
#define ATT_SPLIT_DIAG 0   // diagnostic builds (wrong results): 1 = every slot reads pc row 0 (scalar loads all hit), 2 = no tile DMA,
#endif                     // 3 = no score loop, 4 = no aggregation, 5 = 2 + 3 + 4, 6 = exit after the prologue
#ifndef ATT_SPLIT_G
#define ATT_SPLIT_G 1      // 16-byte chunks of pc (per pair) and of w1 per scalar-load step: 1 = 20 SGPRs per step and buffer, 2 = 40
#endif
#ifndef ATT_SPLIT_PF
#define ATT_SPLIT_PF 0     // 1 = the NEXT row block's pc / w1 lines are touched (one scalar load per 64-byte line) when this block starts:
                           // measured SLOWER (28.2 vs 25.7 us at cfg 3): the extra wait per block costs more than the misses it gathers
#endif
#ifndef ATT_SPLIT_DB
#define ATT_SPLIT_DB 0   // 1: row blocks double buffered in registers (64 VGPRs of rows: spills at the 128-register budget)
#endif

namespace ncf {

struct AttnSplitArgs {
    const float* pc; const float* pr; const float* w1; const float* feat; const float* out_bias;
    const int64_t* rowptr; const int32_t* col; const float* val;
    const int64_t* grp_ptr; const int64_t* pair_ids; const int64_t* wg_ptr; const int32_t* wg_row;
    float* out; float* part;
    int64_t R, I;
    int ldpc, ldpr, ldfeat, ldout, A, Fdim, ppw, nsplit, ldpart;
    float b1;
};

// pc rows and w1 are read-only for the whole launch and wave-uniform: through the CONSTANT address space hipcc reads them with
// scalar loads (through a generic pointer of a by-value struct it cannot prove the memory invariant and emits vector loads)
typedef const f32x4 __attribute__((address_space(4))) * const_f32x4_ptr;

constexpr int kPartHead = 4;   // floats in front of a partial's O: m, l, (2 unused: O stays 16-byte aligned)

// MODE 0: MLP (relu + w1 dot), 2: cosine (dot of normalised rows), 3: MLP on 2^-64-scaled operands (relu = clamp)
// NW = 4 / 8 / 16 waves x 4 pairs (a user's pairs share the staged tile: 64 pairs per workgroup halve the gathered bytes per pair of
// 32, and a (user, tile) unit of a 64-pairs-per-user batch is ONE workgroup per CU instead of two or three); FD = Fdim (compile-time: the aggregation's B-operand reads walk the feat image with a stride of Fdim floats —
// with a run-time stride hipcc keeps 16 precomputed addresses per lane, and spills them)
template <int MODE, int NW, int FD>
__global__ __launch_bounds__(64 * NW, 4) void attn_split_kernel(const AttnSplitArgs a) {
    constexpr int EC = 64, MAXP = 4, PP = MAXP * NW, MT = PP / 16, PS = 66, CB = 8;
    constexpr int Fdim = FD, F4 = FD / 4, NTILES = FD / 16, NJ = (MT * NTILES + NW - 1) / NW;   // NJ = 16x16 aggregation tiles per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int A = a.A, A4 = A >> 2;
    float* prt = reinterpret_cast<float*>(smem);           // [EC][A]     pr tile image (DMA)
    float* fct = prt + EC * A;                              // [EC][Fdim]  feat tile image (DMA)
    float* Pm = fct + EC * Fdim;                            // [PP][PS]    p_e * val_e of the current tile, pair-major
    float* scl = Pm + PP * PS;                              // [PP]        this tile's rescale factor per pair
    int64_t* pid = reinterpret_cast<int64_t*>(scl + PP);    // [PP]        output row of each pair of the group
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t g = blockIdx.x;
    const int split = blockIdx.y;
    if (g >= a.wg_ptr[a.R]) return;                        // the grid is an upper bound (no host sync for its size)
    int64_t r;
    if (a.wg_row) {
        r = a.wg_row[g];
    } else {                                               // row r with wg_ptr[r] <= g < wg_ptr[r+1]
        int64_t lo = 0, hi = a.R;
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (a.wg_ptr[mid] <= g) lo = mid; else hi = mid;
        }
        r = lo;
    }
    const int64_t start = a.grp_ptr[r] + (g - a.wg_ptr[r]) * a.ppw;
    const int64_t left = a.grp_ptr[r + 1] - start;
    const int cnt = (int)(left < a.ppw ? left : a.ppw);
    const int64_t rbeg = a.rowptr[r], rend = a.rowptr[r + 1];
    // this workgroup's slice of the row's tiles
    const int64_t row_tiles = (rend - rbeg + EC - 1) / EC;
    const int64_t t0 = split * row_tiles / a.nsplit, t1 = (split + 1) * row_tiles / a.nsplit;
    const int64_t beg = rbeg + t0 * EC;
    const int64_t end = rbeg + t1 * EC < rend ? rbeg + t1 * EC : rend;
    const int64_t ntiles = t1 - t0;

    // the wave's pairs: slots j = wave + NW*k; their pc rows are wave-uniform pointers (scalar loads)
    const int np = cnt > wave ? (cnt - wave + NW - 1) / NW : 0;
    const float* pcrow[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) {
        const int j = k < np ? wave + NW * k : 0;
        const int64_t b = a.pair_ids[start + j];
        const int blo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffff)), bhi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
        pcrow[k] = a.pc + (((int64_t)bhi << 32) | (unsigned)blo) * a.ldpc;
#if ATT_SPLIT_DIAG == 1
        pcrow[k] = a.pc;
#endif
    }
    if (tid < PP) pid[tid] = tid < cnt ? a.pair_ids[start + tid] : -1;

    const bool swz = (A4 % 16) == 0;
    const int shA = (A4 & (A4 - 1)) == 0 ? __builtin_ctz(A4) : -1;
    constexpr int shF = (F4 & (F4 - 1)) == 0 ? __builtin_ctz(F4) : -1;
    // the tile's own entry of this lane: column and rating, loaded ahead of their use, unconditionally (index clamped into the
    // slice) and back to back.  A masked entry (outside the slice / the catalogue) gathers row 0 and weighs 0.
    auto load_cv = [&](int64_t e0, int& c, float& v) {
        const int64_t e = e0 + lane < end ? e0 + lane : (end > beg ? end - 1 : beg);
        c = end > beg ? a.col[e] : -1;
        v = end > beg ? a.val[e] : 0.f;
    };
    auto valid_c = [&](int64_t e0, int c) { return (e0 + lane < end && c >= 0 && c < a.I) ? c : -1; };
    auto issue = [&](const float* tab, int ld, int X4, int shX, bool xorj, unsigned lds_base, int cols) {
        const int rpp = shX >= 0 ? 64 >> shX : 0;          // whole rows per piece (one wave-instruction, 64 chunks)
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
                if (ATT_SPLIT_DIAG != 2 && ATT_SPLIT_DIAG < 5) dma16(tab + (int64_t)(ci >= 0 ? ci : 0) * ld + 4 * jj, lds_base + (unsigned)piece * 1024u);
            }
            return;
        }
        for (int piece = wave; piece < X4; piece += NW) {  // general row widths: (entry, chunk) by division, col by a shuffle
            const int gi = piece * 64 + lane;
            const int e = shX >= 0 ? gi >> shX : gi / X4;
            const int j = shX >= 0 ? gi & (X4 - 1) : gi - e * X4;
            const int ci = __shfl(cols, e);
            dma16(tab + (int64_t)(ci >= 0 ? ci : 0) * ld + 4 * (xorj ? (j ^ (e & 15)) : j), lds_base + (unsigned)piece * 1024u);
        }
    };
    auto issue_pr = [&](int cols) { issue(a.pr, a.ldpr, A4, shA, swz, lds0, cols); };
    auto issue_feat = [&](int cols) { issue(a.feat, a.ldfeat, F4, shF, false, lds0 + (unsigned)(EC * A * 4), cols); };

    f32x4 acc[NJ];
#pragma unroll
    for (int i = 0; i < NJ; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m[MAXP], l[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) { m[k] = -INFINITY; l[k] = 0.f; }

    int c_cur = -1, c_nxt = -1;
    float v_cur = 0.f, v_nxt = 0.f;
    if (ntiles > 0) {
        load_cv(beg, c_cur, v_cur);
        load_cv(beg + EC, c_nxt, v_nxt);
        c_cur = valid_c(beg, c_cur);
        c_nxt = valid_c(beg + EC, c_nxt);
        issue_pr(c_cur);
        issue_feat(c_cur);
    }
    const int sw = swz ? (lane & 15) : 0;
    const int g4 = lane >> 4, i16 = lane & 15;
    const float* myrow = prt + lane * A;
    const int NB = A4 / CB;
    // Scalar-cache prefetch (ATT_SPLIT_PF, off: measured slower).  A workgroup's pc rows are read exactly once, so every 64-byte line is a
    // compulsory scalar-cache miss, and with all loads hitting (ablation) the kernel is 5.5 us shorter.  touch_block(b) reads one 16-byte
    // chunk of every line of row block b with ordinary, compiler-visible scalar loads and throws the values away, so that a block's misses
    // are taken together.  It did not pay: 28.2 vs 25.7 us.  (A first attempt used inline-asm s_load_dwordx16 into "reserved" SGPRs: hipcc
    // spilled and reused them under the loads in flight and the kernel hung — scalar loads the compiler cannot see are not an option.)
    auto touch_block = [&](int blk) {
#if ATT_SPLIT_PF
        f32x4 pf[2 * MAXP + 2];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const float* rowp = pcrow[k < np ? k : 0];
            pf[2 * k] = *(const_f32x4_ptr)(rowp + 32 * blk);
            pf[2 * k + 1] = *(const_f32x4_ptr)(rowp + 32 * blk + 16);
        }
        const float* wp = MODE != 2 ? a.w1 : pcrow[0];
        pf[2 * MAXP] = *(const_f32x4_ptr)(wp + 32 * blk);
        pf[2 * MAXP + 1] = *(const_f32x4_ptr)(wp + 32 * blk + 16);
#pragma unroll
        for (int i = 0; i < 2 * MAXP + 2; ++i) asm volatile("" ::"s"(pf[i][0]));   // a use: the loads stay, their wait sits here
#endif
    };
    for (int64_t t = 0; t < ntiles; ++t) {
        const int64_t e0 = beg + t * EC;
        touch_block(0);                                    // lands under the DMA wait and the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of pr(t) / feat(t) have landed (and col / val) ...
        __syncthreads();                                   // ... and everybody's; pid is written
        const bool ok = c_cur >= 0;
        const float vl = ok ? v_cur : 0.f;
        int c_nn = -1;
        float v_nn = 0.f;
        if (t + 2 < ntiles) load_cv(e0 + 2 * EC, c_nn, v_nn);   // wave-uniform

        f32x2 s2[MAXP], t2[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) { s2[k] = f32x2{0.f, 0.f}; t2[k] = f32x2{0.f, 0.f}; }
        auto read_rows = [&](f32x4 (&row)[CB], int blk) {
#pragma unroll
            for (int c = 0; c < CB; ++c) row[c] = *reinterpret_cast<const f32x4*>(myrow + 4 * ((blk * CB + c) ^ sw));
        };
        // Scalar operands arrive one 16-byte chunk per pair (+ w1's) at a time, one step ahead: a wait on scalar loads is always
        // lgkmcnt(0) (they return out of order), so a step is  wait(step s) -> issue loads(step s+1) -> VALU(step s).
        auto score_block = [&](const f32x4 (&row)[CB], int blk, auto slots) {
            constexpr int NS = decltype(slots)::value;     // slots computed: 1, 2 or 4 (>= np)
            constexpr int G = ATT_SPLIT_G, NST = CB / G;   // 16-byte chunks per step
            f32x4 qs[2][NS][G], ws[2][G];
            auto load_step = [&](int st, int slot) {           // constant address space + uniform address = s_load_dwordx4 / x8
#pragma unroll
                for (int gk = 0; gk < G; ++gk) {
                    if (MODE != 2) ws[slot][gk] = *(const_f32x4_ptr)(a.w1 + 4 * (blk * CB + st * G + gk));
#pragma unroll
                    for (int k = 0; k < NS; ++k) qs[slot][k][gk] = *(const_f32x4_ptr)(pcrow[k] + 4 * (blk * CB + st * G + gk));
                }
            };
            load_step(0, 0);
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int cur = st & 1;
                asm volatile("" ::"s"(qs[cur][0][0][0]));  // the compiler's wait for step st sits HERE, before the next issue
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
                            if (MODE == 3) {               // relu(p + q) on 2^-64-scaled operands = the [0, 1] clamp of the packed add
                                asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(u) : "v"(p01), "s"(q01));
                                asm("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(v) : "v"(p23), "s"(q23));
                            } else {
                                u = p01 + q01, v = p23 + q23;
                                u = __builtin_elementwise_max(u, (f32x2){0.f, 0.f});
                                v = __builtin_elementwise_max(v, (f32x2){0.f, 0.f});
                            }
                            s2[k] = u * w01 + s2[k];
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
        auto score_dispatch = [&](const f32x4 (&row)[CB], int blk) {
            if (ATT_SPLIT_DIAG == 3 || ATT_SPLIT_DIAG >= 5) return;
            if (np > 2) score_block(row, blk, std::integral_constant<int, 4>{});
            else if (np == 2) score_block(row, blk, std::integral_constant<int, 2>{});
            else if (np == 1) score_block(row, blk, std::integral_constant<int, 1>{});
        };
#if ATT_SPLIT_DB
        {
            f32x4 rA[CB], rB[CB];
            read_rows(rA, 0);
            for (int blk = 0; blk < NB; blk += 2) {        // row blocks double buffered: block b+1 is read under block b's arithmetic
                if (blk + 1 < NB) read_rows(rB, blk + 1);
                score_dispatch(rA, blk);
                if (blk + 1 < NB) {
                    if (blk + 2 < NB) read_rows(rA, blk + 2);
                    score_dispatch(rB, blk + 1);
                }
            }
        }
#else
        for (int blk = 0; blk < NB; ++blk) {               // four waves per SIMD cover a block's LDS reads
            f32x4 rA[CB];
            read_rows(rA, blk);
            if (blk + 1 < NB) touch_block(blk + 1);
            score_dispatch(rA, blk);
        }
#endif
        // Epilogue of the tile, the wave's pairs side by side: tile maximum, running maximum, rescale factor, p_e.
        float sc[MAXP], mx[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const float ssum = (s2[k][0] + t2[k][0]) + (s2[k][1] + t2[k][1]);
            sc[k] = ok ? ssum + (MODE != 2 ? a.b1 : 0.f) : -INFINITY;
            mx[k] = sc[k];
        }
        wave_reduce_dpp_n<MAXP>(mx, [](float x, float y) { return fmaxf(x, y); });
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const float mnew = fmaxf(m[k], mx[k]);
            const bool any = mnew != -INFINITY;            // wave-uniform; false: nothing valid so far, m, l, O stay 0
            const float scale = any ? exp_le0(m[k] - mnew) : 1.f;
            const float pe = (any && ok && k < np) ? exp_le0(sc[k] - mnew) : 0.f;
            l[k] = l[k] * scale + pe;                      // per-lane partial sum; lanes are added once, after the last tile
            m[k] = mnew;
            const int j = wave + NW * k;                   // every slot of the wave writes (slots beyond np: zeros) — nothing
            Pm[j * PS + lane] = pe * vl;                   // uninitialised ever reaches an MFMA
            if (lane == 0) scl[j] = (k < np) ? scale : 0.f;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // P and scl of this tile are complete; every wave is done with the pr image
        if (t + 1 < ntiles) issue_pr(c_nxt);               // tile t+1's rows land under this tile's aggregation
        {
            // jobs: (16-pair row tile mt, 16-feature column tile nt) = (j % MT, j / MT) for j = wave, wave + NW, ...
#pragma unroll
            for (int n = 0; n < NJ; ++n) {
                const int job = wave + NW * n;
                const int mt = job % MT, nt = job / MT;
                if (nt >= NTILES || ATT_SPLIT_DIAG == 4 || ATT_SPLIT_DIAG >= 5) break;   // wave-uniform
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
        if (t + 1 < ntiles) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the feat image / P have returned
            __builtin_amdgcn_s_barrier();
            c_cur = c_nxt; v_cur = v_nxt;
            c_nxt = valid_c(e0 + 2 * EC, c_nn); v_nxt = v_nn;
            issue_feat(c_cur);
        }
    }
    // ---- the slice's softmax partials (nsplit > 1) or the finished rows (nsplit == 1) ----
    float lt[MAXP];
#pragma unroll
    for (int k = 0; k < MAXP; ++k) lt[k] = l[k];
    wave_reduce_dpp_n<MAXP>(lt, [](float x, float y) { return x + y; });
    __syncthreads();                                       // the last tile's MFMAs have read scl; pid is visible (also when ntiles == 0)
    if (a.nsplit > 1) {
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if (k < np && lane == 0) {
                float* ph = a.part + (pid[wave + NW * k] * a.nsplit + split) * (int64_t)a.ldpart;
                ph[0] = m[k];
                ph[1] = lt[k];
            }
#pragma unroll
        for (int n = 0; n < NJ; ++n) {
            const int job = wave + NW * n;
            const int mt = job % MT, nt = job / MT;
            if (nt >= NTILES) break;
            const int f = 16 * nt + i16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * mt + 4 * g4 + i;
                if (row < cnt && f < Fdim) a.part[(pid[row] * a.nsplit + split) * (int64_t)a.ldpart + kPartHead + f] = acc[n][i];
            }
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < MAXP; ++k)
        if (lane == 0) scl[wave + NW * k] = (k < np && lt[k] > 0.f) ? 1.0f / lt[k] : 0.f;
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NJ; ++n) {
        const int job = wave + NW * n;
        const int mt = job % MT, nt = job / MT;
        if (nt >= NTILES) break;
        const f32x4 s4 = *reinterpret_cast<const f32x4*>(scl + 16 * mt + 4 * g4);
        const int f = 16 * nt + i16;
        const float bias = (a.out_bias && f < Fdim) ? a.out_bias[f] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 16 * mt + 4 * g4 + i;
            if (row < cnt && f < Fdim) a.out[pid[row] * a.ldout + f] = acc[n][i] * s4[i] + bias;
        }
    }
}

// out[b, :] = sum_s O_s e^(m_s - M) / sum_s l_s e^(m_s - M) + bias, M = max_s m_s  (slices in index order: deterministic).
// A pair whose slices are all empty (no rated entry: the nan_to_num case, attention_ncf.py:208-209) gets the bias alone.
__global__ __launch_bounds__(256) void attn_combine_kernel(const float* __restrict__ part, int ldpart, int nsplit, int64_t B, int Fdim,
                                                           const float* __restrict__ out_bias, float* __restrict__ out, int64_t ldout) {
    const int F4 = Fdim >> 2;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * F4) return;
    const int64_t b = i / F4;
    const int c = (int)(i - b * F4);
    const float* p0 = part + b * nsplit * (int64_t)ldpart;
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, p0[(int64_t)s * ldpart]);
    float L = 0.f;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
        const float* ps = p0 + (int64_t)s * ldpart;
        const float ms = ps[0];
        const float w = (ms == -INFINITY) ? 0.f : exp_le0(ms - M);
        L += ps[1] * w;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ps + kPartHead + 4 * c);
        o += v * w;
    }
    const float inv = L > 0.f ? 1.0f / L : 0.f;
    f32x4 bias = {0.f, 0.f, 0.f, 0.f};
    if (out_bias) bias = *reinterpret_cast<const f32x4*>(out_bias + 4 * c);
    float* dst = out + b * ldout + 4 * c;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = o[j] * inv + bias[j];
}

}  // namespace ncf

using namespace ncf;

extern "C" size_t ncf_attn_split_workspace_bytes(int64_t B, int Fdim, int nsplit) {
    if (B <= 0 || Fdim <= 0 || nsplit <= 1) return 0;
    return (size_t)B * (size_t)nsplit * (size_t)(Fdim + kPartHead) * sizeof(float);
}

extern "C" int ncf_attn_split_supported(int mode, int A, int Fdim, int pairs_per_wg) {
    if (mode != NCF_ATT_MLP && mode != NCF_ATT_COS && mode != NCF_ATT_MLP_SCALED) return 0;
    if (A <= 0 || A % 32 || A > 256 || (Fdim != 64 && Fdim != 128)) return 0;
    if (pairs_per_wg < 1 || pairs_per_wg > 64) return 0;
    const int pp = pairs_per_wg <= 16 ? 16 : (pairs_per_wg <= 32 ? 32 : 64);
    const size_t lds = ((size_t)64 * (A + Fdim) + pp * 66 + pp + 2 * pp) * 4;
    return lds <= 160 * 1024;
}

extern "C" int ncf_attn_forward_split(int mode, const float* pc, int64_t ldpc, const float* pr, int64_t ldpr, int A, const float* w1,
                                      float b1, const int64_t* rowptr, const int32_t* col, const float* val, int64_t R, int64_t I,
                                      const int64_t* grp_ptr, const int64_t* pair_ids, const int64_t* wg_ptr, const int32_t* wg_row,
                                      int64_t B, int pairs_per_wg, const float* feat, int64_t ldfeat, int Fdim, const float* out_bias,
                                      float* out, int64_t ldout, int nsplit, int merge, void* workspace, size_t workspace_bytes,
                                      ncf_stream_t stream) {
    if (!ncf_attn_split_supported(mode, A, Fdim, pairs_per_wg))
        return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_split: needs MLP / cosine mode, A %% 32 == 0 and <= 256, Fdim 64 or 128 (A = %d, Fdim = %d)", A, Fdim);
    if (B < 0 || R < 0 || I < 0 || nsplit < 1 || nsplit > 64) return fail(NCF_EINVAL, "ncf_attn_forward_split: bad sizes");
    if (B == 0 || R == 0) return NCF_OK;
    if (I == 0) return fail(NCF_EINVAL, "ncf_attn_forward_split: empty catalogue with non-empty rows");
    if (!pc || !pr || !rowptr || !col || !val || !grp_ptr || !pair_ids || !wg_ptr || !feat || !out)
        return fail(NCF_EINVAL, "ncf_attn_forward_split: null pointer");
    if (mode != NCF_ATT_COS && !w1) return fail(NCF_EINVAL, "ncf_attn_forward_split: w1 is null");
    if (ldpc < A || ldpr < A || ldfeat < Fdim || ldout < Fdim) return fail(NCF_EINVAL, "ncf_attn_forward_split: leading dimension smaller than row");
    if (ldpc % 4 || ldpr % 4 || ldfeat % 4 || ldpc >= (1ll << 31) || ldpr >= (1ll << 31) || ldfeat >= (1ll << 31) || ldout >= (1ll << 31))
        return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_split: leading dimensions must be multiples of 4 below 2^31");
    if (!aligned16(pc) || !aligned16(pr) || (w1 && !aligned16(w1)) || !aligned16(feat) || (out_bias && !aligned16(out_bias)))
        return fail(NCF_EINVAL, "ncf_attn_forward_split: operands must be 16-byte aligned");
    if (nsplit > 1 && (!workspace || workspace_bytes < ncf_attn_split_workspace_bytes(B, Fdim, nsplit) || !aligned16(workspace)))
        return fail(NCF_EWORKSPACE, "ncf_attn_forward_split: workspace too small (ncf_attn_split_workspace_bytes) or misaligned");
    hipStream_t s = (hipStream_t)stream;
    AttnSplitArgs a;
    a.pc = pc; a.pr = pr; a.w1 = w1; a.feat = feat; a.out_bias = out_bias;
    a.rowptr = rowptr; a.col = col; a.val = val; a.grp_ptr = grp_ptr; a.pair_ids = pair_ids; a.wg_ptr = wg_ptr; a.wg_row = wg_row;
    a.out = out; a.part = (float*)workspace;
    a.R = R; a.I = I;
    a.ldpc = (int)ldpc; a.ldpr = (int)ldpr; a.ldfeat = (int)ldfeat; a.ldout = (int)ldout; a.A = A; a.Fdim = Fdim;
    a.ppw = pairs_per_wg; a.nsplit = nsplit; a.ldpart = Fdim + kPartHead;
    a.b1 = b1;
    const unsigned blocks = (unsigned)((B + pairs_per_wg - 1) / pairs_per_wg + (R < B ? R : B));   // upper bound on sum_r ceil(n_r / ppw)
    const int pp = pairs_per_wg <= 16 ? 16 : (pairs_per_wg <= 32 ? 32 : 64);   // 4 / 8 / 16 waves of 4 pairs
    const size_t lds = ((size_t)64 * (A + Fdim) + pp * 66 + pp + 2 * pp) * 4;
    auto raise_lds = [&](const void* fn, std::atomic<unsigned long long>& done) -> bool {
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
    const bool scaled = mode == NCF_ATT_MLP_SCALED;
#define LAUNCH_SP(M, W, J)                                                                                              \
    do {                                                                                                                \
        static std::atomic<unsigned long long> done{0};                                                                 \
        if (lds > 64 * 1024 && !raise_lds((const void*)attn_split_kernel<M, W, J>, done))                               \
            return fail(NCF_EUNSUPPORTED, "ncf_attn_forward_split: cannot reserve %zu bytes of LDS", lds);               \
        hipLaunchKernelGGL((attn_split_kernel<M, W, J>), dim3(blocks, (unsigned)nsplit), dim3(64 * W), lds, s, a);        \
    } while (0)
#define LAUNCH_SP_J(M, W) do { if (Fdim == 64) LAUNCH_SP(M, W, 64); else LAUNCH_SP(M, W, 128); } while (0)
#define LAUNCH_SP_W(M) do { if (pp == 16) LAUNCH_SP_J(M, 4); else if (pp == 32) LAUNCH_SP_J(M, 8); else LAUNCH_SP_J(M, 16); } while (0)
    if (mode == NCF_ATT_COS) LAUNCH_SP_W(2);
    else if (scaled) LAUNCH_SP_W(3);
    else LAUNCH_SP_W(0);
#undef LAUNCH_SP_W
#undef LAUNCH_SP_J
#undef LAUNCH_SP
    int rc = check_launch("ncf_attn_forward_split");
    if (rc != NCF_OK || nsplit == 1 || !merge) return rc;   // merge == 0: the partials stay in the workspace (ncf_attn_tail merges them)
    const int64_t n = B * (Fdim / 4);
    hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)workspace, Fdim + kPartHead, nsplit, B,
                       Fdim, out_bias, out, ldout);
    return check_launch("ncf_attn_forward_split (combine)");
}
