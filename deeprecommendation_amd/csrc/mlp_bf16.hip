// K1+K2 fused, bf16 tables + bf16 weights, fp32 accumulate, fp32 output (BASELINE config 5 arithmetic), gfx950.
//
// Same transposed-chain idea as mlp_fused.hip (pair index on the MFMA column, neurons on the accumulator rows,
// layer l's accumulator registers become layer l+1's B operand) on v_mfma_f32_32x32x16_bf16:
//   * layer 1 B operand: lane (m, h) holds X[m][16s + 8h + 0..7] = the 16-byte chunk 2s + h of pair m's concatenated
//     bf16 row, read straight from the tables;
//   * layer 2 B operand: registers 8s'..8s'+7 of accumulator tile kb, ReLU'd and rounded to bf16 (RNE), are the
//     fragment of k-step 2kb + s'; element j of lane-half h is neuron 32kb + 16s' + 8(j>>2) + 4h + (j&3), so W2 is
//     PACKED in that permuted k order (ncf_mlp_pack with dtype NCF_BF16 does it).
// bf16 MFMA is 16x the fp32 rate, so weights can no longer be streamed per wave from L2 (8 waves x 196 KB per tile
// would need 128 B/clk/CU of L1 bandwidth): a 512-thread workgroup (8 waves, 256 pairs) shares each weight slab
// through LDS — slab t+1 is fetched to registers while slab t is consumed, written to the other LDS buffer, one
// barrier per slab; A fragments are lane-linear in LDS (conflict-free ds_read_b128).
// Bound at config 5 (E = 128): bf16 MFMA 196 864 FLOP/pair (2.5 PF -> 12.7 G pairs/s) vs HBM 532 B/pair
// (8 TB/s -> 15 G pairs/s): roughly balanced.
#include "mlp_bf16.h"
#include <stdlib.h>
#include <string.h>

#ifndef NCF_BF16_X_DEPTH
#define NCF_BF16_X_DEPTH 4   // gathered-row prefetch ring, in 16-wide k-steps (measured: 4 -> 20.6 us, 8 -> 21.3, 16 -> 21.7)
#endif
#ifndef NCF_BF16_MSTEP
#define NCF_BF16_MSTEP 4     // 16-wide k-steps per weight slab (= per barrier); measured 2 -> 20.8 us, 4 -> 19.3 us
#endif
#ifndef NCF_BF16_WGW
#define NCF_BF16_WGW 8       // waves per workgroup sharing a weight slab; measured: 8 -> 19.6 us, 4 (two WGs per CU) -> 20.4, 2 -> 24-28
#endif
#ifndef NCF_BF16_ABLATE
#define NCF_BF16_ABLATE 0    // diagnostics: 1 = no gathered-row loads, 2 = no weight slab copies / barriers, 3 = rows from a 1 MiB window,
                             // 5 = weight-stationary kernel reduced to its ds_read + MFMA + barrier skeleton
#endif

namespace ncf {

#if NCF_BF16_STAMP
static unsigned long long* g_bf16_dbg = nullptr;
#define BF16_STAMP(i) do { if (a.dbg && lane == 0) { a.dbg[tile * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BF16_STAMP(i) do { } while (0)
#endif

template <int K0, int N1, int N2>
__global__ __launch_bounds__(NCF_BF16_WGW * 64, 2) void score_fused_bf16_kernel(Bf16Args a) {
    constexpr int WGW = NCF_BF16_WGW;
    constexpr int NT1 = N1 / 32, Q1 = K0 / 16;   // layer 1: Q1 k-steps of 16
    constexpr int NT2 = N2 / 32, Q2 = N1 / 16;   // layer 2
    constexpr int XD = NCF_BF16_X_DEPTH < Q1 ? NCF_BF16_X_DEPTH : Q1;
    constexpr int MS = NCF_BF16_MSTEP;
    constexpr int SLAB1 = MS * NT1 * 1024;       // bytes per macro-step (MS k-steps) of layer 1
    constexpr int SLAB2 = MS * NT2 * 1024;
    constexpr int PIECES1 = SLAB1 / 1024 / WGW;  // 1-KiB pieces per wave per slab
    constexpr int PIECES2 = (SLAB2 / 1024 + WGW - 1) / WGW;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][SLAB1];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * WGW + wave;
    const int64_t p = tile * 32 + m;
    const int64_t pc = p < a.B ? p : a.B - 1;     // waves past the end still take part in copies and barriers
    BF16_STAMP(0);

    const int64_t ia = a.idxA ? a.idxA[pc] : pc;
    const bool okA = (ia >= 0) & (ia < a.rowsA);
    const unsigned short* rowA = a.tabA + (okA ? ia : 0) * a.ldA + 8 * h;
    const int qa = a.EA / 16;                     // k-steps served by table A
    const unsigned short* rowB = rowA;
    bool okB = true;
    if (qa < Q1) {
        const int64_t ib = a.idxB ? a.idxB[pc] : pc;
        okB = (ib >= 0) & (ib < a.rowsB);
        rowB = a.tabB + (okB ? ib : 0) * a.ldB + 8 * h;
    }
    if (!(okA & okB) && a.oob && p < a.B) *a.oob = 1;
    auto xsrc = [&](int q) { return q < qa ? rowA + 16 * q : rowB + 16 * (q - qa); };

    // gathered-row ring: chunk of k-step q is requested XD steps ahead; plain loads survive the LDS barriers
    u32x4 x[XD];
#pragma unroll
    for (int t = 0; t < XD; ++t) x[t] = (NCF_BF16_ABLATE == 1) ? u32x4{1u, 2u, 3u, 4u} : ldg16(xsrc(t));

    // first weight slab -> LDS buffer 0
    {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.Wp1);
#pragma unroll
        for (int i = 0; i < PIECES1; ++i) {
            const int piece = wave * PIECES1 + i;
            *reinterpret_cast<u32x4*>(&lds[0][piece * 1024 + lane * 16]) = ldg16(src + piece * 1024 + lane * 16);
        }
    }

    f32x16 acc1[NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b1 + 32 * nt + 8 * g + 4 * h);
            acc1[nt][4 * g + 0] = bb[0]; acc1[nt][4 * g + 1] = bb[1];
            acc1[nt][4 * g + 2] = bb[2]; acc1[nt][4 * g + 3] = bb[3];
        }
    __syncthreads();
    BF16_STAMP(1);

    // Slab pipeline (one barrier per slab of MS k-steps).  A fragments of k-step s+1 are read from LDS while k-step
    // s's MFMAs run (register double buffer wa[2]); the barrier sits BEFORE the slab's last k-step, after that
    // step's fragments are in registers and after this wave has written its pieces of the NEXT slab, so the last
    // k-step can already prefetch the next slab's first fragments: no MFMA ever waits on an LDS read issued in the
    // same step (hipcc's own schedule put `ds_read; s_waitcnt lgkmcnt(0)` in front of every MFMA after a barrier).
    u32x4 wa[2][NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt) wa[0][nt] = *reinterpret_cast<const u32x4*>(&lds[0][nt * 1024 + lane * 16]);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- layer 1 ----------------
#pragma unroll
    for (int t = 0; t < Q1 / MS; ++t) {
        const int cur = t & 1;
        u32x4 nxt[PIECES1];
        const bool more1 = t + 1 < Q1 / MS;
        const bool have_next = more1 || N2 > 0;
        const int npieces = more1 ? PIECES1 : PIECES2;
        if (have_next) {
            const unsigned char* src = more1 ? reinterpret_cast<const unsigned char*>(a.Wp1) + (size_t)(t + 1) * SLAB1
                                             : reinterpret_cast<const unsigned char*>(a.Wp2);
#pragma unroll
            for (int i = 0; i < PIECES1; ++i)
                if (i < npieces) {
                    const int piece = wave * npieces + i;
                    if (more1 || piece * 1024 < SLAB2) nxt[i] = ldg16(src + piece * 1024 + lane * 16);
                }
        }
#pragma unroll
        for (int qq = 0; qq < MS; ++qq) {
            const int q = MS * t + qq;
            if (qq == MS - 1) {
                if (have_next) {
#pragma unroll
                    for (int i = 0; i < PIECES1; ++i)
                        if (i < npieces) {
                            const int piece = wave * npieces + i;
                            if (more1 || piece * 1024 < SLAB2)
                                *reinterpret_cast<u32x4*>(&lds[cur ^ 1][piece * 1024 + lane * 16]) = nxt[i];
                        }
                }
                __syncthreads();
            }
            u32x4 xr = x[q % XD];
            if (!(q < qa ? okA : okB)) xr = u32x4{0u, 0u, 0u, 0u};  // out-of-range row reads as zeros
            if (q + XD < Q1 && NCF_BF16_ABLATE != 1) x[q % XD] = ldg16(xsrc(q + XD));
            const bf16x8_t xb = as_bf16x8(xr);
            // fragments for the following k-step: same slab, or (last k-step) the first k-step of the next slab
            const bool pf = qq + 1 < MS || have_next;
            const int pf_nt = (qq + 1 < MS || more1) ? NT1 : NT2;
            const unsigned char* pf_base = qq + 1 < MS ? &lds[cur][(qq + 1) * NT1 * 1024] : &lds[cur ^ 1][0];
#pragma unroll
            for (int nt = 0; nt < NT1; ++nt) {
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa[qq & 1][nt]), xb, acc1[nt], 0, 0, 0);
                if (pf && nt < pf_nt)
                    wa[(qq + 1) & 1][nt] = *reinterpret_cast<const u32x4*>(pf_base + nt * 1024 + lane * 16);
            }
            if (pf) {
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    if (nt < pf_nt) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    BF16_STAMP(2);
    float partial = 0.f;
    if constexpr (N2 > 0) {
        // activations of layer 1 -> bf16 B fragments (frees the fp32 accumulators)
        bf16x8_t hb[NT1][2];
#pragma unroll
        for (int kb = 0; kb < NT1; ++kb) {
            hb[kb][0] = pack_relu8(acc1[kb], 0);
            hb[kb][1] = pack_relu8(acc1[kb], 8);
        }
        f32x16 acc2[NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(a.b2 + 32 * nt + 8 * g + 4 * h);
                acc2[nt][4 * g + 0] = bb[0]; acc2[nt][4 * g + 1] = bb[1];
                acc2[nt][4 * g + 2] = bb[2]; acc2[nt][4 * g + 3] = bb[3];
            }
        BF16_STAMP(3);
        constexpr int T0 = (Q1 / MS) & 1;  // LDS buffer holding layer 2's first slab
        // ---------------- layer 2 ----------------
#pragma unroll
        for (int t = 0; t < Q2 / MS; ++t) {
            const int cur = (T0 + t) & 1;
            u32x4 nxt[PIECES2];
            const bool more = t + 1 < Q2 / MS;
            if (more) {
                const unsigned char* src = reinterpret_cast<const unsigned char*>(a.Wp2) + (size_t)(t + 1) * SLAB2;
#pragma unroll
                for (int i = 0; i < PIECES2; ++i) {
                    const int piece = wave * PIECES2 + i;
                    if (piece * 1024 < SLAB2) nxt[i] = ldg16(src + piece * 1024 + lane * 16);
                }
            }
#pragma unroll
            for (int qq = 0; qq < MS; ++qq) {
                const int q = MS * t + qq;  // k-step q = 2*kb + s'
                if (qq == MS - 1 && more) {
#pragma unroll
                    for (int i = 0; i < PIECES2; ++i) {
                        const int piece = wave * PIECES2 + i;
                        if (piece * 1024 < SLAB2) *reinterpret_cast<u32x4*>(&lds[cur ^ 1][piece * 1024 + lane * 16]) = nxt[i];
                    }
                    __syncthreads();
                }
                const bool pf = qq + 1 < MS || more;
                const unsigned char* pf_base = qq + 1 < MS ? &lds[cur][(qq + 1) * NT2 * 1024] : &lds[cur ^ 1][0];
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa[qq & 1][nt]), hb[q >> 1][q & 1], acc2[nt], 0, 0, 0);
                    if (pf) wa[(qq + 1) & 1][nt] = *reinterpret_cast<const u32x4*>(pf_base + nt * 1024 + lane * 16);
                }
                if (pf) {
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        BF16_STAMP(4);
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(a.wl + 32 * nt + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], fmaxf(acc2[nt][4 * g + j], 0.f), partial);
            }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(a.wl + 32 * nt + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], fmaxf(acc1[nt][4 * g + j], 0.f), partial);
            }
    }
    partial += __shfl_xor(partial, 32);
    if (h == 0 && p < a.B) a.out[p] = partial + a.bl[0];
    BF16_STAMP(5);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-stationary persistent form (the default for EA, EB multiples of 64).
//
// One 256-thread workgroup per CU, ONE wave per SIMD, looping over 64-pair tiles.  The weights never move again
// after the prologue: wave w keeps its slice as MFMA A fragments in registers for the whole launch — layer-1 neurons
// [64w, 64w+64) (2 row tiles x K0/16 k-steps) and layer-2 neurons [32w, 32w+32) (1 row tile x N1/16 k-steps),
// 192 VGPRs at 256-256-128 — so nothing is streamed from L2 per tile and no weight byte crosses LDS.
// What crosses LDS instead is the data:
//   * X: the tile's gathered rows, fetched by LDS-DMA (global_load_lds_dwordx4, per-lane SOURCE address) in pieces of
//     8 pairs x 128 B — full cache lines — three tiles deep (ring of 3 buffers, 2 tiles = 64 KB in flight per CU
//     under the MFMAs).  Inside a piece lane position L = (chunk%8)*8 + pair%8, XOR 8 for pair groups 2 and 3:
//     the B-fragment read of k-step s (lane (m,h) <- 16-byte chunk 2s+h of pair m) is then a conflict-free
//     ds_read_b128 (each of its 16-lane groups covers 16 distinct bank quads);
//   * H1: each wave's ReLU'd, bf16-rounded layer-1 slice, written as ready-made layer-2 B fragments (same permuted
//     k order as the streaming kernel, so W2 keeps its packing) — one barrier, then every wave reads all of it;
//   * 64 x 4 partial dots of the last layer.
// Row ids come in by LDS-DMA as well (one dword DMA per wave and tile into the wave's own slot, four tiles ahead) so
// that no VGPR-destination load sits in the vmcnt queue between the DMAs: the loop's only vector-memory ops are the
// DMAs and the output store, waited for with a counted s_waitcnt vmcnt before a raw s_barrier (a __syncthreads()
// would drain the prefetch).
// Per iteration i: ids DMA(i+4), row DMAs(i+2) -> layer 1(i) -> H1 -> barrier -> layer 2(i) -> partial dots ->
// wait ids(i+3) + rows(i+1) -> row pointers(i+3) -> barrier -> store out(i).
#ifndef NCF_BF16_WS
#define NCF_BF16_WS 1
#endif
#ifndef NCF_BF16_WS_MIN_PAIRS
#define NCF_BF16_WS_MIN_PAIRS 1
#endif

#if NCF_BF16_STAMP
// diagnostic builds: stamps of iterations 0..3 and 100..103 of every workgroup, [block][8 iterations][8 stamps]
#define WS_STAMP(k) do { if (a.dbg && lane == 0 && w == 0 && (it < 4 || (it >= 100 && it < 104))) \
    a.dbg[((int64_t)blockIdx.x * 8 + (it < 4 ? it : it - 96)) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WS_STAMP(k) do { } while (0)
#endif

template <int K0, int N1, int N2>
struct WsLayout {
    static constexpr int P = 64, CTN = 2;                 // pairs per tile, 32-pair column tiles per tile
    static constexpr int NCU = K0 / 64;                   // 128-byte column units of the concatenated row
    static constexpr int CT_BYTES = NCU * 4 * 1024;       // X image of one column tile
    static constexpr int XBUF = CTN * CT_BYTES;
    static constexpr int Q2 = N1 / 16;
    static constexpr int H1_BYTES = N2 > 0 ? CTN * Q2 * 1024 : 0;
    static constexpr int LAST = N2 > 0 ? N2 : N1;
    static constexpr int OFF_H1 = 3 * XBUF;
    static constexpr int OFF_RED = OFF_H1 + H1_BYTES;     // float red[4][64]
    static constexpr int OFF_B1 = OFF_RED + 1024;
    static constexpr int OFF_B2 = OFF_B1 + N1 * 4;
    static constexpr int OFF_WL = OFF_B2 + (N2 > 0 ? N2 * 4 : 0);
    static constexpr int OFF_IDS = OFF_WL + LAST * 4;     // int64 ids[4 waves][2 slots][2 ct][2 tables][8]
    static constexpr int TOTAL = OFF_IDS + 4 * 2 * 256;
};

#define SGB_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define SGB_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
#define SGB_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)
#define SGB_DSW(n) __builtin_amdgcn_sched_group_barrier(0x200, n, 0)


template <int K0, int N1, int N2>
__global__ __launch_bounds__(256, 1) void score_ws_bf16_kernel(Bf16Args a, const int64_t* __restrict__ idxA,
                                                               const int64_t* __restrict__ idxB, float* __restrict__ out,
                                                               const unsigned char* __restrict__ zeros, int64_t ntiles) {
    using L = WsLayout<K0, N1, N2>;
    static_assert(N1 == 256 && (N2 == 0 || N2 == 128), "wave w owns 64 layer-1 and 32 layer-2 neurons");
    constexpr int Q1 = K0 / 16, Q2 = L::Q2, CTN = L::CTN, NCU = L::NCU;
    constexpr int ROWS = CTN * NCU;                          // row-DMA pieces per wave and tile
    static_assert(ROWS <= Q1, "row-DMA pieces are spread over the k-steps of phase A");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[L::TOTAL];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const int ncuA = a.EA / 64;
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;        // LDS byte address of the array (M0 of the DMAs)
#if NCF_BF16_STAMP
    if (a.dbg && lane == 0 && w == 0) a.dbg[256 * 64 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // kernel start
#endif
    // DMA lane role inside a piece: pair slot ml = lane & 7 of group w, chunk cl of the 128-byte unit
    const int cl = ((lane ^ (w >= 2 ? 8 : 0)) >> 3);
    const int ml = lane & 7;
    // reader lane offset inside an X column-tile image (k-step s adds (s>>2)*4096 + (s&3)*256)
    const int g = m >> 3;
    const unsigned rd_off = g * 1024 + ((h * 128 + (m & 7) * 16) ^ (g >= 2 ? 128 : 0));

    const int64_t stride = gridDim.x;
    int64_t tile = blockIdx.x;
    const bool e1 = tile + stride < ntiles, e2 = tile + 2 * stride < ntiles, e3 = tile + 3 * stride < ntiles;

    // ---- prologue.  Oldest first in the vector-memory queue: the ids of tiles 0..2 (they head the longest
    // dependent chain: ids -> row addresses -> row DMAs -> first MFMA), then the weights (registers for the whole
    // launch), then the biases (LDS). ----
    int64_t pid[3][CTN][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) {
            const int64_t tt = (t == 0 || (t == 1 && e1) || (t == 2 && e2)) ? tile + t * stride : tile;
            const int64_t p0 = tt * L::P + 32 * ct + 8 * w + ml;
            const int64_t p = p0 < a.B ? p0 : a.B - 1;
            pid[t][ct][0] = idxA[p];
            pid[t][ct][1] = idxB[p];
        }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 wa1[2][Q1];
#pragma unroll
    for (int s = 0; s < Q1; ++s)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            wa1[nt][s] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp1) + ((size_t)(s * (N1 / 32) + 2 * w + nt) * 64 + lane) * 16);
    u32x4 wa2[N2 > 0 ? Q2 : 1];
    if constexpr (N2 > 0) {
#pragma unroll
        for (int s = 0; s < Q2; ++s)
            wa2[s] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp2) + ((size_t)(s * (N2 / 32) + w) * 64 + lane) * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float* lb1 = reinterpret_cast<float*>(lds + L::OFF_B1);
        for (int i = threadIdx.x; i < N1; i += 256) lb1[i] = a.b1[i];
        if constexpr (N2 > 0) {
            float* lb2 = reinterpret_cast<float*>(lds + L::OFF_B2);
            for (int i = threadIdx.x; i < N2; i += 256) lb2[i] = a.b2[i];
        }
        float* lwl = reinterpret_cast<float*>(lds + L::OFF_WL);
        for (int i = threadIdx.x; i < L::LAST; i += 256) lwl[i] = a.wl[i];
    }
    const float bl = a.bl[0];

    // per-lane source rows of one tile: [ct][table]; an out-of-range id reads the 512-byte zero block instead
    typedef const unsigned char* RowSrc[CTN][2];
    bool oob_seen = false;                                   // stored once, after the loop
    auto resolve = [&](RowSrc& src, int ct, int64_t p0, int64_t ia, int64_t ib) {
        const bool okA = (ia >= 0) & (ia < a.rowsA);
        const bool okB = (ncuA == NCU) | ((ib >= 0) & (ib < a.rowsB));
        oob_seen |= !(okA & okB) & (p0 < a.B);
        src[ct][0] = (okA ? reinterpret_cast<const unsigned char*>(a.tabA + ia * a.ldA) : zeros) + cl * 16;
        src[ct][1] = ((okB & (ncuA < NCU)) ? reinterpret_cast<const unsigned char*>(a.tabB + ib * a.ldB) : zeros) + cl * 16;
    };
    // steady state: a tile's 32 ids per wave (2 column tiles x 2 tables x 8 pairs) arrive by ONE dword LDS-DMA into
    // the wave's own 256-byte slot (lane = dword), so that the vmcnt queue holds nothing but DMAs and stores
    auto ids_dma = [&](int64_t t, int slot) {
        const int grp = lane >> 4, dw = lane & 15;
        const int64_t p0 = t * L::P + 32 * (grp >> 1) + 8 * w + (dw >> 1);
        const int64_t p = p0 < a.B ? p0 : a.B - 1;
        const int* gp = reinterpret_cast<const int*>(((grp & 1) ? idxB : idxA) + p) + (dw & 1);
        dma4(gp, lds0 + L::OFF_IDS + (w * 2 + slot) * 256);
    };
    auto locate_lds = [&](RowSrc& src, int64_t t, int slot) {
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) {
            const unsigned char* q = lds + L::OFF_IDS + (w * 2 + slot) * 256 + ct * 128 + ml * 8;
            const int64_t ia = *reinterpret_cast<const int64_t*>(q);
            const int64_t ib = *reinterpret_cast<const int64_t*>(q + 64);
            resolve(src, ct, t * L::P + 32 * ct + 8 * w + ml, ia, ib);
        }
    };
    // one 1-KiB piece (8 pairs x 128 B) of a tile's X image: column tile ct, 128-byte unit cu, pair group w
    auto issue_piece = [&](const RowSrc& src, int buf, int ct, int cu) {
        const bool fromA = cu < ncuA;
        const unsigned char* gp = (fromA ? src[ct][0] : src[ct][1]) + (fromA ? cu : cu - ncuA) * 128;
        if (NCF_BF16_ABLATE == 1 || NCF_BF16_ABLATE == 5) return;  // diagnostics: no row DMAs
        if (NCF_BF16_ABLATE == 3) gp = reinterpret_cast<const unsigned char*>(a.tabA) + ((gp - reinterpret_cast<const unsigned char*>(a.tabA)) & 0xFFFFF);  // diagnostics: rows from a 1 MiB window
        dma16(gp, lds0 + buf * L::XBUF + ct * L::CT_BYTES + (cu * 4 + w) * 1024);
    };

    // vector-memory queue of a wave from here on, oldest first (rows = ROWS DMAs; a missing tile's entries are absent):
    //   ids(3) rows(0) rows(1) | ids(4) rows(2) | ids(5) rows(3) [out(0)] | ids(6) rows(4) [out(1)] | ...
    // In iteration `it` the wave needs ids(it+3) and rows(it+1): everything except the entries of its own phase A.
    RowSrc src;
    {
        RowSrc src0, src1;
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct) {
            resolve(src0, ct, tile * L::P + 32 * ct + 8 * w + ml, pid[0][ct][0], pid[0][ct][1]);
            resolve(src1, ct, (tile + stride) * L::P + 32 * ct + 8 * w + ml, pid[1][ct][0], pid[1][ct][1]);
            resolve(src, ct, (tile + 2 * stride) * L::P + 32 * ct + 8 * w + ml, pid[2][ct][0], pid[2][ct][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (e3) ids_dma(tile + 3 * stride, 1);
#pragma unroll
        for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
            for (int cu = 0; cu < NCU; ++cu) issue_piece(src0, 0, ct, cu);
        if (e1) {
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct)
#pragma unroll
                for (int cu = 0; cu < NCU; ++cu) issue_piece(src1, 1, ct, cu);
        }
    }
    // the weights are older than every DMA: using them here lets hipcc retire its own waits for them now instead of
    // re-emitting `s_waitcnt vmcnt(47..0)` through the first iteration (it cannot see the DMAs, so those would drain them)
#pragma unroll
    for (int s = 0; s < Q1; ++s) asm volatile("" ::"v"(wa1[0][s]), "v"(wa1[1][s]));
    if constexpr (N2 > 0) {
#pragma unroll
        for (int s = 0; s < Q2; ++s) asm volatile("" ::"v"(wa2[s]));
    }
    if (e1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // Steady state.  A wave alone on its SIMD has nobody to cover its non-MFMA work, so each tile is run as a pipeline
    // over its two column tiles and every VALU / LDS / DMA piece is placed under MFMAs of the OTHER column tile:
    //   A: layer 1 (ct0) | row DMAs of tile it+2, ids DMA of tile it+4; last-layer dot of ct1 of tile it-1 -> LDS
    //   B: layer 1 (ct1) | ReLU + bf16 + H1 write of ct0                          -> barrier alpha
    //   C: layer 2 (ct0) | ReLU + bf16 + H1 write of ct1; out(it-1); wait rows(it+1) -> barrier beta
    //   D: layer 2 (ct1) | last-layer dot of ct0, row pointers of tile it+3
    // B fragments come through register rings read 3 (layer 1, 64 MFMA cycles per fragment) or AHEAD2 (layer 2, 32)
    // fragments ahead; sched_group_barrier sequences spread the fillers over the MFMA issue gaps.
#ifndef NCF_WS_AHEAD2
#define NCF_WS_AHEAD2 6
#endif
    constexpr int RING = 4, AHEAD = 3;
    constexpr int AHEAD2 = NCF_WS_AHEAD2, RING2 = AHEAD2 + 2;
    float* const red = reinterpret_cast<float*>(lds + L::OFF_RED);
    constexpr int NPREV = N2 > 0 ? 1 : 2;                    // accumulator tiles of ct1 whose last-layer dot is deferred
    // The accumulators live across iterations: ct1's last hidden layer (acc2[1], or acc1[.][1] without a second
    // layer) is consumed in the NEXT iteration's phase A, before that iteration re-initialises it.
    f32x16 acc1[2][CTN];
    f32x16 acc2[N2 > 0 ? CTN : 1];
    auto prev1 = [&](int i) -> const f32x16& { if constexpr (N2 > 0) return acc2[1]; else return acc1[i][1]; };
    float prev0 = 0.f;                                       // ct0's finished partial dot of the previous tile
    auto dot_quarter = [&](const f32x16& acc, int neuron0, int gq, float part) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(lds + L::OFF_WL + (neuron0 + 8 * gq + 4 * h) * 4);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) part = fmaf(ww[jj], relu1(acc[4 * gq + jj]), part);
        return part;
    };
    auto publish = [&](float p0, float p1) {                 // this wave's share of 64 last-layer dots -> LDS
        const float v0 = p0 + __shfl_xor(p0, 32), v1 = p1 + __shfl_xor(p1, 32);
        if (h == 0) { red[w * 64 + m] = v0; red[w * 64 + 32 + m] = v1; }
    };
    auto store_out = [&](int64_t t) {                        // after the barrier that follows publish(): wave 0
        if (w == 0) {
            const int64_t p = t * L::P + lane;
            const float v = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane] + bl;
            if (p < a.B) out[p] = v;
        }
    };

    // accumulator tiles start as their bias, read from LDS straight into the accumulator registers
    auto bias_tile = [&](int off, int neuron0) {
        f32x16 t;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + off + (neuron0 + 8 * gq + 4 * h) * 4);
            t[4 * gq + 0] = bb[0]; t[4 * gq + 1] = bb[1]; t[4 * gq + 2] = bb[2]; t[4 * gq + 3] = bb[3];
        }
        return t;
    };
    // The barriers are straddled: the last TAIL k-steps of the MFMA stream that precedes a barrier are issued AFTER
    // it, so that they cover the LDS latency of the first fragment reads of the stream that follows it (those reads
    // cannot be issued before the barrier).  The X ring and the bias tiles of the next tile are likewise requested
    // inside the previous tile's last k-steps (its rows are visible since beta).
    constexpr int TAIL = 4;
    constexpr int BODY1 = Q1 - TAIL, BODY2 = Q2 - TAIL;
    static_assert(BODY1 % 4 == 0 && BODY2 % 4 == 0, "four fragments are spread over the body k-steps");
    auto xaddr = [&](int b, int j) {
        return lds + b * L::XBUF + rd_off + (j / Q1) * L::CT_BYTES + ((j % Q1) >> 2) * 4096 + ((j % Q1) & 3) * 256;
    };
    u32x4 xf[RING];
#pragma unroll
    for (int j = 0; j < AHEAD; ++j) xf[j] = *reinterpret_cast<const u32x4*>(xaddr(0, j));
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) acc1[nt][0] = bias_tile(L::OFF_B1, 64 * w + 32 * nt);
    const unsigned char* hbase = lds + L::OFF_H1 + lane * 16;

    for (int it = 0; tile < ntiles; ++it, tile += stride) {
        const int buf = it % 3, nbuf = (it + 1) % 3;
        WS_STAMP(0);
#if NCF_BF16_STAMP
        if (a.dbg && lane == 0 && w == 0 && (it < 4 || (it >= 100 && it < 104)))
            a.dbg[((int64_t)blockIdx.x * 8 + (it < 4 ? it : it - 96)) * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
        const bool has2 = tile + 2 * stride < ntiles, has3 = tile + 3 * stride < ntiles, has4 = tile + 4 * stride < ntiles;
        if (has4 && NCF_BF16_ABLATE != 5) ids_dma(tile + 4 * stride, it & 1);   // slot (it+4)&1; slot (it+3)&1 is read in phase D
        float part1 = 0.f;                                   // deferred dot of the previous tile's ct1
        float part0 = 0.f;                                   // this tile's ct0
        u32x4 hf[CTN][N2 > 0 ? RING2 : 1];
        __builtin_amdgcn_sched_barrier(0);
        // ---------------- layer 1: 2*Q1 fragment steps, column tile ct = j / Q1 ----------------
#pragma unroll
        for (int j = 0; j < CTN * Q1; ++j) {
            const int ct = j / Q1, s = j % Q1;
            if (j + AHEAD < CTN * Q1) xf[(j + AHEAD) % RING] = *reinterpret_cast<const u32x4*>(xaddr(buf, j + AHEAD));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc1[nt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa1[nt][s]), as_bf16x8(xf[j % RING]), acc1[nt][ct], 0, 0, 0);
            if (ct == 0) {
                // phase A fillers: one row-DMA piece per k-step, then the previous tile's ct1 dot in quarters
                if (s < ROWS) {
                    if (has2) issue_piece(src, (it + 2) % 3, s / NCU, s % NCU);
                } else if (it > 0 && NCF_BF16_ABLATE != 5) {
                    constexpr int DQ = 4 * NPREV;            // dot quarters to place
                    const int k = s - ROWS;
                    constexpr int PER = (DQ + (Q1 - ROWS) - 2) / ((Q1 - ROWS) - 1);   // quarters per remaining step but the last
                    if (s < Q1 - 1) {
#pragma unroll
                        for (int d = k * PER; d < (k + 1) * PER && d < DQ; ++d)
                            part1 = dot_quarter(prev1(d / 4), (N2 > 0 ? 32 * w : 64 * w + 32 * (d / 4)), d % 4, part1);
                    } else {
                        publish(prev0, part1);
                    }
                }
                if (s == Q1 - 2) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) acc1[nt][1] = bias_tile(L::OFF_B1, 64 * w + 32 * nt);
                }
                SGB_DSR(1); SGB_MFMA(1); SGB_VALU(4); SGB_MFMA(1); SGB_VALU(4);
                if (s == Q1 - 1) WS_STAMP(1);
                __builtin_amdgcn_sched_barrier(0);
            } else if constexpr (N2 > 0) {
                // phase B fillers: ct0's four finished fragments over the BODY1 k-steps before alpha
                if (s < BODY1 && (s + 1) % (BODY1 / 4) == 0) {
                    const int f = (s + 1) / (BODY1 / 4) - 1, nt = f >> 1, s2 = f & 1;
                    const bf16x8_t hb = NCF_BF16_ABLATE == 5 ? as_bf16x8(xf[0]) : pack_relu8_int(acc1[nt][0], 8 * s2);
                    *reinterpret_cast<bf16x8_t*>(lds + L::OFF_H1 + (0 * Q2 + 2 * (2 * w + nt) + s2) * 1024 + lane * 16) = hb;
#pragma unroll
                    for (int r = 0; r < BODY1 / 4; ++r) { SGB_DSR(1); SGB_MFMA(1); SGB_VALU(8 / (BODY1 / 4)); SGB_MFMA(1); SGB_VALU(8 / (BODY1 / 4)); }
                    SGB_DSW(1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s == BODY1 - 1) {
                    WS_STAMP(2);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();            // alpha: H1(ct0) and the previous tile's dots are visible
                    WS_STAMP(3);
#pragma unroll
                    for (int q = 0; q < AHEAD2; ++q) hf[0][q] = *reinterpret_cast<const u32x4*>(hbase + (0 * Q2 + q) * 1024);
                    acc2[0] = bias_tile(L::OFF_B2, 32 * w);
                    if (it > 0) store_out(tile - stride);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (s >= BODY1) __builtin_amdgcn_sched_barrier(0);
            } else {
                // no second layer: ct0's last-layer dot, a quarter of the k-steps per accumulator half
                if ((s + 1) % (Q1 / 4) == 0) {
                    const int f = (s + 1) / (Q1 / 4) - 1, nt = f >> 1, s2 = f & 1;
                    part0 = dot_quarter(acc1[nt][0], 64 * w + 32 * nt, 2 * s2, part0);
                    part0 = dot_quarter(acc1[nt][0], 64 * w + 32 * nt, 2 * s2 + 1, part0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if constexpr (N2 > 0) {
            // ---------------- layer 2 ----------------
#pragma unroll
            for (int ct = 0; ct < CTN; ++ct) {
                if (ct == 1 && has3 && NCF_BF16_ABLATE != 5) locate_lds(src, tile + 3 * stride, (it + 1) & 1);   // overwrites src[][] (rows(it+2) are issued)
#pragma unroll
                for (int q = 0; q < Q2; ++q) {
                    if (q + AHEAD2 < Q2) hf[ct][(q + AHEAD2) % RING2] = *reinterpret_cast<const u32x4*>(hbase + (ct * Q2 + q + AHEAD2) * 1024);
                    acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa2[q]), as_bf16x8(hf[ct][q % RING2]), acc2[ct], 0, 0, 0);
                    if (q < BODY2 && (q + 1) % (BODY2 / 4) == 0) {
                        const int f = (q + 1) / (BODY2 / 4) - 1;
                        if (ct == 0) {
                            // phase C filler: one finished ct1 fragment of layer 1 per quarter of the body
                            const int nt = f >> 1, s2 = f & 1;
                            const bf16x8_t hb = NCF_BF16_ABLATE == 5 ? as_bf16x8(hf[0][0]) : pack_relu8_int(acc1[nt][1], 8 * s2);
                            *reinterpret_cast<bf16x8_t*>(lds + L::OFF_H1 + (1 * Q2 + 2 * (2 * w + nt) + s2) * 1024 + lane * 16) = hb;
                        } else {
                            // phase D filler: a quarter of ct0's last-layer dot per quarter of the body
                            if (NCF_BF16_ABLATE != 5) part0 = dot_quarter(acc2[0], 32 * w, f, part0);
                        }
#pragma unroll
                        for (int r = 0; r < BODY2 / 4; ++r) { SGB_DSR(1); SGB_MFMA(1); SGB_VALU(16 / (BODY2 / 4)); }
                        if (ct == 0) SGB_DSW(1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (q == BODY2 - 1) {
                        if (ct == 0) {
                            // this wave's pieces of rows(it+1) (and ids(it+3)) have landed once only this iteration's DMAs are left
                            if (has4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS + 1) : "memory");
                            else if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS) : "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            __builtin_amdgcn_s_barrier();    // beta: H1(ct1) and rows(it+1) are visible
                            WS_STAMP(4);
#pragma unroll
                            for (int qq = 0; qq < AHEAD2; ++qq) hf[1][qq] = *reinterpret_cast<const u32x4*>(hbase + (1 * Q2 + qq) * 1024);
                            acc2[1] = bias_tile(L::OFF_B2, 32 * w);
                        } else {
                            // the next tile's first fragments and bias tiles (harmless if there is no next tile)
#pragma unroll
                            for (int j = 0; j < AHEAD; ++j) xf[j] = *reinterpret_cast<const u32x4*>(xaddr(nbuf, j));
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) acc1[nt][0] = bias_tile(L::OFF_B1, 64 * w + 32 * nt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (q >= BODY2) __builtin_amdgcn_sched_barrier(0);
                }
            }
            WS_STAMP(5);
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // alpha: the previous tile's dots are visible
            if (it > 0) store_out(tile - stride);
            if (has4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS + 1) : "memory");
            else if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ROWS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (has3) locate_lds(src, tile + 3 * stride, (it + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                    // beta: rows(it+1) are visible; wave 0 is done with the dots
#pragma unroll
            for (int j = 0; j < AHEAD; ++j) xf[j] = *reinterpret_cast<const u32x4*>(xaddr(nbuf, j));
            // ct1's accumulators still hold the hidden layer the next iteration's phase A reads: only ct0's restart here
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc1[nt][0] = bias_tile(L::OFF_B1, 64 * w + 32 * nt);
            WS_STAMP(5);
        }
        prev0 = part0;
        if (NCF_BF16_ABLATE == 5) {                          // diagnostics: keep the MFMA results alive without reading them
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int ct = 0; ct < CTN; ++ct) asm volatile("" ::"a"(acc1[nt][ct]));
            if constexpr (N2 > 0) asm volatile("" ::"a"(acc2[0]), "a"(acc2[1]));
        }
        WS_STAMP(6);
    }
    if (oob_seen && a.oob) *a.oob = 1;
    // the last tile's deferred dot and output
    if (blockIdx.x < ntiles) {
        float part1 = 0.f;
#pragma unroll
        for (int d = 0; d < 4 * NPREV; ++d)
            part1 = dot_quarter(prev1(d / 4), (N2 > 0 ? 32 * w : 64 * w + 32 * (d / 4)), d % 4, part1);
        publish(prev0, part1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        store_out(tile - stride);
    }
}

// Wp[q][nt][lane][8] (bf16, RNE from fp32) with
//   natural order   (layer 1): element j = W[32nt + (lane&31)][16q + 8(lane>>5) + j]
//   permuted order  (layer 2): element j = W[32nt + (lane&31)][32(q>>1) + 16(q&1) + 8(j>>2) + 4(lane>>5) + (j&3)]
// and, for the 16x16x32 MFMA shape of the 8-wave kernel (modes 2 / 3; nt counts 16-row tiles, q 32-wide k-steps):
//   natural order   (layer 1): element j = W[16nt + (lane&15)][32q + 8(lane>>4) + j]
//   permuted order  (layer 2): element j = W[16nt + (lane&15)][32q + 16(j>>2) + 4(lane>>4) + (j&3)]
__global__ void pack_weight_bf16_kernel(const float* __restrict__ W, int N, int K, int mode, unsigned short* __restrict__ Wp) {
    const int64_t total = (int64_t)N * K;
    const int rows = mode >= 2 ? 16 : 32;
    const int NT = N / rows;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(o & 7);
        const int lane = (int)((o >> 3) & 63);
        const int64_t t = o >> 9;  // q*NT + nt
        const int nt = (int)(t % NT), q = (int)(t / NT);
        int k, row;
        if (mode < 2) {
            const int h = lane >> 5;
            k = mode ? 32 * (q >> 1) + 16 * (q & 1) + 8 * (j >> 2) + 4 * h + (j & 3) : 16 * q + 8 * h + j;
            row = 32 * nt + (lane & 31);
        } else {
            const int kg = lane >> 4;
            k = mode == 3 ? 32 * q + 16 * (j >> 2) + 4 * kg + (j & 3) : 32 * q + 8 * kg + j;
            row = 16 * nt + (lane & 15);
        }
        const float v = W[(int64_t)row * K + k];
        const __bf16 b = (__bf16)v;  // round-to-nearest-even (v_cvt_pk_bf16_f32)
        Wp[o] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

__global__ void copy_or_zero_f32_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : 0.f;
}

struct Bf16Blob {
    size_t wp1, b1, wp2, b2, wl, bl, zeros, wp1m, wp2m, total;  // byte offsets (wp1m / wp2m: 16x16x32 fragments, three-layer MLPs only)
};
static Bf16Blob bf16_blob(const int* dims, int n_layers) {
    Bf16Blob L{};
    const size_t K0 = dims[0], N1 = dims[1];
    size_t off = 0;
    L.wp1 = off; off += N1 * K0 * 2;
    L.b1 = off; off += N1 * 4;
    size_t last = N1;
    if (n_layers == 3) {
        const size_t N2 = dims[2];
        L.wp2 = off; off += N2 * N1 * 2;
        L.b2 = off; off += N2 * 4;
        last = N2;
    }
    L.wl = off; off += last * 4;
    L.bl = off; off += 16;
    L.zeros = off; off += 512;  // source of the rows of out-of-range ids (weight-stationary kernels: 4 x 128-byte units)
    if (n_layers == 3) {
        L.wp1m = off; off += N1 * K0 * 2;
        L.wp2m = off; off += (size_t)dims[2] * N1 * 2;
    }
    L.total = off;
    return L;
}

template <int K0, int N1, int N2>
static void launch_bf16(const Bf16Args& a, hipStream_t s) {
    const int64_t tiles = (a.B + 31) / 32;
    hipLaunchKernelGGL((score_fused_bf16_kernel<K0, N1, N2>), dim3((unsigned)((tiles + NCF_BF16_WGW - 1) / NCF_BF16_WGW)), dim3(NCF_BF16_WGW * 64), 0, s, a);
}

template <int K0, int N1, int N2>
static void launch_ws_bf16(const Bf16Args& a, const unsigned char* zeros, hipStream_t s) {
    const int64_t ntiles = (a.B + 63) / 64;
    const int64_t grid = ntiles < num_cus() ? ntiles : num_cus();
    hipLaunchKernelGGL((score_ws_bf16_kernel<K0, N1, N2>), dim3((unsigned)grid), dim3(256), 0, s, a, a.idxA, a.idxB, a.out, zeros, ntiles);
}

#define NCF_BF16_INSTANCES(X) X(256, 256, 128) X(256, 256, 0) X(128, 256, 128) X(128, 256, 0)

static bool bf16_dispatch(int K0, int N1, int N2, const Bf16Args* a, const unsigned char* zeros, bool ws, hipStream_t s) {
#define X(k, n1, n2) \
    if (K0 == k && N1 == n1 && N2 == n2) { \
        if (a) { if (ws) launch_ws_bf16<k, n1, n2>(*a, zeros, s); else launch_bf16<k, n1, n2>(*a, s); } \
        return true; \
    }
    NCF_BF16_INSTANCES(X)
#undef X
    return false;
}

bool bf16_shape_ok(int EA, int EB, int n_layers, const int* dims) {
    if (!dims || (n_layers != 2 && n_layers != 3) || dims[n_layers] != 1) return false;
    if (EA <= 0 || EB < 0 || EA % 16 || EB % 16 || EA + EB != dims[0]) return false;
    return bf16_dispatch(dims[0], dims[1], n_layers == 3 ? dims[2] : 0, nullptr, nullptr, false, nullptr);
}

size_t bf16_packed_bytes(int n_layers, const int* dims) {
    if (!dims || (n_layers != 2 && n_layers != 3)) return 0;
    return bf16_blob(dims, n_layers).total;
}

int bf16_pack(int n_layers, const int* dims, const void* const* W, const void* const* b, void* packed, size_t packed_bytes,
              hipStream_t s) {
    if (dims[n_layers] != 1) return fail(NCF_EUNSUPPORTED, "ncf_mlp_pack(bf16): last layer must be 1 wide");
    for (int i = 0; i < n_layers - 1; ++i)
        if (dims[i] % 32 || dims[i + 1] % 32)
            return fail(NCF_EUNSUPPORTED, "ncf_mlp_pack(bf16): layer %d dims (%d -> %d) not tileable by 32", i, dims[i], dims[i + 1]);
    const Bf16Blob L = bf16_blob(dims, n_layers);
    if (packed_bytes < L.total) return fail(NCF_EWORKSPACE, "ncf_mlp_pack(bf16): packed buffer too small");
    char* P = (char*)packed;
    auto bias = [&](int i) { return b ? (const float*)b[i] : nullptr; };
    hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(256), dim3(256), 0, s, (const float*)W[0], dims[1], dims[0], 0, (unsigned short*)(P + L.wp1));
    hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3((dims[1] + 255) / 256), dim3(256), 0, s, bias(0), dims[1], (float*)(P + L.b1));
    int last = dims[1];
    if (n_layers == 3) {
        hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(256), dim3(256), 0, s, (const float*)W[1], dims[2], dims[1], 1, (unsigned short*)(P + L.wp2));
        hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(256), dim3(256), 0, s, (const float*)W[0], dims[1], dims[0], 2, (unsigned short*)(P + L.wp1m));
        hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(256), dim3(256), 0, s, (const float*)W[1], dims[2], dims[1], 3, (unsigned short*)(P + L.wp2m));
        hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3((dims[2] + 255) / 256), dim3(256), 0, s, bias(1), dims[2], (float*)(P + L.b2));
        last = dims[2];
    }
    hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3((last + 255) / 256), dim3(256), 0, s, (const float*)W[n_layers - 1], last, (float*)(P + L.wl));
    hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3(1), dim3(256), 0, s, bias(n_layers - 1), 1, (float*)(P + L.bl));
    hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3(1), dim3(256), 0, s, (const float*)nullptr, 128, (float*)(P + L.zeros));
    return check_launch("ncf_mlp_pack(bf16)");
}

int bf16_score(const void* tabA, int64_t rowsA, int64_t ldA, const void* tabB, int64_t rowsB, int64_t ldB, const int64_t* idxA,
               const int64_t* idxB, int64_t B, int EA, int EB, int n_layers, const int* dims, const void* packed, float* out,
               int32_t* oob, hipStream_t s) {
    if (ldA < EA || (EB > 0 && ldB < EB) || ldA % 8 || (EB > 0 && ldB % 8) || !aligned16(tabA) || (EB > 0 && !aligned16(tabB)) || !aligned16(packed))
        return fail(NCF_EINVAL, "ncf_score_fused(bf16): tables must be 16-byte aligned with ld %% 8 == 0");
    const Bf16Blob L = bf16_blob(dims, n_layers);
    const char* P = (const char*)packed;
    Bf16Args a;
    a.tabA = (const unsigned short*)tabA; a.rowsA = rowsA; a.ldA = ldA;
    a.tabB = (const unsigned short*)(EB ? tabB : tabA); a.rowsB = EB ? rowsB : rowsA; a.ldB = EB ? ldB : ldA;
    a.idxA = idxA; a.idxB = idxB; a.B = B; a.EA = EA;
    a.Wp1 = (const unsigned short*)(P + L.wp1); a.b1 = (const float*)(P + L.b1);
    a.Wp2 = n_layers == 3 ? (const unsigned short*)(P + L.wp2) : nullptr;
    a.b2 = n_layers == 3 ? (const float*)(P + L.b2) : nullptr;
    a.Wp1m = n_layers == 3 ? (const unsigned short*)(P + L.wp1m) : nullptr;
    a.Wp2m = n_layers == 3 ? (const unsigned short*)(P + L.wp2m) : nullptr;
    a.wl = (const float*)(P + L.wl); a.bl = (const float*)(P + L.bl);
    a.out = out; a.oob = oob;
#if NCF_BF16_STAMP
    a.dbg = g_bf16_dbg;
#endif
    // Kernel choice (ncf_set_option("bf16_kernel", 1 = ws | 2 = stream | 3 = ws8) overrides it; tests run every batch size through all):
    //   * ws8, the 8-wave weight-stationary kernel (mlp_bf16_ws8.hip), for the NCF shape it is built for: two tables of equal width
    //     (EA = EB = K0/2 in {64, 128}), MLP K0-256-128-1, ids given.  tools/ab_bf16_opt.py (E = 128, 4 M + 1 M rows), ws8 vs ws vs
    //     stream: 16 384 pairs 10.1 / 10.9 / 14.4 us, 32 768: 11.9 / 12.8 / 15.0, 65 536: 16.9 / 18.3 / 18.0, 131 072: 28.3 / 30.2 / 34.4,
    //     262 144: 51.7 / 54.9 / 71.8, 1 M: 174-181 / 210-222 / 288, 4 M: 694-696 / 821-870 / 1 115; below 16 384 pairs all are launch-bound;
    //   * ws, the 4-wave weight-stationary kernel, whenever the row widths are whole 128-byte units and ids are given;
    //   * the slab-streaming kernel otherwise.
    const int force = option(NCF_OPT_BF16_KERNEL);
    bool ws = NCF_BF16_WS && B >= NCF_BF16_WS_MIN_PAIRS;
    if (force) ws = force == 1 || force == 3;
    ws = ws && EA % 64 == 0 && EB % 64 == 0 && idxA && (EB == 0 || idxB);
    if (ws && !idxB) a.idxB = idxA;   // single table: the id DMA's table-B lanes fetch valid (unused) words
    const int N2 = n_layers == 3 ? dims[2] : 0;
    if (ws && force != 1 && EA == EB && B < (int64_t(1) << 31) - 64 && rowsA < (int64_t(1) << 31) && rowsB < (int64_t(1) << 31) &&
        ldA < (int64_t(1) << 30) && ldB < (int64_t(1) << 30) && ws8_shape_ok(dims[0], dims[1], N2)) {
        launch_ws8_bf16(dims[0], a, (const unsigned char*)(P + L.zeros), s);
        return check_launch("ncf_score_fused(bf16)");
    }
    bf16_dispatch(dims[0], dims[1], n_layers == 3 ? dims[2] : 0, &a, (const unsigned char*)(P + L.zeros), ws, s);
    return check_launch("ncf_score_fused(bf16)");
}

}  // namespace ncf

#if NCF_BF16_STAMP
extern "C" void ncf_dev_set_bf16_debug_buffer(void* p) { ncf::g_bf16_dbg = (unsigned long long*)p; }
#endif
