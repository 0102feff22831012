// K1+K2 fused, bf16 tables + bf16 weights, fp32 accumulate: the 8-wave weight-stationary persistent kernel (gfx950).
//
// Why a second weight-stationary form: the 4-wave kernel of mlp_bf16.hip keeps ONE wave per SIMD (the weights fill its
// registers), and a lone wave issues one instruction per ~4-5 cycles — its ~800 instructions per 64-pair tile cost as much
// as the tile's 96 MFMAs (3 072 cycles) before any stall.  Here a 512-thread workgroup puts TWO waves on every SIMD, so one
// wave's VALU / LDS / DMA / scalar work issues under its partner's MFMAs.  The price is 256 registers per wave, which
// forces an 8-way split of the weights:
//   * layer 1: wave w keeps neurons [32w, 32w+32) (Q1 A fragments = 64 registers at K0 = 256) and runs them over BOTH
//     32-pair column tiles of the 64-pair tile: 2 x Q1 MFMAs (v_mfma_f32_32x32x16_bf16);
//   * layer 2: wave w keeps neurons [32(w&3), +32) (16 A fragments = 64 registers; the two waves of a SIMD hold the same
//     slice) and runs them over ONE column tile, ct = w >> 2: 16 MFMAs.
// 48 MFMAs per wave and tile = 96 per SIMD, the same matrix work as before.  What crosses LDS is unchanged (the X image
// filled by LDS-DMA in 8-pair x 128-byte pieces, three tiles deep; H1 as ready-made layer-2 B fragments; the formats are
// the 4-wave kernel's, so W1 / W2 keep their packing), but every wave now reads all of X and half of H1: 384 KB of
// ds_read_b128 per tile and CU = 1 536 LDS cycles of the 3 072.
//
// Schedule.  "Team 0" = waves 0-3, "team 1" = waves 4-7 (SIMD partners: w and w+4).  Team 1 runs layer 1 of column tile 0
// one tile ahead, so that both teams carry the same MFMA count between the two barriers of a tile and H1 needs one buffer:
//     S1(t):  team 0:  L1(t) ct0  | dot(t-1) ct0 -> red      team 1:  L2(t-1) ct1 | pack ct0(t) -> H1
//                      L1(t) ct1  | pack ct0(t) -> H1                 L1(t) ct1   | dot(t-1) ct1 -> red
//     -- wait rows(t+1) -- barrier alpha(t): H1(ct0)(t), red(t-1), X(t+1) visible; X(t) is free --
//     S2(t):  team 0:  L2(t) ct0  | pack ct1(t) -> H1        team 1:  L1(t+1) ct0 | pack ct1(t) -> H1
//                      both: out(t-1) (wave 0), row pointers(t+3), ids DMA(t+4), row DMAs(t+3) into X(t)'s buffer
//     -- barrier beta(t): H1(ct1)(t) visible --
// Vector-memory queue of a wave, oldest first: ... rows(t+1) | [out] ids(t+3) rows(t+2) | [out] ids(t+4) rows(t+3): at
// alpha(t) `s_waitcnt vmcnt(NCU)` leaves only rows(t+2) in flight.  DMAs are issued only for tiles that exist.
#include "mlp_bf16.h"
#include <type_traits>

#ifndef NCF_WS8_RING
#define NCF_WS8_RING 4        // B-fragment register ring of a stream (reads run RING-1 k-steps ahead of their MFMA)
#endif

namespace ncf {

template <int K0>
struct Ws8Layout {
    static constexpr int N1 = 256, N2 = 128;
    static constexpr int P = 64, CTN = 2;
    static constexpr int NCU = K0 / 64;                   // 128-byte units of the concatenated row
    static constexpr int CT_BYTES = NCU * 4 * 1024;       // X image of one 32-pair column tile
    static constexpr int XBUF = CTN * CT_BYTES;
    static constexpr int Q1 = K0 / 16, Q2 = N1 / 16;
    static constexpr int OFF_H1 = 3 * XBUF;
    static constexpr int H1_BYTES = CTN * Q2 * 1024;
    static constexpr int OFF_RED = OFF_H1 + H1_BYTES;     // float red[64 pairs][4 neuron slices][2 lane halves]
    static constexpr int OFF_B1 = OFF_RED + 2048;
    static constexpr int OFF_B2 = OFF_B1 + N1 * 4;
    static constexpr int OFF_WL = OFF_B2 + N2 * 4;
    static constexpr int OFF_IDS = OFF_WL + N2 * 4;       // [8 waves][2 slots][2 tables][8 pairs] int64, twice (lanes 32-63 repeat)
    static constexpr int OFF_STAMP = OFF_IDS + 8 * 2 * 256;   // diagnostic builds: [8 waves][8 iterations][8 stamps] u64
    static constexpr int TOTAL = OFF_STAMP + (NCF_BF16_STAMP ? 4096 : 0);
};

#if NCF_BF16_STAMP
#define W8_STAMP(k) do { if (a.dbg && lane == 0 && it >= 8 && it < 16) \
    reinterpret_cast<unsigned long long*>(lds + L::OFF_STAMP)[(w * 8 + (it - 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W8_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define W8_MFMA(n) __builtin_amdgcn_sched_group_barrier(0x008, n, 0)
#define W8_VALU(n) __builtin_amdgcn_sched_group_barrier(0x002, n, 0)
#define W8_DSR(n) __builtin_amdgcn_sched_group_barrier(0x100, n, 0)

template <int K0>
__global__ __launch_bounds__(512, 1) void score_ws8_bf16_kernel(Bf16Args a, const int64_t* __restrict__ idxA,
                                                                const int64_t* __restrict__ idxB, float* __restrict__ out,
                                                                const unsigned char* __restrict__ zeros, int ntiles) {
    using L = Ws8Layout<K0>;
    constexpr int Q1 = L::Q1, Q2 = L::Q2, NCU = L::NCU, N1 = L::N1, N2 = L::N2;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[L::TOTAL];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int team = w >> 2, g = w & 3;                      // g: pair group of the DMA pieces AND layer-2 neuron slice
    const int m = lane & 31, h = lane >> 5;
    const int ncuA = a.EA / 64;
    const int Bp = (int)a.B;                                 // the launcher keeps B below 2^31 for this kernel
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
    // X image: piece (unit cu, pair group g) = 8 pairs x 128 B, pair-major, the 16-byte chunks of pair p stored at slot
    // chunk ^ p ^ (g >= 2): a DMA instruction's eight consecutive lanes then cover ONE 128-byte line of one row, and the
    // B-fragment read of k-step s (lane (m, h) <- chunk 2s+h of pair m) stays a conflict-free ds_read_b128.
    const int ml = lane >> 3;
    const int cl = (lane & 7) ^ ml ^ (g >= 2 ? 1 : 0);
    const int gm = m >> 3;
    const int key = (m & 7) ^ (gm >= 2 ? 1 : 0);
    unsigned rd4[4];                                          // reader offset of k-steps with s & 3 = j (the XOR does not commute with +)
#pragma unroll
    for (int j = 0; j < 4; ++j) rd4[j] = gm * 1024 + (m & 7) * 128 + (((2 * j + h) ^ key) & 7) * 16;

    const int stride = gridDim.x;
    const int tile0 = blockIdx.x;
    const bool e1 = tile0 + stride < ntiles, e2 = tile0 + 2 * stride < ntiles, e3 = tile0 + 3 * stride < ntiles;
    const int pw = 8 * w + ml;                               // this lane's pair inside a tile (DMA role)

    // ---- prologue: ids of tiles 0..2 -> row DMAs, THEN the weights (their wait drains the queue once, rows included)
    int64_t pid[3][2];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int p0 = (tile0 + t * stride) * L::P + pw;
        const int p = p0 < Bp ? p0 : Bp - 1;
        pid[t][0] = idxA[p];
        pid[t][1] = idxB[p];
    }
    typedef const unsigned char* RowSrc[2];
    bool oob_seen = false;
    auto resolve = [&](RowSrc& src, int p0, int64_t ia, int64_t ib) {
        const bool okA = (ia >= 0) & (ia < a.rowsA);
        const bool okB = (ncuA == NCU) | ((ib >= 0) & (ib < a.rowsB));
        oob_seen |= !(okA & okB) & (p0 < Bp);
        src[0] = (okA ? reinterpret_cast<const unsigned char*>(a.tabA + ia * a.ldA) : zeros) + cl * 16;
        src[1] = ((okB & (ncuA < NCU)) ? reinterpret_cast<const unsigned char*>(a.tabB + ib * a.ldB) : zeros) + cl * 16;
    };
    auto issue_rows = [&](const RowSrc& src, int slot) {
#pragma unroll
        for (int cu = 0; cu < NCU; ++cu) {
            const bool fromA = cu < ncuA;
            const unsigned char* gp = (fromA ? src[0] : src[1]) + (fromA ? cu : cu - ncuA) * 128;
            dma16(gp, lds0 + slot * L::XBUF + team * L::CT_BYTES + (cu * 4 + g) * 1024);
        }
    };
    // a tile's 16 ids per wave (2 tables x 8 pairs) arrive by one dword LDS-DMA (scalar base + 32-bit lane offset)
    const int ids_dw = lane & 15, ids_tb = (lane >> 4) & 1;
    auto ids_dma = [&](int t, int slot) {
        const int p0 = t * L::P + 8 * w + (ids_dw >> 1);
        const int p = p0 < Bp ? p0 : Bp - 1;
        const int* gp = reinterpret_cast<const int*>((ids_tb ? idxB : idxA) + p) + (ids_dw & 1);
        dma4(gp, lds0 + L::OFF_IDS + (w * 2 + slot) * 256);
    };
    auto locate = [&](RowSrc& src, int t, int slot) {
        const unsigned char* q = lds + L::OFF_IDS + (w * 2 + slot) * 256 + ml * 8;
        const int64_t ia = *reinterpret_cast<const int64_t*>(q);
        const int64_t ib = *reinterpret_cast<const int64_t*>(q + 64);
        resolve(src, t * L::P + pw, ia, ib);
    };
    {
        RowSrc s0, s1, s2;
        resolve(s0, tile0 * L::P + pw, pid[0][0], pid[0][1]);
        resolve(s1, (tile0 + stride) * L::P + pw, pid[1][0], pid[1][1]);
        resolve(s2, (tile0 + 2 * stride) * L::P + pw, pid[2][0], pid[2][1]);
        __builtin_amdgcn_sched_barrier(0);
        issue_rows(s0, 0);
        if (e1) issue_rows(s1, 1);
        if (e3) ids_dma(tile0 + 3 * stride, 1);
        if (e2) issue_rows(s2, 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    u32x4 wa1[Q1], wa2[Q2];
#pragma unroll
    for (int s = 0; s < Q1; ++s)
        wa1[s] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp1) + ((size_t)(s * (N1 / 32) + w) * 64 + lane) * 16);
#pragma unroll
    for (int q = 0; q < Q2; ++q)
        wa2[q] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp2) + ((size_t)(q * (N2 / 32) + g) * 64 + lane) * 16);
    {
        float* lb1 = reinterpret_cast<float*>(lds + L::OFF_B1);
        float* lb2 = reinterpret_cast<float*>(lds + L::OFF_B2);
        float* lwl = reinterpret_cast<float*>(lds + L::OFF_WL);
        if (threadIdx.x < N1) lb1[threadIdx.x] = a.b1[threadIdx.x];
        if (threadIdx.x < N2) { lb2[threadIdx.x] = a.b2[threadIdx.x]; lwl[threadIdx.x] = a.wl[threadIdx.x]; }
    }
    const float bl = a.bl[0];
    // the compiler's own waits for the weights come here (it does not see the DMAs: its vmcnt(N) drains them as well)
#pragma unroll
    for (int s = 0; s < Q1; ++s) asm volatile("" ::"v"(wa1[s]));
#pragma unroll
    for (int q = 0; q < Q2; ++q) asm volatile("" ::"v"(wa2[q]));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();

    // ---- pieces of the steady state ----
    constexpr int RING = NCF_WS8_RING, AHEAD = RING - 1;
    const unsigned char* const hbase = lds + L::OFF_H1 + lane * 16;
    unsigned char* const hwr = lds + L::OFF_H1 + (2 * w) * 1024 + lane * 16;          // this wave's two H1 fragments of a column tile
    const unsigned char* const wlp = lds + L::OFF_WL + (32 * g + 4 * h) * 4;
    float* const red = reinterpret_cast<float*>(lds + L::OFF_RED);

    auto bias_tile = [&](int off, int neuron0) {
        f32x16 t;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + off + (neuron0 + 8 * gq + 4 * h) * 4);
            t[4 * gq + 0] = bb[0]; t[4 * gq + 1] = bb[1]; t[4 * gq + 2] = bb[2]; t[4 * gq + 3] = bb[3];
        }
        return t;
    };
    // layer 1 of column tile ct from the X image of the buffer at byte offset xb; fill(s) = work placed after k-step s
    auto l1_stream = [&](f32x16& acc, unsigned xb, int ct, auto&& fill) {
        u32x4 fr[RING];
        auto xp = [&](int s) { return reinterpret_cast<const u32x4*>(lds + (rd4[s & 3] + xb) + ct * L::CT_BYTES + (s >> 2) * 4096); };
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) fr[j] = *xp(j);
#pragma unroll
        for (int s = 0; s < Q1; ++s) {
            if (s + AHEAD < Q1) fr[(s + AHEAD) % RING] = *xp(s + AHEAD);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa1[s]), as_bf16x8(fr[s % RING]), acc, 0, 0, 0);
            fill(s);
        }
    };
    // layer 2 of column tile ct from H1
    auto l2_stream = [&](f32x16& acc, int ct, auto&& fill) {
        u32x4 fr[RING];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) fr[j] = *reinterpret_cast<const u32x4*>(hbase + (ct * Q2 + j) * 1024);
#pragma unroll
        for (int q = 0; q < Q2; ++q) {
            if (q + AHEAD < Q2) fr[(q + AHEAD) % RING] = *reinterpret_cast<const u32x4*>(hbase + (ct * Q2 + q + AHEAD) * 1024);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa2[q]), as_bf16x8(fr[q % RING]), acc, 0, 0, 0);
            fill(q);
        }
    };
    // ReLU + bf16 of half an accumulator tile -> one ready-made layer-2 B fragment in H1
    auto pack_half = [&](const f32x16& acc, int ct, int s2) {
        *reinterpret_cast<bf16x8_t*>(hwr + (ct * Q2 + s2) * 1024) = pack_relu8_int(acc, 8 * s2);
    };
    auto dot_quarter = [&](const f32x16& acc, int gq, float part) {
        const f32x4 ww = *reinterpret_cast<const f32x4*>(wlp + 32 * gq);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) part = fmaf(ww[jj], relu1(acc[4 * gq + jj]), part);
        return part;
    };
    auto publish = [&](float part) { red[((team * 32 + m) * 4 + g) * 2 + h] = part; };
    auto store_out = [&](int t) {
        if (w == 0) {
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(red + lane * 8);
            const f32x4 r1 = *reinterpret_cast<const f32x4*>(red + lane * 8 + 4);
            const int p = t * L::P + lane;
            const float v = ((r0[0] + r0[1]) + (r0[2] + r0[3])) + ((r1[0] + r1[1]) + (r1[2] + r1[3])) + bl;
            if (p < Bp) out[p] = v;
        }
    };
    // positions of the fillers inside a stream of Q k-steps
    auto at = [](int s, int Q, int num, int den) { return s == (Q * num) / den; };
    const int win = team == 0 ? 2 * g : (2 * g + 5) & 7;      // this wave's DMA window (SIMD partners w, w+4: three or more apart)

    // The two teams run separate loops (a join inside the loop costs accumulator copies).
    auto run = [&](auto team_tag) {
        constexpr int TEAM = decltype(team_tag)::value;
        f32x16 acc1[2], acc2;
        acc1[0] = bias_tile(L::OFF_B1, 32 * w);
        if constexpr (TEAM == 1) l1_stream(acc1[0], 0u, 0, [&](int) {});   // peeled: team 1 is one column tile ahead on layer 1
        acc1[1] = bias_tile(L::OFF_B1, 32 * w);
        acc2 = bias_tile(L::OFF_B2, 32 * g);                  // team 0, it = 0: the (unused) dot of a tile that does not exist
        int buf = 0, tile = tile0;
        for (int it = 0; tile < ntiles; ++it, tile += stride) {
            const int nbuf = buf == 2 ? 0 : buf + 1, pbuf = buf == 0 ? 2 : buf - 1;
            const unsigned xb = buf * L::XBUF, xbn = nbuf * L::XBUF;
            const bool has2 = tile + 2 * stride < ntiles, has3 = tile + 3 * stride < ntiles, has4 = tile + 4 * stride < ntiles;
            float part = 0.f;
            // The row DMAs of a tile are spread over the eight waves IN TIME (an LDS-DMA of gathered rows can hold its wave for
            // hundreds of cycles).  Wave w owns window `win`; windows 0-1 lie in S2(t) (rows of tile t+3 into X(t)'s buffer, free
            // since alpha(t)), windows 2-7 in S1(t) (rows of tile t+2 into X(t-1)'s buffer): the same cohort of tiles, and in both
            // cases the wave's newest NCU queue entries at alpha are the rows issued last, so the counted wait is the same.
            auto fetch = [&](int trow, int tids, bool do_rows, bool do_ids, int slot_x, int slot_ids) {
                if (do_rows) {
                    RowSrc src;
                    locate(src, trow, slot_ids);
                    if (do_ids) ids_dma(tids, slot_ids ^ 1);
                    issue_rows(src, slot_x);
                }
            };
            auto win_s2 = [&](int k) { if (win == k) fetch(tile + 3 * stride, tile + 4 * stride, has3, has4, buf, (it + 1) & 1); };
            auto win_s1 = [&](int k) { if (win == k && it > 0) fetch(tile + 2 * stride, tile + 3 * stride, has2, has3, pbuf, it & 1); };
            auto dot_fill = [&](int s) {
                if (s % (Q1 / 4) == Q1 / 4 - 1 && s < Q1 - 1) part = dot_quarter(acc2, s / (Q1 / 4), part);
                if (s == Q1 - 1) { part = dot_quarter(acc2, 3, part); publish(part); }
            };
            auto alpha = [&]() {
                W8_STAMP(2);
                if (has2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NCU) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                W8_STAMP(3);
                wg_barrier();
                W8_STAMP(4);
            };
            auto beta = [&]() {
                W8_STAMP(5);
                wg_barrier();
                W8_STAMP(6);
            };
            W8_STAMP(0);
            if constexpr (TEAM == 0) {
                // ---- S1 ----
                l1_stream(acc1[0], xb, 0, [&](int s) {
                    if (s == 0) win_s1(2);
                    if (at(s, Q1, 6, 8)) win_s1(4);
                    dot_fill(s);
                    if (s == Q1 - 1) acc1[1] = bias_tile(L::OFF_B1, 32 * w);
                });
                W8_STAMP(1);
                l1_stream(acc1[1], xb, 1, [&](int s) {
                    if (at(s, Q1, 1, 4)) pack_half(acc1[0], 0, 0);
                    if (at(s, Q1, 4, 8)) win_s1(6);
                    if (at(s, Q1, 3, 4)) pack_half(acc1[0], 0, 1);
                    if (s == Q1 - 1) acc2 = bias_tile(L::OFF_B2, 32 * g);
                });
                alpha();
                // ---- S2 ----
                l2_stream(acc2, 0, [&](int q) {
                    if (q == 0 && it > 0) store_out(tile - stride);
                    if (q == 1) win_s2(0);
                    if (at(q, Q2, 1, 4)) pack_half(acc1[1], 1, 0);
                    if (at(q, Q2, 3, 4)) pack_half(acc1[1], 1, 1);
                    if (q == Q2 - 1) acc1[0] = bias_tile(L::OFF_B1, 32 * w);
                });
                beta();
            } else {
                // ---- S1 ----
                l2_stream(acc2, 1, [&](int q) {
                    if (at(q, Q2, 1, 4)) pack_half(acc1[0], 0, 0);
                    if (at(q, Q2, 3, 8)) win_s1(3);
                    if (at(q, Q2, 3, 4)) pack_half(acc1[0], 0, 1);
                    if (q == Q2 - 1) acc1[1] = bias_tile(L::OFF_B1, 32 * w);
                });
                W8_STAMP(1);
                l1_stream(acc1[1], xb, 1, [&](int s) {
                    if (at(s, Q1, 1, 8)) win_s1(5);
                    if (at(s, Q1, 6, 8)) win_s1(7);
                    dot_fill(s);
                    if (s == Q1 - 1) acc1[0] = bias_tile(L::OFF_B1, 32 * w);
                });
                alpha();
                // ---- S2 ----
                l1_stream(acc1[0], xbn, 0, [&](int s) {
                    if (at(s, Q1, 1, 4)) pack_half(acc1[1], 1, 0);
                    if (at(s, Q1, 5, 8)) win_s2(1);
                    if (at(s, Q1, 3, 4)) pack_half(acc1[1], 1, 1);
                    if (s == Q1 - 1) acc2 = bias_tile(L::OFF_B2, 32 * g);
                });
                beta();
            }
            buf = nbuf;
        }
        // the last tile: team 1 still owes layer 2 of its column tile; then both teams' dots and the outputs
        float part = 0.f;
        if constexpr (TEAM == 1) l2_stream(acc2, 1, [&](int) {});
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) part = dot_quarter(acc2, gq, part);
        publish(part);
        wg_barrier();
        store_out(tile - stride);
    };
    if (team == 0) run(std::integral_constant<int, 0>{});
    else run(std::integral_constant<int, 1>{});
    if (oob_seen && a.oob) *a.oob = 1;
#if NCF_BF16_STAMP
    if (a.dbg) {
        wg_barrier();
        a.dbg[(int64_t)blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<unsigned long long*>(lds + L::OFF_STAMP)[threadIdx.x];
    }
#endif
}

bool ws8_shape_ok(int K0, int N1, int N2) { return (K0 == 256 || K0 == 128) && N1 == 256 && N2 == 128; }

void launch_ws8_bf16(int K0, const Bf16Args& a, const unsigned char* zeros, hipStream_t s) {
    const int ntiles = (int)((a.B + 63) / 64);
    const int grid = ntiles < num_cus() ? ntiles : num_cus();
    if (K0 == 256)
        hipLaunchKernelGGL((score_ws8_bf16_kernel<256>), dim3((unsigned)grid), dim3(512), 0, s, a, a.idxA, a.idxB, a.out, zeros, ntiles);
    else
        hipLaunchKernelGGL((score_ws8_bf16_kernel<128>), dim3((unsigned)grid), dim3(512), 0, s, a, a.idxA, a.idxB, a.out, zeros, ntiles);
}

}  // namespace ncf
