// K1+K2 fused, bf16 tables + bf16 weights, fp32 accumulate: the 8-wave weight-stationary persistent kernel (gfx950).
//
// Why a second weight-stationary form: the 4-wave kernel of mlp_bf16.hip keeps ONE wave per SIMD (the weights fill its
// registers), and a lone wave issues one instruction per ~4-5 cycles — its ~800 instructions per 64-pair tile cost as much
// as the tile's 96 MFMAs (3 072 cycles) before any stall.  Here a 512-thread workgroup puts TWO waves on every SIMD and
// splits the ROLES between them, so that neither needs more than 256 registers:
//   * "A" waves 0-3 own layer 1: wave a keeps neurons [64a, 64a+64) (2 x Q1 A fragments = 128 registers at K0 = 256) and does
//     nothing but read X fragments, issue MFMAs and turn finished accumulators into H1 (ReLU, bf16, ready-made layer-2 B
//     fragments).  No vector-memory instruction: an A wave never stalls behind an LDS-DMA.
//   * "B" waves 4-7 own layer 2 and all I/O: wave 4+b keeps neurons [32b, 32b+32) of layer 2 (16 A fragments), runs them over
//     H1, does the last layer's dot on its accumulators, and issues every LDS-DMA (ids, gathered rows).  It needs a third of
//     the SIMD's matrix pipe, so the ~100-200 cycles an LDS-DMA of random rows holds its wave are affordable there.
// The unit of work is 32 pairs (one MFMA column tile).  Per SIMD and unit: 32 (A) + 16 (B) MFMAs = 1 536 matrix-pipe cycles;
// per CU and unit 4 x 16 KB (X, each fragment feeds two MFMAs) + 4 x 16 KB (H1) of ds_read_b128 = 512 LDS cycles — the 4-wave
// kernel's traffic, a third less than an 8-way split of both layers.  The H1 format and the W1 / W2 packing are the 4-wave
// kernel's; inside an X piece the chunks are stored pair-major with an XOR swizzle (see below).
//
// Pipeline, one barrier per phase (phase k = unit k of this workgroup; unit j's stages):
//     phase j    A: layer 1 of unit j (X slot j % NU)            | pack(j-1) -> H1[(j-1)&1]; out(j-4) (wave 0)
//     phase j+1  A: pack(j) -> H1[j&1]
//     phase j+2  B: layer 2 of unit j from H1[j&1] -> acc2[j&1]
//     phase j+3  B: last-layer dot of unit j -> red[j&1]
//     phase j+4  A (wave 0): out(j) = sum of red[j&1] + bias
// and B, in phase k: row pointers of unit k+D from the ids that arrived D phases ago, ids DMA of unit k+2D, row DMAs of unit
// k+D into slot (k+D) % NU (D < NU: that slot's unit was consumed before phase k); before the barrier it waits until only
// the DMAs of the last D-2 phases are in flight, i.e. rows(k+2) have landed: A prefetches the first fragments of unit k+1 before
// the barrier of phase k and never waits for LDS after a barrier.
#include "mlp_bf16.h"
#include <type_traits>

#ifndef NCF_WS8_M16
#define NCF_WS8_M16 1         // 1 = v_mfma_f32_16x16x32_bf16 (the chip holds a higher clock on it), 0 = v_mfma_f32_32x32x16_bf16
#endif
#ifndef NCF_WS8_ABLATE
#define NCF_WS8_ABLATE 0      // diagnostics: 1 = no row DMAs, 2 / 3 = rows from a 1 MiB / 64 MiB window of table A
#endif
#ifndef NCF_WS8_PRIO
#define NCF_WS8_PRIO 1        // 1 = B waves at s_setprio 1 (their MFMAs slot in ahead of the A stream: measured best together with NA = 0), 2 = A waves, 0 = none
#endif
#ifndef NCF_WS8_RING
#define NCF_WS8_RING 4        // B-fragment register ring of a stream (reads run RING-1 k-steps ahead of their MFMA)
#endif
#ifndef NCF_WS8_RINGB
#define NCF_WS8_RINGB 8       // the same ring in the B waves' layer-2 stream (they have registers to spare and stall behind DMAs)
#endif
#ifndef NCF_WS8_STAGE
#define NCF_WS8_STAGE 0       // 1 = B gathers through registers (global_load_dwordx4 -> ds_write_b128, all compiler-visible) instead of LDS-DMA
#endif
#ifndef NCF_WS8_NA
#define NCF_WS8_NA 0          // row-DMA pieces (of the NCU per unit and pair group) issued by the A wave, after its MFMA stream (0: B issues all)
#endif
#ifndef NCF_WS8_SPREAD
#define NCF_WS8_SPREAD 3      // B: one row DMA every this many layer-2 k-steps (0 = all DMAs at the head of the phase)
#endif
#ifndef NCF_WS8_DIST
#define NCF_WS8_DIST 3        // D: row DMAs run this many units ahead of layer 1 (3 <= D < NU).  Round 3, measured in one session (3 / 4 / 5): 65 536 pairs
                              // 16.8 / 16.8 / 17.5-18.0 us, 131 072: 26.3 / 26.8 / 27.5, 262 144 to 4 M pairs equal within the noise (174 us at 1 M, 672-675 at 4 M):
                              // a workgroup of config 5's own batch has 8 units, and the prologue waits for the ids of 2 D of them
#endif

namespace ncf {

template <int K0>
struct Ws8Layout {
    static constexpr int N1 = 256, N2 = 128;
    static constexpr int UP = 32;                         // pairs per unit
    static constexpr int NCU = K0 / 64;                   // 128-byte units of the concatenated row
    static constexpr int UB = NCU * 4 * 1024;             // X image of one unit
    static constexpr int NU = NCF_WS8_STAGE ? 4 : 6, D = NCF_WS8_STAGE ? 3 : NCF_WS8_DIST;   // X ring slots, DMA distance
    static constexpr int Q1 = K0 / 16, Q2 = N1 / 16;
    static constexpr int OFF_H1 = NU * UB;
    static constexpr int H1_HALF = Q2 * 1024;
    static constexpr int OFF_RED = OFF_H1 + 2 * H1_HALF;  // float red[2][32 pairs][4 neuron slices][2 lane halves]
    static constexpr int OFF_B1 = OFF_RED + 4096;         // (16x16x32 form: [2][32 pairs][4 slices][4 lane groups])
    static constexpr int OFF_B2 = OFF_B1 + N1 * 4;
    static constexpr int OFF_WL = OFF_B2 + N2 * 4;
    static constexpr int IDS_SLOTS = D + 1;
    static constexpr int OFF_IDS = OFF_WL + N2 * 4;       // [4 B waves][D+1 slots][2 tables][8 pairs] int64, twice (lanes 32-63 repeat)
    static constexpr int OFF_STAMP = OFF_IDS + 4 * IDS_SLOTS * 256;   // diagnostic builds: [8 waves][8 phases][8 stamps] u64
    static constexpr int TOTAL = OFF_STAMP + (NCF_BF16_STAMP ? 4096 : 0);
    static_assert(D >= 3 && D < NU, "DMA distance");
};

#if NCF_BF16_STAMP
#define W8_STAMP(i) do { if (a.dbg && lane == 0 && k >= 16 && k < 24) \
    reinterpret_cast<unsigned long long*>(lds + L::OFF_STAMP)[(w * 8 + (k - 16)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W8_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int K0, bool M16>
__global__ __launch_bounds__(512, 1) void score_ws8_bf16_kernel(Bf16Args a, const int64_t* __restrict__ idxA,
                                                                const int64_t* __restrict__ idxB, float* __restrict__ out,
                                                                const unsigned char* __restrict__ zeros, int nunits) {
    using L = Ws8Layout<K0>;
    constexpr int Q1 = L::Q1, Q2 = L::Q2, NCU = L::NCU, N1 = L::N1, N2 = L::N2, NU = L::NU, D = L::D;
    constexpr int RING = NCF_WS8_RING, AHEAD = RING - 1;
    static_assert(Q1 % RING == 0 && Q2 % RING == 0, "the fragment ring keeps its phase from unit to unit");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[L::TOTAL];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = w & 3;                                     // A: layer-1 neuron slice; B: layer-2 neuron slice AND pair group of its DMA pieces
    const int m = lane & 31, h = lane >> 5;
    const int Bp = (int)a.B;                                 // the launcher keeps B below 2^31 for this kernel
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)lds;
    const int stride = gridDim.x, bid = blockIdx.x;
    const int n = (nunits - bid + stride - 1) / stride;      // units of this workgroup: bid, bid + stride, ...  (grid <= nunits)
    auto unit_pair0 = [&](int k) { return (bid + k * stride) * L::UP; };   // first pair of unit k

    // biases and the last layer's weights -> LDS (all waves); visible after the barrier that ends each role's prologue
    {
        float* lb1 = reinterpret_cast<float*>(lds + L::OFF_B1);
        float* lb2 = reinterpret_cast<float*>(lds + L::OFF_B2);
        float* lwl = reinterpret_cast<float*>(lds + L::OFF_WL);
        if (threadIdx.x < N1) lb1[threadIdx.x] = a.b1[threadIdx.x];
        if (threadIdx.x < N2) { lb2[threadIdx.x] = a.b2[threadIdx.x]; lwl[threadIdx.x] = a.wl[threadIdx.x]; }
    }
    auto bias_tile = [&](int off, int neuron0) {
        f32x16 t;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const f32x4 bb = *reinterpret_cast<const f32x4*>(lds + off + (neuron0 + 8 * gq + 4 * h) * 4);
            t[4 * gq + 0] = bb[0]; t[4 * gq + 1] = bb[1]; t[4 * gq + 2] = bb[2]; t[4 * gq + 3] = bb[3];
        }
        return t;
    };
    float* const red = reinterpret_cast<float*>(lds + L::OFF_RED);
    auto at = [](int s, int Q, int num, int den) { return s == (Q * num) / den; };

    // ---- the gather: ids and rows by LDS-DMA (helpers shared by both roles; wave g of either role serves pairs 8g .. 8g+7) ----
    // X image of a unit (EA = EB = K0/2): per table 32 / PP pieces of PP pairs x RC 16-byte chunks — every piece holds WHOLE table
    // rows, so that one DMA instruction reads PP rows, each by RC consecutive lanes (K0 = 256: 4 rows of 256 B; K0 = 128: 8 rows
    // of 128 B): fewest distinct lines per instruction.  (Measured against pieces of 8 half rows x 128 B: the same 184-185 us at
    // 1 M pairs — DESIGN 4.2 c.)  Inside a piece, chunk cc of pair m sits at slot cc ^ key(m) of the pair's RC slots, which
    // keeps the B-fragment read of a k-step (lane (m, h) <- chunk 2s+h of pair m's concatenated row) a conflict-free ds_read_b128:
    // its 16-lane groups hold 16 different m & 15 (RC = 16: key = m & 15) or 8 different keys in each 128-byte half (RC = 8).
    constexpr int NJ = NCU / 2;                              // pieces per table, wave and unit
    constexpr int PP = 8 / NJ;                               // pairs per piece
    constexpr int RC = 64 / PP;                              // 16-byte chunks per table row
    constexpr int TSTRIDE = (32 / PP) * 1024;                // bytes between the two tables' pieces inside a unit
    auto xkey = [](int mm) { return RC == 16 ? (mm & 15) : ((mm & 7) ^ ((mm >> 3) >= 2 ? 1 : 0)); };
    const int dpp = lane / RC, dpos = lane % RC;             // DMA role: pair dpp of the piece, slot dpos of its row
    typedef const unsigned char* RowSrc[2][NJ];              // [table][piece]
    bool oob_seen = false;
    // (row counts and row strides fit 32 bits — the launcher checks — so a row address is one v_mad_u64_u32 and the range check one
    // unsigned compare: a negative id is a huge unsigned one)
    const uint32_t strideA = (uint32_t)(a.ldA * 2), strideB = (uint32_t)(a.ldB * 2);
    auto resolve = [&](RowSrc& src, int j, int p0, int64_t ia, int64_t ib) {
        const int cc = dpos ^ xkey(8 * g + PP * j + dpp);
        const bool okA = (uint64_t)ia < (uint64_t)a.rowsA;
        const bool okB = (uint64_t)ib < (uint64_t)a.rowsB;
        oob_seen |= !(okA & okB) & (p0 < Bp);
        src[0][j] = (okA ? reinterpret_cast<const unsigned char*>(a.tabA) + (uint64_t)(uint32_t)ia * strideA : zeros) + cc * 16;
        src[1][j] = (okB ? reinterpret_cast<const unsigned char*>(a.tabB) + (uint64_t)(uint32_t)ib * strideB : zeros) + cc * 16;
    };
    auto issue_row = [&](const RowSrc& src, int slot, int i) {   // piece i of this wave: table i / NJ, pairs 8g + PP (i % NJ) ...
        const int tb = i / NJ, j = i % NJ;
        const unsigned char* gp = src[tb][j];
        if (NCF_WS8_ABLATE == 1) return;
        if (NCF_WS8_ABLATE == 2) gp = reinterpret_cast<const unsigned char*>(a.tabA) + ((gp - reinterpret_cast<const unsigned char*>(a.tabA)) & 0xFFFF0);
        if (NCF_WS8_ABLATE == 3) gp = reinterpret_cast<const unsigned char*>(a.tabA) + ((gp - reinterpret_cast<const unsigned char*>(a.tabA)) & 0x3FFFFF0);
        dma16(gp, lds0 + slot * L::UB + tb * TSTRIDE + (g * NJ + j) * 1024);
    };
    auto issue_rows = [&](const RowSrc& src, int slot, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < NCU; ++i)
            if (i >= i0 && i < i1) issue_row(src, slot, i);
    };
    constexpr int NA = NCF_WS8_STAGE ? 0 : (NCF_WS8_NA < NCU ? NCF_WS8_NA : NCU - 1);   // row pieces per unit and pair group issued by the A wave (the rest + ids: B)
    // a unit's 16 ids per B wave (2 tables x 8 pairs) arrive by one dword LDS-DMA; pairs past the end repeat the last pair
    const int ids_dw = lane & 15, ids_tb = (lane >> 4) & 1;
    auto ids_dma = [&](int k, int islot) {
        const int p0 = unit_pair0(k) + 8 * g + (ids_dw >> 1);
        const int p = p0 < Bp ? p0 : Bp - 1;
        const int* gp = reinterpret_cast<const int*>((ids_tb ? idxB : idxA) + p) + (ids_dw & 1);
        dma4(gp, lds0 + L::OFF_IDS + (g * L::IDS_SLOTS + islot) * 256);
    };
    auto locate = [&](RowSrc& src, int k, int islot) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned char* q = lds + L::OFF_IDS + (g * L::IDS_SLOTS + islot) * 256 + (PP * j + dpp) * 8;
            const int64_t ia = *reinterpret_cast<const int64_t*>(q);
            const int64_t ib = *reinterpret_cast<const int64_t*>(q + 64);
            resolve(src, j, unit_pair0(k) + 8 * g + PP * j + dpp, ia, ib);
        }
    };
    if (w < 4) {
        // ================================================= A: layer 1 =================================================
        if (NCF_WS8_PRIO == 2) __builtin_amdgcn_s_setprio(1);
        if constexpr (M16) {
            // ---- 16x16x32 form: 4 row tiles x 2 column tiles of 16 pairs; a unit's 2 KS1 X fragments feed 4 MFMAs each ----
            constexpr int KS1 = K0 / 32, KS2 = N1 / 32, NF = 2 * KS1;
            static_assert(NF % RING == 0, "the fragment ring keeps its phase from unit to unit");
            const int p16 = lane & 15, kg = lane >> 4;
            u32x4 wa[4][KS1];
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt)
                    wa[rt][ks] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp1m) + ((size_t)(ks * (N1 / 16) + 4 * g + rt) * 64 + lane) * 16);
            const float bl = a.bl[0];
            // reader offsets: lane (p16, kg) takes chunk 4 ks + kg of pair 16 ct + p16; one offset per column tile and ks % (RC / 4)
            constexpr int SG = RC / 4;
            unsigned rdx[2][SG];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int j = 0; j < SG; ++j) {
                    const int mm = 16 * ct + p16;
                    rdx[ct][j] = (mm / PP) * 1024 + (mm % PP) * RC * 16 + (((4 * j + kg) ^ xkey(mm)) & (RC - 1)) * 16;
                }
            auto xp = [&](unsigned xb, int j) {              // fragment j of a unit: column tile j / KS1, k-step j % KS1
                const int ct = j / KS1, ks = j % KS1;
                return reinterpret_cast<const u32x4*>(lds + (rdx[ct][ks % SG] + xb) + (ks / SG) * TSTRIDE);
            };
            unsigned char* const hwr = lds + L::OFF_H1 + (2 * g) * 1024 + lane * 16;   // fragments (ct, ks2 = 2g + e) at (ct KS2 + 2g + e) KiB
            const unsigned char* const b1p = lds + L::OFF_B1 + (64 * g + 4 * kg) * 4;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) asm volatile("" ::"v"(wa[0][ks]), "v"(wa[1][ks]), "v"(wa[2][ks]), "v"(wa[3][ks]));
            wg_barrier();                                    // biases in LDS; the B waves have the rows of units 0 and 1 landed

            f32x4 acc[2][4][2];                              // [unit parity][row tile][column tile]
            u32x4 fr[RING];
            auto bias_init = [&](f32x4 (&ac)[4][2], int rt) {
                const f32x4 bb = *reinterpret_cast<const f32x4*>(b1p + 64 * rt);
                ac[rt][0] = bb; ac[rt][1] = bb;
            };
#pragma unroll
            for (int pk = 0; pk < 2; ++pk)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) bias_init(acc[pk], rt);
#pragma unroll
            for (int j = 0; j < AHEAD; ++j) fr[j] = *xp(0u, j);
            auto store_out = [&](int k) {                    // unit k's outputs (wave 0; red[k&1] was published in phase k+3)
                if (w == 0 && lane < 32) {
                    const float* r = red + ((k & 1) * 32 + lane) * 16;
                    f32x4 t = *reinterpret_cast<const f32x4*>(r);
#pragma unroll
                    for (int i = 1; i < 4; ++i) {
                        const f32x4 u = *reinterpret_cast<const f32x4*>(r + 4 * i);
                        t[0] += u[0]; t[1] += u[1]; t[2] += u[2]; t[3] += u[3];
                    }
                    const int p = unit_pair0(k) + lane;
                    if (p < Bp) out[p] = ((t[0] + t[1]) + (t[2] + t[3])) + bl;
                }
            };
            // ReLU + bf16 of two 16-neuron tiles -> one layer-2 B fragment (column tile ct, k-step 2g + e) of H1 half `half`
            auto pack_frag = [&](const f32x4 (&ac)[4][2], int half, int f) {
                const int ct = f >> 1, e = f & 1;
                *reinterpret_cast<bf16x8_t*>(hwr + half * L::H1_HALF + (ct * KS2 + e) * 1024) = pack_relu4x2_int(ac[2 * e][ct], ac[2 * e + 1][ct]);
            };
            int slot = 0, slotA = D % NU, islotA = D % L::IDS_SLOTS;
            auto phase = [&](int k, auto pk_tag) {
                constexpr int PK = decltype(pk_tag)::value;
                const int nslot = slot == NU - 1 ? 0 : slot + 1;
                const unsigned xb = slot * L::UB, xbn = nslot * L::UB;
                W8_STAMP(0);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    const int jn = j + AHEAD;
                    fr[jn % RING] = jn < NF ? *xp(xb, jn) : *xp(xbn, jn - NF);
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt)
                        acc[PK][rt][j / KS1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wa[rt][j % KS1]), as_bf16x8(fr[j % RING]), acc[PK][rt][j / KS1], 0, 0, 0);
                    if (at(j, NF, 1, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 0);
                    if (at(j, NF, 2, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 1);
                    if (at(j, NF, 3, 8)) { if (k >= 4) store_out(k - 4); }
                    if (at(j, NF, 4, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 2);
                    if (at(j, NF, 5, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 3);
                    if (at(j, NF, 6, 8)) { bias_init(acc[PK ^ 1], 0); bias_init(acc[PK ^ 1], 1); }
                    if (at(j, NF, 7, 8)) { bias_init(acc[PK ^ 1], 2); bias_init(acc[PK ^ 1], 3); }
                }
                W8_STAMP(1);
                if (NA > 0) {
                    const bool fetch = k + D < n;
                    if (fetch) {
                        RowSrc src;
                        locate(src, k + D, islotA);
                        issue_rows(src, slotA, NCU - NA, NCU);
                    }
                    W8_STAMP(4);
                    if (!fetch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (w != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NA) : "memory");
                    else if (k >= D + 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (NA + 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    slotA = slotA == NU - 1 ? 0 : slotA + 1;
                    islotA = islotA == L::IDS_SLOTS - 1 ? 0 : islotA + 1;
                }
                W8_STAMP(2);
                wg_barrier();
                W8_STAMP(3);
                slot = nslot;
            };
            int k = 0;
            for (; k + 1 < n; k += 2) {
                phase(k, std::integral_constant<int, 0>{});
                phase(k + 1, std::integral_constant<int, 1>{});
            }
            if (k < n) { phase(k, std::integral_constant<int, 0>{}); ++k; }
            if (n & 1) {
#pragma unroll
                for (int f = 0; f < 4; ++f) pack_frag(acc[0], 0, f);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) pack_frag(acc[1], 1, f);
            }
#pragma unroll
            for (int dph = 0; dph < 4; ++dph) {
                if (n + dph >= 4) store_out(n + dph - 4);
                if (dph < 3) wg_barrier();
            }
        } else {
            u32x4 wa1[2][Q1];
#pragma unroll
            for (int s = 0; s < Q1; ++s)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    wa1[nt][s] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp1) + ((size_t)(s * (N1 / 32) + 2 * g + nt) * 64 + lane) * 16);
            const float bl = a.bl[0];
            // reader offsets of the X image (see the gather helpers): the XOR does not commute with +, one offset per s % (RC/2)
            constexpr int SIG = RC / 2;
            const int key = xkey(m);
            unsigned rdx[SIG];
#pragma unroll
            for (int j = 0; j < SIG; ++j) rdx[j] = (m / PP) * 1024 + (m % PP) * RC * 16 + (((2 * j + h) ^ key) & (RC - 1)) * 16;
            unsigned char* const hwr = lds + L::OFF_H1 + (4 * g) * 1024 + lane * 16;   // this wave's four H1 fragments: q = 2 (2g + nt) + s2
            auto xp = [&](unsigned xb, int s) { return reinterpret_cast<const u32x4*>(lds + (rdx[s % SIG] + xb) + (s / SIG) * TSTRIDE); };
#pragma unroll
            for (int s = 0; s < Q1; ++s) asm volatile("" ::"v"(wa1[0][s]), "v"(wa1[1][s]));
            wg_barrier();                                        // biases in LDS; the B waves have the rows of units 0 and 1 landed

            f32x16 acc[2][2];                                    // [unit parity][row tile]
            u32x4 fr[RING];
#pragma unroll
            for (int pk = 0; pk < 2; ++pk)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[pk][nt] = bias_tile(L::OFF_B1, 64 * g + 32 * nt);
#pragma unroll
            for (int j = 0; j < AHEAD; ++j) fr[j] = *xp(0u, j);
            auto store_out = [&](int k) {                        // unit k's outputs (wave 0; red[k&1] was published in phase k+3)
                if (w == 0 && lane < 32) {
                    const float* r = red + ((k & 1) * 32 + lane) * 8;
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>(r);
                    const f32x4 r1 = *reinterpret_cast<const f32x4*>(r + 4);
                    const int p = unit_pair0(k) + lane;
                    const float v = ((r0[0] + r0[1]) + (r0[2] + r0[3])) + ((r1[0] + r1[1]) + (r1[2] + r1[3])) + bl;
                    if (p < Bp) out[p] = v;
                }
            };
            // ReLU + bf16 of half an accumulator tile -> one ready-made layer-2 B fragment of H1 half `half`
            auto pack_frag = [&](const f32x16 (&ac)[2], int half, int f) {
                if (NCF_WS8_ABLATE == 8) return;
                const int nt = f >> 1, s2 = f & 1;
                if (NCF_WS8_ABLATE == 9) { const bf16x8_t v = pack_relu8_int(ac[nt], 8 * s2); asm volatile("" ::"v"(v)); return; }
                if (NCF_WS8_ABLATE == 10) {
                    const u32x4 raw = {__float_as_uint(ac[nt][8 * s2]), __float_as_uint(ac[nt][8 * s2 + 1]), __float_as_uint(ac[nt][8 * s2 + 2]), __float_as_uint(ac[nt][8 * s2 + 3])};
                    *reinterpret_cast<u32x4*>(hwr + half * L::H1_HALF + (2 * nt + s2) * 1024) = raw;
                    return;
                }
                *reinterpret_cast<bf16x8_t*>(hwr + half * L::H1_HALF + (2 * nt + s2) * 1024) = pack_relu8_int(ac[nt], 8 * s2);
            };
            int slot = 0, slotA = D % NU, islotA = D % L::IDS_SLOTS;
            auto phase = [&](int k, auto pk_tag) {
                constexpr int PK = decltype(pk_tag)::value;
                const int nslot = slot == NU - 1 ? 0 : slot + 1;
                const unsigned xb = slot * L::UB, xbn = nslot * L::UB;
                W8_STAMP(0);
#pragma unroll
                for (int s = 0; s < Q1; ++s) {
                    const int j = s + AHEAD;
                    fr[j % RING] = j < Q1 ? *xp(xb, j) : *xp(xbn, j - Q1);
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        if (NCF_WS8_ABLATE != 11) acc[PK][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa1[nt][s]), as_bf16x8(fr[s % RING]), acc[PK][nt], 0, 0, 0);
                        else asm volatile("" ::"v"(fr[s % RING]));
                    // fillers: the previous unit's accumulators -> H1, then their bias for the next unit; outputs of unit k-4
                    if (at(s, Q1, 1, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 0);
                    if (at(s, Q1, 2, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 1);
                    if (at(s, Q1, 3, 8)) { if (k >= 4) store_out(k - 4); }
                    if (at(s, Q1, 4, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 2);
                    if (at(s, Q1, 5, 8)) pack_frag(acc[PK ^ 1], PK ^ 1, 3);
                    if (at(s, Q1, 6, 8)) acc[PK ^ 1][0] = bias_tile(L::OFF_B1, 64 * g);
                    if (at(s, Q1, 7, 8)) acc[PK ^ 1][1] = bias_tile(L::OFF_B1, 64 * g + 32);
                }
                // A's share of the gather, in the time it would otherwise spend waiting for the B waves at the barrier: the row
                // pieces cu >= NCU - NA of unit k+D (an LDS-DMA of random rows holds its wave ~150-200 cycles)
                if (NA > 0) {
                    const bool fetch = k + D < n;
                    if (fetch) {
                        RowSrc src;
                        locate(src, k + D, islotA);
                        issue_rows(src, slotA, NCU - NA, NCU);
                    }
                    // rows(k+2) have landed once only the DMAs of the last D-2 phases are in flight (wave 0 also has one store per
                    // phase in its queue from phase 4 on)
                    if (!fetch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    else if (w != 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * NA) : "memory");
                    else if (k >= D + 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (NA + 1)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    slotA = slotA == NU - 1 ? 0 : slotA + 1;
                    islotA = islotA == L::IDS_SLOTS - 1 ? 0 : islotA + 1;
                }
                wg_barrier();
                slot = nslot;
            };
            int k = 0;
            for (; k + 1 < n; k += 2) {
                phase(k, std::integral_constant<int, 0>{});
                phase(k + 1, std::integral_constant<int, 1>{});
            }
            if (k < n) { phase(k, std::integral_constant<int, 0>{}); ++k; }
            // drain: phase n packs the last unit; phases n .. n+3 store the last four units' outputs
            if (n & 1) {
#pragma unroll
                for (int f = 0; f < 4; ++f) pack_frag(acc[0], 0, f);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) pack_frag(acc[1], 1, f);
            }
#pragma unroll
            for (int dph = 0; dph < 4; ++dph) {
                if (n + dph >= 4) store_out(n + dph - 4);
                if (dph < 3) wg_barrier();
            }
        }
    } else {
        // ============================================ B: layer 2, last layer, I/O ============================================
        if (NCF_WS8_PRIO == 1) __builtin_amdgcn_s_setprio(1);
        // ---- staged gather (NCF_WS8_STAGE): ids and rows by plain loads, two phases ahead in registers, then ds_write_b128 ----
        struct Ids { int64_t ia[NJ], ib[NJ]; };
        Ids idsbuf[2];                                       // [parity]: ids of unit k+4 while phase k runs (loaded in phase k-2)
        u32x4 rowbuf[2][NCU];                                // [parity]: rows of unit k+2 (loaded in phase k-2)
        auto load_ids = [&](Ids& d, int k) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int p0 = unit_pair0(k) + 8 * g + PP * j + dpp;
                const int p = p0 < Bp ? p0 : Bp - 1;
                d.ia[j] = idxA[p];
                d.ib[j] = idxB[p];
            }
        };
        auto load_rows = [&](u32x4 (&rb)[NCU], const Ids& d, int k) {
            RowSrc src;
#pragma unroll
            for (int j = 0; j < NJ; ++j) resolve(src, j, unit_pair0(k) + 8 * g + PP * j + dpp, d.ia[j], d.ib[j]);
#pragma unroll
            for (int i = 0; i < NCU; ++i) rb[i] = ldg16(src[i / NJ][i % NJ]);
        };
        auto store_rows = [&](const u32x4 (&rb)[NCU], int slot) {
#pragma unroll
            for (int i = 0; i < NCU; ++i)
                *reinterpret_cast<u32x4*>(lds + slot * L::UB + (i / NJ) * TSTRIDE + (g * NJ + i % NJ) * 1024 + lane * 16) = rb[i];
        };
        RowSrc psrc[D];                                      // prologue only
        u32x4 wb[2][N1 / 32];                                // layer-2 A fragments of the 16x16x32 form (M16)
        auto load_wb = [&]() {                               // issued BEHIND the id loads, so that their latencies overlap
            if constexpr (M16) {
#pragma unroll
                for (int ks = 0; ks < N1 / 32; ++ks)
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        wb[rt][ks] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp2m) + ((size_t)(ks * (N2 / 16) + 2 * g + rt) * 64 + lane) * 16);
            }
        };
        if (NCF_WS8_STAGE) {
            Ids i0, i1;
            u32x4 r0[NCU], r1[NCU];
            load_ids(i0, 0); load_ids(i1, 1); load_ids(idsbuf[0], 2); load_ids(idsbuf[1], 3);
            load_rows(r0, i0, 0); load_rows(r1, i1, 1);
            load_rows(rowbuf[0], idsbuf[0], 2); load_rows(rowbuf[1], idsbuf[1], 3);
            load_ids(idsbuf[0], 4); load_ids(idsbuf[1], 5);
            store_rows(r0, 0); store_rows(r1, 1);
        } else {
            // prologue, first part: the ids of units 0..D-1 by direct loads, the rows of units 0 and 1 and the ids of units D..2D-1
            // (phase 0 reads ids(D) at once) by DMA; then the weights.  Only once all of that has arrived (one drain of the queue)
            // the rows of units 2..D-1 are requested: they land during the first phases (per unit they are at least as many queue
            // entries as a steady-state phase issues, so the counted waits of the loop are at worst early).
            {
                int64_t pid[D][NJ][2];
    #pragma unroll
                for (int t = 0; t < D; ++t)
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int p0 = unit_pair0(t) + 8 * g + PP * j + dpp;
                        const int p = p0 < Bp ? p0 : Bp - 1;
                        pid[t][j][0] = idxA[p];
                        pid[t][j][1] = idxB[p];
                    }
                load_wb();
    #pragma unroll
                for (int t = 0; t < D; ++t) {
    #pragma unroll
                    for (int j = 0; j < NJ; ++j) resolve(psrc[t], j, unit_pair0(t) + 8 * g + PP * j + dpp, pid[t][j][0], pid[t][j][1]);
                    if (t < 2 && t < n) issue_rows(psrc[t], t, 0, NCU);
                }
    #pragma unroll
                for (int t = 0; t < D; ++t)                  // phase 0 reads ids(D) at once: they belong to the drained part
                    if (D + t < n) ids_dma(D + t, (D + t) % L::IDS_SLOTS);
            }
        }
        auto prologue_rest = [&]() {
            if (NCF_WS8_STAGE) return;
    #pragma unroll
            for (int t = 2; t < D; ++t)
                if (t < n) issue_rows(psrc[t], t, 0, NCU);
        };
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (M16) {
            // ---- 16x16x32 form: 2 row tiles x 2 column tiles; a unit's 2 KS2 H1 fragments feed 2 MFMAs each ----
            constexpr int KS2 = N1 / 32, NF = 2 * KS2;
            const int p16 = lane & 15, kg = lane >> 4;
            if (NCF_WS8_STAGE) load_wb();
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) asm volatile("" ::"v"(wb[0][ks]), "v"(wb[1][ks]));
            if (!NCF_WS8_STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prologue_rest();
            wg_barrier();

            const unsigned char* const hbase = lds + L::OFF_H1 + lane * 16;
            const unsigned char* const b2p = lds + L::OFF_B2 + (32 * g + 4 * kg) * 4;
            f32x4 wlr[2];                                    // the last layer's weights of this lane's 8 neurons (32g + 16 rt + 4 kg + r)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) wlr[rt] = *reinterpret_cast<const f32x4*>(lds + L::OFF_WL + (32 * g + 16 * rt + 4 * kg) * 4);
            f32x4 acc2[2][2][2];                             // [unit parity][row tile][column tile]
            auto bias_init = [&](f32x4 (&ac)[2][2]) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const f32x4 bb = *reinterpret_cast<const f32x4*>(b2p + 64 * rt);
                    ac[rt][0] = bb; ac[rt][1] = bb;
                }
            };
            bias_init(acc2[0]); bias_init(acc2[1]);
            auto l2_stream = [&](f32x4 (&ac)[2][2], int half, auto&& fill) {
                constexpr int RB = NCF_WS8_RINGB, AB = RB - 1;
                u32x4 fr[RB];
#pragma unroll
                for (int j = 0; j < AB; ++j) fr[j] = *reinterpret_cast<const u32x4*>(hbase + half * L::H1_HALF + j * 1024);
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    if (j + AB < NF) fr[(j + AB) % RB] = *reinterpret_cast<const u32x4*>(hbase + half * L::H1_HALF + (j + AB) * 1024);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
                        ac[rt][j / KS2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(wb[rt][j % KS2]), as_bf16x8(fr[j % RB]), ac[rt][j / KS2], 0, 0, 0);
                    fill(j);
                }
            };
            auto dot_ct = [&](const f32x4 (&ac)[2][2], int ct) {
                float part = 0.f;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) part = fmaf(wlr[rt][r], relu1(ac[rt][ct][r]), part);
                return part;
            };
            auto publish = [&](int par, int ct, float part) { red[((par * 32 + 16 * ct + p16) * 4 + g) * 4 + kg] = part; };
            int slot = (NCF_WS8_STAGE ? 2 : D) % NU, islot = D % L::IDS_SLOTS, islot2 = (2 * D) % L::IDS_SLOTS;
            auto phase = [&](int k, auto pk_tag) {
                constexpr int PK = decltype(pk_tag)::value;  // parity of k: layer 2 of unit k-2 -> acc2[PK], dot of unit k-3 from acc2[PK^1]
                const bool fetch = k + D < n && NCF_WS8_ABLATE != 6;
                RowSrc src;
                W8_STAMP(0);
                l2_stream(acc2[PK], PK, [&](int q) {
                    if (q == 6) W8_STAMP(4);
                    if (NCF_WS8_STAGE) {
                        if (q == 0) {
                            store_rows(rowbuf[PK], slot);
                            const Ids cur = idsbuf[PK];
                            load_ids(idsbuf[PK], k + 6);
                            load_rows(rowbuf[PK], cur, k + 4);
                        }
                    } else if (NCF_WS8_SPREAD) {
                        if (q == 0 && fetch) { locate(src, k + D, islot); ids_dma(k + 2 * D, islot2); }
                        if (q >= 1 && (q - 1) % NCF_WS8_SPREAD == 0 && (q - 1) / NCF_WS8_SPREAD < NCU - NA && fetch) issue_row(src, slot, (q - 1) / NCF_WS8_SPREAD);
                    } else if (q == 0 && fetch) {
                        locate(src, k + D, islot);
                        ids_dma(k + 2 * D, islot2);
                        issue_rows(src, slot, 0, NCU - NA);
                    }
                    if (q == 5) publish(PK ^ 1, 0, dot_ct(acc2[PK ^ 1], 0));
                    if (q == 11) publish(PK ^ 1, 1, dot_ct(acc2[PK ^ 1], 1));
                    if (q == 14) bias_init(acc2[PK ^ 1]);
                });
                W8_STAMP(1);
                if (NCF_WS8_STAGE) { }
                else if (fetch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (NCU - NA + 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                W8_STAMP(2);
                wg_barrier();
                W8_STAMP(3);
                slot = slot == NU - 1 ? 0 : slot + 1;
                islot = islot == L::IDS_SLOTS - 1 ? 0 : islot + 1;
                islot2 = islot2 == L::IDS_SLOTS - 1 ? 0 : islot2 + 1;
            };
            int k = 0;
            for (; k + 1 < n; k += 2) {
                phase(k, std::integral_constant<int, 0>{});
                phase(k + 1, std::integral_constant<int, 1>{});
            }
            if (k < n) { phase(k, std::integral_constant<int, 0>{}); ++k; }
            auto drain = [&](auto pk_tag, bool l2) {
                constexpr int PK = decltype(pk_tag)::value;
                if (l2) l2_stream(acc2[PK], PK, [&](int) {});
                publish(PK ^ 1, 0, dot_ct(acc2[PK ^ 1], 0));
                publish(PK ^ 1, 1, dot_ct(acc2[PK ^ 1], 1));
                bias_init(acc2[PK ^ 1]);
                wg_barrier();
            };
            if (n & 1) {
                drain(std::integral_constant<int, 1>{}, true);
                drain(std::integral_constant<int, 0>{}, true);
                drain(std::integral_constant<int, 1>{}, false);
            } else {
                drain(std::integral_constant<int, 0>{}, true);
                drain(std::integral_constant<int, 1>{}, true);
                drain(std::integral_constant<int, 0>{}, false);
            }
        } else {
            u32x4 wa2[Q2];
#pragma unroll
            for (int q = 0; q < Q2; ++q)
                wa2[q] = ldg16(reinterpret_cast<const unsigned char*>(a.Wp2) + ((size_t)(q * (N2 / 32) + g) * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < Q2; ++q) asm volatile("" ::"v"(wa2[q]));
            if (!NCF_WS8_STAGE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            prologue_rest();
            wg_barrier();

            const unsigned char* const hbase = lds + L::OFF_H1 + lane * 16;
            const unsigned char* const wlp = lds + L::OFF_WL + (32 * g + 4 * h) * 4;
            f32x16 acc2[2];
            acc2[0] = bias_tile(L::OFF_B2, 32 * g);
            acc2[1] = acc2[0];
            auto l2_stream = [&](f32x16& acc, int half, auto&& fill) {
                constexpr int RB = NCF_WS8_RINGB, AB = RB - 1;
                u32x4 fr[RB];
#pragma unroll
                for (int j = 0; j < AB; ++j) fr[j] = *reinterpret_cast<const u32x4*>(hbase + half * L::H1_HALF + j * 1024);
#pragma unroll
                for (int q = 0; q < Q2; ++q) {
                    if (q + AB < Q2) fr[(q + AB) % RB] = *reinterpret_cast<const u32x4*>(hbase + half * L::H1_HALF + (q + AB) * 1024);
                    if (NCF_WS8_ABLATE != 7 && NCF_WS8_ABLATE != 11) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wa2[q]), as_bf16x8(fr[q % RB]), acc, 0, 0, 0);
                    else asm volatile("" ::"v"(fr[q % RB]));
                    fill(q);
                }
            };
            auto dot_quarter = [&](const f32x16& acc, int gq, float part) {
                const f32x4 ww = *reinterpret_cast<const f32x4*>(wlp + 32 * gq);
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) part = fmaf(ww[jj], relu1(acc[4 * gq + jj]), part);
                return part;
            };
            auto publish = [&](int par, float part) { red[((par * 32 + m) * 4 + g) * 2 + h] = part; };
            int slot = (NCF_WS8_STAGE ? 2 : D) % NU, islot = D % L::IDS_SLOTS, islot2 = (2 * D) % L::IDS_SLOTS;   // of units k+D (rows, ids read) and k+2D (ids written)
            auto phase = [&](int k, auto pk_tag) {
                constexpr int PK = decltype(pk_tag)::value;      // parity of k: layer 2 of unit k-2 -> acc2[PK], dot of unit k-3 from acc2[PK^1]
                const bool fetch = k + D < n && NCF_WS8_ABLATE != 6;
                float part = 0.f;
                W8_STAMP(0);
                RowSrc src;
                l2_stream(acc2[PK], PK, [&](int q) {
                    if (NCF_WS8_STAGE) {
                        if (q == 0) {
                            store_rows(rowbuf[PK], slot);        // unit k+2 (slot of unit k+D with D = 3 ... one behind: see below)
                            const Ids cur = idsbuf[PK];          // ids of unit k+4
                            load_ids(idsbuf[PK], k + 6);
                            load_rows(rowbuf[PK], cur, k + 4);
                        }
                    } else
                    // the DMAs are spread over the stream: an LDS-DMA of gathered rows holds the wave ~100-200 cycles, during which the
                    // MFMAs it issued just before keep the matrix pipe busy
                    if (NCF_WS8_SPREAD) {
                        if (q == 0 && fetch) { locate(src, k + D, islot); ids_dma(k + 2 * D, islot2); }
                        if (q >= 1 && (q - 1) % NCF_WS8_SPREAD == 0 && (q - 1) / NCF_WS8_SPREAD < NCU - NA && fetch) issue_row(src, slot, (q - 1) / NCF_WS8_SPREAD);
                    } else if (q == 0 && fetch) {
                        locate(src, k + D, islot);
                        ids_dma(k + 2 * D, islot2);
                        issue_rows(src, slot, 0, NCU - NA);
                    }
                    if (q == 0) W8_STAMP(4);
                    if (q == 8) W8_STAMP(5);
                    if (q % 4 == 1 && NCF_WS8_ABLATE != 5) part = dot_quarter(acc2[PK ^ 1], q / 4, part);
                    if (q == 14 && NCF_WS8_ABLATE != 5) publish(PK ^ 1, part);
                    if (q == 15) acc2[PK ^ 1] = bias_tile(L::OFF_B2, 32 * g);
                });
                W8_STAMP(1);
                // rows(k+2) have landed once only the DMAs of the last D-2 phases are in flight
                if (NCF_WS8_STAGE) { }
                else if (fetch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 2) * (NCU - NA + 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                W8_STAMP(2);
                wg_barrier();
                W8_STAMP(3);
                slot = slot == NU - 1 ? 0 : slot + 1;
                islot = islot == L::IDS_SLOTS - 1 ? 0 : islot + 1;
                islot2 = islot2 == L::IDS_SLOTS - 1 ? 0 : islot2 + 1;
            };
            int k = 0;
            for (; k + 1 < n; k += 2) {
                phase(k, std::integral_constant<int, 0>{});
                phase(k + 1, std::integral_constant<int, 1>{});
            }
            if (k < n) { phase(k, std::integral_constant<int, 0>{}); ++k; }
            // drain: phase n: layer 2 of unit n-2, dot of unit n-3; phase n+1: layer 2 of n-1, dot of n-2; phase n+2: dot of n-1
            auto drain = [&](auto pk_tag, bool l2) {
                constexpr int PK = decltype(pk_tag)::value;
                if (l2) l2_stream(acc2[PK], PK, [&](int) {});
                float part = 0.f;
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) part = dot_quarter(acc2[PK ^ 1], gq, part);
                publish(PK ^ 1, part);
                acc2[PK ^ 1] = bias_tile(L::OFF_B2, 32 * g);
                wg_barrier();
            };
            if (n & 1) {
                drain(std::integral_constant<int, 1>{}, true);
                drain(std::integral_constant<int, 0>{}, true);
                drain(std::integral_constant<int, 1>{}, false);
            } else {
                drain(std::integral_constant<int, 0>{}, true);
                drain(std::integral_constant<int, 1>{}, true);
                drain(std::integral_constant<int, 0>{}, false);
            }
        }
        if (oob_seen && a.oob) *a.oob = 1;
    }
#if NCF_BF16_STAMP
    if (a.dbg) {
        wg_barrier();
        a.dbg[(int64_t)blockIdx.x * 512 + threadIdx.x] = reinterpret_cast<unsigned long long*>(lds + L::OFF_STAMP)[threadIdx.x];
    }
#endif
}

bool ws8_shape_ok(int K0, int N1, int N2) { return (K0 == 256 || K0 == 128) && N1 == 256 && N2 == 128; }   // and EA == EB == K0 / 2 (checked by the caller)

void launch_ws8_bf16(int K0, const Bf16Args& a, const unsigned char* zeros, hipStream_t s) {
    const int nunits = (int)((a.B + 31) / 32);
    const int grid = nunits < num_cus() ? nunits : num_cus();
    constexpr bool M16 = NCF_WS8_M16 != 0;
    if (K0 == 256)
        hipLaunchKernelGGL((score_ws8_bf16_kernel<256, M16>), dim3((unsigned)grid), dim3(512), 0, s, a, a.idxA, a.idxB, a.out, zeros, nunits);
    else
        hipLaunchKernelGGL((score_ws8_bf16_kernel<128, M16>), dim3((unsigned)grid), dim3(512), 0, s, a, a.idxA, a.idxB, a.out, zeros, nunits);
}

}  // namespace ncf
