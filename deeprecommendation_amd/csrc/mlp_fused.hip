// K1+K2 fused — gather -> concat -> MLP -> (B,1) in one launch, fp32, gfx950.
//
// Design (one wave = one tile of 32 pairs, no LDS, no inter-wave communication):
//   Every layer is computed TRANSPOSED:  H^T[n][m] = W[n][:] . X^T[:][m]  with the pair index m on the MFMA
//   column (= lane & 31) and the neuron index n on the accumulator rows (registers).  With
//   v_mfma_f32_32x32x2_f32 the C/D register r of lane-half h holds row (r&3) + 8(r>>2) + 4h, and the B operand
//   of the *next* layer wants B[k = step + h][m] on the same lane — so register r of the previous layer's
//   accumulator IS the next layer's B operand for k-step r (after bias + ReLU).  Activations never leave
//   registers between layers; only weights stream in.
//   Weights are pre-packed (ncf_mlp_pack) as Wp[q][nt][lane][4] = W[32nt + (lane&31)][8q + 4(lane>>5) + 0..3],
//   so one 16-byte load per lane feeds 4 consecutive MFMAs and a wave-instruction reads 1 KiB contiguous.
//   The layer-1 B operand is the gathered embedding row itself: lane (m, h) reads the 16-byte chunk 2q + h of
//   pair m's concatenated row straight from the tables (each 128-B line is consumed by 4 consecutive loads).
//   The K order inside an 8-wide group is permuted identically for A and B, which only changes the fp32
//   summation order (exact FMA chain, one rounding per product).
// Bound: fp32 MFMA (157.3 TFLOP/s).  FLOP per pair = 2 * (K0*N1 + N1*N2 + N2)  (131 328 at 128-256-128-1).
#include "ncf_common.h"

#ifndef NCF_X_DEPTH
#define NCF_X_DEPTH 4       // measured (interleaved A/B, cfg 2): 2 -> 72.4 us, 3 -> 70.0 us, 4 -> 69.5 us
#endif
#ifndef NCF_PAIR
#define NCF_PAIR 0          // 1: MFMAs alternate between two accumulator tiles; measured no gain (70.2 vs 69.7 us): dependent 32x32x2 MFMAs issue back to back at full rate
#endif
#ifndef NCF_ABLATE_LOADS
#define NCF_ABLATE_LOADS 0  // diagnostic: skip the in-loop weight / row loads (wrong results, pure MFMA stream timing)
#endif
#ifndef NCF_WG_WAVES
#define NCF_WG_WAVES 4      // waves per workgroup (4 or 8)
#endif
#ifndef NCF_MIN_WAVES
#define NCF_MIN_WAVES 2     // launch_bounds second argument: waves per SIMD the register budget must allow
#endif
#ifndef NCF_STAGGER
#define NCF_STAGGER 0       // s_sleep units (64 clk) the second half of the workgroup's waves start late by
#endif

#ifndef NCF_STAMP
#define NCF_STAMP 0         // diagnostic builds only (tools/ab_fused.py): per-wave s_memtime / s_memrealtime stamps
#endif

namespace ncf {

#if NCF_STAMP
static unsigned long long* g_dbg = nullptr;  // dev builds only; never present in the shipped library
#endif

struct FusedArgs {
    const float* tabA; int64_t rowsA; int64_t ldA;
    const float* tabB; int64_t rowsB; int64_t ldB;
    const int64_t* idxA; const int64_t* idxB;
    int64_t B; int EA;
    const float* Wp1; const float* b1;
    const float* Wp2; const float* b2;
    const float* wl; const float* bl;   // last (1-wide) layer: weights [Nlast], bias [1]
    float* out; int32_t* oob;
#if NCF_STAMP
    unsigned long long* dbg;
#endif
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int K0, int N1, int N2>
__global__ __launch_bounds__(NCF_WG_WAVES * 64, NCF_MIN_WAVES) void score_fused_f32_kernel(FusedArgs a) {
    constexpr int NT1 = N1 / 32, Q1 = K0 / 8;
    constexpr int NT2 = N2 / 32, Q2 = N1 / 8;
    const int lane = threadIdx.x & 63;
    const int m = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * NCF_WG_WAVES + (threadIdx.x >> 6);
    if (tile * 32 >= a.B) return;  // whole wave exits together
#if NCF_STAMP
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (NCF_STAGGER > 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) >= NCF_WG_WAVES * 32) __builtin_amdgcn_s_sleep(NCF_STAGGER);
    const int64_t p = tile * 32 + m;
    const int64_t pc = p < a.B ? p : a.B - 1;

    // ---- gather setup: row pointers of this lane's pair ----
    const int64_t ia = a.idxA ? a.idxA[pc] : pc;
    const bool okA = (ia >= 0) & (ia < a.rowsA);
    const float* rowA = a.tabA + (okA ? ia : 0) * a.ldA + 4 * h;
    const int qa = a.EA / 8;  // groups served by table A
    const float* rowB = rowA;
    bool okB = true;
    if (qa < Q1) {
        const int64_t ib = a.idxB ? a.idxB[pc] : pc;
        okB = (ib >= 0) & (ib < a.rowsB);
        rowB = a.tabB + (okB ? ib : 0) * a.ldB + 4 * h;
    }
    if (!(okA & okB) && a.oob) *a.oob = 1;
    const float zA = okA ? 1.f : 0.f, zB = okB ? 1.f : 0.f;  // out-of-range rows read as zeros
    // Touch every 128-byte line of this pair's two rows NOW (one dword each, lane half h takes lines h, h+2, ...):
    // the HBM misses of all lines overlap each other and the bias / first-weight loads, instead of surfacing one
    // by one at every 4th k-step of layer 1 (each step only prefetches one step ahead).
#ifndef NCF_TOUCH
#define NCF_TOUCH 0   // measured: touching all lines at kernel start is SLOWER (81 vs 75 us): 2048 waves issue the whole
#endif                // batch's 34 MB of HBM misses at once and every wave's first weight wait queues behind them
    if (NCF_TOUCH) {
        float touch = 0.f;
        for (int l = h; l * 32 < a.EA; l += 2) touch += rowA[l * 32 - 4 * h];
        for (int l = h; l * 32 < K0 - a.EA; l += 2) touch += rowB[l * 32 - 4 * h];
        asm volatile("" ::"v"(touch));
    }

    // ---- layer 1: acc1 = b1 (broadcast over pairs) + W1 . X^T ----
    f32x16 acc1[NT1];
#pragma unroll
    for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bb = ldg4(a.b1 + 32 * nt + 8 * g + 4 * h);
            acc1[nt][4 * g + 0] = bb[0]; acc1[nt][4 * g + 1] = bb[1];
            acc1[nt][4 * g + 2] = bb[2]; acc1[nt][4 * g + 3] = bb[3];
        }
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp1) + lane;  // + (q*NT1 + nt)*64
        // XD = depth of the gathered-row prefetch ring: chunk q + XD - 1 is requested while step q computes.  The
        // X load is issued AFTER the step's weight loads, so the in-order vmcnt wait for the next step's weights
        // (older) never waits for it (younger): a row chunk that misses to HBM gets XD - 1 steps of cover.
        constexpr int XD = NCF_X_DEPTH;
        f32x4 w[2][NT1];
        f32x4 x[XD];
        auto xsrc = [&](int q) { return q < qa ? rowA + 8 * q : rowB + 8 * (q - qa); };  // wave-uniform: EA % 8 == 0
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt) w[0][nt] = wp[nt * 64];
#pragma unroll
        for (int t = 0; t < XD - 1; ++t)
            if (t < Q1) x[t] = ldg4(xsrc(t));
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            // One scheduling region per k-step.  MFMAs run nt-major (4 dependent MFMAs per accumulator: the 64-cycle
            // dependent latency of 32x32x2 equals its issue interval, so that costs nothing) and the NEXT step's
            // weight load for tile nt is issued right behind tile nt's MFMAs, in the order the next step consumes
            // them: every load is issued in the shadow of a running MFMA and gets a full step of cover, and the
            // compiler's counted vmcnt waits only for the tile it is about to use.
            const int cur = q & 1, nxt = cur ^ 1;
            const f32x4 xb = x[q % XD] * (q < qa ? zA : zB);  // zero an out-of-range row at USE time
            if (NCF_PAIR && NT1 % 2 == 0) {
#pragma unroll
                for (int nt = 0; nt < NT1; nt += 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt][j], xb[j], acc1[nt], 0, 0, 0);
                        acc1[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt + 1][j], xb[j], acc1[nt + 1], 0, 0, 0);
                    }
                    if (q + 1 < Q1) {
                        w[nxt][nt] = NCF_ABLATE_LOADS ? w[cur][nt] : wp[((q + 1) * NT1 + nt) * 64];
                        w[nxt][nt + 1] = NCF_ABLATE_LOADS ? w[cur][nt + 1] : wp[((q + 1) * NT1 + nt + 1) * 64];
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt][j], xb[j], acc1[nt], 0, 0, 0);
                    if (q + 1 < Q1) w[nxt][nt] = NCF_ABLATE_LOADS ? w[cur][nt] : wp[((q + 1) * NT1 + nt) * 64];
                }
            }
            if (q + XD - 1 < Q1) x[(q + XD - 1) % XD] = NCF_ABLATE_LOADS ? x[q % XD] : ldg4(xsrc(q + XD - 1));
            if (!NCF_ABLATE_LOADS) {
                if (NCF_PAIR && NT1 % 2 == 0) {
#pragma unroll
                    for (int nt = 0; nt < NT1; nt += 2) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);  // 8 MFMA (two tiles alternating)
                        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // 2 VMEM reads
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // 1 VMEM read
                    }
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

#if NCF_STAMP
    const unsigned long long st_rL1 = __builtin_amdgcn_s_memrealtime();
#endif
    float partial = 0.f;
    if constexpr (N2 > 0) {
        // ---- layer 2: acc2 = b2 + W2 . relu(acc1) ; acc1 registers are the B operands ----
        f32x16 acc2[NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bb = ldg4(a.b2 + 32 * nt + 8 * g + 4 * h);
                acc2[nt][4 * g + 0] = bb[0]; acc2[nt][4 * g + 1] = bb[1];
                acc2[nt][4 * g + 2] = bb[2]; acc2[nt][4 * g + 3] = bb[3];
            }
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp2) + lane;
        f32x4 w[2][NT2];
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) w[0][nt] = wp[nt * 64];
#pragma unroll
        for (int q = 0; q < Q2; ++q) {  // q = 4*kb + g : k-block kb of H1 (= tile kb of acc1), group g
            const int cur = q & 1, nxt = cur ^ 1;
            const int kb = q >> 2, g = q & 3;
            f32x4 hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = fmaxf(acc1[kb][4 * g + j], 0.f);  // ReLU (util.py:15)
            if (NCF_PAIR && NT2 % 2 == 0) {
#pragma unroll
                for (int nt = 0; nt < NT2; nt += 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt][j], hv[j], acc2[nt], 0, 0, 0);
                        acc2[nt + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt + 1][j], hv[j], acc2[nt + 1], 0, 0, 0);
                    }
                    if (q + 1 < Q2) {
                        w[nxt][nt] = NCF_ABLATE_LOADS ? w[cur][nt] : wp[((q + 1) * NT2 + nt) * 64];
                        w[nxt][nt + 1] = NCF_ABLATE_LOADS ? w[cur][nt + 1] : wp[((q + 1) * NT2 + nt + 1) * 64];
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt][j], hv[j], acc2[nt], 0, 0, 0);
                    if (q + 1 < Q2) w[nxt][nt] = NCF_ABLATE_LOADS ? w[cur][nt] : wp[((q + 1) * NT2 + nt) * 64];
                }
            }
            if (!NCF_ABLATE_LOADS) {
                if (NCF_PAIR && NT2 % 2 == 0) {
#pragma unroll
                    for (int nt = 0; nt < NT2; nt += 2) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT2; ++nt) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#if NCF_STAMP
        if (a.dbg && lane == 0) a.dbg[tile * 4 + 0] = __builtin_amdgcn_s_memrealtime();  // end of layer 2
#endif
        // ---- last layer (1 wide): out = bl + sum_n wl[n] * relu(acc2[n]) ----
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ww = ldg4(a.wl + 32 * nt + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], fmaxf(acc2[nt][4 * g + j], 0.f), partial);
            }
    } else {
#pragma unroll
        for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 ww = ldg4(a.wl + 32 * nt + 8 * g + 4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], fmaxf(acc1[nt][4 * g + j], 0.f), partial);
            }
    }
    partial += __shfl_xor(partial, 32);  // the two lane halves hold complementary neuron rows
    if (h == 0 && p < a.B) a.out[p] = partial + a.bl[0];
#if NCF_STAMP
    if (a.dbg && lane == 0) {
        const unsigned long long st_t1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        (void)st_t0; (void)st_t1;
        a.dbg[tile * 4 + 1] = st_rL1;          // end of layer 1 (100 MHz ticks)
        a.dbg[tile * 4 + 2] = st_r0;           // wave start
        a.dbg[tile * 4 + 3] = st_r1;           // wave end
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Small-batch variant.  The main kernel gives one wave a whole 32-pair tile (1024 MFMAs, ~27 us): below ~1000 tiles the
// chip is mostly idle and the latency floor is one tile.  Here the FOUR waves of a workgroup share one tile: each wave
// computes a quarter of layer 1's neurons, the ReLU'd activations go through LDS as [k-chunk][pair] float4 (exactly the
// B-operand fragments of layer 2), each wave computes a quarter of layer 2, and wave 0 finishes the 1-wide layer from
// LDS in the SAME operation order as the main kernel — results are bit-identical to it (and to oracle/ncf_oracle_c.c).
template <int K0, int N1, int N2>
__global__ __launch_bounds__(256) void score_fused_small_f32_kernel(FusedArgs a) {
    constexpr int NT1 = N1 / 32, Q1 = K0 / 8, NT2 = N2 / 32, Q2 = N1 / 8;
    constexpr int T1 = NT1 / 4;                 // layer-1 tiles per wave
    constexpr int T2 = (NT2 + 3) / 4;           // layer-2 tiles per wave (waves beyond NT2 idle in layer 2)
    constexpr int WD = 3, XD = 4;               // weight / row prefetch rings (k-steps)
    __shared__ __attribute__((aligned(16))) f32x4 h1s[N1 / 4][32];
    __shared__ __attribute__((aligned(16))) f32x4 h2s[(N2 > 0 ? N2 : 4) / 4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t p = (int64_t)blockIdx.x * 32 + m;
    const int64_t pc = p < a.B ? p : a.B - 1;

    const int64_t ia = a.idxA ? a.idxA[pc] : pc;
    const bool okA = (ia >= 0) & (ia < a.rowsA);
    const float* rowA = a.tabA + (okA ? ia : 0) * a.ldA + 4 * h;
    const int qa = a.EA / 8;
    const float* rowB = rowA;
    bool okB = true;
    if (qa < Q1) {
        const int64_t ib = a.idxB ? a.idxB[pc] : pc;
        okB = (ib >= 0) & (ib < a.rowsB);
        rowB = a.tabB + (okB ? ib : 0) * a.ldB + 4 * h;
    }
    if (!(okA & okB) && a.oob && wave == 0 && p < a.B) *a.oob = 1;
    const float zA = okA ? 1.f : 0.f, zB = okB ? 1.f : 0.f;
    auto xsrc = [&](int q) { return q < qa ? rowA + 8 * q : rowB + 8 * (q - qa); };

    // ---- layer 1: this wave's tiles nt = wave*T1 + t ----
    f32x16 acc1[T1];
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bb = ldg4(a.b1 + 32 * (wave * T1 + t) + 8 * g + 4 * h);
            acc1[t][4 * g + 0] = bb[0]; acc1[t][4 * g + 1] = bb[1];
            acc1[t][4 * g + 2] = bb[2]; acc1[t][4 * g + 3] = bb[3];
        }
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp1) + lane;
        f32x4 w[WD][T1];
        f32x4 x[XD];
#pragma unroll
        for (int s0 = 0; s0 < WD - 1; ++s0)
            if (s0 < Q1) {
#pragma unroll
                for (int t = 0; t < T1; ++t) w[s0][t] = wp[(s0 * NT1 + wave * T1 + t) * 64];
            }
#pragma unroll
        for (int s0 = 0; s0 < XD - 1; ++s0)
            if (s0 < Q1) x[s0] = ldg4(xsrc(s0));
#pragma unroll
        for (int q = 0; q < Q1; ++q) {
            if (q + WD - 1 < Q1) {
#pragma unroll
                for (int t = 0; t < T1; ++t) w[(q + WD - 1) % WD][t] = wp[((q + WD - 1) * NT1 + wave * T1 + t) * 64];
            }
            if (q + XD - 1 < Q1) x[(q + XD - 1) % XD] = ldg4(xsrc(q + XD - 1));
            __builtin_amdgcn_sched_barrier(0);
            const f32x4 xb = x[q % XD] * (q < qa ? zA : zB);
#pragma unroll
            for (int t = 0; t < T1; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc1[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q % WD][t][j], xb[j], acc1[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ReLU'd activations -> LDS in layer-2 B-fragment order: chunk (8nt + 2g + h) of pair m = neurons 32nt+8g+4h+0..3
#pragma unroll
    for (int t = 0; t < T1; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc1[t][4 * g + j], 0.f);
            h1s[8 * (wave * T1 + t) + 2 * g + h][m] = v;
        }
    __syncthreads();

    if constexpr (N2 > 0) {
        f32x16 acc2[T2];
        const bool l2_active = wave < NT2;      // wave-uniform
        if (l2_active) {
#pragma unroll
            for (int t = 0; t < T2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nt = wave + 4 * t;
                    const f32x4 bb = ldg4(a.b2 + 32 * (nt < NT2 ? nt : 0) + 8 * g + 4 * h);
                    acc2[t][4 * g + 0] = bb[0]; acc2[t][4 * g + 1] = bb[1];
                    acc2[t][4 * g + 2] = bb[2]; acc2[t][4 * g + 3] = bb[3];
                }
            const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp2) + lane;
            f32x4 w[WD][T2];
#pragma unroll
            for (int s0 = 0; s0 < WD - 1; ++s0)
#pragma unroll
                for (int t = 0; t < T2; ++t) {
                    const int nt = wave + 4 * t;
                    w[s0][t] = wp[(s0 * NT2 + (nt < NT2 ? nt : 0)) * 64];
                }
            f32x4 hvr[2];
            hvr[0] = h1s[h][m];
#pragma unroll
            for (int q = 0; q < Q2; ++q) {
                if (q + WD - 1 < Q2) {
#pragma unroll
                    for (int t = 0; t < T2; ++t) {
                        const int nt = wave + 4 * t;
                        w[(q + WD - 1) % WD][t] = wp[((q + WD - 1) * NT2 + (nt < NT2 ? nt : 0)) * 64];
                    }
                }
                if (q + 1 < Q2) hvr[(q + 1) & 1] = h1s[2 * (q + 1) + h][m];
                __builtin_amdgcn_sched_barrier(0);
                const f32x4 hv = hvr[q & 1];
#pragma unroll
                for (int t = 0; t < T2; ++t)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc2[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[q % WD][t][j], hv[j], acc2[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < T2; ++t) {
                const int nt = wave + 4 * t;
                if (nt < NT2) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = fmaxf(acc2[t][4 * g + j], 0.f);
                        h2s[8 * nt + 2 * g + h][m] = v;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (wave != 0) return;
    // ---- last layer, same chain order as the main kernel: per lane half over nt, g, j; then half 0 + half 1 ----
    constexpr int NTL = (N2 > 0 ? N2 : N1) / 32;
    float partial = 0.f;
#pragma unroll
    for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ww = ldg4(a.wl + 32 * nt + 8 * g + 4 * h);
            const f32x4 v = (N2 > 0) ? h2s[8 * nt + 2 * g + h][m] : h1s[8 * nt + 2 * g + h][m];
#pragma unroll
            for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], v[j], partial);
        }
    partial += __shfl_xor(partial, 32);
    if (h == 0 && p < a.B) a.out[p] = partial + a.bl[0];
}

// ---------------------------------------------------------------------------------------------------------------
// Folded first layer (opt-in, inference-time weight folding): with frozen weights
//     relu(W1 . cat(a, b) + b1) = relu(PA[ia] + PB[ib]),   PA = TA . W1[:, :EA]^T + b1,   PB = TB . W1[:, EA:]^T
// so layer 1 needs no matrix work at all: the "embedding" rows become N1-wide pre-activations (4x the table bytes at
// E = 64, N1 = 256) and the kernel is  gather 2 x N1 floats -> add -> ReLU -> layer 2 (MFMA) -> 1-wide layer.
// Lane (m, h) loads the 16-byte chunk (8kb + 2g + h) of pair m's two rows DIRECTLY in accumulator-row order
// (neuron 32kb + 8g + 4h + j), i.e. straight into the B operand of layer 2's k-step 4kb + g.
// Layer 2's packed weights (N1*N2 floats = 128 KB at 256x128) live in LDS for the whole workgroup (8 waves, 256
// pairs): they are read with ds_read_b128 on lgkmcnt, so vmcnt tracks ONLY the HBM row gathers and those can be
// prefetched several k-blocks ahead without the in-order vmcnt making every weight wait queue behind an HBM miss.
// Bound: fp32 MFMA on layer 2 (2*N1*N2 + 2*N2 FLOP/pair = 65 792 at 256x128) vs HBM 2*N1*4 + 20 B/pair = 2068 B.
#ifndef NCF_FOLD_PF
#define NCF_FOLD_PF 2   // accumulator tiles (k-blocks of 32 neurons) of row data requested ahead; measured 2 -> 45.6 us, 3 -> 47.5, 5 -> 55.3, 8 (all up front) -> 69.3
#endif
#ifndef NCF_FOLD_ABLATE
#define NCF_FOLD_ABLATE 0   // diagnostics: 1 = no row gathers in the loop (constant data), 2 = weights from registers only
#endif

struct FoldArgs {
    const float* PA; int64_t rowsA; int64_t ldA;
    const float* PB; int64_t rowsB; int64_t ldB;
    const int64_t* idxA; const int64_t* idxB;
    int64_t B;
    const float* Wp2; const float* b2; const float* wl; const float* bl;
    float* out; int32_t* oob;
};

template <int N1, int N2>
__global__ __launch_bounds__(512, 2) void score_folded_f32_kernel(FoldArgs a) {
    constexpr int NT1 = N1 / 32, NT2 = N2 / 32, Q2 = N1 / 8;
    constexpr int PF = NCF_FOLD_PF < NT1 ? NCF_FOLD_PF : NT1;
    constexpr int RING = PF + 1;
    __shared__ __attribute__((aligned(16))) f32x4 wlds[Q2 * NT2 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 8 + wave;
    const int64_t p = tile * 32 + m;
    const int64_t pc = p < a.B ? p : a.B - 1;

    const int64_t ia = a.idxA ? a.idxA[pc] : pc;
    const int64_t ib = a.idxB ? a.idxB[pc] : pc;
    const bool okA = (ia >= 0) & (ia < a.rowsA), okB = (ib >= 0) & (ib < a.rowsB);
    if (!(okA & okB) && a.oob && p < a.B) *a.oob = 1;
    const float* rowA = a.PA + (okA ? ia : 0) * a.ldA + 4 * h;
    const float* rowB = a.PB + (okB ? ib : 0) * a.ldB + 4 * h;
    const float zA = okA ? 1.f : 0.f, zB = okB ? 1.f : 0.f;

    // row-data ring: tile kb = chunks g = 0..3 at row + 32kb + 8g
    f32x4 pa[RING][4], pb[RING][4];
#pragma unroll
    for (int t = 0; t < PF; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pa[t][g] = ldg4(rowA + 32 * t + 8 * g);
            pb[t][g] = ldg4(rowB + 32 * t + 8 * g);
        }
    // layer-2 weights -> LDS once per workgroup
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(a.Wp2);
        for (int i = threadIdx.x; i < Q2 * NT2 * 64; i += 512) wlds[i] = src[i];
    }
    f32x16 acc2[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bb = ldg4(a.b2 + 32 * nt + 8 * g + 4 * h);
            acc2[nt][4 * g + 0] = bb[0]; acc2[nt][4 * g + 1] = bb[1];
            acc2[nt][4 * g + 2] = bb[2]; acc2[nt][4 * g + 3] = bb[3];
        }
    __syncthreads();

    const f32x4* wq = wlds + lane;  // + (q*NT2 + nt)*64
    f32x4 w[2][NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) w[0][nt] = wq[nt * 64];
#pragma unroll
    for (int kb = 0; kb < NT1; ++kb) {
        if (kb + PF < NT1 && NCF_FOLD_ABLATE != 1) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                pa[(kb + PF) % RING][g] = ldg4(rowA + 32 * (kb + PF) + 8 * g);
                pb[(kb + PF) % RING][g] = ldg4(rowB + 32 * (kb + PF) + 8 * g);
            }
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 4 * kb + g, cur = q & 1, nxt = cur ^ 1;
            const f32x4 s4 = pa[kb % RING][g] * zA + pb[kb % RING][g] * zB;  // bias b1 is folded into PA
            f32x4 hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = fmaxf(s4[j], 0.f);
#pragma unroll
            for (int nt = 0; nt < NT2; ++nt) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt][j], hv[j], acc2[nt], 0, 0, 0);
                if (q + 1 < Q2) w[nxt][nt] = (NCF_FOLD_ABLATE == 2) ? w[cur][nt] : wq[((q + 1) * NT2 + nt) * 64];
            }
            if (NCF_FOLD_ABLATE != 2) {
#pragma unroll
                for (int nt = 0; nt < NT2; ++nt) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // 4 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float partial = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ww = ldg4(a.wl + 32 * nt + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], fmaxf(acc2[nt][4 * g + j], 0.f), partial);
        }
    partial += __shfl_xor(partial, 32);
    if (h == 0 && p < a.B) a.out[p] = partial + a.bl[0];
}

#ifndef NCF_FOLD_DIRECT
#define NCF_FOLD_DIRECT 1   // 1: weights streamed per wave from L2 like the fused kernel (no LDS, no barrier); 0: LDS-resident W2
#endif
#ifndef NCF_FOLD_XD
#define NCF_FOLD_XD 4       // k-steps of row data requested ahead in the direct variant
#endif

// Direct variant: exactly the fused kernel's layer-2 step (next step's weights interleaved behind the MFMAs), with the
// B operand taken from a ring of gathered pre-activation chunks instead of layer-1 accumulators.  One wave = 32 pairs,
// 256-thread workgroups, two per CU, no LDS and no barrier.  The two row loads of k-step q + XD are issued after the
// step's weight loads so the in-order vmcnt wait for the next step's weights never queues behind an HBM miss.
template <int N1, int N2>
__global__ __launch_bounds__(256, 2) void score_folded_direct_kernel(FoldArgs a) {
    constexpr int NT2 = N2 / 32, Q2 = N1 / 8;
    constexpr int XD = NCF_FOLD_XD;
    const int lane = threadIdx.x & 63;
    const int m = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * 32 >= a.B) return;
    const int64_t p = tile * 32 + m;
    const int64_t pc = p < a.B ? p : a.B - 1;
    const int64_t ia = a.idxA ? a.idxA[pc] : pc;
    const int64_t ib = a.idxB ? a.idxB[pc] : pc;
    const bool okA = (ia >= 0) & (ia < a.rowsA), okB = (ib >= 0) & (ib < a.rowsB);
    if (!(okA & okB) && a.oob) *a.oob = 1;
    const float* rowA = a.PA + (okA ? ia : 0) * a.ldA + 4 * h;
    const float* rowB = a.PB + (okB ? ib : 0) * a.ldB + 4 * h;
    const float zA = okA ? 1.f : 0.f, zB = okB ? 1.f : 0.f;

    f32x16 acc2[NT2];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bb = ldg4(a.b2 + 32 * nt + 8 * g + 4 * h);
            acc2[nt][4 * g + 0] = bb[0]; acc2[nt][4 * g + 1] = bb[1];
            acc2[nt][4 * g + 2] = bb[2]; acc2[nt][4 * g + 3] = bb[3];
        }
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.Wp2) + lane;
    f32x4 w[2][NT2];
    f32x4 xa[XD], xb[XD];
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt) w[0][nt] = wp[nt * 64];
#pragma unroll
    for (int t = 0; t < XD - 1; ++t)
        if (t < Q2) { xa[t] = ldg4(rowA + 8 * t); xb[t] = ldg4(rowB + 8 * t); }
#pragma unroll
    for (int q = 0; q < Q2; ++q) {
        const int cur = q & 1, nxt = cur ^ 1;
        const f32x4 s4 = xa[q % XD] * zA + xb[q % XD] * zB;  // b1 is folded into PA
        f32x4 hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[j] = fmaxf(s4[j], 0.f);
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc2[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[cur][nt][j], hv[j], acc2[nt], 0, 0, 0);
            if (q + 1 < Q2) w[nxt][nt] = wp[((q + 1) * NT2 + nt) * 64];
        }
        if (q + XD - 1 < Q2) {
            xa[(q + XD - 1) % XD] = ldg4(rowA + 8 * (q + XD - 1));
            xb[(q + XD - 1) % XD] = ldg4(rowB + 8 * (q + XD - 1));
        }
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float partial = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT2; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 ww = ldg4(a.wl + 32 * nt + 8 * g + 4 * h);
#pragma unroll
            for (int j = 0; j < 4; ++j) partial = fmaf(ww[j], fmaxf(acc2[nt][4 * g + j], 0.f), partial);
        }
    partial += __shfl_xor(partial, 32);
    if (h == 0 && p < a.B) a.out[p] = partial + a.bl[0];
}

template <int N1, int N2>
static void launch_folded(const FoldArgs& a, hipStream_t s) {
    const int64_t tiles = (a.B + 31) / 32;
    if (NCF_FOLD_DIRECT)
        hipLaunchKernelGGL((score_folded_direct_kernel<N1, N2>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((score_folded_f32_kernel<N1, N2>), dim3((unsigned)((tiles + 7) / 8)), dim3(512), 0, s, a);
}

static bool folded_dispatch(int N1, int N2, const FoldArgs* a, hipStream_t s) {
#define X(n1, n2) \
    if (N1 == n1 && N2 == n2) { if (a) launch_folded<n1, n2>(*a, s); return true; }
    X(256, 128) X(128, 64) X(256, 64) X(128, 128)
#undef X
    return false;
}

// Pack W [N][K] row-major into Wp[q][nt][lane][4].
__global__ void pack_weight_kernel(const float* __restrict__ W, int N, int K, float* __restrict__ Wp) {
    const int64_t total = (int64_t)N * K;
    const int NT = N / 32;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(o & 3);
        const int lane = (int)((o >> 2) & 63);
        const int64_t t = o >> 8;  // q*NT + nt
        const int nt = (int)(t % NT), q = (int)(t / NT);
        Wp[o] = W[(int64_t)(32 * nt + (lane & 31)) * K + 8 * q + 4 * (lane >> 5) + j];
    }
}

__global__ void copy_or_zero_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : 0.f;
}

// blob layout (floats): Wp1[N1*K0] b1[N1] (Wp2[N2*N1] b2[N2])? wl[Nlast] bl[1] pad
struct BlobLayout {
    size_t wp1, b1, wp2, b2, wl, bl, total;
};
static BlobLayout blob_layout(const int* dims, int n_layers) {
    BlobLayout L{};
    const size_t K0 = dims[0], N1 = dims[1];
    size_t off = 0;
    L.wp1 = off; off += N1 * K0;
    L.b1 = off; off += N1;
    size_t last = N1;
    if (n_layers == 3) {
        const size_t N2 = dims[2];
        L.wp2 = off; off += N2 * N1;
        L.b2 = off; off += N2;
        last = N2;
    }
    L.wl = off; off += last;
    L.bl = off; off += 4;  // keep 16-byte granularity
    L.total = off;
    return L;
}

typedef void (*fused_fn)(FusedArgs);
#ifndef NCF_SMALL_TILES
#define NCF_SMALL_TILES 768   // below this many 32-pair tiles the 4-waves-per-tile kernel is used (measured crossover)
#endif

#ifndef NCF_HYBRID
#define NCF_HYBRID 1          // ragged last round of the one-wave-per-tile kernel goes to the 4-waves-per-tile kernel
#endif
#ifndef NCF_HYBRID_MAX_PERMILLE
#define NCF_HYBRID_MAX_PERMILLE 850   // ... when it fills less than this share of a round (above: one more full round is cheaper)
#endif

static int fused_num_cus() { return num_cus(); }  // per-device cache in abi.hip

template <int K0, int N1, int N2>
static void launch_inst(const FusedArgs& a, hipStream_t s) {
    const int64_t tiles = (a.B + 31) / 32;
    auto launch_main = [&](const FusedArgs& x) {
        const int64_t t = (x.B + 31) / 32;
        hipLaunchKernelGGL((score_fused_f32_kernel<K0, N1, N2>), dim3((unsigned)((t + NCF_WG_WAVES - 1) / NCF_WG_WAVES)),
                           dim3(NCF_WG_WAVES * 64), 0, s, x);
    };
    if constexpr ((N1 / 32) % 4 == 0 && (N2 / 32) <= 4) {
        auto launch_small = [&](const FusedArgs& x) {
            hipLaunchKernelGGL((score_fused_small_f32_kernel<K0, N1, N2>), dim3((unsigned)((x.B + 31) / 32)), dim3(256), 0, s, x);
        };
        if (tiles < NCF_SMALL_TILES) return launch_small(a);
        // The main kernel's time is a staircase: one step (34.7 us for 128-256-128 on an MI355X) per ROUND of
        // 4 x CUs tiles, whether the round is full or not; the small kernel is linear (1.22 us per 1024 pairs) and
        // computes every pair in the same operation order, so a ragged tail can move to it without changing a bit.
        // Measured (us, main | small): 16 384 pairs 35.8 | 22.3, 24 576: 35.9 | 31.5, 32 768: 38.0 | 42.0,
        // 49 152: 68.3 | 59.9, 65 536: 69.5 | 78.1.  Identity ids (idx == NULL) cannot be offset: no split there.
        const int64_t round = 4 * (int64_t)fused_num_cus();
        const int64_t full = tiles / round, rem = tiles % round;
        const bool ids_ok = a.idxA && (a.EA == K0 || a.idxB);
        if (NCF_HYBRID && ids_ok && rem > 0 && rem * 1000 < round * NCF_HYBRID_MAX_PERMILLE) {
            const int64_t split = full * round * 32;   // pairs handled by the full rounds
            FusedArgs tail = a;
            tail.idxA = a.idxA + split;
            if (a.idxB) tail.idxB = a.idxB + split;
            tail.out = a.out + split;
            tail.B = a.B - split;
            if (full > 0) {
                FusedArgs head = a;
                head.B = split;
                launch_main(head);
            }
            return launch_small(tail);
        }
    }
    launch_main(a);
}

#define NCF_FUSED_INSTANCES(X) \
    X(64, 256, 128) X(64, 256, 0) X(64, 128, 0) X(64, 128, 64) \
    X(128, 256, 128) X(128, 256, 0) X(128, 128, 0) X(128, 128, 64) \
    X(256, 256, 128) X(256, 256, 0) X(256, 128, 0)

static bool fused_dispatch(int K0, int N1, int N2, const FusedArgs* a, hipStream_t s) {
#define X(k, n1, n2) \
    if (K0 == k && N1 == n1 && N2 == n2) { if (a) launch_inst<k, n1, n2>(*a, s); return true; }
    NCF_FUSED_INSTANCES(X)
#undef X
    return false;
}

static bool fused_shape_ok(int dtype, int EA, int EB, int n_layers, const int* dims) {
    if (dtype != NCF_F32 || !dims) return false;
    if (n_layers != 2 && n_layers != 3) return false;
    if (dims[n_layers] != 1) return false;
    if (EA <= 0 || EB < 0 || EA % 8 || EB % 8 || EA + EB != dims[0]) return false;
    return fused_dispatch(dims[0], dims[1], n_layers == 3 ? dims[2] : 0, nullptr, nullptr);
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_score_fused_supported(int dtype, int EA, int EB, int n_layers, const int* dims) {
    if (dtype == NCF_BF16) return bf16_shape_ok(EA, EB, n_layers, dims) ? 1 : 0;
    return fused_shape_ok(dtype, EA, EB, n_layers, dims) ? 1 : 0;
}

extern "C" size_t ncf_mlp_packed_bytes(int dtype, int n_layers, const int* dims) {
    if (dtype == NCF_BF16) return bf16_packed_bytes(n_layers, dims);
    if (dtype != NCF_F32 || !dims || (n_layers != 2 && n_layers != 3)) return 0;
    return blob_layout(dims, n_layers).total * sizeof(float);
}

extern "C" int ncf_mlp_pack(int dtype, int n_layers, const int* dims, const void* const* W, const void* const* b,
                            void* packed, size_t packed_bytes, ncf_stream_t stream) {
    if (dtype != NCF_F32 && dtype != NCF_BF16) return fail(NCF_EINVAL, "ncf_mlp_pack: bad dtype %d", dtype);
    if (!dims || !W || !packed || (n_layers != 2 && n_layers != 3)) return fail(NCF_EINVAL, "ncf_mlp_pack: bad argument");
    if (dtype == NCF_BF16) {
        for (int i = 0; i < n_layers; ++i)
            if (!W[i]) return fail(NCF_EINVAL, "ncf_mlp_pack: W[%d] is null", i);
        return bf16_pack(n_layers, dims, W, b, packed, packed_bytes, (hipStream_t)stream);
    }
    if (dims[n_layers] != 1) return fail(NCF_EUNSUPPORTED, "ncf_mlp_pack: last layer must be 1 wide");
    for (int i = 0; i < n_layers; ++i) {
        if (!W[i]) return fail(NCF_EINVAL, "ncf_mlp_pack: W[%d] is null", i);
        if (i < n_layers - 1 && (dims[i] % 8 || dims[i + 1] % 32))
            return fail(NCF_EUNSUPPORTED, "ncf_mlp_pack: layer %d dims (%d -> %d) not tileable (K %% 8, N %% 32)", i, dims[i], dims[i + 1]);
    }
    const BlobLayout L = blob_layout(dims, n_layers);
    if (packed_bytes < L.total * sizeof(float)) return fail(NCF_EWORKSPACE, "ncf_mlp_pack: packed buffer too small");
    hipStream_t s = (hipStream_t)stream;
    float* P = (float*)packed;
    auto bias = [&](int i) { return b ? (const float*)b[i] : nullptr; };
    hipLaunchKernelGGL(pack_weight_kernel, dim3(256), dim3(256), 0, s, (const float*)W[0], dims[1], dims[0], P + L.wp1);
    hipLaunchKernelGGL(copy_or_zero_kernel, dim3((dims[1] + 255) / 256), dim3(256), 0, s, bias(0), dims[1], P + L.b1);
    int last = dims[1];
    if (n_layers == 3) {
        hipLaunchKernelGGL(pack_weight_kernel, dim3(256), dim3(256), 0, s, (const float*)W[1], dims[2], dims[1], P + L.wp2);
        hipLaunchKernelGGL(copy_or_zero_kernel, dim3((dims[2] + 255) / 256), dim3(256), 0, s, bias(1), dims[2], P + L.b2);
        last = dims[2];
    }
    hipLaunchKernelGGL(copy_or_zero_kernel, dim3((last + 255) / 256), dim3(256), 0, s, (const float*)W[n_layers - 1], last, P + L.wl);
    hipLaunchKernelGGL(copy_or_zero_kernel, dim3(1), dim3(256), 0, s, bias(n_layers - 1), 1, P + L.bl);
    return check_launch("ncf_mlp_pack");
}

extern "C" int ncf_score_fused(int dtype, const void* tabA, int64_t rowsA, int64_t ldA, const void* tabB, int64_t rowsB,
                               int64_t ldB, const int64_t* idxA, const int64_t* idxB, int64_t B, int EA, int EB,
                               int n_layers, const int* dims, const void* packed, float* out, int32_t* oob,
                               ncf_stream_t stream) {
    if (dtype == NCF_BF16) {
        if (!bf16_shape_ok(EA, EB, n_layers, dims))
            return fail(NCF_EUNSUPPORTED, "ncf_score_fused: no bf16 kernel for EA=%d EB=%d layers=%d", EA, EB, n_layers);
        if (B == 0) return NCF_OK;
        if (B < 0 || !tabA || (EB > 0 && !tabB) || !packed || !out) return fail(NCF_EINVAL, "ncf_score_fused: bad argument");
        return bf16_score(tabA, rowsA, ldA, tabB, rowsB, ldB, idxA, idxB, B, EA, EB, n_layers, dims, packed, out, oob, (hipStream_t)stream);
    }
    if (!fused_shape_ok(dtype, EA, EB, n_layers, dims))
        return fail(NCF_EUNSUPPORTED, "ncf_score_fused: no specialised kernel for dtype=%d EA=%d EB=%d layers=%d", dtype, EA, EB, n_layers);
    if (B == 0) return NCF_OK;
    if (B < 0 || !tabA || (EB > 0 && !tabB) || !packed || !out) return fail(NCF_EINVAL, "ncf_score_fused: bad argument");
    if (ldA < EA || (EB > 0 && ldB < EB) || ldA % 4 || (EB > 0 && ldB % 4) || !aligned16(tabA) || (EB > 0 && !aligned16(tabB)) || !aligned16(packed))
        return fail(NCF_EINVAL, "ncf_score_fused: tables must be 16-byte aligned with ld %% 4 == 0");
    if (B == 0) return NCF_OK;
    const BlobLayout L = blob_layout(dims, n_layers);
    const float* P = (const float*)packed;
    FusedArgs a;
    a.tabA = (const float*)tabA; a.rowsA = rowsA; a.ldA = ldA;
    a.tabB = (const float*)(EB ? tabB : tabA); a.rowsB = EB ? rowsB : rowsA; a.ldB = EB ? ldB : ldA;
    a.idxA = idxA; a.idxB = idxB; a.B = B; a.EA = EA;
    a.Wp1 = P + L.wp1; a.b1 = P + L.b1;
    a.Wp2 = n_layers == 3 ? P + L.wp2 : nullptr; a.b2 = n_layers == 3 ? P + L.b2 : nullptr;
    a.wl = P + L.wl; a.bl = P + L.bl;
    a.out = out; a.oob = oob;
#if NCF_STAMP
    a.dbg = g_dbg;
#endif
    fused_dispatch(dims[0], dims[1], n_layers == 3 ? dims[2] : 0, &a, (hipStream_t)stream);
    return check_launch("ncf_score_fused");
}

#if NCF_STAMP
extern "C" void ncf_dev_set_debug_buffer(void* p) { ncf::g_dbg = (unsigned long long*)p; }
#endif

extern "C" int ncf_score_folded_supported(int dtype, int N1, int N2) {
    return (dtype == NCF_F32 && folded_dispatch(N1, N2, nullptr, nullptr)) ? 1 : 0;
}

extern "C" int ncf_score_folded(int dtype, const void* PA, int64_t rowsA, int64_t ldA, const void* PB, int64_t rowsB, int64_t ldB,
                                const int64_t* idxA, const int64_t* idxB, int64_t B, int N1, int N2, const void* packed_tail,
                                float* out, int32_t* oob, ncf_stream_t stream) {
    if (dtype != NCF_F32 || !folded_dispatch(N1, N2, nullptr, nullptr))
        return fail(NCF_EUNSUPPORTED, "ncf_score_folded: no kernel for dtype=%d N1=%d N2=%d", dtype, N1, N2);
    if (B == 0) return NCF_OK;
    if (B < 0 || !PA || !PB || !packed_tail || !out) return fail(NCF_EINVAL, "ncf_score_folded: bad argument");
    if (ldA < N1 || ldB < N1 || ldA % 4 || ldB % 4 || !aligned16(PA) || !aligned16(PB) || !aligned16(packed_tail))
        return fail(NCF_EINVAL, "ncf_score_folded: tables must be 16-byte aligned with ld %% 4 == 0 and ld >= N1");
    const int dims[3] = {N1, N2, 1};
    const BlobLayout L = blob_layout(dims, 2);  // the tail MLP [N1 -> N2 -> 1] packed by ncf_mlp_pack
    const float* P = (const float*)packed_tail;
    FoldArgs a;
    a.PA = (const float*)PA; a.rowsA = rowsA; a.ldA = ldA;
    a.PB = (const float*)PB; a.rowsB = rowsB; a.ldB = ldB;
    a.idxA = idxA; a.idxB = idxB; a.B = B;
    a.Wp2 = P + L.wp1; a.b2 = P + L.b1; a.wl = P + L.wl; a.bl = P + L.bl;
    a.out = out; a.oob = oob;
    folded_dispatch(N1, N2, &a, (hipStream_t)stream);
    return check_launch("ncf_score_folded");
}
