// K2 generic path — C[M][N] = act(A[M][K] . W[N][K]^T + bias) for ANY M, N, K (fp32, gfx950), and the
// layer-by-layer MLP driver ncf_mlp_forward built on it.
//
// LDS-tiled v_mfma_f32_32x32x2_f32 GEMM: workgroup tile 128 (M) x 64 (N), BK = 32, 4 waves each owning 32
// rows x 64 columns (two 32x32 accumulators).  A and W tiles are staged through LDS with +4 float row padding
// (row stride 36 floats: 16 consecutive rows start on 16 distinct 16-byte bank slots, so the ds_read_b128
// operand reads are conflict-free).  Ragged edges (K = 2094, N = 1 ...) are zero-filled at staging time.
// The K order inside each 8-wide group is permuted identically for both operands (lane half h takes k = 8g+4h+j),
// which keeps every MFMA fed by one 16-byte LDS read per operand per 4 k-steps.
// N <= 8 goes to a row-dot kernel (one 16-lane group per output element) instead of wasting a 64-wide tile.
#include "ncf_common.h"
#include <stdlib.h>
#include <string.h>

namespace ncf {

constexpr int BM = 128, BN = 64, BK = 32, LDT = BK + 4;

template <bool RELU>
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, float* __restrict__ C,
                                                         int64_t ldc, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Ws[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // stage A tile: 128 x 32 floats, 16 per thread, k fastest across lanes (coalesced 128-B row segments)
#pragma unroll
        for (int t = 0; t < (BM * BK) / 256; ++t) {
            const int e = t * 256 + tid;
            const int r = e >> 5, k = e & 31;
            const int64_t gm = m0 + r;
            const int gk = k0 + k;
            As[r * LDT + k] = (gm < M && gk < K) ? A[gm * lda + gk] : 0.f;
        }
#pragma unroll
        for (int t = 0; t < (BN * BK) / 256; ++t) {
            const int e = t * 256 + tid;
            const int r = e >> 5, k = e & 31;
            const int gn = n0 + r;
            const int gk = k0 + k;
            Ws[r * LDT + k] = (gn < N && gk < K) ? W[(int64_t)gn * ldw + gk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(&As[(32 * wave + i) * LDT + 8 * g + 4 * h]);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(&Ws[(i)*LDT + 8 * g + 4 * h]);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(&Ws[(32 + i) * LDT + 8 * g + 4 * h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], w0[j], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], w1[j], acc[1], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // D[m][n]: column n = lane & 31, row m = acc_row(r, h)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int gn = n0 + 32 * t + i;
        if (gn >= N) continue;
        const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t gm = m0 + 32 * wave + acc_row(r, h);
            if (gm < M) {
                float v = acc[t][r] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                C[gm * ldc + gn] = v;
            }
        }
    }
}

// Small-N path: out[m][n] = act(bias[n] + sum_k A[m][k] * W[n][k]); one 16-lane group per (m, n).
template <bool RELU>
__global__ __launch_bounds__(256) void rowdot_f32_kernel(const float* __restrict__ A, int64_t lda,
                                                         const float* __restrict__ W, int64_t ldw,
                                                         const float* __restrict__ bias, float* __restrict__ C,
                                                         int64_t ldc, int64_t M, int N, int K) {
    const int sub = threadIdx.x & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    const int64_t total = M * N;
    const int64_t total_pad = (total + 3) & ~int64_t(3);  // uniform trip count inside a wave (4 groups)
    for (int64_t o = grp; o < total_pad; o += ngrp) {
        float acc = 0.f;
        const int64_t m = o / N;
        const int n = (int)(o - m * N);
        if (o < total) {
            const float* a = A + m * lda;
            const float* w = W + (int64_t)n * ldw;
            for (int k = sub; k < K; k += 16) acc = fmaf(a[k], w[k], acc);
        }
        acc += __shfl_xor(acc, 8);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 1);
        if (o < total && sub == 0) {
            float v = acc + (bias ? bias[n] : 0.f);
            if (RELU) v = fmaxf(v, 0.f);
            C[m * ldc + n] = v;
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Row-streaming GEMM for N in {32, 64, 128, 256}: no LDS staging at all.  One wave owns 32 rows of A and ALL N output
// columns (NT = N/32 accumulator tiles).  Lane (i, h) reads the 16-byte chunk k = 8q + 4h .. +3 of ITS row of A
// (A operand) and of row 32nt + i of W (B operand) straight from global memory — the same identical-k-permutation
// trick as the fused kernel, so one 16-byte load per operand feeds 4 MFMAs; W is small and stays L1/L2 resident.
// Output registers hold rows, lanes hold columns -> every store instruction writes 128 contiguous bytes per row.
// Skinny problems (few row tiles, long K: the F = 2094 candidate Linear of AttentionNCF) split K over the KS waves
// of a tile and add the slices through LDS in slice order (deterministic).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // rows of A may be only 4-byte aligned

template <int NT>
struct RsFrag {
    f32x4 a;
    f32x4 w[NT];
};

template <int NT>
__device__ __forceinline__ void rs_load(RsFrag<NT>& f, const float* arow, const float* const (&wrow)[NT], int q) {
    f.a = *reinterpret_cast<const f32x4u*>(arow + 8 * q);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) f.w[nt] = *reinterpret_cast<const f32x4u*>(wrow[nt] + 8 * q);
}

template <int NT>
__device__ __forceinline__ void rs_mma(const RsFrag<NT>& f, f32x16 (&acc)[NT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[j], f.w[nt][j], acc[nt], 0, 0, 0);
}

template <int NT, int KS, bool RELU>
__global__ __launch_bounds__(KS > 4 ? KS * 64 : 256) void linear_rs_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                        int64_t ldw, const float* __restrict__ bias, float* __restrict__ C,
                                                        int64_t ldc, int64_t M, int K) {
    constexpr int WAVES = KS > 4 ? KS : 4;   // KS = 8 / 16: one tile per 512- / 1024-thread workgroup
    // KS >= 8: the same array first stages each wave's A / W blocks (2 x 32 x 32 floats per wave), then carries the partial sums
    __shared__ __attribute__((aligned(16))) float red[KS >= 8 ? WAVES * 2048 : (KS > 1 ? WAVES * NT * 16 * 64 : 4)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t tile = (int64_t)blockIdx.x * (WAVES / KS) + wave / KS;
    const int ks = wave % KS;
    // gridDim.y column blocks of 32*NT outputs each (skinny problems run narrower blocks to fill the chip)
    W += (int64_t)blockIdx.y * 32 * NT * ldw;
    C += (int64_t)blockIdx.y * 32 * NT;
    if (bias) bias += blockIdx.y * 32 * NT;
    const bool tile_ok = tile * 32 < M;  // wave-uniform
    const int64_t m = tile * 32 + i;
    const float* arow = A + (m < M ? m : (M - 1)) * lda + 4 * h;
    const float* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (int64_t)(32 * nt + i) * ldw + 4 * h;

    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

    const int Q = K / 8;
    // KS >= 8 (skinny and deep): the whole 32-wide blocks of K go through the staged loop below, split over the waves; what is
    // left of K (< 32) is ONE wave's, on the direct path
    // (wave 0: the integer split gives the LAST wave the extra block, so the remainder goes to the first)
    constexpr int TAILW = KS >= 8 ? 0 : KS - 1;              // the wave that owns the ragged end of K
    const int qlo = KS >= 8 ? (ks == TAILW ? 4 * (K / 32) : 0) : (int)((int64_t)ks * Q / KS);
    const int qhi = KS >= 8 ? (ks == TAILW ? Q : 0) : (int)((int64_t)(ks + 1) * Q / KS);
    if (tile_ok) {
      if constexpr (KS >= 8) {
        // Staged form.  Read straight into MFMA operand layout (lane = row), a 16-byte load per lane touches 32 rows: 32 pieces of
        // 32 bytes per instruction, and at K = 2094 the CU's address path, not HBM or the MFMAs, set the time (4096 x 2094 -> 64:
        // 30 us for 34 MB and 1.1 GFLOP).  Here a wave loads its 32 x 32 block of A and of W with eight consecutive lanes per row
        // (whole 128-byte runs: 8 lines per instruction instead of 32 pieces), parks them in its own 8 KB of LDS (16-byte slot s
        // of row r at s ^ ((r >> 1) & 7): conflict-free ds_write_b128 and operand-layout ds_read_b128) and reads them back as
        // operands.  No barrier: a wave reads only what it wrote, and a wave's LDS instructions execute in order.  Same k pairing
        // per MFMA as the direct path (k = 8q + 4h + j).
        static_assert(NT == 1, "the staged split-K form keeps one column tile per workgroup");
        float* const stA = red + (size_t)wave * 2048;
        float* const stW = stA + 1024;
        const int NB = K / 32;
        const int blo = (int)((int64_t)ks * NB / KS), bhi = (int)((int64_t)(ks + 1) * NB / KS);
        const int lr = lane >> 3, ls = lane & 7;
        const float* ga[4];
        const float* gw[4];
        unsigned wo[4], ro[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int r = 8 * c + lr;
            const int64_t mm = tile * 32 + r;
            ga[c] = A + (mm < M ? mm : (M - 1)) * lda + 4 * ls;
            gw[c] = W + (int64_t)r * ldw + 4 * ls;
            wo[c] = r * 32 + ((ls ^ ((r >> 1) & 7)) << 2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ro[q] = i * 32 + ((((2 * q + h) ^ ((i >> 1) & 7)) & 7) << 2);
        // one block ahead in registers (two ahead measured no better: 22.6 vs 21.6 us at 4096 x 2094 -> 64)
        f32x4 va[4], vw[4];
        if (bhi > blo) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                va[c] = *reinterpret_cast<const f32x4u*>(ga[c] + 32 * blo);
                vw[c] = *reinterpret_cast<const f32x4u*>(gw[c] + 32 * blo);
            }
        }
        for (int b = blo; b < bhi; ++b) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                *reinterpret_cast<f32x4*>(stA + wo[c]) = va[c];
                *reinterpret_cast<f32x4*>(stW + wo[c]) = vw[c];
            }
            const int bn = b + 1 < bhi ? b + 1 : b;         // the last iteration fetches its own block again (harmless, no branch)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                va[c] = *reinterpret_cast<const f32x4u*>(ga[c] + 32 * bn);
                vw[c] = *reinterpret_cast<const f32x4u*>(gw[c] + 32 * bn);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(stA + ro[q]);
                const f32x4 w = *reinterpret_cast<const f32x4*>(stW + ro[q]);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w[j], acc[0], 0, 0, 0);
            }
        }
      }
      if (qhi > qlo) {
        // A fragments run PD k-steps ahead of the MFMAs in a register ring: the A rows stream from HBM, and with the
        // one-step-ahead pipeline this kernel had, a CU kept 8 KB in flight — 2 MB chip-wide, i.e. 1.9 TB/s at ~1 us of
        // latency, which is exactly what it measured (595 us on 1.1 M x 128 x 128).  W fragments (L2 hits) stay one step
        // ahead.  Steps past qhi load zeros and add nothing; the k order per output is unchanged.
        constexpr int PD = NT >= 2 ? 8 : 16;
        f32x4 ar[PD];
        // no branch around a load (hipcc would wait vmcnt(0) per element): the address is clamped, the VALUE is selected
        const int qlast = qhi > qlo ? qhi - 1 : qlo;
        auto load_a = [&](int q) {
            const f32x4 v = *reinterpret_cast<const f32x4u*>(arow + 8 * (q < qhi ? q : qlast));
            const bool in = q < qhi;
            return f32x4{in ? v[0] : 0.f, in ? v[1] : 0.f, in ? v[2] : 0.f, in ? v[3] : 0.f};
        };
        auto load_w = [&](f32x4 (&w)[NT], int q) {      // a step past qhi multiplies this by A = 0
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) w[nt] = *reinterpret_cast<const f32x4u*>(wrow[nt] + 8 * (q < qhi ? q : qlast));
        };
#pragma unroll
        for (int d = 0; d < PD; ++d) ar[d] = load_a(qlo + d);
        f32x4 w0[NT], w1[NT];
        load_w(w0, qlo);
        for (int qb = qlo; qb < qhi; qb += PD) {
#pragma unroll
            for (int d = 0; d < PD; d += 2) {
                load_w(w1, qb + d + 1);
                {
                    const f32x4 a = ar[d];
                    ar[d] = load_a(qb + d + PD);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w0[nt][j], acc[nt], 0, 0, 0);
                }
                load_w(w0, qb + d + 2);
                {
                    const f32x4 a = ar[d + 1];
                    ar[d + 1] = load_a(qb + d + 1 + PD);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w1[nt][j], acc[nt], 0, 0, 0);
                }
            }
        }
      }
        if ((K & 7) && ks == TAILW) {  // ragged tail of K: guarded scalar loads, zero fill
            RsFrag<NT> t;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = 8 * Q + 4 * h + j < K;
                t.a[j] = ok ? arow[8 * Q + j] : 0.f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) t.w[nt][j] = ok ? wrow[nt][8 * Q + j] : 0.f;
            }
            rs_mma<NT>(t, acc);
        }
    }
    if (KS > 1) {
        if constexpr (KS >= 8) __syncthreads();             // every wave is done with its staging area (the sums reuse the array)
        float* mine = red + (size_t)wave * (NT * 16 * 64);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(nt * 16 + r) * 64 + lane] = acc[nt][r];
        __syncthreads();
        if (ks != 0) return;
#pragma unroll
        for (int s = 1; s < KS; ++s) {
            const float* other = red + (size_t)(wave + s) * (NT * 16 * 64);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] += other[(nt * 16 + r) * 64 + lane];
        }
    }
    if (!tile_ok) return;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = 32 * nt + i;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t gm = tile * 32 + acc_row(r, h);
            if (gm < M) {
                float v = acc[nt][r] + bv;
                if (RELU) v = fmaxf(v, 0.f);
                C[gm * ldc + n] = v;
            }
        }
    }
}

// Persistent form of linear_rs_kernel<NT, 1> for tall problems (K % 64 == 0): a wave walks over its row tiles and the
// A-fragment ring runs PD = 8 k-steps ahead ACROSS tile boundaries, so the next tile's first fragments are in flight
// while the current tile finishes and stores.  In the one-tile-per-wave form every tile paid workgroup dispatch, pointer
// set-up and a cold first fetch (~2 us exposed against ~8 us of MFMAs, measured 571 us on 1.1 M x 128 x 128 where the
// MFMAs alone are 229 us and HBM 225 us); neither deeper prefetch inside a tile (571 vs 586 us), nor staging A or A and W
// through LDS (654 / 702 us) moved it — see tools/ab_linear.py.  Same k order per output: bit-identical results.
#ifndef NCF_RSP_ABLATE
#define NCF_RSP_ABLATE 0   // diagnostics (tools/ab_linear_variants.py): 1 = no C stores, 2 = W fragments loaded once per tile, 3 = both
#endif
template <int NT, bool RELU>
__global__ __launch_bounds__(256) void linear_rsp_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ W,
                                                         int64_t ldw, const float* __restrict__ bias, float* __restrict__ C,
                                                         int64_t ldc, int64_t M, int K) {
    constexpr int PD = 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int64_t tiles = (M + 31) / 32;
    const int64_t first = (int64_t)blockIdx.x * 4 + wave, stride = (int64_t)gridDim.x * 4;
    if (first >= tiles) return;
    const int64_t mine = (tiles - first + stride - 1) / stride;     // row tiles of this wave
    const int Q = K / 8;                                            // multiple of PD (host check)
    const int64_t S = mine * Q;                                     // k-steps of this wave's stream
    const float* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = W + (int64_t)(32 * nt + i) * ldw + 4 * h;
    float bv[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bv[nt] = bias ? bias[32 * nt + i] : 0.f;

    // stream position p -> (tile first + (p / Q) * stride, k-step p % Q); past the end: the last step again (its MFMAs
    // never run: the loops below stop at S)
    auto a_at = [&](int64_t p) {
        if (p >= S) p = S - 1;
        const int64_t t = first + (p / Q) * stride;
        const int64_t m = t * 32 + i;
        return *reinterpret_cast<const f32x4u*>(A + (m < M ? m : (M - 1)) * lda + 4 * h + 8 * (p % Q));
    };
    f32x4 ar[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) ar[d] = a_at(d);

    for (int64_t tk = 0; tk < mine; ++tk) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        f32x4 w0[NT], w1[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w0[nt] = *reinterpret_cast<const f32x4u*>(wrow[nt]);
        const int64_t p0 = tk * Q;
        for (int qb = 0; qb < Q; qb += PD) {
#pragma unroll
            for (int d = 0; d < PD; d += 2) {
                const int q = qb + d;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (!(NCF_RSP_ABLATE & 2)) w1[nt] = *reinterpret_cast<const f32x4u*>(wrow[nt] + 8 * (q + 1));   // q + 1 < Q: Q is even
                    else w1[nt] = w0[nt];
                {
                    const f32x4 a = ar[d];
                    ar[d] = a_at(p0 + q + PD);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w0[nt][j], acc[nt], 0, 0, 0);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    if (!(NCF_RSP_ABLATE & 2)) w0[nt] = *reinterpret_cast<const f32x4u*>(wrow[nt] + 8 * (q + 2 < Q ? q + 2 : 0));
                {
                    const f32x4 a = ar[d + 1];
                    ar[d + 1] = a_at(p0 + q + 1 + PD);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w1[nt][j], acc[nt], 0, 0, 0);
                }
            }
        }
        const int64_t tile = first + tk * stride;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = 32 * nt + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gm = tile * 32 + acc_row(r, h);
                if (gm < M) {
                    float v = acc[nt][r] + bv[nt];
                    if (RELU) v = fmaxf(v, 0.f);
                    if (!(NCF_RSP_ABLATE & 1) || v == 12345.678f) C[gm * ldc + n] = v;
                }
            }
        }
    }
}

template <int NT>
static void launch_rsp(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                       int64_t M, int K, bool relu, hipStream_t s) {
    const int64_t tiles = (M + 31) / 32;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 512) blocks = 512;                         // two 4-wave workgroups per CU, each wave loops over its tiles
    if (relu) hipLaunchKernelGGL((linear_rsp_kernel<NT, true>), dim3((unsigned)blocks), dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, K);
    else hipLaunchKernelGGL((linear_rsp_kernel<NT, false>), dim3((unsigned)blocks), dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, K);
}

template <int NT, int KS>
static void launch_rs(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                      int64_t M, int K, bool relu, hipStream_t s, int col_blocks = 1) {
    const int64_t tiles = (M + 31) / 32;
    constexpr int WAVES = KS > 4 ? KS : 4, TPB = WAVES / KS;
    const dim3 grid((unsigned)((tiles + TPB - 1) / TPB), (unsigned)col_blocks);
    if (relu) hipLaunchKernelGGL((linear_rs_kernel<NT, KS, true>), grid, dim3(WAVES * 64), 0, s, A, lda, W, ldw, bias, C, ldc, M, K);
    else hipLaunchKernelGGL((linear_rs_kernel<NT, KS, false>), grid, dim3(WAVES * 64), 0, s, A, lda, W, ldw, bias, C, ldc, M, K);
}

template <int NT>
static void launch_rs_nt(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                         int64_t M, int K, bool relu, hipStream_t s) {
    const int64_t tiles = (M + 31) / 32;
    const int force = option(NCF_OPT_LINEAR_KERNEL);      // 1 = rs / 2 = rsp: A/B and tests (ncf_set_option); 0: by shape
    // persistent row-streaming form: tall problems with K a multiple of 64 (bit-identical to the one-tile-per-wave form)
    // Measured against the one-tile-per-wave form over the shapes of the hot path (tools/ab_linear.py, rs -> rsp):
    // 1.1 M x 128 x 128 584 -> 488 us; 200 000 x 128 x 128 107 -> 95; 65 536 x 128 -> 256 83 -> 59, x 256 -> 128 74 -> 54,
    // x 64 -> 128 48 -> 23; 32 768 x 128 x 128 67.5 -> 17.5 (the split-K form is slow at short K), x 256 -> 128 70 -> 28,
    // x 128 -> 256 42 -> 30; 16 384 x 256 -> 128 35 -> 27; 4096..16 384 x 128 -> 256 37-39 -> 26-28.  The split-K form keeps
    // deep-and-short problems (8192 x 256 -> 128: 18.7 vs 26.0 us; the 4096 x 2094 candidate Linear).
    // Short K (<= 128) from 128 tiles: 4096 x 64 -> 128 18.3 -> 14.4 us, -> 256 28.1 -> 17.7, 16 384 x 64 -> 128 35.0 -> 14.4
    // (splitting a K of 8 or 16 steps over 4 waves only adds the LDS reduction).
    if (K % 64 == 0 && (force ? force == 2 : (tiles >= 512 || (K <= 128 && tiles >= 128))))
        return launch_rsp<NT>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
    if constexpr (NT <= 4) {
        // Skinny and deep (the 4096 x 2094 -> 64 candidate Linear of AttentionNCF: 128 row tiles): one 32-column block
        // per workgroup so that tiles * NT workgroups share the chip — the rows of A are re-read once per column block
        // (from L2), each wave's MFMA chain is NT times shorter; same split-K order, bit-identical results.
        if (NT > 1 && tiles * NT <= 512 && K >= 512) {
            // ... and with at most one such workgroup per CU, 8 K-slices (512 threads, two waves per SIMD) instead of 4:
            // 4096 x 2094 -> 64: 36.5 -> 29.9 us, 2048 rows 27.1 -> 26.0; NOT beyond 256 workgroups (8192 rows: 48.8 -> 61.0,
            // N = 128: 47.2 -> 55.5) and 16 slices lose everywhere (tools/ab_linear_env.py, ncf_set_option("linear_kslices", 4|8) forces one)
            const int ksf = option(NCF_OPT_LINEAR_KSLICES);
            // (staged A / W blocks since round 2: 4096 x 2094 -> 64 20.8 us vs 36.4 on 4 slices, 8192 rows 37.7 vs 48.1, N = 128 36.3 vs
            // 46.4, 4096 x 1030 -> 64 13.4 vs 18.1: tools/ab_skinny_linear.py)
            const bool ks8 = ksf ? ksf == 8 : K >= 1024;
            if (ks8) return launch_rs<1, 8>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s, NT);
            return launch_rs<1, 4>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s, NT);
        }
        if (tiles <= 1024 && K >= 64) return launch_rs<NT, 4>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
        if (tiles <= 2048 && K >= 32) return launch_rs<NT, 2>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
    }
    launch_rs<NT, 1>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
}

static int launch_linear(const float* A, int64_t lda, const float* W, int64_t ldw, const float* bias, float* C, int64_t ldc,
                         int64_t M, int N, int K, bool relu, hipStream_t s) {
    if (M == 0) return NCF_OK;
    if (K >= 8 && (N == 32 || N == 64 || N == 128 || N == 256)) {
        if (N == 32) launch_rs_nt<1>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
        else if (N == 64) launch_rs_nt<2>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
        else if (N == 128) launch_rs_nt<4>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
        else launch_rs_nt<8>(A, lda, W, ldw, bias, C, ldc, M, K, relu, s);
        return check_launch("linear_rs_f32");
    }
    if (N <= 8) {
        int64_t blocks = (M * N + 15) / 16;
        if (blocks > 16384) blocks = 16384;
        if (relu) hipLaunchKernelGGL(rowdot_f32_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K);
        else hipLaunchKernelGGL(rowdot_f32_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K);
    } else {
        dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((N + BN - 1) / BN));
        if (relu) hipLaunchKernelGGL(linear_f32_kernel<true>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K);
        else hipLaunchKernelGGL(linear_f32_kernel<false>, grid, dim3(256), 0, s, A, lda, W, ldw, bias, C, ldc, M, N, K);
    }
    return check_launch("linear_f32");
}

}  // namespace ncf

using namespace ncf;

static int max_hidden(int n_layers, const int* dims) {
    int mx = 0;
    for (int i = 1; i < n_layers; ++i) mx = dims[i] > mx ? dims[i] : mx;
    return mx;
}

extern "C" size_t ncf_mlp_workspace_bytes(int dtype, int64_t B, int n_layers, const int* dims) {
    if (dtype != NCF_F32 || !dims || n_layers < 1 || B < 0) return 0;
    if (n_layers == 1) return 0;
    const size_t per = (size_t)B * (size_t)max_hidden(n_layers, dims) * sizeof(float);
    const size_t per_al = (per + 255) & ~size_t(255);
    return n_layers == 2 ? per_al : 2 * per_al;
}

extern "C" int ncf_mlp_forward(int dtype, const void* x, int64_t B, int64_t ldx, int n_layers, const int* dims,
                               const void* const* W, const void* const* b, void* workspace, size_t ws_bytes, void* out,
                               int64_t ldOut, ncf_stream_t stream) {
    if (dtype != NCF_F32) return fail(NCF_EUNSUPPORTED, "ncf_mlp_forward: fp32 only on the generic path");
    if (B == 0) return NCF_OK;
    if (!x || !dims || !W || !out || n_layers < 1 || B < 0) return fail(NCF_EINVAL, "ncf_mlp_forward: bad argument");
    for (int i = 0; i <= n_layers; ++i)
        if (dims[i] <= 0) return fail(NCF_EINVAL, "ncf_mlp_forward: dims[%d] = %d", i, dims[i]);
    if (ldx < dims[0] || ldOut < dims[n_layers]) return fail(NCF_EINVAL, "ncf_mlp_forward: leading dimension smaller than row");
    const size_t need = ncf_mlp_workspace_bytes(dtype, B, n_layers, dims);
    if (need > 0 && (!workspace || ws_bytes < need)) return fail(NCF_EWORKSPACE, "ncf_mlp_forward: workspace %zu < %zu bytes", ws_bytes, need);
    if (B == 0) return NCF_OK;
    hipStream_t s = (hipStream_t)stream;
    const int mh = max_hidden(n_layers, dims);
    const size_t per_al = (((size_t)B * mh * sizeof(float)) + 255) & ~size_t(255);
    float* buf[2] = {(float*)workspace, (float*)((char*)workspace + per_al)};
    const float* in = (const float*)x;
    int64_t ldin = ldx;
    for (int i = 0; i < n_layers; ++i) {
        if (!W[i]) return fail(NCF_EINVAL, "ncf_mlp_forward: W[%d] is null", i);
        const bool last = i == n_layers - 1;
        float* o = last ? (float*)out : buf[i & 1];
        const int64_t ldo = last ? ldOut : dims[i + 1];
        const int rc = launch_linear(in, ldin, (const float*)W[i], dims[i], b ? (const float*)b[i] : nullptr, o, ldo, B,
                                     dims[i + 1], dims[i], !last, s);
        if (rc != NCF_OK) return rc;
        in = o;
        ldin = ldo;
    }
    return NCF_OK;
}

extern "C" int ncf_linear_forward(int dtype, const void* x, int64_t M, int64_t ldx, const void* W, const void* b, int K, int N,
                                  int relu, void* out, int64_t ldo, ncf_stream_t stream) {
    if (dtype != NCF_F32) return fail(NCF_EUNSUPPORTED, "ncf_linear_forward: fp32 only");
    if (M == 0) return NCF_OK;
    if (M < 0 || K <= 0 || N <= 0 || !x || !W || !out || ldx < K || ldo < N) return fail(NCF_EINVAL, "ncf_linear_forward: bad argument");
    return launch_linear((const float*)x, ldx, (const float*)W, K, (const float*)b, (float*)out, ldo, M, N, K, relu != 0, (hipStream_t)stream);
}
