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
#include "ncf_common.h"

#ifndef NCF_BF16_X_DEPTH
#define NCF_BF16_X_DEPTH 4   // gathered-row prefetch ring, in 16-wide k-steps (measured: 4 -> 20.6 us, 8 -> 21.3, 16 -> 21.7)
#endif
#ifndef NCF_BF16_MSTEP
#define NCF_BF16_MSTEP 4     // 16-wide k-steps per weight slab (= per barrier); measured 2 -> 20.8 us, 4 -> 19.3 us
#endif
#ifndef NCF_BF16_WGW
#define NCF_BF16_WGW 8       // waves per workgroup sharing a weight slab; measured: 8 -> 19.6 us, 4 (two WGs per CU) -> 20.4, 2 -> 24-28
#endif
#ifndef NCF_BF16_STAMP
#define NCF_BF16_STAMP 0     // diagnostic builds only (tools/ab_bf16.py): phase stamps written behind the outputs
#endif
#ifndef NCF_BF16_ABLATE
#define NCF_BF16_ABLATE 0    // diagnostics: 1 = no gathered-row loads, 2 = no weight slab copies / barriers
#endif

namespace ncf {

#if NCF_BF16_STAMP
static unsigned long long* g_bf16_dbg = nullptr;
#define BF16_STAMP(i) do { if (a.dbg && lane == 0) { a.dbg[tile * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define BF16_STAMP(i) do { } while (0)
#endif

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Bf16Args {
    const unsigned short* tabA; int64_t rowsA; int64_t ldA;
    const unsigned short* tabB; int64_t rowsB; int64_t ldB;
    const int64_t* idxA; const int64_t* idxB;
    int64_t B; int EA;
    const unsigned short* Wp1; const float* b1;
    const unsigned short* Wp2; const float* b2;
    const float* wl; const float* bl;
    float* out; int32_t* oob;
#if NCF_BF16_STAMP
    unsigned long long* dbg;
#endif
};

__device__ __forceinline__ u32x4 ldg16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }

__device__ __forceinline__ bf16x8_t as_bf16x8(u32x4 v) {
    union { u32x4 u; bf16x8_t b; } c;
    c.u = v;
    return c.b;
}

// relu + round-to-nearest-even to bf16 of 8 accumulator registers -> one MFMA B fragment
__device__ __forceinline__ bf16x8_t pack_relu8(const f32x16& acc, int base) {
    bf16x8_t r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        f32x2 v = {fmaxf(acc[base + j], 0.f), fmaxf(acc[base + j + 1], 0.f)};
        bf16x2_t p = __builtin_convertvector(v, bf16x2_t);
        r[j] = p[0];
        r[j + 1] = p[1];
    }
    return r;
}

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

// Wp[q][nt][lane][8] (bf16, RNE from fp32) with
//   natural order   (layer 1): element j = W[32nt + (lane&31)][16q + 8(lane>>5) + j]
//   permuted order  (layer 2): element j = W[32nt + (lane&31)][32(q>>1) + 16(q&1) + 8(j>>2) + 4(lane>>5) + (j&3)]
__global__ void pack_weight_bf16_kernel(const float* __restrict__ W, int N, int K, int permuted, unsigned short* __restrict__ Wp) {
    const int64_t total = (int64_t)N * K;
    const int NT = N / 32;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int j = (int)(o & 7);
        const int lane = (int)((o >> 3) & 63);
        const int64_t t = o >> 9;  // q*NT + nt
        const int nt = (int)(t % NT), q = (int)(t / NT);
        const int h = lane >> 5;
        const int k = permuted ? 32 * (q >> 1) + 16 * (q & 1) + 8 * (j >> 2) + 4 * h + (j & 3) : 16 * q + 8 * h + j;
        const float v = W[(int64_t)(32 * nt + (lane & 31)) * K + k];
        const __bf16 b = (__bf16)v;  // round-to-nearest-even (v_cvt_pk_bf16_f32)
        Wp[o] = *reinterpret_cast<const unsigned short*>(&b);
    }
}

__global__ void copy_or_zero_f32_kernel(const float* __restrict__ src, int n, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : 0.f;
}

struct Bf16Blob {
    size_t wp1, b1, wp2, b2, wl, bl, total;  // byte offsets
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
    L.total = off;
    return L;
}

template <int K0, int N1, int N2>
static void launch_bf16(const Bf16Args& a, hipStream_t s) {
    const int64_t tiles = (a.B + 31) / 32;
    hipLaunchKernelGGL((score_fused_bf16_kernel<K0, N1, N2>), dim3((unsigned)((tiles + NCF_BF16_WGW - 1) / NCF_BF16_WGW)), dim3(NCF_BF16_WGW * 64), 0, s, a);
}

#define NCF_BF16_INSTANCES(X) X(256, 256, 128) X(256, 256, 0) X(128, 256, 128) X(128, 256, 0)

static bool bf16_dispatch(int K0, int N1, int N2, const Bf16Args* a, hipStream_t s) {
#define X(k, n1, n2) \
    if (K0 == k && N1 == n1 && N2 == n2) { if (a) launch_bf16<k, n1, n2>(*a, s); return true; }
    NCF_BF16_INSTANCES(X)
#undef X
    return false;
}

bool bf16_shape_ok(int EA, int EB, int n_layers, const int* dims) {
    if (!dims || (n_layers != 2 && n_layers != 3) || dims[n_layers] != 1) return false;
    if (EA <= 0 || EB < 0 || EA % 16 || EB % 16 || EA + EB != dims[0]) return false;
    return bf16_dispatch(dims[0], dims[1], n_layers == 3 ? dims[2] : 0, nullptr, nullptr);
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
        hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3((dims[2] + 255) / 256), dim3(256), 0, s, bias(1), dims[2], (float*)(P + L.b2));
        last = dims[2];
    }
    hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3((last + 255) / 256), dim3(256), 0, s, (const float*)W[n_layers - 1], last, (float*)(P + L.wl));
    hipLaunchKernelGGL(copy_or_zero_f32_kernel, dim3(1), dim3(256), 0, s, bias(n_layers - 1), 1, (float*)(P + L.bl));
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
    a.wl = (const float*)(P + L.wl); a.bl = (const float*)(P + L.bl);
    a.out = out; a.oob = oob;
#if NCF_BF16_STAMP
    a.dbg = g_bf16_dbg;
#endif
    bf16_dispatch(dims[0], dims[1], n_layers == 3 ? dims[2] : 0, &a, s);
    return check_launch("ncf_score_fused(bf16)");
}

}  // namespace ncf

#if NCF_BF16_STAMP
extern "C" void ncf_dev_set_bf16_debug_buffer(void* p) { ncf::g_bf16_dbg = (unsigned long long*)p; }
#endif
