// K3 tail — the end of an AttentionNCF forward in ONE launch (models/attention_ncf.py:208-222), fp32, gfx950:
//   user_emb = merge of the entry-split attention's softmax partials (attn_split.hip)  [+ UserEmbeddings bias]      (:208-216)
//   out      = MLP(cat(candidate_emb, user_emb))      Linear, ReLU, Linear, ReLU, Linear -> 1                        (:219-222, util.py:5-18)
// Round 3's first form ran attn_combine_kernel (3 us + a launch boundary) and then the small fused scoring kernel, whose 32-pair
// tiles give a 4096-pair batch 128 workgroups of one wave per SIMD: 12.4 us for 0.5 GFLOP.  Here a workgroup takes 16 pairs (256
// workgroups for 4096 pairs: every CU), merges their partials straight into the concatenated input tile in LDS, and runs the three
// layers on v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chains) with 8 waves sharing the tile: layer ℓ's 16-neuron column tiles are
// dealt over the waves, activations live in LDS ([16][width + 4]: bank spread), weights stream from L2 in plain row-major layout
// (lane (i, g) reads W[16 ct + i][16 kk + 4 g ..+3]: the k pairing of a 16-wide block is (4 g + j) for MFMA step j, the same for the
// activation operand, so only the summation order inside a block is permuted).  Weight fragments run two k-blocks ahead in registers
// (measured against requesting ALL of a layer's fragments before its first MFMA: 13.4 vs 16.3 us at 4096 pairs — every workgroup reads
// the same 256 KB of weights, and 256 workgroups asking for all of it at once queue on their XCD's L2).
// No bit-identity with mlp_fused.hip's k order is claimed (another summation order; 1e-5 of the oracle holds).
#include "ncf_common.h"
#include "attn_util.h"
#include <type_traits>

#ifndef ATT_TAIL_DIAG
#define ATT_TAIL_DIAG 0   // diagnostic builds (wrong results): 1 = no merge (partials not read), 2 = no layers 1 / 2, 3 = exit at once,
#endif                    // 4 = layers without their weight loads (one fragment reused)

namespace ncf {

struct TailArgs {
    const float* cand; int64_t ldcand;          // (B, EA) candidate embeddings
    const float* part; int nsplit, ldpart;      // (B, nsplit, UE + 4) softmax partials  (nsplit >= 1)
    const float* user; int64_t lduser;          // OR (B, UE) finished user embeddings (part == nullptr)
    const float* ubias;                         // UserEmbeddings bias added to the merged rows (may be null)
    const float* W1; const float* b1; const float* W2; const float* b2; const float* w3; float b3;
    float* out;
    int64_t B;
};

constexpr int kTailHead = 4;   // floats in front of a partial's O (attn_split.hip kPartHead)

// PACKED: W1 / W2 in MFMA operand order (ncf_attn_tail_pack_weight: [k-block][column tile][lane][4]) — a wave's load of a weight
// fragment is then ONE contiguous 1 KB run instead of 16 rows x 64 bytes.  Measured at the cfg 3 shape: 12.8 -> 9.8 us.  The same
// values reach the same MFMAs in the same order: bit-identical.
template <int EA, int UE, int N1, int N2, bool PACKED>
__global__ __launch_bounds__(512, 2) void attn_tail_kernel(const TailArgs a) {
    constexpr int K0 = EA + UE, TM = 16, NWV = 8;
    constexpr int XS = K0 + 4, H1S = N1 + 4, H2S = N2 + 4;
    __shared__ __attribute__((aligned(16))) float xs[TM * XS];
    __shared__ __attribute__((aligned(16))) float h1[TM * H1S];
    __shared__ __attribute__((aligned(16))) float h2[TM * H2S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * TM;
    if (ATT_TAIL_DIAG == 3) { if (tid == 0) a.out[row0] = 0.f; return; }

    // ---- the input tile: cat(candidate_emb, user_emb) (:219, candidate first) ----
    for (int o = tid; o < TM * (K0 / 4); o += 512) {
        const int r = o / (K0 / 4), c = o - r * (K0 / 4);
        const int64_t b = row0 + r < a.B ? row0 + r : a.B - 1;
        f32x4 v;
        if (c < EA / 4) {
            v = *reinterpret_cast<const f32x4*>(a.cand + b * a.ldcand + 4 * c);
        } else {
            const int cu = c - EA / 4;
            if (a.part && ATT_TAIL_DIAG != 1) {   // out = sum_s O_s e^(m_s - M) / sum_s l_s e^(m_s - M) + bias, slices in index order (attn_combine_kernel's order)
                const float* p0 = a.part + b * a.nsplit * (int64_t)a.ldpart;
                float M = -INFINITY;
                float L = 0.f;
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                if (a.nsplit <= 8) {                       // the usual case: every slice's (m, l) and O chunk requested at once — a loop over
                    const int ns = a.nsplit;               // a run-time count is a chain of dependent round trips
                    f32x2 ml[8];
                    f32x4 ov[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        const float* ps = p0 + (int64_t)(s < ns ? s : ns - 1) * a.ldpart;
                        ml[s] = *reinterpret_cast<const f32x2*>(ps);
                        ov[s] = *reinterpret_cast<const f32x4*>(ps + kTailHead + 4 * cu);
                    }
#pragma unroll
                    for (int s = 0; s < 8; ++s) M = s < ns ? fmaxf(M, ml[s][0]) : M;
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        if (s < ns) {
                            const float w = (ml[s][0] == -INFINITY) ? 0.f : exp_le0(ml[s][0] - M);
                            L += ml[s][1] * w;
                            acc += ov[s] * w;
                        }
                    }
                } else {
                    for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, p0[(int64_t)s * a.ldpart]);
                    for (int s = 0; s < a.nsplit; ++s) {
                        const float* ps = p0 + (int64_t)s * a.ldpart;
                        const float ms = ps[0];
                        const float w = (ms == -INFINITY) ? 0.f : exp_le0(ms - M);
                        L += ps[1] * w;
                        acc += *reinterpret_cast<const f32x4*>(ps + kTailHead + 4 * cu) * w;
                    }
                }
                const float inv = L > 0.f ? 1.0f / L : 0.f;
                f32x4 bias = {0.f, 0.f, 0.f, 0.f};
                if (a.ubias) bias = *reinterpret_cast<const f32x4*>(a.ubias + 4 * cu);
                v = acc * inv + bias;
            } else if (ATT_TAIL_DIAG == 1) {
                v = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
                v = *reinterpret_cast<const f32x4*>(a.user + b * a.lduser + 4 * cu);
            }
        }
        *reinterpret_cast<f32x4*>(xs + r * XS + 4 * c) = v;
    }
    __syncthreads();

    // one layer: OUT[pair][n] = act(bias[n] + sum_k IN[pair][k] W[n][k]); the wave's column tiles ct = wave + 8 t run SIDE BY SIDE (T independent
    // MFMA chains, their weight fragments in flight together), weight fragments PD k-blocks ahead in a register ring
    auto layer = [&](auto tiles, const float* in, int ins, int K, const float* __restrict__ W, const float* __restrict__ bias, float* outp, int outs) {
        constexpr int T = decltype(tiles)::value, PD = 4;
        constexpr int NA = T == 1 ? 2 : 1;                     // a lone tile runs as two chains (even / odd k-blocks), added at the end: a
        const int KB = K / 16;                                 // dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles, the pipe takes one per 32
        const int WSTEP = PACKED ? T * NWV * 256 : 16;         // floats between a lane's fragments of consecutive k-blocks
        f32x4 acc[T][NA], ring[PD][T], aring[PD];
        const float* wrow[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int c = 0; c < NA; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            wrow[t] = PACKED ? W + ((int64_t)(wave + NWV * t) * 64 + lane) * 4
                             : W + (int64_t)(16 * (wave + NWV * t) + i16) * K + 4 * g4;
        }
        const float* irow = in + i16 * ins + 4 * g4;
#pragma unroll
        for (int d = 0; d < PD; ++d) {
            const int kk = d < KB ? d : KB - 1;
            aring[d] = *reinterpret_cast<const f32x4*>(irow + 16 * kk);      // the activation fragments ride the same ring (LDS latency too)
#pragma unroll
            for (int t = 0; t < T; ++t) ring[d][t] = *reinterpret_cast<const f32x4*>(wrow[t] + WSTEP * kk);
        }
        for (int kb = 0; kb < KB; kb += PD) {
#pragma unroll
            for (int d = 0; d < PD; ++d) {
                const int kk = kb + d;
                if (kk < KB) {                                 // wave-uniform
                    f32x4 wv[T];
                    const f32x4 av = aring[d];
                    const int kn = kk + PD < KB ? kk + PD : KB - 1;
                    aring[d] = *reinterpret_cast<const f32x4*>(irow + 16 * kn);
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        wv[t] = ring[d][t];
                        if (ATT_TAIL_DIAG != 4) ring[d][t] = *reinterpret_cast<const f32x4*>(wrow[t] + WSTEP * kn);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int t = 0; t < T; ++t)
                            acc[t][d % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wv[t][j], acc[t][d % NA], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int n = 16 * (wave + NWV * t) + i16;
            const float bv = bias[n];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = NA == 2 ? acc[t][0][i] + acc[t][NA - 1][i] : acc[t][0][i];
                outp[(4 * g4 + i) * outs + n] = fmaxf(v + bv, 0.f);   // ReLU (util.py:12-14)
            }
        }
    };
    static_assert(N1 % (16 * NWV) == 0 && N2 % (16 * NWV) == 0, "column tiles are dealt evenly over the waves");
    if (ATT_TAIL_DIAG != 2) layer(std::integral_constant<int, N1 / 16 / NWV>{}, xs, XS, K0, a.W1, a.b1, h1, H1S);
    __syncthreads();
    if (ATT_TAIL_DIAG != 2) layer(std::integral_constant<int, N2 / 16 / NWV>{}, h1, H1S, N1, a.W2, a.b2, h2, H2S);
    __syncthreads();
    // last layer (N2 -> 1, no activation): wave 0, lane (pair, quarter): a quarter of the dot each, quarters added in order
    if (wave == 0) {
        const float* hrow = h2 + i16 * H2S + g4 * (N2 / 4);
        const float* wq = a.w3 + g4 * (N2 / 4);
        f32x4 wv[N2 / 16];
#pragma unroll
        for (int c = 0; c < N2 / 16; ++c) wv[c] = *reinterpret_cast<const f32x4*>(wq + 4 * c);   // the quarter's weights at once
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < N2 / 16; ++c) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(hrow + 4 * c);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(hv[j], wv[c][j], s);                           // the same k order as a plain loop
        }
        const float s1 = __shfl(s, i16 + 16), s2 = __shfl(s, i16 + 32), s3 = __shfl(s, i16 + 48);
        if (g4 == 0 && row0 + i16 < a.B) a.out[row0 + i16] = ((s + s1) + (s2 + s3)) + a.b3;
    }
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_attn_tail_supported(int EA, int UE, int N1, int N2) {
    return ((EA == 64 && UE == 64) || (EA == 128 && UE == 128)) && N1 == 256 && N2 == 128;
}

namespace ncf {
// packed[((kb * (N / 16) + tile) * 64 + lane) * 4 + j] = W[16 tile + (lane & 15)][16 kb + 4 (lane >> 4) + j]
__global__ __launch_bounds__(256) void tail_pack_kernel(const float* __restrict__ W, int N, int K, float* __restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)N * K) return;
    const int j = (int)(idx & 3), lane = (int)((idx >> 2) & 63);
    const int64_t q = idx >> 8;
    const int tiles = N / 16, tile = (int)(q % tiles), kb = (int)(q / tiles);
    packed[idx] = W[(int64_t)(16 * tile + (lane & 15)) * K + 16 * kb + 4 * (lane >> 4) + j];
}
}  // namespace ncf

/* A hidden layer's (N, K) row-major weight in the operand order ncf_attn_tail(weights_packed = 1) reads (N, K multiples of 16; N * K floats) */
extern "C" int ncf_attn_tail_pack_weight(const float* W, int N, int K, float* packed, ncf_stream_t stream) {
    if (N < 16 || K < 16 || N % 16 || K % 16) return fail(NCF_EINVAL, "ncf_attn_tail_pack_weight: N and K must be multiples of 16");
    if (!W || !packed || !aligned16(packed)) return fail(NCF_EINVAL, "ncf_attn_tail_pack_weight: null or misaligned pointer");
    const int64_t n = (int64_t)N * K;
    hipLaunchKernelGGL(tail_pack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, N, K, packed);
    return check_launch("ncf_attn_tail_pack_weight");
}

extern "C" int ncf_attn_tail(const float* cand, int64_t ldcand, int EA, const float* part, int nsplit, const float* user, int64_t lduser,
                             int UE, const float* ubias, const float* W1, const float* b1, int N1, const float* W2, const float* b2, int N2,
                             const float* w3, float b3, int weights_packed, float* out, int64_t B, ncf_stream_t stream) {
    if (!ncf_attn_tail_supported(EA, UE, N1, N2))
        return fail(NCF_EUNSUPPORTED, "ncf_attn_tail: takes item_emb = user_emb in {64, 128} and MLP [256, 128] (EA = %d, UE = %d, MLP [%d, %d])", EA, UE, N1, N2);
    if (B < 0 || ldcand < EA || (!part && lduser < UE) || (part && (nsplit < 1 || nsplit > 64))) return fail(NCF_EINVAL, "ncf_attn_tail: bad sizes");
    if (B == 0) return NCF_OK;
    if (!cand || (!part && !user) || !W1 || !b1 || !W2 || !b2 || !w3 || !out) return fail(NCF_EINVAL, "ncf_attn_tail: null pointer");
    if (ldcand % 4 || (!part && lduser % 4) || !aligned16(cand) || (part && !aligned16(part)) || (user && !aligned16(user)) || !aligned16(W1) ||
        !aligned16(W2) || (ubias && !aligned16(ubias)))
        return fail(NCF_EINVAL, "ncf_attn_tail: operands must be 16-byte aligned with leading dimensions that are multiples of 4");
    TailArgs a{};
    a.cand = cand; a.ldcand = ldcand; a.part = part; a.nsplit = nsplit; a.ldpart = UE + kTailHead; a.user = user; a.lduser = lduser;
    a.ubias = ubias; a.W1 = W1; a.b1 = b1; a.W2 = W2; a.b2 = b2; a.w3 = w3; a.b3 = b3; a.out = out; a.B = B;
    hipStream_t s = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((B + 15) / 16);
    if (weights_packed) {
        if (EA == 64) hipLaunchKernelGGL((attn_tail_kernel<64, 64, 256, 128, true>), dim3(blocks), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attn_tail_kernel<128, 128, 256, 128, true>), dim3(blocks), dim3(512), 0, s, a);
    } else {
        if (EA == 64) hipLaunchKernelGGL((attn_tail_kernel<64, 64, 256, 128, false>), dim3(blocks), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attn_tail_kernel<128, 128, 256, 128, false>), dim3(blocks), dim3(512), 0, s, a);
    }
    return check_launch("ncf_attn_tail");
}
