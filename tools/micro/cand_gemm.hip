// Skinny-GEMM micro-benchmark (gfx950): emb = x . Wi^T + bi, x (B, K) fp32 row-major with rows only 8-byte aligned (K = 2094),
// Wi (N1, K), N1 = 64 — the big product of attn_cand.hip.  Variants of HOW the operands reach v_mfma_f32_16x16x4_f32:
//   direct<NWV, D, XM>   W PRE-PACKED in operand order (every load instruction = one contiguous 1 KB run, straight into the MFMA's
//                        registers: no LDS round trip for 80 % of the operand bytes); x either loaded directly in operand order
//                        (XM = 0: a load instruction touches 16 rows x 64 B) or staged through a per-wave LDS area from coalesced
//                        loads (XM = 1).  NWV waves = K slices, D steps (32 k each) in flight per wave.
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/cand_gemm.hip -o tools/bin/cand_gemm
// Run (GPU box): tools/bin/cand_gemm [B] [K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// Wp[(((s * NT + nt) * 2 + h) * 64 + lane) * 4 + j] = W[16 nt + (lane & 15)][32 s + 16 h + 4 (lane >> 4) + j]   (0 past K)
__global__ void pack_w_kernel(const float* W, int64_t ldw, int K, int N1, int S, float* Wp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int NT = N1 / 16;
    if (idx >= (int64_t)S * NT * 512) return;
    const int j = idx & 3, lane = (idx >> 2) & 63, h = (idx >> 8) & 1;
    const int64_t q = idx >> 9;
    const int nt = (int)(q % NT), s = (int)(q / NT);
    const int n = 16 * nt + (lane & 15), k = 32 * s + 16 * h + 4 * (lane >> 4) + j;
    Wp[idx] = k < K ? W[(int64_t)n * ldw + k] : 0.f;
}

template <int N1, int NWV, int D, int XM, int DIAG>
__global__ __launch_bounds__(64 * NWV) void cand_direct_kernel(const float* __restrict__ x, const float* __restrict__ Wp, const float* __restrict__ bi,
                                                               float* __restrict__ emb, int64_t B, int64_t ldx, int K, int S) {
    constexpr int NT = N1 / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, g4 = lane >> 4;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    const int64_t m = row0 + i16 < B ? row0 + i16 : B - 1;
    const float* xa = x + m * ldx + 4 * g4;
    const f32x4* wa = reinterpret_cast<const f32x4*>(Wp) + lane;
    const int s_lo = (int)((int64_t)wave * S / NWV), s_hi = (int)((int64_t)(wave + 1) * S / NWV);
    const int SF = K / 32;                                     // steps without a ragged end
    const int e = s_hi < SF ? s_hi : SF;

    // XM = 1: coalesced x loads (8 lanes per row: 128-byte runs), staged through this wave's LDS area
    const int lr = lane >> 3, ls = lane & 7;
    float* const stA = smem + (size_t)wave * (D * 16 * 32);
    const float* xc[2];
    unsigned woA[2], ro[2];
    if (XM == 1) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int r = 8 * c + lr;
            const int64_t mm = row0 + r < B ? row0 + r : B - 1;
            xc[c] = x + mm * ldx + 4 * ls;
            woA[c] = r * 32 + ((ls ^ ((r >> 1) & 7)) << 2);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) ro[h] = i16 * 32 + ((((g4 + 4 * h) ^ ((i16 >> 1) & 7)) & 7) << 2);
    }

    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 va[D][2], vw[D][NT][2];

    auto load = [&](int d, int s) {
        if (DIAG != 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (XM == 0) va[d][h] = *reinterpret_cast<const f32x4u*>(xa + 32 * s + 16 * h);
                else va[d][h] = *reinterpret_cast<const f32x4u*>(xc[h] + 32 * s);
            }
        }
        if (DIAG != 1) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int h = 0; h < 2; ++h) vw[d][nt][h] = wa[(((int64_t)s * NT + nt) * 2 + h) * 64];
        }
    };
    auto compute = [&](int d) {
        f32x4 av[2];
        if (XM == 1) {
            float* st = stA + d * (16 * 32);
#pragma unroll
            for (int c = 0; c < 2; ++c) *reinterpret_cast<f32x4*>(st + woA[c]) = va[d][c];
#pragma unroll
            for (int h = 0; h < 2; ++h) av[h] = *reinterpret_cast<const f32x4*>(st + ro[h]);
        } else {
            av[0] = va[d][0]; av[1] = va[d][1];
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (DIAG == 3) acc[nt][j] += av[h][j] * vw[d][nt][h][j];
                    else acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][j], vw[d][nt][h][j], acc[nt], 0, 0, 0);
                }
    };
    if (DIAG == 1 || DIAG == 2) {                              // diagnostic: the skipped operand is loaded once
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int h = 0; h < 2; ++h) va[d][h] = *reinterpret_cast<const f32x4u*>(xa + 16 * h);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int h = 0; h < 2; ++h) vw[d][nt][h] = wa[(nt * 2 + h) * 64];
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (s_lo + d < e) load(d, s_lo + d);
    for (int s = s_lo; s < e; s += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (s + d < e) {
                compute(d);
                if (s + d + D < e) load(d, s + d + D);
            }
        }
    }
    if (s_hi > SF && s_lo <= SF && SF < S) {                   // the ragged last step (one wave): guarded x loads; packed W is zero-filled
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 32 * SF + 16 * h + 4 * g4 + j;
                v[j] = k < K ? x[m * ldx + k] : 0.f;
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 w = wa[(((int64_t)SF * NT + nt) * 2 + h) * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], w[j], acc[nt], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    float* const red = smem;                                   // [NWV][16][N1]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[(wave * 16 + 4 * g4 + i) * N1 + 16 * nt + i16] = acc[nt][i];
    __syncthreads();
    for (int o = tid; o < 16 * N1; o += 64 * NWV) {
        const int r = o / N1, c = o - r * N1;
        float v = red[o];
#pragma unroll
        for (int w = 1; w < NWV; ++w) v += red[w * 16 * N1 + o];
        v += bi[c];
        if (row0 + r < B) emb[(row0 + r) * N1 + c] = v;
    }
}

// x-read floor: every workgroup reads its 16 rows (one contiguous 16 * K * 4-byte chunk) and adds everything up.  P = 0: lane-linear over the
// chunk (perfectly coalesced); 1: the direct operand pattern (16 rows x 64 B per instruction); 2: 8 rows x 128 B per instruction.  U loads in flight.
template <int P, int U, int NWV>
__global__ __launch_bounds__(64 * NWV) void xread_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t B, int64_t ldx, int K) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row0 = (int64_t)blockIdx.x * 16;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (P == 0) {
        const float* base = x + row0 * ldx;
        const int n4 = (int)(16 * (int64_t)K / 4);                     // float4 pieces of the chunk (8-byte aligned: unaligned loads)
        for (int i = tid; i < n4; i += 64 * NWV * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { const int q = i + u * 64 * NWV; v[u] = *reinterpret_cast<const f32x4u*>(base + 4 * (q < n4 ? q : i)); }
#pragma unroll
            for (int u = 0; u < U; ++u) s += v[u];
        }
    } else {
        const int S = K / 32;
        const int s_lo = (int)((int64_t)wave * S / NWV), s_hi = (int)((int64_t)(wave + 1) * S / NWV);
        const float* p0; const float* p1;
        if (P == 1) { p0 = x + (row0 + (lane & 15)) * ldx + 4 * (lane >> 4); p1 = p0 + 16; }
        else { p0 = x + (row0 + (lane >> 3)) * ldx + 4 * (lane & 7); p1 = p0 + 8 * ldx; }
        for (int st = s_lo; st < s_hi; st += U) {
            f32x4 v[U][2];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = st + u < s_hi ? st + u : st;
                v[u][0] = *reinterpret_cast<const f32x4u*>(p0 + 32 * q);
                v[u][1] = *reinterpret_cast<const f32x4u*>(p1 + 32 * q);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) s += v[u][0] + v[u][1];
        }
    }
    const float t = s[0] + s[1] + s[2] + s[3];
    if (t == 12345.678f) out[blockIdx.x * 64 * NWV + tid] = t;
}

extern "C" int ncf_attn_candidates(const float* x, int64_t B, int64_t ldx, int K, const float* Wi, int64_t ldw, const float* bi, int N1,
                                   const float* Wc, const float* b0, int N2, float* emb, int64_t ldemb, float* pc, int64_t ldpc,
                                   const int64_t* pair_row, int64_t R, int pairs_per_wg, int64_t* grp_ptr, int64_t* pair_ids,
                                   int64_t* wg_ptr, int32_t* wg_row, void* workspace, size_t workspace_bytes, int32_t* oob, void* stream);

template <typename F>
static float time_us(F&& f, int reps = 200) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
    const int K = argc > 2 ? atoi(argv[2]) : 2094;
    constexpr int N1 = 64, N2 = 128;
    const int S = (K + 31) / 32;
    std::vector<float> hx((size_t)B * K), hw((size_t)N1 * K), hb(N1), hwc((size_t)N2 * N1), hb0(N2);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)); };
    for (auto& v : hx) { const float r = rnd(); v = r < 0.02f ? 1.f : (r < 0.5f ? rnd() * 0.01f : 0.f); }
    for (auto& v : hw) v = (rnd() - 0.5f) * 0.05f;
    for (auto& v : hb) v = rnd() - 0.5f;
    for (auto& v : hwc) v = (rnd() - 0.5f) * 0.2f;
    for (auto& v : hb0) v = rnd() - 0.5f;
    float *x, *W, *Wp, *bi, *emb, *emb0, *Wc, *b0, *pc;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&W, hw.size() * 4)); CK(hipMalloc(&Wp, (size_t)S * 32 * N1 * 4)); CK(hipMalloc(&bi, N1 * 4));
    CK(hipMalloc(&emb, (size_t)B * N1 * 4)); CK(hipMalloc(&emb0, (size_t)B * N1 * 4)); CK(hipMalloc(&Wc, hwc.size() * 4)); CK(hipMalloc(&b0, N2 * 4));
    CK(hipMalloc(&pc, (size_t)B * N2 * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bi, hb.data(), N1 * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(Wc, hwc.data(), hwc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b0, hb0.data(), N2 * 4, hipMemcpyHostToDevice));
    {
        const int64_t n = (int64_t)S * (N1 / 16) * 512;
        hipLaunchKernelGGL(pack_w_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, W, (int64_t)K, K, N1, S, Wp);
        CK(hipDeviceSynchronize());
    }
    // fp64 reference on 64 sampled rows
    const int NS = 64;
    std::vector<int64_t> rows(NS);
    std::vector<double> ref((size_t)NS * N1);
    for (int t = 0; t < NS; ++t) {
        const int64_t r = (t * 7919 + (t == NS - 1 ? B - 1 : 0)) % B;
        rows[t] = t == NS - 1 ? B - 1 : r;
        for (int n = 0; n < N1; ++n) {
            double a = hb[n];
            for (int k = 0; k < K; ++k) a += (double)hx[(size_t)rows[t] * K + k] * hw[(size_t)n * K + k];
            ref[(size_t)t * N1 + n] = a;
        }
    }
    std::vector<float> out((size_t)B * N1);
    auto check = [&](const float* d) {
        CK(hipMemcpy(out.data(), d, out.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (size_t i = 0; i < ref.size(); ++i) scale = fmax(scale, fabs(ref[i]));
        for (int t = 0; t < NS; ++t)
            for (int n = 0; n < N1; ++n) worst = fmax(worst, fabs(out[(size_t)rows[t] * N1 + n] - ref[(size_t)t * N1 + n]));
        return worst / scale;
    };
    {
        auto f = [&]() { ncf_attn_candidates(x, B, K, K, W, K, bi, N1, Wc, b0, N2, emb0, N1, pc, N2, nullptr, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr); };
        f(); CK(hipDeviceSynchronize());
        const double err = check(emb0);
        printf("library attn_cand (both products, no grouping): %6.2f us   rel err %.2e\n", time_us(f), err);
    }
    const unsigned blocks = (unsigned)((B + 15) / 16);
#define RUN(NWV, D, XM, DIAG, NAME)                                                                                                        \
    {                                                                                                                                      \
        size_t lds = (size_t)NWV * 16 * N1 * 4;                                                                                            \
        if (XM == 1 && (size_t)NWV * D * 16 * 32 * 4 > lds) lds = (size_t)NWV * D * 16 * 32 * 4;                                           \
        auto kern = cand_direct_kernel<N1, NWV, D, XM, DIAG>;                                                                              \
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                \
        CK(hipMemset(emb, 0, (size_t)B * N1 * 4));                                                                                         \
        auto f = [&]() { hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * NWV), lds, 0, x, Wp, bi, emb, B, (int64_t)K, K, S); };           \
        f(); CK(hipDeviceSynchronize()); CK(hipGetLastError());                                                                            \
        const double err = check(emb);                                                                                                     \
        printf("direct NWV=%2d D=%d XM=%d %-22s: %6.2f us   rel err %.2e\n", NWV, D, XM, NAME, time_us(f), err);                            \
        fflush(stdout);                                                                                                                    \
    }

#define XR(P, U, NWV)                                                                                                                      \
    {                                                                                                                                      \
        auto f = [&]() { hipLaunchKernelGGL((xread_kernel<P, U, NWV>), dim3(blocks), dim3(64 * NWV), 0, 0, x, emb, B, (int64_t)K, K); };   \
        f(); CK(hipDeviceSynchronize()); CK(hipGetLastError());                                                                            \
        printf("xread P=%d U=%d NWV=%2d: %6.2f us\n", P, U, NWV, time_us(f));                                                              \
        fflush(stdout);                                                                                                                    \
    }
    XR(0, 4, 8); XR(0, 8, 8); XR(0, 16, 8); XR(0, 4, 16); XR(0, 8, 16);
    XR(1, 2, 8); XR(1, 4, 8); XR(1, 8, 8); XR(1, 2, 16); XR(1, 4, 16);
    XR(2, 2, 8); XR(2, 4, 8); XR(2, 8, 8); XR(2, 2, 16); XR(2, 4, 16);
    {
        auto f = [&]() { hipLaunchKernelGGL((xread_kernel<0, 4, 8>), dim3(1), dim3(512), 0, 0, x, emb, (int64_t)16, (int64_t)K, K); };
        printf("one workgroup (launch floor): %6.2f us\n", time_us(f));
    }
    RUN(8, 2, 0, 0, "");
    RUN(8, 3, 0, 0, "");
    RUN(8, 4, 0, 0, "");
    RUN(16, 2, 0, 0, "");
    RUN(16, 3, 0, 0, "");
    RUN(8, 2, 1, 0, "");
    RUN(8, 4, 1, 0, "");
    RUN(16, 2, 1, 0, "");
    RUN(16, 2, 0, 1, "(diag: W once)");
    RUN(16, 2, 0, 2, "(diag: x once)");
    RUN(16, 2, 0, 3, "(diag: no MFMA)");
    RUN(8, 4, 0, 1, "(diag: W once)");
    RUN(8, 4, 0, 2, "(diag: x once)");
    RUN(8, 4, 0, 3, "(diag: no MFMA)");
    return 0;
}
