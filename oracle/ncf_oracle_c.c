/* CPU restatement, in plain C, of the ARITHMETIC ORDER of the fused gather -> MLP kernel
 * (deeprecommendation_amd/csrc/mlp_fused.hip) — TEST INFRASTRUCTURE ONLY (see oracle/ncf_oracle.py for the
 * reference-level oracle this is itself checked against).
 *
 * What it restates: models/basic_ncf.py:37-42 of the reference in table form (Linear(onehot(i)) == W[:, i] + b, then
 * cat(user, item) -> util.py:5-18 MLP with ReLU between layers and a 1-wide last layer), evaluated with exactly the
 * floating-point operation order of the GPU kernel, so that the GPU result can be compared BIT FOR BIT:
 *   - v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf chain (one rounding per product, MI355X guide), k = lane
 *     half 0 first, then lane half 1; the kernel feeds k = 8q + j (half 0) and k = 8q + 4 + j (half 1) for j = 0..3;
 *   - accumulators start from the bias;
 *   - the last layer is an fmaf chain per lane half over n = 32nt + 8g + 4h + j (nt, g, j ascending), the two halves
 *     are added (half 0 + half 1), then the output bias.
 * Compile without FMA contraction and without fast-math: the only fused operations are the explicit fmaf() calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static inline float relu(float x) { return x > 0.f ? x : 0.f; }

/* one layer in MFMA order: out[n] = chain over q, j of (k = 8q+j, then k = 8q+4+j), starting from bias[n] */
static void layer_mfma_order(const float* x, int K, const float* W, const float* b, int N, float* out) {
    for (int n = 0; n < N; ++n) {
        float acc = b[n];
        const float* w = W + (size_t)n * K;
        for (int q = 0; q < K / 8; ++q)
            for (int j = 0; j < 4; ++j) {
                acc = fmaf(w[8 * q + j], x[8 * q + j], acc);
                acc = fmaf(w[8 * q + 4 + j], x[8 * q + 4 + j], acc);
            }
        out[n] = acc;
    }
}

/* N2 == 0: single hidden layer.  Returns 0, or -1 on a shape the kernel does not take either. */
int oracle_score_fused_f32(const float* tabA, int64_t ldA, const float* tabB, int64_t ldB,
                           const int64_t* idxA, const int64_t* idxB, int64_t B, int EA, int EB,
                           const float* W1, const float* b1, int N1,
                           const float* W2, const float* b2, int N2,
                           const float* wl, const float* bl, float* out) {
    const int K0 = EA + EB;
    if (K0 % 8 || N1 % 32 || N2 % 32) return -1;
    const int NL = N2 > 0 ? N2 : N1;
#pragma omp parallel
    {
        float* x = (float*)malloc(sizeof(float) * (size_t)(K0 + N1 + (N2 > 0 ? N2 : 1)));
        float* h1 = x + K0;
        float* h2 = h1 + N1;
#pragma omp for
        for (int64_t p = 0; p < B; ++p) {
            const float* ra = tabA + idxA[p] * ldA;
            const float* rb = tabB + idxB[p] * ldB;
            for (int e = 0; e < EA; ++e) x[e] = ra[e];
            for (int e = 0; e < EB; ++e) x[EA + e] = rb[e];
            layer_mfma_order(x, K0, W1, b1, N1, h1);
            const float* last = h1;
            if (N2 > 0) {
                for (int n = 0; n < N1; ++n) h1[n] = relu(h1[n]);
                layer_mfma_order(h1, N1, W2, b2, N2, h2);
                last = h2;
            }
            float part[2] = {0.f, 0.f};
            for (int h = 0; h < 2; ++h)
                for (int nt = 0; nt < NL / 32; ++nt)
                    for (int g = 0; g < 4; ++g)
                        for (int j = 0; j < 4; ++j) {
                            const int n = 32 * nt + 8 * g + 4 * h + j;
                            part[h] = fmaf(wl[n], relu(last[n]), part[h]);
                        }
            out[p] = (part[0] + part[1]) + bl[0];
        }
        free(x);
    }
    return 0;
}
