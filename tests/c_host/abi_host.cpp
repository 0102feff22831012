// A host program that uses libncf_hip.so through include/ncf_abi.h and the HIP runtime only — no Python, no torch:
// what a maintainer binding the library from C / C++ (or cgo / JNI / N-API over the same header) would write.
// Builds tables and a 128-256-128-1 MLP on the host, scores a batch through ncf_gather_concat + ncf_mlp_forward and
// through ncf_mlp_pack + ncf_score_fused on a user-created stream, and checks both against a double-precision host
// evaluation (1e-5 relative, the fp32 bar of BASELINE.json) and the gathered rows bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../include/ncf_abi.h"

#define HIP_OK(x)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } \
    } while (0)
#define NCF_OK_(x)                                                                            \
    do {                                                                                      \
        int r_ = (x);                                                                         \
        if (r_ != NCF_OK) { fprintf(stderr, "ncf error %d (%s) at %s:%d\n", r_, ncf_last_error(), __FILE__, __LINE__); return 3; } \
    } while (0)

static float frand(unsigned& s) {
    s = s * 1664525u + 1013904223u;
    return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f;
}

int main() {
    const int U = 5000, I = 700, E = 64, B = 3001;
    const int dims[4] = {2 * E, 256, 128, 1};
    unsigned seed = 12345;
    std::vector<float> tu((size_t)U * E), ti((size_t)I * E);
    for (auto& v : tu) v = frand(seed);
    for (auto& v : ti) v = frand(seed);
    std::vector<float> W[3], b[3];
    for (int l = 0; l < 3; ++l) {
        W[l].resize((size_t)dims[l + 1] * dims[l]);
        b[l].resize(dims[l + 1]);
        for (auto& v : W[l]) v = frand(seed) * 2.0f / sqrtf((float)dims[l]);
        for (auto& v : b[l]) v = frand(seed) * 0.2f;
    }
    std::vector<int64_t> iu(B), ii(B);
    for (int p = 0; p < B; ++p) { seed = seed * 1664525u + 1013904223u; iu[p] = (seed >> 4) % U; seed = seed * 1664525u + 1013904223u; ii[p] = (seed >> 4) % I; }

    printf("library version %d, built for %s\n", ncf_version(), ncf_build_arch());
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float *d_tu, *d_ti, *d_x, *d_out1, *d_out2, *d_W[3], *d_b[3];
    int64_t *d_iu, *d_ii;
    int32_t* d_oob;
    HIP_OK(hipMalloc(&d_tu, tu.size() * 4)); HIP_OK(hipMalloc(&d_ti, ti.size() * 4));
    HIP_OK(hipMalloc(&d_x, (size_t)B * 2 * E * 4)); HIP_OK(hipMalloc(&d_out1, (size_t)B * 4)); HIP_OK(hipMalloc(&d_out2, (size_t)B * 4));
    HIP_OK(hipMalloc(&d_iu, (size_t)B * 8)); HIP_OK(hipMalloc(&d_ii, (size_t)B * 8)); HIP_OK(hipMalloc(&d_oob, 4));
    HIP_OK(hipMemcpy(d_tu, tu.data(), tu.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ti, ti.data(), ti.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_iu, iu.data(), (size_t)B * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_ii, ii.data(), (size_t)B * 8, hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_oob, 0, 4));
    const void* Wp[3];
    const void* bp[3];
    for (int l = 0; l < 3; ++l) {
        HIP_OK(hipMalloc(&d_W[l], W[l].size() * 4)); HIP_OK(hipMalloc(&d_b[l], b[l].size() * 4));
        HIP_OK(hipMemcpy(d_W[l], W[l].data(), W[l].size() * 4, hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(d_b[l], b[l].data(), b[l].size() * 4, hipMemcpyHostToDevice));
        Wp[l] = d_W[l]; bp[l] = d_b[l];
    }
    // path 1: K1 gather + K2 generic MLP
    NCF_OK_(ncf_gather_concat(NCF_F32, d_tu, U, E, d_ti, I, E, d_iu, d_ii, B, E, E, d_x, 2 * E, d_oob, stream));
    const size_t ws_bytes = ncf_mlp_workspace_bytes(NCF_F32, B, 3, dims);
    void* d_ws;
    HIP_OK(hipMalloc(&d_ws, ws_bytes ? ws_bytes : 16));
    NCF_OK_(ncf_mlp_forward(NCF_F32, d_x, B, 2 * E, 3, dims, Wp, bp, d_ws, ws_bytes, d_out1, 1, stream));
    // path 2: packed weights + fused gather->MLP
    if (!ncf_score_fused_supported(NCF_F32, E, E, 3, dims)) { fprintf(stderr, "no fused instance for 128-256-128-1\n"); return 4; }
    const size_t pk_bytes = ncf_mlp_packed_bytes(NCF_F32, 3, dims);
    void* d_pk;
    HIP_OK(hipMalloc(&d_pk, pk_bytes));
    NCF_OK_(ncf_mlp_pack(NCF_F32, 3, dims, Wp, bp, d_pk, pk_bytes, stream));
    NCF_OK_(ncf_score_fused(NCF_F32, d_tu, U, E, d_ti, I, E, d_iu, d_ii, B, E, E, 3, dims, d_pk, d_out2, d_oob, stream));
    HIP_OK(hipStreamSynchronize(stream));

    std::vector<float> x((size_t)B * 2 * E), o1(B), o2(B);
    int32_t oob = -1;
    HIP_OK(hipMemcpy(x.data(), d_x, x.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o1.data(), d_out1, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o2.data(), d_out2, (size_t)B * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&oob, d_oob, 4, hipMemcpyDeviceToHost));
    if (oob != 0) { fprintf(stderr, "out-of-range flag set\n"); return 5; }

    double max_ref = 0, err1 = 0, err2 = 0;
    int bad_rows = 0;
    std::vector<double> h1(256), h2(128);
    for (int p = 0; p < B; ++p) {
        if (memcmp(&x[(size_t)p * 2 * E], &tu[(size_t)iu[p] * E], E * 4) || memcmp(&x[(size_t)p * 2 * E + E], &ti[(size_t)ii[p] * E], E * 4)) ++bad_rows;
        for (int n = 0; n < 256; ++n) {
            double s = b[0][n];
            for (int k = 0; k < 2 * E; ++k) s += (double)W[0][(size_t)n * 2 * E + k] * x[(size_t)p * 2 * E + k];
            h1[n] = s > 0 ? s : 0;
        }
        for (int n = 0; n < 128; ++n) {
            double s = b[1][n];
            for (int k = 0; k < 256; ++k) s += (double)W[1][(size_t)n * 256 + k] * h1[k];
            h2[n] = s > 0 ? s : 0;
        }
        double s = b[2][0];
        for (int k = 0; k < 128; ++k) s += (double)W[2][k] * h2[k];
        max_ref = fmax(max_ref, fabs(s));
        err1 = fmax(err1, fabs(o1[p] - s));
        err2 = fmax(err2, fabs(o2[p] - s));
    }
    printf("gathered rows differing: %d; max |ref| %.4f; max err generic %.3e, fused %.3e\n", bad_rows, max_ref, err1, err2);
    // error behaviour: a bad argument returns a code and a message, never a fault
    const int rc = ncf_gather_concat(7, d_tu, U, E, d_ti, I, E, d_iu, d_ii, B, E, E, d_x, 2 * E, d_oob, stream);
    printf("bad dtype -> %d (%s)\n", rc, ncf_last_error());
    const bool ok = bad_rows == 0 && err1 <= 1e-5 * max_ref + 1e-6 && err2 <= 1e-5 * max_ref + 1e-6 && rc == NCF_EINVAL;
    printf(ok ? "ABI HOST OK\n" : "ABI HOST FAILED\n");
    return ok ? 0 : 1;
}
