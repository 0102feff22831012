// Training-step kernels: the backward of K1 (embedding-gradient scatter-add) and of K2 (weight
// gradient dW = dY^T . X as a "TN" MFMA GEMM with an ordered split over the batch, bias gradient, ReLU mask).
// The data gradient dX = dY . W is the forward row-streaming GEMM with the transposed weight.
#include "ncf_common.h"

namespace ncf {

// out[N1][N2] = sum_m A[m][n1] * Bm[m][n2]  (A: [M][N1], Bm: [M][N2], both row-major).
// v_mfma_f32_32x32x2_f32 with k = batch row: lane (i, h) supplies A[m0 + h][n1_0 + i] and Bm[m0 + h][n2_0 + i] — lanes
// run along the contiguous dimension of both operands, so every operand load is a coalesced 128-byte half-wave read.
// Workgroup = 4 waves = a 64 x 64 output block (wave (wi, wj) owns a 32 x 32 tile); grid.z splits M into slices whose
// partial blocks go to `partial[slice]` and are added in slice order by gemm_tn_reduce_kernel (deterministic).
template <int UNROLL>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm,
                                                      int64_t ldb, int64_t M, int N1, int N2, int64_t rows_per_slice,
                                                      float* __restrict__ partial) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int n1 = blockIdx.x * 64 + 32 * (wave >> 1) + i;
    const int n2 = blockIdx.y * 64 + 32 * (wave & 1) + i;
    const bool ok1 = n1 < N1, ok2 = n2 < N2;
    const int64_t m_lo = (int64_t)blockIdx.z * rows_per_slice;
    const int64_t m_hi = m_lo + rows_per_slice < M ? m_lo + rows_per_slice : M;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* pa = A + (ok1 ? n1 : 0);
    const float* pb = Bm + (ok2 ? n2 : 0);
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += 2 * UNROLL) {
        float av[UNROLL], bv[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t m = m0 + 2 * u + h;
            const bool in = m < m_hi;
            av[u] = (in && ok1) ? pa[m * lda] : 0.f;
            bv[u] = (in && ok2) ? pb[m * ldb] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
    }
    // D[i1][i2]: column i2 = lane & 31 (n2), row i1 = acc_row(r, h) (n1)
    float* out = partial + (size_t)blockIdx.z * N1 * N2;
    if (ok2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = blockIdx.x * 64 + 32 * (wave >> 1) + acc_row(r, h);
            if (row < N1) out[(size_t)row * N2 + n2] = acc[r];
        }
    }
}

__global__ void gemm_tn_reduce_kernel(const float* __restrict__ partial, int slices, int64_t n, float* __restrict__ out, int64_t ldo,
                                      int N2) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < slices; ++k) s += partial[(size_t)k * n + e];  // slice order: deterministic
        out[(e / N2) * ldo + (e % N2)] = s;
    }
}

// partial[slice][n] = sum over the slice's rows of X[m][n]: block (x = 64 columns, y = row slice), rows strided over the
// block's 4 waves, fixed-order LDS combine; gemm_tn_reduce_kernel then adds the slices in order (deterministic).
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int64_t ldx, int64_t M, int N, int64_t rows_per_slice,
                                                     float* __restrict__ partial) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const int64_t m_lo = (int64_t)blockIdx.y * rows_per_slice;
    const int64_t m_hi = m_lo + rows_per_slice < M ? m_lo + rows_per_slice : M;
    float s = 0.f;
    if (n < N)
        for (int64_t m = m_lo + wave; m < m_hi; m += 4) s += X[m * ldx + n];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && n < N) partial[(size_t)blockIdx.y * N + n] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// dY[m][n] = Y[m][n] > 0 ? dY[m][n] : 0   (backward of the ReLU fused into the forward GEMM epilogue)
__global__ void relu_backward_kernel(float* __restrict__ dY, int64_t ldd, const float* __restrict__ Y, int64_t ldy, int64_t M, int N) {
    const int64_t total = M * N;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / N;
        const int n = (int)(e - m * N);
        if (!(Y[m * ldy + n] > 0.f)) dY[m * ldd + n] = 0.f;
    }
}

// dst[idx[p], 0:E] += src[p, 0:E]  — embedding-table gradient.  Lanes run along E (a 256-byte row per 64 lanes at
// E = 64: the full-rate shape for global float atomics); duplicates of an id are added by the memory-side atomics, so
// the sum is order-dependent in the last bits (like torch's CUDA index_add_).
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(const float* __restrict__ src, int64_t lds_, const int64_t* __restrict__ idx,
                                                               int64_t B, int E, float* __restrict__ dst, int64_t ldd, int64_t rows,
                                                               int32_t* oob) {
    const int64_t total = B * E;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / E;
        const int c = (int)(e - p * E);
        const int64_t r = idx ? idx[p] : p;
        if (r >= 0 && r < rows) atomicAdd(dst + r * ldd + c, src[p * lds_ + c]);
        else if (oob) *oob = 1;
    }
}

}  // namespace ncf

using namespace ncf;

extern "C" size_t ncf_gemm_tn_workspace_bytes(int64_t M, int N1, int N2) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
    int64_t slices = (M + 2047) / 2048;
    if (slices > 64) slices = 64;
    return (size_t)slices * N1 * N2 * sizeof(float);
}

extern "C" int ncf_gemm_tn(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int N1, int N2, float* out,
                           int64_t ldo, void* workspace, size_t ws_bytes, ncf_stream_t stream) {
    if (N1 <= 0 || N2 <= 0 || M < 0 || !out || ldo < N2) return fail(NCF_EINVAL, "ncf_gemm_tn: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return hipMemset2DAsync(out, ldo * sizeof(float), 0, N2 * sizeof(float), N1, s) == hipSuccess ? NCF_OK : fail(NCF_ELAUNCH, "ncf_gemm_tn: memset failed");
    if (!A || !Bm || lda < N1 || ldb < N2) return fail(NCF_EINVAL, "ncf_gemm_tn: bad operand");
    const size_t need = ncf_gemm_tn_workspace_bytes(M, N1, N2);
    if (!workspace || ws_bytes < need) return fail(NCF_EWORKSPACE, "ncf_gemm_tn: workspace %zu < %zu bytes", ws_bytes, need);
    int64_t slices = (M + 2047) / 2048;
    if (slices > 64) slices = 64;
    int64_t rps = (M + slices - 1) / slices;
    rps = (rps + 15) & ~int64_t(15);
    dim3 grid((N1 + 63) / 64, (N2 + 63) / 64, (unsigned)slices);
    hipLaunchKernelGGL(gemm_tn_kernel<8>, grid, dim3(256), 0, s, A, lda, Bm, ldb, M, N1, N2, rps, (float*)workspace);
    const int64_t n = (int64_t)N1 * N2;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, (int)slices, n, out, ldo, N2);
    return check_launch("ncf_gemm_tn");
}

static int64_t colsum_slices(int64_t M) {
    int64_t slices = (M + 255) / 256;
    return slices > 256 ? 256 : (slices < 1 ? 1 : slices);
}

extern "C" size_t ncf_colsum_workspace_bytes(int64_t M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)colsum_slices(M) * N * sizeof(float);
}

extern "C" int ncf_colsum(const float* X, int64_t ldx, int64_t M, int N, float* out, void* workspace, size_t ws_bytes, ncf_stream_t stream) {
    if (N <= 0 || M < 0 || !out || (M > 0 && (!X || ldx < N))) return fail(NCF_EINVAL, "ncf_colsum: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return hipMemsetAsync(out, 0, N * sizeof(float), s) == hipSuccess ? NCF_OK : fail(NCF_ELAUNCH, "ncf_colsum: memset failed");
    const size_t need = ncf_colsum_workspace_bytes(M, N);
    if (!workspace || ws_bytes < need) return fail(NCF_EWORKSPACE, "ncf_colsum: workspace %zu < %zu bytes", ws_bytes, need);
    const int64_t slices = colsum_slices(M);
    const int64_t rps = (M + slices - 1) / slices;
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 63) / 64, (unsigned)slices), dim3(256), 0, s, X, ldx, M, N, rps, (float*)workspace);
    int64_t blocks = (N + 255) / 256;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)workspace, (int)slices, (int64_t)N, out, (int64_t)N, N);
    return check_launch("ncf_colsum");
}

extern "C" int ncf_relu_backward(float* dY, int64_t ldd, const float* Y, int64_t ldy, int64_t M, int N, ncf_stream_t stream) {
    if (M == 0) return NCF_OK;
    if (N <= 0 || M < 0 || !dY || !Y || ldd < N || ldy < N) return fail(NCF_EINVAL, "ncf_relu_backward: bad argument");
    int64_t blocks = (M * N + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(relu_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dY, ldd, Y, ldy, M, N);
    return check_launch("ncf_relu_backward");
}

extern "C" int ncf_scatter_add_rows(const float* src, int64_t ld_src, const int64_t* idx, int64_t B, int E, float* dst, int64_t ld_dst,
                                    int64_t rows, int32_t* oob, ncf_stream_t stream) {
    if (B == 0) return NCF_OK;
    if (B < 0 || E <= 0 || !src || !dst || ld_src < E || ld_dst < E) return fail(NCF_EINVAL, "ncf_scatter_add_rows: bad argument");
    int64_t blocks = (B * E + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx, B, E, dst, ld_dst, rows, oob);
    return check_launch("ncf_scatter_add_rows");
}
