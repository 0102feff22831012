// Training-step kernels: the backward of K1 (embedding-gradient scatter-add) and of K2 (weight
// gradient dW = dY^T . X as a "TN" MFMA GEMM with an ordered split over the batch, bias gradient, ReLU mask).
// The data gradient dX = dY . W is the forward row-streaming GEMM with the transposed weight.
#include "ncf_common.h"
#include <math.h>

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
    // two register sets: the loads of rows m0 + 2*UNROLL.. fly while the MFMAs of rows m0.. run (one or two waves per
    // SIMD cannot hide a 1-2 us activation fetch behind 8 MFMAs otherwise: measured 131 us -> see DESIGN 4.8)
    float av[2][UNROLL], bv[2][UNROLL];
    auto fetch = [&](int set, int64_t m0) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t m = m0 + 2 * u + h;
            const bool in = m < m_hi;
            av[set][u] = (in && ok1) ? pa[m * lda] : 0.f;
            bv[set][u] = (in && ok2) ? pb[m * ldb] : 0.f;
        }
    };
    fetch(0, m_lo);
    for (int64_t m0 = m_lo; m0 < m_hi; m0 += 4 * UNROLL) {
        fetch(1, m0 + 2 * UNROLL);                         // rows past m_hi read as zeros
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][u], bv[0][u], acc, 0, 0, 0);
        fetch(0, m0 + 4 * UNROLL);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1][u], bv[1][u], acc, 0, 0, 0);
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
        int k = 0;
        for (; k + 8 <= slices; k += 8) {                  // 8 independent loads in flight, added in slice order: deterministic
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = partial[(size_t)(k + j) * n + e];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += t[j];
        }
        for (; k < slices; ++k) s += partial[(size_t)k * n + e];
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

// out[m][n] = Y[m][n] > 0 ? scale * dY[m][n] : 0, dY left untouched (autograd hands gradients it may still share): one
// pass instead of a copy followed by the in-place mask.  With Y = dropout(relu(.)) and scale = 1/(1-p) this is the backward
// of ReLU AND of the dropout behind it (Y > 0 exactly where the ReLU was active and the element was kept).
// VEC: 16-byte accesses when widths, strides and bases allow.
template <bool VEC>
__global__ void relu_backward_out_kernel(const float* __restrict__ dY, int64_t ldd, const float* __restrict__ Y, int64_t ldy,
                                         float* __restrict__ out, int64_t ldo, int64_t M, int N, float scale) {
    const int W = VEC ? N / 4 : N;
    const int64_t total = M * W;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = e / W;
        const int c = (int)(e - m * W);
        if (VEC) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(dY + m * ldd + 4 * c);
            const f32x4 y = *reinterpret_cast<const f32x4*>(Y + m * ldy + 4 * c);
            f32x4 r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = y[j] > 0.f ? g[j] * scale : 0.f;
            *reinterpret_cast<f32x4*>(out + m * ldo + 4 * c) = r;
        } else {
            out[m * ldo + c] = Y[m * ldy + c] > 0.f ? dY[m * ldd + c] * scale : 0.f;
        }
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

// ---- nn.Linear-layout embeddings (basic_ncf.py:25-33 keeps W as [E, U]: Linear(onehot(i)) = W[:, i] + b) ----
// out[p, e] = W[e, idx[p]] + b[e]: a COLUMN gather (lanes along e: 64 rows of W, one 4-byte element each)
__global__ __launch_bounds__(256) void gather_cols_kernel(const float* __restrict__ W, int64_t ldw, const float* __restrict__ bias,
                                                          const int64_t* __restrict__ idx, int64_t B, int E, int64_t cols,
                                                          float* __restrict__ out, int64_t ldo, int32_t* oob) {
    const int64_t total = B * E;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / E;
        const int e = (int)(t - p * E);
        const int64_t c = idx[p];
        float v = 0.f;
        if (c >= 0 && c < cols) v = W[e * ldw + c] + (bias ? bias[e] : 0.f);
        else if (oob) *oob = 1;
        out[p * ldo + e] = v;
    }
}
// dW[e, idx[p]] += src[p, e]: the gradient of that gather, straight into the [E, U] layout of the parameter (torch's
// index_put backward builds it through a sort and two table-sized copies)
__global__ __launch_bounds__(256) void scatter_add_cols_kernel(const float* __restrict__ src, int64_t lds_, const int64_t* __restrict__ idx,
                                                               int64_t B, int E, float* __restrict__ dst, int64_t ldd, int64_t cols,
                                                               int32_t* oob) {
    const int64_t total = B * E;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = t / E;
        const int e = (int)(t - p * E);
        const int64_t c = idx[p];
        if (c >= 0 && c < cols) atomicAdd(dst + e * ldd + c, src[p * lds_ + e]);
        else if (oob) *oob = 1;
    }
}

// One pass of torch.optim.Adam's update (train.py:55: Adam(lr, weight_decay); amsgrad / maximize off) over one tensor:
//   g += wd * p;  m += (1 - b1) * (g - m);  v = b2 * v + (1 - b2) * g * g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// — the operation order of torch's own single-tensor path.  7 table passes (read p, g, m, v; write p, m, v) in ONE kernel;
// torch's default (foreach) path runs ~9 elementwise kernels over the same tensors.
__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, int64_t n, float lr_over_bc1, float beta1, float beta2,
                                                        float inv_sqrt_bc2, float eps, float wd) {
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pv = reinterpret_cast<const f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<const f32x4*>(m)[i], vv = reinterpret_cast<const f32x4*>(v)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = wd != 0.f ? gv[j] + wd * pv[j] : gv[j];
            mv[j] = mv[j] + (1.f - beta1) * (gg - mv[j]);
            vv[j] = beta2 * vv[j] + (1.f - beta2) * gg * gg;
            const float denom = sqrtf(vv[j]) * inv_sqrt_bc2 + eps;
            pv[j] = pv[j] - lr_over_bc1 * (mv[j] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0) {
        const int64_t i = 4 * n4 + threadIdx.x;
        if (i < n) {
            const float gg = wd != 0.f ? g[i] + wd * p[i] : g[i];
            const float mm = m[i] + (1.f - beta1) * (gg - m[i]);
            const float vv = beta2 * v[i] + (1.f - beta2) * gg * gg;
            m[i] = mm;
            v[i] = vv;
            p[i] = p[i] - lr_over_bc1 * (mm / (sqrtf(vv) * inv_sqrt_bc2 + eps));
        }
    }
}

}  // namespace ncf

using namespace ncf;

extern "C" size_t ncf_gemm_tn_workspace_bytes(int64_t M, int N1, int N2) {
    if (M <= 0 || N1 <= 0 || N2 <= 0) return 0;
    int64_t slices = (M + 1023) / 1024;
    if (slices > 64) slices = 64;
    return (size_t)slices * N1 * N2 * sizeof(float);
}

extern "C" int ncf_gemm_tn(const float* A, int64_t lda, const float* Bm, int64_t ldb, int64_t M, int N1, int N2, float* out,
                           int64_t ldo, void* workspace, size_t ws_bytes, ncf_stream_t stream) {
    if (N1 <= 0 || N2 <= 0 || M < 0 || !out || ldo < N2) return fail(NCF_EINVAL, "ncf_gemm_tn: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {                                              // an empty batch: the gradient is zero (a kernel, not a memset: ncf_common.h)
        for (int r = 0; r < N1; ++r) fill_u32_async(out + (int64_t)r * ldo, 0u, (size_t)N2 * sizeof(float), s);
        return check_launch("ncf_gemm_tn (empty)");
    }
    if (!A || !Bm || lda < N1 || ldb < N2) return fail(NCF_EINVAL, "ncf_gemm_tn: bad operand");
    const size_t need = ncf_gemm_tn_workspace_bytes(M, N1, N2);
    if (!workspace || ws_bytes < need) return fail(NCF_EWORKSPACE, "ncf_gemm_tn: workspace %zu < %zu bytes", ws_bytes, need);
    int64_t slices = (M + 1023) / 1024;
    if (slices > 64) slices = 64;
    int64_t rps = (M + slices - 1) / slices;
    rps = (rps + 31) & ~int64_t(31);
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
    if (M == 0) { fill_u32_async(out, 0u, (size_t)N * sizeof(float), s); return check_launch("ncf_colsum (empty)"); }
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

extern "C" int ncf_relu_backward_out(const float* dY, int64_t ldd, const float* Y, int64_t ldy, float* out, int64_t ldo, int64_t M, int N,
                                     float scale, ncf_stream_t stream) {
    if (M == 0) return NCF_OK;
    if (N <= 0 || M < 0 || !dY || !Y || !out || ldd < N || ldy < N || ldo < N) return fail(NCF_EINVAL, "ncf_relu_backward_out: bad argument");
    const bool vec = N % 4 == 0 && ldd % 4 == 0 && ldy % 4 == 0 && ldo % 4 == 0 && aligned16(dY) && aligned16(Y) && aligned16(out);
    int64_t blocks = (M * (vec ? N / 4 : N) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (vec) hipLaunchKernelGGL(relu_backward_out_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dY, ldd, Y, ldy, out, ldo, M, N, scale);
    else hipLaunchKernelGGL(relu_backward_out_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dY, ldd, Y, ldy, out, ldo, M, N, scale);
    return check_launch("ncf_relu_backward_out");
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

extern "C" int ncf_gather_cols(const float* W, int64_t ldw, const float* bias, const int64_t* idx, int64_t B, int E, int64_t cols,
                               float* out, int64_t ldo, int32_t* oob, ncf_stream_t stream) {
    if (B == 0) return NCF_OK;
    if (B < 0 || E <= 0 || cols < 0 || !W || !idx || !out || ldw < cols || ldo < E) return fail(NCF_EINVAL, "ncf_gather_cols: bad argument");
    int64_t blocks = (B * E + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, W, ldw, bias, idx, B, E, cols, out, ldo, oob);
    return check_launch("ncf_gather_cols");
}

extern "C" int ncf_scatter_add_cols(const float* src, int64_t ld_src, const int64_t* idx, int64_t B, int E, float* dst, int64_t ld_dst,
                                    int64_t cols, int32_t* oob, ncf_stream_t stream) {
    if (B == 0) return NCF_OK;
    if (B < 0 || E <= 0 || cols < 0 || !src || !idx || !dst || ld_src < E || ld_dst < cols) return fail(NCF_EINVAL, "ncf_scatter_add_cols: bad argument");
    int64_t blocks = (B * E + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(scatter_add_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx, B, E, dst, ld_dst, cols, oob);
    return check_launch("ncf_scatter_add_cols");
}

extern "C" int ncf_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                             float weight_decay, int64_t step, ncf_stream_t stream) {
    if (n == 0) return NCF_OK;
    if (n < 0 || step < 1 || !p || !g || !m || !v) return fail(NCF_EINVAL, "ncf_adam_step: bad argument");
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return fail(NCF_EINVAL, "ncf_adam_step: tensors must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    int64_t blocks = (n / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)(lr / bc1), beta1, beta2,
                       (float)(1.0 / sqrt(bc2)), eps, weight_decay);
    return check_launch("ncf_adam_step");
}
