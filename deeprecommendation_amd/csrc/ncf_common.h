// Shared helpers for libncf_hip.so (gfx950 only).  Internal — the public surface is include/ncf_abi.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ncf_abi.h"

namespace ncf {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NCF_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return NCF_OK;
}

// Process-wide kernel-selection overrides (ncf_set_option; 0 = choose by shape) and the per-device CU count (abi.hip).
enum {
    NCF_OPT_BF16_KERNEL = 0,         // 1 = 4-wave weight-stationary kernel, 2 = slab-streaming kernel, 3 = 8-wave weight-stationary kernel
    NCF_OPT_LINEAR_KERNEL = 1,       // 1 = one tile per wave (rs), 2 = persistent row-streaming (rsp)
    NCF_OPT_LINEAR_KSLICES = 2,      // 4 / 8 K-slices of the skinny-deep Linear form
    NCF_OPT_ATTN_GROUPED_KERNEL = 3, // 1 = first LDS-broadcast form, 2 = scalar-operand form
    NCF_OPT_GATHER_KERNEL = 4,       // 1 = one step per wave, 2 = persistent prefetching waves
    NCF_OPT_COUNT_
};
int option(int opt);
int num_cus();

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Stream-ordered fill of 32-bit words BY A KERNEL.  hipMemsetAsync is not used in this library: captured into a HIP graph, memset
// nodes and the kernel nodes around them lost their order from the second back-to-back replay on (ROCm 7.2 — a replayed
// dense-user_matrix forward hung in the hash-table probe; tools/graph_dense_probe.py); kernel nodes keep it.
static __global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t* __restrict__ p, uint32_t v, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
inline void fill_u32_async(void* p, uint32_t v, size_t nbytes, hipStream_t s) {   // nbytes: a multiple of 4
    const size_t n = nbytes / 4;
    if (n) hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (uint32_t*)p, v, n);
}

constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float(((unsigned int)v) << 16); }

// Row of a 32x32 MFMA accumulator held in register r of a lane in half h (= lane >> 5):
// row = (r & 3) + 8 * (r >> 2) + 4 * h   (MI355X C/D layout, dtype independent).
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// bf16 fused path (mlp_bf16.hip), reached through the dtype argument of the ncf_mlp_pack / ncf_score_fused entries
bool bf16_shape_ok(int EA, int EB, int n_layers, const int* dims);
size_t bf16_packed_bytes(int n_layers, const int* dims);
int bf16_pack(int n_layers, const int* dims, const void* const* W, const void* const* b, void* packed, size_t packed_bytes,
              hipStream_t s);
int bf16_score(const void* tabA, int64_t rowsA, int64_t ldA, const void* tabB, int64_t rowsB, int64_t ldB, const int64_t* idxA,
               const int64_t* idxB, int64_t B, int EA, int EB, int n_layers, const int* dims, const void* packed, float* out,
               int32_t* oob, hipStream_t s);

}  // namespace ncf
