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

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short v) { return __uint_as_float(((unsigned int)v) << 16); }

// Row of a 32x32 MFMA accumulator held in register r of a lane in half h (= lane >> 5):
// row = (r & 3) + 8 * (r >> 2) + 4 * h   (MI355X C/D layout, dtype independent).
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace ncf
