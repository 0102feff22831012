// Device helpers shared by the attention kernels (attn.hip, attn_split.hip).  Internal.
#pragma once
#include "ncf_common.h"
#include <math.h>
#include <type_traits>

namespace ncf {

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// Wave-wide reductions on the VALU (DPP row operations + 4 readlanes) instead of six dependent ds_bpermute round trips:
// quad swaps, then half-row and row mirrors give every lane its 16-lane row's total; the four row totals are combined
// from lanes 15 / 31 / 47 / 63.  The result is wave-uniform.  Used by the grouped kernel, whose single wave per SIMD
// pair cannot hide LDS latency.  (Same pairing for max and sum: the sum's association order is fixed, run to run.)
template <typename Op>
__device__ __forceinline__ float wave_reduce_dpp(float v, Op op) {
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));    // quad_perm [1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));    // quad_perm [2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));   // row_half_mirror
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));   // row_mirror
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 31));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 63));
    return op(op(r0, r1), op(r2, r3));
}

// The same reduction for N values at once: every DPP stage is applied to all of them before the next one, so the N
// dependency chains interleave in the instruction stream.  Results are wave-uniform.
template <int N, typename Op>
__device__ __forceinline__ void wave_reduce_dpp_n(float (&v)[N], Op op) {
    auto dpp = [](float x, auto ctrl) {
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = op(v[i], dpp(v[i], std::integral_constant<int, 0xB1>{}));
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = op(v[i], dpp(v[i], std::integral_constant<int, 0x4E>{}));
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = op(v[i], dpp(v[i], std::integral_constant<int, 0x141>{}));
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = op(v[i], dpp(v[i], std::integral_constant<int, 0x140>{}));
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int b = __float_as_int(v[i]);
        const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 15)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 31));
        const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 47)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 63));
        v[i] = op(op(r0, r1), op(r2, r3));
    }
}

// e^x for x <= 0 (softmax arguments; x = -inf gives 0): x log2(e) as an exact product hi + lo, 2^hi on the hardware
// exponential, the low part as a first-order factor — about 2 ulp, 9 instructions, no range reduction (a result below the
// normal range may flush to 0, which for a softmax term next to a term of 1 is 0 anyway).
__device__ __forceinline__ float exp_le0(float x) {
    const float hi = x * 1.44269504088896341f;
    const float lo = fmaf(x, 1.44269504088896341f, -hi) + x * 1.92596299112661746e-8f;
    const float r = __builtin_amdgcn_exp2f(hi);
    return x == -INFINITY ? 0.f : fmaf(r, lo * 0.693147180559945309f, r);
}

// LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land contiguously from the wave-uniform LDS byte address in M0, each
// lane fetching from ITS OWN global address).  Inline asm: M0 is saved and restored inside the statement; completion is
// counted by hand (s_waitcnt vmcnt) before the barrier that precedes the reads.
__device__ __forceinline__ void dma16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

}  // namespace ncf
