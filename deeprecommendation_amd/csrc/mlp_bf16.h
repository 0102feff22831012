// Pieces shared by the bf16 scoring kernels (mlp_bf16.hip: slab-streaming and 4-wave weight-stationary; mlp_bf16_ws8.hip: 8-wave
// weight-stationary).  Internal.
#pragma once
#include "ncf_common.h"

#ifndef NCF_BF16_STAMP
#define NCF_BF16_STAMP 0     // diagnostic builds only (tools/ab_bf16.py): phase stamps written behind the outputs
#endif

namespace ncf {

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
    const unsigned short* Wp1m; const unsigned short* Wp2m;   // the same weights as 16x16x32 A fragments (8-wave kernel), or null
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

typedef short s16x2_t __attribute__((ext_vector_type(2)));
// two 16x16 accumulator tiles (4 + 4 registers) -> one 16x16x32 B fragment: RNE to bf16, ReLU on the packed pairs as int16
__device__ __forceinline__ bf16x8_t pack_relu4x2_int(const f32x4& lo, const f32x4& hi) {
    union { bf16x8_t b; s16x2_t s[4]; } r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4& t = j < 2 ? lo : hi;
        f32x2 v = {t[2 * (j & 1)], t[2 * (j & 1) + 1]};
        union { bf16x2_t b; s16x2_t s; } p;
        p.b = __builtin_convertvector(v, bf16x2_t);
        r.s[j] = __builtin_elementwise_max(p.s, (s16x2_t){0, 0});
    }
    return r.b;
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

// the same fragment with the ReLU applied AFTER the rounding, on the packed pairs as signed 16-bit integers (a negative
// bf16 is a negative int16; rounding keeps the sign, so the result is bit-identical): 4 cvt + 4 v_pk_max_i16
// instead of 16 v_max_f32 + 4 cvt — this matters where the conversion has to hide in MFMA issue gaps
__device__ __forceinline__ bf16x8_t pack_relu8_int(const f32x16& acc, int base) {
    union { bf16x8_t b; s16x2_t s[4]; } r;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        f32x2 v = {acc[base + j], acc[base + j + 1]};
        union { bf16x2_t b; s16x2_t s; } p;
        p.b = __builtin_convertvector(v, bf16x2_t);
        r.s[j >> 1] = __builtin_elementwise_max(p.s, (s16x2_t){0, 0});
    }
    return r.b;
}

typedef __attribute__((address_space(3))) void* lptr_t;

// LDS-DMA by inline asm: the compiler's waitcnt pass treats the builtin form as an out-of-order LGKM event and turns
// every later `s_waitcnt lgkmcnt(N)` of the kernel into lgkmcnt(0) (measured: ~100 stall cycles per k-step at one wave
// per SIMD); hidden in asm, its own LDS reads keep their counted waits and the DMAs are counted by hand (vmcnt).
// M0 (the wave-uniform LDS destination) is saved and restored inside the statement.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void dma4(const void* g, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_addr) : "memory");
}

// ReLU as ONE v_max_i32 on the bit pattern (a negative float is a negative int; -0.0 and negative NaNs become +0.0): the
// fmaxf / fmed3 builtins compile to a canonicalising v_max x,x followed by the max itself, and an inline-asm v_max_f32 hides
// the MFMA -> VALU read hazard from the compiler's hazard recogniser (it read stale accumulators right behind an MFMA).
__device__ __forceinline__ float relu1(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}

// 8-wave weight-stationary kernel (mlp_bf16_ws8.hip); K0 in {128, 256}, MLP K0-256-128-1
bool ws8_shape_ok(int K0, int N1, int N2);
void launch_ws8_bf16(int K0, const Bf16Args& a, const unsigned char* zeros, hipStream_t s);

}  // namespace ncf
