// VALU issue micro-benchmark (gfx950): packed fp32 ops with a SCALAR (SGPR pair) source against the same ops on VGPR sources —
// the attention score loop feeds pc rows and w1 as scalar operands (attn.hip / attn_split.hip).  Cycles per wave-instruction at
// 1, 2, 4 waves per SIMD (s_memtime around a 4096-instruction unrolled stream of 8 independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed, const float* sc) {
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{seed + i + threadIdx.x, seed - i};
    f32x2 c = {seed * 0.5f, seed * 0.25f};
    f32x2 s0 = {sc[0], sc[1]}, s1 = {sc[2], sc[3]};
    s0[0] = __builtin_amdgcn_readfirstlane(s0[0]); s0[1] = __builtin_amdgcn_readfirstlane(s0[1]);
    s1[0] = __builtin_amdgcn_readfirstlane(s1[0]); s1[1] = __builtin_amdgcn_readfirstlane(s1[1]);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
                if (OP == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x[i]) : "s"(s0));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(x[i]) : "s"(s0));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(x[i]) : "s"(s1));
                if (OP == 5) { asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(x[i]) : "s"(s0)); asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(x[(i + 4) & 7]) : "s"(s1)); }
                if (OP == 6) { asm volatile("v_add_f32 %0, %1, %0" : "+v"(x[i][0]) : "s"(s0[0])); }
                if (OP == 7) { asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(x[i][0]) : "s"(s1[0])); }
                if (OP == 8) { asm volatile("v_pk_add_f32 %0, %0, %1 clamp" : "+v"(x[i]) : "v"(c)); asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(x[(i + 4) & 7]) : "v"(c)); }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += x[i][0] + x[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int OP>
void run(const char* name, int ninstr_per_iter) {
    float* out; unsigned long long* cyc; float* sc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 2048 * 4 * 8); hipMalloc(&sc, 16);
    float hs[4] = {0.5f, 0.25f, 1.0f, 0.75f};
    hipMemcpy(sc, hs, 16, hipMemcpyHostToDevice);
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f, sc);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f, sc);
        hipDeviceSynchronize();
        static unsigned long long h[8192];
        hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < blocks * 4; ++i) sum += h[i];
        double per = sum / (blocks * 4) / (64.0 * ninstr_per_iter);
        printf("%-44s waves/SIMD=%d: %.2f cycles per wave-instruction (wave view), %.2f SIMD-cycles per instruction\n", name, wps, per, per / wps);
    }
}
int main() {
    run<0>("v_pk_add_f32 vgpr", 64); run<1>("v_pk_add_f32 sgpr-pair", 64); run<2>("v_pk_add_f32 sgpr-pair clamp", 64);
    run<3>("v_pk_fma_f32 vgpr", 64); run<4>("v_pk_fma_f32 sgpr-pair", 64); run<5>("pk_add clamp + pk_fma, sgpr-pairs", 128);
    run<8>("pk_add clamp + pk_fma, vgprs", 128); run<6>("v_add_f32 sgpr", 64); run<7>("v_fma_f32 sgpr", 64);
    return 0;
}
