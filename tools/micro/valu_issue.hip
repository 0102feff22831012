// VALU issue micro-benchmark (gfx950): cycles per wave-instruction of v_max_f32 / v_fma_f32 / v_pk_add_f32 / v_pk_fma_f32 at 1, 2, 4
// waves per SIMD (s_memtime around a 4096-instruction unrolled stream of 8 independent chains).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, float seed) {
    f32x2 x[8];
    for (int i = 0; i < 8; ++i) x[i] = f32x2{seed + i + threadIdx.x, seed - i};
    f32x2 c = {seed * 0.5f, seed * 0.25f};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll 1
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[i][0]) : "v"(c[0]));
                if (OP == 1) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(x[i][0]) : "v"(c[0]));
                if (OP == 2) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(x[i]) : "v"(c));
                if (OP == 4) { asm volatile("v_pk_add_f32 %0, %1, %0 clamp" : "+v"(x[i]) : "v"(c)); }
                if (OP == 5) { asm volatile("v_max_f32 %0, %1, %0" : "+v"(x[i][0]) : "v"(c[0])); asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(x[(i + 4) & 7]) : "v"(c)); }
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
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    for (int wps = 1; wps <= 4; wps *= 2) {
        int blocks = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.5f);
        hipDeviceSynchronize();
        static unsigned long long h[8192];
        hipMemcpy(h, cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (int i = 0; i < blocks * 4; ++i) sum += h[i];
        double per = sum / (blocks * 4) / (64.0 * ninstr_per_iter);
        printf("%-28s waves/SIMD=%d: %.2f cycles per wave-instruction (wave view), %.2f SIMD-cycles per instruction\n", name, wps, per, per / wps);
    }
}
int main() {
    run<0>("v_max_f32", 64); run<1>("v_fma_f32", 64); run<2>("v_pk_add_f32", 64); run<3>("v_pk_fma_f32", 64);
    run<4>("v_pk_add_f32 clamp", 64); run<5>("v_max + v_pk_fma pairs", 128);
    return 0;
}
