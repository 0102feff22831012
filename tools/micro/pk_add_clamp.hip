#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* __restrict__ a, const float* __restrict__ b, float* out, int n) {
    int i = threadIdx.x;
    f32x2 p = {a[2 * i], a[2 * i + 1]};
    f32x2 q = {b[0], b[1]};   // uniform -> SGPR
    f32x2 u;
    asm volatile("v_pk_add_f32 %0, %1, %2 clamp" : "=v"(u) : "v"(p), "s"(q));
    out[2 * i] = u[0];
    out[2 * i + 1] = u[1];
    f32x2 w = {b[2], b[3]};
    f32x2 s = {1.0f, 2.0f};
    f32x2 r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(u), "s"(w), "v"(s));
    out[128 + 2 * i] = r[0];
    out[128 + 2 * i + 1] = r[1];
}
int main() {
    float ha[128], hb[4] = {0.25f, -0.5f, 3.0f, 5.0f}, ho[256];
    for (int i = 0; i < 128; ++i) ha[i] = (i - 64) * 0.05f;
    ha[5] = __builtin_nanf(""); ha[6] = 1e30f; ha[7] = -1e30f;
    float *a, *b, *o;
    hipMalloc(&a, sizeof(ha)); hipMalloc(&b, sizeof(hb)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(a, ha, sizeof(ha), hipMemcpyHostToDevice); hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, 128);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 128; ++i) {
        float x = ha[i] + hb[i & 1];
        float ref = x != x ? 0.f : (x < 0 ? 0.f : (x > 1 ? 1.f : x));
        if (ho[i] != ref) { printf("clamp mismatch i=%d in=%g got=%g ref=%g\n", i, x, ho[i], ref); ++bad; }
        float rr = __builtin_fmaf(ref, hb[2 + (i & 1)], (i & 1) ? 2.0f : 1.0f);
        if (ho[128 + i] != rr) { printf("fma mismatch i=%d got=%g ref=%g\n", i, ho[128 + i], rr); ++bad; }
    }
    printf("bad=%d  sample: %g %g %g %g nan->%g big->%g\n", bad, ho[0], ho[64], ho[70], ho[127], ho[5], ho[6]);
    return bad != 0;
}
