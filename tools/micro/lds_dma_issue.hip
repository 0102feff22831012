// LDS-DMA issue micro-benchmark: one or four waves per CU issue 32 gather pieces (1 KiB each: 2 rows x 512 B) from a 51 MB table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int VAR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ tab, const int* __restrict__ rows, unsigned long long* cyc, int npieces, int rowfloats) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) void* lp;
    const unsigned lds0 = (unsigned)(size_t)(lp)smem;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int* myrows = rows + (blockIdx.x * 4 + wave) * npieces * 2;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int p = 0; p < npieces; ++p) {
        const int r = myrows[2 * p + (lane >> 5)];     // uniform per half wave
        const float* src = tab + (long long)r * rowfloats + 4 * (lane & 31);
        const unsigned dst = lds0 + (unsigned)(wave * 8 + (p & 7)) * 1024u;
        if (VAR == 0) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        } else if (VAR == 1) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
        } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (lp)(smem + (wave * 8 + (p & 7)) * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (lane == 0) { cyc[(blockIdx.x * 4 + wave) * 2] = t1 - t0; cyc[(blockIdx.x * 4 + wave) * 2 + 1] = t2 - t0; }
}
int main() {
    const int I = 100000, A = 128, NP = 32, BLK = 256;
    float* tab; hipMalloc(&tab, (size_t)I * A * 4); hipMemset(tab, 0, (size_t)I * A * 4);
    int* rows; int* hrows = (int*)malloc(BLK * 4 * NP * 2 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, BLK * 4 * 2 * 8); hipMalloc(&rows, BLK * 4 * NP * 2 * 4);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (int i = 0; i < BLK * 4 * NP * 2; ++i) hrows[i] = mode == 0 ? rand() % I : (i % I);
        hipMemcpy(rows, hrows, BLK * 4 * NP * 2 * 4, hipMemcpyHostToDevice);
        for (int threads = 64; threads <= 256; threads *= 4) {
            for (int var = 0; var < 3; ++var) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (var == 0) hipLaunchKernelGGL(k<0>, dim3(BLK), dim3(threads), 32768, 0, tab, rows, cyc, NP, A);
                    if (var == 1) hipLaunchKernelGGL(k<1>, dim3(BLK), dim3(threads), 32768, 0, tab, rows, cyc, NP, A);
                    if (var == 2) hipLaunchKernelGGL(k<2>, dim3(BLK), dim3(threads), 32768, 0, tab, rows, cyc, NP, A);
                }
                hipDeviceSynchronize();
                static unsigned long long h[BLK * 4 * 2];
                hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
                double a = 0, b = 0; int n = 0;
                for (int blk = 0; blk < BLK; ++blk) for (int w = 0; w < threads / 64; ++w) { a += h[(blk * 4 + w) * 2]; b += h[(blk * 4 + w) * 2 + 1]; ++n; }
                printf("%s rows, %d wave(s)/CU, variant %d (%s): issue %.0f cycles per piece, issue+land %.0f per piece\n", mode == 0 ? "random" : "sequential",
                       threads / 64, var, var == 0 ? "asm m0 save/restore" : var == 1 ? "asm m0 clobber" : "builtin", a / n / NP, b / n / NP);
            }
        }
    }
    return 0;
}
