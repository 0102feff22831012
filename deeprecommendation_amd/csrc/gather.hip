// K1 — embedding gather (+ fused concat) and gather-dot for gfx950.
//
// HBM-bound byte movement.  A pair's output row is (EA + EB) elements; it is moved as 16-byte chunks with
// consecutive lanes on consecutive chunks of the same row, so one wave-instruction reads whole contiguous
// table rows (256 B rows at E = 64 fp32: 16 lanes x 16 B) and writes a contiguous piece of `out`.
// Algorithmic bytes per pair: (EA + EB) * elt read + 16 B of indices + (EA + EB) * elt written.
#include "ncf_common.h"

#ifndef NCF_G_UNROLL
#define NCF_G_UNROLL 1        // wave steps per wave; measured (cfg 2, A/B): 4 -> 13.3 us, 2 -> 12.8, 1 -> 13.1; with NT stores 2 -> 12.33, 1 -> 12.28
#endif
#ifndef NCF_G_NT_STORE
#define NCF_G_NT_STORE 1      // non-temporal stores for the streamed output rows (13.3 -> 13.0 us alone; NT LOADS are slower: 15.5)
#endif
#ifndef NCF_G_NT_LOAD
#define NCF_G_NT_LOAD 0       // 1: non-temporal loads for the table rows
#endif
#ifndef NCF_G_NT_IDX
#define NCF_G_NT_IDX 0        // 1: non-temporal loads for the (read-once) index arrays
#endif
#ifndef NCF_G_THREADS
#define NCF_G_THREADS 256     // workgroup size of the vector gather kernel
#endif
#ifndef NCF_G_MAXBLOCKS
#define NCF_G_MAXBLOCKS (256 * 128)
#endif

namespace ncf {

// LPP lanes cooperate on one pair; a wave moves 64 / LPP pairs per step and UNROLL steps are in flight.
template <int LPP, int UNROLL>
__global__ __launch_bounds__(NCF_G_THREADS) void gather_concat_vec16(
    const char* __restrict__ tabA, int64_t rowsA, int64_t ldA_bytes,
    const char* __restrict__ tabB, int64_t rowsB, int64_t ldB_bytes,
    const int64_t* __restrict__ idxA, const int64_t* __restrict__ idxB,
    int64_t B, int chunksA, int chunksB, char* __restrict__ out, int64_t ldOut_bytes, int32_t* oob) {
    constexpr int PPW = kWave / LPP;  // pairs per wave step
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPP;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int cpp = chunksA + chunksB;
    for (int64_t base = wave * (PPW * UNROLL); base < B; base += nwaves * (PPW * UNROLL)) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t p = base + u * PPW + lane / LPP;
            if (p >= B) continue;
            int64_t ia = idxA ? (NCF_G_NT_IDX ? __builtin_nontemporal_load(idxA + p) : idxA[p]) : p;
            int64_t ib = chunksB ? (idxB ? (NCF_G_NT_IDX ? __builtin_nontemporal_load(idxB + p) : idxB[p]) : p) : 0;
            const bool okA = (ia >= 0) & (ia < rowsA);
            const bool okB = chunksB == 0 || ((ib >= 0) & (ib < rowsB));
            if (!(okA && okB) && oob && sub == 0) *oob = 1;
            const char* ra = tabA + ia * ldA_bytes;
            const char* rb = tabB + ib * ldB_bytes;
            char* o = out + p * ldOut_bytes;
            for (int c = sub; c < cpp; c += LPP) {
                u32x4 v = {0u, 0u, 0u, 0u};
                const u32x4* src = reinterpret_cast<const u32x4*>(c < chunksA ? ra + (int64_t)c * 16 : rb + (int64_t)(c - chunksA) * 16);
                if (c < chunksA ? okA : okB) v = NCF_G_NT_LOAD ? __builtin_nontemporal_load(src) : *src;
                if (NCF_G_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(o + (int64_t)c * 16));
                else *reinterpret_cast<u32x4*>(o + (int64_t)c * 16) = v;
            }
        }
    }
}

// Persistent form: a fixed population of waves (8 workgroups of 256 threads per CU = every wave slot of the chip) walks
// the batch; a wave keeps THREE steps in flight — the ids of step s+2, the table rows of step s+1 and the store of step s —
// so the dependent chain id -> row -> store of one step overlaps the next two instead of being hidden only by other
// waves, and no wave is launched (or its registers initialised) per kilobyte moved.  Same bytes, same order per row:
// bit-identical output.  Rows of up to 64 chunks (1 KiB).
#ifndef NCF_G_PERSIST_U
#define NCF_G_PERSIST_U 1     // wave steps per pipeline stage of the persistent form; interleaved A/B at cfg 2 on one box (tools/ab_gather_opt.py):
                              // one-step-per-wave kernel 12.70-12.76 us, persistent U = 1 12.60, U = 2 12.56, U = 4 13.24 (registers); on a faster
                              // box 12.20 -> 11.82 (U = 1), 262 144 pairs 48.1 -> 45.1
#endif
template <int LPP>
__global__ __launch_bounds__(NCF_G_THREADS) void gather_concat_persistent(
    const char* __restrict__ tabA, int64_t rowsA, int64_t ldA_bytes,
    const char* __restrict__ tabB, int64_t rowsB, int64_t ldB_bytes,
    const int64_t* __restrict__ idxA, const int64_t* __restrict__ idxB,
    int64_t B, int chunksA, int chunksB, char* __restrict__ out, int64_t ldOut_bytes, int32_t* oob) {
    constexpr int PPW = kWave / LPP;  // pairs per wave step
    constexpr int U = NCF_G_PERSIST_U;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPP;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int cpp = chunksA + chunksB;
    const bool isA = sub < chunksA;
    const bool has = sub < cpp;                              // lanes beyond the row's chunks idle (cpp < LPP)
    const int64_t coff = isA ? (int64_t)sub * 16 : (int64_t)(sub - chunksA) * 16;
    const int64_t step = nwaves * PPW;                       // pairs between two consecutive steps of this wave
    const int64_t stride = step * U;                         // ... and between two pipeline stages
    auto load_id = [&](int64_t p) -> int64_t {               // the id this lane's chunk needs (its table's index array)
        if (p >= B || !has) return -1;
        const int64_t* ix = isA ? idxA : idxB;
        return ix ? ix[p] : p;
    };
    auto load_row = [&](int64_t p, int64_t id, u32x4& v) {
        v = u32x4{0u, 0u, 0u, 0u};
        if (p >= B || !has) return;
        const bool ok = (id >= 0) & (id < (isA ? rowsA : rowsB));
        if (ok) v = *reinterpret_cast<const u32x4*>((isA ? tabA + id * ldA_bytes : tabB + id * ldB_bytes) + coff);
        else if (oob) *oob = 1;
    };
    int64_t p0 = wave * PPW + lane / LPP;                    // this lane's pair of step 0
    int64_t id0[U], id1[U];
    u32x4 v0[U], v1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) id0[u] = load_id(p0 + u * step);
#pragma unroll
    for (int u = 0; u < U; ++u) id1[u] = load_id(p0 + stride + u * step);
#pragma unroll
    for (int u = 0; u < U; ++u) load_row(p0 + u * step, id0[u], v0[u]);
    for (; p0 < B; p0 += stride) {                           // (the wave leaves when its first pair is past the batch: uniform
        int64_t id2[U];                                      //  up to the last partial wave step, whose idle lanes run along)
#pragma unroll
        for (int u = 0; u < U; ++u) id2[u] = load_id(p0 + 2 * stride + u * step);
#pragma unroll
        for (int u = 0; u < U; ++u) load_row(p0 + stride + u * step, id1[u], v1[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = p0 + u * step;
            if (has && p < B) {
                if (NCF_G_NT_STORE) __builtin_nontemporal_store(v0[u], reinterpret_cast<u32x4*>(out + p * ldOut_bytes + (int64_t)sub * 16));
                else *reinterpret_cast<u32x4*>(out + p * ldOut_bytes + (int64_t)sub * 16) = v0[u];
            }
            v0[u] = v1[u];
            id1[u] = id2[u];
        }
    }
}

// Fallback for rows that are not 16-byte tileable / aligned: element-granular, ELT-byte elements.
template <typename T>
__global__ __launch_bounds__(256) void gather_concat_scalar(
    const T* __restrict__ tabA, int64_t rowsA, int64_t ldA, const T* __restrict__ tabB, int64_t rowsB, int64_t ldB,
    const int64_t* __restrict__ idxA, const int64_t* __restrict__ idxB, int64_t B, int EA, int EB,
    T* __restrict__ out, int64_t ldOut, int32_t* oob) {
    const int E = EA + EB;
    const int64_t total = B * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / E;
        const int e = (int)(i - p * E);
        T v = T(0);
        if (e < EA) {
            const int64_t ia = idxA ? idxA[p] : p;
            if (ia >= 0 && ia < rowsA) v = tabA[ia * ldA + e];
            else if (oob) *oob = 1;
        } else {
            const int64_t ib = idxB ? idxB[p] : p;
            if (ib >= 0 && ib < rowsB) v = tabB[ib * ldB + (e - EA)];
            else if (oob) *oob = 1;
        }
        out[p * ldOut + e] = v;
    }
}

// out[p] = <tabA[idxA[p]], tabB[idxB[p]]>, fp32 accumulate.  One pair per 16-lane group.
template <bool BF16>
__global__ __launch_bounds__(256) void gather_dot_kernel(
    const void* __restrict__ tabA_, int64_t rowsA, int64_t ldA, const void* __restrict__ tabB_, int64_t rowsB, int64_t ldB,
    const int64_t* __restrict__ idxA, const int64_t* __restrict__ idxB, int64_t B, int E, float* __restrict__ out,
    int32_t* oob) {
    const int lane = threadIdx.x & 63;
    const int sub = lane & 15;
    const int64_t grp = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int64_t ngrp = ((int64_t)gridDim.x * blockDim.x) >> 4;
    // every lane of a wave runs the same trip count (B rounded up per wave) so the shuffles stay convergent
    const int64_t Bpad = (B + 3) & ~int64_t(3);
    for (int64_t p = grp; p < Bpad; p += ngrp) {
        float acc = 0.f;
        if (p < B) {
            const int64_t ia = idxA ? idxA[p] : p;
            const int64_t ib = idxB ? idxB[p] : p;
            const bool ok = (ia >= 0) & (ia < rowsA) & (ib >= 0) & (ib < rowsB);
            if (!ok && oob && sub == 0) *oob = 1;
            if (ok) {
                if (BF16) {
                    const unsigned short* a = (const unsigned short*)tabA_ + ia * ldA;
                    const unsigned short* b = (const unsigned short*)tabB_ + ib * ldB;
                    for (int e = sub; e < E; e += 16) acc = fmaf(bf16_to_f32(a[e]), bf16_to_f32(b[e]), acc);
                } else {
                    const float* a = (const float*)tabA_ + ia * ldA;
                    const float* b = (const float*)tabB_ + ib * ldB;
                    for (int e = sub; e < E; e += 16) acc = fmaf(a[e], b[e], acc);
                }
            }
        }
        acc += __shfl_xor(acc, 8);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 1);
        if (p < B && sub == 0) out[p] = acc;
    }
}

template <int LPP>
static void launch_vec16(const char* tabA, int64_t rowsA, int64_t ldA_b, const char* tabB, int64_t rowsB, int64_t ldB_b,
                         const int64_t* idxA, const int64_t* idxB, int64_t B, int cA, int cB, char* out, int64_t ldO_b,
                         int32_t* oob, hipStream_t s) {
    constexpr int UNROLL = NCF_G_UNROLL;
    constexpr int PPW = kWave / LPP;
    // Persistent form once the batch gives every resident wave at least two steps; below that (and for rows beyond 64
    // chunks) one step per wave.  ncf_set_option("gather_kernel", 1 | 2) forces one (A/B, tests).
    const int force = option(NCF_OPT_GATHER_KERNEL);
    const int64_t resident_waves = (int64_t)num_cus() * 8 * (NCF_G_THREADS / 64);
    const bool persistent = cA + cB <= 64 && (force ? force == 2 : B >= 2 * resident_waves * PPW);   // >= 2 steps per resident wave
    if (persistent) {
        int64_t blocks = (int64_t)num_cus() * 8;
        const int64_t need = (B + (int64_t)(NCF_G_THREADS / 64) * PPW - 1) / ((int64_t)(NCF_G_THREADS / 64) * PPW);
        if (blocks > need) blocks = need;
        hipLaunchKernelGGL((gather_concat_persistent<LPP>), dim3((unsigned)blocks), dim3(NCF_G_THREADS), 0, s, tabA, rowsA, ldA_b, tabB,
                           rowsB, ldB_b, idxA, idxB, B, cA, cB, out, ldO_b, oob);
        return;
    }
    const int64_t pairs_per_block = (int64_t)(NCF_G_THREADS / 64) * PPW * UNROLL;
    int64_t blocks = (B + pairs_per_block - 1) / pairs_per_block;
    if (blocks > NCF_G_MAXBLOCKS) blocks = NCF_G_MAXBLOCKS;  // grid-stride beyond
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((gather_concat_vec16<LPP, UNROLL>), dim3((unsigned)blocks), dim3(NCF_G_THREADS), 0, s, tabA, rowsA, ldA_b,
                       tabB, rowsB, ldB_b, idxA, idxB, B, cA, cB, out, ldO_b, oob);
}

}  // namespace ncf

using namespace ncf;

extern "C" int ncf_gather_concat(int dtype, const void* tabA, int64_t rowsA, int64_t ldA, const void* tabB,
                                 int64_t rowsB, int64_t ldB, const int64_t* idxA, const int64_t* idxB, int64_t B,
                                 int EA, int EB, void* out, int64_t ldOut, int32_t* oob, ncf_stream_t stream) {
    if (dtype != NCF_F32 && dtype != NCF_BF16) return fail(NCF_EINVAL, "ncf_gather_concat: bad dtype %d", dtype);
    if (B < 0 || EA <= 0 || EB < 0) return fail(NCF_EINVAL, "ncf_gather_concat: bad sizes B=%lld EA=%d EB=%d", (long long)B, EA, EB);
    if (B == 0) return NCF_OK;  // an empty batch is legal and touches nothing (its pointers may be null)
    if (!tabA || !out || (EB > 0 && !tabB)) return fail(NCF_EINVAL, "ncf_gather_concat: null pointer");
    if (ldA < EA || (EB > 0 && ldB < EB) || ldOut < EA + EB) return fail(NCF_EINVAL, "ncf_gather_concat: leading dimension smaller than row");
    if (B == 0) return NCF_OK;
    hipStream_t s = (hipStream_t)stream;
    const int elt = dtype == NCF_F32 ? 4 : 2;
    const bool vec = (EA * elt) % 16 == 0 && (EB * elt) % 16 == 0 && (ldA * elt) % 16 == 0 && (ldOut * elt) % 16 == 0 &&
                     (EB == 0 || (ldB * elt) % 16 == 0) && aligned16(tabA) && aligned16(out) && (EB == 0 || aligned16(tabB));
    if (vec) {
        const int cA = EA * elt / 16, cB = EB * elt / 16, cpp = cA + cB;
        const char* a = (const char*)tabA;
        const char* b = (const char*)(EB ? tabB : tabA);
        char* o = (char*)out;
        if (cpp <= 8) launch_vec16<8>(a, rowsA, ldA * elt, b, rowsB, ldB * elt, idxA, idxB, B, cA, cB, o, ldOut * elt, oob, s);
        else if (cpp <= 16) launch_vec16<16>(a, rowsA, ldA * elt, b, rowsB, ldB * elt, idxA, idxB, B, cA, cB, o, ldOut * elt, oob, s);
        else if (cpp <= 32) launch_vec16<32>(a, rowsA, ldA * elt, b, rowsB, ldB * elt, idxA, idxB, B, cA, cB, o, ldOut * elt, oob, s);
        else launch_vec16<64>(a, rowsA, ldA * elt, b, rowsB, ldB * elt, idxA, idxB, B, cA, cB, o, ldOut * elt, oob, s);
    } else {
        int64_t total = B * (EA + EB);
        int64_t blocks = (total + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (dtype == NCF_F32)
            hipLaunchKernelGGL(gather_concat_scalar<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)tabA, rowsA,
                               ldA, (const float*)tabB, rowsB, ldB, idxA, idxB, B, EA, EB, (float*)out, ldOut, oob);
        else
            hipLaunchKernelGGL(gather_concat_scalar<unsigned short>, dim3((unsigned)blocks), dim3(256), 0, s,
                               (const unsigned short*)tabA, rowsA, ldA, (const unsigned short*)tabB, rowsB, ldB, idxA, idxB, B,
                               EA, EB, (unsigned short*)out, ldOut, oob);
    }
    return check_launch("ncf_gather_concat");
}

extern "C" int ncf_gather_dot(int dtype, const void* tabA, int64_t rowsA, int64_t ldA, const void* tabB, int64_t rowsB,
                              int64_t ldB, const int64_t* idxA, const int64_t* idxB, int64_t B, int E, float* out,
                              int32_t* oob, ncf_stream_t stream) {
    if (dtype != NCF_F32 && dtype != NCF_BF16) return fail(NCF_EINVAL, "ncf_gather_dot: bad dtype %d", dtype);
    if (B == 0) return NCF_OK;
    if (B < 0 || E <= 0 || !tabA || !tabB || !out) return fail(NCF_EINVAL, "ncf_gather_dot: bad argument");
    if (ldA < E || ldB < E) return fail(NCF_EINVAL, "ncf_gather_dot: leading dimension smaller than row");
    if (B == 0) return NCF_OK;
    int64_t blocks = (B + 15) / 16;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == NCF_F32)
        hipLaunchKernelGGL(gather_dot_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, tabA, rowsA, ldA, tabB, rowsB, ldB,
                           idxA, idxB, B, E, out, oob);
    else
        hipLaunchKernelGGL(gather_dot_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, tabA, rowsA, ldA, tabB, rowsB, ldB,
                           idxA, idxB, B, E, out, oob);
    return check_launch("ncf_gather_dot");
}
