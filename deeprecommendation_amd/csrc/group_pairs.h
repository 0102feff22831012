// Pairs listed row by row for the grouped attention kernels: the single-workgroup form of the counting sort (ncf_group_pairs), as a
// device function so that another kernel's spare workgroup can run it (attn_cand.hip: the grouping of a batch depends on pair_row
// only, and hides under the candidate projection instead of being a launch of its own).  Internal.
#pragma once
#include "ncf_common.h"

namespace ncf {

// Wave-aggregated atomics: the lanes of a wave that target the same row elect a leader which adds their count once and
// hands every lane its rank.  One user scored against a whole catalogue (the web backend's call: 65 536 pairs, ONE row)
// otherwise serialises 65 536 atomics on a single counter in each pass (~0.65 ms each, measured 2.0 ms per request).
// Returns the value the lane's own atomicAdd(&counter[r], 1) would have returned in SOME valid order; inactive lanes
// (valid == false) take no part.
__device__ __forceinline__ int wave_aggregated_inc(int* __restrict__ counter, int64_t r, bool valid, bool aggregate = true) {
    int result = 0;
    bool pending = valid;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int round = 0; round < 2 && aggregate; ++round) {  // two rounds take the one or two hot rows of a wave ...
        const unsigned long long todo = __ballot(pending);
        if (todo == 0) break;
        const int leader = __ffsll((long long)todo) - 1;
        const int64_t r0 = __shfl(r, leader, 64);
        const unsigned long long same = __ballot(pending && r == r0);
        int base = 0;
        if (lane == leader) base = atomicAdd(&counter[r0], __popcll(same));
        base = __shfl(base, leader, 64);
        if (pending && r == r0) {
            result = base + __popcll(same & ((1ull << lane) - 1ull));
            pending = false;
        }
    }
    if (pending) result = atomicAdd(&counter[r], 1);        // ... lanes on other rows add in parallel as before
    return result;
}


constexpr int kGroupLdsRows = 4096;   // rows whose counters / cursors fit LDS

template <bool LDS_COUNTERS>
constexpr int group_small_lds_ints(int nthreads) {
    return (LDS_COUNTERS ? 2 * kGroupLdsRows : 2) + 2 * (nthreads / 64) + 2;
}

// The three steps (count, scan, scatter) in ONE NT-thread workgroup (an evaluation batch is a few thousand pairs: three launches
// and a memset cost more than the work).  Counters and cursors live in LDS when the rows fit (LDS atomics: a batch of 4096 pairs
// over 64 users puts 64 increments on each counter — 20 us with global atomics, 4 with LDS ones).  B, R <= 32768 (32-bit sums).
// `lds`: group_small_lds_ints<LDS_COUNTERS>(NT) ints.  Every thread of the workgroup must call it (barriers inside).
template <bool LDS_COUNTERS, int NT>
__device__ __forceinline__ void group_small_body(const int64_t* __restrict__ pair_row, int64_t B, int64_t R, int ppw,
                                                 int* __restrict__ gcounts, int* __restrict__ gcursor, int* __restrict__ bad,
                                                 int64_t* __restrict__ grp_ptr, int64_t* __restrict__ wg_ptr,
                                                 int64_t* __restrict__ pair_ids, int32_t* __restrict__ wg_row, int* lds) {
    constexpr int NWV = NT / 64;
    // pairs spread over many rows (an evaluation batch in random user order) rarely meet in a wave: the aggregation rounds would be
    // ~80 instructions of pure overhead per 64 pairs; they pay when a few rows hold most pairs (one user x a whole catalogue)
    const bool aggregate = R <= 16;                        // >= 4 lanes of a wave per row on average
    if (B <= 0) {                                           // nothing to list: empty groups
        for (int64_t i = threadIdx.x; i <= R; i += NT) { grp_ptr[i] = 0; wg_ptr[i] = 0; }
        return;
    }
    int* counts = LDS_COUNTERS ? lds : gcounts;
    int* cursor = LDS_COUNTERS ? lds + kGroupLdsRows : gcursor;
    int* sa = lds + (LDS_COUNTERS ? 2 * kGroupLdsRows : 2);
    int* sb = sa + NWV;
    int* carry = sb + NWV;                                  // [2]
    for (int64_t i = threadIdx.x; i < R; i += NT) counts[i] = 0;
    __syncthreads();
    // The rows of U iterations are loaded FIRST, back to back (clamped index, value selected afterwards): one memory round trip per
    // U x NT pairs instead of one per NT — as a spare workgroup beside a bandwidth-bound kernel a round trip is microseconds.  The
    // first U x NT pairs (a whole evaluation batch) stay in registers for the scatter pass.
    constexpr int U = 8;
    int64_t rr0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t b = (int64_t)u * NT + threadIdx.x;
        rr0[u] = pair_row[b < B ? b : B - 1];
    }
    for (int64_t base = 0; base < B; base += (int64_t)U * NT) {   // uniform trip counts: the aggregated atomic is a wave-level operation
        int64_t rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t b = base + (int64_t)u * NT + threadIdx.x;
            rr[u] = base == 0 ? rr0[u] : pair_row[b < B ? b : B - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t b = base + (int64_t)u * NT + threadIdx.x;
            if (base + (int64_t)u * NT >= B) break;
            const int64_t r = b < B ? rr[u] : -1;
            const bool ok = b < B && r >= 0 && r < R;
            if (b < B && !ok) *bad = 1;
            (void)wave_aggregated_inc(counts, r, ok, aggregate);
        }
    }
    __syncthreads();
    // Exclusive scans of the row counts and of the per-row workgroup counts.  Each thread owns a contiguous run of rows: serial
    // sums over its run, ONE block scan of the NT run totals (wave scans by shuffles + a scan of the wave totals: three barriers
    // whatever R is), then the run is written out.
    {
        const int per = (int)((R + NT - 1) / NT);
        const int64_t lo = (int64_t)threadIdx.x * per;
        const int64_t hi = lo + per < R ? lo + per : R;
        int ta = 0, tb = 0;
        for (int64_t i = lo; i < hi; ++i) {
            const int c = counts[i];
            ta += c;
            tb += (c + ppw - 1) / ppw;
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int ia = ta, ib = tb;                                   // inclusive scan inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int ua = __shfl_up(ia, off, 64), ub = __shfl_up(ib, off, 64);
            if (lane >= off) { ia += ua; ib += ub; }
        }
        if (lane == 63) { sa[wave] = ia; sb[wave] = ib; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int ca = 0, cb = 0;
            for (int w = 0; w < NWV; ++w) {                     // exclusive scan of the wave totals
                const int va = sa[w], vb = sb[w];
                sa[w] = ca; sb[w] = cb;
                ca += va; cb += vb;
            }
            carry[0] = ca; carry[1] = cb;
        }
        __syncthreads();
        int ea = sa[wave] + ia - ta, eb = sb[wave] + ib - tb;   // exclusive prefix of this thread's run
        for (int64_t i = lo; i < hi; ++i) {
            const int c = counts[i];
            grp_ptr[i] = ea;
            wg_ptr[i] = eb;
            cursor[i] = ea;
            const int wgs = (c + ppw - 1) / ppw;
            if (wg_row)
                for (int w = 0; w < wgs; ++w) wg_row[eb + w] = (int32_t)i;   // workgroup -> row, read instead of a binary search
            ea += c;
            eb += wgs;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { grp_ptr[R] = carry[0]; wg_ptr[R] = carry[1]; }
    for (int64_t base = 0; base < B; base += (int64_t)U * NT) {
        int64_t rr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t b = base + (int64_t)u * NT + threadIdx.x;
            rr[u] = base == 0 ? rr0[u] : pair_row[b < B ? b : B - 1];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t b = base + (int64_t)u * NT + threadIdx.x;
            if (base + (int64_t)u * NT >= B) break;
            const int64_t r = b < B ? rr[u] : -1;
            const bool ok = b < B && r >= 0 && r < R;
            const int slot = wave_aggregated_inc(cursor, r, ok, aggregate);
            if (ok) pair_ids[slot] = b;
        }
    }
}

}  // namespace ncf
