// Workgroup reduction of the packed normal-equation record (21 J^T J + 6 J^T r + cost + count, padded to 32 doubles) shared by
// the scan-to-map and scan-to-scan linearisation kernels.
#pragma once
#include "ctx.hpp"

namespace mlh {

// blockIdx -> tile, XCD-aware: consecutive tiles go to the same XCD (workgroups are dealt round-robin over the 8 XCDs)
__device__ __forceinline__ int xcd_tile(int n_tiles)
{
    const int per = (n_tiles + 7) >> 3;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// One exchange of the transposed butterfly across the wavefront halves (SWAP32) or across neighbouring 16-lane rows (SWAP16), with gfx950's
// v_permlane32_swap / v_permlane16_swap: given x = the value this half keeps if it is the lower one and y = the value it keeps if it is the
// upper one, the swap leaves (own x, partner's x) in the lower lanes and (partner's y, own y) in the upper lanes of the two result registers,
// so ONE add finishes the exchange -- no LDS round trip (ds_bpermute) and no per-lane selects (the keep / send pair costs four v_cndmask
// per double). Same operands as keep + shfl_xor(send), so the sums are bit-identical.
template <bool SWAP32>
__device__ __forceinline__ double swap_add(double x, double y)
{
    const unsigned xl = (unsigned)__double2loint(x), xh = (unsigned)__double2hiint(x), yl = (unsigned)__double2loint(y), yh = (unsigned)__double2hiint(y);
    if constexpr (SWAP32) {
        const auto lo = __builtin_amdgcn_permlane32_swap(xl, yl, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap(xh, yh, false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    } else {
        const auto lo = __builtin_amdgcn_permlane16_swap(xl, yl, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap(xh, yh, false, false);
        return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
}

// acc: this lane's 32 values (entries 29..31 zero). 256 threads. partial_out[0..31]: the workgroup's sums; slots 29 / 30 mirror the
// count for feature kind 0 / 1.
// COH: the record is stored with agent-scope monotonic stores (readable by other workgroups of the SAME launch through agent-scope loads, without an L2 write-back fence)
// MODE 2 (round 6, lm_loop_kernel): the record leaves TAGGED -- every double as two 8-byte words {high half | tag, low half | tag}, agent-scope monotonic stores:
// a reader that finds the iteration's tag in a word has that word's half of the value (8-byte single-copy atomicity), so the record needs no barrier behind it
// (lmc_sum_records_tagged). partial_out then points at the tile's 64 words.
template <int MODE = 0>
__device__ __forceinline__ void reduce_acc32(double (&acc)[32], int kind, double *lds_red /*4*32*/, double *__restrict__ partial_out, unsigned tag = 0u)
{
    constexpr bool COH = MODE == 1;
    // transposed butterfly: at each step a lane keeps one half of its values and trades the other half with its partner, so the
    // wavefront total of value i ends up in lanes 2i and 2i+1 after 16+8+4+2+1+1 = 32 exchanges (a plain per-value butterfly
    // takes 29*6 = 174). Fixed tree -> deterministic sums.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#define MLH_RED_STEP(H, MASK)                                              \
    {                                                                      \
        const bool up = (lane & (MASK)) != 0;                              \
        _Pragma("unroll") for (int i = 0; i < (H); ++i) {                  \
            const double keep = up ? acc[i + (H)] : acc[i];                \
            const double send = up ? acc[i] : acc[i + (H)];                \
            acc[i] = keep + __shfl_xor(send, (MASK));                      \
        }                                                                  \
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = swap_add<true>(acc[i], acc[i + 16]);      // partner = lane ^ 32
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = swap_add<false>(acc[i], acc[i + 8]);        // partner = lane ^ 16
    MLH_RED_STEP(4, 8)
    MLH_RED_STEP(2, 4)
    MLH_RED_STEP(1, 2)
#undef MLH_RED_STEP
    acc[0] += __shfl_xor(acc[0], 1);
    if ((lane & 1) == 0) lds_red[wave * 32 + (lane >> 1)] = acc[0];
    __syncthreads();
    if (threadIdx.x < 32) {
        double v = 0.0;
        const int src = (threadIdx.x == NE_CNT + 1 + kind) ? NE_CNT : threadIdx.x;   // per-kind count mirrors the count column
        if (src < 29) v = ((lds_red[src] + lds_red[32 + src]) + lds_red[64 + src]) + lds_red[96 + src];
        if constexpr (MODE == 2) {
            unsigned long long *out = reinterpret_cast<unsigned long long *>(partial_out) + 2 * threadIdx.x;
            const unsigned long long w0 = ((unsigned long long)(unsigned)__double2hiint(v) << 32) | tag, w1 = ((unsigned long long)(unsigned)__double2loint(v) << 32) | tag;
            __hip_atomic_store(out, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(out + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else if constexpr (COH) __hip_atomic_store(partial_out + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else partial_out[threadIdx.x] = v;
    }
}

// ---- shared by the one-launch Levenberg-Marquardt loops (match.hip: lm_loop_kernel, track.hip: track_lm_loop_kernel)
// the tiles' records summed by EVERY workgroup in sum_partials<TPB, 12>'s order (same slices, same four chains, same association: the same bits everywhere);
// COH: the records were stored with agent-scope monotonic stores by other workgroups of the SAME launch and are read with agent-scope loads
template <bool COH = false>
__device__ __forceinline__ void lmc_sum_records(const double *rec, int ntot, double *f_ne, double *f_scratch)
{
    constexpr int NS = TPB / 32, U = 12;
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double ch[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = sl; j < ntot; j += U * NS) {
        double tv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = j + NS * u;
            if constexpr (COH) tv[u] = jj < ntot ? __hip_atomic_load(rec + size_t(jj) * NE_STRIDE + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            else tv[u] = jj < ntot ? rec[size_t(jj) * NE_STRIDE + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) ch[u & 3] += tv[u];
    }
    f_scratch[sl * 32 + c] = (ch[0] + ch[1]) + (ch[2] + ch[3]);
    __syncthreads();
    if (threadIdx.x < 32) {
        double tsum = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) tsum += f_scratch[q * 32 + c];
        f_ne[c] = tsum;
    }
    __syncthreads();
}

// The same sum over TAGGED records (reduce_acc32<2>): no barrier in front -- every word is polled until it carries this iteration's tag. The workgroups of a loop
// run in step (their loop tops are 0.1-0.3 us apart), so the first sweep finds most words in place and the wait is the last record's store-to-load latency instead of
// an arrival atomic + a poll of its counter + the loads. Same slices, same chains, same association as lmc_sum_records: the same bits. A word that does not arrive
// within timeout_ticks of the 100 MHz wall clock (a workgroup that never became resident, a fault) gives the loop up exactly as the barrier did: counters[3] tells
// the other workgroups, *s_timeout (LDS, read by the caller behind this function's barriers) this one.
#ifndef MLH_LOOP_TAG_SLEEP
#define MLH_LOOP_TAG_SLEEP 1
#endif
__device__ __forceinline__ void lmc_sum_records_tagged(const unsigned long long *rec, int ntot, unsigned tag, double *f_ne, double *f_scratch, unsigned *counters,
                                                       unsigned long long timeout_ticks, int *s_timeout)
{
    constexpr int NS = TPB / 32, U = 12;
    const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
    double ch[4] = {0.0, 0.0, 0.0, 0.0};
    bool gave_up = false;
    for (int j = sl; j < ntot; j += U * NS) {
        unsigned long long w0[U], w1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int jj = j + NS * u;
            const unsigned long long *p = rec + (size_t(jj) * NE_STRIDE + c) * 2;
            w0[u] = jj < ntot ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)tag;
            w1[u] = jj < ntot ? __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (unsigned long long)tag;
        }
        unsigned spins = 0;
        long long t0 = 0;
        while (!gave_up) {
            bool stale = false;
#pragma unroll
            for (int u = 0; u < U; ++u) stale = stale || unsigned(w0[u]) != tag || unsigned(w1[u]) != tag;
            if (!stale) break;
            if (MLH_LOOP_TAG_SLEEP) __builtin_amdgcn_s_sleep(MLH_LOOP_TAG_SLEEP);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + NS * u;
                const unsigned long long *p = rec + (size_t(jj) * NE_STRIDE + c) * 2;
                if (unsigned(w0[u]) != tag) w0[u] = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (unsigned(w1[u]) != tag) w1[u] = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if ((++spins & 63u) == 0u) {
                const long long now = wall_clock64();
                if (t0 == 0) t0 = now;
                const bool late = (unsigned long long)(now - t0) > timeout_ticks;
                if (late) __hip_atomic_store(counters + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (late || __hip_atomic_load(counters + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { gave_up = true; *s_timeout = 1; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) ch[u & 3] += __hiloint2double(int(unsigned(w0[u] >> 32)), int(unsigned(w1[u] >> 32)));
    }
    f_scratch[sl * 32 + c] = (ch[0] + ch[1]) + (ch[2] + ch[3]);
    __syncthreads();
    if (threadIdx.x < 32) {
        double tsum = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) tsum += f_scratch[q * 32 + c];
        f_ne[c] = tsum;
    }
    __syncthreads();
}

// One arrival at the loop kernels' grid barrier, by thread 0 of a workgroup whose record stores have been issued (by lanes of thread 0's own wavefront): wait for
// their acknowledgement, count the arrival, poll until all `total` workgroups of barrier number `nth` (1, 2, ...) have arrived. counters[1] = arrivals (monotonic
// over the launch), counters[2] = departures (loop_barrier_leave), counters[3] = "given up". Returns false when the wait outlasted timeout_ticks of the 100 MHz
// wall clock (a workgroup that never became resident) or another workgroup of the launch has given up; the first to give up says so in counters[3], so that the
// others (and workgroups that only start now: loop_barrier_given_up) do not each wait the limit out.
__device__ __forceinline__ bool loop_barrier_arrive(unsigned *counters, int total, int nth, unsigned long long timeout_ticks)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned target = unsigned(total) * unsigned(nth);
    unsigned spins = 0;
    long long t0 = 0;
    while (__hip_atomic_load(counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0u) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            const bool late = (unsigned long long)(now - t0) > timeout_ticks;
            if (late) __hip_atomic_store(counters + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (late || __hip_atomic_load(counters + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
        }
    }
    asm volatile("" ::: "memory");
    return true;
}
__device__ __forceinline__ bool loop_barrier_given_up(unsigned *counters)
{
    return __hip_atomic_load(counters + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}
// the last workgroup to leave re-arms the counters for the next launch (thread 0 of every participating workgroup, once, at the kernel's end)
__device__ __forceinline__ void loop_barrier_leave(unsigned *counters, int total)
{
    const unsigned left = atomicAdd(counters + 2, 1u);
    if (left == unsigned(total - 1)) {
        __hip_atomic_store(counters + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counters + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(counters + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace mlh
