// Register-resident bitonic sorts of 64-bit keys (extract.hip: the sectors' curvature order; voxel.hip: the rings' voxel order).
#pragma once
#include <hip/hip_runtime.h>

namespace mlh {

// Ascending bitonic sort of 64*KPL 64-bit keys held by ONE wavefront: lane l owns elements l*KPL .. l*KPL+KPL-1 in registers.
// Exchange distances below KPL stay inside a lane (static register indices), the others are lane-xor shuffles: no LDS array,
// no workgroup barrier -- the six sectors of a ring sort concurrently on six wavefronts.
template <int KPL>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&v)[KPL], int lane)
{
#pragma unroll
    for (int k = 2; k <= 64 * KPL; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= KPL) {
                const int lx = j / KPL;                          // partner lane = lane ^ lx
                const bool lower = (lane & lx) == 0;
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const bool up = (((lane * KPL + r) & k) == 0);
                    const unsigned long long a = v[r];
                    unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
                    lo = __shfl_xor(lo, lx); hi = __shfl_xor(hi, lx);
                    const unsigned long long b = ((unsigned long long)hi << 32) | lo;
                    const bool take_min = (lower == up);
                    v[r] = take_min ? (a < b ? a : b) : (a < b ? b : a);
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    if ((r & j) == 0) {
                        const bool up = (((lane * KPL + r) & k) == 0);
                        const unsigned long long a = v[r], b = v[r | j];
                        const bool sw = (a > b) == up;
                        v[r] = sw ? b : a;
                        v[r | j] = sw ? a : b;
                    }
                }
            }
        }
    }
}

// The same network over a whole workgroup of WAVES wavefronts: thread t owns elements t*KPL .. t*KPL+KPL-1 in registers. Exchange
// distances below KPL stay inside a thread, those below 64*KPL are lane-xor shuffles inside a wavefront, and only the
// log2(WAVES)(log2(WAVES)+1)/2 stages whose partner sits in another wavefront go through LDS (`xch`: WAVES*64*KPL keys, stored
// r-major so that consecutive lanes touch consecutive words) with a workgroup barrier on either side.
template <int KPL, int WAVES>
__device__ __forceinline__ void block_bitonic_sort(unsigned long long (&v)[KPL], unsigned long long *xch, int tid)
{
    constexpr int T = WAVES * 64;
    const int lane = tid & 63;
#pragma unroll
    for (int k = 2; k <= T * KPL; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * KPL) {
                const int tx = j / KPL;                          // partner thread = tid ^ tx
                const bool lower = (tid & tx) == 0;
#pragma unroll
                for (int r = 0; r < KPL; ++r) xch[r * T + tid] = v[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const bool up = (((tid * KPL + r) & k) == 0);
                    const unsigned long long a = v[r], b = xch[r * T + (tid ^ tx)];
                    const bool take_min = (lower == up);
                    v[r] = take_min ? (a < b ? a : b) : (a < b ? b : a);
                }
                __syncthreads();
            } else if (j >= KPL) {
                const int lx = j / KPL;                          // partner lane = lane ^ lx
                const bool lower = (lane & lx) == 0;
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    const bool up = (((tid * KPL + r) & k) == 0);
                    const unsigned long long a = v[r];
                    unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
                    lo = __shfl_xor(lo, lx); hi = __shfl_xor(hi, lx);
                    const unsigned long long b = ((unsigned long long)hi << 32) | lo;
                    const bool take_min = (lower == up);
                    v[r] = take_min ? (a < b ? a : b) : (a < b ? b : a);
                }
            } else {
#pragma unroll
                for (int r = 0; r < KPL; ++r) {
                    if ((r & j) == 0) {
                        const bool up = (((tid * KPL + r) & k) == 0);
                        const unsigned long long a = v[r], b = v[r | j];
                        const bool sw = (a > b) == up;
                        v[r] = sw ? b : a;
                        v[r | j] = sw ? a : b;
                    }
                }
            }
        }
    }
}

}  // namespace mlh
