// The mailbox communicator's device side (comm.hip has the protocol's description and the host side): the mailbox layout and the exchange itself, as a
// function a workgroup calls on a record it holds in LDS -- the stand-alone all-reduce kernel (comm.hip) and the fit kernel's finishing workgroup (match.hip:
// a sharded Gauss-Newton iteration is then the same two launches as an unsharded one, with one hop to the peers inside the second) both run it.
#pragma once
#include "ctx.hpp"

namespace mlh {

constexpr int P2P_MAX_RANKS = 16, P2P_MAX_DOUBLES = 512;
struct P2pMailbox {
    double slot[2][P2P_MAX_RANKS][P2P_MAX_DOUBLES];
    unsigned long long flag[2][P2P_MAX_RANKS];
};
struct P2pDev {
    P2pMailbox *peer[P2P_MAX_RANKS];     // rank r's mailbox as this process sees it
    int n_ranks, rank;                   // n_ranks <= 1: no exchange
    unsigned long long *counter;         // device word: exchanges this rank has COMPLETED. The next one's sequence number is *counter + 1 -- counted where the
                                         // exchanges happen, not where launches are enqueued: a launch that finds the LM loop already terminated exchanges nothing
                                         // (on every rank alike), and the parity argument below needs consecutive numbers for consecutive exchanges
    int *err;                            // pinned host word: 2 = a peer did not show up within the bound
};

// rec[0 .. n) (LDS) <- sum over the ranks, added in rank order. All NT threads of the workgroup, converged; n <= P2P_MAX_DOUBLES.
// Returns false (uniform over the workgroup) when a peer did not show up within the bound: the pinned error word is 2 then, rec is NOT a sum over the job, and
// the caller must not solve with it -- the host reports the failure from every solve entry point (device_error_check).
template <int NT>
__device__ __forceinline__ bool p2p_exchange(const P2pDev &a, double *rec, int n)
{
    __shared__ int s_timeout;
    if (threadIdx.x == 0) s_timeout = 0;
    __syncthreads();
    const unsigned long long seq = __hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1ull;
    const int t = threadIdx.x, par = int(seq & 1ull);
    for (int r = 0; r < a.n_ranks; ++r)
        for (int i = t; i < n; i += NT) __hip_atomic_store(&a.peer[r]->slot[par][a.rank][i], rec[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (t < a.n_ranks) __hip_atomic_store(&a.peer[t]->flag[par][a.rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    P2pMailbox *mine = a.peer[a.rank];
    if (t < a.n_ranks) {
        const unsigned long long t0 = wall_clock64();
        unsigned spins = 0;
        while (__hip_atomic_load(&mine->flag[par][t], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 1023u) == 0 && wall_clock64() - t0 > 500000000ull) {        // 5 s at 100 MHz: the peer is not coming
                if (a.err) __hip_atomic_store(a.err, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&s_timeout, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
        }
    }
    __syncthreads();
    for (int i = t; i < n; i += NT) {
        double s = __hip_atomic_load(&mine->slot[par][0][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 1; r < a.n_ranks; ++r) s += __hip_atomic_load(&mine->slot[par][r][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        rec[i] = s;
    }
    __syncthreads();                     // (every thread has read the counter long before this point)
    if (t == 0) __hip_atomic_store(a.counter, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool ok = __hip_atomic_load(&s_timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == 0;      // into a register BEFORE the last barrier: the caller may
    __syncthreads();                                                                                           // re-enter at once and thread 0 clears the flag again
    return ok;
}

// fills the device-side descriptor from the context
inline void p2p_fill(mlh_ctx *ctx, P2pDev &d)
{
    for (int r = 0; r < P2P_MAX_RANKS; ++r) d.peer[r] = static_cast<P2pMailbox *>(ctx->p2p.peer[r]);
    d.n_ranks = ctx->p2p.active ? ctx->n_ranks : 1;
    d.rank = ctx->rank;
    d.counter = static_cast<unsigned long long *>(ctx->p2p.counter);
    d.err = ctx->p2p.active ? device_error_word(ctx) : nullptr;
}

}  // namespace mlh
