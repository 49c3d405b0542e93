// Multi-GPU plumbing: map-shard ownership planes and the one collective of the path -- an all-reduce (sum, f64) of the
// packed normal equations per evaluation, over RCCL on the context's stream. librccl is dlopen'ed on first use so the
// single-GPU library has no hard dependency on it. xGMI is point-to-point and this message is 256 B: the collective is
// pure latency, so it is issued exactly once per evaluation, in place on the device-resident solver state.
#include "ctx.hpp"
#include "p2p_dev.hpp"
#include <dlfcn.h>

namespace mlh {

// minimal RCCL surface (rccl.h: ncclUniqueId is 128 bytes; ncclFloat64 = 8, ncclSum = 0)
typedef struct { char internal[128]; } rccl_unique_id;
typedef int (*fn_get_unique_id)(rccl_unique_id *);
typedef int (*fn_comm_init_rank)(void **, int, rccl_unique_id, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_get_error_string)(int);

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_get_error_string get_error_string = nullptr;
    bool ok = false;
};

static Rccl &rccl()
{
    static Rccl r;
    if (!r.handle) {
        // one RCCL per process: a copy that is already mapped (a PyTorch host brings its own librccl.so) is the one to bind -- a second copy
        // next to it would run its own static teardown at exit. Only a process without one loads the ROCm installation's.
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) { r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD); if (r.handle) break; }
        for (const char *n : names) { if (r.handle) break; r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL); }
        if (r.handle) {
            r.get_unique_id = (fn_get_unique_id)dlsym(r.handle, "ncclGetUniqueId");
            r.comm_init_rank = (fn_comm_init_rank)dlsym(r.handle, "ncclCommInitRank");
            r.comm_destroy = (fn_comm_destroy)dlsym(r.handle, "ncclCommDestroy");
            r.all_reduce = (fn_all_reduce)dlsym(r.handle, "ncclAllReduce");
            r.get_error_string = (fn_get_error_string)dlsym(r.handle, "ncclGetErrorString");
            r.ok = r.get_unique_id && r.comm_init_rank && r.comm_destroy && r.all_reduce;
        }
    }
    return r;
}

static int rccl_fail(mlh_ctx *ctx, const char *what, int code)
{
    std::string m = what;
    Rccl &r = rccl();
    if (r.get_error_string) { m += ": "; m += r.get_error_string(code); }
    return fail(ctx, MLH_ERR_HIP, m.c_str());
}

// ---------------------------------------------------------------- the mailbox communicator
// This path's only collective is an all-reduce of a few hundred bytes per evaluation: pure latency. A library collective pays for generality there (protocol
// selection, proxy threads, a ring or tree over the ranks); what the message needs is ONE hop. Every rank owns a mailbox in its device memory, exported through
// hipIpc and mapped by every other rank (peer access over xGMI between GPUs; the same memory through a second mapping when ranks share a GPU). An all-reduce is
// what ONE workgroup per rank does on a record it holds in LDS (p2p_dev.hpp: p2p_exchange) -- the finishing workgroup of a Gauss-Newton / LM launch (match.hip:
// no launch of its own), or the stand-alone kernel below (mlh_allreduce_f64, the good-feature paths) -- on the context's stream:
//   1. store my record into slot [parity][my rank] of EVERY mailbox (system-scope stores: they leave the L2 for the owner's memory),
//   2. fence, then store the all-reduce's sequence number into flag [parity][my rank] of every mailbox,
//   3. wait until the n flags of MY mailbox carry that sequence number (bounded: a missing peer raises the context's device error word instead of hanging),
//   4. sum the n slots of my mailbox in RANK ORDER -- every rank adds the same numbers in the same order, so all ranks hold the same bits and apply the same update.
// Two halves (parity of the sequence number) suffice: a rank can only get one all-reduce ahead of a peer, because finishing all-reduce s + 1 needs that peer's
// contribution to s + 1, which the peer writes after it has read everything of s. Sequence numbers are counted on the device, one per exchange actually made.
// Tear-down is collective: a rank must not free its mailbox (mlh_comm_finalize / mlh_destroy) while a peer may still write into it.
struct P2pArgs {
    P2pDev d;
    double *buf;
    int n;
};

__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2pArgs a)
{
    __shared__ double rec[P2P_MAX_DOUBLES];
    for (int i = threadIdx.x; i < a.n; i += 256) rec[i] = a.buf[i];
    __syncthreads();
    p2p_exchange<256>(a.d, rec, a.n);
    for (int i = threadIdx.x; i < a.n; i += 256) a.buf[i] = rec[i];
}

static int p2p_allreduce(mlh_ctx *ctx, double *buf, int n)
{
    if (n > P2P_MAX_DOUBLES) return fail(ctx, MLH_ERR_UNSUPPORTED, "the mailbox communicator carries records of up to 512 doubles");
    P2pArgs a;
    p2p_fill(ctx, a.d);
    a.buf = buf; a.n = n;
    MLH_LAUNCH(p2p_allreduce_kernel, dim3(1), dim3(256), 0, ctx->stream, a);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// in-place sum over the ranks of n doubles in device memory, on the context's stream: the mailbox communicator when it is set up, RCCL otherwise
static int allreduce_device(mlh_ctx *ctx, double *buf, size_t n)
{
    prof_begin(ctx, MLH_K_ALLREDUCE);
    int rc = MLH_OK;
    if (ctx->p2p.active) rc = p2p_allreduce(ctx, buf, int(n));
    else {
        const int nc = rccl().all_reduce(buf, buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
        if (nc != 0) rc = rccl_fail(ctx, "ncclAllReduce", nc);
    }
    prof_end(ctx, MLH_K_ALLREDUCE);
    return rc;
}

int comm_allreduce_state(mlh_ctx *ctx, int to_ce)
{
    if (!distributed(ctx)) return MLH_OK;
    SolverState *S = ctx->state.as<SolverState>();
    return allreduce_device(ctx, to_ce ? S->ce : S->ne, NE_STRIDE);
}

int comm_allreduce_blocks(mlh_ctx *ctx, int n_blocks)
{
    if (!distributed(ctx)) return MLH_OK;
    SolverState *S = ctx->state.as<SolverState>();
    return allreduce_device(ctx, &S->neb[0][0], size_t(NE_STRIDE) * size_t(n_blocks));
}

void comm_destroy(mlh_ctx *ctx)
{
    if (ctx->comm) { rccl().comm_destroy(ctx->comm); ctx->comm = nullptr; }
    if (ctx->p2p.mailbox || ctx->p2p.active || ctx->p2p.counter) {
        for (int r = 0; r < P2P_MAX_RANKS; ++r) {
            if (ctx->p2p.peer[r] && ctx->p2p.peer[r] != ctx->p2p.mailbox) (void)hipIpcCloseMemHandle(ctx->p2p.peer[r]);
            ctx->p2p.peer[r] = nullptr;
        }
        if (ctx->p2p.mailbox) (void)hipFree(ctx->p2p.mailbox);
        if (ctx->p2p.counter) (void)hipFree(ctx->p2p.counter);
        ctx->p2p.mailbox = nullptr; ctx->p2p.counter = nullptr; ctx->p2p.active = false;
    }
}

}  // namespace mlh

using namespace mlh;

extern "C" {

int mlh_shard_set(mlh_ctx *ctx, const float *lo_plane4, const float *hi_plane4)
{
    if (!ctx) return MLH_ERR_INVALID;
    ctx->shard_lo = lo_plane4 != nullptr;
    ctx->shard_hi = hi_plane4 != nullptr;
    for (int i = 0; i < 4; ++i) {
        ctx->lo_plane[i] = lo_plane4 ? lo_plane4[i] : 0.f;
        ctx->hi_plane[i] = hi_plane4 ? hi_plane4[i] : 0.f;
    }
    return MLH_OK;
}

int mlh_shard_set_features(mlh_ctx *ctx, int n_ranks, int rank)
{
    if (!ctx || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return MLH_ERR_INVALID;
    ctx->own_mod = n_ranks;
    ctx->own_rem = rank;
    return MLH_OK;
}

int mlh_comm_unique_id(void *out_128_bytes)
{
    if (!out_128_bytes) return MLH_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return MLH_ERR_UNSUPPORTED;
    rccl_unique_id id;
    int rc = r.get_unique_id(&id);
    if (rc != 0) return MLH_ERR_HIP;
    std::memcpy(out_128_bytes, &id, sizeof(id));
    return MLH_OK;
}

int mlh_comm_init(mlh_ctx *ctx, int n_ranks, int rank, const void *unique_id_128_bytes)
{
    if (!ctx || n_ranks <= 0 || rank < 0 || rank >= n_ranks || !unique_id_128_bytes) return MLH_ERR_INVALID;
    Rccl &r = rccl();
    if (!r.ok) return fail(ctx, MLH_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    comm_destroy(ctx);
    rccl_unique_id id;
    std::memcpy(&id, unique_id_128_bytes, sizeof(id));
    void *comm = nullptr;
    int rc = r.comm_init_rank(&comm, n_ranks, id, rank);
    if (rc != 0) return rccl_fail(ctx, "ncclCommInitRank", rc);
    ctx->comm = comm;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return MLH_OK;
}

int mlh_comm_finalize(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    comm_destroy(ctx);
    ctx->n_ranks = 1;
    ctx->rank = 0;
    return MLH_OK;
}

int mlh_allreduce_f64(mlh_ctx *ctx, double *host_inout, int n)
{
    if (!ctx || !host_inout || n <= 0) return MLH_ERR_INVALID;
    if (!distributed(ctx)) return MLH_OK;   // single rank: identity
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    // any length: the 32-double record of the map solver, the 326 doubles of the 24-dimensional window problem (mlh_pure_odom_normal_eq), ...
    MLH_HIP(ctx, ctx->allreduce_buf.ensure(sizeof(double) * size_t(n)));
    double *buf = ctx->allreduce_buf.as<double>();
    MLH_HIP(ctx, hipMemcpyAsync(buf, host_inout, sizeof(double) * size_t(n), hipMemcpyHostToDevice, ctx->stream));
    const int rc = allreduce_device(ctx, buf, size_t(n));
    if (rc) return rc;
    MLH_HIP(ctx, hipMemcpyAsync(host_inout, buf, sizeof(double) * size_t(n), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return device_error_check(ctx);
}

int mlh_p2p_mailbox(mlh_ctx *ctx, void *ipc_handle_64_bytes)
{
    if (!ctx || !ipc_handle_64_bytes) return MLH_ERR_INVALID;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the C-ABI promises a 64-byte handle");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    comm_destroy(ctx);
    // Fine-grained device memory when it can be had and exported: another GPU's stores into it are then visible to this GPU's system-scope loads without
    // relying on what an XCD's L2 does with remotely written lines of ordinary (coarse-grained) memory. Ordinary memory otherwise (ranks sharing a device).
    hipIpcMemHandle_t h;
    bool have = false;
    if (!std::getenv("MLH_P2P_COARSE") && hipExtMallocWithFlags(&ctx->p2p.mailbox, sizeof(P2pMailbox), hipDeviceMallocFinegrained) == hipSuccess) {
        if (hipMemset(ctx->p2p.mailbox, 0, sizeof(P2pMailbox)) == hipSuccess && hipIpcGetMemHandle(&h, ctx->p2p.mailbox) == hipSuccess) have = true;
        else { (void)hipFree(ctx->p2p.mailbox); ctx->p2p.mailbox = nullptr; }
    }
    (void)hipGetLastError();
    if (!have) {
        MLH_HIP(ctx, hipMalloc(&ctx->p2p.mailbox, sizeof(P2pMailbox)));
        MLH_HIP(ctx, hipMemset(ctx->p2p.mailbox, 0, sizeof(P2pMailbox)));
        MLH_HIP(ctx, hipIpcGetMemHandle(&h, ctx->p2p.mailbox));
    }
    std::memcpy(ipc_handle_64_bytes, &h, sizeof(h));
    return MLH_OK;
}

int mlh_p2p_comm_init(mlh_ctx *ctx, int n_ranks, int rank, const void *ipc_handles)
{
    if (!ctx || n_ranks <= 0 || n_ranks > P2P_MAX_RANKS || rank < 0 || rank >= n_ranks || !ipc_handles) return MLH_ERR_INVALID;
    if (!ctx->p2p.mailbox) return fail(ctx, MLH_ERR_STATE, "mlh_p2p_mailbox first: the handles passed here are what it returned on every rank");
    if (ctx->p2p.active) return fail(ctx, MLH_ERR_STATE, "the mailbox communicator is already set up (mlh_comm_finalize, then mlh_p2p_mailbox again, to rebuild it)");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    const char *hs = static_cast<const char *>(ipc_handles);
    for (int r = 0; r < n_ranks; ++r) {
        if (r == rank) { ctx->p2p.peer[r] = ctx->p2p.mailbox; continue; }
        hipIpcMemHandle_t h;
        std::memcpy(&h, hs + size_t(r) * sizeof(h), sizeof(h));
        void *p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { comm_destroy(ctx); return fail(ctx, MLH_ERR_HIP, "hipIpcOpenMemHandle of a peer's mailbox", e); }
        ctx->p2p.peer[r] = p;
    }
    MLH_HIP(ctx, hipMalloc(&ctx->p2p.counter, sizeof(unsigned long long)));
    MLH_HIP(ctx, hipMemset(ctx->p2p.counter, 0, sizeof(unsigned long long)));
    ctx->p2p.active = true;
    ctx->n_ranks = n_ranks;
    ctx->rank = rank;
    return MLH_OK;
}

}  // extern "C"
