// Multi-GPU plumbing: map-shard ownership planes and the one collective of the path -- an all-reduce (sum, f64) of the
// packed normal equations per evaluation, over RCCL on the context's stream. librccl is dlopen'ed on first use so the
// single-GPU library has no hard dependency on it. xGMI is point-to-point and this message is 256 B: the collective is
// pure latency, so it is issued exactly once per evaluation, in place on the device-resident solver state.
#include "ctx.hpp"
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

int comm_allreduce_state(mlh_ctx *ctx, int to_ce)
{
    if (!ctx->comm) return MLH_OK;
    SolverState *S = ctx->state.as<SolverState>();
    double *buf = to_ce ? S->ce : S->ne;
    prof_begin(ctx, MLH_K_ALLREDUCE);
    int rc = rccl().all_reduce(buf, buf, NE_STRIDE, /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    prof_end(ctx, MLH_K_ALLREDUCE);
    if (rc != 0) return rccl_fail(ctx, "ncclAllReduce", rc);
    return MLH_OK;
}

int comm_allreduce_blocks(mlh_ctx *ctx, int n_blocks)
{
    if (!ctx->comm) return MLH_OK;
    SolverState *S = ctx->state.as<SolverState>();
    double *buf = &S->neb[0][0];
    prof_begin(ctx, MLH_K_ALLREDUCE);
    int rc = rccl().all_reduce(buf, buf, size_t(NE_STRIDE) * size_t(n_blocks), /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    prof_end(ctx, MLH_K_ALLREDUCE);
    if (rc != 0) return rccl_fail(ctx, "ncclAllReduce", rc);
    return MLH_OK;
}

void comm_destroy(mlh_ctx *ctx)
{
    if (ctx->comm) { rccl().comm_destroy(ctx->comm); ctx->comm = nullptr; }
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
    if (!ctx->comm) return MLH_OK;   // single rank: identity
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    // any length: the 32-double record of the map solver, the 326 doubles of the 24-dimensional window problem (mlh_pure_odom_normal_eq), ...
    MLH_HIP(ctx, ctx->allreduce_buf.ensure(sizeof(double) * size_t(n)));
    double *buf = ctx->allreduce_buf.as<double>();
    MLH_HIP(ctx, hipMemcpyAsync(buf, host_inout, sizeof(double) * size_t(n), hipMemcpyHostToDevice, ctx->stream));
    prof_begin(ctx, MLH_K_ALLREDUCE);
    int rc = rccl().all_reduce(buf, buf, size_t(n), /*ncclFloat64*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    prof_end(ctx, MLH_K_ALLREDUCE);
    if (rc != 0) return rccl_fail(ctx, "ncclAllReduce", rc);
    MLH_HIP(ctx, hipMemcpyAsync(host_inout, buf, sizeof(double) * size_t(n), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLH_OK;
}

}  // extern "C"
