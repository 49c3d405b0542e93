// extern "C" surface of libmloam_hip.so (see include/mloam_hip.h for the contract and the reference interfaces each
// entry point replaces). Host logic only: staging, launch sequencing, result unpacking.
#include "ctx.hpp"
#include <chrono>
#include <cstdlib>
#include "dev_math.hpp"
#include <algorithm>
#include <cmath>
#include <new>
#include <random>

namespace mlh {

int fail(mlh_ctx *ctx, int code, const char *what, hipError_t e)
{
    if (ctx) {
        ctx->err = what ? what : "";
        if (e != hipSuccess) { ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    }
    return code;
}

bool launch_check_enabled()
{
    static const bool on = std::getenv("MLH_CHECK_LAUNCH") && std::atoi(std::getenv("MLH_CHECK_LAUNCH")) != 0;
    return on;
}
namespace { thread_local hipError_t t_launch_err = hipSuccess; thread_local const char *t_launch_kernel = nullptr; }
void launch_check(const char *kernel)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess && t_launch_err == hipSuccess) { t_launch_err = e; t_launch_kernel = kernel; }     // the FIRST failing launch is the one to name
}
hipError_t launch_check_take(const char **kernel)
{
    const hipError_t e = t_launch_err;
    if (kernel) *kernel = t_launch_kernel;
    t_launch_err = hipSuccess; t_launch_kernel = nullptr;
    return e;
}
int fail_launch(mlh_ctx *ctx, const char *kernel, hipError_t e)
{
    if (ctx) { ctx->err = "launch of "; ctx->err += kernel ? kernel : "?"; ctx->err += ": "; ctx->err += hipGetErrorString(e); }
    return MLH_ERR_HIP;
}

static hipEvent_t prof_event(mlh_ctx *ctx)
{
    Profile &p = ctx->prof;
    if (!p.pool.empty()) { hipEvent_t e = p.pool.back(); p.pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void prof_begin(mlh_ctx *ctx, int id)
{
    if (!(ctx->prof.mask & (1u << id))) return;
    Profile::Pending pd;
    pd.id = id; pd.a = prof_event(ctx); pd.b = nullptr;
    (void)hipEventRecord(pd.a, ctx->stream);
    ctx->prof.pending.push_back(pd);
}

void prof_end(mlh_ctx *ctx, int id)
{
    if (!(ctx->prof.mask & (1u << id)) || ctx->prof.pending.empty()) return;
    Profile::Pending &pd = ctx->prof.pending.back();
    if (pd.id != id || pd.b) return;
    pd.b = prof_event(ctx);
    (void)hipEventRecord(pd.b, ctx->stream);
}

bool prof_kernel_events(mlh_ctx *ctx, int id, hipEvent_t *start, hipEvent_t *stop)
{
    if (!(ctx->prof.mask & (1u << id))) return false;
    if ((ctx->prof.seen[id]++ % ctx->prof.every) != 0) return false;
    Profile::Pending pd;
    pd.id = id; pd.a = prof_event(ctx); pd.b = prof_event(ctx);
    if (!pd.a || !pd.b) return false;
    ctx->prof.pending.push_back(pd);
    *start = pd.a; *stop = pd.b;
    return true;
}

void prof_collect(mlh_ctx *ctx)
{
    Profile &p = ctx->prof;
    for (auto &pd : p.pending) {
        if (pd.a && pd.b) {
            float ms = 0.f;
            if (hipEventSynchronize(pd.b) == hipSuccess && hipEventElapsedTime(&ms, pd.a, pd.b) == hipSuccess) {
                p.total_ms[pd.id] += ms;
                p.launches[pd.id] += 1;
            }
        }
        if (pd.a) p.pool.push_back(pd.a);
        if (pd.b) p.pool.push_back(pd.b);
    }
    p.pending.clear();
}

// AoS records (stride bytes) -> float4 {x,y,z,w}; w = f32 at w_off, or the record index (as int bits) when w_off == -2, or 0
__global__ __launch_bounds__(256) void pack_points_kernel(const unsigned char *__restrict__ src, int stride, int n, int w_off,
                                                          int cov_off, float4 *__restrict__ out, float4 *__restrict__ covd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *rec = reinterpret_cast<const float *>(src + size_t(i) * stride);
    float4 p;
    p.x = rec[0]; p.y = rec[1]; p.z = rec[2];
    if (w_off == -2) p.w = __int_as_float(i);
    else if (w_off >= 0) p.w = *reinterpret_cast<const float *>(src + size_t(i) * stride + w_off);
    else p.w = 0.f;
    out[i] = p;
    if (covd) {
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (cov_off >= 0) {
            const float *cv = reinterpret_cast<const float *>(src + size_t(i) * stride + cov_off);
            c.x = cv[0]; c.y = cv[3]; c.z = cv[5];
        }
        covd[i] = c;
    }
}

static int stage_points(mlh_ctx *ctx, const void *points, int stride, int n, int mem, int w_off, int cov_off, DevBuf &dst, DevBuf *covd,
                        DevBuf &tmp)
{
    if (!points || n <= 0 || stride < 12 || (stride & 3)) return fail(ctx, MLH_ERR_INVALID, "bad point buffer (null, n <= 0, or stride not a multiple of 4 >= 12)");
    MLH_HIP(ctx, dst.ensure(sizeof(float4) * size_t(n)));
    if (covd) MLH_HIP(ctx, covd->ensure(sizeof(float4) * size_t(n)));
    const unsigned char *src = static_cast<const unsigned char *>(points);
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, tmp.ensure(size_t(n) * stride));
        MLH_HIP(ctx, hipMemcpyAsync(tmp.p, points, size_t(n) * stride, hipMemcpyHostToDevice, ctx->stream));
        src = tmp.as<unsigned char>();
    }
    MLH_LAUNCH(pack_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, src, stride, n, w_off, cov_off,
                       dst.as<float4>(), covd ? covd->as<float4>() : nullptr);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

static int ensure_state(mlh_ctx *ctx, int n_stats)
{
    if (!ctx->state.p) {
        MLH_HIP(ctx, ctx->state.ensure(sizeof(SolverState)));
        MLH_HIP(ctx, hipMemsetAsync(ctx->state.p, 0, sizeof(SolverState), ctx->stream));
    }
    if (n_stats > 0) MLH_HIP(ctx, ctx->stats.ensure(sizeof(IterStatDev) * size_t(n_stats)));
    return MLH_OK;
}

// ---- host <-> device hand-over of the pose without copy engines or blocking waits
// In: the pose travels in the kernel-argument segment of a one-wavefront launch that also resets the solver state (no staging
// buffer, no hipMemcpyAsync set-up latency, nothing for the host to wait on). Out: a one-wavefront launch writes the pose(s) into
// pinned host memory and then stores a sequence number with system-scope release; the host spins on that word (acquire) --
// microseconds instead of the tens of microseconds an interrupt-driven hipStreamSynchronize wake-up costs per frame.
struct PoseArg { double p[7]; };
__global__ void init_state_kernel(SolverState *S, PoseArg pose)
{
    double *w = reinterpret_cast<double *>(S);
    for (int i = threadIdx.x; i < int(sizeof(SolverState) / sizeof(double)); i += blockDim.x) w[i] = 0.0;
    __syncthreads();
    if (threadIdx.x < 7) { S->x[threadIdx.x] = pose.p[threadIdx.x]; S->cand[threadIdx.x] = pose.p[threadIdx.x]; }
    if (threadIdx.x < 6) S->V[threadIdx.x * 7] = 1.0;
}

// The pose the mapper starts frame k+1 from, computed where frame k's result lives (dev_math.hpp: chain_start_pose). One lane; the solve enqueued behind it reads S->x.
__global__ void chain_pose_kernel(SolverState *S, PoseArg wodom_prev, PoseArg wodom_cur, HostPublish *start_out)
{
    if (threadIdx.x != 0) return;
    double xc[7], out[7];
    for (int i = 0; i < 7; ++i) xc[i] = S->x[i];
    chain_start_pose(xc, wodom_prev.p, wodom_cur.p, out);
    for (int i = 0; i < 7; ++i) { S->x[i] = out[i]; S->cand[i] = out[i]; }
    if (start_out) for (int i = 0; i < 7; ++i) start_out->xb[1][i] = out[i];      // the frame's start pose, for a host that may have to solve the frame again
}

__global__ void set_block_pose_kernel(SolverState *S, int b, PoseArg pose)
{
    if (threadIdx.x < 7) S->xb[b][threadIdx.x] = pose.p[threadIdx.x];
}

__global__ void publish_kernel(const SolverState *S, HostPublish *h, unsigned long long seq)
{
    if (threadIdx.x < 7) h->x[threadIdx.x] = S->x[threadIdx.x];
    if (threadIdx.x < 56) h->xb[threadIdx.x / 7][threadIdx.x % 7] = S->xb[threadIdx.x / 7][threadIdx.x % 7];
    if (threadIdx.x == 63) h->done = S->done;
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&h->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int upload_pose(mlh_ctx *ctx, const double pose[7])
{
    static_assert(sizeof(SolverState) % sizeof(double) == 0, "SolverState is cleared in doubles");
    // a solve submitted with mlh_gn_solve_begin* may have left its last iteration as tile records and its pose in SolverState::xi[slot]: this launch zeroes the whole
    // state, so that solve has to be completed (and published to ITS host record) before, not by the match launch that follows
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }
    PoseArg a;
    for (int i = 0; i < 7; ++i) a.p[i] = pose[i];
    MLH_LAUNCH(init_state_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->state.as<SolverState>(), a);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

static int upload_block_pose(mlh_ctx *ctx, int b, const double pose[7])
{
    PoseArg a;
    for (int i = 0; i < 7; ++i) a.p[i] = pose[i];
    MLH_LAUNCH(set_block_pose_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->state.as<SolverState>(), b, a);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

__global__ void stream_flag_kernel(unsigned long long *h, unsigned long long seq)
{
    if (threadIdx.x == 0) __hip_atomic_store(h, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

hipError_t stream_flag_post(mlh_ctx *ctx, unsigned long long *seq_out)
{
    if (!ctx->h_sync) {
        void *p = nullptr;
        hipError_t e = hipHostMalloc(&p, 64, hipHostMallocDefault);
        if (e != hipSuccess) return e;
        ctx->h_sync = static_cast<unsigned long long *>(p);
        *ctx->h_sync = 0;
    }
    const unsigned long long seq = ++ctx->sync_seq;
    MLH_LAUNCH(stream_flag_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->h_sync, seq);
    *seq_out = seq;
    return hipGetLastError();
}

// the word only grows (one stream, launches in order): anything at or past `seq` means the work enqueued before that post is done
hipError_t stream_flag_wait(mlh_ctx *ctx, unsigned long long seq)
{
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(ctx->h_sync, __ATOMIC_ACQUIRE) < seq) {
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return hipStreamSynchronize(ctx->stream);
        host_wait_relax(spins);
    }
    return hipSuccess;
}

hipError_t stream_wait_spin(mlh_ctx *ctx)
{
    unsigned long long seq = 0;
    hipError_t e = stream_flag_post(ctx, &seq);
    if (e != hipSuccess) { (void)hipGetLastError(); return hipStreamSynchronize(ctx->stream); }
    return stream_flag_wait(ctx, seq);
}

// the pinned record and the next sequence number (allocated on first use)
// Three pinned records. Record 0 is everybody's; records 1 and 2 belong to scan2map's LM driver, which has one publication being read and the next chunk's
// already enqueued -- and may return with that one still to arrive: it must not land in a record some later call (a map staging on the other stream, say)
// is polling. which < 0: record 0; otherwise record 1 + (which & 1).
static int publish_slot(mlh_ctx *ctx, HostPublish **h, unsigned long long *seq, int which = -1)
{
    if (!ctx->h_state) {
        MLH_HIP(ctx, hipHostMalloc(&ctx->h_state, 3 * sizeof(HostPublish), hipHostMallocDefault));
        std::memset(ctx->h_state, 0, 3 * sizeof(HostPublish));
    }
    *h = static_cast<HostPublish *>(ctx->h_state) + (which < 0 ? 0 : 1 + (which & 1));
    *seq = ++ctx->publish_seq;
    return MLH_OK;
}

// spin until the device has stored `seq`; every kernel enqueued before the publishing one has completed when this returns
static int wait_published(mlh_ctx *ctx, unsigned long long seq, HostPublish &out, void *record = nullptr)
{
    HostPublish *h = static_cast<HostPublish *>(record ? record : ctx->h_state);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) != seq) {
        if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
            // not the fast case (a long LM run, a profiler, a fault): fall back to the blocking wait, which also surfaces errors
            MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (__atomic_load_n(&h->seq, __ATOMIC_ACQUIRE) != seq) return fail(ctx, MLH_ERR_HIP, "pose publication did not arrive");
            break;
        }
        host_wait_relax(spins);
    }
    out = *h;
    // several ranks joined by the mailbox communicator: the launches behind this publication exchanged records with the peers inside their finish. A peer that
    // never arrived leaves the error word set (p2p_dev.hpp): no solve entry point returns a pose built on partial sums as if it were the job's
    if (ctx->p2p.active) return device_error_check(ctx);
    return MLH_OK;
}

// enqueue a stand-alone publication of the pose(s) and wait for it
static int fetch_published(mlh_ctx *ctx, HostPublish &out)
{
    HostPublish *h;
    unsigned long long seq;
    int rc = publish_slot(ctx, &h, &seq);
    if (rc) return rc;
    MLH_LAUNCH(publish_kernel, dim3(1), dim3(64), 0, ctx->stream, (const SolverState *)ctx->state.as<SolverState>(), h, seq);
    MLH_HIP(ctx, hipGetLastError());
    return wait_published(ctx, seq, out);
}

static void copy_stat(const IterStatDev &d, mlh_iter_stat &o)
{
    o.n_surf = d.n_surf; o.n_corner = d.n_corner; o.is_degenerate = d.is_degenerate; o.lm_iterations = d.lm_iterations;
    o.successful_steps = d.successful_steps; o.termination = d.termination; o.cost = d.cost; o.final_cost = d.final_cost;
    std::memcpy(o.eigval, d.eigval, sizeof(o.eigval));
    std::memcpy(o.H, d.H, sizeof(o.H));
    std::memcpy(o.g, d.g, sizeof(o.g));
    std::memcpy(o.pose_after, d.pose_after, sizeof(o.pose_after));
}


// ---------------------------------------------------------------- what the device admits (mlh_ctx::caps)
// "hex[,hex...]" -> CU-mask words (32 compute units each), cut to the device's size; an empty vector = no mask
static std::vector<uint32_t> parse_cu_mask(const char *text, int cu_count)
{
    const int n_words = (cu_count + 31) / 32;
    std::vector<uint32_t> w(size_t(n_words), 0u);
    const char *p = text;
    bool any = false;
    for (int i = 0; i < n_words && p && *p; ++i) {
        char *rest = nullptr;
        w[size_t(i)] = uint32_t(std::strtoul(p, &rest, 16));
        any = any || rest != p;
        p = (rest && *rest == ',') ? rest + 1 : nullptr;
    }
    if (!any) return {};
    if (cu_count % 32) w.back() &= (1u << (cu_count % 32)) - 1u;      // (no bits beyond the device's last compute unit)
    return w;
}

// The first half of the device's compute units as mask words: the default of the staging stream (mlh_map_set_pair_overlapped)
static std::vector<uint32_t> lower_half_cu_mask(int cu_count)
{
    std::vector<uint32_t> w(size_t((cu_count + 31) / 32), 0u);
    for (int cu = 0; cu < cu_count / 2; ++cu) w[size_t(cu / 32)] |= 1u << (cu % 32);
    return w;
}

static int query_device_caps(mlh_ctx *c)
{
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return MLH_ERR_HIP;
    c->caps.cu_count = prop.multiProcessorCount;
    c->caps.cu_solver = prop.multiProcessorCount;
    int occ[2] = {0, 0}, occ_t = 0;
    if (lm_loop_occupancy(occ) != MLH_OK || track_loop_occupancy(&occ_t) != MLH_OK) return MLH_ERR_HIP;
    c->caps.blocks_per_cu[0] = occ[0]; c->caps.blocks_per_cu[1] = occ[1]; c->caps.blocks_per_cu[2] = occ_t;
    unsigned long long us = 20000;                // a completed barrier takes ~2 us; another context's longest kernel in the way, < 1 ms
    if (const char *e = std::getenv("MLH_LOOP_TIMEOUT_US")) { const long long v = std::atoll(e); if (v > 0) us = (unsigned long long)v; }
    c->caps.loop_timeout_ticks = us * 100ull;     // wall_clock64(): 100 MHz
    return MLH_OK;
}

// Workgroups of the loop kernels that may stand behind one in-kernel barrier. Residency: blocks per compute unit = min(occupancy query, 8, 6) -- the query is
// known to answer one too many where the SCALAR registers bind (MI355X_MICROARCH.md "Residency and cooperative launch": admitted = min(API, 8,
// floor(800 / (ceil(sgpr / 16) * 16 + 16)))); the runtime does not report a kernel's scalar register count, so the rule is applied for the most a wavefront can
// have (112 -> 6; these kernels: 106 SGPRs, 218 VGPRs -> the query's 2 is the vector registers' and exact) -- x the compute units the solver's stream may use,
// less a margin of an eighth of them (at least 8 workgroups). What ELSE runs on those compute units -- the staging stream's index builds, other contexts'
// kernels -- is not subtracted: such kernels end by themselves, so they can delay an arrival by their own duration (tens of microseconds, far inside the
// barrier's time limit) but cannot keep a workgroup of ours out for good; only our own grid exceeding the device could (tests/test_gpu_residency.py: four
// contexts' whole frames at once, 1 024-thread sort workgroups in every wave slot, no barrier given up on). Redundancy: every workgroup sums every tile's record, which
// stops paying beyond GN_DEFER_MAX_TILES (measured, profiles/r04_feature_sweep.txt; the fused thinning + solve call sizes its grid for the un-thinned clouds and
// accepts up to FUSED_LOOP_MAX_TILES). The smaller of the two; MLH_LOOP_MAX_TILES lowers it further (0: never).
constexpr int GN_DEFER_MAX_TILES = 160;
constexpr int FUSED_LOOP_MAX_TILES = 512;
static void set_loop_gates(mlh_ctx *c)
{
    const int by_size[3] = {GN_DEFER_MAX_TILES, FUSED_LOOP_MAX_TILES, GN_DEFER_MAX_TILES};
    int by_hand = -1;
    if (const char *e = std::getenv("MLH_LOOP_MAX_TILES")) by_hand = std::max(0, std::atoi(e));
    for (int i = 0; i < 3; ++i) {
        const int per_cu = std::max(0, std::min(c->caps.blocks_per_cu[i], 6));
        long long resident = (long long)per_cu * c->caps.cu_solver;
        resident = std::max(0ll, resident - std::max<long long>(c->caps.cu_solver / 8, 8));
        int gate = int(std::min<long long>(resident, by_size[i]));
        if (by_hand >= 0) gate = std::min(gate, by_hand);
        if (c->caps.loop_demoted[i] >= 0) gate = std::min(gate, c->caps.loop_demoted[i]);
        c->caps.loop_max_tiles[i] = gate;
    }
}

// A loop kernel of kind `which` came back with its barrier given up on at `tiles` workgroups: that many are evidently not resident together here. Half of it from
// now on (a second failure halves again; 0 = the launch-per-iteration forms for good).
static void demote_loop_gate(mlh_ctx *c, int which, int tiles)
{
    ++c->caps.loop_timeouts;
    const int cur = c->caps.loop_max_tiles[which];
    c->caps.loop_demoted[which] = std::min(cur, tiles) / 2;
    set_loop_gates(c);
}
static bool loop_tiles_ok(const mlh_ctx *c, int which, int tiles) { return tiles > 0 && tiles <= c->caps.loop_max_tiles[which]; }

}  // namespace mlh

using namespace mlh;

extern "C" {

const char *mlh_version(void) { return "mloam_hip 0.1 (gfx950)"; }

int mlh_create(mlh_ctx **out, int device_id)
{
    if (!out) return MLH_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MLH_ERR_HIP;
    if (device_id < 0 || device_id >= count) return MLH_ERR_INVALID;
    if (hipSetDevice(device_id) != hipSuccess) return MLH_ERR_HIP;
    mlh_ctx *c = new (std::nothrow) mlh_ctx;
    if (!c) return MLH_ERR_NOMEM;
    c->device = device_id;
    if (const char *e = std::getenv("MLH_KNN_LANES")) c->knn_lanes_override = std::atoi(e);
    if (const char *e = std::getenv("MLH_GN_DEFER")) c->gn_defer = std::atoi(e);
    if (const char *e = std::getenv("MLH_KNN_WARM")) c->knn_warm = std::atoi(e);
    if (const char *e = std::getenv("MLH_GN_FINAL_DEFER")) c->gn_final_defer = std::atoi(e);
    if (query_device_caps(c) != MLH_OK) { delete c; return MLH_ERR_HIP; }
    // the solver's stream: the whole device, or -- MLH_SOLVER_CU_MASK=<hex word>[,<hex word>...], bit i of word w = compute unit 32 w + i -- a part of it (a
    // deployment that keeps compute units for other work; the tests of the residency gates)
    std::vector<uint32_t> mask;
    if (const char *m = std::getenv("MLH_SOLVER_CU_MASK")) mask = parse_cu_mask(m, c->caps.cu_count);
    hipError_t se;
    if (!mask.empty()) {
        int bits = 0;
        for (uint32_t w : mask) bits += __builtin_popcount(w);
        if (bits <= 0) { delete c; return MLH_ERR_INVALID; }
        se = hipExtStreamCreateWithCUMask(&c->stream, uint32_t(mask.size()), mask.data());
        c->caps.cu_solver = bits; c->caps.solver_masked = true;
    } else se = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete c; return MLH_ERR_HIP; }
    set_loop_gates(c);
    *out = c;
    return MLH_OK;
}

int mlh_get_info(mlh_ctx *ctx, mlh_device_info *out)
{
    if (!ctx || !out) return MLH_ERR_INVALID;
    std::memset(out, 0, sizeof(*out));
    out->cu_count = ctx->caps.cu_count; out->cu_solver = ctx->caps.cu_solver;
    for (int i = 0; i < 3; ++i) { out->loop_blocks_per_cu[i] = ctx->caps.blocks_per_cu[i]; out->loop_max_tiles[i] = ctx->caps.loop_max_tiles[i]; }
    out->scan_uploads_from_ahead = int32_t(ctx->ahead.used & 0x7fffffffull);
    out->loop_launches = ctx->caps.loop_launches; out->loop_timeouts = ctx->caps.loop_timeouts; out->loop_fallbacks = ctx->caps.loop_fallbacks;
    return MLH_OK;
}

void mlh_destroy(mlh_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    prof_collect(ctx);
    for (auto e : ctx->prof.pool) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) {
        for (int set = 0; set < 2; ++set) {
            MapGrid &m = ctx->map_sets[set][k];
            m.raw.release(); m.sorted.release(); m.cell_id.release(); m.cell_start.release(); m.cell_fill.release(); m.block_sums.release(); m.bounds.release(); m.occ.release();
        }
        FeatSet &f = ctx->feat[k];
        f.pts.release(); f.covd.release(); f.corr.release(); f.nbr.release(); f.r.release(); f.J.release(); f.flag8.release(); f.fps_order.release();
    }
    ScanBuf &s = ctx->scan;
    s.pts.release(); s.start.release(); s.end.release(); s.curvature.release(); s.label.release(); s.picked.release(); s.stage.release();
    s.ring_counts.release(); s.ring_offsets.release(); s.totals.release();
    for (int i = 0; i < 4; ++i) s.lists[i].release();
    s.vox_stage.release(); s.vox_out.release(); s.ring_vox.release(); s.vox_keys.release(); s.vox_perm.release(); ctx->uct_buf.release(); ctx->fused[0].release(); ctx->fused[1].release(); ctx->fused_cnt.release(); ctx->fused_part.release();
    { TrackSet &t = ctx->track; for (int k = 0; k < 2; ++k) { MapGrid &m = t.grid[k]; m.raw.release(); m.sorted.release(); m.cell_id.release(); m.cell_start.release(); m.cell_fill.release(); m.block_sums.release(); m.bounds.release(); t.ring[k].release(); t.ring_start[k].release(); t.walk[k].release(); t.cur[k].release(); t.corr[k].release(); } }
    { OdomSet &o = ctx->odom; o.tab.release(); o.idx.release(); o.poses.release(); o.r.release(); o.J.release(); o.perm.release(); o.tile_group.release(); o.partial.release(); o.ne_out.release(); o.solve_aux.release(); }
    { SegBuf &g = ctx->seg; g.raw.release(); g.pix.release(); g.owner.release(); g.range.release(); g.ground.release(); g.keep.release(); }
    { VoxBuf &v = ctx->vox; v.in.release(); v.bounds.release(); v.cell.release(); v.word_of.release(); v.wpre.release(); v.cnt.release(); v.members.release(); v.vox_of.release(); v.sorted_idx.release(); v.leader.release(); v.out.release(); v.sums.release(); v.total.release(); }
    ctx->state.release(); ctx->partials.release(); ctx->ticket.release(); ctx->stats.release(); ctx->knn_q.release(); ctx->knn_idx.release(); ctx->knn_d.release(); ctx->tmp.release(); ctx->stdsort.release(); ctx->allreduce_buf.release(); ctx->oob_flag.release();
    comm_destroy(ctx);
    if (ctx->h_state) (void)hipHostFree(ctx->h_state);
    if (ctx->h_solve) (void)hipHostFree(ctx->h_solve);
    if (ctx->h_occ) (void)hipHostFree(ctx->h_occ);
    for (int k = 0; k < 2; ++k) if (ctx->select_host[k]) (void)hipHostFree(ctx->select_host[k]);
    if (ctx->vox_order_host) (void)hipHostFree(ctx->vox_order_host);
    if (ctx->fused_host) (void)hipHostFree(ctx->fused_host);
    if (ctx->h_scratch) (void)hipHostFree(ctx->h_scratch);
    if (ctx->h_sync) (void)hipHostFree(ctx->h_sync);
    if (ctx->ev_handover) (void)hipEventDestroy(ctx->ev_handover);
    if (ctx->h_rings) (void)hipHostFree(ctx->h_rings);
    if (ctx->h_pts) (void)hipHostFree(ctx->h_pts);
    for (int i = 0; i < 2; ++i) if (ctx->ev_pts[i]) (void)hipEventDestroy(ctx->ev_pts[i]);
    for (int i = 0; i < 2; ++i) if (ctx->ev_rings[i]) (void)hipEventDestroy(ctx->ev_rings[i]);
    if (ctx->h_dev_err) (void)hipHostFree(ctx->h_dev_err);
    for (int i = 0; i < 2; ++i) if (ctx->ev_set_built[i]) (void)hipEventDestroy(ctx->ev_set_built[i]);
    if (ctx->stream2) { (void)hipStreamSynchronize(ctx->stream2); (void)hipStreamDestroy(ctx->stream2); }
    if (ctx->ev_scan_reader) (void)hipEventDestroy(ctx->ev_scan_reader);
    if (ctx->ahead.cs) {
        (void)hipStreamSynchronize(ctx->ahead.cs); (void)hipStreamDestroy(ctx->ahead.cs);
        if (ctx->ahead.ev_arrived) (void)hipEventDestroy(ctx->ahead.ev_arrived);
        if (ctx->ahead.ev_consumed) (void)hipEventDestroy(ctx->ahead.ev_consumed);
        ctx->ahead.buf.release();
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mlh_last_error(const mlh_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
void *mlh_stream(mlh_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int mlh_synchronize(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    return device_error_check(ctx);
}

int mlh_profile_enable(mlh_ctx *ctx, int kernel_mask)
{
    if (!ctx) return MLH_ERR_INVALID;
    ctx->prof.mask = unsigned(kernel_mask) & ((1u << MLH_K_COUNT) - 1u);
    return MLH_OK;
}

int mlh_profile_sample(mlh_ctx *ctx, int every_n)
{
    if (!ctx || every_n < 1) return MLH_ERR_INVALID;
    ctx->prof.every = every_n;
    for (int i = 0; i < MLH_K_COUNT; ++i) ctx->prof.seen[i] = 0;
    return MLH_OK;
}

int mlh_profile_reset(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    for (int i = 0; i < MLH_K_COUNT; ++i) { ctx->prof.total_ms[i] = 0; ctx->prof.launches[i] = 0; }
    return MLH_OK;
}

int mlh_profile_get(mlh_ctx *ctx, int kernel_id, double *total_ms, long long *launches)
{
    if (!ctx || kernel_id < 0 || kernel_id >= MLH_K_COUNT) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (total_ms) *total_ms = ctx->prof.total_ms[kernel_id];
    if (launches) *launches = ctx->prof.launches[kernel_id];
    return MLH_OK;
}

// ---------------------------------------------------------------- extraction
// Whatever rewrites this context's scan buffers goes behind the launch of another context that still reads them (mlh_fuse_add_scan_from)
static int scan_wait_readers(mlh_ctx *ctx)
{
    if (ctx->scan_reader_pending.exchange(false, std::memory_order_acq_rel)) MLH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_scan_reader, 0));
    return MLH_OK;
}

// The points of the scan a LATER mlh_scan_upload will stage, sent to the device now, on a copy stream of the context's own: the copy engine moves them beside whatever
// kernels the context's stream is running (the previous scan's extraction, thinning, ...), and the upload leaves the frame's chain (framebench, the estimator / mapper
// pair: period 0.48-0.50 -> 0.42-0.43 ms with the caller's own prefetch; this is the same inside the library). One scan ahead; a second call replaces the first.
int mlh_scan_upload_ahead(mlh_ctx *ctx, const void *points, int stride_bytes, int n)
{
    if (!ctx) return MLH_ERR_INVALID;
    if (!points || n <= 0 || stride_bytes < 12 || (stride_bytes & 3)) return fail(ctx, MLH_ERR_INVALID, "bad point buffer (null, n <= 0, or stride not a multiple of 4 >= 12)");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    mlh_ctx::ScanAhead &A = ctx->ahead;
    if (!A.cs) {
        MLH_HIP(ctx, hipStreamCreateWithFlags(&A.cs, hipStreamNonBlocking));
        MLH_HIP(ctx, hipEventCreateWithFlags(&A.ev_arrived, hipEventDisableTiming));
        MLH_HIP(ctx, hipEventCreateWithFlags(&A.ev_consumed, hipEventDisableTiming));
    }
    A.valid = false;
    const size_t bytes = size_t(n) * size_t(stride_bytes);
    if (A.buf.cap < bytes) {
        // a pack kernel of the main stream may still be reading the old block
        if (A.consumed_recorded) MLH_HIP(ctx, hipEventSynchronize(A.ev_consumed));
        MLH_HIP(ctx, hipStreamSynchronize(A.cs));
        MLH_HIP(ctx, A.buf.ensure(bytes));
    }
    // the block is overwritten only behind the pack kernel that read the previous scan out of it
    if (A.consumed_recorded) MLH_HIP(ctx, hipStreamWaitEvent(A.cs, A.ev_consumed, 0));
    MLH_HIP(ctx, hipMemcpyAsync(A.buf.p, points, bytes, hipMemcpyHostToDevice, A.cs));
    MLH_HIP(ctx, hipEventRecord(A.ev_arrived, A.cs));
    // a page-locked source is read by the copy engine in place, later: the consuming mlh_scan_upload gives it back to the caller (it waits for the arrival there);
    // a pageable one has been taken into the runtime's staging memory when hipMemcpyAsync returns
    {
        hipPointerAttribute_t at;
        A.src_pinned = hipPointerGetAttributes(&at, points) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();
    }
    A.src = points; A.n = n; A.stride = stride_bytes; A.valid = true;
    ++A.issued;
    return MLH_OK;
}

int mlh_scan_upload(mlh_ctx *ctx, const void *points, int stride_bytes, int intensity_offset_bytes, int n, const int *scan_start,
                    const int *scan_end, int n_rings, int mem)
{
    if (!ctx) return MLH_ERR_INVALID;
    if (!scan_start || !scan_end || n_rings <= 0) return fail(ctx, MLH_ERR_INVALID, "bad ring table");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int wrc = scan_wait_readers(ctx); if (wrc) return wrc; }
    ScanBuf &sb = ctx->scan;
    sb.extracted = false;
    sb.voxelised = false;
    sb.h_lists_valid = sb.h_vox_valid = false;
    if (intensity_offset_bytes >= 0 && intensity_offset_bytes + 4 > stride_bytes) return fail(ctx, MLH_ERR_INVALID, "intensity offset outside the record");
    // A caller's PAGEABLE buffer (a ROS message) goes through hipMemcpyAsync as it is: like cudaMemcpyAsync, the call returns once the pageable source has been
    // copied into the runtime's own staging memory (the DMA to the device may still be in flight), so the buffer is the caller's again when this call returns,
    // and nothing here waits for the stream. MLH_SCAN_STAGE_PINNED=1 makes that independent of the runtime: the points are first copied into a pinned block the
    // context owns (two halves, an event per half) and go to the device from there -- measured +0.08 ms per two-LiDAR frame (the runtime writes small pageable
    // copies faster than an explicit memcpy + DMA pair: scripts/framebench.py, C++ leg, 0.859 against 0.940 ms, two alternations), hence not the default.
    // (A buffer the caller pinned is copied from in place; that case waits below.)
    const void *src_points = points;
    bool caller_pinned = false;
    int pts_half = -1;
    if (mem == MLH_MEM_HOST && points && n > 0 && stride_bytes >= 12) {
        hipPointerAttribute_t at;
        caller_pinned = hipPointerGetAttributes(&at, points) == hipSuccess && at.type == hipMemoryTypeHost;
        (void)hipGetLastError();
        static const bool stage_pinned = std::getenv("MLH_SCAN_STAGE_PINNED") && std::atoi(std::getenv("MLH_SCAN_STAGE_PINNED")) != 0;
        if (!caller_pinned && stage_pinned) {
            const size_t bytes = size_t(n) * stride_bytes;
            if (bytes > ctx->h_pts_cap) {
                MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
                if (ctx->h_pts) (void)hipHostFree(ctx->h_pts);
                ctx->h_pts = nullptr; ctx->h_pts_cap = 0;
                const size_t cap = ((bytes + bytes / 4 + 4095) / 4096) * 4096;
                MLH_HIP(ctx, hipHostMalloc(&ctx->h_pts, 2 * cap, hipHostMallocDefault));
                ctx->h_pts_cap = cap;
                for (int i = 0; i < 2; ++i) if (!ctx->ev_pts[i]) MLH_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_pts[i], hipEventDisableTiming));
                ctx->ev_pts_used[0] = ctx->ev_pts_used[1] = false;
            }
            pts_half = int(ctx->pts_turn++ & 1);
            if (ctx->ev_pts_used[pts_half]) MLH_HIP(ctx, hipEventSynchronize(ctx->ev_pts[pts_half]));      // two uploads ago
            void *dst = static_cast<char *>(ctx->h_pts) + size_t(pts_half) * ctx->h_pts_cap;
            std::memcpy(dst, points, bytes);
            src_points = dst;
        }
    }
    // sent ahead (mlh_scan_upload_ahead with this very buffer)? then the points are on the device already, or on their way: the main stream waits for their arrival and
    // packs from there. Any other host upload drops what was sent ahead (it was another scan's).
    mlh_ctx::ScanAhead &AH = ctx->ahead;
    int pts_mem = mem;
    bool from_ahead = false;
    if (mem == MLH_MEM_HOST && AH.valid) {
        if (AH.src == points && AH.n == n && AH.stride == stride_bytes && pts_half < 0) {
            MLH_HIP(ctx, hipStreamWaitEvent(ctx->stream, AH.ev_arrived, 0));
            if (AH.src_pinned) MLH_HIP(ctx, hipEventSynchronize(AH.ev_arrived));      // issued a frame ago: long done
            src_points = AH.buf.p; pts_mem = MLH_MEM_DEVICE; from_ahead = true; caller_pinned = false;
            ++AH.used;
        }
        AH.valid = false;
    }
    int rc = stage_points(ctx, src_points, stride_bytes, n, pts_mem, intensity_offset_bytes >= 0 ? intensity_offset_bytes : -1, -1, sb.pts, nullptr, ctx->tmp);
    if (rc) return rc;
    if (from_ahead) { MLH_HIP(ctx, hipEventRecord(AH.ev_consumed, ctx->stream)); AH.consumed_recorded = true; }
    if (pts_half >= 0) {
        MLH_HIP(ctx, hipEventRecord(ctx->ev_pts[pts_half], ctx->stream));
        ctx->ev_pts_used[pts_half] = true;
    }
    // The ring tables go to the device from a pinned block the context owns (two halves, used alternately; an event says when a half's copy has been read):
    // nothing has to be waited for before this call returns, so the host enqueues the extraction while the points are still being uploaded -- a blocking
    // wait here was ~40 us of idle GPU per frame (profiles/r03_frame_timeline.txt: the gap in front of the curvature kernel).
    const size_t ring_bytes = sizeof(int) * 2 * size_t(n_rings);
    if (ring_bytes > ctx->h_rings_cap) {
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->h_rings) (void)hipHostFree(ctx->h_rings);
        ctx->h_rings = nullptr; ctx->h_rings_cap = 0;
        MLH_HIP(ctx, hipHostMalloc(&ctx->h_rings, 2 * (ring_bytes + ring_bytes / 2), hipHostMallocDefault));
        ctx->h_rings_cap = ring_bytes + ring_bytes / 2;
        for (int i = 0; i < 2; ++i) if (!ctx->ev_rings[i]) MLH_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_rings[i], hipEventDisableTiming));
        ctx->ev_rings_used[0] = ctx->ev_rings_used[1] = false;
    }
    const int half = int(ctx->rings_turn++ & 1);
    if (ctx->ev_rings_used[half]) MLH_HIP(ctx, hipEventSynchronize(ctx->ev_rings[half]));        // two uploads ago: long done
    int *hs = reinterpret_cast<int *>(static_cast<char *>(ctx->h_rings) + size_t(half) * ctx->h_rings_cap), *he = hs + n_rings;
    if (mem == MLH_MEM_HOST) {
        std::memcpy(hs, scan_start, sizeof(int) * n_rings);
        std::memcpy(he, scan_end, sizeof(int) * n_rings);
    } else {
        MLH_HIP(ctx, hipMemcpyAsync(hs, scan_start, sizeof(int) * n_rings, hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, hipMemcpyAsync(he, scan_end, sizeof(int) * n_rings, hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, stream_wait_spin(ctx));
    }
    // rings are labelled by independent workgroups: that is exact only when the neighbour-suppression reach (+-5) of one
    // ring cannot touch another ring's labelled span, which the ImageSegmenter insets (+5 / -6) guarantee.
    int max_len = 0, prev_end = -1000000;
    for (int r = 0; r < n_rings; ++r) {
        if (he[r] - hs[r] < 6) continue;   // skipped by the extractor (cpp:155)
        if (hs[r] < 5 || he[r] + 5 > n) return fail(ctx, MLH_ERR_INVALID, "scan_start/scan_end must be inset by 5 from the cloud ends");
        if (hs[r] <= prev_end + 4) return fail(ctx, MLH_ERR_UNSUPPORTED, "rings must be ascending and separated by the +5/-6 insets of ScanInfo");
        prev_end = he[r];
        max_len = std::max(max_len, he[r] - hs[r]);
    }
    MLH_HIP(ctx, sb.start.ensure(sizeof(int) * 2 * size_t(n_rings)));                  // [start | end], as they sit in the pinned block: one copy
    MLH_HIP(ctx, hipMemcpyAsync(sb.start.p, hs, sizeof(int) * 2 * n_rings, hipMemcpyHostToDevice, ctx->stream));
    sb.end_alias = sb.start.as<int>() + n_rings;
    MLH_HIP(ctx, hipEventRecord(ctx->ev_rings[half], ctx->stream));
    ctx->ev_rings_used[half] = true;
    // a buffer the CALLER pinned is read by the copy engine in place, asynchronously, and is the caller's to reuse after this call: that (rare) case waits here
    if (caller_pinned) MLH_HIP(ctx, stream_wait_spin(ctx));
    sb.n = n; sb.n_rings = n_rings; sb.max_ring_len = max_len;
    return MLH_OK;
}

void mlh_segment_params_default(mlh_segment_params *p)
{
    if (!p) return;
    p->vertical_scans = 16; p->horizon_scans = 1800; p->min_cluster_size = 30; p->segment_valid_point_num = 5; p->segment_valid_line_num = 3;
    p->segment_theta = 1.047f; p->roi_range = 1.0; p->segment_flag = 1;
}

int mlh_segment_cloud(mlh_ctx *ctx, const void *points, int stride_bytes, int intensity_offset_bytes, int n, int mem, const mlh_segment_params *prm,
                      float *cloud_out, int32_t *n_out, int32_t *scan_start, int32_t *scan_end, float *outlier_out, int32_t outlier_capacity, int32_t *n_outlier)
{
    if (!ctx || !prm) return MLH_ERR_INVALID;
    if (outlier_out && outlier_capacity < 0) return fail(ctx, MLH_ERR_INVALID, "negative outlier capacity");
    if (intensity_offset_bytes >= 0 && intensity_offset_bytes + 4 > stride_bytes) return fail(ctx, MLH_ERR_INVALID, "intensity offset outside the record");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int wrc = scan_wait_readers(ctx); if (wrc) return wrc; }
    return segment_cloud_run(ctx, points, stride_bytes, intensity_offset_bytes, n, mem, *prm, cloud_out, n_out, scan_start, scan_end, outlier_out, outlier_capacity, n_outlier);
}

int mlh_extract_run(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int wrc = scan_wait_readers(ctx); if (wrc) return wrc; }
    return extract_run(ctx);
}

int mlh_extract_fetch(mlh_ctx *ctx, int32_t *label, float *curvature, int32_t *picked, int32_t *idx_out[4], int32_t n_out[4])
{
    if (!ctx) return MLH_ERR_INVALID;
    ScanBuf &sb = ctx->scan;
    if (!sb.extracted) return fail(ctx, MLH_ERR_STATE, "extract_run has not been called");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int totals[4] = {0, 0, 0, 0};
    MLH_HIP(ctx, hipMemcpyAsync(totals, sb.totals.p, sizeof(totals), hipMemcpyDeviceToHost, st));
    if (label) MLH_HIP(ctx, hipMemcpyAsync(label, sb.label.p, sizeof(int) * size_t(sb.n), hipMemcpyDeviceToHost, st));
    if (curvature) MLH_HIP(ctx, hipMemcpyAsync(curvature, sb.curvature.p, sizeof(float) * size_t(sb.n), hipMemcpyDeviceToHost, st));
    if (picked) MLH_HIP(ctx, hipMemcpyAsync(picked, sb.picked.p, sizeof(int) * size_t(sb.n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) {
        if (n_out) n_out[i] = totals[i];
        if (idx_out && idx_out[i] && totals[i] > 0)
            MLH_HIP(ctx, hipMemcpyAsync(idx_out[i], sb.lists[i].p, sizeof(int) * size_t(totals[i]), hipMemcpyDeviceToHost, st));
    }
    MLH_HIP(ctx, hipStreamSynchronize(st));
    prof_collect(ctx);
    return MLH_OK;
}

int mlh_extract_voxel_run(mlh_ctx *ctx, float leaf)
{
    if (!ctx || !(leaf > 0.f)) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int wrc = scan_wait_readers(ctx); if (wrc) return wrc; }
    return ring_voxel_run(ctx, leaf);
}

int mlh_extract_fetch_voxel(mlh_ctx *ctx, float *xyzi_out, int32_t *n_out)
{
    if (!ctx || !n_out) return MLH_ERR_INVALID;
    ScanBuf &sb = ctx->scan;
    if (!sb.voxelised) return fail(ctx, MLH_ERR_STATE, "extract_voxel_run has not been called");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int total = 0;
    MLH_HIP(ctx, hipMemcpyAsync(&total, sb.ring_vox.as<int>() + 2 * sb.n_rings, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = total;
    if (xyzi_out && total > 0) {
        MLH_HIP(ctx, hipMemcpyAsync(xyzi_out, sb.vox_out.p, sizeof(float4) * size_t(total), hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    prof_collect(ctx);
    return device_error_check(ctx);
}

int mlh_point_uncertainty(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                          const double *ext_poses, const double *ext_covs, int n_lidar, const double cov_measurement[9],
                          double trace_threshold, float *cov_vec_out, int32_t *keep_out)
{
    if (!ctx || !ext_poses || !ext_covs || !cov_measurement || !cov_vec_out || !keep_out) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return point_uncertainty_run(ctx, points, stride_bytes, n, intensity_offset_bytes, mem, ext_poses, ext_covs, n_lidar, cov_measurement,
                                 trace_threshold, cov_vec_out, keep_out);
}

int mlh_pure_odom_set(mlh_ctx *ctx, int n, const int32_t *type, const double *points, const double *coeffs, const double *sqrt_info,
                      const int32_t *frame_idx, const int32_t *ext_idx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return pure_odom_set(ctx, n, type, points, coeffs, sqrt_info, frame_idx, ext_idx);
}

int mlh_pure_odom_evaluate(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                           double *residuals, double *jacobians)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return pure_odom_evaluate(ctx, pivot, frames, n_frames, exts, n_ext, residuals, jacobians);
}

int mlh_pure_odom_normal_eq(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                            double huber_delta, double *JtJ, double *Jtr, double *cost, int32_t *n_residuals)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return pure_odom_normal_eq(ctx, pivot, frames, n_frames, exts, n_ext, huber_delta, JtJ, Jtr, cost, n_residuals);
}

int mlh_pure_odom_gn_solve(mlh_ctx *ctx, const double pivot[7], double *frames, int n_frames, double *exts, int n_ext, double huber_delta, int n_iters,
                           uint32_t const_block_mask, const double *V_update, double *cost, int32_t *n_residuals, int32_t *status)
{
    if (!ctx) return MLH_ERR_INVALID;
    // the window solve factorises THIS rank's J^T J / J^T r: with the factor table sharded over several ranks every rank would apply a different update
    if (distributed(ctx)) return fail(ctx, MLH_ERR_UNSUPPORTED, "mlh_pure_odom_gn_solve under a communicator: all-reduce mlh_pure_odom_normal_eq's H / g (mlh_allreduce_f64) and solve on the host, or build the whole factor table on every rank");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return pure_odom_gn_solve(ctx, pivot, frames, n_frames, exts, n_ext, huber_delta, n_iters, const_block_mask, V_update, cost, n_residuals, status);
}

int mlh_pure_odom_begin(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    return pure_odom_begin(ctx);
}

int mlh_pure_odom_add_matches(mlh_ctx *ctx, int kind, const double rel_pose[7], int k_neigh, uint32_t flags, float min_match_sq_dis,
                              float min_plane_dis, int frame_idx, int ext_idx)
{
    if (!ctx || kind < 0 || kind > 1 || !rel_pose) return MLH_ERR_INVALID;
    if (k_neigh != 5 && k_neigh != 10) return fail(ctx, MLH_ERR_UNSUPPORTED, "N_NEIGH is 5 or 10");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    if ((rc = upload_pose(ctx, rel_pose))) return rc;
    MatchArgs a;
    a.kind_mask = 1 << kind; a.flags = flags & MLH_FLAG_CHECK_FOV; a.min_match_sq_dis = min_match_sq_dis; a.min_plane_dis = min_plane_dis;
    a.huber_delta = 0.0; a.cov_measurement_trace = 0.0; a.dense = false; a.pose_sel = 0; a.k_neigh[0] = k_neigh;
    if ((rc = match_launch(ctx, a))) return rc;          // correspondences (validity + f32 coefficients) stay in HBM
    ctx->map_read_unsynced = true;                       // (nothing here waits for that launch: see mlh_map_set_pair_overlapped)
    return pure_odom_add_matches(ctx, kind, frame_idx, ext_idx);
}

int mlh_pure_odom_add_matches_gf(mlh_ctx *ctx, int kind, const double rel_pose[7], const double pivot[7], const double pose_i[7], const double ext[7], int k_neigh,
                                 uint32_t flags, float min_match_sq_dis, float min_plane_dis, int frame_idx, int ext_idx, float gf_ratio, uint64_t seed,
                                 int32_t *sel_out, int32_t *n_sel)
{
    if (!ctx || kind < 0 || kind > 1 || !rel_pose || !pivot || !pose_i || !ext || !(gf_ratio > 0.f)) return MLH_ERR_INVALID;
    if (k_neigh != 5 && k_neigh != 10) return fail(ctx, MLH_ERR_UNSUPPORTED, "N_NEIGH is 5 or 10");
    if (distributed(ctx)) return fail(ctx, MLH_ERR_UNSUPPORTED, "mlh_pure_odom_add_matches_gf: the selection loop is sequential over ALL features of the group; a sharded feature set cannot run it");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    if ((rc = upload_pose(ctx, rel_pose))) return rc;
    MatchArgs a;
    a.kind_mask = 1 << kind; a.flags = flags & MLH_FLAG_CHECK_FOV; a.min_match_sq_dis = min_match_sq_dis; a.min_plane_dis = min_plane_dis;
    a.huber_delta = 0.0; a.cov_measurement_trace = 0.0; a.dense = false; a.pose_sel = 0; a.k_neigh[0] = k_neigh;
    if ((rc = match_launch(ctx, a))) return rc;
    FeatSet &f = ctx->feat[kind];
    std::vector<int32_t> sel;
    if (gf_ratio == 1.0f) {
        // "if (gf_ratio == 1.0)": every matched feature, in feature order (estimator.cpp:1379-1412)
        if (sel_out || n_sel) {
            MLH_HIP(ctx, f.flag8.ensure((size_t(f.m) + 63) & ~size_t(63)));
            if ((rc = pure_odom_feature_rows(ctx, kind, pivot, pose_i, ext))) return rc;      // (its validity bytes)
            std::vector<uint8_t> v(size_t(f.m));
            MLH_HIP(ctx, hipMemcpyAsync(v.data(), f.flag8.p, size_t(f.m), hipMemcpyDeviceToHost, ctx->stream));
            MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            for (int i = 0; i < f.m; ++i) if (v[size_t(i)]) sel.push_back(i);
        } else ctx->map_read_unsynced = true;
    } else {
        if ((rc = pure_odom_feature_rows(ctx, kind, pivot, pose_i, ext))) return rc;
        std::mt19937 rng(static_cast<uint32_t>(seed));
        if ((rc = odom_good_feature_select(ctx, kind, gf_ratio, rng, sel))) return rc;
    }
    if (n_sel) *n_sel = int32_t(sel.size());
    if (sel_out) std::copy(sel.begin(), sel.end(), sel_out);
    return pure_odom_add_matches(ctx, kind, frame_idx, ext_idx);
}

int mlh_cloud_uct_associate_to_map(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int cov_offset_bytes,
                                   int trace_offset_bytes, const double pose_global[7], const double cov_global[36], const double *ext_poses,
                                   const double *ext_covs, int n_lidar, const double cov_measurement[9], int with_ua, double trace_threshold,
                                   void *out, int32_t *n_out, int mem)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return cloud_uct_associate_run(ctx, points, stride_bytes, n, intensity_offset_bytes, cov_offset_bytes, trace_offset_bytes, pose_global,
                                   cov_global, ext_poses, ext_covs, n_lidar, cov_measurement, with_ua, trace_threshold, out, n_out, mem);
}

int mlh_compound_pose_with_cov(const double pose_1[7], const double cov_1[36], const double pose_2[7], const double cov_2[36],
                               double pose_cp[7], double cov_cp[36])
{
    if (!pose_1 || !cov_1 || !pose_2 || !cov_2 || !pose_cp || !cov_cp) return MLH_ERR_INVALID;
    compound_pose_with_cov(pose_1, cov_1, pose_2, cov_2, pose_cp, cov_cp);
    return MLH_OK;
}

int mlh_voxel_filter(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int cov_offset_bytes,
                     int trace_offset_bytes, float leaf, float trace_threshold, void *out, int32_t *n_out, int mem)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return voxel_filter_run(ctx, points, stride_bytes, n, intensity_offset_bytes, cov_offset_bytes, trace_offset_bytes, leaf, trace_threshold,
                            out, n_out, mem);
}

// pcl::VoxelGrid<PointXYZI>::filter (PCL 1.8.0): one centroid per occupied voxel, every field averaged (CentroidPoint), output in
// ascending voxel index -- Estimator::buildLocalMap / buildCalibMap thin the window's clouds with it (estimator.cpp:1124-1130, 1194-1203)
int mlh_voxel_grid(mlh_ctx *ctx, const void *points, int stride_bytes, int n, int intensity_offset_bytes, float leaf, void *out, int32_t *n_out, int mem)
{
    if (!ctx || intensity_offset_bytes < 12) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return voxel_filter_run(ctx, points, stride_bytes, n, intensity_offset_bytes, -1, -1, leaf, 0.f, out, n_out, mem, nullptr, true, true);
}

int mlh_transform_point_cloud(mlh_ctx *ctx, void *points, int stride_bytes, int n, const double pose[7], int mem)
{
    if (!ctx || !points || !pose || n < 0 || stride_bytes < 12 || (stride_bytes & 3)) return MLH_ERR_INVALID;
    if (n == 0) return MLH_OK;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    unsigned char *dev = static_cast<unsigned char *>(points);
    const size_t bytes = size_t(n) * stride_bytes;
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, ctx->tmp.ensure(bytes));
        MLH_HIP(ctx, hipMemcpyAsync(ctx->tmp.p, points, bytes, hipMemcpyHostToDevice, ctx->stream));
        dev = ctx->tmp.as<unsigned char>();
    }
    { int rc = transform_cloud_launch(ctx, dev, stride_bytes, n, pose); if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; } }
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, hipMemcpyAsync(points, ctx->tmp.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MLH_OK;
}

// ---------------------------------------------------------------- map
// host records are copied to the staging buffer first (both clouds back to back); device records are read in place
static int map_set_impl(mlh_ctx *ctx, int n_maps, const int *kinds, const void *const *points, const int *n, int stride_bytes, float min_match_sq_dis, int mem)
{
    if (ctx) ++ctx->stage_epoch;      // (mlh_scan2map_end: a frame in flight may only be re-solved on the inputs it was submitted with)
    if (!(min_match_sq_dis > 0.f)) return fail(ctx, MLH_ERR_INVALID, "min_match_sq_dis must be positive");
    if (stride_bytes < 12 || (stride_bytes & 3)) return fail(ctx, MLH_ERR_INVALID, "bad point buffer (stride not a multiple of 4 >= 12)");
    for (int k = 0; k < n_maps; ++k) if (!points[k] || n[k] <= 0) return fail(ctx, MLH_ERR_INVALID, "bad point buffer (null or n <= 0)");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned char *src[2] = {nullptr, nullptr};
    if (mem == MLH_MEM_HOST) {
        size_t off[2] = {0, 0}, total = 0;
        for (int k = 0; k < n_maps; ++k) { off[k] = total; total += ((size_t(n[k]) * stride_bytes + 255) / 256) * 256; }
        // the overlapped path runs on the staging stream beside whatever the main stream has queued (a scan upload's pack kernel may still be reading ctx->tmp):
        // it stages through a buffer of its own
        DevBuf &tmp = (ctx->stream2 && ctx->stream == ctx->stream2) ? ctx->tmp_stage : ctx->tmp;
        MLH_HIP(ctx, tmp.ensure(total));
        for (int k = 0; k < n_maps; ++k) {
            MLH_HIP(ctx, hipMemcpyAsync(tmp.as<unsigned char>() + off[k], points[k], size_t(n[k]) * stride_bytes, hipMemcpyHostToDevice, ctx->stream));
            src[k] = tmp.as<unsigned char>() + off[k];
        }
    } else {
        for (int k = 0; k < n_maps; ++k) src[k] = static_cast<const unsigned char *>(points[k]);
    }
    HostPublish *pub;
    unsigned long long seq;
    int rc = publish_slot(ctx, &pub, &seq);
    if (rc) return rc;
    float sq[2] = {min_match_sq_dis, min_match_sq_dis};
    rc = map_stage_and_build(ctx, n_maps, kinds, src, n, stride_bytes, sq, pub, seq);
    if (rc) return rc;
    for (int k = 0; k < n_maps; ++k) ctx->feat[kinds[k]].matched = false;
    return MLH_OK;
}

int mlh_map_set(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, float min_match_sq_dis, int mem)
{
    if (!ctx || kind < 0 || kind > 1) return MLH_ERR_INVALID;
    const void *pts[1] = {points};
    return map_set_impl(ctx, 1, &kind, pts, &n, stride_bytes, min_match_sq_dis, mem);
}

int mlh_map_set_pair(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                     float min_match_sq_dis, int mem)
{
    if (!ctx) return MLH_ERR_INVALID;
    const int kinds[2] = {MLH_SURF, MLH_CORNER};
    const void *pts[2] = {surf_points, corner_points};
    const int n[2] = {n_surf, n_corner};
    return map_set_impl(ctx, 2, kinds, pts, n, stride_bytes, min_match_sq_dis, mem);
}

// The next maps staged WHILE the main stream is busy with something that does not read a map: a submitted solve (mlh_gn_solve_begin) still running on the
// current set, or -- no solve in flight, device-resident clouds -- the frame's own front end (upload, extractCloud, fusion, thinning: the local map comes from
// earlier keyframes and does not wait for the scan). Pack + fit check + index build go to the context's second stream and into the other map set; the call
// returns when they are complete (the host waits, the main stream does not) and the context has switched to the new set, so every launch enqueued after this
// call reads it. With host-resident clouds and no solve in flight it behaves like mlh_map_set_pair (the upload staging buffer is shared with the main stream).
int mlh_map_set_pair_overlapped(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                                float min_match_sq_dis, int mem)
{
    if (!ctx) return MLH_ERR_INVALID;
    if (!ctx->solve_pending && mem != MLH_MEM_DEVICE) return mlh_map_set_pair(ctx, surf_points, n_surf, corner_points, n_corner, stride_bytes, min_match_sq_dis, mem);
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->map_read_unsynced) {
        // mlh_pure_odom_add_matches returns with its map-reading launch still queued. Whatever set it reads, it must be done before a set is rewritten -- also
        // with a solve in flight (begin(A); add_matches(A); overlapped -> B; end; begin(B); overlapped -> A would otherwise rewrite A under that launch): the
        // main stream is drained, the solve included. Rare (the window's factor table is not built beside a pipelined mapper solve), so correctness first.
        MLH_HIP(ctx, stream_wait_spin(ctx));
        ctx->map_read_unsynced = false;
    }
    if (!ctx->stream2) {
        // The staging stream may use the first half of the compute units: its kernels are atomics- / latency-bound and lose ~5 % on half the device, while the
        // solver's launches they run beside lose less to them (step 0.1452 -> 0.1415 ms, three alternations in one gpurun call; every other CU, a quarter of the
        // CUs and 8 CUs per 32 measured no better). The mask is sized from the device's compute-unit count. MLH_STAGE_CU_MASK=<hex word>[,<hex word>...]
        // overrides (32 compute units per word; a single word is repeated over the first half of the words, a second one over the other half -- round 3's
        // "lo,hi" form; "ffffffff,ffffffff" = no mask).
        std::vector<uint32_t> words = lower_half_cu_mask(ctx->caps.cu_count);
        if (const char *m = std::getenv("MLH_STAGE_CU_MASK")) {
            const int nw = int(words.size());
            std::vector<uint32_t> given = parse_cu_mask(m, 32 * nw);
            int n_given = 1;
            for (const char *q = m; *q; ++q) n_given += *q == ',';
            if (!given.empty() && n_given <= 2 && nw > 2) {
                const uint32_t lo = given[0], hi = n_given == 2 ? given[1] : given[0];
                for (int i = 0; i < nw; ++i) given[size_t(i)] = i < nw / 2 ? lo : hi;
            }
            if (!given.empty()) {
                if (ctx->caps.cu_count % 32) given.back() &= (1u << (ctx->caps.cu_count % 32)) - 1u;
                words = given;
            }
        }
        int bits = 0;
        for (uint32_t w : words) bits += __builtin_popcount(w);
        if (bits <= 0 || bits >= ctx->caps.cu_count) MLH_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        else {
            MLH_HIP(ctx, hipExtStreamCreateWithCUMask(&ctx->stream2, uint32_t(words.size()), words.data()));
            ctx->caps.staging_masked = true;
        }
        for (int i = 0; i < 2; ++i) MLH_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_set_built[i], hipEventDisableTiming));
    }
    // The other set's last reader is the solve submitted BEFORE the one in flight; with at most one solve in flight here it has been collected, i.e. it is done:
    // no device-side wait is needed before its buffers are overwritten. (Round 3 first ordered the two streams with events on the solver's stream -- a record
    // after every solve, a wait before the next: two barrier packets, ~10 us of idle stream per frame in the kernel trace. Both are gone: the host orders.)
    if (ctx->solve_seq - ctx->solve_collected > 1) return fail(ctx, MLH_ERR_STATE, "collect the older solve (mlh_gn_solve_end) before staging the next frame's maps");
    const int target = 1 - ctx->map_set_cur;
    // ... unless the caller stages twice beside ONE solve (begin on A; overlapped -> B; overlapped -> A again with that solve uncollected): then the target's reader
    // is the solve in flight, and the set is rewritten only once the main stream has run dry (scripts/soak_schedule.py found this as a rare, timing-dependent
    // difference of 1e-5 m: the index of A rebuilt under a correspondence launch)
    if (ctx->set_reader_seq[target] > ctx->solve_collected) MLH_HIP(ctx, stream_wait_spin(ctx));
    hipStream_t main_stream = ctx->stream;
    ctx->stream = ctx->stream2;                    // everything map_set_impl enqueues (and its profiling brackets) goes to the staging stream ...
    ctx->map = ctx->map_sets[target];              // ... and into the other set
    const int kinds[2] = {MLH_SURF, MLH_CORNER};
    const void *pts[2] = {surf_points, corner_points};
    const int n[2] = {n_surf, n_corner};
    const int rc = map_set_impl(ctx, 2, kinds, pts, n, stride_bytes, min_match_sq_dis, mem);
    ctx->stream = main_stream;
    if (rc) { ctx->map = ctx->map_sets[ctx->map_set_cur]; return rc; }
    ctx->map_set_cur = target;
    // the index must be complete before anything that reads it is enqueued on the solver's stream: waited for HERE, on the host (the build is ~30 us of
    // small launches, the solve in flight another ~100), so that the solver's stream carries no cross-stream wait at all
    MLH_HIP(ctx, hipEventRecord(ctx->ev_set_built[target], ctx->stream2));
    {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (hipEventQuery(ctx->ev_set_built[target]) == hipErrorNotReady) {
            if ((++spins & 0xff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream2)); break; }
            host_wait_relax(spins);
        }
    }
    return MLH_OK;
}

int mlh_map_rebuild(mlh_ctx *ctx, int kind)
{
    if (!ctx || kind < MLH_ALL_KINDS || kind > 1) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return grid_build(ctx, kind == MLH_ALL_KINDS ? 3 : (1 << kind), false);
}

int mlh_set_extract_tie_order(mlh_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 1) return MLH_ERR_INVALID;
    ctx->extract_tie_ref = mode;
    return MLH_OK;
}

int mlh_set_gn_schedule(mlh_ctx *ctx, int deferred_finish, int knn_warm_start, int final_in_successor)
{
    if (!ctx) return MLH_ERR_INVALID;
    if ((deferred_finish != 0 && deferred_finish != 1) || (knn_warm_start != 0 && knn_warm_start != 1) || (final_in_successor != 0 && final_in_successor != 1))
        return fail(ctx, MLH_ERR_INVALID, "schedule switches are 0 or 1");
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }
    ctx->gn_defer = deferred_finish;
    ctx->knn_warm = knn_warm_start;
    ctx->gn_final_defer = final_in_successor;
    return MLH_OK;
}

int mlh_set_voxel_member_order(mlh_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MLH_ERR_INVALID;
    ctx->vox_member_order = mode;
    return MLH_OK;
}

__global__ void debug_noop_kernel(int *p) { if (p && threadIdx.x == 4096) *p = 0; }
// Debug: a launch with an impossible configuration (4096 threads per workgroup). Under MLH_CHECK_LAUNCH=1 the call returns MLH_ERR_HIP and mlh_last_error names
// debug_noop_kernel; without it the call returns MLH_OK and the runtime's sticky error is left for whoever asks next (what every launch of the library used to do).
int mlh_debug_bad_launch(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    MLH_LAUNCH(debug_noop_kernel, dim3(1), dim3(4096), 0, ctx->stream, (int *)nullptr);
    MLH_HIP(ctx, hipSuccess);                 // (the check every entry point runs behind its launches)
    if (!launch_check_enabled()) (void)hipGetLastError();      // do not leave the provoked error behind in a run that does not check launches
    return MLH_OK;
}

int mlh_std_sort_permutation(mlh_ctx *ctx, const int32_t *keys, int n0, int n, int32_t *perm_out, int mode)
{
    if (!ctx) return MLH_ERR_INVALID;
    if (!keys || !perm_out || n <= 0 || n0 < 0 || n0 > n || (mode != 1 && mode != 2)) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    if (mode == 2) {
        host_std_sort_permutation(keys, 0, n0, perm_out);
        host_std_sort_permutation(keys, n0, n, perm_out);
        return MLH_OK;
    }
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    MLH_HIP(ctx, ctx->tmp.ensure(sizeof(int) * size_t(n) * 2));
    int *d_keys = ctx->tmp.as<int>(), *d_perm = d_keys + n;
    MLH_HIP(ctx, hipMemcpyAsync(d_keys, keys, sizeof(int) * size_t(n), hipMemcpyHostToDevice, ctx->stream));
    int rc = device_std_sort_by_key(ctx, d_keys, n0, n, d_perm);
    if (rc) return rc;
    MLH_HIP(ctx, hipMemcpyAsync(perm_out, d_perm, sizeof(int) * size_t(n), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return device_error_check(ctx);
}

int mlh_map_info(mlh_ctx *ctx, int kind, int32_t *n_points, int32_t *occupied_cells, double *mean_cell_population, int32_t *knn_lanes)
{
    if (!ctx || kind < 0 || kind > 1) return MLH_ERR_INVALID;
    const MapGrid &g = ctx->map[kind];
    if (n_points) *n_points = g.built ? g.n : 0;
    if (occupied_cells) *occupied_cells = g.built ? g.occupied : 0;
    if (mean_cell_population) *mean_cell_population = (g.built && g.n > 0) ? double(g.pop_sq) / double(g.n) : 0.0;
    if (knn_lanes) { int lanes[2]; knn_lanes_for(ctx, 3, lanes); *knn_lanes = lanes[kind]; }
    return MLH_OK;
}

int mlh_knn(mlh_ctx *ctx, int kind, const float *queries_xyz, int nq, int k, int32_t *idx, float *sqdist)
{
    if (!ctx || kind < 0 || kind > 1 || !queries_xyz || !idx || !sqdist) return MLH_ERR_INVALID;
    if (k != 5) return fail(ctx, MLH_ERR_UNSUPPORTED, "only k = 5 is implemented");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    return knn_launch(ctx, kind, queries_xyz, nq, idx, sqdist);
}

// ---------------------------------------------------------------- features
// marks the padding slots between two pose blocks (intensity < 0 = "not a feature")
__global__ void pad_fill_kernel(float4 *pts, float4 *covd, int from, int to)
{
    const int i = from + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < to) { pts[i] = make_float4(0.f, 0.f, 0.f, -1.f); covd[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
}

// AoS records -> float4 at an offset; the stored w is the pose-block id (>= 0)
__global__ __launch_bounds__(256) void pack_block_kernel(const unsigned char *__restrict__ src, int stride, int n, int cov_off, float wval,
                                                         float4 *__restrict__ out, float4 *__restrict__ covd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *rec = reinterpret_cast<const float *>(src + size_t(i) * stride);
    out[i] = make_float4(rec[0], rec[1], rec[2], wval);
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    if (cov_off >= 0) {
        const float *cv = reinterpret_cast<const float *>(src + size_t(i) * stride + cov_off);
        c.x = cv[0]; c.y = cv[3]; c.z = cv[5];
    }
    covd[i] = c;
}

int mlh_features_set_block(mlh_ctx *ctx, int kind, int block, const void *points, int stride_bytes, int n, int cov_offset_bytes, int mem)
{
    if (ctx) ++ctx->stage_epoch;      // (mlh_scan2map_end: a frame in flight may only be re-solved on the inputs it was submitted with)
    if (!ctx || kind < 0 || kind > 1 || block < 0 || block >= 8) return MLH_ERR_INVALID;
    // (n == 0: a LiDAR without features of this kind in this frame -- the block exists, holds nothing, contributes no residuals; `points` may be null then)
    if ((!points && n > 0) || n < 0 || stride_bytes < 12 || (stride_bytes & 3)) return fail(ctx, MLH_ERR_INVALID, "bad point buffer");
    if (cov_offset_bytes >= 0 && cov_offset_bytes + 24 > stride_bytes) return fail(ctx, MLH_ERR_INVALID, "cov_offset_bytes + 24 exceeds the record stride");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    FeatSet &f = ctx->feat[kind];
    if (block == 0) { f.m = 0; f.n_blocks = 0; f.has_cov = false; }
    if (block != f.n_blocks) return fail(ctx, MLH_ERR_STATE, "pose blocks must be staged in ascending order starting at 0");
    f.matched = false;
    const int start = block == 0 ? 0 : ((f.m + 255) / 256) * 256;     // blocks begin on fit-tile boundaries
    const size_t total = size_t(start) + size_t(n);
    MLH_HIP(ctx, f.pts.grow(sizeof(float4) * total, sizeof(float4) * size_t(f.m), ctx->stream));
    MLH_HIP(ctx, f.covd.grow(sizeof(float4) * total, sizeof(float4) * size_t(f.m), ctx->stream));
    if (start > f.m)
        MLH_LAUNCH(pad_fill_kernel, dim3((start - f.m + 255) / 256), dim3(256), 0, ctx->stream, f.pts.as<float4>(), f.covd.as<float4>(), f.m, start);
    const unsigned char *src = static_cast<const unsigned char *>(points);
    if (mem == MLH_MEM_HOST && n > 0) {
        MLH_HIP(ctx, ctx->tmp.ensure(size_t(n) * stride_bytes));
        MLH_HIP(ctx, hipMemcpyAsync(ctx->tmp.p, points, size_t(n) * stride_bytes, hipMemcpyHostToDevice, ctx->stream));
        src = ctx->tmp.as<unsigned char>();
    }
    if (n > 0)
        MLH_LAUNCH(pack_block_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, src, stride_bytes, n, cov_offset_bytes, float(block),
                           f.pts.as<float4>() + start, f.covd.as<float4>() + start);
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f.blk_start[block] = start;
    f.blk_real[block] = n;
    f.m = int(total);
    f.n_blocks = block + 1;
    for (int b = f.n_blocks; b <= 8; ++b) f.blk_start[b] = f.m;
    f.has_cov = f.has_cov || cov_offset_bytes >= 0;
    return MLH_OK;
}

int mlh_features_set(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes,
                     int cov_offset_bytes, int mem)
{
    if (ctx) ++ctx->stage_epoch;      // (mlh_scan2map_end: a frame in flight may only be re-solved on the inputs it was submitted with)
    if (!ctx || kind < 0 || kind > 1) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    FeatSet &f = ctx->feat[kind];
    f.matched = false;
    f.m = 0;
    if (cov_offset_bytes >= 0 && cov_offset_bytes + 24 > stride_bytes) return fail(ctx, MLH_ERR_INVALID, "cov_offset_bytes + 24 exceeds the record stride");
    if (intensity_offset_bytes >= 0 && intensity_offset_bytes + 4 > stride_bytes) return fail(ctx, MLH_ERR_INVALID, "intensity offset exceeds the record stride");
    (void)intensity_offset_bytes;   // the LiDAR index is not needed by the single-pose path; the slot marks padding in block mode
    int rc = stage_points(ctx, points, stride_bytes, n, mem, -1, cov_offset_bytes, f.pts, &f.covd, ctx->tmp);
    if (rc) return rc;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    f.m = n;
    f.n_blocks = 1;
    f.blk_start[0] = 0;
    f.blk_real[0] = n;
    for (int b = 1; b <= 8; ++b) f.blk_start[b] = n;
    f.has_cov = cov_offset_bytes >= 0;
    return MLH_OK;
}

// The staged feature set of `kind` handed from one context to another on the same GPU, device to device: how an estimator-side context (extraction, fusion,
// thinning) feeds a mapper-side context (index, scan2MapOptimization) when the two run as separate pipelines, as the reference's estimator and mapper nodes do
// (there: a ROS message through host memory). Ordered after everything `src` has enqueued (waited for here); the copies run on dst's stream.
int mlh_features_copy(mlh_ctx *dst, mlh_ctx *src, int kind)
{
    if (dst) ++dst->stage_epoch;      // (mlh_scan2map_end: a frame in flight may only be re-solved on the inputs it was submitted with)
    if (!dst || !src || dst == src || kind < 0 || kind > 1) return MLH_ERR_INVALID;
    if (dst->device != src->device) return fail(dst, MLH_ERR_UNSUPPORTED, "mlh_features_copy: both contexts must be on the same device");
    MLH_HIP(dst, hipSetDevice(dst->device));
    const FeatSet &a = src->feat[kind];
    FeatSet &f = dst->feat[kind];
    f.matched = false;
    f.m = 0;
    if (a.m <= 0) return fail(dst, MLH_ERR_STATE, "mlh_features_copy: the source context has no staged features of this kind");
    // ordered behind the source's producers ON THE DEVICE (an event of dst's, recorded on src's stream): the caller is typically another thread than the one
    // driving src, and nothing of src's host-side state is touched here (its feature set must simply not be restaged before this call returns)
    if (!dst->ev_handover) MLH_HIP(dst, hipEventCreateWithFlags(&dst->ev_handover, hipEventDisableTiming));
    MLH_HIP(dst, hipEventRecord(dst->ev_handover, src->stream));
    MLH_HIP(dst, hipStreamWaitEvent(dst->stream, dst->ev_handover, 0));
    MLH_HIP(dst, f.pts.ensure(sizeof(float4) * size_t(a.m)));
    MLH_HIP(dst, f.covd.ensure(sizeof(float4) * size_t(a.m)));
    MLH_HIP(dst, hipMemcpyAsync(f.pts.p, a.pts.p, sizeof(float4) * size_t(a.m), hipMemcpyDeviceToDevice, dst->stream));
    MLH_HIP(dst, hipMemcpyAsync(f.covd.p, a.covd.p, sizeof(float4) * size_t(a.m), hipMemcpyDeviceToDevice, dst->stream));
    MLH_HIP(dst, stream_wait_spin(dst));                           // the source may overwrite its set as soon as this returns
    f.m = a.m; f.n_blocks = a.n_blocks; f.nbr_stride = a.nbr_stride; f.has_cov = a.has_cov;
    for (int b = 0; b < 9; ++b) f.blk_start[b] = a.blk_start[b];
    for (int b = 0; b < 8; ++b) f.blk_real[b] = a.blk_real[b];
    return MLH_OK;
}

// downsampleCurrentScan for one feature kind, device-resident: the result IS the kind's feature set
int mlh_downsample_current_scan(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                                float leaf, const double *ext_poses, const double *ext_covs, int n_lidar, const double cov_measurement[9],
                                int with_ua, double trace_threshold, float *features_out, int32_t *n_features)
{
    if (ctx) ++ctx->stage_epoch;      // (mlh_scan2map_end: a frame in flight may only be re-solved on the inputs it was submitted with)
    if (!ctx || kind < 0 || kind > 1 || !n_features) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    FeatSet &f = ctx->feat[kind];
    f.matched = false;
    f.m = 0;
    float *d_out = nullptr;
    if (features_out) {
        MLH_HIP(ctx, ctx->tmp.ensure(sizeof(float) * 11 * size_t(n > 0 ? n : 1)));
        d_out = ctx->tmp.as<float>();
    }
    int m = 0;
    // a fused cloud of this context brings its exact bounding box along (mlh_fused_cloud): no bounds pass
    const float *known_bounds = nullptr;
    for (int k = 0; k < 2; ++k)
        if (mem == MLH_MEM_DEVICE && !ctx->fused_dirty && n > 0 && points == ctx->fused[k].p && n == ctx->fused_n[k] && stride_bytes == 16) known_bounds = ctx->fused_minmax[k];
    int rc = downsample_current_scan_run(ctx, points, stride_bytes, n, intensity_offset_bytes, mem, leaf, ext_poses, ext_covs, n_lidar, cov_measurement,
                                         with_ua, trace_threshold, f.pts, f.covd, d_out, &m, known_bounds);
    if (rc) return rc;
    if (features_out && m > 0) {
        MLH_HIP(ctx, hipMemcpyAsync(features_out, d_out, sizeof(float) * 11 * size_t(m), hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    *n_features = m;
    f.m = m;
    f.n_blocks = 1;
    f.blk_start[0] = 0;
    f.blk_real[0] = m;
    for (int b = 1; b <= 8; ++b) f.blk_start[b] = m;
    f.has_cov = true;
    return MLH_OK;
}

// downsampleCurrentScan for BOTH feature kinds (lidar_mapper_keyframe.cpp:359-368 thins the surf and the corner cloud back to back).
// When both clouds are this context's fused clouds (device-resident, bounding boxes known) they go through ONE thinning pipeline --
// the chains are dispatch-bound, so one chain for both halves the time; otherwise the two single calls are made.
int mlh_downsample_current_scan_pair(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                                     int intensity_offset_bytes, int mem, float leaf_surf, float leaf_corner, const double *ext_poses, const double *ext_covs,
                                     int n_lidar, const double cov_measurement[9], int with_ua, double trace_threshold, int32_t *n_surf_features,
                                     int32_t *n_corner_features)
{
    if (ctx) ++ctx->stage_epoch;      // (mlh_scan2map_end: a frame in flight may only be re-solved on the inputs it was submitted with)
    if (!ctx || !n_surf_features || !n_corner_features) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    const bool fused_pair = mem == MLH_MEM_DEVICE && !ctx->fused_dirty && stride_bytes == 16 && intensity_offset_bytes == 12 && n_surf > 0 && n_corner > 0 &&
                            surf_points == ctx->fused[MLH_SURF].p && n_surf == ctx->fused_n[MLH_SURF] &&
                            corner_points == ctx->fused[MLH_CORNER].p && n_corner == ctx->fused_n[MLH_CORNER];
    if (fused_pair) {
        for (int k = 0; k < 2; ++k) { ctx->feat[k].matched = false; ctx->feat[k].m = 0; }
        int m[2] = {0, 0};
        int rc = downsample_current_scan_pair_run(ctx, surf_points, n_surf, ctx->fused_minmax[MLH_SURF], leaf_surf, corner_points, n_corner,
                                                  ctx->fused_minmax[MLH_CORNER], leaf_corner, stride_bytes, intensity_offset_bytes, ext_poses, ext_covs, n_lidar,
                                                  cov_measurement, with_ua, trace_threshold, &m[0], &m[1]);
        if (rc == MLH_OK) {
            for (int k = 0; k < 2; ++k) {
                FeatSet &f = ctx->feat[k];
                f.m = m[k]; f.n_blocks = 1; f.blk_start[0] = 0; f.blk_real[0] = m[k];
                for (int b = 1; b <= 8; ++b) f.blk_start[b] = m[k];
                f.has_cov = true;
            }
            *n_surf_features = m[0];
            *n_corner_features = m[1];
            return MLH_OK;
        }
        if (rc != MLH_ERR_UNSUPPORTED) return rc;            // a grid too large for the shared index space: one by one below
    }
    int rc = mlh_downsample_current_scan(ctx, MLH_SURF, surf_points, stride_bytes, n_surf, intensity_offset_bytes, mem, leaf_surf, ext_poses, ext_covs, n_lidar,
                                         cov_measurement, with_ua, trace_threshold, nullptr, n_surf_features);
    if (rc) return rc;
    return mlh_downsample_current_scan(ctx, MLH_CORNER, corner_points, stride_bytes, n_corner, intensity_offset_bytes, mem, leaf_corner, ext_poses, ext_covs, n_lidar,
                                       cov_measurement, with_ua, trace_threshold, nullptr, n_corner_features);
}

// ---------------------------------------------------------------- host-driven match / linearise
static int fetch_dense_and_reduced(mlh_ctx *ctx, int kind, bool want_corr, uint8_t *valid, double *coeffs, double *r, double *J,
                                   double *JtJ, double *Jtr, double *cost, int32_t *n_valid)
{
    FeatSet &f = ctx->feat[kind];
    hipStream_t st = ctx->stream;
    std::vector<Corr> hc;
    if (want_corr && (valid || coeffs)) {
        hc.resize(f.m);
        MLH_HIP(ctx, hipMemcpyAsync(hc.data(), f.corr.p, sizeof(Corr) * size_t(f.m), hipMemcpyDeviceToHost, st));
    }
    if (r) MLH_HIP(ctx, hipMemcpyAsync(r, f.r.p, sizeof(double) * size_t(f.m), hipMemcpyDeviceToHost, st));
    if (J) MLH_HIP(ctx, hipMemcpyAsync(J, f.J.p, sizeof(double) * 6 * size_t(f.m), hipMemcpyDeviceToHost, st));
    SolverState hs;
    MLH_HIP(ctx, hipMemcpyAsync(&hs, ctx->state.p, sizeof(hs), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    prof_collect(ctx);
    if (!hc.empty()) {
        for (int i = 0; i < f.m; ++i) {
            if (valid) valid[i] = hc[i].valid ? 1 : 0;
            if (coeffs) for (int k = 0; k < 6; ++k) coeffs[size_t(i) * 6 + k] = double(hc[i].c[k]);
        }
    }
    if (JtJ) {
        int q = 0;
        for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { JtJ[i * 6 + j] = hs.ne[q]; JtJ[j * 6 + i] = hs.ne[q]; ++q; }
    }
    if (Jtr) for (int i = 0; i < 6; ++i) Jtr[i] = hs.ne[NE_G + i];
    if (cost) *cost = hs.ne[NE_COST];
    if (n_valid) *n_valid = int(hs.ne[NE_CNT] + 0.5);
    return MLH_OK;
}

int mlh_match_coeffs(mlh_ctx *ctx, int kind, uint8_t *valid, double *coeffs, int32_t *n_valid)
{
    if (!ctx || kind < 0 || kind > 1) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    FeatSet &f = ctx->feat[kind];
    if (!f.matched || f.m <= 0 || !f.corr.p) return fail(ctx, MLH_ERR_STATE, "no correspondences of this kind on the device (match first)");
    std::vector<Corr> hc(size_t(f.m));
    MLH_HIP(ctx, hipMemcpyAsync(hc.data(), f.corr.p, sizeof(Corr) * size_t(f.m), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    int n = 0;
    for (int i = 0; i < f.m; ++i) {
        if (valid) valid[i] = hc[i].valid ? 1 : 0;
        if (coeffs) for (int k = 0; k < 6; ++k) coeffs[size_t(i) * 6 + k] = double(hc[i].c[k]);
        n += hc[i].valid ? 1 : 0;
    }
    if (n_valid) *n_valid = n;
    return MLH_OK;
}

int mlh_match_linearize(mlh_ctx *ctx, int kind, const double pose[7], int k_neigh, uint32_t flags,
                        float min_match_sq_dis, float min_plane_dis, double huber_delta, double cov_measurement_trace,
                        uint8_t *valid, double *coeffs, double *r, double *J,
                        double *JtJ, double *Jtr, double *cost, int32_t *n_valid)
{
    if (!ctx || kind < 0 || kind > 1 || !pose) return MLH_ERR_INVALID;
    if (k_neigh != 5 && k_neigh != 10) return fail(ctx, MLH_ERR_UNSUPPORTED, "N_NEIGH is 5 (the mapper, the reference LiDAR) or 10 (buildCalibMap's other LiDARs, estimator.cpp:1135)");
    if ((r == nullptr) != (J == nullptr)) return fail(ctx, MLH_ERR_INVALID, "r and J must be requested together");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    if ((rc = upload_pose(ctx, pose))) return rc;
    MatchArgs a;
    a.kind_mask = 1 << kind; a.flags = flags; a.min_match_sq_dis = min_match_sq_dis; a.min_plane_dis = min_plane_dis;
    a.huber_delta = huber_delta; a.cov_measurement_trace = cov_measurement_trace; a.dense = (r != nullptr); a.pose_sel = 0;
    a.k_neigh[0] = k_neigh;
    if ((rc = match_launch(ctx, a))) return rc;
    if ((rc = reduce_only_launch(ctx, 0))) return rc;
    return fetch_dense_and_reduced(ctx, kind, true, valid, coeffs, r, J, JtJ, Jtr, cost, n_valid);
}

int mlh_linearize(mlh_ctx *ctx, int kind, const double pose[7], uint32_t flags, double huber_delta, double cov_measurement_trace,
                  double *r, double *J, double *JtJ, double *Jtr, double *cost, int32_t *n_valid)
{
    if (!ctx || kind < 0 || kind > 1 || !pose) return MLH_ERR_INVALID;
    if ((r == nullptr) != (J == nullptr)) return fail(ctx, MLH_ERR_INVALID, "r and J must be requested together");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    if ((rc = upload_pose(ctx, pose))) return rc;
    MatchArgs a;
    a.kind_mask = 1 << kind; a.flags = flags; a.min_match_sq_dis = 0.f; a.min_plane_dis = 0.f;
    a.huber_delta = huber_delta; a.cov_measurement_trace = cov_measurement_trace; a.dense = (r != nullptr); a.pose_sel = 0;
    if ((rc = linearize_launch(ctx, a))) return rc;
    if ((rc = reduce_only_launch(ctx, 0))) return rc;
    return fetch_dense_and_reduced(ctx, kind, false, nullptr, nullptr, r, J, JtJ, Jtr, cost, n_valid);
}

// ---------------------------------------------------------------- device-resident solvers
void mlh_solver_opts_default(mlh_solver_opts *o)
{
    if (!o) return;
    o->min_match_sq_dis = 1.0f;
    o->min_plane_dis = 0.2f;
    o->huber_delta = 0.1;
    o->map_eig_thre = 100.0;
    o->cov_measurement_trace = 0.0075;
    o->flags = 0;
    o->max_outer = 2;
    o->max_lm_iterations = 30;
    o->gf_method = MLH_GF_WO;
    o->gf_ratio = 1.0;
    o->gf_seed = 0;
}

static MatchArgs args_from_opts(const mlh_solver_opts *o, int kind_mask, int pose_sel)
{
    MatchArgs a;
    a.kind_mask = kind_mask; a.eig_thre[0] = o->map_eig_thre; a.flags = o->flags & (MLH_FLAG_CHECK_FOV | MLH_FLAG_WITH_UA);
    a.min_match_sq_dis = o->min_match_sq_dis; a.min_plane_dis = o->min_plane_dis;
    a.huber_delta = o->huber_delta; a.cov_measurement_trace = o->cov_measurement_trace; a.dense = false; a.pose_sel = pose_sel;
    return a;
}

// The deferred finish makes EVERY workgroup of the next correspondence launch sum the fit tiles' records: cheap for a mapper frame (88 records, ~1 000
// workgroups: +3.5 us against the 5 us the fit kernel's serial tail costs), quadratic in the launch size beyond it -- measured (profiles/r04_feature_sweep.txt):
// even at 27-34 k features (108-133 tiles), a loss from ~100 k features on (447 tiles: +14 us), 4x the launch at 450 k. Launches with more tiles than this keep the
// classic finish, whose one serial tail is noise next to a 100 us kernel.
static bool gn_defer_applies(const mlh_ctx *ctx, int kind_mask)
{
    if (!ctx->gn_defer || distributed(ctx)) return false;
    int tiles = 0;
    for (int k = 0; k < 2; ++k) if (kind_mask & (1 << k)) tiles += (ctx->feat[k].m + 256 - 1) / 256;      // (the fit kernel's tile: 256 features, match.hip)
    return tiles <= GN_DEFER_MAX_TILES;
}

static int fetch_pose_and_stats(mlh_ctx *ctx, double pose[7], mlh_iter_stat *stats, int n_stats)
{
    if (!stats || n_stats <= 0) {
        HostPublish hp;
        int rc = fetch_published(ctx, hp);
        if (rc) return rc;
        if (!ctx->prof.pending.empty()) prof_collect(ctx);     // their events precede the publication in stream order: complete
        for (int i = 0; i < 7; ++i) pose[i] = hp.x[i];
        return MLH_OK;
    }
    SolverState hs;
    std::vector<IterStatDev> hd(n_stats);
    MLH_HIP(ctx, hipMemcpyAsync(&hs, ctx->state.p, sizeof(hs), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(hd.data(), ctx->stats.p, sizeof(IterStatDev) * size_t(n_stats), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (ctx->p2p.active) { const int erc = device_error_check(ctx); if (erc) return erc; }      // (see wait_published)
    for (int i = 0; i < 7; ++i) pose[i] = hs.x[i];
    for (int i = 0; i < n_stats; ++i) copy_stat(hd[i], stats[i]);
    return MLH_OK;
}

int mlh_gn_solve(mlh_ctx *ctx, double pose_inout[7], int n_iters, const mlh_solver_opts *opts, mlh_iter_stat *stats)
{
    if (!ctx || !pose_inout || !opts || n_iters <= 0) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }      // (a solve submitted with mlh_gn_solve_begin* may have left its last iteration as records)
    int rc = ensure_state(ctx, n_iters);
    if (rc) return rc;
    const int mask = ((ctx->feat[0].m > 0 && ctx->map[0].built) ? 1 : 0) | ((ctx->feat[1].m > 0 && ctx->map[1].built) ? 2 : 0);
    if (!mask) return fail(ctx, MLH_ERR_STATE, "no map/features staged");
    // the pose goes in with the first iteration's kernel arguments and (single GPU, no statistics wanted) comes back through
    // pinned host memory written by the last iteration's finish: 2 launches per iteration and nothing else
    unsigned long long seq = 0;
    bool fused_publish = false;
    for (int it = 0; it < n_iters; ++it) {
        MatchArgs a = args_from_opts(opts, mask, 0);
        if (it == 0) a.init_pose = pose_inout;
        // iterations >= 1 re-find the neighbours of the same features in the same map: the previous iteration's five bound the search (not with ownership
        // planes: a feature that changes hands between iterations would bring another frame's records)
        a.warm = it >= 1 && ctx->knn_warm && !ctx->shard_lo && !ctx->shard_hi;
        if (!distributed(ctx) || ctx->p2p.active) {
            // single GPU: two launches per iteration; the fit kernel's last workgroup reduces, solves and updates the pose. Several ranks joined by the
            // mailbox communicator: the same two launches -- that workgroup exchanges the summed record with the peers (one hop) before it solves
            a.finish = 1;
            a.stat_slot = stats ? it : -1;
            if (!stats && n_iters >= 2 && gn_defer_applies(ctx, mask)) {
                // one GPU, no per-iteration statistics: only the LAST iteration keeps that finish; the others leave their tiles' records to the next
                // iteration's correspondence launch, whose every workgroup sums and solves for itself (match.hip: knn_features_kernel<.., PRE>)
                a.gn_iter = it; a.gn_iters = n_iters;
                if (it == 1) a.init_pose = pose_inout;       // iteration 0's pose, the one iteration 1 updates
                if (it < n_iters - 1) a.finish = 0;
            }
            if (it == n_iters - 1 && !stats) {
                if ((rc = publish_slot(ctx, &a.publish, &seq))) return rc;
                a.publish_seq = seq;
                fused_publish = true;
            }
            if ((rc = match_launch(ctx, a))) return rc;
        } else {
            // multi-GPU: the fit kernel's last workgroup leaves this rank's sums in the solver state, then ONE all-reduce of
            // 32 doubles, then every rank runs the identical solve
            a.finish = 2;
            if ((rc = match_launch(ctx, a))) return rc;
            if ((rc = comm_allreduce_state(ctx, 0))) return rc;
            if ((rc = gn_update_prereduced_launch(ctx, opts->map_eig_thre, stats ? it : -1))) return rc;
        }
    }
    if (fused_publish) {
        HostPublish hp;
        if ((rc = wait_published(ctx, seq, hp))) return rc;
        if (!ctx->prof.pending.empty()) {
            // the fit kernel that published may still be retiring: its own stop event (if one was requested) needs the stream to drain
            if (ctx->prof.mask & ((1u << MLH_K_FIT) | (1u << MLH_K_SOLVE))) MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
            prof_collect(ctx);
        }
        for (int i = 0; i < 7; ++i) pose_inout[i] = hp.x[i];
        return MLH_OK;
    }
    return fetch_pose_and_stats(ctx, pose_inout, stats, n_iters);
}

// ---- the same solve, submitted and collected separately: mlh_gn_solve_begin enqueues the iterations and returns; mlh_gn_solve_end waits for the pose. A caller that
// stages the NEXT frame's maps between the two (mlh_map_set_pair: its launches queue up behind this solve on the context's stream) keeps the GPU busy across
// the frame boundary -- the ~16 us of host turn-around between "pose published" and "next frame's first launch" (profiles/r03_step_timeline.txt) disappear.
// Up to TWO solves may be in flight (frame k + 1 submitted before frame k's pose is collected: the GPU starts it the moment frame k is done instead of after the
// host has seen the pose and enqueued ten launches); each publishes into its own pinned record, apart from the one the staging hand-shake uses.
static int gn_solve_submit(mlh_ctx *ctx, const double *pose_in, const double *wodom_prev, const double *wodom_cur, int n_iters, const mlh_solver_opts *opts)
{
    // several ranks: fine with the mailbox communicator (the exchange happens inside the launches, the host is not involved); not over RCCL
    if (ctx->comm) return fail(ctx, MLH_ERR_UNSUPPORTED, "mlh_gn_solve_begin under an RCCL communicator: the sharded solve there is a host-driven sequence of launches and collectives (use the mailbox communicator)");
    if (ctx->solve_seq - ctx->solve_collected >= 2) return fail(ctx, MLH_ERR_STATE, "two solves are already in flight: collect the older one with mlh_gn_solve_end first");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    const int mask = ((ctx->feat[0].m > 0 && ctx->map[0].built) ? 1 : 0) | ((ctx->feat[1].m > 0 && ctx->map[1].built) ? 2 : 0);
    if (!mask) return fail(ctx, MLH_ERR_STATE, "no map/features staged");
    if (!ctx->h_solve) {
        MLH_HIP(ctx, hipHostMalloc(&ctx->h_solve, 2 * sizeof(HostPublish), hipHostMallocDefault));      // one record per solve in flight
        std::memset(ctx->h_solve, 0, 2 * sizeof(HostPublish));
    }
    const unsigned long long seq = ctx->solve_seq + 1;
    ctx->solve_slot[seq & 1].kind = 0;
    HostPublish *rec = static_cast<HostPublish *>(ctx->h_solve) + (seq & 1);
    const bool defer = n_iters >= 2 && gn_defer_applies(ctx, mask);
    // The previous solve may have left its LAST iteration as tile records (gn_pending). A chained, deferred solve completes it in its own first launch -- unless this
    // frame needs a larger record buffer (the records would not survive the reallocation); everything else completes it now, before the chain launch reads the pose.
    size_t tiles_now = 0;
    for (int k = 0; k < 2; ++k) if (mask & (1 << k)) tiles_now += size_t((ctx->feat[k].m + 255) / 256);
    const bool consume = !pose_in && defer && ctx->gn_pending.active && sizeof(double) * NE_STRIDE * tiles_now <= ctx->partials.cap;
    mlh_ctx::GnPending pend = ctx->gn_pending;
    if (ctx->gn_pending.active && !consume && (rc = gn_flush_pending(ctx))) return rc;
    if (!pose_in && !consume) {                    // chained: the start pose is made on the device from the pose the previous solve left there
        PoseArg pa, pb;
        for (int i = 0; i < 7; ++i) { pa.p[i] = wodom_prev[i]; pb.p[i] = wodom_cur[i]; }
        MLH_LAUNCH(chain_pose_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->state.as<SolverState>(), pa, pb, static_cast<HostPublish *>(nullptr));
        MLH_HIP(ctx, hipGetLastError());
    }
    // this solve's own last iteration: left to a successor (or to mlh_gn_solve_end) when the schedule says so
    const bool leave_final = defer && ctx->gn_final_defer;
    const int base = ctx->gn_slot_base;
    ctx->gn_slot_base ^= 2;
    for (int it = 0; it < n_iters; ++it) {
        MatchArgs a = args_from_opts(opts, mask, 0);
        if (it == 0) a.init_pose = pose_in;        // null when chained: the kernels read the state's pose (or compute it: `consume`)
        a.finish = 1;
        a.stat_slot = -1;
        a.warm = it >= 1 && ctx->knn_warm && !ctx->shard_lo && !ctx->shard_hi;
        if (defer) {     // the finish moves into the next iteration's correspondence launch (see mlh_gn_solve)
            a.gn_iter = it; a.gn_iters = n_iters; a.gn_slot_base = base;
            if (it == 1) a.init_pose = pose_in;
            if (it < n_iters - 1 || leave_final) a.finish = 0;
            if (consume) {
                a.pre_final = true;                // iteration 0: complete the predecessor, publish its pose, chain; iterations >= 1: pose 0 sits in its slot
                if (it == 0) {
                    a.pre_final_tiles = pend.tiles; a.pre_final_slot = pend.slot; a.pre_final_thre = pend.thre; a.pre_final_freeze = pend.freeze;
                    a.pre_final_publish = static_cast<HostPublish *>(pend.rec); a.pre_final_seq = pend.seq;
                    a.chain_prev = wodom_prev; a.chain_cur = wodom_cur;
                }
            }
        }
        if (it == n_iters - 1 && !leave_final) {
            a.publish = rec;
            a.publish_seq = seq;
        }
        if ((rc = match_launch(ctx, a))) return rc;
        if (it == 0 && consume) ctx->gn_pending.active = false;       // consumed by the launch just enqueued
    }
    if (leave_final) {
        ctx->gn_pending.active = true;
        ctx->gn_pending.tiles = ctx->n_partial_tiles;
        ctx->gn_pending.slot = base + ((n_iters - 1) & 1);
        ctx->gn_pending.thre = opts->map_eig_thre;
        ctx->gn_pending.freeze = 0;
        ctx->gn_pending.rec = rec;
        ctx->gn_pending.seq = seq;
    }
    ctx->solve_seq = seq;
    ctx->set_reader_seq[ctx->map_set_cur] = seq;
    ctx->solve_pending = true;
    return MLH_OK;
}

int mlh_gn_solve_begin(mlh_ctx *ctx, const double pose_in[7], int n_iters, const mlh_solver_opts *opts)
{
    if (!ctx || !pose_in || !opts || n_iters <= 0) return MLH_ERR_INVALID;
    return gn_solve_submit(ctx, pose_in, nullptr, nullptr, n_iters, opts);
}

int mlh_gn_solve_begin_chained(mlh_ctx *ctx, const double wodom_prev[7], const double wodom_cur[7], int n_iters, const mlh_solver_opts *opts)
{
    if (!ctx || !wodom_prev || !wodom_cur || !opts || n_iters <= 0) return MLH_ERR_INVALID;
    if (ctx->solve_seq == 0) return fail(ctx, MLH_ERR_STATE, "mlh_gn_solve_begin_chained continues from the pose a previous solve left on the device: submit the first frame with mlh_gn_solve_begin");
    return gn_solve_submit(ctx, nullptr, wodom_prev, wodom_cur, n_iters, opts);
}

int mlh_gn_solve_end(mlh_ctx *ctx, double pose_out[7])
{
    if (!ctx || !pose_out) return MLH_ERR_INVALID;
    if (ctx->solve_seq == ctx->solve_collected) return fail(ctx, MLH_ERR_STATE, "no solve in flight (mlh_gn_solve_begin)");
    const unsigned long long seq = ctx->solve_collected + 1;        // the oldest one
    if (ctx->solve_slot[seq & 1].kind != 0) return fail(ctx, MLH_ERR_STATE, "the oldest solve in flight was submitted with mlh_scan2map_begin: collect it with mlh_scan2map_end");
    if (ctx->gn_pending.active && ctx->gn_pending.seq == seq) {      // nobody chained a successor behind it: its last iteration is completed here
        const int frc = gn_flush_pending(ctx);
        if (frc) return frc;
    }
    HostPublish hp;
    int rc = wait_published(ctx, seq, hp, static_cast<HostPublish *>(ctx->h_solve) + (seq & 1));
    ctx->solve_collected = seq;
    ctx->solve_pending = ctx->solve_seq != ctx->solve_collected;
    if (rc) return rc;
    if (!ctx->prof.pending.empty() && !ctx->solve_pending) {
        if (ctx->prof.mask & ((1u << MLH_K_FIT) | (1u << MLH_K_SOLVE))) MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        prof_collect(ctx);
    }
    for (int i = 0; i < 7; ++i) pose_out[i] = hp.x[i];
    return MLH_OK;
}

int mlh_gn_solve_blocks(mlh_ctx *ctx, double *poses_inout, int n_iters, const mlh_solver_opts *opts, const mlh_block_opts *bo,
                        mlh_iter_stat *stats)
{
    if (!ctx || !poses_inout || !opts || !bo || n_iters <= 0 || bo->n_blocks <= 0 || bo->n_blocks > 8) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }      // (a solve submitted with mlh_gn_solve_begin* may have left its last iteration as records)
    const int nb = bo->n_blocks;
    int rc = ensure_state(ctx, n_iters * nb);
    if (rc) return rc;
    if ((rc = upload_pose(ctx, poses_inout))) return rc;
    for (int b = 1; b < nb; ++b) if ((rc = upload_block_pose(ctx, b, poses_inout + 7 * b))) return rc;
    const int mask = ((ctx->feat[0].m > 0 && ctx->map[0].built) ? 1 : 0) | ((ctx->feat[1].m > 0 && ctx->map[1].built) ? 2 : 0);
    if (!mask) return fail(ctx, MLH_ERR_STATE, "no map/features staged");
    // The finish in the consumer, per pose block (one GPU, no statistics): a correspondence workgroup serves features of ONE block, so its prologue sums that block's
    // records only and solves that block only -- the four serial finishes of the classic last workgroup (20 us of a 40 us fit launch on config 4's frame) become one
    // parallel prologue, and the redundancy is per block (at most 160 tiles each). Together with the bounded search (N_NEIGH 5 or 10); every block needs surf features
    // (its first surf tile's workgroup is the one that leaves the block's pose in HBM).
    bool defer_blocks = !distributed(ctx) && !stats && n_iters >= 2 && ctx->gn_defer && ctx->knn_warm && !ctx->shard_lo && !ctx->shard_hi;
    for (int b = 0; b < nb && defer_blocks; ++b) {
        int tiles = 0;
        for (int k = 0; k < 2; ++k) if (mask & (1 << k)) tiles += (ctx->feat[k].blk_start[b + 1] - ctx->feat[k].blk_start[b] + 255) / 256;
        if (!(mask & 1) || ctx->feat[0].n_blocks != nb || ctx->feat[0].blk_real[b] <= 0 || tiles > GN_DEFER_MAX_TILES) defer_blocks = false;
    }
    for (int it = 0; it < n_iters; ++it) {
        MatchArgs a = args_from_opts(opts, mask, 0);
        a.n_blocks = nb;
        for (int b = 0; b < nb; ++b) { a.k_neigh[b] = bo->k_neigh[b]; a.eig_thre[b] = bo->eig_thre[b]; a.freeze[b] = bo->freeze[b]; }
        if (!distributed(ctx) || ctx->p2p.active) {      // (mailbox communicator: the blocks' records are exchanged inside the finish, one after the other)
            a.finish = 1;
            a.stat_slot = stats ? it * nb : -1;
            if (defer_blocks) {
                a.gn_iter = it; a.gn_iters = n_iters; a.gn_blocks = true;
                a.warm = it >= 1;
                if (it < n_iters - 1) a.finish = 0;
            }
            if ((rc = match_launch(ctx, a))) return rc;
        } else {
            // multi-GPU: per-block local sums in the fit kernel's last workgroup, ONE all-reduce of nb x 32 doubles, identical updates
            a.finish = 2;
            if ((rc = match_launch(ctx, a))) return rc;
            if ((rc = comm_allreduce_blocks(ctx, nb))) return rc;
            if ((rc = gn_update_blocks_prereduced_launch(ctx, nb, bo->eig_thre, bo->freeze, stats ? it * nb : -1))) return rc;
        }
    }
    HostPublish hs;
    std::vector<IterStatDev> hd(stats ? size_t(n_iters) * nb : 0);
    if (stats) MLH_HIP(ctx, hipMemcpyAsync(hd.data(), ctx->stats.p, sizeof(IterStatDev) * hd.size(), hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = fetch_published(ctx, hs))) return rc;
    if (stats) MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    for (int i = 0; i < 7; ++i) poses_inout[i] = hs.x[i];
    for (int b = 1; b < nb; ++b) for (int i = 0; i < 7; ++i) poses_inout[7 * b + i] = hs.xb[b][i];
    for (size_t i = 0; stats && i < hd.size(); ++i) copy_stat(hd[i], stats[i]);
    return MLH_OK;
}

// The host-polled form: chunks of LM launches with the loop's verdict read between them (statistics, good-feature selections, RCCL ranks, and the re-solve of a frame
// whose loop outgrew its look-ahead).
// The Levenberg-Marquardt launches of scan2map in the consumer-side form (match.hip: lm_consume_kernel) -- the default where it applies (one GPU, every feature
// used, no statistics asked for); MLH_LM_CONSUMER=0 keeps the classic launches (read at every call: an A/B can flip it between two frames of one process).
// ... and the whole LM loop of an outer iteration as ONE launch whose workgroups synchronise among themselves (match.hip: lm_loop_kernel); MLH_LM_LOOP=0 keeps
// one launch per LM iteration (read at every call, as above)
static bool lm_loop_enabled()
{
    const char *e = std::getenv("MLH_LM_LOOP");
    return !(e && std::atoi(e) == 0);
}
static int feature_tiles(const mlh_ctx *ctx)
{
    int tiles = 0;
    for (int k = 0; k < 2; ++k) tiles += (ctx->feat[k].m + 256 - 1) / 256;      // (the fit / linearise kernels' tile: 256 features, match.hip)
    return tiles;
}
// ... where every tile's workgroup can be resident at once on what this context's stream may use of the device (mlh_ctx::caps, asked at mlh_create)
static bool lm_loop_applies(const mlh_ctx *ctx, int which, int tiles) { return lm_loop_enabled() && loop_tiles_ok(ctx, which, tiles); }
// The barrier of a one-launch loop was given up on (`done` bit 2 of its publication): the gate comes down, and the caller solves the frame again through the
// launch-per-iteration form -- same arithmetic, same pose bits, no residency requirement
static void note_loop_timeout(mlh_ctx *ctx, int which, int tiles, bool solved_again)
{
    demote_loop_gate(ctx, which, tiles);
    if (solved_again) ++ctx->caps.loop_fallbacks;
}

// MLH_S2M_WARM=0: every outer iteration searches unbounded (A/B)
static bool s2m_warm_applies(const mlh_ctx *ctx)
{
    const char *e = std::getenv("MLH_S2M_WARM");
    if (e && std::atoi(e) == 0) return false;
    return ctx->knn_warm && !ctx->shard_lo && !ctx->shard_hi && ctx->own_mod <= 1;
}

static bool lm_consumer_switch()
{
    const char *e = std::getenv("MLH_LM_CONSUMER");
    return !(e && std::atoi(e) == 0);
}
static bool lm_consumer_enabled(const mlh_ctx *ctx)
{
    if (!lm_consumer_switch()) return false;
    // every workgroup sums every tile's record: the same size limit as the Gauss-Newton path's deferred finish (GN_DEFER_MAX_TILES)
    int tiles = 0;
    for (int k = 0; k < 2; ++k) tiles += (ctx->feat[k].m + 256 - 1) / 256;
    return tiles <= GN_DEFER_MAX_TILES;
}

// scan2MapOptimization matches through ActiveFeatureSelection::goodFeatureMatching, which passes n_neigh = 5 and CHECK_FOV = false to match*PointFromMap whatever
// the caller's configuration says (lidar_mapper.h:256-283, every gf_method): MLH_FLAG_CHECK_FOV does not apply to the scan2map entry points. (The flag is for
// mlh_gn_solve*, mlh_match_linearize and the odometry's matches, where the reference does pass CHECK_FOV: estimator.cpp:1142, 1149. Found by the "hard" scene
// family of round 6: tall poles put corner features outside the +-60 degree cone, and a scan2map that honoured the flag dropped what the reference keeps.)
static mlh_solver_opts scan2map_opts(const mlh_solver_opts *o)
{
    mlh_solver_opts c = *o;
    c.flags &= ~uint32_t(MLH_FLAG_CHECK_FOV);
    return c;
}

static int scan2map_polled(mlh_ctx *ctx, double pose_inout[7], const mlh_solver_opts *opts_in, mlh_iter_stat *stats, bool allow_loop = true)
{
    if (!ctx || !pose_inout || !opts_in || opts_in->max_outer <= 0) return MLH_ERR_INVALID;
    const mlh_solver_opts opts_v = scan2map_opts(opts_in), *opts = &opts_v;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }      // (a solve submitted with mlh_gn_solve_begin* may have left its last iteration as records)
    int rc = ensure_state(ctx, opts->max_outer);
    if (rc) return rc;
    // scan2MapOptimization runs only when the map has > 50 surf and > 10 corner points (lidar_mapper_keyframe.cpp:429)
    if (!(ctx->map[MLH_SURF].built && ctx->map[MLH_CORNER].built && ctx->map[MLH_SURF].n > 50 && ctx->map[MLH_CORNER].n > 10)) {
        if (stats) std::memset(stats, 0, sizeof(mlh_iter_stat) * size_t(opts->max_outer));
        return MLH_OK;
    }
    if (ctx->feat[0].m <= 0 || ctx->feat[1].m <= 0) return fail(ctx, MLH_ERR_STATE, "features_set is required for both kinds");
    // one GPU, every feature used (wo_gf): the LM begin rides in the match launch and every LM step in its linearise launch -- an outer
    // iteration is 2 + (LM iterations) launches, the pose goes in with the first launch's kernel arguments
    const bool fused = (!distributed(ctx) || ctx->p2p.active) && opts->gf_method == MLH_GF_WO;      // (mailbox communicator: the exchange rides in the finish)
    // with a feature selection (one GPU): the dense passes and the selection come first, then the LM begin rides in the launch that evaluates the
    // selected rows and every LM step in its linearise launch, as above
    const bool fused_lm = fused || !distributed(ctx);
    // ... and with nobody asking for per-iteration statistics, on one GPU: the LM step rides in the CONSUMER of the records -- the match launch leaves its tiles'
    // records, every LM launch begins by summing its predecessor's and running the begin / step in all workgroups, then evaluates at the candidate (one launch more
    // per loop: the last evaluation's verdict is the next launch's)
    const bool lmc = fused && !stats && !distributed(ctx) && lm_consumer_enabled(ctx);
    const int loop_tiles = feature_tiles(ctx);
    if (lmc && allow_loop && lm_loop_applies(ctx, 0, loop_tiles)) {
        // the LM loop of every outer iteration is one launch that ends when the loop does: the whole frame is enqueued at once, the last launch publishes
        HostPublish *rec = nullptr;
        unsigned long long seq = 0;
        if ((rc = publish_slot(ctx, &rec, &seq, 0))) return rc;
        for (int outer = 0; outer < opts->max_outer; ++outer) {
            MatchArgs a = args_from_opts(opts, 3, 0);
            a.finish = 0; a.lm_max_it = opts->max_lm_iterations;
            if (outer == 0) a.init_pose = pose_inout;
            a.warm = outer >= 1 && s2m_warm_applies(ctx);
            a.no_fit = loop_fit_fusable(a);                 // (the fit rides in the loop launch below)
            if ((rc = match_launch(ctx, a))) return rc;
            MatchArgs b = args_from_opts(opts, 3, 1);
            b.finish = 0; b.lmc = 3; b.lmc_j = 1; b.lm_max_it = opts->max_lm_iterations; b.lm_min_blocks = 0; b.fit_in_loop = a.no_fit;
            b.lm_expect_done = outer == 0 ? -1 : 1;
            if (outer == 0) b.init_pose = pose_inout;
            if (outer == opts->max_outer - 1) { b.publish = rec; b.publish_seq = seq; }
            if ((rc = lm_consume_launch(ctx, b))) return rc;
        }
        HostPublish hp;
        if ((rc = wait_published(ctx, seq, hp, rec))) return rc;
        if (hp.done & 4) {
            // the loop's workgroups did not all arrive at a barrier (not all resident at once beside whatever else runs here): the frame again, from the start
            // pose the caller still holds, through the launch-per-iteration form
            note_loop_timeout(ctx, 0, loop_tiles, true);
            return scan2map_polled(ctx, pose_inout, opts, stats, false);
        }
        if (!ctx->prof.pending.empty()) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
        for (int i = 0; i < 7; ++i) pose_inout[i] = hp.x[i];
        return MLH_OK;
    }
    if (!fused && (rc = upload_pose(ctx, pose_inout))) return rc;
    // LM iterations enqueued between two looks at the device-side `done` flag: six first (the mapper's solves converge in 5-7), then two at a time -- launches
    // enqueued after convergence are no-ops, but each still costs a dispatch (profiles/r03_frame_timeline.txt: five of them behind a 7-iteration solve)
    const int first_chunk = 6, next_chunk = 2;
    HostPublish last_hp;
    bool have_hp = false;  // fused path: the last chunk's publication already carries the pose
    int lm_chunk_idx = 0;  // which of the two pinned publication records the next chunk writes
    HostPublish *gf_loop_rec = nullptr;      // a good-feature frame whose LM loops ran as one launch each: the record its last launch publishes into
    unsigned long long gf_loop_seq = 0;
    std::mt19937 rng((uint32_t)opts->gf_seed);
    for (int outer = 0; outer < opts->max_outer; ++outer) {
        if (fused) {
            MatchArgs a = args_from_opts(opts, 3, 0);
            a.finish = lmc ? 0 : 3; a.stat_slot = stats ? outer : -1; a.lm_max_it = opts->max_lm_iterations; a.lm_min_blocks = 0;
            if (outer == 0) { a.init_pose = pose_inout; a.lm_expect_done = -1; }
            // outer iterations behind the first search the same map for the same features from a pose a few centimetres away: bounded by the neighbours the
            // previous outer iteration left (knn_feature_warm: still the exact 5-NN)
            a.warm = outer >= 1 && s2m_warm_applies(ctx);
            if ((rc = match_launch(ctx, a))) return rc;
        } else if (opts->gf_method == MLH_GF_WO) {
            if ((rc = match_launch(ctx, args_from_opts(opts, 3, 0)))) return rc;
        } else {
            // goodFeatureMatching for corners, then surfs (cpp:503-533), each against a fresh 1e-6*I; then the evaluation of the
            // selected residual blocks at the current pose (problem.Evaluate, cpp:575-581)
            // Both kinds' dense passes and their copies to the host are enqueued first: the corner selection loop runs on the host while the surf pass
            // and its copies are still in flight, and each kind's flags go back without a wait (the linearise launch is behind them on the stream)
            std::vector<int32_t> sel;
            for (int kind : {MLH_CORNER, MLH_SURF})
                if ((rc = good_feature_stage(ctx, kind, opts->gf_method, opts->gf_ratio, rng, opts->min_match_sq_dis, opts->min_plane_dis, true))) return rc;
            if ((rc = good_feature_fps_flush(ctx))) return rc;      // ('fps': the two kinds' loops side by side in one launch; nothing pending otherwise)
            for (int kind : {MLH_CORNER, MLH_SURF}) {
                double Hsel[36];
                for (int i = 0; i < 36; ++i) Hsel[i] = (i % 7 == 0) ? 1e-6 : 0.0;
                if ((rc = good_feature_finish(ctx, kind, opts->gf_method, opts->gf_ratio, rng, sel, Hsel, nullptr))) return rc;
            }
            MatchArgs a = args_from_opts(opts, 3, 0);
            if (fused_lm) { a.finish = 3; a.stat_slot = stats ? outer : -1; a.lm_max_it = opts->max_lm_iterations; a.lm_min_blocks = 0; }
            // the selected rows' LM loop as one launch (as the wo_gf frame's, above): the linearisation of the selection only leaves its records
            const bool loop_gf = fused_lm && !stats && !distributed(ctx) && lm_consumer_enabled(ctx) && allow_loop && lm_loop_applies(ctx, 0, loop_tiles);
            if (loop_gf) a.finish = 0;
            if ((rc = linearize_launch(ctx, a))) return rc;
            if (loop_gf) {
                MatchArgs b = args_from_opts(opts, 3, 1);
                b.finish = 0; b.lmc = 3; b.lmc_j = 1; b.lm_max_it = opts->max_lm_iterations; b.lm_min_blocks = 0;
                b.lm_expect_done = outer == 0 ? -1 : 1;
                if (outer == opts->max_outer - 1) {         // the last loop launch publishes pose (and a barrier given up on, if any)
                    if ((rc = publish_slot(ctx, &gf_loop_rec, &gf_loop_seq, 0))) return rc;
                    b.publish = gf_loop_rec; b.publish_seq = gf_loop_seq;
                }
                if ((rc = lm_consume_launch(ctx, b))) return rc;
                continue;                                   // (the pose stays on the device: the next outer iteration's selection starts from it)
            }
        }
        if (!fused_lm && (rc = lm_begin_launch(ctx, opts->map_eig_thre, opts->max_lm_iterations, stats ? outer : -1))) return rc;
        if (fused_lm) {
            // Chunks of LM launches, each ending in a launch that publishes pose + `done` itself. The chunk after the one whose verdict the host is waiting for
            // is already enqueued (into the other pinned record): when the verdict is "not yet" the GPU has gone on without the host's round trip (~15 us of idle
            // stream per poll in profiles/r03_frame_cpp_timeline_*.txt), when it is "done" the chunk ahead is two launches that find `done` set and leave.
            struct Pending { unsigned long long seq; HostPublish *rec; } pend[2];
            int n_pend = 0, enq = 0, lm_j = 0;
            const int lm_cap = opts->max_lm_iterations + (lmc ? 1 : 0);     // launches after which the loop has terminated by itself
            auto enqueue_chunk = [&](int count) -> int {
                for (int j = 0; j < count; ++j) {
                    MatchArgs a = args_from_opts(opts, 3, 1);
                    a.finish = 4; a.lm_max_it = opts->max_lm_iterations;
                    if (lmc) {
                        a.finish = 0; a.lmc = lm_j == 0 ? 1 : 2; a.lmc_j = ++lm_j; a.lm_min_blocks = 0;
                        if (a.lmc == 1 && outer == 0) { a.init_pose = pose_inout; a.lm_expect_done = -1; }
                    }
                    if (j == count - 1) {                   // the chunk's last launch publishes (no publication launch)
                        unsigned long long seq = 0;
                        int prc = publish_slot(ctx, &a.publish, &seq, lm_chunk_idx);
                        if (prc) return prc;
                        a.publish_seq = seq;
                        pend[n_pend++] = Pending{seq, a.publish};
                    }
                    int lrc = lmc ? lm_consume_launch(ctx, a) : linearize_launch(ctx, a);
                    if (lrc) return lrc;
                }
                ++lm_chunk_idx;
                enq += count;
                return MLH_OK;
            };
            if ((rc = enqueue_chunk(std::min(first_chunk + (lmc ? 1 : 0), lm_cap)))) return rc;
            for (;;) {
                if (enq < lm_cap && n_pend < 2 && (rc = enqueue_chunk(std::min(next_chunk, lm_cap - enq)))) return rc;
                HostPublish hp;                             // pinned-memory poll of the device-side `done` flag (no copy engine, no blocking wait)
                if ((rc = wait_published(ctx, pend[0].seq, hp, pend[0].rec))) return rc;
                last_hp = hp; have_hp = true;
                pend[0] = pend[1]; --n_pend;
                if ((hp.done & 1) || n_pend == 0) break;
            }
        } else {
            for (int it = 0, j_end = 0; it < opts->max_lm_iterations; it = j_end) {
                j_end = std::min(it + (it == 0 ? first_chunk : next_chunk), opts->max_lm_iterations);
                for (int j = it; j < j_end; ++j) {
                    if ((rc = linearize_launch(ctx, args_from_opts(opts, 3, 1)))) return rc;
                    if ((rc = lm_step_launch(ctx, opts->max_lm_iterations, -1))) return rc;
                }
                HostPublish hp;
                if ((rc = fetch_published(ctx, hp))) return rc;
                if (hp.done) break;
            }
        }
        if (stats && (rc = lm_finish_launch(ctx, outer))) return rc;     // fills the record's LM summary
    }
    if (gf_loop_rec) {
        HostPublish hp;
        if ((rc = wait_published(ctx, gf_loop_seq, hp, gf_loop_rec))) return rc;
        if (hp.done & 4) {               // as above (the selection's draws start from opts->gf_seed again: the same frame)
            note_loop_timeout(ctx, 0, loop_tiles, true);
            return scan2map_polled(ctx, pose_inout, opts, stats, false);
        }
        if (!ctx->prof.pending.empty()) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
        for (int i = 0; i < 7; ++i) pose_inout[i] = hp.x[i];
        return MLH_OK;
    }
    if (fused_lm && !stats && have_hp) {
        if (!ctx->prof.pending.empty()) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
        for (int i = 0; i < 7; ++i) pose_inout[i] = last_hp.x[i];
        return MLH_OK;
    }
    return fetch_pose_and_stats(ctx, pose_inout, stats, opts->max_outer);
}

// ---- scan2MapOptimization, submitted and collected separately (the call the reference makes once per frame, pipelined like mlh_gn_solve_begin / _end).
// The host does not read the Levenberg-Marquardt loop's verdict between launches: per outer iteration the match launch (LM begin in its finish) and `lm_lookahead`
// LM launches are enqueued at once; launches behind the loop's termination find `done` on the device and leave (~3 us each). The last launch publishes pose and
// verdict. A frame whose LM loop needs MORE than the look-ahead is detected on the device (the next outer iteration finds the loop unterminated: lm_overflow; or the
// last loop is unterminated at the publication) -- its result is then not scan2MapOptimization's and is never returned as such: mlh_scan2map_end solves the frame
// again synchronously when that is sound (nothing restaged since the submission, no younger solve chained behind it), and says so otherwise.
static int scan2map_submit(mlh_ctx *ctx, const double *pose_in, const double *wodom_prev, const double *wodom_cur, const mlh_solver_opts *opts_in, int lm_lookahead)
{
    const mlh_solver_opts opts_v = scan2map_opts(opts_in), *opts = &opts_v;
    if (ctx->comm) return fail(ctx, MLH_ERR_UNSUPPORTED, "mlh_scan2map_begin under an RCCL communicator: the sharded LM iteration there is a host-driven sequence of launches and collectives (use the mailbox communicator)");
    if (opts->gf_method != MLH_GF_WO) return fail(ctx, MLH_ERR_UNSUPPORTED, "mlh_scan2map_begin with a good-feature selection: the selection loops run on the host between the launches (use mlh_scan2map)");
    if (opts->max_outer <= 0) return fail(ctx, MLH_ERR_INVALID, "max_outer must be positive");
    if (ctx->solve_seq - ctx->solve_collected >= 2) return fail(ctx, MLH_ERR_STATE, "two solves are already in flight: collect the older one first");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }      // (a solve submitted with mlh_gn_solve_begin* may have left its last iteration as records)
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    if (!ctx->h_solve) {
        MLH_HIP(ctx, hipHostMalloc(&ctx->h_solve, 2 * sizeof(HostPublish), hipHostMallocDefault));
        std::memset(ctx->h_solve, 0, 2 * sizeof(HostPublish));
    }
    const unsigned long long seq = ctx->solve_seq + 1;
    HostPublish *rec = static_cast<HostPublish *>(ctx->h_solve) + (seq & 1);
    mlh_ctx::SolveSlot &slot = ctx->solve_slot[seq & 1];
    // everything that can refuse the frame is checked BEFORE the chain launch below rewrites the device pose: a refused mlh_scan2map_begin_chained leaves the
    // state as it found it, so the caller's retry does not apply transformUpdate / transformAssociateToMap twice
    const bool have_maps = ctx->map[MLH_SURF].built && ctx->map[MLH_CORNER].built && ctx->map[MLH_SURF].n > 50 && ctx->map[MLH_CORNER].n > 10;
    if (have_maps && (ctx->feat[0].m <= 0 || ctx->feat[1].m <= 0)) return fail(ctx, MLH_ERR_STATE, "features_set is required for both kinds");
    slot.kind = 1; slot.chained = pose_in == nullptr; slot.opts = *opts; slot.epoch = ctx->stage_epoch; slot.tainted = false;
    if (pose_in) for (int i = 0; i < 7; ++i) slot.start[i] = pose_in[i];
    if (!pose_in) {
        PoseArg pa, pb;
        for (int i = 0; i < 7; ++i) { pa.p[i] = wodom_prev[i]; pb.p[i] = wodom_cur[i]; }
        MLH_LAUNCH(chain_pose_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->state.as<SolverState>(), pa, pb, rec);
        MLH_HIP(ctx, hipGetLastError());
    }
    // scan2MapOptimization runs only when the map has > 50 surf and > 10 corner points (lidar_mapper_keyframe.cpp:429): otherwise the start pose is the result
    if (!have_maps) {
        slot.kind = 2;
        if (!pose_in) {          // chained: the start pose exists on the device only
            MLH_LAUNCH(publish_kernel, dim3(1), dim3(64), 0, ctx->stream, (const SolverState *)ctx->state.as<SolverState>(), rec, seq);
            MLH_HIP(ctx, hipGetLastError());
        }
        ctx->solve_seq = seq; ctx->solve_pending = true;
        return MLH_OK;
    }
    const int budget = std::max(1, std::min(lm_lookahead > 0 ? lm_lookahead : ctx->lm_lookahead_auto, opts->max_lm_iterations));
    // the consumer-side form of the LM launches (scan2map_polled): budget + 1 launches run `budget` LM steps
    const bool lmc = !ctx->p2p.active && lm_consumer_enabled(ctx);
    // ... or one launch per LM loop, which ends on the device when the loop does: no budget, nothing to overflow (an explicit lm_lookahead keeps the launches it counts)
    const bool loop = lmc && lm_lookahead <= 0 && lm_loop_applies(ctx, 0, feature_tiles(ctx));
    slot.loop_tiles = loop ? feature_tiles(ctx) : 0;
    for (int outer = 0; loop && outer < opts->max_outer; ++outer) {
        MatchArgs a = args_from_opts(opts, 3, 0);
        a.finish = 0; a.lm_max_it = opts->max_lm_iterations;
        if (outer == 0) a.init_pose = pose_in;
        a.warm = outer >= 1 && s2m_warm_applies(ctx);
        a.no_fit = loop_fit_fusable(a);
        if ((rc = match_launch(ctx, a))) return rc;
        MatchArgs b = args_from_opts(opts, 3, 1);
        b.finish = 0; b.lmc = 3; b.lmc_j = 1; b.lm_max_it = opts->max_lm_iterations; b.lm_min_blocks = 0; b.fit_in_loop = a.no_fit;
        b.lm_expect_done = outer == 0 ? -1 : 1;
        if (outer == 0) b.init_pose = pose_in;
        if (outer == opts->max_outer - 1) { b.publish = rec; b.publish_seq = seq; }
        if ((rc = lm_consume_launch(ctx, b))) return rc;
    }
    for (int outer = 0; !loop && outer < opts->max_outer; ++outer) {
        MatchArgs a = args_from_opts(opts, 3, 0);
        a.finish = lmc ? 0 : 3; a.stat_slot = -1; a.lm_max_it = opts->max_lm_iterations; a.lm_min_blocks = 0;
        a.lm_expect_done = outer == 0 ? -1 : 1;
        if (outer == 0) a.init_pose = pose_in;
        a.warm = outer >= 1 && s2m_warm_applies(ctx);
        if ((rc = match_launch(ctx, a))) return rc;
        const int n_launch = budget + (lmc ? 1 : 0);
        for (int j = 0; j < n_launch; ++j) {
            MatchArgs b = args_from_opts(opts, 3, 1);
            b.finish = 4; b.lm_max_it = opts->max_lm_iterations;
            if (lmc) {
                b.finish = 0; b.lmc = j == 0 ? 1 : 2; b.lmc_j = j + 1; b.lm_min_blocks = 0;
                if (j == 0) { b.lm_expect_done = outer == 0 ? -1 : 1; if (outer == 0) b.init_pose = pose_in; }
            }
            if (outer == opts->max_outer - 1 && j == n_launch - 1) { b.publish = rec; b.publish_seq = seq; }
            if ((rc = lmc ? lm_consume_launch(ctx, b) : linearize_launch(ctx, b))) return rc;
        }
    }
    ctx->solve_seq = seq;
    ctx->set_reader_seq[ctx->map_set_cur] = seq;
    ctx->solve_pending = true;
    return MLH_OK;
}

int mlh_scan2map_begin(mlh_ctx *ctx, const double pose_in[7], const mlh_solver_opts *opts, int lm_lookahead)
{
    if (!ctx || !pose_in || !opts) return MLH_ERR_INVALID;
    return scan2map_submit(ctx, pose_in, nullptr, nullptr, opts, lm_lookahead);
}

int mlh_scan2map_begin_chained(mlh_ctx *ctx, const double wodom_prev[7], const double wodom_cur[7], const mlh_solver_opts *opts, int lm_lookahead)
{
    if (!ctx || !wodom_prev || !wodom_cur || !opts) return MLH_ERR_INVALID;
    if (ctx->solve_seq == 0) return fail(ctx, MLH_ERR_STATE, "mlh_scan2map_begin_chained continues from the pose a previous solve left on the device: submit the first frame with mlh_scan2map_begin");
    return scan2map_submit(ctx, nullptr, wodom_prev, wodom_cur, opts, lm_lookahead);
}

int mlh_scan2map_end(mlh_ctx *ctx, double pose_out[7], int32_t *status_out)
{
    if (!ctx || !pose_out) return MLH_ERR_INVALID;
    if (status_out) *status_out = 0;
    if (ctx->solve_seq == ctx->solve_collected) return fail(ctx, MLH_ERR_STATE, "no solve in flight (mlh_scan2map_begin)");
    const unsigned long long seq = ctx->solve_collected + 1;        // the oldest one
    mlh_ctx::SolveSlot slot = ctx->solve_slot[seq & 1];
    if (slot.kind == 0) return fail(ctx, MLH_ERR_STATE, "the oldest solve in flight was submitted with mlh_gn_solve_begin: collect it with mlh_gn_solve_end");
    HostPublish hp;
    int rc = MLH_OK;
    if (slot.kind == 2 && !slot.chained) {
        for (int i = 0; i < 7; ++i) hp.x[i] = slot.start[i];
        hp.done = 1;
    } else {
        rc = wait_published(ctx, seq, hp, static_cast<HostPublish *>(ctx->h_solve) + (seq & 1));
    }
    ctx->solve_collected = seq;
    ctx->solve_pending = ctx->solve_seq != ctx->solve_collected;
    // a younger solve chained behind THIS one began from whatever pose this one left on the device: if this one did not produce a result, neither did that one
    auto taint_successor = [&]() { if (ctx->solve_pending && ctx->solve_slot[(seq + 1) & 1].chained) ctx->solve_slot[(seq + 1) & 1].tainted = true; };
    if (rc) { taint_successor(); return rc; }
    if (!ctx->prof.pending.empty() && !ctx->solve_pending) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
    const bool barrier_given_up = slot.kind == 1 && (hp.done & 4);       // (a one-launch LM loop whose workgroups were not all resident: lm_loop_kernel)
    if (barrier_given_up) demote_loop_gate(ctx, 0, slot.loop_tiles);
    if (slot.kind == 1 && !barrier_given_up) {
        // the next frame's automatic look-ahead: what this frame's longest LM loop used, plus two (consecutive mapper frames need about the same); a frame that
        // overflowed doubles it
        const int used = int(hp.xb[2][0]);
        const bool ok = (hp.done & 1) && !(hp.done & 2);
        ctx->lm_lookahead_auto = ok ? std::max(3, used + 2) : std::min(2 * std::max(ctx->lm_lookahead_auto, 4), slot.opts.max_lm_iterations);
    }
    if (slot.kind == 2 || ((hp.done & 1) && !(hp.done & 6))) {
        for (int i = 0; i < 7; ++i) pose_out[i] = hp.x[i];
        if (slot.tainted) {
            // this frame was chained behind one whose LM loop outgrew its look-ahead: it began from that frame's UNFINISHED pose. Its own loops terminated, but the
            // result is not the mapper's; a solve chained behind THIS one inherits the mark.
            taint_successor();
            if (status_out) { *status_out = 3; return MLH_OK; }
            return fail(ctx, MLH_ERR_INCOMPLETE, "mlh_scan2map_end: the frame was chained behind one that did not finish inside its look-ahead (status 3) and status_out is NULL");
        }
        return MLH_OK;
    }
    // the look-ahead was too short for this frame -- or its one-launch loop gave its barrier up
    double start[7];
    for (int i = 0; i < 7; ++i) start[i] = slot.chained ? hp.xb[1][i] : slot.start[i];
    if (!ctx->solve_pending && ctx->stage_epoch == slot.epoch && !slot.tainted) {
        // nothing younger is chained behind it and the frame's maps and features are still the staged ones: solve it as mlh_scan2map would have (after a barrier
        // given up on: through the launch-per-iteration form)
        for (int i = 0; i < 7; ++i) pose_out[i] = start[i];
        if (status_out) *status_out = 2;
        if (barrier_given_up) ++ctx->caps.loop_fallbacks;
        return scan2map_polled(ctx, pose_out, &slot.opts, nullptr, !barrier_given_up);
    }
    for (int i = 0; i < 7; ++i) pose_out[i] = start[i];
    // a younger solve chained behind this frame started from its unfinished pose: marked, and reported at ITS end (status 3)
    taint_successor();
    if (status_out) { *status_out = slot.tainted ? 3 : 1; return MLH_OK; }
    // the pose handed back is NOT a result, and this caller has no way to see that: a distinct return code instead of success
    return fail(ctx, MLH_ERR_INCOMPLETE, "mlh_scan2map_end: the frame did not finish inside its look-ahead and cannot be solved again here (status 1); status_out is NULL");
}

// The synchronous call: the polled form. MLH_S2M_LOOKAHEAD=1 (A/B runs) makes it the split submission collected at once where nothing stands against that -- no
// statistics, every feature used (wo_gf), no RCCL communicator, no solve in flight: the whole frame (per outer iteration the match launch + the look-ahead budget of
// LM launches) enqueued without the host reading the loop's verdict in between, a frame that outgrows the budget solved again by the polled form (status 2); the
// same poses, bit for bit. Measured in one gpurun call, two alternations (profiles/r05_knockout_experiments.txt): 0.2436 / 0.2435 ms per frame against the polled
// form's 0.2421 / 0.2422 -- the polls were already hidden behind the chunk enqueued ahead, and the launches that find `done` cost what the polls did. Not the default.
// (Measured on round 4's launches. Since the LM loop of an outer iteration is ONE launch that ends on the device -- scan2map_polled's first branch -- the default
// synchronous call enqueues the whole frame at once anyway and there is no budget left to look ahead of; the switch remains for the launches-per-iteration forms.)
int mlh_scan2map(mlh_ctx *ctx, double pose_inout[7], const mlh_solver_opts *opts, mlh_iter_stat *stats)
{
    if (!ctx || !pose_inout || !opts || opts->max_outer <= 0) return MLH_ERR_INVALID;
    static const bool lookahead = std::getenv("MLH_S2M_LOOKAHEAD") && std::atoi(std::getenv("MLH_S2M_LOOKAHEAD")) != 0;
    const bool eligible = lookahead && !stats && opts->gf_method == MLH_GF_WO && !ctx->comm && ctx->solve_seq == ctx->solve_collected;
    if (!eligible) return scan2map_polled(ctx, pose_inout, opts, stats);
    int rc = scan2map_submit(ctx, pose_inout, nullptr, nullptr, opts, 0);
    if (rc) return rc;
    double out[7];
    int32_t status = 0;
    rc = mlh_scan2map_end(ctx, out, &status);
    if (rc) return rc;
    if (status == 1 || status == 3) return fail(ctx, MLH_ERR_STATE, "mlh_scan2map: the frame could not be completed (internal: look-ahead overflow without a re-solve)");
    for (int i = 0; i < 7; ++i) pose_inout[i] = out[i];
    return MLH_OK;
}

// downsampleCurrentScan + scan2MapOptimization with no host read between them (include/mloam_hip.h)
int mlh_downsample_scan2map(mlh_ctx *ctx, const void *surf_points, int n_surf, const void *corner_points, int n_corner, int stride_bytes,
                            int intensity_offset_bytes, int mem, float leaf_surf, float leaf_corner, const double *ext_poses, const double *ext_covs,
                            int n_lidar, const double cov_measurement[9], int with_ua, double trace_threshold, double pose_inout[7],
                            const mlh_solver_opts *opts_in, int32_t *n_surf_features, int32_t *n_corner_features)
{
    if (!ctx || !pose_inout || !opts_in || !n_surf_features || !n_corner_features || opts_in->max_outer <= 0) return MLH_ERR_INVALID;
    const mlh_solver_opts opts_v = scan2map_opts(opts_in), *opts = &opts_v;      // (CHECK_FOV does not apply to scan2map: see scan2map_opts)
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    auto two_calls = [&]() -> int {
        int rc = mlh_downsample_current_scan_pair(ctx, surf_points, n_surf, corner_points, n_corner, stride_bytes, intensity_offset_bytes, mem, leaf_surf, leaf_corner,
                                                  ext_poses, ext_covs, n_lidar, cov_measurement, with_ua, trace_threshold, n_surf_features, n_corner_features);
        if (rc) return rc;
        return mlh_scan2map(ctx, pose_inout, opts, nullptr);
    };
    const bool fused_pair = mem == MLH_MEM_DEVICE && !ctx->fused_dirty && stride_bytes == 16 && intensity_offset_bytes == 12 && n_surf > 0 && n_corner > 0 &&
                            surf_points == ctx->fused[MLH_SURF].p && n_surf == ctx->fused_n[MLH_SURF] &&
                            corner_points == ctx->fused[MLH_CORNER].p && n_corner == ctx->fused_n[MLH_CORNER];
    const bool have_maps = ctx->map[MLH_SURF].built && ctx->map[MLH_CORNER].built && ctx->map[MLH_SURF].n > 50 && ctx->map[MLH_CORNER].n > 10;
    // the loop kernel's barrier wants every tile's workgroup resident: the BOUND's tiles, since the real count is not known here
    const int bound_tiles = (n_surf + 255) / 256 + (n_corner + 255) / 256;
    static const bool off = std::getenv("MLH_FUSED_THIN_SOLVE") && std::atoi(std::getenv("MLH_FUSED_THIN_SOLVE")) == 0;      // (A/B: always the two calls)
    if (off || !fused_pair || !have_maps || distributed(ctx) || ctx->comm || opts->gf_method != MLH_GF_WO || !lm_consumer_switch() || !lm_loop_applies(ctx, 1, bound_tiles) ||
        ctx->solve_seq != ctx->solve_collected || ctx->vox_member_order != 1)
        return two_calls();
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    ++ctx->stage_epoch;
    for (int k = 0; k < 2; ++k) { ctx->feat[k].matched = false; ctx->feat[k].m = 0; }
    int m[2] = {0, 0};
    rc = downsample_current_scan_pair_run(ctx, surf_points, n_surf, ctx->fused_minmax[MLH_SURF], leaf_surf, corner_points, n_corner, ctx->fused_minmax[MLH_CORNER],
                                          leaf_corner, stride_bytes, intensity_offset_bytes, ext_poses, ext_covs, n_lidar, cov_measurement, with_ua, trace_threshold,
                                          &m[0], &m[1], true);
    if (rc == MLH_ERR_UNSUPPORTED) return two_calls();
    if (rc) return rc;
    auto stage_counts = [&](const int cnt[2]) {
        for (int k = 0; k < 2; ++k) {
            FeatSet &f = ctx->feat[k];
            f.m = cnt[k]; f.n_blocks = 1; f.blk_start[0] = 0; f.blk_real[0] = cnt[k];
            for (int b = 1; b <= 8; ++b) f.blk_start[b] = cnt[k];
            f.has_cov = true;
        }
    };
    if (m[0] >= 0) {                       // the thinning took a pipeline that waits for its counts anyway: the solve as usual
        stage_counts(m);
        *n_surf_features = m[0]; *n_corner_features = m[1];
        return mlh_scan2map(ctx, pose_inout, opts, nullptr);
    }
    const int bound[2] = {n_surf, n_corner};
    stage_counts(bound);                   // upper bounds: the launches' grids and the buffers; the kernels read the real counts
    HostPublish *rec = nullptr;
    unsigned long long seq = 0;
    if ((rc = publish_slot(ctx, &rec, &seq, 0))) return rc;
    for (int outer = 0; outer < opts->max_outer && !rc; ++outer) {
        MatchArgs a = args_from_opts(opts, 3, 0);
        a.finish = 0; a.lm_max_it = opts->max_lm_iterations; a.m_dev = ctx->thin_counts_dev;
        if (outer == 0) a.init_pose = pose_inout;
        a.warm = outer >= 1 && s2m_warm_applies(ctx);
        a.no_fit = loop_fit_fusable(a);
        if ((rc = match_launch(ctx, a))) break;
        MatchArgs b = args_from_opts(opts, 3, 1);
        b.finish = 0; b.lmc = 3; b.lmc_j = 1; b.lm_max_it = opts->max_lm_iterations; b.lm_min_blocks = 0; b.m_dev = ctx->thin_counts_dev; b.fit_in_loop = a.no_fit;
        b.lm_expect_done = outer == 0 ? -1 : 1;
        if (outer == 0) b.init_pose = pose_inout;
        if (outer == opts->max_outer - 1) { b.publish = rec; b.publish_seq = seq; }
        rc = lm_consume_launch(ctx, b);
    }
    HostPublish hp;
    if (!rc) rc = wait_published(ctx, seq, hp, rec);
    // the counts were published by the thinning's last launch, long before the pose: no wait here in practice
    int real[2] = {0, 0};
    if (!rc) {
        if (__atomic_load_n(ctx->thin_seq_host, __ATOMIC_ACQUIRE) != ctx->thin_seq) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); }
        if (__atomic_load_n(ctx->thin_seq_host, __ATOMIC_ACQUIRE) != ctx->thin_seq) rc = fail(ctx, MLH_ERR_HIP, "the thinned feature counts did not arrive");
        else { real[0] = ctx->thin_counts_host[0]; real[1] = ctx->thin_counts_host[1]; }
    }
    if (rc) { (void)hipStreamSynchronize(ctx->stream); for (int k = 0; k < 2; ++k) { ctx->feat[k].m = 0; ctx->feat[k].matched = false; } return rc; }
    stage_counts(real);
    *n_surf_features = real[0]; *n_corner_features = real[1];
    if ((rc = device_error_check(ctx))) return rc;
    if (real[0] <= 0 || real[1] <= 0) return fail(ctx, MLH_ERR_STATE, "features_set is required for both kinds");      // (what mlh_scan2map says of an empty kind)
    if (hp.done & 4) {
        // the loop's workgroups (sized for the un-thinned clouds) did not all arrive at a barrier: the thinned sets are staged and counted by now -- the solve again,
        // as the second of the two calls, through the launch-per-iteration form
        note_loop_timeout(ctx, 1, bound_tiles, true);
        return scan2map_polled(ctx, pose_inout, opts, nullptr, false);
    }
    if (!ctx->prof.pending.empty()) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
    for (int i = 0; i < 7; ++i) pose_inout[i] = hp.x[i];
    return MLH_OK;
}

// ---------------------------------------------------------------- scan-to-scan odometry (LidarTracker)
void mlh_track_opts_default(mlh_track_opts *o)
{
    if (!o) return;
    o->distance_sq_threshold = 25.0f; o->nearby_scan = 2.5f; o->huber_delta = 0.1; o->max_outer = 2; o->max_lm_iterations = 4;
}

static TrackArgs track_args(const mlh_track_opts *o, int pose_sel)
{
    TrackArgs a;
    a.pose_sel = pose_sel; a.dist_sq_thr = o->distance_sq_threshold; a.nearby_scan = o->nearby_scan; a.huber_delta = o->huber_delta;
    return a;
}

// previous-frame cloud of one kind: packed copies + ring table, everything enqueued and nothing waited for; *host_bad is valid after
// the next synchronisation of the stream
static int track_stage_prev(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                            float distance_sq_threshold, int *host_bad)
{
    TrackSet &T = ctx->track;
    MapGrid &g = T.grid[kind];
    g.built = false;
    int rc = stage_points(ctx, points, stride_bytes, n, mem, -2, -1, g.raw, nullptr, ctx->tmp);
    if (rc) return rc;
    const unsigned char *d_src = (mem == MLH_MEM_HOST) ? ctx->tmp.as<unsigned char>() : static_cast<const unsigned char *>(points);
    // second copy in the original order with the ring id in w: what the scan-line walks stream over
    MLH_HIP(ctx, T.walk[kind].ensure(sizeof(float4) * size_t(n)));
    MLH_LAUNCH(pack_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_src, stride_bytes, n, intensity_offset_bytes, -1,
                       T.walk[kind].as<float4>(), (float4 *)nullptr);
    if ((rc = track_set_prev_rings(ctx, kind, d_src, stride_bytes, n, intensity_offset_bytes, host_bad))) return rc;
    g.n = n;
    g.min_match_sq_dis = distance_sq_threshold / float(TRACK_SHELLS * TRACK_SHELLS);   // cell edge = 1.001 * sqrt(thr) / TRACK_SHELLS
    return MLH_OK;
}

// index build of the staged previous-frame clouds (one host round trip for all of them: the bounding boxes) + the deferred ring check
static int track_build_prev(mlh_ctx *ctx, int kind_mask, const int *host_bad)
{
    MapGrid *gp[2];
    int ng = 0;
    for (int k = 0; k < 2; ++k) if (kind_mask & (1 << k)) gp[ng++] = &ctx->track.grid[k];
    int rc = grid_build_grids(ctx, gp, ng, true);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); for (int k = 0; k < ng; ++k) gp[k]->built = false; return rc; }
    for (int k = 0; k < 2; ++k)
        if ((kind_mask & (1 << k)) && host_bad[k]) {
            ctx->track.grid[k].built = false;
            return fail(ctx, MLH_ERR_INVALID, "previous-frame cloud must be ordered by ring id (int(intensity) non-decreasing, 0 <= id < 255)");
        }
    return MLH_OK;
}

int mlh_track_set_prev(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int n, int intensity_offset_bytes, int mem,
                       float distance_sq_threshold)
{
    if (!ctx || kind < 0 || kind > 1 || intensity_offset_bytes < 0 || !(distance_sq_threshold > 0.f)) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int bad[2] = {0, 0};
    int rc = track_stage_prev(ctx, kind, points, stride_bytes, n, intensity_offset_bytes, mem, distance_sq_threshold, &bad[kind]);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    if ((rc = track_build_prev(ctx, 1 << kind, bad))) return rc;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));      // host buffers: the staging copy is reused by the next call
    return MLH_OK;
}

static int track_stage_cur(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int m, int intensity_offset_bytes, int mem)
{
    TrackSet &T = ctx->track;
    int rc = stage_points(ctx, points, stride_bytes, m, mem, intensity_offset_bytes >= 0 ? intensity_offset_bytes : -1, -1, T.cur[kind], nullptr, ctx->tmp);
    if (rc) return rc;
    MLH_HIP(ctx, T.corr[kind].ensure(sizeof(Corr) * size_t(m)));
    T.m[kind] = m;
    return MLH_OK;
}

int mlh_track_set_cur(mlh_ctx *ctx, int kind, const void *points, int stride_bytes, int m, int intensity_offset_bytes, int mem)
{
    if (!ctx || kind < 0 || kind > 1) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = track_stage_cur(ctx, kind, points, stride_bytes, m, intensity_offset_bytes, mem);
    if (rc) return rc;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLH_OK;
}

// the list sizes of the scan the context holds (and the thinned less-flat count once it exists): one host round trip per scan
static int scan_totals(mlh_ctx *ctx, bool want_vox)
{
    ScanBuf &sb = ctx->scan;
    const bool need_lists = !sb.h_lists_valid, need_vox = want_vox && !sb.h_vox_valid;
    if (!need_lists && !need_vox) return MLH_OK;
    if (need_lists) MLH_HIP(ctx, hipMemcpyAsync(sb.h_totals, sb.totals.p, sizeof(int) * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (need_vox || (sb.voxelised && !sb.h_vox_valid))
        MLH_HIP(ctx, hipMemcpyAsync(sb.h_totals + 4, sb.ring_vox.as<int>() + 2 * sb.n_rings, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sb.h_lists_valid = true;
    if (sb.voxelised) sb.h_vox_valid = true;
    return MLH_OK;
}

// the features of the scan the context holds become the tracker's current (which = 0: sharp corners, flat surfs) or previous
// (which = 1: less-sharp corners, voxel-thinned less-flat surfs) frame, device to device
int mlh_track_set_from_scan(mlh_ctx *ctx, int which, float distance_sq_threshold)
{
    if (!ctx || which < 0 || which > 1) return MLH_ERR_INVALID;
    ScanBuf &sb = ctx->scan;
    if (!sb.extracted) return fail(ctx, MLH_ERR_STATE, "extract_run has not been called");
    if (which == 1 && !sb.voxelised) return fail(ctx, MLH_ERR_STATE, "extract_voxel_run has not been called (the previous frame's surf cloud is the thinned one)");
    if (which == 1 && !(distance_sq_threshold > 0.f)) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = scan_totals(ctx, which == 1);
    if (rc) return rc;
    const int corner_list = which == 0 ? 0 : 1;                 // corner_points_sharp / corner_points_less_sharp
    const int n_corner = sb.h_totals[corner_list], n_flat = sb.h_totals[2], n_vox = sb.h_totals[4];
    if (n_corner <= 0 || (which == 0 ? n_flat : n_vox) <= 0) return fail(ctx, MLH_ERR_STATE, "the scan produced no features of one kind");
    MLH_HIP(ctx, ctx->knn_q.ensure(sizeof(float4) * size_t(std::max(n_corner, n_flat))));     // gather scratch (not ctx->tmp: the staging calls may use that)
    float4 *g = ctx->knn_q.as<float4>();
    gather_points_launch(ctx, sb.pts.as<float4>(), sb.lists[corner_list].as<int>(), n_corner, g);
    if (which == 0) {
        // nothing to wait for: every consumer of these buffers is a later launch on the same stream
        if ((rc = track_stage_cur(ctx, MLH_CORNER, g, 16, n_corner, 12, MLH_MEM_DEVICE))) return rc;
        gather_points_launch(ctx, sb.pts.as<float4>(), sb.lists[2].as<int>(), n_flat, g);
        return track_stage_cur(ctx, MLH_SURF, g, 16, n_flat, 12, MLH_MEM_DEVICE);
    }
    int bad[2] = {0, 0};
    if ((rc = track_stage_prev(ctx, MLH_CORNER, g, 16, n_corner, 12, MLH_MEM_DEVICE, distance_sq_threshold, &bad[MLH_CORNER]))) { (void)hipStreamSynchronize(st); return rc; }
    if ((rc = track_stage_prev(ctx, MLH_SURF, sb.vox_out.p, 16, n_vox, 12, MLH_MEM_DEVICE, distance_sq_threshold, &bad[MLH_SURF]))) { (void)hipStreamSynchronize(st); return rc; }
    return track_build_prev(ctx, 3, bad);      // both indices in one set of launches, one host round trip
}

int mlh_transform_to_end(mlh_ctx *ctx, void *points, int stride_bytes, int n, int intensity_offset_bytes, const double pose[7], int b_distortion,
                         float scan_period, int mem)
{
    if (!ctx || !points || !pose || n < 0 || stride_bytes < 16 || (stride_bytes & 3) || intensity_offset_bytes < 12 || !(scan_period > 0.f)) return MLH_ERR_INVALID;
    if (n == 0) return MLH_OK;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    if (mem == MLH_MEM_DEVICE) return transform_to_end_launch(ctx, points, stride_bytes, n, intensity_offset_bytes, pose, b_distortion, scan_period);
    const size_t bytes = size_t(n) * stride_bytes;
    MLH_HIP(ctx, ctx->tmp.ensure(bytes));
    MLH_HIP(ctx, hipMemcpyAsync(ctx->tmp.p, points, bytes, hipMemcpyHostToDevice, ctx->stream));
    int rc = transform_to_end_launch(ctx, ctx->tmp.p, stride_bytes, n, intensity_offset_bytes, pose, b_distortion, scan_period);
    if (rc) { (void)hipStreamSynchronize(ctx->stream); return rc; }
    MLH_HIP(ctx, hipMemcpyAsync(points, ctx->tmp.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLH_OK;
}

// Estimator::undistortMeasurements (estimator.cpp:376-410) for the scan the context holds: its points (laser_cloud, and with them the
// less-sharp corners the lists index) and its thinned less-flat cloud move to the end of the sweep, in place, on the device
int mlh_scan_undistort(mlh_ctx *ctx, const double pose_undist[7], float scan_period)
{
    if (!ctx || !pose_undist || !(scan_period > 0.f)) return MLH_ERR_INVALID;
    ScanBuf &sb = ctx->scan;
    if (!sb.extracted) return fail(ctx, MLH_ERR_STATE, "extract_run has not been called");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    { const int wrc = scan_wait_readers(ctx); if (wrc) return wrc; }
    int rc = transform_to_end_launch(ctx, sb.pts.p, 16, sb.n, 12, pose_undist, 1, scan_period);
    if (rc || !sb.voxelised) return rc;
    if ((rc = scan_totals(ctx, true))) return rc;      // the thinned cloud's count (fetched once per scan, shared with the hand-overs)
    return transform_to_end_launch(ctx, sb.vox_out.p, 16, sb.h_totals[4], 12, pose_undist, 1, scan_period);
}

int mlh_fuse_reset(mlh_ctx *ctx)
{
    if (!ctx) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    MLH_HIP(ctx, ctx->fused_cnt.ensure(sizeof(int) * 2 * 8));                      // prefix table of the record counts, one pair per append (grows);
                                                                                   // its first row is never read: the first append starts from zero by argument
    ctx->fused_n[0] = ctx->fused_n[1] = 0;
    ctx->fused_bound[0] = ctx->fused_bound[1] = 0;
    ctx->fused_parts = 0;
    for (int k = 0; k < 2; ++k) for (int d = 0; d < 6; ++d) ctx->fused_minmax[k][d] = d < 3 ? FLT_MAX : -FLT_MAX;
    ctx->fused_dirty = false;
    return MLH_OK;
}

int mlh_fuse_add_rings(mlh_ctx *ctx, int ring_begin, int ring_end, int lidar_idx, const double ext_pose[7])
{
    if (!ctx || !ext_pose || lidar_idx < 0) return MLH_ERR_INVALID;
    ScanBuf &sb = ctx->scan;
    if (!sb.extracted || !sb.voxelised) return fail(ctx, MLH_ERR_STATE, "mlh_extract_run and mlh_extract_voxel_run come first");
    if (ring_begin < 0 || ring_end > sb.n_rings || ring_begin >= ring_end) return fail(ctx, MLH_ERR_INVALID, "bad ring range");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->fused_cnt.p) { int rc = mlh_fuse_reset(ctx); if (rc) return rc; }
    { int rc = fuse_append_launch(ctx, sb, ring_begin, ring_end, lidar_idx, ext_pose); if (rc) return rc; }
    ctx->fused_dirty = true;
    return MLH_OK;
}

int mlh_fuse_add_scan(mlh_ctx *ctx, int lidar_idx, const double ext_pose[7])
{
    if (!ctx) return MLH_ERR_INVALID;
    return mlh_fuse_add_rings(ctx, 0, ctx->scan.n_rings, lidar_idx, ext_pose);
}

// mlh_fuse_add_scan with the scan ANOTHER context of the same device holds: every LiDAR's segmentCloud -> extractCloud on a context (a thread) of its own -- the
// host-side cluster searches side by side -- and one context gathers their features for the mapper without a host hop. Ordered on the device: ctx's stream waits for
// src's work so far (an event of ctx's on src's stream), and whatever rewrites src's scan next waits for this append (an event of src's on ctx's stream).
int mlh_fuse_add_scan_from(mlh_ctx *ctx, mlh_ctx *src, int lidar_idx, const double ext_pose[7])
{
    if (!ctx || !src || !ext_pose || lidar_idx < 0) return MLH_ERR_INVALID;
    if (src == ctx) return mlh_fuse_add_scan(ctx, lidar_idx, ext_pose);
    if (ctx->device != src->device) return fail(ctx, MLH_ERR_UNSUPPORTED, "mlh_fuse_add_scan_from: both contexts must be on the same device");
    ScanBuf &sb = src->scan;
    if (!sb.extracted || !sb.voxelised) return fail(ctx, MLH_ERR_STATE, "mlh_fuse_add_scan_from: mlh_extract_run and mlh_extract_voxel_run on the source context come first");
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->fused_cnt.p) { int rc = mlh_fuse_reset(ctx); if (rc) return rc; }
    if (!ctx->ev_handover) MLH_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_handover, hipEventDisableTiming));
    MLH_HIP(ctx, hipEventRecord(ctx->ev_handover, src->stream));
    MLH_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_handover, 0));
    { int rc = fuse_append_launch(ctx, sb, 0, sb.n_rings, lidar_idx, ext_pose); if (rc) return rc; }
    ctx->fused_dirty = true;
    if (!src->ev_scan_reader) MLH_HIP(ctx, hipEventCreateWithFlags(&src->ev_scan_reader, hipEventDisableTiming));
    MLH_HIP(ctx, hipEventRecord(src->ev_scan_reader, ctx->stream));
    src->scan_reader_pending.store(true, std::memory_order_release);
    return MLH_OK;
}

// The two record counts and the two bounding boxes of the fused clouds, reduced over the appends' per-workgroup partial boxes and written straight into pinned
// host memory by one launch (sequence word last, system-scope release): what used to be two copies and a marker launch behind them.
__global__ __launch_bounds__(256) void fused_publish_kernel(const int *__restrict__ cnt2, const float *__restrict__ part, int n_boxes_per_kind_and_part, int parts,
                                                            int *h_cnt, float *h_box, unsigned long long *h_seq, unsigned long long seq)
{
    __shared__ float s_red[4][12];
    const int t = threadIdx.x;
    float v[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) v[c] = (c % 6) < 3 ? FLT_MAX : -FLT_MAX;
    for (int i = t; i < parts * n_boxes_per_kind_and_part; i += 256) {          // one partial box of each kind per step: [append][kind][workgroup][6]
        const int a = i / n_boxes_per_kind_and_part, b = i - a * n_boxes_per_kind_and_part;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float *q = part + (size_t(a) * 2 * n_boxes_per_kind_and_part + size_t(k) * n_boxes_per_kind_and_part + b) * 6;
#pragma unroll
            for (int d = 0; d < 6; ++d) v[k * 6 + d] = d < 3 ? fminf(v[k * 6 + d], q[d]) : fmaxf(v[k * 6 + d], q[d]);
        }
    }
#pragma unroll
    for (int c = 0; c < 12; ++c) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { const float o = __shfl_xor(v[c], off, 64); v[c] = (c % 6) < 3 ? fminf(v[c], o) : fmaxf(v[c], o); }
    }
    if ((t & 63) == 0) {
#pragma unroll
        for (int c = 0; c < 12; ++c) s_red[t >> 6][c] = v[c];
    }
    __syncthreads();
    if (t < 12) {
        const bool lo = (t % 6) < 3;
        float r = s_red[0][t];
        for (int w = 1; w < 4; ++w) r = lo ? fminf(r, s_red[w][t]) : fmaxf(r, s_red[w][t]);
        h_box[t] = r;
    }
    if (t == 0) { h_cnt[0] = cnt2[0]; h_cnt[1] = cnt2[1]; }
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(h_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int mlh_fused_cloud(mlh_ctx *ctx, int kind, const void **device_points, int32_t *n)
{
    if (!ctx || kind < 0 || kind > 1 || !device_points || !n) return MLH_ERR_INVALID;
    if (ctx->fused_dirty) {
        MLH_HIP(ctx, hipSetDevice(ctx->device));
        if (ctx->fused_parts == 0) {                                // nothing appended since the reset
            ctx->fused_n[0] = ctx->fused_n[1] = 0;
            for (int k = 0; k < 2; ++k) for (int d = 0; d < 6; ++d) ctx->fused_minmax[k][d] = d < 3 ? FLT_MAX : -FLT_MAX;
            ctx->fused_dirty = false;
            *device_points = ctx->fused[kind].p;
            *n = 0;
            return MLH_OK;
        }
        if (!ctx->fused_host) {
            MLH_HIP(ctx, hipHostMalloc(&ctx->fused_host, 128, hipHostMallocDefault));
            std::memset(ctx->fused_host, 0, 128);
            ctx->fused_host_cap = 128;
        }
        int *h_cnt = static_cast<int *>(ctx->fused_host);
        float *h_box = reinterpret_cast<float *>(static_cast<char *>(ctx->fused_host) + 16);
        unsigned long long *h_seq = reinterpret_cast<unsigned long long *>(static_cast<char *>(ctx->fused_host) + 64);
        const unsigned long long seq = ++ctx->fused_seq;
        MLH_LAUNCH(fused_publish_kernel, dim3(1), dim3(256), 0, ctx->stream, (const int *)(ctx->fused_cnt.as<int>() + 2 * ctx->fused_parts),
                           (const float *)ctx->fused_part.as<float>(), FUSE_BLOCKS, ctx->fused_parts, h_cnt, h_box, h_seq, seq);
        MLH_HIP(ctx, hipGetLastError());
        {
            const auto t0 = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
                    if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) return fail(ctx, MLH_ERR_HIP, "the fused clouds' sizes did not arrive");
                    break;
                }
                host_wait_relax(spins);
            }
        }
        ctx->fused_n[0] = h_cnt[0]; ctx->fused_n[1] = h_cnt[1];
        for (int k = 0; k < 2; ++k) for (int d = 0; d < 6; ++d) ctx->fused_minmax[k][d] = h_box[k * 6 + d];
        ctx->fused_dirty = false;
    }
    *device_points = ctx->fused[kind].p;
    *n = ctx->fused_n[kind];
    return MLH_OK;
}

int mlh_track_match(mlh_ctx *ctx, int kind, const double pose[7], const mlh_track_opts *opts, uint8_t *valid, double *coeffs)
{
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }      // (a solve submitted with mlh_gn_solve_begin* may have left its last iteration as records)
    if (!ctx || kind < 0 || kind > 1 || !pose || !opts) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    TrackArgs a = track_args(opts, 0);
    a.init_pose = pose;
    if ((rc = track_match_launch(ctx, 1 << kind, a))) return rc;
    const int m = ctx->track.m[kind];
    std::vector<Corr> hc(m);
    MLH_HIP(ctx, hipMemcpyAsync(hc.data(), ctx->track.corr[kind].p, sizeof(Corr) * size_t(m), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < m; ++i) {
        if (valid) valid[i] = hc[i].valid ? 1 : 0;
        if (coeffs) for (int k = 0; k < 6; ++k) coeffs[size_t(i) * 6 + k] = double(hc[i].c[k]);
    }
    return MLH_OK;
}

static int track_cloud_impl(mlh_ctx *ctx, double pose_inout[7], const mlh_track_opts *opts, mlh_iter_stat *stats, bool allow_loop)
{
    { const int frc = gn_flush_pending(ctx); if (frc) return frc; }      // (a solve submitted with mlh_gn_solve_begin* may have left its last iteration as records)
    if (!ctx || !pose_inout || !opts || opts->max_outer <= 0 || opts->max_lm_iterations <= 0) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    TrackSet &T = ctx->track;
    if (!(T.grid[0].built && T.grid[1].built && T.m[0] > 0 && T.m[1] > 0)) return fail(ctx, MLH_ERR_STATE, "track_set_prev / track_set_cur are required for both kinds");
    int rc = ensure_state(ctx, opts->max_outer);
    if (rc) return rc;
    // Without per-round records the pose goes in with the first round's kernel arguments and comes back from the last launch through
    // pinned host memory: 2 + max_lm_iterations launches per round and nothing else. With records: the state is uploaded and read back.
    const bool lean = stats == nullptr;
    if (!lean && (rc = upload_pose(ctx, pose_inout))) return rc;
    unsigned long long seq = 0;
    // lean, one GPU, a frame's worth of features: a round is TWO launches -- the match and one launch that runs the round's LM loop to its end on the device
    // (track.hip: track_lm_loop_kernel); MLH_TRACK_LOOP=0 keeps 2 + max_lm_iterations launches per round (A/B; read at every call)
    {
        const char *e = std::getenv("MLH_TRACK_LOOP");
        const int tiles = (T.m[0] + 255) / 256 + (T.m[1] + 255) / 256;
        if (lean && allow_loop && !(e && std::atoi(e) == 0) && !distributed(ctx) && loop_tiles_ok(ctx, 2, tiles)) {
            HostPublish *rec = nullptr;
            if ((rc = publish_slot(ctx, &rec, &seq))) return rc;
            for (int outer = 0; outer < opts->max_outer; ++outer) {
                TrackArgs m = track_args(opts, 0);
                if (outer == 0) m.init_pose = pose_inout;
                if ((rc = track_match_launch(ctx, 3, m))) return rc;
                TrackArgs b = track_args(opts, 0);
                b.finish = 0; b.lm_max_it = opts->max_lm_iterations; b.lm_min_blocks = 10; b.stat_slot = -1;
                if (outer == 0) b.init_pose = pose_inout;
                if (outer == opts->max_outer - 1) { b.publish = rec; b.publish_seq = seq; }
                if ((rc = track_lm_loop_launch(ctx, 3, b))) return rc;
            }
            HostPublish hp;
            if ((rc = wait_published(ctx, seq, hp))) return rc;
            if (hp.done & 4) {           // a barrier given up on: the rounds again, from the pose the caller still holds, a launch per LM iteration
                note_loop_timeout(ctx, 2, tiles, true);
                return track_cloud_impl(ctx, pose_inout, opts, stats, false);
            }
            if (!ctx->prof.pending.empty()) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
            for (int i = 0; i < 7; ++i) pose_inout[i] = hp.x[i];
            return MLH_OK;
        }
    }
    for (int outer = 0; outer < opts->max_outer; ++outer) {
        // lidar_tracker.cpp:42-121: match at the current estimate, then Ceres on the fixed correspondences (Huber 0.1, <= 4 iterations,
        // no degeneracy handling); fewer than 10 correspondences -> the round is skipped
        // (the LM begin / step run in the linearisation kernel's last workgroup: 2 + max_lm_iterations launches per round)
        TrackArgs m = track_args(opts, 0);
        if (lean && outer == 0) m.init_pose = pose_inout;
        if ((rc = track_match_launch(ctx, 3, m))) return rc;
        TrackArgs b = track_args(opts, 0);
        b.finish = 3; b.lm_max_it = opts->max_lm_iterations; b.lm_min_blocks = 10; b.stat_slot = stats ? outer : -1;
        if (lean && outer == 0) b.init_pose = pose_inout;
        if ((rc = track_linearize_launch(ctx, 3, b))) return rc;
        for (int it = 0; it < opts->max_lm_iterations; ++it) {
            TrackArgs s = track_args(opts, 1);
            s.finish = 4; s.lm_max_it = opts->max_lm_iterations;
            if (lean && outer == opts->max_outer - 1 && it == opts->max_lm_iterations - 1) {
                if ((rc = publish_slot(ctx, &s.publish, &seq))) return rc;
                s.publish_seq = seq;
            }
            if ((rc = track_linearize_launch(ctx, 3, s))) return rc;
        }
        if (stats && (rc = lm_finish_launch(ctx, outer))) return rc;     // fills the record's LM summary
    }
    if (lean) {
        HostPublish hp;
        if ((rc = wait_published(ctx, seq, hp))) return rc;
        if (!ctx->prof.pending.empty()) { MLH_HIP(ctx, hipStreamSynchronize(ctx->stream)); prof_collect(ctx); }
        for (int i = 0; i < 7; ++i) pose_inout[i] = hp.x[i];
        return MLH_OK;
    }
    return fetch_pose_and_stats(ctx, pose_inout, stats, opts->max_outer);
}

int mlh_track_cloud(mlh_ctx *ctx, double pose_inout[7], const mlh_track_opts *opts, mlh_iter_stat *stats)
{
    return track_cloud_impl(ctx, pose_inout, opts, stats, true);
}

int mlh_good_feature_matching(mlh_ctx *ctx, int kind, const double pose[7], int gf_method, double gf_ratio, uint64_t seed,
                              float min_match_sq_dis, float min_plane_dis, int32_t *sel_idx, int32_t *n_sel,
                              double sub_mat_H[36], uint8_t *matched)
{
    if (!ctx || kind < 0 || kind > 1 || !pose || !sel_idx || !n_sel || !sub_mat_H) return MLH_ERR_INVALID;
    MLH_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_state(ctx, 0);
    if (rc) return rc;
    if ((rc = upload_pose(ctx, pose))) return rc;
    std::mt19937 rng((uint32_t)seed);
    std::vector<int32_t> sel;
    if ((rc = good_feature_select(ctx, kind, gf_method, gf_ratio, rng, min_match_sq_dis, min_plane_dis, sel, sub_mat_H, matched))) return rc;
    *n_sel = (int32_t)sel.size();
    std::copy(sel.begin(), sel.end(), sel_idx);
    return MLH_OK;
}

// ---------------------------------------------------------------- small host helpers
int mlh_pose_plus(const double x[7], const double delta[6], const double *V_update, double x_plus_delta[7])
{
    if (!x || !delta || !x_plus_delta) return MLH_ERR_INVALID;
    pose_plus(x, delta, V_update, x_plus_delta);
    return MLH_OK;
}

int mlh_eval_degeneracy(const double H[36], double eig_thre, double eigval[6], double V_update[36])
{
    if (!H || !eigval || !V_update) return MLH_ERR_INVALID;
    // cyclic Jacobi on the host (same procedure as the device kernel)
    double a[36], V[36];
    for (int i = 0; i < 36; ++i) a[i] = H[i];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) V[i * 6 + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < 6; ++i) { dg += a[i * 6 + i] * a[i * 6 + i]; for (int j = i + 1; j < 6; ++j) off += a[i * 6 + j] * a[i * 6 + j]; }
        if (off <= 1e-32 * dg || off == 0.0) break;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double apq = a[p * 6 + q];
                if (apq == 0.0) continue;
                double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) { double u = a[k * 6 + p], v = a[k * 6 + q]; a[k * 6 + p] = c * u - s * v; a[k * 6 + q] = s * u + c * v; }
                for (int k = 0; k < 6; ++k) { double u = a[p * 6 + k], v = a[q * 6 + k]; a[p * 6 + k] = c * u - s * v; a[q * 6 + k] = s * u + c * v; }
                for (int k = 0; k < 6; ++k) { double u = V[k * 6 + p], v = V[k * 6 + q]; V[k * 6 + p] = c * u - s * v; V[k * 6 + q] = s * u + c * v; }
            }
    }
    int order[6] = {0, 1, 2, 3, 4, 5};
    for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 6; ++j) if (a[order[j] * 7] < a[order[i] * 7]) std::swap(order[i], order[j]);
    bool deg = false, stop = false;
    bool keep[6];
    for (int j = 0; j < 6; ++j) {
        eigval[j] = a[order[j] * 7];
        if (!stop && eigval[j] < eig_thre) { keep[j] = false; deg = true; } else { keep[j] = true; stop = true; }
    }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            if (deg) { for (int j = 0; j < 6; ++j) if (keep[j]) s += V[r * 6 + order[j]] * V[c * 6 + order[j]]; }
            else s = (r == c) ? 1.0 : 0.0;
            V_update[r * 6 + c] = s;
        }
    return deg ? 1 : 0;
}

}  // extern "C"
