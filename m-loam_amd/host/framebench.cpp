// Mapper frames timed THROUGH THE C-ABI FROM C++ (scripts/framebench.py times the same sequence through ctypes: a dozen calls per frame, each with its interpreter
// overhead). A frame: two raw 64-ring scans in (host buffers, one joint upload), extractCloud + per-ring voxel grid, fusion, downsampleCurrentScan for both kinds,
// index build of the local map, scan2MapOptimization, pose out. Inputs are the files scripts/framebench.py / bench.py write (the bench workload).
//   usage: framebench <dir> [frames] [mode] [pipeline counts, e.g. 1,2,4]
//   mode single   (default) one context, one thread: ms per frame by stage, index rebuilt on the critical path / staged beside the front end   (text lines)
//        two_ctx  the reference's process structure on one GPU: an ESTIMATOR-side thread (upload -> extract -> fuse -> thin; estimator.cpp:100 process_thread_,
//                 :248-270) and a MAPPER-side thread (index -> scan2map; lidar_mapper_keyframe.cpp:1315 mapping_process) on two contexts, device-to-device
//                 hand-over (mlh_features_copy): frame PERIOD                                                                                 (one JSON line)
//        pipes    K = 1 / 2 / 4 independent frame pipelines (K sensor rigs / K bags replayed at once), each a thread with its own context running whole
//                 frames back to back: aggregate frames per second                                                                            (one JSON line)
//        all      two_ctx + pipes in one JSON line (what bench.py puts into its `frame` object)
//        raw      the frame from RAW clouds, ImageSegmenter included (the reference's default front end: segmentCloud -> extractCloud per LiDAR, estimator.cpp:248-263,
//                 then fusion -> thinning -> index -> scan2map): one context, the LiDARs one after the other | a context + thread per LiDAR (the host-side cluster
//                 searches side by side) gathered by mlh_fuse_add_scan_from | the same with the next frame's front end started behind the appends (period)  (one JSON line)
#include "../../include/mloam_hip.h"
#include <hip/hip_runtime_api.h>   // only for the device-resident copy of the local map the loops stage from
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

template <typename T> static std::vector<T> read_file(const std::string &p)
{
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", p.c_str()); std::exit(2); }
    const size_t n = size_t(f.tellg()) / sizeof(T);
    std::vector<T> v(n);
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), std::streamsize(n * sizeof(T)));
    return v;
}
#define CK(x) do { const int rc_ = (x); if (rc_) { std::fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mlh_last_error(ctx)); return 1; } } while (0)

using Clock = std::chrono::steady_clock;
static double ms_between(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

struct Work {
    std::vector<float> pts, surf_map, corner_map;
    std::vector<int32_t> rings, ring_ofs, meta;
    std::vector<double> ext, covs, meas, p0;
    int R = 0, n = 0, n_lidar = 0, map_stride = 0, n_surf_map = 0, n_corner_map = 0;
    void *d_surf = nullptr, *d_corner = nullptr;     // the local map, device-resident (a mapper that assembles it from keyframe clouds on the GPU has it there)
    void *d_pts = nullptr, *d_rings = nullptr;       // FB_DEVICE_SCAN=1: the scans resident in HBM too (no upload inside the frame: bench.py's `frame` convention)
    mlh_solver_opts o;
};

static int front_end(mlh_ctx *ctx, const Work &W, const void *d_pts_now = nullptr, const void *d_rings_now = nullptr)
{
    CK(mlh_fuse_reset(ctx));
    if (d_pts_now) CK(mlh_scan_upload(ctx, d_pts_now, 16, 12, W.n, static_cast<const int32_t *>(d_rings_now), static_cast<const int32_t *>(d_rings_now) + W.R, W.R, MLH_MEM_DEVICE));
    else if (W.d_pts) CK(mlh_scan_upload(ctx, W.d_pts, 16, 12, W.n, static_cast<const int32_t *>(W.d_rings), static_cast<const int32_t *>(W.d_rings) + W.R, W.R, MLH_MEM_DEVICE));
    else CK(mlh_scan_upload(ctx, W.pts.data(), 16, 12, W.n, W.rings.data(), W.rings.data() + W.R, W.R, MLH_MEM_HOST));
    CK(mlh_extract_run(ctx));
    CK(mlh_extract_voxel_run(ctx, 0.2f));
    for (int i = 0; i < W.n_lidar; ++i) CK(mlh_fuse_add_rings(ctx, W.ring_ofs[size_t(i)], W.ring_ofs[size_t(i) + 1], i, W.ext.data() + 7 * i));
    return 0;
}
static int thin(mlh_ctx *ctx, const Work &W, int32_t *ms_, int32_t *mc)
{
    const void *fs = nullptr, *fc = nullptr;
    int32_t ns = 0, nc = 0;
    CK(mlh_fused_cloud(ctx, MLH_SURF, &fs, &ns));
    CK(mlh_fused_cloud(ctx, MLH_CORNER, &fc, &nc));
    CK(mlh_downsample_current_scan_pair(ctx, fs, ns, fc, nc, 16, 12, MLH_MEM_DEVICE, 0.4f, 0.2f, W.ext.data(), W.covs.data(), W.n_lidar, W.meas.data(), W.meta[1], 0.6, ms_, mc));
    return 0;
}
static int stage_map_beside(mlh_ctx *ctx, const Work &W)
{
    CK(mlh_map_set_pair_overlapped(ctx, W.d_surf, W.n_surf_map, W.d_corner, W.n_corner_map, W.map_stride, 1.0f, MLH_MEM_DEVICE));
    return 0;
}
// one whole frame on one context, the local map staged and indexed beside the front end; st (nullable): host wall time of the four parts, accumulated [ms]
// FB_ONE_STREAM=1: the index is rebuilt on the context's own stream between thinning and solve instead -- one stream per pipeline. A process gets FOUR hardware
// queues out of this GPU (scripts/exp/launch_rate.hip: streams beyond four share queues and their kernels serialise; asking the runtime for more queues is
// slower still), so K = 4 pipelines with a staging stream each are eight streams on four queues.
static bool g_one_stream = std::getenv("FB_ONE_STREAM") && std::atoi(std::getenv("FB_ONE_STREAM")) != 0;
// FB_LIB_AHEAD=1: every whole frame sends the NEXT frame's scan ahead (mlh_scan_upload_ahead) right behind its front end's launches (A/B runs)
static bool g_lib_ahead = std::getenv("FB_LIB_AHEAD") && std::atoi(std::getenv("FB_LIB_AHEAD")) != 0;
static int whole_frame(mlh_ctx *ctx, const Work &W, double pose[7], double *st = nullptr)
{
    const auto t0 = Clock::now();
    if (front_end(ctx, W)) return 1;
    if (g_lib_ahead && !W.d_pts) CK(mlh_scan_upload_ahead(ctx, W.pts.data(), 16, W.n));
    const auto t1 = Clock::now();
    if (!g_one_stream && stage_map_beside(ctx, W)) return 1;
    const auto t2 = Clock::now();
    int32_t a = 0, b = 0;
    if (thin(ctx, W, &a, &b)) return 1;
    const auto t3 = Clock::now();
    if (g_one_stream) CK(mlh_map_rebuild(ctx, MLH_ALL_KINDS));
    for (int i = 0; i < 7; ++i) pose[i] = W.p0[size_t(i)];
    CK(mlh_scan2map(ctx, pose, &W.o, nullptr));
    if (st) { const auto t4 = Clock::now(); st[0] += ms_between(t0, t1); st[1] += ms_between(t1, t2); st[2] += ms_between(t2, t3); st[3] += ms_between(t3, t4); }
    return 0;
}
static mlh_ctx *make_ctx(const Work &W)
{
    mlh_ctx *ctx = nullptr;
    if (mlh_create(&ctx, 0)) { std::fprintf(stderr, "no GPU context\n"); return nullptr; }
    if (mlh_map_set_pair(ctx, W.d_surf, W.n_surf_map, W.d_corner, W.n_corner_map, W.map_stride, 1.0f, MLH_MEM_DEVICE)) { std::fprintf(stderr, "map_set_pair: %s\n", mlh_last_error(ctx)); return nullptr; }
    return ctx;
}
static bool same_pose(const double a[7], const double b[7]) { return std::memcmp(a, b, 7 * sizeof(double)) == 0; }

// a counter the other thread waits on (C++17: no std::counting_semaphore); the waits are tens of microseconds
struct Signal {
    std::atomic<long> v{0};
    void post() { v.fetch_add(1, std::memory_order_release); }
    void wait_for(long k) { unsigned s = 0; while (v.load(std::memory_order_acquire) < k) { if (++s > 256) std::this_thread::yield(); } }
};

// ---- two contexts, two threads: the frame period of the estimator / mapper pair
// prefetch: the estimator side's NEXT scan goes to the device while the current one is being worked on -- what a caller that has its scans ahead of time (a bag
// replayed faster than real time; a driver that fills pinned buffers) does with plain HIP calls and MLH_MEM_DEVICE: scans in page-locked host memory, a copy stream of
// the caller's own, two device buffers, an event per buffer. Nothing of the library changes; the upload leaves the estimator side's chain (copy engine beside compute).
struct Prefetch {
    hipStream_t cs = nullptr;
    void *d_pts[2] = {nullptr, nullptr}, *d_rings[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    bool registered[2] = {false, false};
    int setup(const Work &W)
    {
        if (hipStreamCreateWithFlags(&cs, hipStreamNonBlocking) != hipSuccess) return 1;
        for (int i = 0; i < 2; ++i)
            if (hipMalloc(&d_pts[i], W.pts.size() * 4) != hipSuccess || hipMalloc(&d_rings[i], W.rings.size() * 4) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess) return 1;
        registered[0] = hipHostRegister(const_cast<float *>(W.pts.data()), W.pts.size() * 4, hipHostRegisterDefault) == hipSuccess;
        registered[1] = hipHostRegister(const_cast<int32_t *>(W.rings.data()), W.rings.size() * 4, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();
        return 0;
    }
    int issue(const Work &W, int k)
    {
        const int b = k & 1;
        if (hipMemcpyAsync(d_pts[b], W.pts.data(), W.pts.size() * 4, hipMemcpyHostToDevice, cs) != hipSuccess ||
            hipMemcpyAsync(d_rings[b], W.rings.data(), W.rings.size() * 4, hipMemcpyHostToDevice, cs) != hipSuccess || hipEventRecord(ev[b], cs) != hipSuccess) return 1;
        return 0;
    }
    int arrived(int k) { return hipEventSynchronize(ev[k & 1]) == hipSuccess ? 0 : 1; }
    void release(const Work &W)
    {
        if (cs) (void)hipStreamSynchronize(cs);
        if (registered[0]) (void)hipHostUnregister(const_cast<float *>(W.pts.data()));
        if (registered[1]) (void)hipHostUnregister(const_cast<int32_t *>(W.rings.data()));
        for (int i = 0; i < 2; ++i) { (void)hipFree(d_pts[i]); (void)hipFree(d_rings[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
        if (cs) (void)hipStreamDestroy(cs);
    }
};

static int run_two_ctx(const Work &W, int frames, const double ref_pose[7], double *period_ms, bool *same, double *est_alone_ms, double *map_alone_ms, bool prefetch = false,
                       bool library_ahead = false)
{
    mlh_ctx *E = make_ctx(W), *M = make_ctx(W);
    if (!E || !M) return 1;
    const int warm = 5, total = frames + warm;
    Prefetch P;
    if (prefetch && P.setup(W)) { std::fprintf(stderr, "prefetch set-up failed\n"); return 1; }
    // each side alone first (what the period cannot be shorter than)
    {
        mlh_ctx *ctx = E;
        int32_t a = 0, b = 0;
        if (prefetch && P.issue(W, 0)) return 1;
        auto est_frame = [&](int k) -> int {
            // library_ahead: mlh_scan_upload_ahead on the caller's own (pageable) buffer right behind the front end's launches -- the next frame's mlh_scan_upload finds it
            if (library_ahead) return (front_end(ctx, W) || mlh_scan_upload_ahead(ctx, W.pts.data(), 16, W.n) || thin(ctx, W, &a, &b)) ? 1 : 0;
            if (!prefetch) return (front_end(ctx, W) || thin(ctx, W, &a, &b)) ? 1 : 0;
            if (P.arrived(k) || front_end(ctx, W, P.d_pts[k & 1], P.d_rings[k & 1]) || P.issue(W, k + 1) || thin(ctx, W, &a, &b)) return 1;
            return 0;
        };
        for (int k = 0; k < 3; ++k) if (est_frame(k)) return 1;
        const auto t0 = Clock::now();
        for (int k = 0; k < frames; ++k) if (est_frame(k + 3)) return 1;
        *est_alone_ms = ms_between(t0, Clock::now()) / frames;
        if (prefetch && P.arrived(frames + 3)) return 1;              // the last look-ahead copy: nothing reads it
        ctx = M;
        double pose[7];
        CK(mlh_features_copy(M, E, MLH_SURF)); CK(mlh_features_copy(M, E, MLH_CORNER));
        for (int k = 0; k < 3; ++k) { CK(mlh_map_rebuild(ctx, MLH_ALL_KINDS)); for (int i = 0; i < 7; ++i) pose[i] = W.p0[size_t(i)]; CK(mlh_scan2map(ctx, pose, &W.o, nullptr)); }
        const auto t1 = Clock::now();
        for (int k = 0; k < frames; ++k) {
            CK(mlh_features_copy(M, E, MLH_SURF)); CK(mlh_features_copy(M, E, MLH_CORNER));
            CK(mlh_map_rebuild(ctx, MLH_ALL_KINDS));
            for (int i = 0; i < 7; ++i) pose[i] = W.p0[size_t(i)];
            CK(mlh_scan2map(ctx, pose, &W.o, nullptr));
        }
        *map_alone_ms = ms_between(t1, Clock::now()) / frames;
    }
    Signal ready, copied;
    std::atomic<int> failed{0};
    std::vector<double> poses(size_t(total) * 7, 0.0);
    Clock::time_point t_start, t_end;
    std::thread est([&] {
        mlh_ctx *ctx = E;
        for (int k = 0; k < total && !failed.load(); ++k) {
            if (k == warm) t_start = Clock::now();
            int32_t a = 0, b = 0;
            if (prefetch) {
                if (k == 0 && P.issue(W, 0)) { failed = 1; break; }
                if (P.arrived(k) || front_end(ctx, W, P.d_pts[k & 1], P.d_rings[k & 1]) || (k + 1 < total && P.issue(W, k + 1))) { failed = 1; break; }
            } else if (front_end(ctx, W) || (library_ahead && k + 1 < total && mlh_scan_upload_ahead(ctx, W.pts.data(), 16, W.n))) { failed = 1; break; }
            copied.wait_for(k);                      // the mapper side has taken frame k - 1's features: this context's sets may be overwritten
            if (thin(ctx, W, &a, &b)) { failed = 1; break; }
            ready.post();
        }
        ready.v.store(1L << 40);
    });
    std::thread map([&] {
        mlh_ctx *ctx = M;
        for (int k = 0; k < total && !failed.load(); ++k) {
            // the local map does not wait for the scan (it is made of earlier keyframes): indexed while the estimator side is still busy with frame k
            if (mlh_map_rebuild(ctx, MLH_ALL_KINDS)) { failed = 1; break; }
            ready.wait_for(k + 1);
            if (failed.load()) break;
            if (mlh_features_copy(M, E, MLH_SURF) || mlh_features_copy(M, E, MLH_CORNER)) { std::fprintf(stderr, "features_copy: %s\n", mlh_last_error(M)); failed = 1; break; }
            copied.post();
            double *pose = poses.data() + size_t(k) * 7;
            for (int i = 0; i < 7; ++i) pose[i] = W.p0[size_t(i)];
            if (mlh_scan2map(ctx, pose, &W.o, nullptr)) { std::fprintf(stderr, "scan2map: %s\n", mlh_last_error(M)); failed = 1; break; }
        }
        t_end = Clock::now();
        copied.v.store(1L << 40);
    });
    est.join(); map.join();
    if (failed.load()) return 1;
    *period_ms = ms_between(t_start, t_end) / frames;
    *same = true;
    for (int k = 0; k < total; ++k) *same = *same && same_pose(poses.data() + size_t(k) * 7, ref_pose);
    mlh_destroy(E); mlh_destroy(M);
    if (prefetch) P.release(W);
    return 0;
}

// ---- K independent pipelines: aggregate frames per second
static int run_pipes(const Work &W, int K, int frames, const double ref_pose[7], double *fps, bool *same, unsigned long long *timeouts, double stage_ms[4])
{
    // more than two pipelines: one stream each (the index on the pipeline's own stream), unless FB_ONE_STREAM says otherwise -- see whole_frame
    if (!std::getenv("FB_ONE_STREAM")) g_one_stream = K > 2;
    std::vector<double> st(size_t(K) * 4, 0.0);
    std::vector<mlh_ctx *> ctxs;
    for (int i = 0; i < K; ++i) { mlh_ctx *c = make_ctx(W); if (!c) return 1; ctxs.push_back(c); }
    std::atomic<int> failed{0}, started{0};
    std::atomic<bool> go{false};
    std::vector<char> ok(size_t(K), 1);
    std::vector<std::thread> th;
    for (int i = 0; i < K; ++i)
        th.emplace_back([&, i] {
            double pose[7];
            for (int k = 0; k < 5; ++k) if (whole_frame(ctxs[size_t(i)], W, pose)) { failed = 1; }
            started.fetch_add(1);
            while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
            for (int k = 0; k < frames && !failed.load(); ++k) {
                if (whole_frame(ctxs[size_t(i)], W, pose, st.data() + 4 * size_t(i))) { failed = 1; break; }
                if (!same_pose(pose, ref_pose)) ok[size_t(i)] = 0;
            }
        });
    while (started.load() < K) std::this_thread::yield();
    const auto t0 = Clock::now();
    go.store(true, std::memory_order_release);
    for (auto &t : th) t.join();
    const double ms = ms_between(t0, Clock::now());
    if (failed.load()) return 1;
    *fps = 1e3 * double(K) * double(frames) / ms;
    *same = true;
    for (char c : ok) *same = *same && c;
    for (int j = 0; j < 4; ++j) { stage_ms[j] = 0.0; for (int i = 0; i < K; ++i) stage_ms[j] += st[size_t(i) * 4 + size_t(j)] / (double(K) * frames); }
    *timeouts = 0;
    for (mlh_ctx *c : ctxs) { mlh_device_info di; if (!mlh_get_info(c, &di)) *timeouts += di.loop_timeouts; mlh_destroy(c); }
    return 0;
}

// ---- the frame from raw clouds, segmenter included
struct RawLidar { const float *pts; int n; int rings; };
static int seg_front(mlh_ctx *ctx, const RawLidar &L)
{
    mlh_segment_params prm;
    mlh_segment_params_default(&prm);
    prm.vertical_scans = L.rings; prm.horizon_scans = 1800; prm.segment_flag = 1;
    int32_t n_out = 0, n_outl = 0;
    CK(mlh_segment_cloud(ctx, L.pts, 16, -1, L.n, MLH_MEM_HOST, &prm, nullptr, &n_out, nullptr, nullptr, nullptr, 0, &n_outl));
    CK(mlh_extract_run(ctx));
    CK(mlh_extract_voxel_run(ctx, 0.2f));
    return 0;
}
static int raw_back(mlh_ctx *ctx, const Work &W, double pose[7])
{
    int32_t a = 0, b = 0;
    if (thin(ctx, W, &a, &b)) return 1;
    CK(mlh_map_rebuild(ctx, MLH_ALL_KINDS));
    for (int i = 0; i < 7; ++i) pose[i] = W.p0[size_t(i)];
    CK(mlh_scan2map(ctx, pose, &W.o, nullptr));
    return 0;
}
static int run_raw(const Work &W, int frames)
{
    std::vector<RawLidar> Ls;
    for (int l = 0; l < W.n_lidar; ++l) {
        const int r0 = W.ring_ofs[size_t(l)], r1 = W.ring_ofs[size_t(l) + 1];
        const int b = W.rings[size_t(r0)] - 5, e = W.rings[size_t(W.R + r1 - 1)] + 6;          // ScanInfo's insets (image_segmenter.hpp:385-387)
        if (r1 - r0 != 16 && r1 - r0 != 32 && r1 - r0 != 64) { std::fprintf(stderr, "raw mode: %d rings per LiDAR\n", r1 - r0); return 1; }
        Ls.push_back(RawLidar{W.pts.data() + 4 * size_t(b), e - b, r1 - r0});
    }
    const int L = int(Ls.size()), warm = 5;
    // (a) one context, one thread
    mlh_ctx *ctx = make_ctx(W);
    if (!ctx) return 1;
    double pose_serial[7], pose[7];
    auto serial_frame = [&](double out[7]) -> int {
        CK(mlh_fuse_reset(ctx));
        for (int l = 0; l < L; ++l) { if (seg_front(ctx, Ls[size_t(l)])) return 1; CK(mlh_fuse_add_scan(ctx, l, W.ext.data() + 7 * l)); }
        return raw_back(ctx, W, out);
    };
    for (int k = 0; k < warm; ++k) if (serial_frame(pose_serial)) return 1;
    auto t0 = Clock::now();
    for (int k = 0; k < frames; ++k) if (serial_frame(pose)) return 1;
    const double ms_serial = ms_between(t0, Clock::now()) / frames;
    bool same = same_pose(pose, pose_serial);
    // (b), (c) a context + a persistent thread per LiDAR; the first context gathers
    std::vector<mlh_ctx *> lane;
    lane.resize(size_t(L), nullptr);
    for (auto &c : lane) { if (mlh_create(&c, 0)) return 1; }
    const size_t n_lanes = size_t(L);
    std::vector<Signal> go(n_lanes), done(n_lanes);
    std::atomic<int> failed{0};
    std::atomic<bool> quit{false};
    std::vector<std::thread> th;
    for (int l = 0; l < L; ++l)
        th.emplace_back([&, l] {
            for (long k = 1;; ++k) {
                go[size_t(l)].wait_for(k);
                if (quit.load()) return;
                if (seg_front(lane[size_t(l)], Ls[size_t(l)])) failed = 1;
                done[size_t(l)].post();
            }
        });
    long issued = 0;
    auto lanes_start = [&] { ++issued; for (int l = 0; l < L; ++l) go[size_t(l)].post(); };
    auto gather = [&]() -> int {
        for (int l = 0; l < L; ++l) done[size_t(l)].wait_for(issued);
        if (failed.load()) return 1;
        CK(mlh_fuse_reset(ctx));
        for (int l = 0; l < L; ++l) CK(mlh_fuse_add_scan_from(ctx, lane[size_t(l)], l, W.ext.data() + 7 * l));
        return 0;
    };
    double ms_lanes = 0, ms_period = 0;
    for (int pipelined = 0; pipelined < 2; ++pipelined) {
        lanes_start();
        Clock::time_point t1;
        for (int k = -warm; k < frames; ++k) {
            if (k == 0) t1 = Clock::now();
            if (gather()) return 1;
            const bool more = k + 1 < frames;
            if (pipelined && more) lanes_start();              // the next frame's front end behind the appends (ordered by events), beside this frame's thinning + solve
            if (raw_back(ctx, W, pose)) return 1;
            same = same && same_pose(pose, pose_serial);
            if (!pipelined && more) lanes_start();
        }
        (pipelined ? ms_period : ms_lanes) = ms_between(t1, Clock::now()) / frames;
    }
    quit = true;
    for (int l = 0; l < L; ++l) go[size_t(l)].v.store(1L << 40);
    for (auto &t : th) t.join();
    std::printf("{\"raw_frame_lidars\": %d, \"frames\": %d, \"raw_frame_ms_one_context\": %.4f, \"raw_frame_ms_context_per_lidar\": %.4f, \"raw_frame_period_ms_context_per_lidar_pipelined\": %.4f, "
                "\"raw_frame_same_pose\": %s}\n", L, frames, ms_serial, ms_lanes, ms_period, same ? "true" : "false");
    for (auto c : lane) mlh_destroy(c);
    mlh_destroy(ctx);
    return same ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <dir> [frames] [single|two_ctx|pipes|all]\n", argv[0]); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    const int frames = argc > 2 ? std::atoi(argv[2]) : 50;
    const std::string mode = argc > 3 ? argv[3] : "single";
    Work W;
    W.pts = read_file<float>(d + "fb_points.f32");           // both scans, rings back to back: x y z intensity
    W.rings = read_file<int32_t>(d + "fb_rings.i32");        // [start (R)] [end (R)]
    W.ring_ofs = read_file<int32_t>(d + "fb_ring_ofs.i32");  // ring range of every LiDAR (n_lidar + 1)
    W.ext = read_file<double>(d + "fb_ext.f64");             // n_lidar x 7
    W.covs = read_file<double>(d + "fb_covs.f64");           // n_lidar x 36
    W.meas = read_file<double>(d + "fb_meas.f64");           // 9
    W.surf_map = read_file<float>(d + "fb_surf_map.f32"); W.corner_map = read_file<float>(d + "fb_corner_map.f32");   // x y z (+ fields): stride in fb_meta
    W.meta = read_file<int32_t>(d + "fb_meta.i32");          // [map stride bytes, with_ua]
    W.p0 = read_file<double>(d + "fb_pose.f64");
    W.R = int(W.rings.size() / 2); W.n = int(W.pts.size() / 4); W.n_lidar = int(W.ring_ofs.size()) - 1; W.map_stride = W.meta[0];
    W.n_surf_map = int(W.surf_map.size() * 4 / size_t(W.map_stride)); W.n_corner_map = int(W.corner_map.size() * 4 / size_t(W.map_stride));
    mlh_solver_opts_default(&W.o);
    if (W.meta[1]) W.o.flags |= MLH_FLAG_WITH_UA;
    mlh_ctx *ctx = nullptr;
    if (mlh_create(&ctx, 0)) { std::fprintf(stderr, "no GPU context\n"); return 1; }
    if (hipMalloc(&W.d_surf, W.surf_map.size() * 4) != hipSuccess || hipMalloc(&W.d_corner, W.corner_map.size() * 4) != hipSuccess ||
        hipMemcpy(W.d_surf, W.surf_map.data(), W.surf_map.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(W.d_corner, W.corner_map.data(), W.corner_map.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { std::fprintf(stderr, "device copy of the maps failed\n"); return 1; }
    if (std::getenv("FB_DEVICE_SCAN") && std::atoi(std::getenv("FB_DEVICE_SCAN")) != 0) {
        if (hipMalloc(&W.d_pts, W.pts.size() * 4) != hipSuccess || hipMalloc(&W.d_rings, W.rings.size() * 4) != hipSuccess ||
            hipMemcpy(W.d_pts, W.pts.data(), W.pts.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(W.d_rings, W.rings.data(), W.rings.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { std::fprintf(stderr, "device copy of the scans failed\n"); return 1; }
    }
    CK(mlh_map_set_pair(ctx, W.surf_map.data(), W.n_surf_map, W.corner_map.data(), W.n_corner_map, W.map_stride, 1.0f, MLH_MEM_HOST));
    double pose[7], t_stage[3] = {0, 0, 0};
    if (mode == "single") {
        // mode 0: the index of the local map rebuilt between the thinning and the solve (on the frame's critical path);
        // mode 1: the local map staged and indexed beside the front end: mlh_map_set_pair_overlapped after the front end's launches are enqueued, before the
        //         first call that waits for them
        for (int m = 0; m < 2; ++m) {
            t_stage[0] = t_stage[1] = t_stage[2] = 0.0;
            for (int it = -5; it < frames; ++it) {
                const auto t0 = Clock::now();
                if (front_end(ctx, W)) return 1;
                if (m == 1 && stage_map_beside(ctx, W)) return 1;
                const auto t1 = Clock::now();
                int32_t ms_ = 0, mc = 0;
                if (thin(ctx, W, &ms_, &mc)) return 1;
                const auto t2 = Clock::now();
                if (m == 0) CK(mlh_map_rebuild(ctx, MLH_ALL_KINDS));
                for (int i = 0; i < 7; ++i) pose[i] = W.p0[size_t(i)];
                CK(mlh_scan2map(ctx, pose, &W.o, nullptr));
                const auto t3 = Clock::now();
                if (it >= 0) { t_stage[0] += ms_between(t0, t1); t_stage[1] += ms_between(t1, t2); t_stage[2] += ms_between(t2, t3); }
            }
            std::printf("%s, ms per frame: upload+extract+fuse%s %.3f downsample %.3f scan2map %.3f total %.3f  pose %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n",
                        m == 0 ? "C++ over the C-ABI, device-resident, one launch set, both kinds thinned in one pipeline"
                               : "  the same with the local map staged and indexed beside the front end (second stream, other map set)",
                        m == 0 ? "" : "+map staging", t_stage[0] / frames, t_stage[1] / frames, t_stage[2] / frames, (t_stage[0] + t_stage[1] + t_stage[2]) / frames,
                        pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], pose[6]);
        }
        mlh_destroy(ctx);
        return 0;
    }
    if (mode == "raw") { mlh_destroy(ctx); return run_raw(W, frames); }
    // the pose every pipeline has to reproduce, and one pipeline's latency
    double ref_pose[7];
    for (int k = 0; k < 5; ++k) if (whole_frame(ctx, W, ref_pose)) return 1;
    const auto tl = Clock::now();
    for (int k = 0; k < frames; ++k) if (whole_frame(ctx, W, pose)) return 1;
    const double latency_ms = ms_between(tl, Clock::now()) / frames;
    mlh_destroy(ctx);
    std::printf("{\"frames\": %d, \"ms_per_frame_one_pipeline\": %.4f", frames, latency_ms);
    if (mode == "two_ctx" || mode == "all") {
        double period = 0, ea = 0, ma = 0;
        bool same = false;
        if (run_two_ctx(W, frames, ref_pose, &period, &same, &ea, &ma)) return 1;
        std::printf(", \"period_ms_two_contexts\": %.4f, \"two_contexts_same_pose\": %s, \"estimator_side_alone_ms\": %.4f, \"mapper_side_alone_ms\": %.4f", period, same ? "true" : "false", ea, ma);
        // the same pair with the next scan's upload issued ahead by the caller (page-locked scans, a copy stream, MLH_MEM_DEVICE): the copy engine beside compute
        double period_p = 0, ea_p = 0, ma_p = 0;
        bool same_p = false;
        if (run_two_ctx(W, frames, ref_pose, &period_p, &same_p, &ea_p, &ma_p, true)) return 1;
        std::printf(", \"period_ms_two_contexts_upload_ahead\": %.4f, \"upload_ahead_same_pose\": %s, \"estimator_side_alone_upload_ahead_ms\": %.4f", period_p, same_p ? "true" : "false", ea_p);
        // ... and with the library's own look-ahead (mlh_scan_upload_ahead on the caller's pageable buffer: no HIP call on the caller's side)
        double period_l = 0, ea_l = 0, ma_l = 0;
        bool same_l = false;
        if (run_two_ctx(W, frames, ref_pose, &period_l, &same_l, &ea_l, &ma_l, false, true)) return 1;
        std::printf(", \"period_ms_two_contexts_mlh_scan_upload_ahead\": %.4f, \"mlh_scan_upload_ahead_same_pose\": %s, \"estimator_side_alone_mlh_scan_upload_ahead_ms\": %.4f", period_l, same_l ? "true" : "false", ea_l);
    }
    if (mode == "pipes" || mode == "all") {
        std::printf(", \"frames_per_s_at_K\": {");
        double fps1 = 0;
        std::vector<int> Ks = {1, 2, 3, 4, 6, 8};
        if (argc > 4) {                  // "1,2,4": the pipeline counts to run
            Ks.clear();
            for (const char *q = argv[4]; *q;) { Ks.push_back(std::atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
        }
        for (size_t i = 0; i < Ks.size(); ++i) {
            double fps = 0;
            bool same = false;
            unsigned long long to = 0;
            double sm[4];
            if (run_pipes(W, Ks[i], frames, ref_pose, &fps, &same, &to, sm)) return 1;
            if (i == 0) fps1 = fps;
            std::printf("%s\"%d\": {\"frames_per_s\": %.1f, \"vs_K1\": %.3f, \"streams_per_pipeline\": %d, \"same_pose\": %s, \"barriers_given_up\": %llu, "
                        "\"host_ms_per_frame\": {\"upload_extract_fuse\": %.3f, \"map_staged_beside\": %.3f, \"thin\": %.3f, \"scan2map\": %.3f}}",
                        i ? ", " : "", Ks[i], fps, fps / fps1, g_one_stream ? 1 : 2, same ? "true" : "false", to, sm[0], sm[1], sm[2], sm[3]);
        }
        std::printf("}");
        // the same pipelines with every frame sending the NEXT frame's scan ahead (mlh_scan_upload_ahead: what K replayed bags can do, a live sensor cannot)
        if (!g_lib_ahead) {
            g_lib_ahead = true;
            std::printf(", \"frames_per_s_at_K_upload_ahead\": {");
            bool first = true;
            for (size_t i = 0; i < Ks.size(); ++i) {
                if (Ks[i] > 3) continue;
                double fps = 0;
                bool same = false;
                unsigned long long to = 0;
                double sm[4];
                if (run_pipes(W, Ks[i], frames, ref_pose, &fps, &same, &to, sm)) return 1;
                std::printf("%s\"%d\": {\"frames_per_s\": %.1f, \"same_pose\": %s}", first ? "" : ", ", Ks[i], fps, same ? "true" : "false");
                first = false;
            }
            std::printf("}");
            g_lib_ahead = false;
        }
    }
    std::printf("}\n");
    (void)hipFree(W.d_surf); (void)hipFree(W.d_corner);
    return 0;
}
