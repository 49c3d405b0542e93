// One device-resident mapper frame timed THROUGH THE C-ABI FROM C++ (scripts/framebench.py times the same sequence through ctypes: a dozen calls per frame, each
// with its interpreter overhead): two raw 64-ring scans in (as one joint upload), extractCloud + per-ring voxel grid, fusion, downsampleCurrentScan for both
// kinds, index rebuild, scan2MapOptimization, pose out. Inputs are the files scripts/framebench.py writes (the bench workload); prints ms per frame by stage.
//   usage: framebench <dir> [frames]
#include "../../include/mloam_hip.h"
#include <hip/hip_runtime_api.h>   // only for the device-resident copy of the local map the second loop stages from
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
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

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <dir> [frames]\n", argv[0]); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    const int frames = argc > 2 ? std::atoi(argv[2]) : 50;
    const auto pts = read_file<float>(d + "fb_points.f32");           // both scans, rings back to back: x y z intensity
    const auto rings = read_file<int32_t>(d + "fb_rings.i32");        // [start (R)] [end (R)]
    const auto ring_ofs = read_file<int32_t>(d + "fb_ring_ofs.i32");  // ring range of every LiDAR (n_lidar + 1)
    const auto ext = read_file<double>(d + "fb_ext.f64");             // n_lidar x 7
    const auto covs = read_file<double>(d + "fb_covs.f64");           // n_lidar x 36
    const auto meas = read_file<double>(d + "fb_meas.f64");           // 9
    const auto surf_map = read_file<float>(d + "fb_surf_map.f32"), corner_map = read_file<float>(d + "fb_corner_map.f32");   // x y z (+ fields): stride in fb_meta
    const auto meta = read_file<int32_t>(d + "fb_meta.i32");          // [map stride bytes, with_ua]
    const auto p0 = read_file<double>(d + "fb_pose.f64");
    const int R = int(rings.size() / 2), n = int(pts.size() / 4), n_lidar = int(ring_ofs.size()) - 1, map_stride = meta[0];
    mlh_ctx *ctx = nullptr;
    if (mlh_create(&ctx, 0)) { std::fprintf(stderr, "no GPU context\n"); return 1; }
    CK(mlh_map_set_pair(ctx, surf_map.data(), int(surf_map.size() * 4 / size_t(map_stride)), corner_map.data(), int(corner_map.size() * 4 / size_t(map_stride)), map_stride, 1.0f, MLH_MEM_HOST));
    mlh_solver_opts o;
    mlh_solver_opts_default(&o);
    if (meta[1]) o.flags |= MLH_FLAG_WITH_UA;
    double pose[7], t_stage[3] = {0, 0, 0};
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    // mode 0: the index of the local map rebuilt between the thinning and the solve (on the frame's critical path);
    // mode 1: the local map (device-resident, as a mapper that assembles it from keyframe clouds on the GPU has it) staged and indexed beside the front end:
    //         mlh_map_set_pair_overlapped after the front end's launches are enqueued, before the first call that waits for them
    void *d_surf = nullptr, *d_corner = nullptr;
    if (hipMalloc(&d_surf, surf_map.size() * 4) != hipSuccess || hipMalloc(&d_corner, corner_map.size() * 4) != hipSuccess ||
        hipMemcpy(d_surf, surf_map.data(), surf_map.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d_corner, corner_map.data(), corner_map.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { std::fprintf(stderr, "device copy of the maps failed\n"); return 1; }
    const int n_surf_map = int(surf_map.size() * 4 / size_t(map_stride)), n_corner_map = int(corner_map.size() * 4 / size_t(map_stride));
    const int stage_pos = std::getenv("FB_STAGE_POS") ? std::atoi(std::getenv("FB_STAGE_POS")) : 3;   // where in the front end the staging call sits (3: after the last launch)
    for (int mode = 0; mode < 2; ++mode) {
        t_stage[0] = t_stage[1] = t_stage[2] = 0.0;
        for (int it = -5; it < frames; ++it) {
            const auto t0 = now();
            CK(mlh_fuse_reset(ctx));
            if (mode == 1 && stage_pos == 0) CK(mlh_map_set_pair_overlapped(ctx, d_surf, n_surf_map, d_corner, n_corner_map, map_stride, 1.0f, MLH_MEM_DEVICE));
            CK(mlh_scan_upload(ctx, pts.data(), 16, 12, n, rings.data(), rings.data() + R, R, MLH_MEM_HOST));
            if (mode == 1 && stage_pos == 1) CK(mlh_map_set_pair_overlapped(ctx, d_surf, n_surf_map, d_corner, n_corner_map, map_stride, 1.0f, MLH_MEM_DEVICE));
            CK(mlh_extract_run(ctx));
            if (mode == 1 && stage_pos == 2) CK(mlh_map_set_pair_overlapped(ctx, d_surf, n_surf_map, d_corner, n_corner_map, map_stride, 1.0f, MLH_MEM_DEVICE));
            CK(mlh_extract_voxel_run(ctx, 0.2f));
            for (int i = 0; i < n_lidar; ++i) CK(mlh_fuse_add_rings(ctx, ring_ofs[size_t(i)], ring_ofs[size_t(i) + 1], i, ext.data() + 7 * i));
            if (mode == 1 && stage_pos == 3) CK(mlh_map_set_pair_overlapped(ctx, d_surf, n_surf_map, d_corner, n_corner_map, map_stride, 1.0f, MLH_MEM_DEVICE));
            const auto t1 = now();
            const void *fs = nullptr, *fc = nullptr;
            int32_t ns = 0, nc = 0, ms_ = 0, mc = 0;
            CK(mlh_fused_cloud(ctx, MLH_SURF, &fs, &ns));
            CK(mlh_fused_cloud(ctx, MLH_CORNER, &fc, &nc));
            CK(mlh_downsample_current_scan_pair(ctx, fs, ns, fc, nc, 16, 12, MLH_MEM_DEVICE, 0.4f, 0.2f, ext.data(), covs.data(), n_lidar, meas.data(), meta[1], 0.6, &ms_, &mc));
            const auto t2 = now();
            if (mode == 0) CK(mlh_map_rebuild(ctx, MLH_ALL_KINDS));
            for (int i = 0; i < 7; ++i) pose[i] = p0[size_t(i)];
            CK(mlh_scan2map(ctx, pose, &o, nullptr));
            const auto t3 = now();
            if (it >= 0) { t_stage[0] += ms(t0, t1); t_stage[1] += ms(t1, t2); t_stage[2] += ms(t2, t3); }
        }
        std::printf("%s, ms per frame: upload+extract+fuse%s %.3f downsample %.3f scan2map %.3f total %.3f  pose %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n",
                    mode == 0 ? "C++ over the C-ABI, device-resident, one launch set, both kinds thinned in one pipeline"
                              : "  the same with the local map staged and indexed beside the front end (second stream, other map set)",
                    mode == 0 ? "" : "+map staging", t_stage[0] / frames, t_stage[1] / frames, t_stage[2] / frames, (t_stage[0] + t_stage[1] + t_stage[2]) / frames,
                    pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], pose[6]);
    }
    (void)hipFree(d_surf); (void)hipFree(d_corner);
    mlh_destroy(ctx);
    return 0;
}
