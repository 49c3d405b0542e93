// The estimator's front end THROUGH THE FACADE from C++, timed as a caller sees it: NUM_OF_LASER raw 64-ring clouds in (host, pcl-shaped),
// per LiDAR calTimestamp -> ImageSegmenter::segmentCloud -> FeatureExtract::extractCloud (estimator.cpp:248-263), feature clouds out (host).
//   serial   one thread, the LiDARs one after the other on one context                      (what a caller without threads got until round 6)
//   lanes    one calling thread, FrontEndLanes::processAllLasers (the facade's own persistent workers, a context each)
//   openmp   the reference's own loop: #pragma omp parallel for num_threads(NUM_OF_LASER)    (what the reference's build gets)
//   device-resident (C-ABI: raw cloud in HBM -> mlh_segment_cloud, nothing fetched -> mlh_extract_run -> mlh_extract_voxel_run; what a pipeline that keeps its clouds
//            on the GPU pays, and the unit VERDICT r05 item 8 quotes its mark in): one thread one LiDAR after the other | the same jobs on the lanes
// and, per LiDAR, whether lanes / openmp returned the serial run's clouds bit for bit. One JSON line.
//   usage: frontbench <dir with raw_<i>.f32 (x y z 0 per point, firing order)> <n_lidars> <n_scans> [frames]
#include "mloam_facade.hpp"
#include <hip/hip_runtime_api.h>   // only for the device-resident copies of the raw clouds the last two legs start from
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include <omp.h>

using namespace mloam_hip;
using Clock = std::chrono::steady_clock;
static double ms_between(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

static std::vector<float> read_f32(const std::string &p)
{
    std::ifstream f(p, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot read %s\n", p.c_str()); std::exit(2); }
    std::vector<float> v(size_t(f.tellg()) / sizeof(float));
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), std::streamsize(v.size() * sizeof(float)));
    return v;
}

static void flatten(cloudFeature &cf, std::vector<float> &o)
{
    o.clear();
    for (const char *k : {"laser_cloud", "corner_points_sharp", "corner_points_less_sharp", "surf_points_flat", "surf_points_less_flat", "laser_cloud_outlier"}) {
        o.push_back(float(cf[k].size()));
        for (const auto &q : cf[k].points) { o.push_back(q.x); o.push_back(q.y); o.push_back(q.z); o.push_back(q.intensity); }
    }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char **argv)
{
    if (argc < 4) { std::fprintf(stderr, "usage: frontbench <dir> <n_lidars> <n_scans> [frames]\n"); return 2; }
    const std::string d = std::string(argv[1]) + "/";
    const int L = std::atoi(argv[2]), N_SCANS = std::atoi(argv[3]), frames = argc > 4 ? std::atoi(argv[4]) : 40;
    try {
        std::vector<PointCloud<PointXYZ>> v_laser_cloud_in{size_t(L)};
        size_t n_points = 0;
        for (int l = 0; l < L; ++l) {
            const auto raw = read_f32(d + "raw_" + std::to_string(l) + ".f32");
            for (size_t i = 0; i + 4 <= raw.size(); i += 4) { PointXYZ q; q.x = raw[i]; q.y = raw[i + 1]; q.z = raw[i + 2]; v_laser_cloud_in[size_t(l)].push_back(q); }
            n_points += v_laser_cloud_in[size_t(l)].size();
        }
        ImageSegmenter img_segment_;
        img_segment_.setParameter(N_SCANS, 1800, 30, 5, 3);
        FeatureExtract f_extract_;
        auto body = [&](size_t i, cloudFeature &out) {          // estimator.cpp:252-262
            PointICloud laser_cloud, laser_cloud_segment, laser_cloud_outlier;
            f_extract_.calTimestamp(v_laser_cloud_in[i], laser_cloud);
            ScanInfo scan_info(N_SCANS, true);
            img_segment_.segmentCloud(laser_cloud, laser_cloud_segment, laser_cloud_outlier, scan_info);
            out.clear();
            f_extract_.extractCloud(laser_cloud_segment, scan_info, out);
            out.insert(std::pair<std::string, PointICloud>("laser_cloud_outlier", laser_cloud_outlier));
        };
        std::vector<cloudFeature> ff_serial{size_t(L)}, ff_lanes, ff_omp{size_t(L)};
        FrontEndLanes lanes(L);
        auto run_serial = [&] { for (size_t i = 0; i < size_t(L); ++i) body(i, ff_serial[i]); };
        auto run_lanes = [&] { lanes.processAllLasers(img_segment_, f_extract_, v_laser_cloud_in, N_SCANS, true, ff_lanes); };
        auto run_omp = [&] {
#pragma omp parallel for num_threads(L)
            for (size_t i = 0; i < v_laser_cloud_in.size(); i++) body(i, ff_omp[i]);
        };
        for (int w = 0; w < 3; ++w) { run_serial(); run_lanes(); run_omp(); }          // contexts created, buffers sized, threads up
        int eq_lanes = 0, eq_omp = 0;
        for (size_t i = 0; i < size_t(L); ++i) {
            std::vector<float> a, b, c;
            flatten(ff_serial[i], a); flatten(ff_lanes[i], b); flatten(ff_omp[i], c);
            eq_lanes += a == b ? 1 : 0; eq_omp += a == c ? 1 : 0;
        }
        // three alternations of `frames` frames each; per form the median of the three
        std::vector<double> t_serial, t_lanes, t_omp;
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = Clock::now();
            for (int k = 0; k < frames; ++k) run_serial();
            auto t1 = Clock::now();
            for (int k = 0; k < frames; ++k) run_lanes();
            auto t2 = Clock::now();
            for (int k = 0; k < frames; ++k) run_omp();
            auto t3 = Clock::now();
            t_serial.push_back(ms_between(t0, t1) / frames); t_lanes.push_back(ms_between(t1, t2) / frames); t_omp.push_back(ms_between(t2, t3) / frames);
        }
        // the front end WITHOUT a host hop through the facade: pcl-shaped raw clouds in, per lane calTimestamp -> segmentCloudOnDevice -> extractStagedCloudOnDevice, gathered
        // on the calling thread's Device (fuseCloudFeatureFrom), both fused clouds thinned there (downsampleFusedScans: the first call that waits) -- against the same
        // calls on ONE Device, the LiDARs one after the other
        std::vector<double> t_dev_facade_serial, t_dev_facade_lanes;
        int thinned_equal = 0;
        {
            std::vector<Pose> pose_ext{size_t(L)};
            for (int i = 0; i < L; ++i) pose_ext[size_t(i)].t_(1) = 0.3 * i;
            Device &g = threadDevice();
            ImageSegmenter seg_b(g);
            seg_b.setParameter(N_SCANS, 1800, 30, 5, 3);
            FeatureExtract fe_b(g);
            auto serial_dev = [&]() -> std::pair<int, int> {
                fuseReset(g);
                for (int i = 0; i < L; ++i) {
                    PointICloud laser_cloud;
                    fe_b.calTimestamp(v_laser_cloud_in[size_t(i)], laser_cloud);
                    ScanInfo scan_info(N_SCANS, true);
                    seg_b.segmentCloudOnDevice(laser_cloud, scan_info);
                    fe_b.extractStagedCloudOnDevice();
                    fuseCloudFeature(g, i, pose_ext[size_t(i)]);
                }
                return downsampleFusedScans(g, 0.4f, 0.2f, pose_ext, true);
            };
            auto lanes_dev = [&]() -> std::pair<int, int> {
                lanes.processAllLasersOnDevice(img_segment_, f_extract_, v_laser_cloud_in, N_SCANS, true, g, pose_ext);
                return downsampleFusedScans(g, 0.4f, 0.2f, pose_ext, true);
            };
            std::pair<int, int> a, b;
            for (int w = 0; w < 3; ++w) { a = serial_dev(); b = lanes_dev(); }
            thinned_equal = a == b ? 1 : 0;
            for (int rep = 0; rep < 3; ++rep) {
                auto t0 = Clock::now();
                for (int k = 0; k < frames; ++k) serial_dev();
                auto t1 = Clock::now();
                for (int k = 0; k < frames; ++k) lanes_dev();
                auto t2 = Clock::now();
                t_dev_facade_serial.push_back(ms_between(t0, t1) / frames); t_dev_facade_lanes.push_back(ms_between(t1, t2) / frames);
            }
        }
        // the calls alone, device-resident results (what a device-resident pipeline pays: no feature clouds to the host): segmentCloud's share of the serial form
        std::vector<double> t_seg;
        {
            ImageSegmenter seg_b(threadDevice());
            seg_b.setParameter(N_SCANS, 1800, 30, 5, 3);
            PointICloud laser_cloud, a, b;
            f_extract_.calTimestamp(v_laser_cloud_in[0], laser_cloud);
            for (int rep = 0; rep < 3; ++rep) {
                auto t0 = Clock::now();
                for (int k = 0; k < frames; ++k) { ScanInfo si(N_SCANS, true); seg_b.segmentCloud(laser_cloud, a, b, si); }
                t_seg.push_back(ms_between(t0, Clock::now()) / frames);
            }
        }
        // device-resident legs
        std::vector<double> t_dev_serial, t_dev_lanes;
        {
            std::vector<void *> d_raw(size_t(L), nullptr);
            for (int l = 0; l < L; ++l) {
                const auto &c = v_laser_cloud_in[size_t(l)];
                if (hipMalloc(&d_raw[size_t(l)], c.size() * sizeof(PointXYZ)) != hipSuccess ||
                    hipMemcpy(d_raw[size_t(l)], c.points.data(), c.size() * sizeof(PointXYZ), hipMemcpyHostToDevice) != hipSuccess) throw Error("hipMalloc / hipMemcpy of a raw cloud");
            }
            mlh_segment_params prm;
            mlh_segment_params_default(&prm);
            prm.vertical_scans = N_SCANS; prm.horizon_scans = 1800; prm.segment_flag = 1;
            auto dev_job = [&](size_t i) {
                Device &dev = threadDevice();
                int32_t n_out = 0, n_outl = 0;
                dev.check(mlh_segment_cloud(dev.ctx(), d_raw[i], int(sizeof(PointXYZ)), -1, int(v_laser_cloud_in[i].size()), MLH_MEM_DEVICE, &prm, nullptr, &n_out, nullptr, nullptr,
                                            nullptr, 0, &n_outl));
                dev.check(mlh_extract_run(dev.ctx()));
                dev.check(mlh_extract_voxel_run(dev.ctx(), 0.2f));
                dev.check(mlh_synchronize(dev.ctx()));
            };
            auto dev_serial = [&] { for (size_t i = 0; i < size_t(L); ++i) dev_job(i); };
            auto dev_lanes = [&] { for (int i = 0; i < L; ++i) lanes.post(i, [&dev_job, i] { dev_job(size_t(i)); }); for (int i = 0; i < L; ++i) lanes.wait(i); };
            for (int w = 0; w < 3; ++w) { dev_serial(); dev_lanes(); }
            for (int rep = 0; rep < 3; ++rep) {
                auto t0 = Clock::now();
                for (int k = 0; k < frames; ++k) dev_serial();
                auto t1 = Clock::now();
                for (int k = 0; k < frames; ++k) dev_lanes();
                auto t2 = Clock::now();
                t_dev_serial.push_back(ms_between(t0, t1) / frames); t_dev_lanes.push_back(ms_between(t1, t2) / frames);
            }
            for (void *q : d_raw) (void)hipFree(q);
        }
        std::printf("{\"device_resident_ms_per_frame_serial\": %.4f, \"device_resident_ms_per_frame_lanes\": %.4f, ", median(t_dev_serial), median(t_dev_lanes));
        std::printf("\"facade_no_host_hop_ms_per_frame_one_device\": %.4f, \"facade_no_host_hop_ms_per_frame_lanes\": %.4f, \"facade_no_host_hop_thinned_counts_equal\": %d, ",
                    median(t_dev_facade_serial), median(t_dev_facade_lanes), thinned_equal);
        std::printf("\"n_lidars\": %d, \"n_scans\": %d, \"points\": %zu, \"frames\": %d, \"ms_per_frame_serial_one_thread\": %.4f, \"ms_per_frame_lanes_one_calling_thread\": %.4f, "
                    "\"ms_per_frame_openmp\": %.4f, \"ms_segment_cloud_alone_one_lidar\": %.4f, \"lidars_equal_lanes\": %d, \"lidars_equal_openmp\": %d, "
                    "\"serial_runs\": [%.4f, %.4f, %.4f], \"lanes_runs\": [%.4f, %.4f, %.4f], \"openmp_runs\": [%.4f, %.4f, %.4f]}\n",
                    L, N_SCANS, n_points, frames, median(t_serial), median(t_lanes), median(t_omp), median(t_seg), eq_lanes, eq_omp,
                    t_serial[0], t_serial[1], t_serial[2], t_lanes[0], t_lanes[1], t_lanes[2], t_omp[0], t_omp[1], t_omp[2]);
        return (eq_lanes == L && eq_omp == L) ? 0 : 1;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "frontbench: %s\n", e.what());
        return 1;
    }
}
