// TEST INFRASTRUCTURE ONLY -- the reference's own call sites of the hot path, compiled VERBATIM against the drop-in boundary.
//
// tests/host/refcut/build_refcut.py cuts the line ranges named below out of /root/reference into _gen/*.inc (never committed, deleted after the compile); this
// file supplies what surrounds those lines in the reference -- the declarations of the variables they name -- and nothing else. Everything the cut lines CALL is
// m-loam_amd/host/mloam_facade.hpp (-DMLOAM_FACADE_USE_PCL_TYPES -DMLOAM_FACADE_CERES_BASES) over headers shaped like PCL / boost / Eigen / Ceres
// (tests/host/pcl_stub); every TYPE they name is the reference's own, cut as well: pcl::PointXYZIWithCov, common::PointICloud, cloudFeature, PointPlaneFeature,
// ScanInfo, Pose. What a maintainer changes in the reference is what is written out here by hand: the include of the façade, five using-declarations, and the
// declared type of the two kd-tree pointers.
//
//   refcut_selftest cpu <dir>    host only: <dir>/factors.f64 (n x 26 rows: kind, point, coeff[6], cov[9], pose[7]) -> the AddResidualBlock loops of
//                                lidar_mapper_keyframe.cpp:537-571 build one residual block per row in a ceres::Problem that OWNS them; every block evaluated
//                                through the interface Ceres calls -> <dir>/factors_out.f64 (n x 8: r, J[7]), with and without with_ua_flag
//   refcut_selftest gpu <dir>    on the GPU: the estimator's OpenMP front end (estimator.cpp:248-270) on four scans; kdtree_*_from_map->setInputCloud
//                                (cpp:433-434); the façade's match functions filling the REFERENCE's PointPlaneFeature at the REFERENCE's Pose; the
//                                AddResidualBlock loops on those features; every block's r / J against the batched device evaluation
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#include <Eigen/Dense>
#include <ceres/ceres.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <new>
#include <string>
#include <vector>
#include <omp.h>

// ---- live heap allocations (the ownership check: what ceres::Problem took must be gone when it is)
static std::atomic<long> g_live{0};
void *operator new(std::size_t n) { void *p = std::malloc(n ? n : 1); if (!p) throw std::bad_alloc(); ++g_live; return p; }
void operator delete(void *p) noexcept { if (p) { --g_live; std::free(p); } }
void operator delete(void *p, std::size_t) noexcept { if (p) { --g_live; std::free(p); } }

// ---------------------------------------------------------------- the reference's own types
#include "_gen/point_with_cov.inc"                    // namespace pcl { struct _PointXYZIWithCov; struct PointXYZIWithCov }      mloam_pcl/point_with_cov.hpp:43-112
namespace common {
#include "_gen/extract_cov.inc"                       // void extractCov(const pcl::PointXYZIWithCov &, Eigen::Matrix3d &)       point_with_cov.hpp:202-214
}
#include "_gen/common_types.inc"                      // namespace common { typedef pcl::PointXYZ Point; ... PointICloudConstPtr;   common/types/type.h:13-23
}                                                     // (the cut stops before the typedefs of point types this path does not use)
using namespace std;                                  // pose.h:35, parameters.h
using namespace common;
#include "_gen/parameter_types.inc"                   // cloudFeature, PointPlaneFeature, FeatureWithScore, ScanInfo                 parameters.h:161-207
namespace nav_msgs { struct Odometry; }
namespace geometry_msgs { struct Pose; }
#include "_gen/pose_class.inc"                        // class Pose                                                                  pose.h:38-66
#include "_gen/pose_ctor_default.inc"                 // pose.cpp:16-23
#include "_gen/pose_ctor_copy.inc"                    // pose.cpp:25-32
#include "_gen/pose_ctor_qt.inc"                      // pose.cpp:34-41
#include "_gen/pose_inverse_update.inc"               // pose.cpp:99-108
#include "_gen/pose_mul.inc"                          // pose.cpp:110-113
typedef pcl::PointXYZIWithCov PointIWithCov;          // lidar_mapper.h
typedef pcl::PointCloud<PointIWithCov> PointICovCloud;

// ---------------------------------------------------------------- what the maintainer adds
#include "mloam_facade.hpp"
using mloam_hip::FeatureExtract;
using mloam_hip::ImageSegmenter;
using mloam_hip::LidarMapPlaneNormFactor;
using mloam_hip::LidarMapEdgeFactor;
using mloam_hip::PoseLocalParameterization;

// ---------------------------------------------------------------- file I/O of the harness
template <typename T>
static std::vector<T> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    size_t bytes = (size_t)f.tellg();
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), bytes);
    return v;
}
template <typename T>
static void write_file(const std::string &path, const std::vector<T> &v)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), sizeof(T) * v.size());
}

// ---------------------------------------------------------------- the mapper's file-scope state the cut lines name (lidar_mapper_keyframe.cpp:59-131)
static mloam_hip::MapIndex<PointIWithCov>::Ptr kdtree_surf_from_map, kdtree_corner_from_map;      // reference: pcl::KdTreeFLANN<PointIWithCov>::Ptr (cpp:59-62)
static PointICovCloud::Ptr laser_cloud_surf_from_map_cov_ds(new PointICovCloud()), laser_cloud_corner_from_map_cov_ds(new PointICovCloud());
static PointICovCloud::Ptr laser_cloud_surf_cov(new PointICovCloud()), laser_cloud_corner_cov(new PointICovCloud());
static std::vector<PointPlaneFeature> all_surf_features, all_corner_features;
static std::vector<size_t> sel_surf_feature_idx, sel_corner_feature_idx;
static bool with_ua_flag = true;
static Eigen::Matrix3d COV_MEASUREMENT;                                                            // parameters.cpp
static int CHECK_JACOBIAN = 0;
static double para_pose[7];

// lidar_mapper_keyframe.cpp:433-434 inside a function
static void build_map_indices()
{
#include "_gen/set_input_cloud.inc"
}

// lidar_mapper_keyframe.cpp:537-571 with the four declarations that precede it in scan2MapOptimization (cpp:441-442, 446-452)
static void add_residual_blocks(ceres::Problem &problem, ceres::LossFunction *loss_function, std::vector<ceres::internal::ResidualBlock *> &res_ids_proj)
{
#include "_gen/add_residual_blocks.inc"
}

struct BlockEval { double r; double J[7]; };
static BlockEval evaluate_block(const ceres::Problem &problem, ceres::internal::ResidualBlock *id)
{
    BlockEval e{};
    std::vector<std::vector<double>> J;
    problem.EvaluateBlock(id, false, &e.r, &J);
    for (int k = 0; k < 7 && k < (int)J[0].size(); ++k) e.J[k] = J[0][k];
    return e;
}

// ---------------------------------------------------------------- cpu mode
static int run_cpu(const std::string &d)
{
    auto rows = read_file<double>(d + "factors.f64");
    const size_t n = rows.size() / 26;
    std::vector<double> out;
    int owned_ok = 1;
    for (int ua = 0; ua < 2; ++ua) {
        with_ua_flag = ua == 1;
        all_surf_features.clear(); all_corner_features.clear(); sel_surf_feature_idx.clear(); sel_corner_feature_idx.clear();
        laser_cloud_surf_cov->clear(); laser_cloud_corner_cov->clear();
        std::vector<size_t> row_of_surf, row_of_corner;
        for (size_t i = 0; i < n; ++i) {
            const double *w = rows.data() + i * 26;
            PointPlaneFeature f;
            f.point_ = Eigen::Vector3d(w[1], w[2], w[3]);
            Eigen::Matrix3f cov;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) cov(r, c) = float(w[10 + r * 3 + c]);
            pcl::PointXYZI pi;
            pi.x = float(w[1]); pi.y = float(w[2]); pi.z = float(w[3]);
            if (w[0] == 0.0) {
                f.type_ = 's';
                f.coeffs_.resize(4);
                for (int k = 0; k < 4; ++k) f.coeffs_[k] = w[4 + k];
                f.idx_ = laser_cloud_surf_cov->size();
                laser_cloud_surf_cov->push_back(PointIWithCov(pi, cov));                       // point_with_cov.hpp:90-100: the reference's own constructor
                sel_surf_feature_idx.push_back(all_surf_features.size());
                all_surf_features.push_back(f);
                row_of_surf.push_back(i);
            } else {
                f.type_ = 'c';
                f.coeffs_.resize(6);
                for (int k = 0; k < 6; ++k) f.coeffs_[k] = w[4 + k];
                f.idx_ = laser_cloud_corner_cov->size();
                laser_cloud_corner_cov->push_back(PointIWithCov(pi, cov));
                sel_corner_feature_idx.push_back(all_corner_features.size());
                all_corner_features.push_back(f);
                row_of_corner.push_back(i);
            }
        }
        std::vector<double> res(n * 8, 0.0);
        const long live_before = g_live.load();
        {
            ceres::Problem problem;                                                            // cpp:441
            ceres::LossFunction *loss_function = new ceres::HuberLoss(0.1);                    // cpp:442
            PoseLocalParameterization *local_parameterization = new PoseLocalParameterization();      // cpp:446-450
            local_parameterization->setParameter();
            problem.AddParameterBlock(para_pose, 7, local_parameterization);
            std::vector<ceres::internal::ResidualBlock *> res_ids_proj;                        // cpp:451
            add_residual_blocks(problem, loss_function, res_ids_proj);
            if (res_ids_proj.size() != n || problem.NumOwnedCostFunctions() != n || problem.NumOwnedLossFunctions() != 1) owned_ok = 0;
            // the blocks come in the loops' order: all surf features, then all corner features; every row has its own pose
            for (size_t b = 0; b < res_ids_proj.size(); ++b) {
                const size_t i = b < row_of_surf.size() ? row_of_surf[b] : row_of_corner[b - row_of_surf.size()];
                for (int k = 0; k < 7; ++k) para_pose[k] = rows[i * 26 + 19 + k];
                const BlockEval e = evaluate_block(problem, res_ids_proj[b]);
                res[i * 8] = e.r;
                for (int k = 0; k < 7; ++k) res[i * 8 + 1 + k] = e.J[k];
            }
        }
        if (g_live.load() != live_before) { std::fprintf(stderr, "ownership: %ld allocations outlive the ceres::Problem\n", g_live.load() - live_before); owned_ok = 0; }
        out.insert(out.end(), res.begin(), res.end());
    }
    write_file(d + "factors_out.f64", out);
    std::printf("refcut cpu: %zu residual blocks through lidar_mapper_keyframe.cpp:537-571, owned and released by ceres::Problem: %s\n", n, owned_ok ? "yes" : "NO");
    return owned_ok ? 0 : 1;
}

// ---------------------------------------------------------------- gpu mode
// the Estimator members / parameters.cpp globals estimator.cpp:248-270 names
static int NUM_OF_LASER = 4, N_SCANS = 16, SEGMENT_CLOUD = 1, ESTIMATE_EXTRINSIC = 0;
static FeatureExtract f_extract_;                     // estimator.h:190 -- the façade's, default-constructed as the member is
static ImageSegmenter img_segment_;                   // estimator.h:189
static size_t total_corner_feature_ = 0, total_surf_feature_ = 0;

static void estimator_front_end(const std::vector<PointCloud> &v_laser_cloud_in, std::vector<cloudFeature> &feature_frame, std::vector<int> &thread_of)
{
    // thread_of is filled by a second, identical-shaped parallel region: the cut lines themselves are not touched
#include "_gen/estimator_front_end.inc"
#pragma omp parallel for num_threads(NUM_OF_LASER)
    for (size_t i = 0; i < v_laser_cloud_in.size(); i++) thread_of[i] = omp_get_thread_num();
    for (cloudFeature *p : feature_frame_ptr) delete p;                                         // (the reference leaks them)
}

static void flatten(cloudFeature &cf, std::vector<float> &o)
{
    o.clear();
    for (const char *k : {"laser_cloud", "corner_points_sharp", "corner_points_less_sharp", "surf_points_flat", "surf_points_less_flat", "laser_cloud_outlier"}) {
        o.push_back(float(cf[k].size()));
        for (const auto &q : cf[k].points) { o.push_back(q.x); o.push_back(q.y); o.push_back(q.z); o.push_back(q.intensity); }
    }
}

static void fill_cov_cloud(PointICovCloud &c, const std::vector<float> &a, int cols)
{
    c.clear();
    for (size_t i = 0; i + cols <= a.size(); i += cols) {
        PointIWithCov p;
        p.x = a[i]; p.y = a[i + 1]; p.z = a[i + 2];
        p.intensity = cols > 3 ? a[i + 3] : 0.f;
        c.push_back(p);
    }
}

static int run_gpu(const std::string &d)
{
    mloam_hip::Device dev(0);
    std::vector<int> verdict;
    // ---- (1) estimator.cpp:248-270 on four scans, against the same calls one after the other on objects bound to one context
    {
        auto raw = read_file<float>(d + "raw_cloud.f32");
        std::vector<PointCloud> v_laser_cloud_in(NUM_OF_LASER);
        for (int l = 0; l < NUM_OF_LASER; ++l) {             // the harness's cloud turned about z by 0 / 17 / 34 / 51 degrees
            const double a = l * 17.0 * M_PI / 180.0, ca = std::cos(a), sa = std::sin(a);
            for (size_t i = 0; i + 4 <= raw.size(); i += 4) {
                pcl::PointXYZ q;
                q.x = float(ca * raw[i] - sa * raw[i + 1]); q.y = float(sa * raw[i] + ca * raw[i + 1]); q.z = raw[i + 2];
                v_laser_cloud_in[l].push_back(q);
            }
        }
        img_segment_.setParameter(N_SCANS, 1800, 30, 5, 3);
        std::vector<std::vector<float>> seq(NUM_OF_LASER), par(NUM_OF_LASER);
        {
            ImageSegmenter seg_b(dev);
            seg_b.setParameter(N_SCANS, 1800, 30, 5, 3);
            FeatureExtract fe_b(dev);
            for (int i = 0; i < NUM_OF_LASER; ++i) {
                PointICloud laser_cloud, laser_cloud_segment, laser_cloud_outlier;
                fe_b.calTimestamp(v_laser_cloud_in[i], laser_cloud);
                ScanInfo scan_info(N_SCANS, SEGMENT_CLOUD);
                seg_b.segmentCloud(laser_cloud, laser_cloud_segment, laser_cloud_outlier, scan_info);
                cloudFeature cfb;
                fe_b.extractCloud(laser_cloud_segment, scan_info, cfb);
                cfb.insert(pair<std::string, PointICloud>("laser_cloud_outlier", laser_cloud_outlier));
                flatten(cfb, seq[i]);
            }
        }
        std::vector<int> thread_of(NUM_OF_LASER, -1);
        size_t n_corner = 0, n_surf = 0;
        for (int round = 0; round < 3; ++round) {               // three frames: the per-thread contexts are created once and reused
            std::vector<cloudFeature> feature_frame(NUM_OF_LASER);
            total_corner_feature_ = total_surf_feature_ = 0;
            estimator_front_end(v_laser_cloud_in, feature_frame, thread_of);
            for (int i = 0; i < NUM_OF_LASER; ++i) flatten(feature_frame[i], par[i]);
            n_corner = total_corner_feature_; n_surf = total_surf_feature_;
        }
        int distinct = 0;
        { std::vector<int> seen(64, 0); for (int t : thread_of) if (t >= 0 && t < 64 && !seen[t]) { seen[t] = 1; ++distinct; } }
        for (int i = 0; i < NUM_OF_LASER; ++i) { verdict.push_back(par[i] == seq[i] ? 1 : 0); verdict.push_back(int(seq[i].size())); }
        verdict.push_back(distinct);
        verdict.push_back(int(n_corner)); verdict.push_back(int(n_surf));
        std::printf("refcut gpu: estimator.cpp:248-270 -- %d LiDARs on %d threads, clouds equal to the sequential calls: %d %d %d %d; corner %zu surf %zu\n", NUM_OF_LASER, distinct,
                    verdict[0], verdict[2], verdict[4], verdict[6], n_corner, n_surf);
    }
    // ---- (2) lidar_mapper_keyframe.cpp:433-434, then the façade's match functions on the reference's own Pose / PointPlaneFeature
    fill_cov_cloud(*laser_cloud_surf_from_map_cov_ds, read_file<float>(d + "surf_map.f32"), 3);
    fill_cov_cloud(*laser_cloud_corner_from_map_cov_ds, read_file<float>(d + "corner_map.f32"), 3);
    fill_cov_cloud(*laser_cloud_surf_cov, read_file<float>(d + "surf.f32"), 4);
    fill_cov_cloud(*laser_cloud_corner_cov, read_file<float>(d + "corner.f32"), 4);
    for (auto *c : {laser_cloud_surf_cov.get(), laser_cloud_corner_cov.get()}) {
        size_t k = 0;
        for (auto &p : c->points) {          // COV_MEASUREMENT for a third of the features (weight 1), two heavier classes whose weight is below 1 (lidar_map_factor.hpp:36-40)
            const float v = 0.0025f + 0.05f * float(k++ % 3);
            p.cov_vec[0] = p.cov_vec[3] = p.cov_vec[5] = v; p.cov_trace = v + v + v;
        }
    }
    kdtree_surf_from_map.reset(new mloam_hip::MapIndex<PointIWithCov>(dev, MLH_SURF));
    kdtree_corner_from_map.reset(new mloam_hip::MapIndex<PointIWithCov>(dev, MLH_CORNER));
    build_map_indices();
    auto pv = read_file<double>(d + "pose.f64");
    Pose pose_wmap_curr(Eigen::Quaterniond(pv[6], pv[3], pv[4], pv[5]), Eigen::Vector3d(pv[0], pv[1], pv[2]));      // the reference's Pose(q, t): pose.cpp:34-41
    FeatureExtract f_extract(dev);
    // ActiveFeatureSelection::goodFeatureMatching's `wo_gf` branch (lidar_mapper.h:250-262): match every feature, keep the matched ones
    f_extract.matchSurfFromMap(kdtree_surf_from_map, *laser_cloud_surf_from_map_cov_ds, *laser_cloud_surf_cov, pose_wmap_curr, all_surf_features, 5, false);
    f_extract.matchCornerFromMap(kdtree_corner_from_map, *laser_cloud_corner_from_map_cov_ds, *laser_cloud_corner_cov, pose_wmap_curr, all_corner_features, 5, false);
    sel_surf_feature_idx.clear(); sel_corner_feature_idx.clear();
    for (size_t i = 0; i < all_surf_features.size(); ++i) sel_surf_feature_idx.push_back(i);
    for (size_t i = 0; i < all_corner_features.size(); ++i) sel_corner_feature_idx.push_back(i);
    {
        std::vector<uint8_t> vs(laser_cloud_surf_cov->size(), 0), vc(laser_cloud_corner_cov->size(), 0);
        for (const PointPlaneFeature &f : all_surf_features) vs[f.idx_] = 1;
        for (const PointPlaneFeature &f : all_corner_features) vc[f.idx_] = 1;
        write_file(d + "refcut_valid_surf.u8", vs);
        write_file(d + "refcut_valid_corner.u8", vc);
    }
    // ---- (3) lidar_mapper_keyframe.cpp:537-571 on those features; every block against the batched device evaluation of the same correspondences
    double max_dr = 0.0, max_dJ = 0.0;
    size_t n_blocks = 0;
    int owned_ok = 1;
    for (int ua = 0; ua < 2; ++ua) {
        with_ua_flag = ua == 1;
        para_pose[0] = pv[0]; para_pose[1] = pv[1]; para_pose[2] = pv[2]; para_pose[3] = pv[3]; para_pose[4] = pv[4]; para_pose[5] = pv[5]; para_pose[6] = pv[6];
        // the batched device evaluation first (its buffers and the runtime's own bookkeeping stay out of the ownership count below): the correspondences of the two
        // match calls above are still staged, kind by kind; rows of unmatched features are zero
        std::vector<double> dev_r[2], dev_J[2];
        for (int kind = 0; kind < 2; ++kind) {
            const PointICovCloud &feat = kind == 0 ? *laser_cloud_surf_cov : *laser_cloud_corner_cov;
            mloam_hip::LidarMapBatchFactor batch(dev, kind == 0 ? MLH_SURF : MLH_CORNER, (int)feat.size(), with_ua_flag);
            dev_r[kind].assign(feat.size(), 0.0); dev_J[kind].assign(feat.size() * 7, 0.0);
            const double *pp[1] = {para_pose};
            double *Jp[1] = {dev_J[kind].data()};
            if (!batch.Evaluate(pp, dev_r[kind].data(), Jp)) throw std::runtime_error("LidarMapBatchFactor::Evaluate failed");
        }
        const long live_before = g_live.load();
        {
            ceres::Problem problem;
            ceres::LossFunction *loss_function = new ceres::HuberLoss(0.1);
            PoseLocalParameterization *local_parameterization = new PoseLocalParameterization();
            local_parameterization->setParameter();
            problem.AddParameterBlock(para_pose, 7, local_parameterization);
            std::vector<ceres::internal::ResidualBlock *> res_ids_proj;
            add_residual_blocks(problem, loss_function, res_ids_proj);
            n_blocks = res_ids_proj.size();
            if (n_blocks != all_surf_features.size() + all_corner_features.size() || problem.NumOwnedCostFunctions() != n_blocks) owned_ok = 0;
            for (int kind = 0; kind < 2; ++kind) {
                const std::vector<PointPlaneFeature> &fs = kind == 0 ? all_surf_features : all_corner_features;
                for (size_t b = 0; b < fs.size(); ++b) {
                    const BlockEval e = evaluate_block(problem, res_ids_proj[(kind == 0 ? 0 : all_surf_features.size()) + b]);
                    const size_t i = fs[b].idx_;
                    max_dr = std::max(max_dr, std::fabs(e.r - dev_r[kind][i]));
                    for (int k = 0; k < 7; ++k) max_dJ = std::max(max_dJ, std::fabs(e.J[k] - dev_J[kind][i * 7 + k]));
                }
            }
        }
        if (g_live.load() != live_before) owned_ok = 0;
    }
    verdict.push_back(int(n_blocks));
    verdict.push_back(owned_ok);
    write_file(d + "refcut_verdict.i32", verdict);
    write_file(d + "refcut_diff.f64", std::vector<double>{max_dr, max_dJ});
    std::printf("refcut gpu: cpp:433-434 + match on the reference's Pose / PointPlaneFeature: %zu surf + %zu corner matched; cpp:537-571: %zu blocks owned by ceres::Problem (%s), "
                "per-block Evaluate vs the batched device evaluation: |dr| <= %.3e, |dJ| <= %.3e\n", all_surf_features.size(), all_corner_features.size(), n_blocks,
                owned_ok ? "released" : "LEAKED", max_dr, max_dJ);
    kdtree_surf_from_map.reset(); kdtree_corner_from_map.reset();
    return owned_ok ? 0 : 1;
}

int main(int argc, char **argv)
{
    std::setvbuf(stdout, nullptr, _IONBF, 0);
    if (argc < 3) { std::fprintf(stderr, "usage: %s cpu|gpu <dir>\n", argv[0]); return 2; }
    COV_MEASUREMENT.setZero();
    COV_MEASUREMENT(0, 0) = COV_MEASUREMENT(1, 1) = COV_MEASUREMENT(2, 2) = 0.0025;
    const std::string mode = argv[1], d = std::string(argv[2]) + "/";
    try {
        if (mode == "cpu") return run_cpu(d);
        if (mode == "gpu") return run_gpu(d);
    } catch (const std::exception &e) {
        std::fprintf(stderr, "refcut_selftest: %s\n", e.what());
        return 1;
    }
    return 2;
}
