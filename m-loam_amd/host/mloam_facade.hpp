// C++ facade over the C-ABI (include/mloam_hip.h) that keeps M-LOAM's host-side API surface for the scan-to-map hot path,
// so the mapper / estimator sources can switch to the MI355X path by swapping includes.
//
// Mirrored reference interfaces (names, argument meaning, output ordering and error behaviour):
//   FeatureExtract::extractCloud / match{Surf,Corner}FromMap / match{Surf,Corner}PointFromMap
//                                         estimator/src/featureExtract/feature_extract.hpp:56-128
//   pcl::KdTreeFLANN<PointT>::setInputCloud (the object handed to the match functions)     lidar_mapper_keyframe.cpp:433-434
//   PointPlaneFeature, ScanInfo, cloudFeature                                              estimator/src/estimator/parameters.h:161-207
//   Pose (q_, t_)                                                                           estimator/src/estimator/pose.h:38-66
//   PoseLocalParameterization {Plus, ComputeJacobian, GlobalSize, LocalSize, setParameter, is_degenerate_, V_update_}
//                                         estimator/src/factor/pose_local_parameterization.h:21-32
//   evalDegenracy(mat_H, local_parameterization)                                            lidar_mapper_keyframe.cpp:1172-1204
//   scan2MapOptimization()                                                                  lidar_mapper_keyframe.cpp:423-639
//   ceres::CostFunction::Evaluate(double const* const*, double*, double**) for the map factors  lidar_map_factor.hpp:44,143
//
// The image this was developed in has no PCL / Eigen / Ceres, so the few types the signatures need are declared here with the
// reference's field names and memory layout (pcl::PointXYZI is 32 bytes, pcl::PointXYZIWithCov 48 bytes); inside the
// reference tree, define MLOAM_FACADE_USE_PCL_TYPES and the real headers' types are used instead (same layouts).
// All heavy work goes through libmloam_hip.so; this header contains no numerical fallback.
#pragma once
#include <algorithm>
#include <array>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <cmath>
#include <condition_variable>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/mloam_hip.h"

// Inside the reference tree (-DMLOAM_FACADE_CERES_BASES): the per-factor classes ARE ceres::SizedCostFunction<...> and PoseLocalParameterization IS a
// ceres::LocalParameterization -- `new`ed by the caller, handed to problem.AddResidualBlock / AddParameterBlock and owned (deleted) by ceres::Problem, as the
// reference's own classes are (lidar_mapper_keyframe.cpp:446-450, 547-548, 562-563). Without the macro they are plain classes with the same members.
#ifdef MLOAM_FACADE_CERES_BASES
#include <ceres/ceres.h>
#define MLOAM_FACADE_COST_BASE(...) : public ceres::SizedCostFunction<__VA_ARGS__>
#define MLOAM_FACADE_LOCAL_PARAM_BASE : public ceres::LocalParameterization
#define MLOAM_FACADE_VIRTUAL virtual
#else
#define MLOAM_FACADE_COST_BASE(...)
#define MLOAM_FACADE_LOCAL_PARAM_BASE
#define MLOAM_FACADE_VIRTUAL
#endif

namespace mloam_hip {

// ------------------------------------------------------------------ point / cloud types
// Inside the reference tree (-DMLOAM_FACADE_USE_PCL_TYPES): the real pcl::PointXYZ / pcl::PointXYZI / pcl::PointXYZIWithCov and pcl::PointCloud<T> -- this header
// includes <pcl/point_types.h> and <pcl/point_cloud.h> itself and, when it is on the include path, the reference's own <mloam_pcl/point_with_cov.hpp>
// (otherwise include the header that declares pcl::PointXYZIWithCov before this one). Nothing else has to be declared by the includer.
// Everywhere else: layout-compatible stand-ins with the same member names.
#ifdef MLOAM_FACADE_USE_PCL_TYPES
}  // namespace mloam_hip
#include <pcl/point_types.h>
#include <pcl/point_cloud.h>
#if defined(__has_include)
#if __has_include(<mloam_pcl/point_with_cov.hpp>)
#include <mloam_pcl/point_with_cov.hpp>
#endif
#endif
namespace mloam_hip {
using PointI = pcl::PointXYZI;
using PointXYZ = pcl::PointXYZ;                    // the raw driver cloud's point (FeatureExtract::calTimestamp's input)
using PointIWithCov = pcl::PointXYZIWithCov;       // mloam_pcl/point_with_cov.hpp:45-101
template <typename PointT> using PointCloud = pcl::PointCloud<PointT>;
#else
struct alignas(16) PointI {           // pcl::PointXYZI: float data[4]; float intensity; pad[3]  -> 32 bytes
    float x = 0, y = 0, z = 0, pad_ = 1.f;
    float intensity = 0, pad2_[3] = {0, 0, 0};
};
struct alignas(16) PointXYZ {         // pcl::PointXYZ: float data[4] -> 16 bytes (the raw driver cloud, FeatureExtract::calTimestamp's input)
    float x = 0, y = 0, z = 0, pad_ = 1.f;
};
struct alignas(16) PointIWithCov {    // pcl::PointXYZIWithCov (mloam_pcl/point_with_cov.hpp:45-53) -> 48 bytes
    float x = 0, y = 0, z = 0, pad_ = 1.f;
    float intensity = 0;
    float cov_vec[6] = {0, 0, 0, 0, 0, 0};   // cxx cxy cxz cyy cyz czz
    float cov_trace = 0;
};
template <typename PointT>
struct PointCloud {
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    size_t size() const { return points.size(); }
    void push_back(const PointT &p) { points.push_back(p); }
    void clear() { points.clear(); }
    const PointT &operator[](size_t i) const { return points[i]; }
    PointT &operator[](size_t i) { return points[i]; }
};
#endif
static_assert(sizeof(PointXYZ) == 16, "pcl::PointXYZ layout");
static_assert(sizeof(PointI) == 32 && offsetof(PointI, intensity) == 16, "pcl::PointXYZI layout");
static_assert(sizeof(PointIWithCov) == 48 && offsetof(PointIWithCov, intensity) == 16 && offsetof(PointIWithCov, cov_vec) == 20 && offsetof(PointIWithCov, cov_trace) == 44,
              "pcl::PointXYZIWithCov layout (mloam_pcl/point_with_cov.hpp:45-53)");
typedef PointCloud<PointXYZ> PointXYZCloud;      // the reference's `PointCloud` (common/types/type.h:16: pcl::PointCloud<pcl::PointXYZ>)
typedef PointCloud<PointI> PointICloud;
typedef PointCloud<PointIWithCov> PointICovCloud;
typedef std::map<std::string, PointICloud> cloudFeature;   // parameters.h:161

template <typename PointT> struct point_traits;
template <> struct point_traits<PointI> { static constexpr int intensity_off = 16, cov_off = -1; };
template <> struct point_traits<PointIWithCov> { static constexpr int intensity_off = 16, cov_off = 20; };

struct Quat { double w = 1, x = 0, y = 0, z = 0; };       // Eigen::Quaterniond accessor order w,x,y,z
struct Vec3 { double v[3] = {0, 0, 0}; double &operator()(int i) { return v[i]; } double operator()(int i) const { return v[i]; } };

class Pose {                                              // pose.h:38-66 (q_, t_ only: what the hot path reads)
public:
    Quat q_;
    Vec3 t_;
    std::array<double, 36> cov_{};                        // pose_wmap_curr.cov_ = cov_mapping (cpp:631)
    void toParam(double p[7]) const { p[0] = t_(0); p[1] = t_(1); p[2] = t_(2); p[3] = q_.x; p[4] = q_.y; p[5] = q_.z; p[6] = q_.w; }
    void fromParam(const double p[7]) { t_(0) = p[0]; t_(1) = p[1]; t_(2) = p[2]; q_.x = p[3]; q_.y = p[4]; q_.z = p[5]; q_.w = p[6]; }
};

class PointPlaneFeature {                                 // parameters.h:163-175
public:
    PointPlaneFeature() : idx_(0), laser_idx_(0), type_('n') {}
    size_t idx_, laser_idx_;
    std::array<double, 3> point_{};
    std::vector<double> coeffs_;                          // Eigen::VectorXd: 4 for 's', 6 for 'c'
    std::vector<double> jaco_;                            // 1x6
    char type_;
};

class ScanInfo {                                          // parameters.h:193-207
public:
    ScanInfo(const int &n_scan, const bool &segment_flag) : segment_flag_(segment_flag) { scan_start_ind_.resize(n_scan); scan_end_ind_.resize(n_scan); }
    std::vector<int> scan_start_ind_, scan_end_ind_;
    bool segment_flag_;
    std::vector<bool> ground_flag_;
};

// ------------------------------------------------------------------ the reference's OWN types at the call sites (duck typing)
// Inside the reference tree the call sites hold the reference's types, not the stand-ins above: Pose with Eigen::Quaterniond q_ / Eigen::Vector3d t_ /
// Eigen::Matrix<double, 6, 6> cov_ (pose.h:38-66), PointPlaneFeature with Eigen::Vector3d point_ / Eigen::VectorXd coeffs_ / Eigen::MatrixXd jaco_
// (parameters.h:163-175), ScanInfo (parameters.h:193-207), Eigen::Matrix3d covariances, boost::shared_ptr'd clouds and indices. Every entry point of this
// header that mirrors a reference interface is a template over those types and touches them only through the members both families have; the few places
// where the spelling differs (q.w() against q.w, m(r, c) against m[r * n + c], MatrixXd::resize(1, n) against vector::resize(n), *ptr against a
// reference) go through the helpers below. tests/host/refcut compiles the reference's own call sites, cut verbatim, against exactly this.
namespace detail {
// a cloud of m points to be written by index (pcl::PointCloud keeps width / height beside `points`; push_back per point is ~4.5 ns x 115 k points per 64-ring cloud)
template <class C> auto cloud_set_dims(C &c, size_t m, int) -> decltype(void(c.width = 0)) { c.width = uint32_t(m); c.height = 1; }
template <class C> void cloud_set_dims(C &, size_t, long) {}
template <class C> void cloud_resize(C &c, size_t m) { c.points.resize(m); cloud_set_dims(c, m, 0); }
// the calling thread's scratch floats (slot 0 / 1): a fresh zero-filled vector of a 64-ring cloud's size per call is ~70 us of page faults and fills
inline std::vector<float> &thread_floats(int slot, size_t n) { thread_local std::vector<float> v[2]; if (v[slot].size() < n) v[slot].resize(n); return v[slot]; }
template <class Q> auto quat_w(const Q &q, int) -> decltype(double(q.w())) { return q.w(); }
template <class Q> auto quat_w(const Q &q, long) -> decltype(double(q.w)) { return q.w; }
template <class Q> auto quat_x(const Q &q, int) -> decltype(double(q.x())) { return q.x(); }
template <class Q> auto quat_x(const Q &q, long) -> decltype(double(q.x)) { return q.x; }
template <class Q> auto quat_y(const Q &q, int) -> decltype(double(q.y())) { return q.y(); }
template <class Q> auto quat_y(const Q &q, long) -> decltype(double(q.y)) { return q.y; }
template <class Q> auto quat_z(const Q &q, int) -> decltype(double(q.z())) { return q.z(); }
template <class Q> auto quat_z(const Q &q, long) -> decltype(double(q.z)) { return q.z; }
template <class Q> auto quat_set(Q &q, double w, double x, double y, double z, int) -> decltype(void(q.w() = w)) { q.w() = w; q.x() = x; q.y() = y; q.z() = z; }
template <class Q> auto quat_set(Q &q, double w, double x, double y, double z, long) -> decltype(void(q.w = w)) { q.w = w; q.x = x; q.y = y; q.z = z; }
// element (r, c) of an n-column matrix: Eigen's m(r, c), or a row-major std::array / std::vector / C array
template <class M> auto mat_at(const M &m, int r, int c, int /*n*/, int) -> decltype(double(m(r, c))) { return m(r, c); }
template <class M> auto mat_at(const M &m, int r, int c, int n, long) -> decltype(double(m[0])) { return m[size_t(r * n + c)]; }
template <class M> auto mat_ref(M &m, int r, int c, int /*n*/, int) -> decltype(m(r, c)) { return m(r, c); }
template <class M> auto mat_ref(M &m, int r, int c, int n, long) -> decltype(m[0]) { return m[size_t(r * n + c)]; }
// the reference's Pose keeps T_ in step with q_ / t_ through update() (pose.cpp:105-108); the stand-in has no such member
template <class P> auto pose_update(P &pose, int) -> decltype(void(pose.update())) { pose.update(); }
template <class P> void pose_update(P &, long) {}
template <class P> void pose_to_param(const P &pose, double p[7])          // [tx ty tz qx qy qz qw]: the parameter-block layout (vector2Double, cpp:236-243)
{
    p[0] = pose.t_(0); p[1] = pose.t_(1); p[2] = pose.t_(2);
    p[3] = quat_x(pose.q_, 0); p[4] = quat_y(pose.q_, 0); p[5] = quat_z(pose.q_, 0); p[6] = quat_w(pose.q_, 0);
}
template <class P> void pose_from_param(P &pose, const double p[7])
{
    pose.t_(0) = p[0]; pose.t_(1) = p[1]; pose.t_(2) = p[2];
    quat_set(pose.q_, p[6], p[3], p[4], p[5], 0);
    pose_update(pose, 0);
}
template <class P> void pose_cov_get(const P &pose, double c[36]) { for (int r = 0; r < 6; ++r) for (int k = 0; k < 6; ++k) c[r * 6 + k] = mat_at(pose.cov_, r, k, 6, 0); }
template <class P> void pose_cov_set(P &pose, const double c[36]) { for (int r = 0; r < 6; ++r) for (int k = 0; k < 6; ++k) mat_ref(pose.cov_, r, k, 6, 0) = c[r * 6 + k]; }
// PointPlaneFeature::jaco_: Eigen::MatrixXd (1 x 6) in the reference, std::vector<double> in the stand-in; both expose data()
template <class J> auto jaco_resize(J &j, int n, int) -> decltype(void(j.cols())) { j.resize(1, n); }
template <class J> void jaco_resize(J &j, int n, long) { j.resize(size_t(n)); }
template <class F> void feature_set(F &f, size_t idx, size_t laser_idx, char type, double x, double y, double z, const double *coeffs, int n_coeffs, const double *jaco6)
{
    f.idx_ = idx; f.laser_idx_ = laser_idx; f.type_ = type;
    f.point_[0] = x; f.point_[1] = y; f.point_[2] = z;
    f.coeffs_.resize(n_coeffs);
    for (int k = 0; k < n_coeffs; ++k) f.coeffs_[k] = coeffs[k];
    if (jaco6) { jaco_resize(f.jaco_, 6, 0); for (int k = 0; k < 6; ++k) f.jaco_.data()[k] = jaco6[k]; }
}
// a kd-tree / cloud handed over as the object or as a (boost / std) shared_ptr to it
template <class T> auto deref(const T &x, int) -> decltype(*x) { return *x; }
template <class T> const T &deref(const T &x, long) { return x; }
}  // namespace detail
// reference Pose <-> stand-in Pose (either direction, any two pose types with q_ / t_ / cov_)
template <class PoseA, class PoseB> void copyPose(const PoseA &from, PoseB &to)
{
    double p[7], c[36];
    detail::pose_to_param(from, p); detail::pose_cov_get(from, c);
    detail::pose_from_param(to, p); detail::pose_cov_set(to, c);
}

// hot-path globals of parameters.h that the match functions read (MIN_MATCH_SQ_DIS, MIN_PLANE_DIS, ...)
struct Params {
    float MIN_MATCH_SQ_DIS = 1.0f, MIN_PLANE_DIS = 0.2f;
    double MAP_EIG_THRE = 100.0, HUBER_DELTA = 0.1, COV_MEASUREMENT_TRACE = 0.0075;
    double COV_MEASUREMENT[9] = {0.0025, 0, 0, 0, 0.0025, 0, 0, 0, 0.0025};   // config uct_measurement
    double TRACE_THRESHOLD_MAPPING = 0.6;
    int N_SCANS = 16;
    // the odometry window (parameters.h:58-111)
    int OPT_WINDOW_SIZE = 4, NUM_OF_LASER = 2, ESTIMATE_EXTRINSIC = 1, N_CUMU_FEATURE = 10;
    double LAMBDA_THRE_CALIB = 70.0;
    // keyframes of the mapper (parameters.cpp:98-100, 268-270; values of config_realvehicle_hercules.yaml:142-144)
    float DISTANCE_KEYFRAMES = 1.0f, ORIENTATION_KEYFRAMES = 1.0f, SURROUNDING_KF_RADIUS = 50.0f;
};
inline Params &params() { static Params p; return p; }

class Error : public std::runtime_error { public: using std::runtime_error::runtime_error; };

// ------------------------------------------------------------------ one device context (RAII)
class Device {
public:
    explicit Device(int device_id = 0)
    {
        int rc = mlh_create(&ctx_, device_id);
        if (rc != MLH_OK) throw Error("mlh_create failed: no usable MI355X / HIP device (the product path has no CPU fallback)");
    }
    ~Device() { mlh_destroy(ctx_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    mlh_ctx *ctx() const { return ctx_; }
    void check(int rc) const { if (rc != MLH_OK) throw Error(std::string("mloam_hip: ") + mlh_last_error(ctx_)); }
private:
    mlh_ctx *ctx_ = nullptr;
};

// A context is thread-compatible, not re-entrant (one stream, one set of scan buffers). The reference calls ImageSegmenter::segmentCloud and
// FeatureExtract::extractCloud on ONE object from NUM_OF_LASER OpenMP threads at once (estimator.cpp:249-263), so the default-constructed facade objects do not
// hold a context: every calling thread gets its own from here, created on the thread's first call and kept for the thread's lifetime (OpenMP keeps its workers
// between parallel regions, so a frame pays no context creation). setThreadDeviceId picks the GPU the pool creates contexts on (default 0).
inline int &threadDeviceId() { static int id = 0; return id; }
inline void setThreadDeviceId(int device_id) { threadDeviceId() = device_id; }
inline Device &threadDevice()
{
    thread_local std::unique_ptr<Device> dev;
    if (!dev) dev.reset(new Device(threadDeviceId()));
    return *dev;
}

// ------------------------------------------------------------------ the "kd-tree" handed to the match functions
// Mirrors pcl::KdTreeFLANN<PointT>: setInputCloud(cloud) (re)builds the index. kind selects which of the context's two
// resident maps this object stands for (the mapper keeps kdtree_surf_from_map / kdtree_corner_from_map, cpp:59-62).
template <typename PointT>
class MapIndex {
public:
    typedef std::shared_ptr<MapIndex<PointT>> Ptr;       // the mapper holds pcl::KdTreeFLANN<PointIWithCov>::Ptr kdtree_*_from_map (lidar_mapper_keyframe.cpp:59-62)
    MapIndex(Device &dev, int kind) : dev_(dev), kind_(kind) {}
    // kdtree_surf_from_map->setInputCloud(laser_cloud_surf_from_map_cov_ds) (cpp:433-434) hands a PointCloud::Ptr over
    template <typename CloudPtr>
    auto setInputCloud(const CloudPtr &cloud) -> decltype(void((*cloud).points)) { setInputCloud(*cloud); }
    void setInputCloud(const PointCloud<PointT> &cloud)
    {
        if (cloud.size() == 0) throw Error("setInputCloud: empty cloud");
        dev_.check(mlh_map_set(dev_.ctx(), kind_, cloud.points.data(), (int)sizeof(PointT), (int)cloud.size(), params().MIN_MATCH_SQ_DIS, MLH_MEM_HOST));
        n_ = cloud.size();
    }
    // nearestKSearch for k = 5 (feature_extract.hpp:666): indices into the cloud given to setInputCloud. Exact (= a kd-tree's answer)
    // for every neighbour closer than the acceptance radius the index was built for (sqrt(min_match_sq_dis)); a neighbour reported
    // beyond that radius may not be the true k-th nearest (the 27-cell search does not look farther) -- see mlh_knn in mloam_hip.h
    int nearestKSearch(const PointT &p, int k, std::vector<int> &idx, std::vector<float> &sqd) const
    {
        idx.assign(k, -1); sqd.assign(k, 0.f);
        const float q[3] = {p.x, p.y, p.z};
        dev_.check(mlh_knn(dev_.ctx(), kind_, q, 1, k, idx.data(), sqd.data()));
        int found = 0;
        for (int i = 0; i < k; ++i) if (idx[i] >= 0) ++found;
        return found;
    }
    Device &device() const { return dev_; }
    int kind() const { return kind_; }
    size_t size() const { return n_; }
private:
    Device &dev_;
    int kind_;
    size_t n_ = 0;
};

// ------------------------------------------------------------------ pcl::VoxelGridCovarianceMLOAM<PointT>
// The covariance-aware voxel filter (mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam.h:95-399): same setter names, setInputCloud
// takes the cloud by reference instead of a boost::shared_ptr. PointIWithCov selects the covariance branch, PointI the plain one
// (the `cov_index >= 0` test of voxel_grid_covariance_mloam_impl.hpp:286).
template <typename PointT> struct VoxelFields;
template <> struct VoxelFields<PointI> { static constexpr int intensity = 16, cov = -1, trace = -1; };
template <> struct VoxelFields<PointIWithCov> { static constexpr int intensity = 16, cov = 20, trace = 44; };

template <typename PointT>
class VoxelGridCovarianceMLOAM {
public:
    explicit VoxelGridCovarianceMLOAM(Device &dev) : dev_(dev) {}
    void setLeafSize(float lx, float ly, float lz)
    {
        if (lx != ly || lx != lz) throw Error("VoxelGridCovarianceMLOAM: cubic leaves only (every call site of the reference uses them)");
        leaf_ = lx;
    }
    void setTraceThreshold(const float trace_threshold) { trace_threshold_ = trace_threshold; }   // .h:349
    void setInputCloud(const PointCloud<PointT> &cloud) { input_ = &cloud; keep_.reset(); }
    // as the reference calls it: a PointCloud::Ptr, possibly a temporary (`setInputCloud(boost::make_shared<PointICovCloud>(cloud))`) -- a copy of the smart
    // pointer is kept until the next setInputCloud, as pcl::Filter keeps input_
    template <typename CloudPtr>
    auto setInputCloud(const CloudPtr &cloud) -> decltype(void((*cloud).points)) { auto k = std::make_shared<CloudPtr>(cloud); input_ = &**k; keep_ = k; }
    void filter(PointCloud<PointT> &output)
    {
        if (!input_ || input_->size() == 0) { output.points.clear(); return; }   // "No input dataset given" -> empty output (impl.hpp:71-77)
        std::vector<PointT> out(input_->size());
        int32_t n_out = 0;
        dev_.check(mlh_voxel_filter(dev_.ctx(), input_->points.data(), (int)sizeof(PointT), (int)input_->size(), VoxelFields<PointT>::intensity,
                                    VoxelFields<PointT>::cov, VoxelFields<PointT>::trace, leaf_, trace_threshold_, out.data(), &n_out, MLH_MEM_HOST));
        out.resize(n_out);
        output.points.assign(out.begin(), out.end());
    }
private:
    Device &dev_;
    const PointCloud<PointT> *input_ = nullptr;
    std::shared_ptr<void> keep_;
    float leaf_ = 0.4f, trace_threshold_ = 2.0f;   // .h:102
};

// ------------------------------------------------------------------ local-map assembly (lidar_mapper.h:118-119, associate_uct.hpp:90-147)
// compoundPoseWithCov(pose_1, pose_2, pose_cp): method 2, the only one the mapper uses
template <typename PoseT>
inline void compoundPoseWithCov(const PoseT &pose_1, const PoseT &pose_2, PoseT &pose_cp)
{
    double p1[7], p2[7], pc[7], c1[36], c2[36], cc[36];
    detail::pose_to_param(pose_1, p1); detail::pose_to_param(pose_2, p2);
    detail::pose_cov_get(pose_1, c1); detail::pose_cov_get(pose_2, c2);
    if (mlh_compound_pose_with_cov(p1, c1, p2, c2, pc, cc) != MLH_OK) throw Error("compoundPoseWithCov");
    detail::pose_from_param(pose_cp, pc);
    detail::pose_cov_set(pose_cp, cc);
}
namespace detail {
template <typename PoseVec> void pack_extrinsics(const PoseVec &pose_ext, std::vector<double> &ext, std::vector<double> &ext_cov)
{
    ext.resize(pose_ext.size() * 7); ext_cov.resize(pose_ext.size() * 36);
    for (size_t n = 0; n < pose_ext.size(); ++n) { pose_to_param(pose_ext[n], ext.data() + n * 7); pose_cov_get(pose_ext[n], ext_cov.data() + n * 36); }
}
}  // namespace detail

// cloudUCTAssociateToMap(cloud_local, cloud_global, pose_global, pose_ext): the reference reads the globals with_ua_flag,
// COV_MEASUREMENT and TRACE_THRESHOLD_MAPPING; here they are the last argument and params().
template <typename PoseT, typename PoseVec>
inline void cloudUCTAssociateToMap(Device &dev, const PointICovCloud &cloud_local, PointICovCloud &cloud_global, const PoseT &pose_global,
                                   const PoseVec &pose_ext, bool with_ua_flag)
{
    cloud_global.points.clear();
    if (cloud_local.size() == 0) return;
    std::vector<double> ext, ext_cov;
    detail::pack_extrinsics(pose_ext, ext, ext_cov);
    double pg[7], pg_cov[36];
    detail::pose_to_param(pose_global, pg);
    detail::pose_cov_get(pose_global, pg_cov);
    std::vector<PointIWithCov> out(cloud_local.size());
    int32_t n_out = 0;
    dev.check(mlh_cloud_uct_associate_to_map(dev.ctx(), cloud_local.points.data(), (int)sizeof(PointIWithCov), (int)cloud_local.size(),
                                             VoxelFields<PointIWithCov>::intensity, VoxelFields<PointIWithCov>::cov, VoxelFields<PointIWithCov>::trace,
                                             pg, pg_cov, ext.data(), ext_cov.data(), (int)pose_ext.size(), params().COV_MEASUREMENT,
                                             with_ua_flag ? 1 : 0, params().TRACE_THRESHOLD_MAPPING, out.data(), &n_out, MLH_MEM_HOST));
    out.resize(n_out);
    cloud_global.points.assign(out.begin(), out.end());
}

// downsampleCurrentScan() for one feature cloud (lidar_mapper_keyframe.cpp:356-421): thin at `leaf`, attach the extrinsic-induced
// covariance, drop what exceeds TRACE_THRESHOLD_MAPPING. The result is returned AND stays on the device as the kind's feature set,
// so a following scan2map call needs no mlh_features_set for it.
template <typename PoseVec>
inline void downsampleCurrentScan(Device &dev, int kind, const PointICloud &laser_cloud_last, float leaf, const PoseVec &pose_ext,
                                  bool with_ua_flag, PointICovCloud &laser_cloud_cov)
{
    laser_cloud_cov.points.clear();
    if (laser_cloud_last.size() == 0) return;
    std::vector<double> ext, ext_cov;
    detail::pack_extrinsics(pose_ext, ext, ext_cov);
    std::vector<float> out(laser_cloud_last.size() * 11);
    int32_t m = 0;
    dev.check(mlh_downsample_current_scan(dev.ctx(), kind, laser_cloud_last.points.data(), (int)sizeof(PointI), (int)laser_cloud_last.size(),
                                          point_traits<PointI>::intensity_off, MLH_MEM_HOST, leaf, ext.data(), ext_cov.data(), (int)pose_ext.size(),
                                          params().COV_MEASUREMENT, with_ua_flag ? 1 : 0, params().TRACE_THRESHOLD_MAPPING, out.data(), &m));
    for (int i = 0; i < m; ++i) {
        PointIWithCov p;
        const float *o = out.data() + size_t(i) * 11;
        p.x = o[0]; p.y = o[1]; p.z = o[2]; p.intensity = o[3];
        for (int k = 0; k < 6; ++k) p.cov_vec[k] = o[4 + k];
        p.cov_trace = o[10];
        laser_cloud_cov.push_back(p);
    }
}

// ------------------------------------------------------------------ FeatureExtract
class FeatureExtract {
public:
    // As the reference's member `FeatureExtract f_extract_;` (estimator.h:190): no context of its own -- every calling thread works on its own (threadDevice()),
    // so extractCloud is RE-ENTRANT on one object, as estimator.cpp:249-263 needs it (one object, NUM_OF_LASER OpenMP threads).
    FeatureExtract() : bound_(nullptr) {}
    // Bound to one context: for callers that keep the extraction's device-resident results (extractCloudOnDevice -> LidarTracker / fuseCloudFeature on the same
    // Device). One thread at a time.
    explicit FeatureExtract(Device &dev) : bound_(&dev) {}

    // feature_extract.cpp:38-113: a point's relative time inside the sweep from its azimuth (the `intensity` field of the output), the reference's two-phase
    // unwrapping (`half_passed`) restated. Host code by nature: one sequential pass with carried state over a cloud that is still on the host; ImageSegmenter
    // overwrites the field with the ring id anyway (image_segmenter.hpp:371). Here so that estimator.cpp:249-263 compiles against this header as it stands.
    template <typename PointXYZ>
    void calTimestamp(const PointCloud<PointXYZ> &laser_cloud_in, PointICloud &laser_cloud_out, float scan_period = 0.1f) const
    {
        const size_t n = laser_cloud_in.size();
        laser_cloud_out.points.resize(n);
        if (n == 0) return;
        // The reference's variables are float, its constants (M_PI) double: every `ori += 2 * M_PI` is a DOUBLE addition rounded back to float, every comparison
        // against `start_ori - M_PI / 2` a double comparison -- spelled out here, term by term (pinned against the reference's own lines:
        // tests/test_abi.py::test_facade_cal_timestamp_is_the_references). The angle itself: atan2 on two floats = the float overload.
        float start_ori = -std::atan2(laser_cloud_in.points[0].y, laser_cloud_in.points[0].x);
        float end_ori = float(double(-std::atan2(laser_cloud_in.points[n - 1].y, laser_cloud_in.points[n - 1].x)) + 2 * M_PI);
        if (double(end_ori - start_ori) > 3 * M_PI) end_ori = float(double(end_ori) - 2 * M_PI);
        else if (double(end_ori - start_ori) < M_PI) end_ori = float(double(end_ori) + 2 * M_PI);
        bool half_passed = false;
        for (size_t i = 0; i < n; ++i) {
            PointI q;
            q.x = laser_cloud_in.points[i].x; q.y = laser_cloud_in.points[i].y; q.z = laser_cloud_in.points[i].z;
            float ori = -std::atan2(q.y, q.x);
            if (!half_passed) {
                if (double(ori) < double(start_ori) - M_PI / 2) ori = float(double(ori) + 2 * M_PI);
                else if (double(ori) > double(start_ori) + M_PI * 3 / 2) ori = float(double(ori) - 2 * M_PI);
                if (double(ori - start_ori) > M_PI) half_passed = true;
            } else {
                ori = float(double(ori) + 2 * M_PI);
                if (double(ori) < double(end_ori) - M_PI * 3 / 2) ori = float(double(ori) + 2 * M_PI);
                else if (double(ori) > double(end_ori) + M_PI / 2) ori = float(double(ori) - 2 * M_PI);
            }
            q.intensity = (ori - start_ori) / (end_ori - start_ori) * scan_period;
            laser_cloud_out.points[i] = q;
        }
    }

    // feature_extract.cpp:118-297. Output keys and ordering as the reference (cpp:281-285), "surf_points_less_flat" thinned by the
    // per-ring 0.2 m VoxelGrid (cpp:266-271) with the intensity field (ring id) averaged along, as PCL does.
    // Re-entrancy: see the constructors. Nothing of a call lives in the object (the labels of a thread's last call: cloudLabel()).
    template <typename ScanInfoT>
    void extractCloud(const PointICloud &laser_cloud_in, const ScanInfoT &scan_info, cloudFeature &cloud_feature)
    {
        Device &dev_ = device();
        std::vector<int32_t> &labels_ = threadLabels();
        const int n = (int)laser_cloud_in.size();
        const int rings = (int)scan_info.scan_start_ind_.size();
        dev_.check(mlh_scan_upload(dev_.ctx(), laser_cloud_in.points.data(), (int)sizeof(PointI), point_traits<PointI>::intensity_off, n, scan_info.scan_start_ind_.data(),
                                   scan_info.scan_end_ind_.data(), rings, MLH_MEM_HOST));
        dev_.check(mlh_extract_run(dev_.ctx()));
        thread_local std::vector<int32_t> lists[4];        // (the calling thread's: four fresh index vectors of the cloud's size are 1.8 MB of fills per 64-ring call)
        int32_t *ptrs[4];
        int32_t counts[4] = {0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) { if (lists[i].size() < size_t(n > 0 ? n : 1)) lists[i].resize(n > 0 ? n : 1); ptrs[i] = lists[i].data(); }
        labels_.resize(n);
        dev_.check(mlh_extract_fetch(dev_.ctx(), labels_.data(), nullptr, nullptr, ptrs, counts));
        cloud_feature.clear();
        cloud_feature["laser_cloud"] = laser_cloud_in;
        static const char *names[3] = {"corner_points_sharp", "corner_points_less_sharp", "surf_points_flat"};
        for (int i = 0; i < 3; ++i) {
            PointICloud &c = cloud_feature[names[i]];
            detail::cloud_resize(c, size_t(counts[i]));
            for (int k = 0; k < counts[i]; ++k) c.points[size_t(k)] = laser_cloud_in.points[lists[i][k]];
        }
        // "surf_points_less_flat": the label <= 0 points after the per-ring 0.2 m pcl::VoxelGrid (cpp:266-271), ring by ring
        dev_.check(mlh_extract_voxel_run(dev_.ctx(), 0.2f));
        std::vector<float> &vox = detail::thread_floats(0, size_t(counts[3] > 0 ? counts[3] : 1) * 4);
        int32_t n_vox = 0;
        dev_.check(mlh_extract_fetch_voxel(dev_.ctx(), vox.data(), &n_vox));
        PointICloud &lf = cloud_feature["surf_points_less_flat"];
        detail::cloud_resize(lf, size_t(n_vox));
        for (int k = 0; k < n_vox; ++k) { PointI p; p.x = vox[4 * k]; p.y = vox[4 * k + 1]; p.z = vox[4 * k + 2]; p.intensity = vox[4 * k + 3]; lf.points[size_t(k)] = p; }
    }
    const std::vector<int32_t> &cloudLabel() const { return threadLabels(); }   // cloud_label[] of the calling thread's last extractCloud
    // The NEXT cloud's points on their way to the device while this thread's context still works on the current one (mlh_scan_upload_ahead): the extractCloud /
    // extractCloudOnDevice call that is handed the SAME cloud object (unchanged in between) finds them there. For callers that have the next sweep already -- a
    // replayed bag; nothing in the reference corresponds to it.
    void sendAhead(const PointICloud &next_laser_cloud_in)
    {
        if (next_laser_cloud_in.size() == 0) return;
        Device &dev_ = device();
        dev_.check(mlh_scan_upload_ahead(dev_.ctx(), next_laser_cloud_in.points.data(), (int)sizeof(PointI), (int)next_laser_cloud_in.size()));
    }
    Device &device() const { return bound_ ? *bound_ : threadDevice(); }

    // The same extraction with nothing fetched: the four feature lists and the thinned less-flat cloud stay in HBM for
    // LidarTracker::set*FromExtractor and fuseCloudFeature (one scan per Device at a time).
    template <typename ScanInfoT>
    void extractCloudOnDevice(const PointICloud &laser_cloud_in, const ScanInfoT &scan_info)
    {
        Device &dev_ = device();
        dev_.check(mlh_scan_upload(dev_.ctx(), laser_cloud_in.points.data(), (int)sizeof(PointI), point_traits<PointI>::intensity_off, (int)laser_cloud_in.size(),
                                   scan_info.scan_start_ind_.data(), scan_info.scan_end_ind_.data(), (int)scan_info.scan_start_ind_.size(), MLH_MEM_HOST));
        dev_.check(mlh_extract_run(dev_.ctx()));
        dev_.check(mlh_extract_voxel_run(dev_.ctx(), 0.2f));
    }

    // ... on the scan the Device already holds (ImageSegmenter::segmentCloudOnDevice left it there, ring-major, with its ring table): no upload at all
    void extractStagedCloudOnDevice()
    {
        Device &dev_ = device();
        dev_.check(mlh_extract_run(dev_.ctx()));
        dev_.check(mlh_extract_voxel_run(dev_.ctx(), 0.2f));
    }

    // feature_extract.hpp:542-643 / 379-538: batch matching, matches compacted in input order. The kd-tree argument is the MapIndex (or a shared_ptr to it: the
    // reference passes `const typename pcl::KdTreeFLANN<PointType>::Ptr &`); pose and feature types are the caller's (the stand-ins above or the reference's
    // Pose / PointPlaneFeature with their Eigen members).
    template <typename KdTreeT, typename PointT, typename PoseT, typename FeatureVec>
    void matchSurfFromMap(const KdTreeT &kdtree_surf_from_map, const PointCloud<PointT> & /*cloud_map*/, const PointCloud<PointT> &cloud_data,
                          const PoseT &pose_local, FeatureVec &features, const size_t &N_NEIGH = 5, const bool &CHECK_FOV = true)
    { matchFromMap(detail::deref(kdtree_surf_from_map, 0), cloud_data, pose_local, features, N_NEIGH, CHECK_FOV, 's'); }

    template <typename KdTreeT, typename PointT, typename PoseT, typename FeatureVec>
    void matchCornerFromMap(const KdTreeT &kdtree_corner_from_map, const PointCloud<PointT> & /*cloud_map*/, const PointCloud<PointT> &cloud_data,
                            const PoseT &pose_local, FeatureVec &features, const size_t &N_NEIGH = 5, const bool &CHECK_FOV = true)
    { matchFromMap(detail::deref(kdtree_corner_from_map, 0), cloud_data, pose_local, features, N_NEIGH, CHECK_FOV, 'c'); }

    // feature_extract.hpp:646-883: single-point versions (one-element batch; prefer the batch calls)
    template <typename KdTreeT, typename PointT, typename PoseT, typename FeatureT>
    bool matchSurfPointFromMap(const KdTreeT &kdtree, const PointCloud<PointT> &cloud_map, const PointT &point_ori, const PoseT &pose_local,
                               FeatureT &feature, const size_t &idx, const size_t &N_NEIGH = 5, const bool &CHECK_FOV = true)
    { return matchPoint(detail::deref(kdtree, 0), cloud_map, point_ori, pose_local, feature, idx, N_NEIGH, CHECK_FOV, 's'); }

    template <typename KdTreeT, typename PointT, typename PoseT, typename FeatureT>
    bool matchCornerPointFromMap(const KdTreeT &kdtree, const PointCloud<PointT> &cloud_map, const PointT &point_ori, const PoseT &pose_local,
                                 FeatureT &feature, const size_t &idx, const size_t &N_NEIGH = 5, const bool &CHECK_FOV = true)
    { return matchPoint(detail::deref(kdtree, 0), cloud_map, point_ori, pose_local, feature, idx, N_NEIGH, CHECK_FOV, 'c'); }

private:
    template <typename PointT, typename PoseT, typename FeatureVec>
    void matchFromMap(const MapIndex<PointT> &kd, const PointCloud<PointT> &cloud_data, const PoseT &pose_local,
                      FeatureVec &features, size_t n_neigh, bool check_fov, char type)
    {
        typedef typename FeatureVec::value_type FeatureT;
        features.clear();
        const int m = (int)cloud_data.size();
        if (m == 0) return;
        Device &dev = kd.device();
        dev.check(mlh_features_set(dev.ctx(), kd.kind(), cloud_data.points.data(), (int)sizeof(PointT), m, point_traits<PointT>::intensity_off,
                                   point_traits<PointT>::cov_off, MLH_MEM_HOST));
        double pose[7];
        detail::pose_to_param(pose_local, pose);
        std::vector<uint8_t> valid(m);
        std::vector<double> coeffs(size_t(m) * 6), r(m), J(size_t(m) * 6);
        const Params &P = params();
        dev.check(mlh_match_linearize(dev.ctx(), kd.kind(), pose, (int)n_neigh, (check_fov ? MLH_FLAG_CHECK_FOV : 0u) | MLH_FLAG_WITH_UA | MLH_FLAG_NO_LOSS,
                                      P.MIN_MATCH_SQ_DIS, P.MIN_PLANE_DIS, 0.0, P.COV_MEASUREMENT_TRACE, valid.data(), coeffs.data(), r.data(), J.data(),
                                      nullptr, nullptr, nullptr, nullptr));
        for (int i = 0; i < m; ++i) {
            if (!valid[i]) continue;
            FeatureT f;
            // jaco_: what evaluateFeatJacobianMatching stores (lidar_mapper.h:162-164)
            detail::feature_set(f, size_t(i), (size_t)cloud_data.points[i].intensity, type, double(cloud_data.points[i].x), double(cloud_data.points[i].y),
                                double(cloud_data.points[i].z), coeffs.data() + size_t(i) * 6, type == 's' ? 4 : 6, J.data() + size_t(i) * 6);
            features.push_back(f);
        }
    }
    template <typename PointT, typename PoseT, typename FeatureT>
    bool matchPoint(const MapIndex<PointT> &kd, const PointCloud<PointT> &cloud_map, const PointT &point_ori, const PoseT &pose_local,
                    FeatureT &feature, size_t idx, size_t n_neigh, bool check_fov, char type)
    {
        PointCloud<PointT> one;
        one.push_back(point_ori);
        std::vector<FeatureT> out;
        if (type == 's') matchSurfFromMap(kd, cloud_map, one, pose_local, out, n_neigh, check_fov);
        else matchCornerFromMap(kd, cloud_map, one, pose_local, out, n_neigh, check_fov);
        if (out.empty()) return false;
        feature = out[0];
        feature.idx_ = idx;
        return true;
    }
    static std::vector<int32_t> &threadLabels() { thread_local std::vector<int32_t> l; return l; }
    Device *bound_;
};

// ------------------------------------------------------------------ PoseLocalParameterization (host side of the GN step)
class PoseLocalParameterization MLOAM_FACADE_LOCAL_PARAM_BASE {       // pose_local_parameterization.h:21-32
public:
    MLOAM_FACADE_VIRTUAL ~PoseLocalParameterization() {}
    MLOAM_FACADE_VIRTUAL bool Plus(const double *x, const double *delta, double *x_plus_delta) const { return mlh_pose_plus(x, delta, V_update_.data(), x_plus_delta) == MLH_OK; }
    MLOAM_FACADE_VIRTUAL bool ComputeJacobian(const double * /*x*/, double *jacobian) const   // [I6; 0], row-major 7x6
    {
        std::memset(jacobian, 0, sizeof(double) * 42);
        for (int i = 0; i < 6; ++i) jacobian[i * 6 + i] = 1.0;
        return true;
    }
    MLOAM_FACADE_VIRTUAL int GlobalSize() const { return 7; }
    MLOAM_FACADE_VIRTUAL int LocalSize() const { return 6; }
    void setParameter()
    {
        is_degenerate_ = false;
        V_update_.fill(0.0);
        for (int i = 0; i < 6; ++i) V_update_[i * 6 + i] = 1.0;
    }
    bool is_degenerate_ = false;
    std::array<double, 36> V_update_{};   // row-major 6x6
};

// lidar_mapper_keyframe.cpp:1172-1204
// mat_H: the reference's Eigen::Matrix<double, 6, 6> or a row-major std::array<double, 36>
template <typename Mat6>
inline void evalDegenracy(const Mat6 &mat_H, PoseLocalParameterization *local_parameterization, std::array<double, 6> *mat_E = nullptr)
{
    double ev[6], V[36], H[36];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[r * 6 + c] = detail::mat_at(mat_H, r, c, 6, 0);
    int deg = mlh_eval_degeneracy(H, params().MAP_EIG_THRE, ev, V);
    if (mat_E) std::memcpy(mat_E->data(), ev, sizeof(ev));
    if (deg > 0) {
        local_parameterization->is_degenerate_ = true;
        std::memcpy(local_parameterization->V_update_.data(), V, sizeof(V));
    }
}

// ------------------------------------------------------------------ a Ceres-shaped aggregate cost function
// One residual block for ALL features of a kind: Evaluate() runs the linearise kernel on the correspondences of the last
// match at parameters[0] and returns m residuals and the m x 7 row-major Jacobian -- what m LidarMap{PlaneNorm,Edge}Factor
// blocks (lidar_map_factor.hpp:26-235) return one by one. Inside the reference tree derive it from ceres::CostFunction
// (set_num_residuals(m), mutable_parameter_block_sizes()->push_back(7)).
class LidarMapBatchFactor {
public:
    LidarMapBatchFactor(Device &dev, int kind, int m, bool with_ua) : dev_(dev), kind_(kind), m_(m), with_ua_(with_ua), r_(m), J_(size_t(m) * 6) {}
    int num_residuals() const { return m_; }
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const
    {
        int rc = mlh_linearize(dev_.ctx(), kind_, parameters[0], (with_ua_ ? MLH_FLAG_WITH_UA : 0u) | MLH_FLAG_NO_LOSS, 0.0,
                               params().COV_MEASUREMENT_TRACE, r_.data(), J_.data(), nullptr, nullptr, nullptr, nullptr);
        if (rc != MLH_OK) return false;
        std::memcpy(residuals, r_.data(), sizeof(double) * m_);
        if (jacobians && jacobians[0]) {
            for (int i = 0; i < m_; ++i) {
                std::memcpy(jacobians[0] + size_t(i) * 7, J_.data() + size_t(i) * 6, sizeof(double) * 6);
                jacobians[0][size_t(i) * 7 + 6] = 0.0;
            }
        }
        return true;
    }
private:
    Device &dev_;
    int kind_, m_;
    bool with_ua_;
    mutable std::vector<double> r_, J_;
};

// ------------------------------------------------------------------ the odometry window's LidarPureOdom factors as one cost function
// Estimator::optimizeMap adds, for every window frame i > pivot, LiDAR n and matched feature, a LidarPureOdomPlaneNormFactor or
// LidarPureOdomEdgeFactor over (para_pose_[0], para_pose_[i - pivot_idx], para_ex_pose_[n]) (estimator.cpp:700-780). This
// aggregate stands for all of them: add(feature, frame, laser) while building the problem, then Evaluate() with the Ceres
// contract over the parameter blocks [pivot, frame_1 .. frame_W, ext_0 .. ext_{L-1}] (each double[7]); jacobians[b] is
// num_residuals x 7 row-major, rows of factors that do not touch block b are zero.
class LidarPureOdomBatchFactor {
public:
    LidarPureOdomBatchFactor(Device &dev, int n_window_frames, int n_lasers) : dev_(dev), W_(n_window_frames), L_(n_lasers) {}
    // feature: as produced by the match functions ('s': coeffs_ = [n, d]; 'c': coeffs_ = the two line points); s = sqrt_info (1.0 in the reference)
    template <typename FeatureT>
    void add(const FeatureT &feature, int frame /*1..W*/, int laser /*0..L-1*/, double s = 1.0)
    {
        type_.push_back(feature.type_ == 's' ? 0 : 1);
        for (int k = 0; k < 3; ++k) pts_.push_back(feature.point_[k]);
        for (int k = 0; k < 6; ++k) coef_.push_back(k < (int)feature.coeffs_.size() ? feature.coeffs_[k] : 0.0);
        s_.push_back(s); fi_.push_back(frame - 1); ei_.push_back(laser);
        staged_ = false;
    }
    int num_residuals() const { return (int)type_.size(); }
    int num_parameter_blocks() const { return 1 + W_ + L_; }
    bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const
    {
        const int n = num_residuals();
        if (n == 0) return true;
        if (!staged_) {
            if (mlh_pure_odom_set(dev_.ctx(), n, type_.data(), pts_.data(), coef_.data(), s_.data(), fi_.data(), ei_.data()) != MLH_OK) return false;
            staged_ = true;
            J_.resize(size_t(n) * 21);
        }
        std::vector<double> frames(size_t(W_) * 7), exts(size_t(L_) * 7);
        for (int w = 0; w < W_; ++w) std::memcpy(frames.data() + w * 7, parameters[1 + w], sizeof(double) * 7);
        for (int l = 0; l < L_; ++l) std::memcpy(exts.data() + l * 7, parameters[1 + W_ + l], sizeof(double) * 7);
        if (mlh_pure_odom_evaluate(dev_.ctx(), parameters[0], frames.data(), W_, exts.data(), L_, residuals, jacobians ? J_.data() : nullptr) != MLH_OK) return false;
        if (!jacobians) return true;
        for (int b = 0; b < num_parameter_blocks(); ++b) if (jacobians[b]) std::memset(jacobians[b], 0, sizeof(double) * 7 * size_t(n));
        for (int i = 0; i < n; ++i) {
            const double *J = J_.data() + size_t(i) * 21;
            if (jacobians[0]) std::memcpy(jacobians[0] + size_t(i) * 7, J, sizeof(double) * 7);
            if (jacobians[1 + fi_[i]]) std::memcpy(jacobians[1 + fi_[i]] + size_t(i) * 7, J + 7, sizeof(double) * 7);
            if (jacobians[1 + W_ + ei_[i]]) std::memcpy(jacobians[1 + W_ + ei_[i]] + size_t(i) * 7, J + 14, sizeof(double) * 7);
        }
        return true;
    }
private:
    Device &dev_;
    int W_, L_;
    std::vector<int32_t> type_, fi_, ei_;
    std::vector<double> pts_, coef_, s_;
    mutable bool staged_ = false;
    mutable std::vector<double> J_;
};

// ------------------------------------------------------------------ LidarTracker (estimator/src/lidarTracker/lidar_tracker.h)
// trackCloud(prev_cloud_feature, cur_cloud_feature, pose_ini) -> pose_prev_cur: scan-to-scan odometry of one LiDAR
// (lidar_tracker.cpp:23-129). The clouds' intensity carries the ring id, as ImageSegmenter leaves it.
class LidarTracker {
public:
    explicit LidarTracker(Device &dev) : dev_(dev) { mlh_track_opts_default(&opts_); }
    mlh_track_opts &options() { return opts_; }       // DISTANCE_SQ_THRESHOLD, NEARBY_SCAN (parameters.cpp:226-227)
    template <typename PoseT>
    PoseT trackCloud(const cloudFeature &prev_cloud_feature, const cloudFeature &cur_cloud_feature, const PoseT &pose_ini)
    {
        const PointICloud &corner_last = prev_cloud_feature.find("corner_points_less_sharp")->second;
        const PointICloud &surf_last = prev_cloud_feature.find("surf_points_less_flat")->second;
        const PointICloud &corner_sharp = cur_cloud_feature.find("corner_points_sharp")->second;
        const PointICloud &surf_flat = cur_cloud_feature.find("surf_points_flat")->second;
        const int io = point_traits<PointI>::intensity_off, sz = (int)sizeof(PointI);
        dev_.check(mlh_track_set_prev(dev_.ctx(), MLH_CORNER, corner_last.points.data(), sz, (int)corner_last.size(), io, MLH_MEM_HOST, opts_.distance_sq_threshold));
        dev_.check(mlh_track_set_prev(dev_.ctx(), MLH_SURF, surf_last.points.data(), sz, (int)surf_last.size(), io, MLH_MEM_HOST, opts_.distance_sq_threshold));
        dev_.check(mlh_track_set_cur(dev_.ctx(), MLH_CORNER, corner_sharp.points.data(), sz, (int)corner_sharp.size(), io, MLH_MEM_HOST));
        dev_.check(mlh_track_set_cur(dev_.ctx(), MLH_SURF, surf_flat.points.data(), sz, (int)surf_flat.size(), io, MLH_MEM_HOST));
        double p[7];
        detail::pose_to_param(pose_ini, p);
        dev_.check(mlh_track_cloud(dev_.ctx(), p, &opts_, nullptr));
        PoseT pose_prev_cur;
        detail::pose_from_param(pose_prev_cur, p);
        return pose_prev_cur;
    }
    // Device-resident hand-over (estimator.cpp:426-427, 532-543: copied before any undistortion): the scan FeatureExtract::extractCloudOnDevice
    // left on this Device becomes the current frame (sharp / flat) or, after tracking, the next call's previous frame (less sharp /
    // thinned less flat); trackCloudOnDevice is trackCloud on whatever was staged.
    void setCurFromExtractor() { dev_.check(mlh_track_set_from_scan(dev_.ctx(), 0, opts_.distance_sq_threshold)); }
    void setPrevFromExtractor() { dev_.check(mlh_track_set_from_scan(dev_.ctx(), 1, opts_.distance_sq_threshold)); }
    template <typename PoseT>
    PoseT trackCloudOnDevice(const PoseT &pose_ini)
    {
        double p[7];
        detail::pose_to_param(pose_ini, p);
        dev_.check(mlh_track_cloud(dev_.ctx(), p, &opts_, nullptr));
        PoseT pose_prev_cur;
        detail::pose_from_param(pose_prev_cur, p);
        return pose_prev_cur;
    }
private:
    Device &dev_;
    mlh_track_opts opts_;
};

// both kinds at once: the two fused clouds go through ONE thinning pipeline (half the dependent launches, one host round trip)
template <typename PoseVec>
inline std::pair<int, int> downsampleFusedScans(Device &dev, float leaf_surf, float leaf_corner, const PoseVec &pose_ext, bool with_ua_flag)
{
    std::vector<double> ext, ext_cov;
    detail::pack_extrinsics(pose_ext, ext, ext_cov);
    const void *cs = nullptr, *cc = nullptr;
    int32_t ns = 0, nc = 0, ms = 0, mc = 0;
    dev.check(mlh_fused_cloud(dev.ctx(), MLH_SURF, &cs, &ns));
    dev.check(mlh_fused_cloud(dev.ctx(), MLH_CORNER, &cc, &nc));
    if (ns <= 0 || nc <= 0) return {0, 0};
    dev.check(mlh_downsample_current_scan_pair(dev.ctx(), cs, ns, cc, nc, 16, 12, MLH_MEM_DEVICE, leaf_surf, leaf_corner, ext.data(), ext_cov.data(),
                                               (int)pose_ext.size(), params().COV_MEASUREMENT, with_ua_flag ? 1 : 0, params().TRACE_THRESHOLD_MAPPING, &ms, &mc));
    return {ms, mc};
}

// ------------------------------------------------------------------ the odometry's window map (estimator.cpp:1160-1203)
// pcl::transformPointCloud(cloud_in, cloud_out, pose.T_.cast<float>())
template <typename PoseT>
inline void transformPointCloud(Device &dev, const PointICloud &cloud_in, PointICloud &cloud_out, const PoseT &pose)
{
    cloud_out = cloud_in;
    if (cloud_out.size() == 0) return;
    double p[7];
    detail::pose_to_param(pose, p);
    dev.check(mlh_transform_point_cloud(dev.ctx(), cloud_out.points.data(), (int)sizeof(PointI), (int)cloud_out.size(), p, MLH_MEM_HOST));
}
// pcl::VoxelGrid<PointI> with the reference's call sequence: setLeafSize / setInputCloud / filter
class VoxelGrid {
public:
    explicit VoxelGrid(Device &dev) : dev_(dev) {}
    void setLeafSize(float lx, float ly, float lz) { (void)ly; (void)lz; leaf_ = lx; }     // the reference only ever sets cubic leaves
    void setInputCloud(const PointICloud &cloud) { input_ = &cloud; keep_.reset(); }
    template <typename CloudPtr>
    auto setInputCloud(const CloudPtr &cloud) -> decltype(void((*cloud).points)) { auto k = std::make_shared<CloudPtr>(cloud); input_ = &**k; keep_ = k; }
    void filter(PointICloud &output)
    {
        if (!input_ || input_->size() == 0) { output.points.clear(); return; }
        std::vector<PointI> out(input_->size());
        int32_t n_out = 0;
        dev_.check(mlh_voxel_grid(dev_.ctx(), input_->points.data(), (int)sizeof(PointI), (int)input_->size(), point_traits<PointI>::intensity_off, leaf_,
                                  out.data(), &n_out, MLH_MEM_HOST));
        out.resize(n_out);
        output.points.swap(out);
    }
private:
    Device &dev_;
    const PointICloud *input_ = nullptr;
    std::shared_ptr<void> keep_;
    float leaf_ = 0.4f;
};

// ------------------------------------------------------------------ undistortion (utility.h:79-100, estimator.cpp:376-410)
// TransformToEnd over a whole cloud (the reference loops `for (PointI &point : cloud) TransformToEnd(point, point, pose, true, SCAN_PERIOD)`)
template <typename PoseT>
inline void TransformToEnd(Device &dev, PointICloud &cloud, const PoseT &pose, const bool &b_distortion, const float &scan_period = 0.1f)
{
    if (cloud.size() == 0) return;
    double p[7];
    detail::pose_to_param(pose, p);
    dev.check(mlh_transform_to_end(dev.ctx(), cloud.points.data(), (int)sizeof(PointI), (int)cloud.size(), point_traits<PointI>::intensity_off, p,
                                   b_distortion ? 1 : 0, scan_period, MLH_MEM_HOST));
}
// the same for the scan FeatureExtract::extractCloudOnDevice left on the device (laser_cloud + the thinned less-flat cloud)
template <typename PoseT>
inline void undistortMeasurementsOnDevice(Device &dev, const PoseT &pose_undist, float scan_period = 0.1f)
{
    double p[7];
    detail::pose_to_param(pose_undist, p);
    dev.check(mlh_scan_undistort(dev.ctx(), p, scan_period));
}

// ------------------------------------------------------------------ the mapper's input clouds without a host hop
// transformCloudFeature (visualization.cpp:39-51) + the concatenation the mapper receives: fuseReset once per frame, then for every
// LiDAR extractCloudOnDevice + fuseCloudFeature(laser index, its extrinsic); downsampleFusedScan is downsampleCurrentScan on the
// fused cloud of `kind` (the result becomes the kind's feature set for scan2map) and returns the number of features kept.
inline void fuseReset(Device &dev) { dev.check(mlh_fuse_reset(dev.ctx())); }
template <typename PoseT>
inline void fuseCloudFeature(Device &dev, int laser_idx, const PoseT &pose_ext)
{
    double e[7];
    detail::pose_to_param(pose_ext, e);
    dev.check(mlh_fuse_add_scan(dev.ctx(), laser_idx, e));
}
// ... of the scan ANOTHER Device of the same GPU holds (a front-end lane's: FrontEndLanes, threadDevice()): device to device, no host hop; `src` must be idle
template <class PoseT>
inline void fuseCloudFeatureFrom(Device &dev, Device &src, int laser_idx, const PoseT &pose_ext)
{
    double e[7];
    detail::pose_to_param(pose_ext, e);
    dev.check(mlh_fuse_add_scan_from(dev.ctx(), src.ctx(), laser_idx, e));
}
template <typename PoseVec>
inline int downsampleFusedScan(Device &dev, int kind, float leaf, const PoseVec &pose_ext, bool with_ua_flag)
{
    std::vector<double> ext, ext_cov;
    detail::pack_extrinsics(pose_ext, ext, ext_cov);
    const void *cloud = nullptr;
    int32_t n = 0, m = 0;
    dev.check(mlh_fused_cloud(dev.ctx(), kind, &cloud, &n));
    if (n <= 0) return 0;
    dev.check(mlh_downsample_current_scan(dev.ctx(), kind, cloud, 16, n, 12, MLH_MEM_DEVICE, leaf, ext.data(), ext_cov.data(), (int)pose_ext.size(),
                                          params().COV_MEASUREMENT, with_ua_flag ? 1 : 0, params().TRACE_THRESHOLD_MAPPING, nullptr, &m));
    return m;
}

// ------------------------------------------------------------------ ActiveFeatureSelection::evalFullHessian + the gf_ratio policy
// evalFullHessian (lidar_mapper.h:176-227): match ALL features of a kind at pose_local and add their un-corrected, uncertainty-weighted
// J^T J to mat_H; feat_num += matched features. The map index of `kind` and its features must be staged (MapIndex::setInputCloud,
// mlh_features_set / downsampleCurrentScan).
template <typename PoseT>
inline void evalFullHessian(Device &dev, int kind, const PoseT &pose_local, double mat_H[36], int &feat_num)
{
    double p[7], JtJ[36];
    detail::pose_to_param(pose_local, p);
    int32_t n_valid = 0;
    dev.check(mlh_match_linearize(dev.ctx(), kind, p, 5, MLH_FLAG_WITH_UA | MLH_FLAG_NO_LOSS, params().MIN_MATCH_SQ_DIS, params().MIN_PLANE_DIS, 0.0,
                                  params().COV_MEASUREMENT_TRACE, nullptr, nullptr, nullptr, nullptr, JtJ, nullptr, nullptr, &n_valid));
    for (int i = 0; i < 36; ++i) mat_H[i] += JtJ[i];
    feat_num += n_valid;
}

// common::logDet(M, true) of a 6x6 (mloam_common/libs/include/common/algos/math.hpp:173-187): 2 * sum(log(diag(chol(M))))
inline double logDet6(const double M[36])
{
    double L[36] = {0}, ld = 0.0;
    for (int j = 0; j < 6; ++j) {
        double s = M[j * 6 + j];
        for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
        if (!(s > 0.0)) return std::nan("");
        const double ljj = std::sqrt(s);
        L[j * 6 + j] = ljj;
        ld += std::log(ljj);
        for (int i = j + 1; i < 6; ++i) {
            double t = M[i * 6 + j];
            for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = t / ljj;
        }
    }
    return 2.0 * ld;
}

// the every-10th-frame policy of scan2MapOptimization (lidar_mapper_keyframe.cpp:456-494): returns the new gf_ratio_cur. gf_ratio_cur (in): the value so far -- it
// stays when no branch of the reference's chain is taken (a method name it does not know; gd_float with a NaN factor: neither `>` nor `<=` holds)
inline double gfRatioPolicy(const std::string &gf_method, double gf_ratio_ini, double gf_deg_factor, double MAP_DEG_THRE, double gf_ratio_cur = 1.0)
{
    if (gf_method == "wo_gf") return 1.0;
    if (gf_method == "rnd" || gf_method == "fps" || gf_method == "gd_fix") return gf_ratio_ini;
    if (gf_method == "gd_float") {
        if (gf_deg_factor > MAP_DEG_THRE) return gf_ratio_ini;
        if (gf_deg_factor <= MAP_DEG_THRE) return 0.8;
    }
    return gf_ratio_cur;
}

// ------------------------------------------------------------------ ImageSegmenter (estimator/src/imageSegmenter/image_segmenter.hpp:36-83)
// setParameter / segmentCloud with the reference's argument lists; the segmented, ring-major cloud ALSO stays staged on the device as
// the Device's scan, so FeatureExtract::extractCloudOnDevice-style calls (mlh_extract_run) can follow without an upload.
class ImageSegmenter {
public:
    // as the reference's member `ImageSegmenter img_segment_;` (estimator.h:189): re-entrant on one object, every calling thread on its own context
    ImageSegmenter() : bound_(nullptr) { mlh_segment_params_default(&prm_); }
    explicit ImageSegmenter(Device &dev) : bound_(&dev) { mlh_segment_params_default(&prm_); }
    void setParameter(const int &vertical_scans, const int &horizon_scans, const int &min_cluster_size, const int &segment_valid_point_num,
                      const int &segment_valid_line_num)
    {
        prm_.vertical_scans = vertical_scans; prm_.horizon_scans = horizon_scans; prm_.min_cluster_size = min_cluster_size;
        prm_.segment_valid_point_num = segment_valid_point_num; prm_.segment_valid_line_num = segment_valid_line_num;
    }
    float &SEGMENT_THETA() { return prm_.segment_theta; }        // parameters.h globals the reference's segmentCloud reads
    double &ROI_RANGE() { return prm_.roi_range; }
    template <typename ScanInfoT>
    void segmentCloud(const PointICloud &laser_cloud_in, PointICloud &laser_cloud_out, PointICloud &laser_cloud_outlier, ScanInfoT &scan_info)
    {
        Device &dev_ = bound_ ? *bound_ : threadDevice();
        mlh_segment_params prm_ = this->prm_;          // the call's own copy: concurrent calls differ in segment_flag
        const int n = (int)laser_cloud_in.size();
        prm_.segment_flag = scan_info.segment_flag_ ? 1 : 0;
        // laser_cloud_outlier holds at most one point per pixel of a column that is a multiple of 5 (and never more than n), plus one
        const int outl_cap = std::min(n, prm_.vertical_scans * ((prm_.horizon_scans + 4) / 5)) + 1;
        std::vector<float> &out = detail::thread_floats(0, size_t(n > 0 ? n : 1) * 4), &outl = detail::thread_floats(1, size_t(outl_cap) * 4);
        int32_t n_out = 0, n_outl = 0;
        scan_info.scan_start_ind_.resize(prm_.vertical_scans);
        scan_info.scan_end_ind_.resize(prm_.vertical_scans);
        dev_.check(mlh_segment_cloud(dev_.ctx(), laser_cloud_in.points.data(), (int)sizeof(PointI), point_traits<PointI>::intensity_off, n, MLH_MEM_HOST, &prm_,
                                     out.data(), &n_out, scan_info.scan_start_ind_.data(), scan_info.scan_end_ind_.data(), outl.data(), outl_cap, &n_outl));
        if (n_outl > outl_cap) n_outl = outl_cap;      // cannot happen with the bound above; never read past the buffer
        auto fill = [](PointICloud &c, const std::vector<float> &v, int m) {
            detail::cloud_resize(c, size_t(m));
            for (int k = 0; k < m; ++k) { PointI p; p.x = v[4 * k]; p.y = v[4 * k + 1]; p.z = v[4 * k + 2]; p.intensity = v[4 * k + 3]; c.points[size_t(k)] = p; }
        };
        fill(laser_cloud_out, out, n_out);
        fill(laser_cloud_outlier, outl, n_outl);
    }
    // The same call with nothing fetched: scan_info gets its ring table, the segmented ring-major cloud stays on the Device as its scan -- for
    // FeatureExtract::extractStagedCloudOnDevice and fuseCloudFeature / fuseCloudFeatureFrom behind it (INTEGRATION 4b': the frame from raw clouds without a host hop)
    template <typename ScanInfoT>
    void segmentCloudOnDevice(const PointICloud &laser_cloud_in, ScanInfoT &scan_info)
    {
        Device &dev_ = bound_ ? *bound_ : threadDevice();
        mlh_segment_params prm_ = this->prm_;
        prm_.segment_flag = scan_info.segment_flag_ ? 1 : 0;
        int32_t n_out = 0, n_outl = 0;
        scan_info.scan_start_ind_.resize(prm_.vertical_scans);
        scan_info.scan_end_ind_.resize(prm_.vertical_scans);
        dev_.check(mlh_segment_cloud(dev_.ctx(), laser_cloud_in.points.data(), (int)sizeof(PointI), point_traits<PointI>::intensity_off, (int)laser_cloud_in.size(),
                                     MLH_MEM_HOST, &prm_, nullptr, &n_out, scan_info.scan_start_ind_.data(), scan_info.scan_end_ind_.data(), nullptr, 0, &n_outl));
    }
private:
    Device *bound_;
    mlh_segment_params prm_;
};

// ------------------------------------------------------------------ the front end's lanes: the reference's OpenMP team for callers that have none
// estimator.cpp:249 runs calTimestamp / segmentCloud / extractCloud of the NUM_OF_LASER LiDARs on NUM_OF_LASER OpenMP threads. Most of segmentCloud is its cluster
// search -- sequential by definition (csrc/segment.hip), on the host, ~1.3 ms of a 1.6 ms call on a 64-ring scan -- so what can run beside one LiDAR's search is the
// NEXT LiDAR's search and the device work around it. A caller built without OpenMP (or calling from one thread) gets that from here: FrontEndLanes keeps one
// persistent worker thread per LiDAR; each worker runs the loop body of estimator.cpp:252-262 for its LiDAR on the default-constructed (re-entrant) facade objects,
// i.e. on the context threadDevice() keeps for that worker -- created on the worker's first job, reused for every later frame. The results are the same calls'
// results (facade_selftest: equal to the one-after-the-other run, bit for bit).
class FrontEndLanes {
public:
    explicit FrontEndLanes(int n_lanes) : lanes_(size_t(n_lanes > 0 ? n_lanes : 1))
    {
        for (auto &l : lanes_) { l.reset(new Lane); Lane *lp = l.get(); lp->th = std::thread([lp] { lp->run(); }); }
    }
    ~FrontEndLanes()
    {
        for (auto &l : lanes_) { { std::lock_guard<std::mutex> g(l->mu); l->quit = true; } l->cv.notify_all(); }
        for (auto &l : lanes_) if (l->th.joinable()) l->th.join();
    }
    FrontEndLanes(const FrontEndLanes &) = delete;
    FrontEndLanes &operator=(const FrontEndLanes &) = delete;
    int size() const { return int(lanes_.size()); }
    // hands `job` to lane i's worker (behind the job it may still be running) and returns; everything the job references must stay alive until wait(i)
    void post(int i, std::function<void()> job)
    {
        Lane &l = *lanes_.at(size_t(i));
        std::unique_lock<std::mutex> g(l.mu);
        l.cv.wait(g, [&] { return !l.busy; });
        l.job = std::move(job); l.busy = true; l.err = nullptr;
        g.unlock();
        l.cv.notify_all();
    }
    // returns when lane i's job has run; what the job threw is thrown again here
    void wait(int i)
    {
        Lane &l = *lanes_.at(size_t(i));
        std::unique_lock<std::mutex> g(l.mu);
        l.cv.wait(g, [&] { return !l.busy; });
        std::exception_ptr e = l.err;
        l.err = nullptr;
        g.unlock();
        if (e) std::rethrow_exception(e);
    }
    // estimator.cpp:248-263 from ONE calling thread: every LiDAR's calTimestamp -> segmentCloud -> extractCloud (+ "laser_cloud_outlier") on its own lane, all lanes
    // at once; returns when all are done. img_segment / f_extract: the default-constructed (context-less) objects, as the reference's members are.
    // segment_flag: ScanInfo's (SEGMENT_CLOUD, false while the extrinsics are being estimated: estimator.cpp:258).
    template <class RawCloudVec, class SegmenterT, class ExtractT, class FeatureVec>
    void processAllLasers(SegmenterT &img_segment, ExtractT &f_extract, const RawCloudVec &v_laser_cloud_in, int n_scans, bool segment_flag, FeatureVec &feature_frame)
    {
        const size_t n = v_laser_cloud_in.size();
        if (feature_frame.size() < n) feature_frame.resize(n);
        for (size_t i0 = 0; i0 < n; i0 += lanes_.size()) {
            const size_t i1 = std::min(n, i0 + lanes_.size());
            for (size_t i = i0; i < i1; ++i)
                post(int(i - i0), [&img_segment, &f_extract, &v_laser_cloud_in, &feature_frame, n_scans, segment_flag, i] {
                    PointICloud laser_cloud, laser_cloud_segment, laser_cloud_outlier;
                    f_extract.calTimestamp(v_laser_cloud_in[i], laser_cloud);
                    ScanInfo scan_info(n_scans, segment_flag);
                    img_segment.segmentCloud(laser_cloud, laser_cloud_segment, laser_cloud_outlier, scan_info);
                    feature_frame[i].clear();
                    f_extract.extractCloud(laser_cloud_segment, scan_info, feature_frame[i]);
                    feature_frame[i].insert(std::pair<std::string, PointICloud>("laser_cloud_outlier", laser_cloud_outlier));
                });
            std::exception_ptr first;
            for (size_t i = i0; i < i1; ++i) { try { wait(int(i - i0)); } catch (...) { if (!first) first = std::current_exception(); } }
            if (first) std::rethrow_exception(first);
        }
    }
    // the context lane i's worker keeps (threadDevice() of that thread; created on first use): what fuseCloudFeatureFrom reads from. Valid while the lanes live.
    Device &laneDevice(int i)
    {
        Lane &l = *lanes_.at(size_t(i));
        if (!l.dev) {
            Device **slot = &l.dev;
            post(i, [slot] { *slot = &threadDevice(); });
            wait(i);
        }
        return *l.dev;
    }
    // The frame's front end WITHOUT a host hop (INTEGRATION 4b'): every LiDAR's calTimestamp -> segmentCloudOnDevice -> extractStagedCloudOnDevice on its lane's
    // context, then `gather` collects the lanes' mapping features device to device (fuseReset + fuseCloudFeatureFrom: transformCloudFeature with pose_ext[i], LiDAR
    // index i). Returns when the appends are ENQUEUED: the lanes are free for the next frame (what rewrites their scans waits for the appends on the device), and
    // downsampleFusedScans / scan2MapOptimization on `gather` follow.
    template <class RawCloudVec, class SegmenterT, class ExtractT, class PoseVecT>
    void processAllLasersOnDevice(SegmenterT &img_segment, ExtractT &f_extract, const RawCloudVec &v_laser_cloud_in, int n_scans, bool segment_flag, Device &gather,
                                  const PoseVecT &pose_ext)
    {
        const size_t n = v_laser_cloud_in.size();
        fuseReset(gather);
        for (size_t i0 = 0; i0 < n; i0 += lanes_.size()) {
            const size_t i1 = std::min(n, i0 + lanes_.size());
            for (size_t i = i0; i < i1; ++i)
                post(int(i - i0), [&img_segment, &f_extract, &v_laser_cloud_in, n_scans, segment_flag, i] {
                    PointICloud laser_cloud;
                    f_extract.calTimestamp(v_laser_cloud_in[i], laser_cloud);
                    ScanInfo scan_info(n_scans, segment_flag);
                    img_segment.segmentCloudOnDevice(laser_cloud, scan_info);
                    f_extract.extractStagedCloudOnDevice();
                });
            std::exception_ptr first;
            for (size_t i = i0; i < i1; ++i) { try { wait(int(i - i0)); } catch (...) { if (!first) first = std::current_exception(); } }
            if (first) std::rethrow_exception(first);
            for (size_t i = i0; i < i1; ++i) fuseCloudFeatureFrom(gather, laneDevice(int(i - i0)), int(i), pose_ext[i]);
        }
    }
private:
    struct Lane {
        Device *dev = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<void()> job;
        std::exception_ptr err;
        bool busy = false, quit = false;
        void run()
        {
            std::unique_lock<std::mutex> g(mu);
            for (;;) {
                cv.wait(g, [&] { return busy || quit; });
                if (!busy && quit) return;
                std::function<void()> j = std::move(job);
                g.unlock();
                std::exception_ptr e;
                try { j(); } catch (...) { e = std::current_exception(); }
                j = nullptr;
                g.lock();
                err = e; busy = false;
                cv.notify_all();
            }
        }
    };
    std::vector<std::unique_ptr<Lane>> lanes_;
};

// ------------------------------------------------------------------ the per-feature Ceres contract (lidar_map_factor.hpp:26-71, 130-174)
// LidarMapPlaneNormFactor / LidarMapEdgeFactor with the reference's constructor (point, coeff, cov_matrix) and
// SizedCostFunction<1, 7>::Evaluate(parameters, residuals, jacobians) -- so the reference's block-assembly loop
// (lidar_mapper_keyframe.cpp:537-571: `new LidarMapPlaneNormFactor(feature.point_, feature.coeffs_, cov_matrix)`, AddResidualBlock) compiles
// against this header unchanged. A residual block that Ceres evaluates one at a time is a HOST call by contract (one virtual call per
// residual per iteration): these classes hold the closed form for that case. The accelerated path does not use them -- it evaluates all
// blocks of a kind in one launch (LidarMapBatchFactor / mlh_linearize) or keeps the whole solve on the device (mlh_scan2map).
namespace detail {
inline void quat_rot(const double q[4] /*x y z w*/, const double v[3], double out[3])
{
    const double ux = q[1] * v[2] - q[2] * v[1], uy = q[2] * v[0] - q[0] * v[2], uz = q[0] * v[1] - q[1] * v[0];
    const double tx = ux + ux, ty = uy + uy, tz = uz + uz;
    out[0] = v[0] + q[3] * tx + (q[1] * tz - q[2] * ty);
    out[1] = v[1] + q[3] * ty + (q[2] * tx - q[0] * tz);
    out[2] = v[2] + q[3] * tz + (q[0] * ty - q[1] * tx);
}
inline void quat_to_rot(const double q[4], double R[9])
{
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3], txx = tx * q[0], txy = ty * q[0], txz = tz * q[0], tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
// lidar_map_factor.hpp:36-40: sqrt(1 / trace(cov_matrix)), >= 3 -> 1, else / 3. cov_matrix: Eigen::Matrix3d or a row-major 3 x 3 array
template <class M3> inline double sqrt_info_of(const M3 &cov) { const double s = std::sqrt(1 / (mat_at(cov, 0, 0, 3, 0) + mat_at(cov, 1, 1, 3, 0) + mat_at(cov, 2, 2, 3, 0))); return s >= 3.0 ? 1.0 : s / 3.0; }
static const std::array<double, 9> kIdentity3 = {{1, 0, 0, 0, 1, 0, 0, 0, 1}};
template <class V3> inline std::array<double, 3> vec3_of(const V3 &v) { return {{double(v[0]), double(v[1]), double(v[2])}}; }
template <class VX> inline std::vector<double> vecx_of(const VX &v) { std::vector<double> o(size_t(v.size())); for (size_t i = 0; i < o.size(); ++i) o[i] = double(v[i]); return o; }
// the reference's check() (lidar_map_factor.hpp:98-118, 204-227): analytic against central-difference Jacobian of one factor, printed. F: any class with the
// Ceres Evaluate contract over one 7-parameter block; the numeric columns perturb the pose through PoseLocalParameterization's Plus (identity projector)
template <class F> inline void check_factor(const F &f, double **param, const char *name)
{
    double r0 = 0.0, J[7] = {0, 0, 0, 0, 0, 0, 0};
    double *Jp[1] = {J};
    f.Evaluate(param, &r0, Jp);
    double num[6];
    double V[36] = {0};
    for (int i = 0; i < 6; ++i) V[i * 7] = 1.0;
    const double eps = 1e-6;
    for (int k = 0; k < 6; ++k) {
        double d[6] = {0, 0, 0, 0, 0, 0}, xp[7], xm[7], rp = 0.0, rm = 0.0;
        d[k] = eps; mlh_pose_plus(param[0], d, V, xp);
        d[k] = -eps; mlh_pose_plus(param[0], d, V, xm);
        const double *pp[1] = {xp}, *pm[1] = {xm};
        f.Evaluate(pp, &rp, nullptr); f.Evaluate(pm, &rm, nullptr);
        num[k] = (rp - rm) / (2 * eps);
    }
    std::printf("%s check: residual %.9f\n  analytic", name, r0);
    for (int k = 0; k < 6; ++k) std::printf(" %.6f", J[k]);
    std::printf("\n  numeric ");
    for (int k = 0; k < 6; ++k) std::printf(" %.6f", num[k]);
    std::printf("\n");
}
// row (1x3) times [p]x
inline void row_skew(const double a[3], const double p[3], double out[3]) { out[0] = a[1] * p[2] - a[2] * p[1]; out[1] = a[2] * p[0] - a[0] * p[2]; out[2] = a[0] * p[1] - a[1] * p[0]; }
}  // namespace detail

class LidarMapPlaneNormFactor MLOAM_FACADE_COST_BASE(1, 7) {      // lidar_map_factor.hpp:26-126
public:
    // (point, coeff, cov_matrix) as lidar_map_factor.hpp:28-30: Eigen::Vector3d / Eigen::VectorXd / Eigen::Matrix3d, or std::array / std::vector / row-major array
    template <class V3, class VX, class M3>
    LidarMapPlaneNormFactor(const V3 &point, const VX &coeff, const M3 &cov_matrix)
        : point_(detail::vec3_of(point)), coeff_(detail::vecx_of(coeff)), sqrt_info_(detail::sqrt_info_of(cov_matrix)) {}
    template <class V3, class VX>
    LidarMapPlaneNormFactor(const V3 &point, const VX &coeff) : point_(detail::vec3_of(point)), coeff_(detail::vecx_of(coeff)), sqrt_info_(detail::sqrt_info_of(detail::kIdentity3)) {}
    MLOAM_FACADE_VIRTUAL ~LidarMapPlaneNormFactor() {}
    void check(double **param) { detail::check_factor(*this, param, "LidarMapPlaneNormFactor"); }
    MLOAM_FACADE_VIRTUAL bool Evaluate(double const *const *param, double *residuals, double **jacobians) const
    {
        const double *x = param[0];
        double lp[3];
        detail::quat_rot(x + 3, point_.data(), lp);
        const double w[3] = {coeff_[0], coeff_[1], coeff_[2]};
        const double a = (w[0] * (lp[0] + x[0]) + w[1] * (lp[1] + x[1]) + w[2] * (lp[2] + x[2])) + coeff_[3];
        residuals[0] = sqrt_info_ * a;
        if (jacobians && jacobians[0]) {
            double R[9], wr[3], rot[3];
            detail::quat_to_rot(x + 3, R);
            for (int c = 0; c < 3; ++c) wr[c] = -(w[0] * R[c] + w[1] * R[3 + c] + w[2] * R[6 + c]);      // -w^T R
            detail::row_skew(wr, point_.data(), rot);                                                  // ... [p]x
            double *J = jacobians[0];
            for (int c = 0; c < 3; ++c) { J[c] = sqrt_info_ * w[c]; J[3 + c] = sqrt_info_ * rot[c]; }
            J[6] = 0.0;
        }
        return true;
    }
private:
    const std::array<double, 3> point_;
    const std::vector<double> coeff_;
    double sqrt_info_;
};

class LidarMapEdgeFactor MLOAM_FACADE_COST_BASE(1, 7) {           // lidar_map_factor.hpp:130-235
public:
    template <class V3, class VX, class M3>
    LidarMapEdgeFactor(const V3 &point, const VX &coeff, const M3 &cov_matrix)
        : point_(detail::vec3_of(point)), coeff_(detail::vecx_of(coeff)), sqrt_info_(detail::sqrt_info_of(cov_matrix)) {}
    template <class V3, class VX>
    LidarMapEdgeFactor(const V3 &point, const VX &coeff) : point_(detail::vec3_of(point)), coeff_(detail::vecx_of(coeff)), sqrt_info_(detail::sqrt_info_of(detail::kIdentity3)) {}
    MLOAM_FACADE_VIRTUAL ~LidarMapEdgeFactor() {}
    void check(double **param) { detail::check_factor(*this, param, "LidarMapEdgeFactor"); }      // lidar_map_factor.hpp:176-229; named at lidar_mapper_keyframe.cpp:568
    MLOAM_FACADE_VIRTUAL bool Evaluate(double const *const *param, double *residuals, double **jacobians) const
    {
        const double *x = param[0];
        double lp[3];
        detail::quat_rot(x + 3, point_.data(), lp);
        for (int c = 0; c < 3; ++c) lp[c] += x[c];
        const double a[3] = {lp[0] - coeff_[0], lp[1] - coeff_[1], lp[2] - coeff_[2]}, b[3] = {lp[0] - coeff_[3], lp[1] - coeff_[4], lp[2] - coeff_[5]};
        const double nu[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
        const double de[3] = {coeff_[0] - coeff_[3], coeff_[1] - coeff_[4], coeff_[2] - coeff_[5]};
        const double nu_n = std::sqrt(nu[0] * nu[0] + nu[1] * nu[1] + nu[2] * nu[2]), de_n = std::sqrt(de[0] * de[0] + de[1] * de[1] + de[2] * de[2]);
        residuals[0] = sqrt_info_ * nu_n / de_n;
        if (jacobians && jacobians[0]) {
            double eta[3] = {nu[0], nu[1], nu[2]};
            if (nu_n > 0.0) for (int c = 0; c < 3; ++c) eta[c] /= nu_n;                      // Eigen normalized(): a zero vector stays zero
            for (int c = 0; c < 3; ++c) eta[c] *= 1.0 / de_n;
            double eD[3], R[9], eDR[3], rot[3];
            detail::row_skew(eta, de, eD);                                                   // eta [lpa - lpb]x
            detail::quat_to_rot(x + 3, R);
            for (int c = 0; c < 3; ++c) eDR[c] = eD[0] * R[c] + eD[1] * R[3 + c] + eD[2] * R[6 + c];
            detail::row_skew(eDR, point_.data(), rot);
            double *J = jacobians[0];
            for (int c = 0; c < 3; ++c) { J[c] = sqrt_info_ * (-eD[c]); J[3 + c] = sqrt_info_ * rot[c]; }
            J[6] = 0.0;
        }
        return true;
    }
private:
    const std::array<double, 3> point_;
    const std::vector<double> coeff_;
    double sqrt_info_;
};

// ------------------------------------------------------------------ the odometry window's and the calibration's per-factor classes
// For callers that keep Estimator::optimizeMap's AddResidualBlock loops as they are (estimator.cpp:733-813): LidarPureOdom{PlaneNorm,Edge}Factor
// (lidar_pure_odom_factor.hpp:27-102, 198-282; ceres::SizedCostFunction<1, 7, 7, 7> over [pivot, frame, extrinsic]) and LidarOnlineCalib{PlaneNorm,Edge}Factor
// (lidar_online_calib_factor.hpp:24-62, 125-165; <1, 7> over the extrinsic, the weight handed in). Host code, one factor per call, the arithmetic of the batched
// device kernel (csrc/odom.hip: odom_factor_core) term by term -- including the two Jacobian columns of the reference that are not exact derivatives.
namespace detail {
inline void qmul4(const double a[4], const double b[4], double o[4])      // [x y z w]
{
    o[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
    o[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
    o[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
    o[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
}
inline void matv(const double M[9], const double v[3], double o[3]) { for (int r = 0; r < 3; ++r) o[r] = M[3 * r] * v[0] + M[3 * r + 1] * v[1] + M[3 * r + 2] * v[2]; }
inline void tmatv(const double M[9], const double v[3], double o[3]) { for (int c = 0; c < 3; ++c) o[c] = M[c] * v[0] + M[3 + c] * v[1] + M[6 + c] * v[2]; }
inline void rowm(const double a[3], const double M[9], double o[3]) { for (int c = 0; c < 3; ++c) o[c] = a[0] * M[c] + a[1] * M[3 + c] + a[2] * M[6 + c]; }
// type 0: plane (coeff = n, d), 1: edge (coeff = the two line points); J: three 1 x 7 rows [pivot | frame | extrinsic], any may be null
inline void pure_odom_factor(int type, const double p[3], const double *coeff, double s, const double *pp, const double *pi, const double *pe, double &r_out,
                             double *J0, double *J1, double *J2)
{
    const double Qpc[4] = {-pp[3], -pp[4], -pp[5], pp[6]};
    double Qpi[4], Qx[4], tpi[3], rte[3], rp[3];
    qmul4(Qpc, pi + 3, Qpi);
    const double dt[3] = {pi[0] - pp[0], pi[1] - pp[1], pi[2] - pp[2]};
    quat_rot(Qpc, dt, tpi);
    qmul4(Qpi, pe + 3, Qx);
    quat_rot(Qpi, pe, rte);
    quat_rot(Qx, p, rp);
    const double lp[3] = {rp[0] + (rte[0] + tpi[0]), rp[1] + (rte[1] + tpi[1]), rp[2] + (rte[2] + tpi[2])};
    double Rp[9], Ri[9], Re[9], Rep[3], Rite[3], RiRep[3];
    quat_to_rot(pp + 3, Rp); quat_to_rot(pi + 3, Ri); quat_to_rot(pe + 3, Re);
    matv(Re, p, Rep); matv(Ri, pe, Rite); matv(Ri, Rep, RiRep);
    const double v[3] = {RiRep[0] + Rite[0] + pi[0] - pp[0], RiRep[1] + Rite[1] + pi[1] - pp[1], RiRep[2] + Rite[2] + pi[2] - pp[2]};
    double a[3], res, dab[3] = {0, 0, 0};
    if (type == 0) {
        a[0] = coeff[0]; a[1] = coeff[1]; a[2] = coeff[2];
        res = (a[0] * lp[0] + a[1] * lp[1] + a[2] * lp[2]) + coeff[3];
    } else {
        const double ba[3] = {lp[0] - coeff[0], lp[1] - coeff[1], lp[2] - coeff[2]}, bb[3] = {lp[0] - coeff[3], lp[1] - coeff[4], lp[2] - coeff[5]};
        const double nu[3] = {ba[1] * bb[2] - ba[2] * bb[1], ba[2] * bb[0] - ba[0] * bb[2], ba[0] * bb[1] - ba[1] * bb[0]};
        const double de[3] = {coeff[0] - coeff[3], coeff[1] - coeff[4], coeff[2] - coeff[5]};
        const double n2 = nu[0] * nu[0] + nu[1] * nu[1] + nu[2] * nu[2];
        const double nu_n = std::sqrt(n2), de_n = std::sqrt(de[0] * de[0] + de[1] * de[1] + de[2] * de[2]);
        res = nu_n / de_n;
        double nh[3] = {nu[0], nu[1], nu[2]};
        if (n2 > 0.0) { const double nn = std::sqrt(n2); nh[0] = nu[0] / nn; nh[1] = nu[1] / nn; nh[2] = nu[2] / nn; }      // Eigen normalized(): zero stays zero
        const double k = 1.0 / de_n;
        const double eta[3] = {k * nh[0], k * nh[1], k * nh[2]};
        dab[0] = ba[0] - bb[0]; dab[1] = ba[1] - bb[1]; dab[2] = ba[2] - bb[2];
        row_skew(eta, dab, a);
    }
    r_out = s * res;
    double row0[3], row1[3], rot[3], t1[3], t2[3], tmp[3];
    matv(Rp, a, row0);                                                                     // a^T Rp^T
    if (J0) {
        if (type == 0) row_skew(row0, v, rot);
        else { tmatv(Rp, v, tmp); row_skew(a, tmp, rot); }
        for (int c = 0; c < 3; ++c) { J0[c] = s * (-row0[c]); J0[3 + c] = s * rot[c]; }
        J0[6] = 0.0;
    }
    rowm(row0, Ri, row1);
    if (J1) {
        const double q[3] = {Rep[0] + pe[0], Rep[1] + pe[1], Rep[2] + pe[2]};
        row_skew(row1, q, rot);
        for (int c = 0; c < 3; ++c) { J1[c] = s * row0[c]; J1[3 + c] = s * (-rot[c]); }
        J1[6] = 0.0;
    }
    if (J2) {
        if (type == 0) row_skew(row1, Rep, rot);
        else { rowm(row1, Re, tmp); row_skew(tmp, p, t1); row_skew(row1, pe, t2); rot[0] = t1[0] + t2[0]; rot[1] = t1[1] + t2[1]; rot[2] = t1[2] + t2[2]; }
        for (int c = 0; c < 3; ++c) { J2[c] = s * row1[c]; J2[3 + c] = s * (-rot[c]); }
        J2[6] = 0.0;
    }
}
}  // namespace detail

template <int TYPE>
class LidarPureOdomFactorT MLOAM_FACADE_COST_BASE(1, 7, 7, 7) {          // lidar_pure_odom_factor.hpp:27-102, 198-282
public:
    template <class V3, class VX>
    LidarPureOdomFactorT(const V3 &point, const VX &coeff, const double &s = 1.0) : point_(detail::vec3_of(point)), coeff_(detail::vecx_of(coeff)), s_(s) {}
    MLOAM_FACADE_VIRTUAL ~LidarPureOdomFactorT() {}
    MLOAM_FACADE_VIRTUAL bool Evaluate(double const *const *param, double *residuals, double **jacobians) const
    {
        detail::pure_odom_factor(TYPE, point_.data(), coeff_.data(), s_, param[0], param[1], param[2], residuals[0], jacobians ? jacobians[0] : nullptr,
                                 jacobians ? jacobians[1] : nullptr, jacobians ? jacobians[2] : nullptr);
        return true;
    }
private:
    const std::array<double, 3> point_;
    const std::vector<double> coeff_;
    const double s_;
};
typedef LidarPureOdomFactorT<0> LidarPureOdomPlaneNormFactor;      // lidar_pure_odom_factor.hpp:27-102
typedef LidarPureOdomFactorT<1> LidarPureOdomEdgeFactor;           // lidar_pure_odom_factor.hpp:198-282

// the map factors' form on the extrinsic alone, the weight handed in (1.0 at estimator.cpp:757, 813): residual = sqrt_info * (the map factor's unweighted residual)
namespace detail { static const std::array<double, 9> kUnitWeightCov = {{0.001, 0, 0, 0, 0.001, 0, 0, 0, 0.001}}; }      // trace 0.003 -> sqrt(1 / trace) = 18 >= 3 -> the map factor's weight is exactly 1
class LidarOnlineCalibPlaneNormFactor MLOAM_FACADE_COST_BASE(1, 7) {      // lidar_online_calib_factor.hpp:24-62
public:
    template <class V3, class VX>
    LidarOnlineCalibPlaneNormFactor(const V3 &point, const VX &coeff, const double &sqrt_info = 1.0) : unit_(point, coeff, detail::kUnitWeightCov), s_(sqrt_info) {}
    MLOAM_FACADE_VIRTUAL ~LidarOnlineCalibPlaneNormFactor() {}
    MLOAM_FACADE_VIRTUAL bool Evaluate(double const *const *param, double *residuals, double **jacobians) const
    {
        unit_.Evaluate(param, residuals, jacobians);
        residuals[0] *= s_;
        if (jacobians && jacobians[0]) for (int c = 0; c < 6; ++c) jacobians[0][c] *= s_;
        return true;
    }
private:
    LidarMapPlaneNormFactor unit_;
    double s_;
};
class LidarOnlineCalibEdgeFactor MLOAM_FACADE_COST_BASE(1, 7) {           // lidar_online_calib_factor.hpp:125-165
public:
    template <class V3, class VX>
    LidarOnlineCalibEdgeFactor(const V3 &point, const VX &coeff, const double &sqrt_info = 1.0) : unit_(point, coeff, detail::kUnitWeightCov), s_(sqrt_info) {}
    MLOAM_FACADE_VIRTUAL ~LidarOnlineCalibEdgeFactor() {}
    MLOAM_FACADE_VIRTUAL bool Evaluate(double const *const *param, double *residuals, double **jacobians) const
    {
        unit_.Evaluate(param, residuals, jacobians);
        residuals[0] *= s_;
        if (jacobians && jacobians[0]) for (int c = 0; c < 6; ++c) jacobians[0][c] *= s_;
        return true;
    }
private:
    LidarMapEdgeFactor unit_;
    double s_;
};

// ------------------------------------------------------------------ ActiveFeatureSelection (lidar_mapper.h:126-631)
// evalFullHessian / goodFeatureMatching with the reference's argument lists. The kd-tree argument is the MapIndex the caller built with
// setInputCloud(laser_map); laser_cloud is the feature cloud (PointIWithCov: the cov_vec weights the rows when WITH_UA is on).
class ActiveFeatureSelection {
public:
    explicit ActiveFeatureSelection(Device &dev, bool with_ua = true, uint64_t seed = 0) : dev_(dev), with_ua_(with_ua), seed_(seed) {}
    // lidar_mapper.h:176-227: match ALL features, mat_H += J^T J of the matched ones (weighted, not loss-corrected), feat_num += matches
    // (kdtree: the MapIndex or a shared_ptr to it; pose / 6 x 6 matrix: the stand-ins or the reference's Pose / Eigen::Matrix<double, 6, 6>)
    template <typename KdTreeT, typename PoseT, typename Mat6>
    void evalFullHessian(const KdTreeT &kdtree_from_map_, const PointICovCloud &laser_map, const PointICovCloud &laser_cloud, const PoseT &pose_local,
                         const char feature_type, Mat6 &mat_H, int &feat_num)
    {
        (void)laser_map;
        const MapIndex<PointIWithCov> &kdtree_from_map = detail::deref(kdtree_from_map_, 0);
        const int kind = feature_type == 's' ? MLH_SURF : MLH_CORNER;
        if (kdtree_from_map.kind() != kind) throw Error("evalFullHessian: the index handed in was built for the other feature kind");
        stage(kind, laser_cloud);
        double p[7], JtJ[36];
        detail::pose_to_param(pose_local, p);
        int32_t n_valid = 0;
        dev_.check(mlh_match_linearize(dev_.ctx(), kind, p, 5, (with_ua_ ? MLH_FLAG_WITH_UA : 0u) | MLH_FLAG_NO_LOSS, params().MIN_MATCH_SQ_DIS, params().MIN_PLANE_DIS, 0.0,
                                       params().COV_MEASUREMENT_TRACE, nullptr, nullptr, nullptr, nullptr, JtJ, nullptr, nullptr, &n_valid));
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) detail::mat_ref(mat_H, r, c, 6, 0) += JtJ[r * 6 + c];
        feat_num += n_valid;
    }
    // lidar_mapper.h:229-573. all_features[i].type_ stays 'n' for features that were not matched; sel_feature_idx lists the chosen ones in
    // pick order; sub_mat_H comes in as the caller initialised it (1e-6 * I, cpp:505/520) and returns with the selected rows added.
    template <typename KdTreeT, typename PoseT, typename FeatureVec, typename Mat6>
    void goodFeatureMatching(const KdTreeT &kdtree_from_map_, const PointICovCloud &laser_map, const PointICovCloud &laser_cloud, const PoseT &pose_local,
                             FeatureVec &all_features, std::vector<size_t> &sel_feature_idx, const char feature_type,
                             const std::string gf_method, const double gf_ratio, Mat6 &sub_mat_H)
    {
        typedef typename FeatureVec::value_type FeatureT;
        (void)laser_map;
        const MapIndex<PointIWithCov> &kdtree_from_map = detail::deref(kdtree_from_map_, 0);
        const int kind = feature_type == 's' ? MLH_SURF : MLH_CORNER;
        if (kdtree_from_map.kind() != kind) throw Error("goodFeatureMatching: the index handed in was built for the other feature kind");
        static const std::map<std::string, int> methods = {{"wo_gf", MLH_GF_WO}, {"rnd", MLH_GF_RND}, {"fps", MLH_GF_FPS}, {"gd_fix", MLH_GF_GD_FIX}, {"gd_float", MLH_GF_GD_FLOAT}};
        auto it = methods.find(gf_method);
        if (it == methods.end()) throw Error("goodFeatureMatching: unknown gf_method " + gf_method);
        stage(kind, laser_cloud);
        const int m = (int)laser_cloud.size();
        double p[7], H36[36];
        detail::pose_to_param(pose_local, p);
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H36[r * 6 + c] = detail::mat_at(sub_mat_H, r, c, 6, 0);
        std::vector<int32_t> sel(size_t(m > 0 ? m : 1));
        std::vector<uint8_t> matched(size_t(m > 0 ? m : 1));
        int32_t n_sel = 0;
        dev_.check(mlh_good_feature_matching(dev_.ctx(), kind, p, it->second, gf_ratio, seed_, params().MIN_MATCH_SQ_DIS, params().MIN_PLANE_DIS, sel.data(), &n_sel,
                                             H36, matched.data()));
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) detail::mat_ref(sub_mat_H, r, c, 6, 0) = H36[r * 6 + c];
        // coefficients of the matched features (the selected ones are what the block-assembly loop reads)
        std::vector<uint8_t> valid(size_t(m > 0 ? m : 1));
        std::vector<double> coeffs(size_t(m > 0 ? m : 1) * 6);
        all_features.assign(size_t(m), FeatureT());
        sel_feature_idx.assign(sel.begin(), sel.begin() + n_sel);
        int32_t n_valid = 0;
        // after the selection only the chosen correspondences are live on the device: one linearise pass returns their coefficient rows
        dev_.check(mlh_match_coeffs(dev_.ctx(), kind, valid.data(), coeffs.data(), &n_valid));
        for (int i = 0; i < m; ++i) {
            const auto &pt = laser_cloud.points[size_t(i)];
            const bool ok = valid[size_t(i)] != 0;
            detail::feature_set(all_features[size_t(i)], size_t(i), size_t(pt.intensity), ok ? feature_type : 'n', double(pt.x), double(pt.y), double(pt.z),
                                coeffs.data() + size_t(i) * 6, ok ? (feature_type == 's' ? 4 : 6) : 0, nullptr);
        }
    }
private:
    void stage(int kind, const PointICovCloud &laser_cloud)
    {
        dev_.check(mlh_features_set(dev_.ctx(), kind, laser_cloud.points.data(), (int)sizeof(PointIWithCov), (int)laser_cloud.size(),
                                    point_traits<PointIWithCov>::intensity_off, point_traits<PointIWithCov>::cov_off, MLH_MEM_HOST));
    }
    Device &dev_;
    bool with_ua_;
    uint64_t seed_;
};

// ------------------------------------------------------------------ the odometry window's coupled normal equations (estimator.cpp:687-848, 1577-1680)
// J^T J / J^T r of the LidarPureOdom factors staged with LidarPureOdomBatchFactor (mlh_pure_odom_set) over [pivot | frames | extrinsics], and the
// pose half of Estimator::evalDegenracy on its diagonal blocks: V_update_ / is_degenerate_ of every PoseLocalParameterization in para_ids order.
struct WindowNormalEquations {
    int D = 0;
    std::vector<double> JtJ, Jtr;      // D x D row-major, D
    double cost = 0.0;
    int n_residuals = 0;
};
inline void evalWindowNormalEquations(Device &dev, const double pivot[7], const std::vector<std::array<double, 7>> &frames, const std::vector<std::array<double, 7>> &exts,
                                      double huber_delta, WindowNormalEquations &ne)
{
    ne.D = 6 * (1 + (int)frames.size() + (int)exts.size());
    ne.JtJ.assign(size_t(ne.D) * ne.D, 0.0);
    ne.Jtr.assign(size_t(ne.D), 0.0);
    int32_t n = 0;
    dev.check(mlh_pure_odom_normal_eq(dev.ctx(), pivot, frames.empty() ? nullptr : frames[0].data(), (int)frames.size(), exts.empty() ? nullptr : exts[0].data(),
                                      (int)exts.size(), huber_delta, ne.JtJ.data(), ne.Jtr.data(), &ne.cost, &n));
    ne.n_residuals = n;
}

// ceres::Solve of Estimator::optimizeMap's LiDAR factors (estimator.cpp:593-680), device-resident: n_iters Gauss-Newton iterations on the staged factor table
// with the reference's constant blocks (para_pose_[0], estimator.cpp:636; para_ex_pose_[IDX_REF], :642) and every block's V_update_ as evalDegenracy left it
// (local_param_ids in the reference's order: pivot + window poses, then extrinsics). frames / exts: in the linearisation point, out the result.
// Returns the solver status (0 solved, 1 regularised, 2 an update was skipped).
inline int solveWindow(Device &dev, const double pivot[7], std::vector<std::array<double, 7>> &frames, std::vector<std::array<double, 7>> &exts, double huber_delta,
                       int n_iters, const std::vector<PoseLocalParameterization *> *local_param_ids = nullptr, int idx_ref = 0, double *final_cost = nullptr)
{
    const int nb = 1 + (int)frames.size() + (int)exts.size();
    std::vector<double> V;
    if (local_param_ids && (int)local_param_ids->size() == nb) {
        V.resize(size_t(nb) * 36);
        for (int b = 0; b < nb; ++b) for (int k = 0; k < 36; ++k) V[size_t(b) * 36 + k] = (*local_param_ids)[size_t(b)]->V_update_[k];
    }
    const uint32_t const_mask = 1u | (1u << (1 + (int)frames.size() + idx_ref));
    int32_t n = 0, status = 0;
    double cost = 0.0;
    dev.check(mlh_pure_odom_gn_solve(dev.ctx(), pivot, frames.empty() ? nullptr : frames[0].data(), (int)frames.size(), exts.empty() ? nullptr : exts[0].data(),
                                     (int)exts.size(), huber_delta, n_iters, const_mask, V.empty() ? nullptr : V.data(), &cost, &n, &status));
    if (final_cost) *final_cost = cost;
    return status;
}

// Estimator::evalDegenracy (estimator.cpp:1598-1680) on the window's normal equations, in the reference's argument order minus the Jacobian
// (J^T J comes from evalWindowNormalEquations instead of a ceres::CRSMatrix): local_param_ids = the OPT_WINDOW_SIZE + 1 pose blocks, then one per
// LiDAR extrinsic. Pose blocks: the mapper's rule per diagonal block with that block's threshold eig_thre[i]; a degenerate block gets
// is_degenerate_ and its projector V_update_. Extrinsic blocks (ESTIMATE_EXTRINSIC != 0): on calibration frames (frame_cnt % N_CUMU_FEATURE == 0)
// lambda = lambda_min / N_CUMU_FEATURE -- >= LAMBDA_THRE_CALIB: the threshold is set to it and d_factor_calib records lambda; above the running
// threshold: the threshold rises to lambda; otherwise, and on every other frame, the extrinsic is frozen (is_degenerate_, V_update_ = 0).
// Host code over mlh_eval_degeneracy; st carries what the Estimator keeps between calls (eig_thre_, d_factor_calib_, log_lambda_).
struct WindowDegeneracyState {
    std::vector<double> eig_thre;         // eig_thre_: one per block of local_param_ids
    std::vector<double> d_factor_calib;   // d_factor_calib_: one per LiDAR
    std::vector<double> log_lambda;       // log_lambda_
};
inline void evalDegenracy(std::vector<PoseLocalParameterization *> &local_param_ids, const WindowNormalEquations &ne, int frame_cnt, WindowDegeneracyState &st)
{
    const Params &P = params();
    if (ne.n_residuals == 0) return;
    const int D = ne.D, n_pose = P.OPT_WINDOW_SIZE + 1;
    if (D != 6 * (int)local_param_ids.size() || (int)st.eig_thre.size() != (int)local_param_ids.size()) throw Error("evalDegenracy: block count mismatch");
    auto block = [&](size_t i) {
        std::array<double, 36> H;
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[size_t(r * 6 + c)] = ne.JtJ[size_t(6 * i + r) * D + 6 * i + c];
        return H;
    };
    for (size_t i = 0; i < size_t(n_pose) && i < local_param_ids.size(); ++i) {
        double ev[6], V[36];
        if (mlh_eval_degeneracy(block(i).data(), st.eig_thre[i], ev, V) > 0) {
            local_param_ids[i]->is_degenerate_ = true;
            std::memcpy(local_param_ids[i]->V_update_.data(), V, sizeof(V));
        }
    }
    if (P.ESTIMATE_EXTRINSIC != 0) {
        st.d_factor_calib.assign(size_t(P.NUM_OF_LASER), 0.0);
        for (size_t i = size_t(n_pose); i < local_param_ids.size(); ++i) {
            bool freeze = true;
            if (frame_cnt % P.N_CUMU_FEATURE == 0) {
                double ev[6], V[36];
                (void)mlh_eval_degeneracy(block(i).data(), 0.0, ev, V);
                const double lambda = ev[0] / P.N_CUMU_FEATURE;
                st.log_lambda.push_back(lambda);
                if (lambda >= P.LAMBDA_THRE_CALIB) { st.eig_thre[i] = P.LAMBDA_THRE_CALIB; st.d_factor_calib[i - size_t(n_pose)] = lambda; freeze = false; }
                else if (lambda > st.eig_thre[i]) { st.eig_thre[i] = lambda; freeze = false; }
            }
            if (freeze) { local_param_ids[i]->is_degenerate_ = true; local_param_ids[i]->V_update_.fill(0.0); }
        }
    }
}

inline Pose poseMul(const Pose &a, const Pose &b);          // (defined with the mapper's pose chain below)
inline Pose poseInverse(const Pose &a);

// The factor table of Estimator::optimizeMap built ON THE DEVICE (estimator.cpp:700-780): in place of
//     f_extract_.matchCornerFromMap / matchSurfFromMap(kdtree, local_map, features_of(frame i, LiDAR n), pose_local, all_features, ...)
//     for (feature : all_features) problem.AddResidualBlock(new LidarPureOdom{PlaneNorm,Edge}Factor(point, coeffs, 1.0), loss, para_pose_[0], para_pose_[i - pivot], para_ex_pose_[n])
// one addMatches per (frame, LiDAR, kind): the match pass runs against the resident local map (MapIndex::setInputCloud) at
// pose_local = T_pivot^-1 T_i T_ext and its valid correspondences become factors in HBM; evalWindowNormalEquations then reduces them.
class WindowFactorTable {
public:
    explicit WindowFactorTable(Device &dev) : dev_(dev) { dev_.check(mlh_pure_odom_begin(dev_.ctx())); }
    // frame: 1..W as in LidarPureOdomBatchFactor::add; laser: 0..L-1; type 's' / 'c'
    template <typename PointT, typename PoseT>
    void addMatches(const PointCloud<PointT> &features_in_lidar_frame, char type, const PoseT &pose_local, int frame, int laser, size_t n_neigh = 5, bool check_fov = true)
    {
        const int m = (int)features_in_lidar_frame.size();
        if (m == 0) return;
        const int kind = type == 's' ? MLH_SURF : MLH_CORNER;
        dev_.check(mlh_features_set(dev_.ctx(), kind, features_in_lidar_frame.points.data(), (int)sizeof(PointT), m, point_traits<PointT>::intensity_off,
                                    point_traits<PointT>::cov_off, MLH_MEM_HOST));
        double pose[7];
        detail::pose_to_param(pose_local, pose);
        const Params &P = params();
        dev_.check(mlh_pure_odom_add_matches(dev_.ctx(), kind, pose, (int)n_neigh, check_fov ? MLH_FLAG_CHECK_FOV : 0u, P.MIN_MATCH_SQ_DIS, P.MIN_PLANE_DIS, frame - 1, laser));
    }
    // Estimator::goodFeatureMatching (estimator.cpp:1347-1517) as buildLocalMap calls it per (frame, LiDAR) and kind (cpp:1241-1263) -- the odometry's own selection:
    // all features matched and their scored rows evaluated on the GPU, the reference's draw loop (seeded: the reference seeds rgi_ from std::random_device) on the
    // host; sel_feature_idx comes back in selection order and the selected correspondences are already factors of this table (no AddResidualBlock loop, cpp:733-780).
    // gf_ratio: ODOM_GF_RATIO, a float in the reference (parameters.cpp:85). n_neigh = 5 and no FOV check, as the reference's call has them (cpp:1384-1404).
    template <typename PointT>
    void goodFeatureMatching(const PointCloud<PointT> &laser_cloud, std::vector<size_t> &sel_feature_idx, char feature_type, const Pose &pose_pivot, const Pose &pose_i,
                             const Pose &pose_ext, float gf_ratio, int frame, int laser, uint64_t seed)
    {
        sel_feature_idx.clear();
        const int m = (int)laser_cloud.size();
        if (m == 0) return;
        const int kind = feature_type == 's' ? MLH_SURF : MLH_CORNER;
        dev_.check(mlh_features_set(dev_.ctx(), kind, laser_cloud.points.data(), (int)sizeof(PointT), m, point_traits<PointT>::intensity_off, point_traits<PointT>::cov_off,
                                    MLH_MEM_HOST));
        const Pose pose_local = poseMul(poseMul(poseInverse(pose_pivot), pose_i), pose_ext);      // Pose(T_pivot^-1 T_i T_ext), cpp:1358
        double rel[7], pv[7], pi[7], pe[7];
        pose_local.toParam(rel); pose_pivot.toParam(pv); pose_i.toParam(pi); pose_ext.toParam(pe);
        const Params &P = params();
        std::vector<int32_t> sel((size_t)m);
        int32_t n_sel = 0;
        dev_.check(mlh_pure_odom_add_matches_gf(dev_.ctx(), kind, rel, pv, pi, pe, 5, 0u, P.MIN_MATCH_SQ_DIS, P.MIN_PLANE_DIS, frame - 1, laser, gf_ratio, seed, sel.data(), &n_sel));
        sel_feature_idx.assign(sel.begin(), sel.begin() + n_sel);
    }
private:
    Device &dev_;
};

// every voxel filter of the device walks a voxel's members in the order libstdc++'s unstable std::sort leaves them, as the reference's filters do
// (voxel_grid_covariance_mloam_impl.hpp:227) -- exact LiDAR ids in mixed voxels of fused clouds. ON by default, produced on the device
// (mode 1); mode 2 = the same through a host pass with the platform's own std::sort; false / mode 0 walks members by point index
// instead (cheaper, ids of mixed voxels may differ)
inline void setVoxelMemberOrderAsReference(Device &dev, bool on) { dev.check(mlh_set_voxel_member_order(dev.ctx(), on ? 1 : 0)); }
// extractCloud's order among equal curvatures: the reference's (std::sort, default) or (curvature, index)
inline void setExtractTieOrderAsReference(Device &dev, bool on) { dev.check(mlh_set_extract_tie_order(dev.ctx(), on ? 1 : 0)); }
inline void setVoxelMemberOrderMode(Device &dev, int mode) { dev.check(mlh_set_voxel_member_order(dev.ctx(), mode)); }

// ------------------------------------------------------------------ scan2MapOptimization() (gf_method "wo_gf")
struct Scan2MapReport {
    std::vector<mlh_iter_stat> outer;   // one per outer iteration: matched counts, H, eigenvalues, LM iterations, costs
};

// Replaces the body of scan2MapOptimization (lidar_mapper_keyframe.cpp:423-639): index build for both maps, max_iter x
// { match all features, evalHessian + evalDegenracy, Levenberg-Marquardt with Ceres' trust-region semantics }, all on the GPU.
// pose_wmap_curr: the stand-in Pose or the reference's own (Eigen members; its T_ is refreshed through update()).
template <typename PoseT>
inline void scan2MapOptimization(Device &dev, const PointICovCloud &laser_cloud_surf_from_map_cov_ds, const PointICovCloud &laser_cloud_corner_from_map_cov_ds,
                                 const PointICovCloud &laser_cloud_surf_cov, const PointICovCloud &laser_cloud_corner_cov, PoseT &pose_wmap_curr,
                                 bool with_ua_flag, Scan2MapReport *report = nullptr, int max_iter = 2)
{
    const Params &P = params();
    if (!(laser_cloud_surf_from_map_cov_ds.size() > 50 && laser_cloud_corner_from_map_cov_ds.size() > 10)) {   // cpp:429
        const double zero[36] = {0};
        detail::pose_cov_set(pose_wmap_curr, zero);
        return;
    }
    MapIndex<PointIWithCov> kdtree_surf_from_map(dev, MLH_SURF), kdtree_corner_from_map(dev, MLH_CORNER);
    kdtree_surf_from_map.setInputCloud(laser_cloud_surf_from_map_cov_ds);       // cpp:433
    kdtree_corner_from_map.setInputCloud(laser_cloud_corner_from_map_cov_ds);   // cpp:434
    dev.check(mlh_features_set(dev.ctx(), MLH_SURF, laser_cloud_surf_cov.points.data(), 48, (int)laser_cloud_surf_cov.size(), 16, 20, MLH_MEM_HOST));
    dev.check(mlh_features_set(dev.ctx(), MLH_CORNER, laser_cloud_corner_cov.points.data(), 48, (int)laser_cloud_corner_cov.size(), 16, 20, MLH_MEM_HOST));
    mlh_solver_opts o;
    mlh_solver_opts_default(&o);
    o.min_match_sq_dis = P.MIN_MATCH_SQ_DIS; o.min_plane_dis = P.MIN_PLANE_DIS; o.huber_delta = P.HUBER_DELTA; o.map_eig_thre = P.MAP_EIG_THRE;
    o.cov_measurement_trace = P.COV_MEASUREMENT_TRACE; o.flags = with_ua_flag ? MLH_FLAG_WITH_UA : 0u; o.max_outer = max_iter;
    double pose[7];
    detail::pose_to_param(pose_wmap_curr, pose);
    std::vector<mlh_iter_stat> stats(max_iter);
    dev.check(mlh_scan2map(dev.ctx(), pose, &o, stats.data()));
    detail::pose_from_param(pose_wmap_curr, pose);
    if (report) report->outer = stats;
}

// The feature-cloud "message" between an estimator-side Device and a mapper-side Device on the same GPU (the reference's nodes exchange host clouds over ROS:
// estimator.cpp publishes, lidar_mapper_keyframe.cpp:162-190 queues): the staged feature sets of `from` become those of `to`, device to device.
inline void handOverFeatures(Device &to, Device &from)
{
    to.check(mlh_features_copy(to.ctx(), from.ctx(), MLH_SURF));
    to.check(mlh_features_copy(to.ctx(), from.ctx(), MLH_CORNER));
}

// ------------------------------------------------------------------ the mapper's frame loop with the GPU kept busy across frames
// LidarMapper::process() (lidar_mapper_keyframe.cpp:1000-1100) runs transformAssociateToMap -> extractSurroundingKeyFrames -> downsampleCurrentScan ->
// scan2MapOptimization -> transformUpdate per frame. FramePipeline issues the same per-frame work so that frame k + 1's map index is built on a second stream
// while frame k is being solved, and frame k + 1's solve is queued behind frame k's with its start pose chained on the device (cpp:145-160). Up to two frames
// in flight; poses come back in submission order.
class FramePipeline {
public:
    explicit FramePipeline(Device &dev, int gn_iters = 5, bool with_ua_flag = false) : dev_(dev), iters_(gn_iters)
    {
        const Params &P = params();
        mlh_solver_opts_default(&o_);
        o_.min_match_sq_dis = P.MIN_MATCH_SQ_DIS; o_.min_plane_dis = P.MIN_PLANE_DIS; o_.huber_delta = P.HUBER_DELTA; o_.map_eig_thre = P.MAP_EIG_THRE;
        o_.cov_measurement_trace = P.COV_MEASUREMENT_TRACE; o_.flags = with_ua_flag ? MLH_FLAG_WITH_UA : 0u;
    }
    // kdtree_*_from_map->setInputCloud (cpp:433-434) for the frame about to be submitted; overlapped with the solve in flight when there is one
    void setInputClouds(const PointICovCloud &surf_map, const PointICovCloud &corner_map)
    {
        dev_.check(mlh_map_set_pair_overlapped(dev_.ctx(), surf_map.points.data(), (int)surf_map.size(), corner_map.points.data(), (int)corner_map.size(),
                                               (int)sizeof(PointIWithCov), params().MIN_MATCH_SQ_DIS, MLH_MEM_HOST));
    }
    void setFeatures(const PointICovCloud &surf_cov, const PointICovCloud &corner_cov)
    {
        dev_.check(mlh_features_set(dev_.ctx(), MLH_SURF, surf_cov.points.data(), 48, (int)surf_cov.size(), 16, 20, MLH_MEM_HOST));
        dev_.check(mlh_features_set(dev_.ctx(), MLH_CORNER, corner_cov.points.data(), 48, (int)corner_cov.size(), 16, 20, MLH_MEM_HOST));
    }
    // first frame (or a frame redone after a keyframe-selection mismatch): start pose given
    void submit(const Pose &pose_wmap_curr)
    {
        double p[7];
        pose_wmap_curr.toParam(p);
        if (scan2map_) dev_.check(mlh_scan2map_begin(dev_.ctx(), p, &o_, lm_lookahead_));
        else dev_.check(mlh_gn_solve_begin(dev_.ctx(), p, iters_, &o_));
        ++in_flight_;
    }
    // every later frame: start pose = (previous result * pose_wodom_prev.inverse()) * pose_wodom_curr, evaluated on the device
    void submitChained(const Pose &pose_wodom_prev, const Pose &pose_wodom_curr)
    {
        double a[7], b[7];
        pose_wodom_prev.toParam(a); pose_wodom_curr.toParam(b);
        if (scan2map_) dev_.check(mlh_scan2map_begin_chained(dev_.ctx(), a, b, &o_, lm_lookahead_));
        else dev_.check(mlh_gn_solve_begin_chained(dev_.ctx(), a, b, iters_, &o_));
        ++in_flight_;
    }
    // The frames are solved by the reference's own per-frame call, scan2MapOptimization (2 outer iterations x Levenberg-Marquardt), submitted and collected
    // separately (mlh_scan2map_begin / _end) instead of `gn_iters` Gauss-Newton iterations. lm_lookahead 0: automatic. After collect(), lastStatus() is
    // mlh_scan2map_end's status: 0 / 2 = the pose is scan2MapOptimization's; 1 = the pose returned is the frame's START pose and the caller has to solve the frame
    // with scan2MapOptimization(...) on its inputs (it overflowed the look-ahead with a younger frame chained behind it); 3 = the frame was chained behind a frame
    // that ended with 1 / 3 (it began from an unfinished pose): resubmit it after the predecessor has been solved.
    void useScan2Map(bool on, int lm_lookahead = 0) { scan2map_ = on; lm_lookahead_ = lm_lookahead; }
    int lastStatus() const { return last_status_; }
    Pose collect()
    {
        double p[7];
        last_status_ = 0;
        if (scan2map_) { int32_t st = 0; dev_.check(mlh_scan2map_end(dev_.ctx(), p, &st)); last_status_ = st; }
        else dev_.check(mlh_gn_solve_end(dev_.ctx(), p));
        --in_flight_;
        Pose r;
        r.fromParam(p);
        return r;
    }
    int inFlight() const { return in_flight_; }
private:
    Device &dev_;
    mlh_solver_opts o_;
    int iters_, in_flight_ = 0;
    bool scan2map_ = false;
    int lm_lookahead_ = 0, last_status_ = 0;
};

// ------------------------------------------------------------------ what makes staging beside the solve legal, as code
// Pose::operator* / Pose::inverse (pose.cpp:99-113): both construct their result through Pose(q, t), which normalises the quaternion; the rotation of a vector is
// Eigen's v + w (2 u x v) + u x (2 u x v), Quaterniond::inverse is conjugate / squaredNorm, the quaternion product in Eigen's term order -- the arithmetic of the
// device's chain (csrc/dev_math.hpp: chain_start_pose), so a host-side prior and the device's chained start pose are the same bits
// (tests/test_abi.py::test_facade_keyframe_policy_and_pose_chain holds both against the reference's own lines).
namespace detail {
inline void quat_rotate(const Quat &q, const double v[3], double o[3])
{
    double uv[3] = {q.y * v[2] - q.z * v[1], q.z * v[0] - q.x * v[2], q.x * v[1] - q.y * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c2[3] = {q.y * uv[2] - q.z * uv[1], q.z * uv[0] - q.x * uv[2], q.x * uv[1] - q.y * uv[0]};
    o[0] = v[0] + q.w * uv[0] + c2[0]; o[1] = v[1] + q.w * uv[1] + c2[1]; o[2] = v[2] + q.w * uv[2] + c2[2];
}
inline Quat quat_mul(const Quat &a, const Quat &b)
{
    Quat r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}
inline Pose pose_from(const Quat &q, const double t[3])          // Pose(q, t): q normalised
{
    Pose r;
    const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    r.q_.x = q.x / n; r.q_.y = q.y / n; r.q_.z = q.z / n; r.q_.w = q.w / n;
    r.t_(0) = t[0]; r.t_(1) = t[1]; r.t_(2) = t[2];
    return r;
}
}  // namespace detail
inline Pose poseMul(const Pose &a, const Pose &b)
{
    double rt[3];
    detail::quat_rotate(a.q_, b.t_.v, rt);
    const double t[3] = {rt[0] + a.t_(0), rt[1] + a.t_(1), rt[2] + a.t_(2)};
    return detail::pose_from(detail::quat_mul(a.q_, b.q_), t);
}
inline Pose poseInverse(const Pose &a)
{
    const double n2 = a.q_.x * a.q_.x + a.q_.y * a.q_.y + a.q_.z * a.q_.z + a.q_.w * a.q_.w;
    Quat qi;
    if (n2 > 0.0) { qi.x = -a.q_.x / n2; qi.y = -a.q_.y / n2; qi.z = -a.q_.z / n2; qi.w = a.q_.w / n2; }
    else { qi.x = qi.y = qi.z = qi.w = 0.0; }
    double mt[3];
    detail::quat_rotate(qi, a.t_.v, mt);
    const double t[3] = {-mt[0], -mt[1], -mt[2]};
    return detail::pose_from(qi, t);
}

// The mapper's keyframe bookkeeping: saveKeyframe's test (lidar_mapper_keyframe.cpp:641-657) and extractSurroundingKeyFrames' selection (cpp:263-272). The
// reference rebuilds the local map ONLY in the frame after a keyframe was saved (cpp:257-261 return early while the map clouds are non-empty; :1101 clears them
// when save_new_keyframe) -- that is what decides whether a frame's maps may be staged while the previous frame is still being solved.
class KeyframePolicy {
public:
    KeyframePolicy() : dist_(params().DISTANCE_KEYFRAMES), ori_deg_(params().ORIENTATION_KEYFRAMES), radius_(params().SURROUNDING_KF_RADIUS) {}
    KeyframePolicy(float distance_keyframes, float orientation_keyframes_deg, float surrounding_kf_radius)
        : dist_(distance_keyframes), ori_deg_(orientation_keyframes_deg), radius_(surrounding_kf_radius) {}
    // would saveKeyframe() save a frame that ends at this pose? (no state change)
    bool wouldSave(const Pose &pose_wmap_curr) const
    {
        if (pose_keyframes_3d.size() == 0) return true;
        const float cx = float(pose_wmap_curr.t_(0)), cy = float(pose_wmap_curr.t_(1)), cz = float(pose_wmap_curr.t_(2));      // PointI fields: f32, as pose_point_cur
        const double d = std::sqrt(double((cx - prev_[0]) * (cx - prev_[0]) + (cy - prev_[1]) * (cy - prev_[1]) + (cz - prev_[2]) * (cz - prev_[2])));
        const Quat &a = pose_wmap_curr.q_, &b = q_prev_;
        const double dot = std::fabs(a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z);
        const double ang = 2.0 * std::acos(std::min(1.0, dot));              // Eigen's angularDistance
        return d > dist_ || ang / M_PI * 180 > ori_deg_;
    }
    // saveKeyframe(): the test, and on success the bookkeeping (the caller stores the frame's clouds under the returned index); -1 when not saved
    int save(const Pose &pose_wmap_curr)
    {
        if (!wouldSave(pose_wmap_curr)) return -1;
        PointI p;
        p.x = float(pose_wmap_curr.t_(0)); p.y = float(pose_wmap_curr.t_(1)); p.z = float(pose_wmap_curr.t_(2));
        p.intensity = float(pose_keyframes_3d.size());
        prev_[0] = p.x; prev_[1] = p.y; prev_[2] = p.z;
        q_prev_ = pose_wmap_curr.q_;
        pose_keyframes_3d.push_back(p);
        pose_keyframes_6d.push_back(pose_wmap_curr);
        return int(pose_keyframes_3d.size()) - 1;
    }
    // kdtree_surrounding_keyframes->radiusSearch(pose_point_cur, SURROUNDING_KF_RADIUS, ...) (cpp:266-272): keyframe indices within the radius, nearest first
    // (PCL returns a radius search sorted by distance; equal distances by index here)
    std::vector<int> surrounding(const Pose &pose_wmap_curr) const
    {
        const float cx = float(pose_wmap_curr.t_(0)), cy = float(pose_wmap_curr.t_(1)), cz = float(pose_wmap_curr.t_(2));
        std::vector<std::pair<float, int>> hit;
        for (size_t i = 0; i < pose_keyframes_3d.size(); ++i) {
            const PointI &k = pose_keyframes_3d.points[i];
            const float dx = k.x - cx, dy = k.y - cy, dz = k.z - cz, d2 = dx * dx + dy * dy + dz * dz;
            if (d2 <= radius_ * radius_) hit.emplace_back(d2, int(i));
        }
        std::sort(hit.begin(), hit.end());
        std::vector<int> ids;
        for (const auto &h : hit) ids.push_back(h.second);
        return ids;
    }
    PointICloud pose_keyframes_3d;
    std::vector<Pose> pose_keyframes_6d;
private:
    float dist_, ori_deg_, radius_, prev_[3] = {0, 0, 0};
    Quat q_prev_;
};

// The mapper's frame loop, pipelined, with the precondition of the overlap checked per frame instead of assumed:
//   frame k's local map is the previous frame's, unchanged, unless frame k - 1 was saved as a keyframe (then it is rebuilt around the prior of frame k from the
//   keyframes' clouds -- frame k - 1's own cloud at its SOLVED pose among them). So while frame k - 1 is being solved, frame k's maps can be staged beside it
//   exactly when frame k - 1 will not be a keyframe. That is predicted from frame k - 1's prior (the pose before scan-to-map's correction, centimetres from the
//   result), verified on its result, and on a wrong prediction frame k is solved again on the rebuilt map, synchronously.
// process() takes frame k's inputs and returns frame k - 1's pose (one frame late, as FramePipeline); finish() returns the last one.
// assemble(ids, prior, surf_map, corner_map): the caller's local-map assembly for the keyframe indices `ids` (cloudUCTAssociateToMap + the two voxel filters,
// cpp:296-344) -- called only when the reference would rebuild.
class PipelinedMapper {
public:
    typedef std::function<void(const std::vector<int> &, const Pose &, PointICovCloud &, PointICovCloud &)> Assemble;
    typedef std::function<void(int, const Pose &)> OnKeyframe;      // (keyframe index, its pose): store the frame's clouds (saveKeyframe, cpp:673-683)
    PipelinedMapper(Device &dev, KeyframePolicy &kf, Assemble assemble, OnKeyframe on_keyframe, int gn_iters = 5, bool with_ua_flag = false)
        : pipe_(dev, gn_iters, with_ua_flag), kf_(kf), assemble_(std::move(assemble)), on_keyframe_(std::move(on_keyframe)) {}
    void setInitialMap(const PointICovCloud &surf_map, const PointICovCloud &corner_map) { surf_map_ = surf_map; corner_map_ = corner_map; }
    void setInitialPose(const Pose &pose_wmap_curr, const Pose &pose_wodom_curr) { wmap_wodom_ = poseMul(pose_wmap_curr, poseInverse(pose_wodom_curr)); }
    struct Counters { int frames = 0, overlapped = 0, waited = 0, redone = 0, keyframes = 0; } counters;

    // returns true and fills pose_prev when a previous frame's pose came back with this call
    bool process(const PointICovCloud &surf_cov, const PointICovCloud &corner_cov, const Pose &pose_wodom_curr, Pose &pose_prev)
    {
        ++counters.frames;
        const Pose prior = poseMul(wmap_wodom_, pose_wodom_curr);             // transformAssociateToMap with the correction known NOW (frame k - 2's while k - 1 is in flight)
        if (!in_flight_) {
            pipe_.setInputClouds(surf_map_, corner_map_);
            pipe_.setFeatures(surf_cov, corner_cov);
            pipe_.submit(prior);
            in_flight_ = true; prior_in_flight_ = prior; wodom_in_flight_ = pose_wodom_curr;
            return false;
        }
        const bool predicted_keyframe = kf_.wouldSave(prior_in_flight_);
        if (!predicted_keyframe) {
            // frame k - 1 is not expected to be saved: frame k matches against the same local map -- stage it and submit frame k behind the solve in flight
            pipe_.setInputClouds(surf_map_, corner_map_);
            pipe_.setFeatures(surf_cov, corner_cov);
            pipe_.submitChained(wodom_in_flight_, pose_wodom_curr);
            pose_prev = pipe_.collect();                                      // frame k - 1
            const bool saved = closeFrame(pose_prev);
            if (saved) {
                // wrong prediction (the result crossed the keyframe threshold the prior stayed under): frame k was matched against a map without keyframe k - 1.
                // Its solve is allowed to finish and dropped; the map is rebuilt as the reference would have, and frame k is solved again from the host-side prior.
                (void)pipe_.collect();
                ++counters.redone;
                const Pose prior_k = poseMul(wmap_wodom_, pose_wodom_curr);
                rebuild(prior_k);
                pipe_.setInputClouds(surf_map_, corner_map_);
                pipe_.submit(prior_k);
                prior_in_flight_ = prior_k;
            } else {
                ++counters.overlapped;
                prior_in_flight_ = poseMul(wmap_wodom_, pose_wodom_curr);     // (what the device computed for frame k, restated for the next prediction)
            }
        } else {
            // frame k - 1 is expected to be saved as a keyframe: frame k's map contains its cloud at its solved pose -- nothing to stage before that pose is known
            pose_prev = pipe_.collect();
            ++counters.waited;
            const bool saved = closeFrame(pose_prev);
            const Pose prior_k = poseMul(wmap_wodom_, pose_wodom_curr);
            if (saved) rebuild(prior_k);
            pipe_.setInputClouds(surf_map_, corner_map_);
            pipe_.setFeatures(surf_cov, corner_cov);
            pipe_.submit(prior_k);
            prior_in_flight_ = prior_k;
        }
        wodom_in_flight_ = pose_wodom_curr;
        return true;
    }
    Pose finish()
    {
        Pose p = pipe_.collect();
        in_flight_ = false;
        closeFrame(p);
        return p;
    }
    const PointICovCloud &surfMap() const { return surf_map_; }
    const PointICovCloud &cornerMap() const { return corner_map_; }
private:
    // transformUpdate + saveKeyframe for the frame whose pose just came back (cpp:1076-1079)
    bool closeFrame(const Pose &pose_wmap_curr)
    {
        wmap_wodom_ = poseMul(pose_wmap_curr, poseInverse(wodom_in_flight_));
        const int idx = kf_.save(pose_wmap_curr);
        if (idx < 0) return false;
        ++counters.keyframes;
        if (on_keyframe_) on_keyframe_(idx, pose_wmap_curr);
        return true;
    }
    void rebuild(const Pose &prior)
    {
        PointICovCloud s, c;
        assemble_(kf_.surrounding(prior), prior, s, c);
        surf_map_ = std::move(s); corner_map_ = std::move(c);
    }
    FramePipeline pipe_;
    KeyframePolicy &kf_;
    Assemble assemble_;
    OnKeyframe on_keyframe_;
    PointICovCloud surf_map_, corner_map_;
    Pose wmap_wodom_, prior_in_flight_, wodom_in_flight_;
    bool in_flight_ = false;
};

}  // namespace mloam_hip
