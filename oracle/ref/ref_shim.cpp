// TEST INFRASTRUCTURE ONLY -- oracle/_ref: the reference's OWN SOURCE LINES, compiled where they lie.
// The reference cannot be built as a whole in this image (every translation unit includes Eigen, PCL + FLANN, Ceres, glog, ROS headers,
// none of which exist here), but the functions on the hot path are plain C++ over a few dozen library calls. oracle/ref/build_ref.py cuts
// the line ranges of those functions out of the files under /root/reference into oracle/_ref/gen/*.inc (never committed, deleted after
// the compile) and this file supplies just enough context -- a mini Eigen (mini_eigen.hpp), a vector-backed pcl::PointCloud, a
// pcl::VoxelGrid that calls the oracle's restatement, empty ceres / ROS bases -- for g++ to compile them VERBATIM:
//   ImageSegmenter (class, projectCloud, segmentCloud, setParameter)   estimator/src/imageSegmenter/image_segmenter.hpp:36-83, 88-136, 138-393;
//                                          image_segmenter.cpp:18-63  (its three UB spots compile to whatever g++ makes of them: see oracle/image_segmenter.hpp)
//   FeatureExtract::extractCloud           estimator/src/featureExtract/feature_extract.cpp:118-297  (+ compObject, feature_extract.hpp:48-53)
//   LidarMapPlaneNormFactor ctor/Evaluate  estimator/src/factor/lidar_map_factor.hpp:26-71, 122-126
//   LidarMapEdgeFactor ctor/Evaluate       estimator/src/factor/lidar_map_factor.hpp:130-174, 231-235
//   FeatureExtract::match{Corner,Surf}PointFromMap   estimator/src/featureExtract/feature_extract.hpp:645-788, 790-883 (+ decls 111-128)
//   pointAssociateToMap, common::sqrSum    estimator/src/utility/utility.h:102-117; mloam_common/.../algos/math.hpp:10-14
//   Utility::deltaQ, skewSymmetric         estimator/src/utility/utility.h:169-195
//   LidarPureOdom{PlaneNorm,Edge}Factor    estimator/src/factor/lidar_pure_odom_factor.hpp:27-102, 191-195; 198-282, 377-381
//   LidarOnlineCalib{PlaneNorm,Edge}Factor estimator/src/factor/lidar_online_calib_factor.hpp:24-62, 117-121; 125-165, 223-227
//   PoseLocalParameterization::{setParameter, Plus}   estimator/src/factor/pose_local_parameterization.h:21-33, .cpp:16-45
//   pointToFS, evalPointUncertainty (both overloads)  estimator/src/lidarMapper/associate_uct.hpp:150-156, 164-193, 195-215
//   adjointMatrix, covop1, covop2, compoundPoseWithCov (explicit-covariance overload)   associate_uct.hpp:9-86
//   TransformToStart / TransformToEnd      estimator/src/utility/utility.h:54-100
//   FeatureExtract::match{Corner,Surf}FromScan        estimator/src/featureExtract/feature_extract.hpp:131-376 (+ decls 78-90)
//   LidarScanPlaneNormFactor, LidarScanEdgeFactorVector   estimator/src/factor/lidar_scan_factor.hpp:25-62, 122-126; 236-279, 339-343
//   Pose::{Pose(), Pose(const Pose &), Pose(q, t, td), inverse, update, operator*}   estimator/src/estimator/pose.cpp:16-41, 99-113
//   compoundPoseWithCov (pose.cov_ overload), cloudUCTAssociateToMap, evalDegenracy   associate_uct.hpp:88-147; lidar_mapper_keyframe.cpp:1116-1158, 1171-1204
//   downsampleCurrentScan                  estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:356-421 (+ PointXYZIWithCov ctors point_with_cov.hpp:57-62, 91-101)
//   ActiveFeatureSelection::{evaluateFeatJacobianMatching, evalFullHessian, goodFeatureMatching}   estimator/src/lidarMapper/lidar_mapper.h:130-573
//     (+ PointPlaneFeature / FeatureWithScore parameters.h:163-191, extractCov point_with_cov.hpp:202-214, common::logDet math.hpp:172-202,
//      common::RandomGeneratorInt random_generator.hpp:52-66, the two limits lidar_mapper.h:82-83)
//   VoxelGridCovarianceMLOAM<PointT>::applyFilter   mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp:68-457 (both branches; std::sort at :227)
//   PoseLocalParameterization::ComputeJacobian       estimator/src/factor/pose_local_parameterization.cpp:47-55
//   scan2MapOptimization (+ vector2Double / double2Vector)   estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:423-639, 236-252 -- over a Ceres-SHAPED
//   LidarTracker::trackCloud                                 estimator/src/lidarTracker/lidar_tracker.cpp:23-129            -- shim (below) whose minimiser is oracle/lm.hpp
// What this pins: every decision, loop bound, comparison, term and sign the reference's own code makes (the labels and the four feature
// lists; residual and Jacobian formulas). What it does not: the arithmetic INSIDE the third-party calls (Eigen products, PCL's voxel
// centroids), which here is the shim's / the oracle's restatement.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <memory>
#include <numeric>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <vector>
#include "mini_eigen.hpp"
#include "../feature_extract.hpp"      // orc::PointI, orc::voxel_grid_xyzi (the oracle's pcl::VoxelGrid restatement)
#include "../kdtree.hpp"               // orc::KdTree (the pcl::KdTreeFLANN restatement: exact k-NN, f32 L2_Simple accumulation)

using namespace std;                   // the reference's headers pull this in (parameters.h); its .cpp files rely on it (`pair`, `sqrt`)

// ---------------------------------------------------------------- pcl / boost / ROS / ceres context
namespace boost {
using std::shared_ptr;
template <typename T, typename... A> std::shared_ptr<T> make_shared(A &&...a) { return std::make_shared<T>(std::forward<A>(a)...); }
}  // namespace boost
namespace pcl {
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };
struct PointXYZ { float x = 0, y = 0, z = 0; };                 // the raw driver cloud's point (FeatureExtract::calTimestamp's input)
template <typename P> struct PointCloud {
    typedef boost::shared_ptr<PointCloud<P>> Ptr;
    std::vector<P> points;
    unsigned width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    void resize(size_t n) { points.resize(n); }
    typename std::vector<P>::iterator begin() { return points.begin(); }
    typename std::vector<P>::const_iterator begin() const { return points.begin(); }
    typename std::vector<P>::const_iterator end() const { return points.end(); }
    P &operator[](size_t i) { return points[i]; }
    const P &operator[](size_t i) const { return points[i]; }
    // a position that is no longer inside the row erases nothing (std::vector::erase there is undefined: oracle/image_segmenter.hpp U2)
    typename std::vector<P>::iterator erase(typename std::vector<P>::iterator it) { return (it < points.begin() || it >= points.end()) ? points.end() : points.erase(it); }
    void push_back(const P &p) { points.push_back(p); }
    void clear() { points.clear(); }
    PointCloud &operator+=(const PointCloud &o) { points.insert(points.end(), o.points.begin(), o.points.end()); return *this; }
};
template <typename P> struct VoxelGrid {       // 3p: centroid per voxel -> the oracle's restatement of pcl::VoxelGrid::applyFilter
    typename PointCloud<P>::Ptr in;
    float leaf = 0.f;
    void setInputCloud(const typename PointCloud<P>::Ptr &c) { in = c; }
    void setLeafSize(float lx, float, float) { leaf = lx; }
    void filter(PointCloud<P> &out)
    {
        std::vector<orc::PointI> src(in->points.size()), dst;
        for (size_t i = 0; i < src.size(); ++i) { src[i].x = in->points[i].x; src[i].y = in->points[i].y; src[i].z = in->points[i].z; src[i].intensity = in->points[i].intensity; }
        orc::voxel_grid_xyzi(src.data(), int(src.size()), leaf, dst);
        out.points.resize(dst.size());
        for (size_t i = 0; i < dst.size(); ++i) { out.points[i].x = dst[i].x; out.points[i].y = dst[i].y; out.points[i].z = dst[i].z; out.points[i].intensity = dst[i].intensity; }
    }
};
}  // namespace pcl
namespace pcl {
namespace fields { struct intensity {}; }
namespace traits { template <typename P, typename F> struct has_field { static const bool value = true; }; }
template <typename P> struct KdTreeFLANN {                    // 3p: the oracle's exact k-NN
    typedef boost::shared_ptr<KdTreeFLANN<P>> Ptr;
    orc::KdTree tree;
    void setInputCloud(const PointCloud<P> &c) { tree.build(&c.points[0].x, sizeof(P) / sizeof(float), int(c.size())); }
    void setInputCloud(const typename PointCloud<P>::Ptr &c) { setInputCloud(*c); }
    int nearestKSearch(const P &p, int k, std::vector<int> &idx, std::vector<float> &sqd) const
    {
        const float q[3] = {p.x, p.y, p.z};
        idx.resize(size_t(k)); sqd.resize(size_t(k));          // pcl::KdTreeFLANN::nearestKSearch sizes its outputs itself (kdtree_flann.hpp)
        return tree.knn(q, k, idx.data(), sqd.data());
    }
};
}  // namespace pcl
struct NullLog { template <typename T> NullLog &operator<<(const T &) { return *this; } };
#define LOG(x) NullLog()
class Pose {                                                  // pose.h:37-66: the members and the methods this path uses; their bodies are pose.cpp's own lines (below)
public:
    Pose();
    Pose(const Pose &pose);
    Pose(const Eigen::Quaterniond &q, const Eigen::Vector3d &t, const double &td = 0);
    Pose(const Eigen::Matrix4d &T, const double &td = 0);
    void update();
    Pose inverse() const;
    Pose operator * (const Pose &pose);
    double td_;
    Eigen::Quaterniond q_;
    Eigen::Vector3d t_;
    Eigen::Matrix4d T_;
    Eigen::Matrix<double, 6, 6> cov_;
};
#include "../_ref/gen/pose_ctor_default.inc"                  // Pose::Pose()                                pose.cpp:16-23
#include "../_ref/gen/pose_ctor_copy.inc"                     // Pose::Pose(const Pose &)                    pose.cpp:25-32
#include "../_ref/gen/pose_ctor_qt.inc"                       // Pose::Pose(q, t, td)                        pose.cpp:34-41
#include "../_ref/gen/pose_ctor_T.inc"                        // Pose::Pose(T, td)                           pose.cpp:52-59
#include "../_ref/gen/pose_inverse_update.inc"                // Pose::inverse, Pose::update                 pose.cpp:99-108
#include "../_ref/gen/pose_mul.inc"                           // Pose::operator*                             pose.cpp:110-113
#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
#include "../_ref/gen/feature_structs.inc"                    // class PointPlaneFeature, class FeatureWithScore  (parameters.h:163-191)
float MIN_MATCH_SQ_DIS = 1.0f, MIN_PLANE_DIS = 0.2f;         // parameters.cpp:232-233
namespace common {
#include "../_ref/gen/sqr_sum.inc"                            // template <typename T> inline T sqrSum(x, y, z)
}
using namespace common;
#include "../_ref/gen/point_associate_to_map.inc"             // pointAssociateToMap

namespace ceres {
struct CostFunction {                                         // ceres/cost_function.h: the interface ResidualBlock::Evaluate calls
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    int num_residuals_ = 0;
};
template <int kNumResiduals, int... N> struct SizedCostFunction : CostFunction { SizedCostFunction() { num_residuals_ = kNumResiduals; } };
struct CRSMatrix {                                            // ceres/crs_matrix.h
    int num_rows = 0, num_cols = 0;
    std::vector<int> cols, rows;
    std::vector<double> values;
};
struct LocalParameterization {
    virtual ~LocalParameterization() {}
    virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};
}  // namespace ceres
typedef pcl::PointXYZI PointI;
typedef pcl::PointCloud<PointI> PointICloud;
typedef std::map<std::string, PointICloud> cloudFeature;      // parameters.h:161
struct ScanInfo { std::vector<int> scan_start_ind_, scan_end_ind_; bool segment_flag_ = true; };     // parameters.h:193-207 (the members read on this path)
double ROI_RANGE = 1.0;                                       // parameters.cpp:58
float SEGMENT_THETA = 1.047f;                                 // parameters.cpp:39
// A clock that advances by 0.1 us per reading: the selection loops read it once per round, so an ordinary selection (at most one round per feature) never reaches
// MAX_FEATURE_SELECT_TIME = 20 ms, while a loop that can only end through the cut-off (fps, every point visited, feature 1 unmatched: lidar_mapper.h:391-399)
// ends after 200 000 rounds instead of never.
struct TicToc { long reads = 0; double toc() { return 1e-4 * double(++reads); } };
#define ROS_WARN(...) do { } while (0)
int N_SCANS = 0;                                              // parameters.cpp global

#include "../_ref/gen/image_segmenter_class.inc"              // class ImageSegmenter { ... }
#include "../_ref/gen/image_segmenter_project.inc"            // template projectCloud
#include "../_ref/gen/image_segmenter_segment.inc"            // template segmentCloud
#include "../_ref/gen/image_segmenter_setparam.inc"           // ImageSegmenter::setParameter

#include "../_ref/gen/comp_object.inc"                        // class compObject
class FeatureExtract {
public:
    void extractCloud(const PointICloud &laser_cloud_in, const ScanInfo &scan_info, cloudFeature &cloud_feature);
#include "../_ref/gen/scan_match_decls.inc"                   // declarations of match{Corner,Surf}FromScan (feature_extract.hpp:78-90)
#include "../_ref/gen/match_point_decls.inc"                  // the declarations of match{Corner,Surf}PointFromMap (default arguments live here)
#include "../_ref/gen/match_batch_decls.inc"                  // ... and of the whole-cloud match{Corner,Surf}FromMap (feature_extract.hpp:92-108)
    typedef pcl::PointCloud<pcl::PointXYZ> PointCloud;        // common_header.h: the raw driver cloud
    void findStartEndAngle(const PointCloud &laser_cloud_in, float &start_ori, float &end_ori);      // feature_extract.hpp:60-62
    void calTimestamp(const PointCloud &laser_cloud_in, PointICloud &laser_cloud_out);               // feature_extract.hpp:68-69
};
typedef FeatureExtract::PointCloud PointCloud;
#include "../_ref/gen/extract_cloud.inc"                      // void FeatureExtract::extractCloud(...) { ... }
#include "../_ref/gen/match_corner_point.inc"                 // template <typename PointType> bool FeatureExtract::matchCornerPointFromMap(...)
#include "../_ref/gen/match_surf_point.inc"
#include "../_ref/gen/match_corner_batch.inc"                 // FeatureExtract::matchCornerFromMap (whole cloud; buildCalibMap's call, estimator.cpp:1143)   feature_extract.hpp:378-538
#include "../_ref/gen/match_surf_batch.inc"                   // FeatureExtract::matchSurfFromMap                                                          feature_extract.hpp:541-643
float SCAN_PERIOD = 0.1f, DISTANCE_SQ_THRESHOLD = 25.0f, NEARBY_SCAN = 2.5f;      // parameters.cpp:50-52 (set by the test entry points)
namespace pcl {
// pcl::copyPointCloud between clouds of the same point type, and from PointXYZ to PointXYZI (common/io.h: the fields both types have are copied, the rest keep
// their default -- intensity 0)
template <typename P> void copyPointCloud(const PointCloud<P> &in, PointCloud<P> &out) { out.points.assign(in.points.begin(), in.points.end()); out.width = in.width; out.height = in.height; }
inline void copyPointCloud(const PointCloud<PointXYZ> &in, PointCloud<PointXYZI> &out)
{
    out.points.resize(in.points.size());
    for (size_t i = 0; i < in.points.size(); ++i) { out.points[i].x = in.points[i].x; out.points[i].y = in.points[i].y; out.points[i].z = in.points[i].z; out.points[i].intensity = 0.f; }
    out.width = in.width; out.height = in.height;
}
}
#include "../_ref/gen/cal_timestamp.inc"                      // FeatureExtract::findStartEndAngle, calTimestamp(PointCloud)   feature_extract.cpp:54-114
#include "../_ref/gen/transform_start_end.inc"                // TransformToStart, TransformToEnd   utility.h:54-100
#include "../_ref/gen/match_from_scan.inc"                    // FeatureExtract::matchCornerFromScan / matchSurfFromScan   feature_extract.hpp:131-376

#include "../_ref/gen/utility_head.inc"                       // class Utility { public: deltaQ, skewSymmetric
};
#include "../_ref/gen/plane_factor_head.inc"                  // class LidarMapPlaneNormFactor ... Evaluate
#include "../_ref/gen/plane_factor_tail.inc"                  // private members, };
#include "../_ref/gen/edge_factor_head.inc"
    void check(double **) {}                                  // lidar_map_factor.hpp:176-229 (finite-difference print-out; scan2MapOptimization names it behind CHECK_JACOBIAN = 0)
#include "../_ref/gen/edge_factor_tail.inc"
#include "../_ref/gen/odom_plane_head.inc"                    // LidarPureOdomPlaneNormFactor
    void check(double **) {}                                  // lidar_pure_odom_factor.hpp:105-189 (finite-difference print-out; optimizeMap names it behind CHECK_JACOBIAN = 0)
#include "../_ref/gen/odom_plane_tail.inc"
#include "../_ref/gen/odom_edge_head.inc"                     // LidarPureOdomEdgeFactor
    void check(double **) {}                                  // lidar_pure_odom_factor.hpp:285-375
#include "../_ref/gen/odom_edge_tail.inc"
#include "../_ref/gen/calib_plane_head.inc"                   // LidarOnlineCalibPlaneNormFactor
#include "../_ref/gen/calib_plane_tail.inc"
#include "../_ref/gen/calib_edge_head.inc"                    // LidarOnlineCalibEdgeFactor
#include "../_ref/gen/calib_edge_tail.inc"
#include "../_ref/gen/scan_plane_head.inc"                    // LidarScanPlaneNormFactor   lidar_scan_factor.hpp:25-62, 122-126
#include "../_ref/gen/scan_plane_tail.inc"
#include "../_ref/gen/scan_edge_vec_head.inc"                 // LidarScanEdgeFactorVector  lidar_scan_factor.hpp:236-279, 339-343
    void check(double **) {}                                  // lidar_scan_factor.hpp:281-337 (as above; trackCloud names it behind CHECK_JACOBIAN = 0)
#include "../_ref/gen/scan_edge_vec_tail.inc"
#include "../_ref/gen/plp_class.inc"                          // class PoseLocalParameterization
#include "../_ref/gen/plp_plus.inc"                           // setParameter, Plus
#include "../_ref/gen/plp_jacobian.inc"                       // ComputeJacobian   pose_local_parameterization.cpp:47-55
Eigen::Matrix<double, 3, 3> COV_MEASUREMENT;                  // parameters.cpp:104 (set by the test entry point below)
#include "../_ref/gen/uct_compound.inc"                       // adjointMatrix, covop1, covop2, compoundPoseWithCov(pose_1, cov_1, pose_2, cov_2, pose_cp, cov_cp, method)
#include "../_ref/gen/uct_point_to_fs.inc"                    // inline Eigen::Matrix<double, 4, 6> pointToFS(const Eigen::Vector4d &)
#include "../_ref/gen/uct_eval_point_cov.inc"                 // evalPointUncertainty(pi, cov_point, pose, cov_pose)
#include "../_ref/gen/uct_eval_point.inc"                     // evalPointUncertainty(pi, cov_point, pose)   -- reads pose.cov_

// ---------------------------------------------------------------- ActiveFeatureSelection (lidar_mapper.h:126-573) from the reference's own lines
namespace pcl {
struct PointXYZIWithCov {                                     // point_with_cov.hpp:45-101: the fields and the two constructors this path uses (their own lines)
    float x, y, z, intensity;
    float cov_vec[6];
    float cov_trace;
#include "../_ref/gen/point_cov_ctor_default.inc"
#include "../_ref/gen/point_cov_ctor_from_point.inc"
};
}
typedef pcl::PointXYZIWithCov PointIWithCov;
typedef pcl::PointCloud<PointIWithCov> PointICovCloud;
namespace common {
#include "../_ref/gen/extract_cov.inc"                        // void extractCov(const pcl::PointXYZIWithCov &, Eigen::Matrix3d &)
#include "../_ref/gen/log_det.inc"                            // template logDet(M, use_cholesky): the LLT / PartialPivLU it calls are mini_eigen's
#include "../_ref/gen/rgi_head.inc"                           // struct RandomGeneratorInt { ..., geneRandUniform }  -- the tests re-seed m_random_engine
    };
}
namespace ceres {
struct LossFunction { virtual ~LossFunction() {} virtual void Evaluate(double sq_norm, double out[3]) const = 0; };
struct HuberLoss : LossFunction {                             // ceres/loss_function.cc (restated; its output is computed and discarded by evaluateFeatJacobianMatching)
    double a_, b_;
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    void Evaluate(double s, double rho[3]) const override
    {
        if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
        else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    }
};
}  // namespace ceres
#define SIZE_POSE 7                                           // parameters.h
#define LOG_EVERY_N(severity, n) NullLog()
#include "../_ref/gen/afs_limits.inc"                         // MAX_FEATURE_SELECT_TIME, MAX_RANDOM_QUEUE_TIME
FeatureExtract f_extract;                                     // lidar_mapper.h:92
class ActiveFeatureSelection {
public:
#include "../_ref/gen/afs_eval_jaco.inc"                      // evaluateFeatJacobianMatching   lidar_mapper.h:130-174
#include "../_ref/gen/afs_full_hessian.inc"                   // evalFullHessian                lidar_mapper.h:176-227
#define printf(...) ((void)0)                                 // its closing progress line (lidar_mapper.h:570) stays out of the test logs
#include "../_ref/gen/afs_gfm.inc"                            // goodFeatureMatching            lidar_mapper.h:229-573
#undef printf
    template <typename... A> void pubFeature(A &&...) {}      // lidar_mapper.h:575-606: rviz markers of the selected features (I/O, not on the path)
    ceres::LossFunction *loss_function_;                      // lidar_mapper.h:628-629
    common::RandomGeneratorInt<size_t> rgi_;
};

// ---------------------------------------------------------------- cloudUCTAssociateToMap, evalDegenracy (lidar_mapper_keyframe.cpp) from the reference's own lines
size_t NUM_OF_LASER = 2;                                      // parameters.cpp
bool with_ua_flag = true, is_degenerate = false;              // lidar_mapper_keyframe.cpp:130-131
double TRACE_THRESHOLD_MAPPING = 0.6, MAP_EIG_THRE = 100.0;   // parameters.cpp
std::vector<Eigen::Matrix<double, 1, 6>> d_factor_list;       // lidar_mapper_keyframe.cpp:118-121
std::vector<Eigen::Matrix<double, 6, 6>> d_eigvec_list;
Eigen::Matrix<double, 6, 6> mat_P;
namespace common {
#include "../_ref/gen/update_cov.inc"                         // void updateCov(pcl::PointXYZIWithCov &po, const Eigen::Matrix3d &)   point_with_cov.hpp:191-200
}
#include "../_ref/gen/uct_compound_pose.inc"                  // compoundPoseWithCov(pose_1, pose_2, pose_cp, method)   associate_uct.hpp:88-147
#include "../_ref/gen/cloud_uct_associate.inc"                // cloudUCTAssociateToMap                                   lidar_mapper_keyframe.cpp:1116-1158
#define INFO 0
#include "../_ref/gen/eval_degeneracy.inc"                    // evalDegenracy(mat_H, local_parameterization)            lidar_mapper_keyframe.cpp:1171-1204
// ---------------------------------------------------------------- VoxelGridCovarianceMLOAM<PointT>::applyFilter from the reference's own lines
// mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp:68-457 is the REFERENCE'S file (a PCL VoxelGrid rewritten for the covariance
// record: the trace gate, the (threshold - trace) weights, "the heaviest member's intensity", the LAST member's intensity of the plain branch, and
// the std::sort whose order inside a voxel decides both). What it calls into PCL / boost is supplied here: the Filter base's members
// (voxel_grid_covariance_mloam.h:71-390, pcl/filters/filter.h), getMinMax3D (pcl/common/impl/common.hpp), the field list of the two point types
// (POINT_CLOUD_REGISTER_POINT_STRUCT, point_with_cov.hpp:72-84), NdCopy{Point,Eigen}...Functor (pcl/common/impl/centroid.hpp: one float per
// registered field, in registration order), cloud_point_index_idx (pcl/filters/voxel_grid.h).
#include <cfloat>
#include <stdexcept>
#define PCL_WARN(...) do { } while (0)
#define pcl_isfinite(x) std::isfinite(x)
namespace boost { namespace mpl { template <typename FL> struct size { static const int value = FL::n; }; } }
namespace pcl {
struct PCLPointField { std::string name; unsigned offset; };
struct RGB { unsigned char b, g, r, a; };
struct PCLException : std::runtime_error { PCLException(const std::string &m, const char *, const char *) : std::runtime_error(m) {} };
struct cloud_point_index_idx {
    unsigned int idx, cloud_point_index;
    cloud_point_index_idx(unsigned int idx_, unsigned int cloud_point_index_) : idx(idx_), cloud_point_index(cloud_point_index_) {}
    bool operator<(const cloud_point_index_idx &p) const { return idx < p.idx; }
};
// registered fields of the two point types, in registration order
inline int field_count(const PointXYZI *) { return 4; }
inline float *field_ptr(PointXYZI &p, int i) { return i == 0 ? &p.x : (i == 1 ? &p.y : (i == 2 ? &p.z : &p.intensity)); }
inline const char *field_name(const PointXYZI *, int i) { static const char *n[] = {"x", "y", "z", "intensity"}; return n[i]; }
inline int field_count(const PointXYZIWithCov *) { return 11; }
inline float *field_ptr(PointXYZIWithCov &p, int i) { return i < 3 ? (&p.x + i) : (i == 3 ? &p.intensity : (i < 10 ? &p.cov_vec[i - 4] : &p.cov_trace)); }
inline const char *field_name(const PointXYZIWithCov *, int i)
{
    static const char *n[] = {"x", "y", "z", "intensity", "cov_xx", "cov_xy", "cov_xz", "cov_yy", "cov_yz", "cov_zz", "cov_trace"};
    return n[i];
}
template <typename P> struct FieldListOf;
template <> struct FieldListOf<PointXYZI> { static const int n = 4; };
template <> struct FieldListOf<PointXYZIWithCov> { static const int n = 11; };
template <typename P> int getFieldIndex(const PointCloud<P> &, const std::string &name, std::vector<PCLPointField> &fields)
{
    fields.clear();
    P probe;
    int found = -1;
    for (int i = 0; i < field_count((const P *)nullptr); ++i) {
        fields.push_back(PCLPointField{field_name((const P *)nullptr, i), unsigned(reinterpret_cast<char *>(field_ptr(probe, i)) - reinterpret_cast<char *>(&probe))});
        if (name == fields.back().name) found = i;
    }
    return found;
}
template <typename P> struct NdCopyPointEigenFunctor {
    const P &p; Eigen::VectorXf &v;
    NdCopyPointEigenFunctor(const P &p_, Eigen::VectorXf &v_) : p(p_), v(v_) {}
    void run() const { for (int i = 0; i < field_count((const P *)nullptr); ++i) v[i] = *field_ptr(const_cast<P &>(p), i); }
};
template <typename P> struct NdCopyEigenPointFunctor {
    const Eigen::VectorXf &v; P &p;
    NdCopyEigenPointFunctor(const Eigen::VectorXf &v_, P &p_) : v(v_), p(p_) {}
    void run() const { for (int i = 0; i < field_count((const P *)nullptr); ++i) *field_ptr(p, i) = v[i]; }
};
template <typename FL, typename F> void for_each_type(F f) { f.run(); }
// pcl/common/impl/common.hpp: component-wise min / max over the indexed points (all of them finite in a dense cloud)
template <typename P> void getMinMax3D(const PointCloud<P> &cloud, const std::vector<int> &indices, Eigen::Vector4f &min_pt, Eigen::Vector4f &max_pt)
{
    for (int k = 0; k < 4; ++k) { min_pt[k] = FLT_MAX; max_pt[k] = -FLT_MAX; }
    for (int i : indices) {
        const P &q = cloud.points[size_t(i)];
        if (!cloud.is_dense && (!std::isfinite(q.x) || !std::isfinite(q.y) || !std::isfinite(q.z))) continue;
        const float a[4] = {q.x, q.y, q.z, 1.f};
        for (int k = 0; k < 4; ++k) { min_pt[k] = std::min(min_pt[k], a[k]); max_pt[k] = std::max(max_pt[k], a[k]); }
    }
}
template <typename P> void getMinMax3D(const typename PointCloud<P>::Ptr &, const std::vector<int> &, const std::string &, float, float, Eigen::Vector4f &, Eigen::Vector4f &, bool)
{
    throw std::logic_error("filter-field path of getMinMax3D: not on the mapper's path (filter_field_name_ stays empty)");
}
template <typename PointT> class VoxelGridCovarianceMLOAM {   // the members applyFilter reads (voxel_grid_covariance_mloam.h:71-390, pcl/filters/filter.h)
public:
    typedef pcl::PointCloud<PointT> PointCloud;
    typedef FieldListOf<PointT> FieldList;
    VoxelGridCovarianceMLOAM() : inverse_leaf_size_(), downsample_all_data_(true), save_leaf_layout_(false), filter_limit_min_(-FLT_MAX), filter_limit_max_(FLT_MAX),
                                 filter_limit_negative_(false), min_points_per_voxel_(0), trace_threshold_(2.0) {}          // voxel_grid_covariance_mloam.h:88-102
    void setInputCloud(const typename PointCloud::Ptr &c)
    {
        input_ = c;
        indices_.reset(new std::vector<int>(c->points.size()));
        for (size_t i = 0; i < c->points.size(); ++i) (*indices_)[i] = int(i);
    }
    void setLeafSize(float lx, float ly, float lz)             // voxel_grid_covariance_mloam.h:118-127: inverse_leaf_size_ = Array4f::Ones() / leaf_size_.array()
    {
        const float leaf[4] = {lx, ly, lz, 1.f};
        for (int k = 0; k < 4; ++k) inverse_leaf_size_[k] = 1.f / leaf[k];
    }
    void setTraceThreshold(const float trace_threshold) { trace_threshold_ = trace_threshold; }
    void filter(PointCloud &output) { applyFilter(output); }
    std::string getClassName() const { return "VoxelGridCovarianceMLOAM"; }
protected:
    void applyFilter(PointCloud &output);
    typename PointCloud::Ptr input_;
    boost::shared_ptr<std::vector<int>> indices_;
    Eigen::Vector4f inverse_leaf_size_;
    bool downsample_all_data_, save_leaf_layout_;
    std::vector<int> leaf_layout_;
    Eigen::Vector4i min_b_, max_b_, div_b_, divb_mul_;
    std::string filter_field_name_;
    double filter_limit_min_, filter_limit_max_;
    bool filter_limit_negative_;
    unsigned int min_points_per_voxel_;
    float trace_threshold_;
};
}  // namespace pcl
#include "../_ref/gen/voxel_filter_apply.inc"                 // VoxelGridCovarianceMLOAM<PointT>::applyFilter   voxel_grid_covariance_mloam_impl.hpp:68-457
// downsampleCurrentScan (lidar_mapper_keyframe.cpp:356-421) now runs on that filter: everything on its path is the reference's text
#include "../uct.hpp"
PointICloud::Ptr laser_cloud_surf_last(new PointICloud()), laser_cloud_corner_last(new PointICloud()), laser_cloud_outlier(new PointICloud());      // lidar_mapper_keyframe.cpp:41-57
PointICloud::Ptr laser_cloud_surf_last_ds(new PointICloud()), laser_cloud_corner_last_ds(new PointICloud()), laser_cloud_outlier_ds(new PointICloud());
PointICovCloud::Ptr laser_cloud_surf_cov(new PointICovCloud()), laser_cloud_corner_cov(new PointICovCloud()), laser_cloud_outlier_cov(new PointICovCloud());
std::vector<Pose> pose_ext;
pcl::VoxelGridCovarianceMLOAM<PointI> down_size_filter_surf, down_size_filter_corner, down_size_filter_outlier;      // lidar_mapper_keyframe.cpp:79-81
#include "../_ref/gen/downsample_current_scan.inc"

// ---------------------------------------------------------------- Estimator::evalDegenracy (estimator.cpp:1598-1680) from the reference's own lines
// The odometry window's degeneracy / calibration policy on J^T J: the window's pose blocks by the mapper's rule with per-block thresholds, the
// extrinsic blocks by lambda_min / N_CUMU_FEATURE against LAMBDA_THRE_CALIB and the running eig_thre_. Ceres hands it the Jacobian as a
// CRSMatrix; the struct below has ceres/crs_matrix.h's fields.
#include "../_ref/gen/crs_to_sparse.inc"                      // CRSMatrix2EigenMatrix(crs, Eigen::SparseMatrix<T, RowMajor> &)   utility.h:152-166
int ESTIMATE_EXTRINSIC = 1, OPT_WINDOW_SIZE = 4, N_CUMU_FEATURE = 10;      // parameters.cpp
double LAMBDA_THRE_CALIB = 70.0;
namespace ceres { struct Problem; namespace internal { struct ResidualBlock; } }
class MarginalizationInfo;                                    // marginalization_factor.h: out of scope; only the pointer type is named here
class Estimator {                                             // the members evalDegenracy and optimizeMap (up to its marginalisation section) touch (estimator.h:85-230)
public:
    void evalDegenracy(std::vector<PoseLocalParameterization *> &local_param_ids, const ceres::CRSMatrix &jaco);
    void optimizeMap();
    void vector2Double();
    void double2Vector();
    void buildCalibMap() {}                                   // estimator.cpp:1067-1157 / 1159-1268: matching against the window map; the test entry fills *_map_features_
    void buildLocalMap() {}                                   //   and sel_*_feature_idx_ itself (those steps are pinned through match*FromMap and goodFeatureMatching above)
    void evalResidual(ceres::Problem &problem, std::vector<PoseLocalParameterization *> &local_param_ids, const std::vector<double *> &para_ids,
                      const std::vector<ceres::internal::ResidualBlock *> &res_ids_proj, const MarginalizationInfo *last_marginalization_info_,
                      const std::vector<ceres::internal::ResidualBlock *> &res_ids_marg);
    void evaluateFeatJacobian(const Pose &pose_pivot, const Pose &pose_i, const Pose &pose_ext, PointPlaneFeature &feature);          // estimator.cpp:1273-1345
    void goodFeatureMatching(const pcl::KdTreeFLANN<PointI>::Ptr &kdtree_from_map, const PointICloud &laser_map, const PointICloud &laser_cloud,
                             std::vector<PointPlaneFeature> &all_features, std::vector<size_t> &sel_feature_idx, const char feature_type, const Pose &pose_pivot,
                             const Pose &pose_i, const Pose &pose_ext, const double &gf_ratio);                                     // estimator.cpp:1347-1517
    FeatureExtract f_extract_;                                // estimator.h
    common::RandomGeneratorInt<size_t> rgi_;
    std::vector<Eigen::Quaterniond> Qs_;                      // CircularBuffer<> in the reference (estimator.h:166-167): only indexed here
    std::vector<Eigen::Vector3d> Ts_;
    std::vector<std::vector<std::vector<PointPlaneFeature>>> surf_map_features_, corner_map_features_;
    std::vector<std::vector<PointPlaneFeature>> cumu_surf_map_features_, cumu_corner_map_features_;
    std::vector<std::vector<std::vector<size_t>>> sel_surf_feature_idx_, sel_corner_feature_idx_;
    double **para_pose_{};
    double **para_ex_pose_{};
    MarginalizationInfo *last_marginalization_info_{};
    std::vector<double *> last_marginalization_parameter_blocks_;
    std::vector<Eigen::Quaterniond> qbl_;
    std::vector<Eigen::Vector3d> tbl_;
    int frame_cnt_{};
    Eigen::VectorXd eig_thre_;
    std::vector<double> log_lambda_;
    std::vector<Pose> log_extrinsics_;
    std::vector<double> d_factor_calib_;
};
#define printf(...) ((void)0)                                 // "%lu: calib eig is %f" stays out of the test logs; std::cout is pointed at a null buffer by the caller
#include "../_ref/gen/estimator_eval_degeneracy.inc"
#undef printf
#include "../_ref/gen/eval_hessian.inc"                         // evalHessian(jaco, mat_H)   lidar_mapper_keyframe.cpp:1160-1169

// ---------------------------------------------------------------- ceres::Problem / ceres::Solve as the driver loops use them
// A Ceres-SHAPED interface (AddParameterBlock / AddResidualBlock / Evaluate -> CRSMatrix / Solve / Summary) over the reference's own cost functions and
// its own PoseLocalParameterization: a residual block is evaluated by cost_function->Evaluate (the reference's lines), corrected by the loss as
// ceres::internal::ResidualBlock::Evaluate + Corrector do (rho'' <= 0 for Huber: rows and residuals scaled by sqrt(rho')), moved to the local
// parameterisation by J_global * ComputeJacobian() (the reference's lines), accumulated in block order; the minimiser is the oracle's restatement of
// Ceres 1.12's trust-region LM (oracle/lm.hpp) with x (+) delta = the reference's Plus. So under scan2MapOptimization / trackCloud everything but the
// LM iteration itself and the kd-tree is the reference's text.
#include "../lm.hpp"
namespace ceres {
namespace internal { struct ResidualBlock { CostFunction *cost; LossFunction *loss; std::vector<double *> params; }; }
enum LinearSolverType { DENSE_NORMAL_CHOLESKY, DENSE_QR, SPARSE_NORMAL_CHOLESKY, DENSE_SCHUR, SPARSE_SCHUR };
static CRSMatrix g_last_jacobian;                             // what the last Problem::Evaluate produced (read by the test entry points)
static double g_last_cost = 0.0;
struct Problem {
    struct EvaluateOptions {
        std::vector<double *> parameter_blocks;
        std::vector<internal::ResidualBlock *> residual_blocks;
        bool apply_loss_function = true;
    };
    // parameter blocks in program order (the order of AddParameterBlock; a block first seen in AddResidualBlock is appended, as Ceres does)
    struct ParamBlock { double *values; int size; LocalParameterization *lp; bool constant; };
    std::vector<ParamBlock> params;
    std::vector<std::unique_ptr<internal::ResidualBlock>> blocks;
    std::vector<std::unique_ptr<CostFunction>> owned_costs;       // Problem takes ownership (ceres/problem.h)
    static constexpr int kLocal = 6, kGlobal = 7;                 // every block of this path is a pose with the reference's PoseLocalParameterization
    int find(const double *v) const
    {
        for (size_t i = 0; i < params.size(); ++i) if (params[i].values == v) return int(i);
        return -1;
    }
    void AddParameterBlock(double *values, int size, LocalParameterization *local_parameterization = nullptr)
    {
        const int i = find(values);
        if (i >= 0) { if (local_parameterization) params[size_t(i)].lp = local_parameterization; return; }
        params.push_back(ParamBlock{values, size, local_parameterization, false});
    }
    void SetParameterBlockConstant(double *values) { AddParameterBlock(values, kGlobal); params[size_t(find(values))].constant = true; }
    internal::ResidualBlock *AddResidualBlock(CostFunction *cost_function, LossFunction *loss_function, const std::vector<double *> &xs)
    {
        for (double *x : xs) AddParameterBlock(x, kGlobal);
        blocks.emplace_back(new internal::ResidualBlock{cost_function, loss_function, xs});
        owned_costs.emplace_back(cost_function);
        return blocks.back().get();
    }
    internal::ResidualBlock *AddResidualBlock(CostFunction *c, LossFunction *l, double *x0) { return AddResidualBlock(c, l, std::vector<double *>{x0}); }
    internal::ResidualBlock *AddResidualBlock(CostFunction *c, LossFunction *l, double *x0, double *x1, double *x2) { return AddResidualBlock(c, l, std::vector<double *>{x0, x1, x2}); }
    // One residual block with parameter block p's values taken from at[p]: loss-corrected residuals r[nres] and, per parameter of the block, the LOCAL
    // Jacobian rows Jl[q][nres][6] (left untouched for constant blocks: ceres::internal::ResidualBlock::Evaluate hands the cost function a null pointer
    // for those, and the Jacobian writers skip them). Returns 0.5 * rho(|r|^2).
    double evaluate_block(const internal::ResidualBlock &b, const std::vector<const double *> &at, double *r, double (*Jl)[3 * kLocal], bool apply_loss) const
    {
        const int nres = b.cost->num_residuals_, np = int(b.params.size());
        double Jg[3][3 * kGlobal];
        const double *pv[3];
        double *jac[3];
        int pi[3];
        for (int q = 0; q < np; ++q) {
            pi[q] = find(b.params[size_t(q)]);
            pv[q] = at[size_t(pi[q])];
            jac[q] = params[size_t(pi[q])].constant ? nullptr : Jg[q];
        }
        b.cost->Evaluate(pv, r, jac);
        double sq = 0.0;
        for (int q = 0; q < nres; ++q) sq += r[q] * r[q];
        double rho[3] = {sq, 1.0, 0.0};
        if (apply_loss && b.loss) b.loss->Evaluate(sq, rho);
        const double s = std::sqrt(rho[1]);                      // Corrector: sq_norm == 0 or rho[2] <= 0 -> both scalings are sqrt(rho[1])
        for (int q = 0; q < np; ++q) {
            if (!jac[q]) continue;
            double P[kGlobal * kLocal];
            params[size_t(pi[q])].lp->ComputeJacobian(pv[q], P);
            for (int a = 0; a < nres; ++a)
                for (int k = 0; k < kLocal; ++k) {
                    double acc = 0.0;
                    for (int c = 0; c < kGlobal; ++c) acc += (Jg[q][a * kGlobal + c] * s) * P[c * kLocal + k];
                    Jl[q][a * kLocal + k] = acc;
                }
        }
        for (int a = 0; a < nres; ++a) r[a] *= s;
        return 0.5 * rho[0];
    }
    // the variable blocks in program order: what Program::RemoveFixedBlocks leaves for the minimiser -- neither constant nor without a residual block
    // (a block nothing depends on is dropped from the reduced program and keeps its value bit for bit)
    std::vector<int> free_blocks() const
    {
        std::vector<char> used(params.size(), 0);
        for (const auto &b : blocks) for (double *x : b->params) used[size_t(find(x))] = 1;
        std::vector<int> f;
        for (size_t i = 0; i < params.size(); ++i) if (!params[i].constant && used[i]) f.push_back(int(i));
        return f;
    }
    // J^T J / J^T r / cost over all residual blocks with the variable blocks' values taken from x (kGlobal values per variable block, program order)
    void normal_equations(const double *x, orc::NormalEqDyn &ne) const
    {
        const std::vector<int> fb = free_blocks();
        std::vector<int> slot(params.size(), -1);
        std::vector<const double *> at(params.size());
        for (size_t i = 0; i < params.size(); ++i) at[i] = params[i].values;
        for (size_t k = 0; k < fb.size(); ++k) { slot[size_t(fb[k])] = int(k); at[size_t(fb[k])] = x + k * kGlobal; }
        const size_t N = fb.size() * kLocal;
        ne.H.assign(N * N, 0.0); ne.g.assign(N, 0.0); ne.cost = 0.0; ne.count = 0;
        for (const auto &b : blocks) {
            double r[3], Jl[3][3 * kLocal];
            ne.cost += evaluate_block(*b, at, r, Jl, true);
            ne.count++;
            const int np = int(b->params.size());
            for (int q = 0; q < b->cost->num_residuals_; ++q)
                for (int pa = 0; pa < np; ++pa) {
                    const int sa = slot[size_t(find(b->params[size_t(pa)]))];
                    if (sa < 0) continue;
                    const double *ja = Jl[pa] + q * kLocal;
                    for (int k = 0; k < kLocal; ++k) ne.g[size_t(sa) * kLocal + k] += ja[k] * r[q];
                    for (int pb = 0; pb < np; ++pb) {
                        const int sb = slot[size_t(find(b->params[size_t(pb)]))];
                        if (sb < 0) continue;
                        const double *jb = Jl[pb] + q * kLocal;
                        for (int a = 0; a < kLocal; ++a) for (int c = 0; c < kLocal; ++c) ne.H[(size_t(sa) * kLocal + a) * N + size_t(sb) * kLocal + c] += ja[a] * jb[c];
                    }
                }
        }
    }
    // Problem::Evaluate (problem_impl.cc): columns = the LISTED parameter blocks in the listed order (all blocks, program order, when the list is empty), LocalSize
    // each; a constant block keeps its columns but gets no entries; a row's entries ascend by column (CompressedRowJacobianWriter sorts the blocks of a row)
    bool Evaluate(const EvaluateOptions &o, double *cost, std::vector<double> *residuals, std::vector<double> *gradient, CRSMatrix *jacobian) const
    {
        std::vector<int> col_of(params.size(), -1);
        int ncols = 0;
        if (o.parameter_blocks.empty()) for (size_t i = 0; i < params.size(); ++i) { col_of[i] = ncols; ncols += kLocal; }
        else for (double *p : o.parameter_blocks) { col_of[size_t(find(p))] = ncols; ncols += kLocal; }
        std::vector<const double *> at(params.size());
        for (size_t i = 0; i < params.size(); ++i) at[i] = params[i].values;
        double total = 0.0;
        if (jacobian) { jacobian->num_rows = 0; jacobian->num_cols = ncols; jacobian->rows.assign(1, 0); jacobian->cols.clear(); jacobian->values.clear(); }
        if (residuals) residuals->clear();
        if (gradient) gradient->assign(size_t(ncols), 0.0);
        std::vector<internal::ResidualBlock *> all;
        if (o.residual_blocks.empty()) for (const auto &b : blocks) all.push_back(b.get());
        for (const internal::ResidualBlock *b : (o.residual_blocks.empty() ? all : o.residual_blocks)) {
            double r[3], Jl[3][3 * kLocal];
            total += evaluate_block(*b, at, r, Jl, o.apply_loss_function);
            const int np = int(b->params.size());
            int order[3] = {0, 1, 2};
            std::sort(order, order + np, [&](int a, int c) { return col_of[size_t(find(b->params[size_t(a)]))] < col_of[size_t(find(b->params[size_t(c)]))]; });
            for (int q = 0; q < b->cost->num_residuals_; ++q) {
                if (residuals) residuals->push_back(r[q]);
                for (int oi = 0; oi < np; ++oi) {
                    const int pa = order[oi], pidx = find(b->params[size_t(pa)]), c0 = col_of[size_t(pidx)];
                    if (params[size_t(pidx)].constant || c0 < 0) continue;
                    for (int k = 0; k < kLocal; ++k) {
                        if (gradient) (*gradient)[size_t(c0 + k)] += Jl[pa][q * kLocal + k] * r[q];
                        if (jacobian) { jacobian->cols.push_back(c0 + k); jacobian->values.push_back(Jl[pa][q * kLocal + k]); }
                    }
                }
                if (jacobian) { jacobian->rows.push_back(int(jacobian->values.size())); jacobian->num_rows++; }
            }
        }
        if (cost) *cost = total;
        if (jacobian) g_last_jacobian = *jacobian;
        g_last_cost = total;
        return true;
    }
};
struct Solver {
    struct Options {
        LinearSolverType linear_solver_type = SPARSE_NORMAL_CHOLESKY;
        int max_num_iterations = 50;
        bool minimizer_progress_to_stdout = false, check_gradients = false, update_state_every_iteration = false;
        double gradient_check_relative_precision = 1e-8, max_solver_time_in_seconds = 1e9;
    };
    struct Summary {
        orc::SolveSummary s;
        int num_residual_blocks = 0;
        std::string BriefReport() const { return "Ceres-shaped solver report: iterations " + std::to_string(s.num_iterations); }
        std::string FullReport() const { return BriefReport(); }
    };
};
std::vector<Solver::Summary> g_solve_log;                     // every ceres::Solve of a driver-loop call, in order (read by the test entry points)
void Solve(const Solver::Options &options, Problem *problem, Solver::Summary *summary)
{
    summary->num_residual_blocks = int(problem->blocks.size());
    const std::vector<int> fb = problem->free_blocks();
    const int G = Problem::kGlobal, L = Problem::kLocal;
    std::vector<double> x(fb.size() * size_t(G));
    for (size_t k = 0; k < fb.size(); ++k) std::memcpy(&x[k * size_t(G)], problem->params[size_t(fb[k])].values, sizeof(double) * size_t(G));
    if (!fb.empty())
        orc::ceres_like_solve_dyn([&](const double *at, orc::NormalEqDyn &ne) { problem->normal_equations(at, ne); },
                                  [&](const double *at, const double *delta, double *out) {
                                      for (size_t k = 0; k < fb.size(); ++k) problem->params[size_t(fb[k])].lp->Plus(at + k * size_t(G), delta + k * size_t(L), out + k * size_t(G));
                                  },
                                  x.data(), int(fb.size()) * L, int(fb.size()) * G, options.max_num_iterations, summary->s);
    for (size_t k = 0; k < fb.size(); ++k) std::memcpy(problem->params[size_t(fb[k])].values, &x[k * size_t(G)], sizeof(double) * size_t(G));
    g_solve_log.push_back(*summary);
}
}  // namespace ceres

// ---------------------------------------------------------------- scan2MapOptimization (lidar_mapper_keyframe.cpp:423-639) from the reference's own lines
namespace common {
const std::string YELLOW("\033[1;33m"), RESET("\033[0m");     // color.hpp:52, 55
// timing.hpp:180-192 (wall-clock bookkeeping). GetCountTime: seconds; advances by 10 ns per reading -- the odometry's selection reads it two or three times per draw,
// so an ordinary selection stays far below its 7 ms cut-off while a loop that can only end through the cut-off (nothing left to draw) ends after 700 000 readings
namespace timing { struct Timer { long reads = 0; explicit Timer(const std::string &) {} double Stop() { return 0.0; } double GetCountTime() { return 1e-8 * double(++reads); } }; }
}
std::ostream &operator<<(std::ostream &o, const Pose &p) { return o << "t: [" << p.t_(0) << " " << p.t_(1) << " " << p.t_(2) << "]"; }   // pose.cpp:110-117 (printing)
PointICovCloud::Ptr laser_cloud_surf_from_map_cov_ds(new PointICovCloud()), laser_cloud_corner_from_map_cov_ds(new PointICovCloud());   // lidar_mapper_keyframe.cpp:66-67
pcl::KdTreeFLANN<PointIWithCov>::Ptr kdtree_surf_from_map(new pcl::KdTreeFLANN<PointIWithCov>()), kdtree_corner_from_map(new pcl::KdTreeFLANN<PointIWithCov>());   // :73-74
ActiveFeatureSelection afs;                                   // :133
double para_pose[SIZE_POSE];                                  // :103
Pose pose_wmap_curr, pose_wmap_wodom, pose_wodom_curr;        // :96-98 (the three poses transformAssociateToMap / transformUpdate chain)
#include "../_ref/gen/transform_associate_update.inc"         // transformAssociateToMap, transformUpdate   lidar_mapper_keyframe.cpp:145-160
int frame_cnt = 0, CHECK_JACOBIAN = 0;                        // :30, parameters.cpp
int POINT_PLANE_FACTOR = 1, POINT_EDGE_FACTOR = 1;            // parameters.cpp (config point_plane_factor / point_edge_factor)
std::string FLAGS_gf_method = "wo_gf";                        // lidar_mapper_keyframe.cpp:20-22 (gflags)
double FLAGS_gf_ratio_ini = 0.2, gf_ratio_cur = 1.0, MAP_DEG_THRE = 42.0;
std::vector<double> gf_deg_factor_list, gf_logdet_H_list;     // :123-124
std::vector<std::vector<double>> mapping_sp_list;             // :127
Eigen::Matrix<double, 6, 6> cov_mapping;                      // :99
std::vector<std::pair<double, Pose>> pose_keyframes_6d;       // :60
int pub_good_surf_feature = 0;                                // ros::Publisher :115
double time_laser_odometry = 0.0;                             // :36
#include "../_ref/gen/vector2double.inc"                      // vector2Double, double2Vector   lidar_mapper_keyframe.cpp:236-252
#define printf(...) ((void)0)
#include "../_ref/gen/scan2map_optimization.inc"              // scan2MapOptimization           lidar_mapper_keyframe.cpp:423-639
#undef printf

// ---------------------------------------------------------------- saveKeyframe (lidar_mapper_keyframe.cpp:641-683) from the reference's own lines
bool save_new_keyframe = false;                               // :64
PointI pose_point_cur, pose_point_prev;                       // :93
Eigen::Quaterniond q_ori_cur, q_ori_prev;                     // :94
float DISTANCE_KEYFRAMES = 1.0f, ORIENTATION_KEYFRAMES = 1.0f; // parameters.cpp:98-99
PointICloud::Ptr pose_keyframes_3d(new PointICloud());        // :59
std::vector<PointICovCloud::Ptr> surf_cloud_keyframes_cov, corner_cloud_keyframes_cov, outlier_cloud_keyframes_cov;   // :74-76
#define printf(...) ((void)0)
#include "../_ref/gen/save_keyframe.inc"
#undef printf

// ---------------------------------------------------------------- LidarTracker::trackCloud (lidar_tracker.cpp:23-129) from the reference's own lines
typedef PointICloud::Ptr PointICloudPtr;                      // common_header.h
class LidarTracker {                                          // lidar_tracker.h:44-52
public:
    Pose trackCloud(const cloudFeature &prev_cloud_feature, const cloudFeature &cur_cloud_feature, const Pose &pose_ini);
    FeatureExtract f_extract_;
};
#define printf(...) ((void)0)
#include "../_ref/gen/track_cloud.inc"
#undef printf

// ---------------------------------------------------------------- Estimator::goodFeatureMatching + evaluateFeatJacobian (estimator.cpp:1273-1517) from the reference's own lines
#undef MAX_FEATURE_SELECT_TIME
#undef MAX_RANDOM_QUEUE_TIME
#define MAX_FEATURE_SELECT_TIME 7                             // estimator.h:62-63 (the mapper's limits, lidar_mapper.h:82-83, are 20 / 20)
#define MAX_RANDOM_QUEUE_TIME 10
#include "../_ref/gen/estimator_eval_feat_jacobian.inc"
#include "../_ref/gen/estimator_gfm.inc"

// ---------------------------------------------------------------- C API for the tests
extern "C" {
// FeatureExtract::calTimestamp on a raw cloud (n x 3 floats) -> the relative times it writes into the intensity field
int ref_cal_timestamp(const float *xyz, int n, float scan_period, float *rel_time)
{
    SCAN_PERIOD = scan_period;
    FeatureExtract::PointCloud in;
    in.points.resize(size_t(n));
    for (int i = 0; i < n; ++i) { in.points[size_t(i)].x = xyz[3 * i]; in.points[size_t(i)].y = xyz[3 * i + 1]; in.points[size_t(i)].z = xyz[3 * i + 2]; }
    PointICloud out;
    FeatureExtract fe;
    fe.calTimestamp(in, out);
    for (int i = 0; i < n; ++i) rel_time[i] = out.points[size_t(i)].intensity;
    return int(out.size());
}

// saveKeyframe over a sequence of mapper poses (n x 7: t, q): saved[i] = 1 where the reference saves frame i as a keyframe; from a clean state
int ref_save_keyframes(const double *poses7, int n, float distance_keyframes, float orientation_keyframes, unsigned char *saved)
{
    DISTANCE_KEYFRAMES = distance_keyframes; ORIENTATION_KEYFRAMES = orientation_keyframes;
    pose_keyframes_3d->points.clear(); pose_keyframes_6d.clear();
    surf_cloud_keyframes_cov.clear(); corner_cloud_keyframes_cov.clear(); outlier_cloud_keyframes_cov.clear();
    pose_point_prev = PointI(); q_ori_prev = Eigen::Quaterniond::Identity();
    for (int i = 0; i < n; ++i) {
        const double *p = poses7 + 7 * i;
        pose_wmap_curr = Pose(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2]));
        saveKeyframe();
        saved[i] = save_new_keyframe ? 1 : 0;
    }
    const int n_kf = int(pose_keyframes_3d->size());
    pose_keyframes_3d->points.clear(); pose_keyframes_6d.clear();
    surf_cloud_keyframes_cov.clear(); corner_cloud_keyframes_cov.clear(); outlier_cloud_keyframes_cov.clear();
    return n_kf;
}

// FeatureExtract::matchCornerFromMap / matchSurfFromMap (the whole-cloud forms buildCalibMap calls, estimator.cpp:1136-1150): features out as idx, coeffs[6]
int ref_match_cloud(char kind, const float *map4, int n_map, const float *feat4, int n_feat, const double pose7[7], int n_neigh, int check_fov, float min_match_sq_dis,
                    float min_plane_dis, int *idx_out, double *coeffs_out, int *n_out)
{
    MIN_MATCH_SQ_DIS = min_match_sq_dis; MIN_PLANE_DIS = min_plane_dis;
    PointICloud map, feat;
    auto fill = [](PointICloud &c, const float *p, int n) {
        c.points.resize(size_t(n));
        for (int i = 0; i < n; ++i) { c.points[size_t(i)].x = p[4 * i]; c.points[size_t(i)].y = p[4 * i + 1]; c.points[size_t(i)].z = p[4 * i + 2]; c.points[size_t(i)].intensity = p[4 * i + 3]; }
    };
    fill(map, map4, n_map); fill(feat, feat4, n_feat);
    pcl::KdTreeFLANN<PointI>::Ptr kd(new pcl::KdTreeFLANN<PointI>());
    kd->setInputCloud(PointICloud::Ptr(new PointICloud(map)));
    const Pose pose(Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]), Eigen::Vector3d(pose7[0], pose7[1], pose7[2]));
    FeatureExtract fe;
    std::vector<PointPlaneFeature> features;
    if (kind == 's') fe.matchSurfFromMap<PointI>(kd, map, feat, pose, features, size_t(n_neigh), check_fov != 0);
    else fe.matchCornerFromMap<PointI>(kd, map, feat, pose, features, size_t(n_neigh), check_fov != 0);
    *n_out = int(features.size());
    for (size_t i = 0; i < features.size(); ++i) {
        idx_out[i] = int(features[i].idx_);
        for (int k = 0; k < 6; ++k) coeffs_out[i * 6 + size_t(k)] = k < int(features[i].coeffs_.size()) ? features[i].coeffs_(k) : 0.0;
    }
    return 0;
}

// Estimator::goodFeatureMatching on one (frame, LiDAR) group: map / features as n x 4 floats; the three poses as [t, q]; gf_ratio = ODOM_GF_RATIO (a float in the
// reference, widened at the call, estimator.cpp:1250). rel_out: the Pose the function matches at, Pose(T_pivot^-1 T_i T_ext), as this build computes it.
int ref_odom_good_feature_matching(char kind, const float *map4, int n_map, const float *feat4, int n_feat, const double pivot7[7], const double posei7[7],
                                   const double ext7[7], float gf_ratio, unsigned seed, float min_match_sq_dis, float min_plane_dis, int *sel, int *n_sel,
                                   double rel_out[7])
{
    MIN_MATCH_SQ_DIS = min_match_sq_dis; MIN_PLANE_DIS = min_plane_dis;
    PointICloud map, feat;
    auto fill = [](PointICloud &c, const float *p, int n) {
        c.points.resize(size_t(n));
        for (int i = 0; i < n; ++i) { c.points[size_t(i)].x = p[4 * i]; c.points[size_t(i)].y = p[4 * i + 1]; c.points[size_t(i)].z = p[4 * i + 2]; c.points[size_t(i)].intensity = p[4 * i + 3]; }
    };
    fill(map, map4, n_map); fill(feat, feat4, n_feat);
    pcl::KdTreeFLANN<PointI>::Ptr kd(new pcl::KdTreeFLANN<PointI>());
    kd->setInputCloud(PointICloud::Ptr(new PointICloud(map)));
    auto mk = [](const double *p) { return Pose(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2])); };
    const Pose pose_pivot = mk(pivot7), pose_i = mk(posei7), pose_ext = mk(ext7);
    Estimator est;
    est.rgi_.m_random_engine.seed(seed);
    std::vector<PointPlaneFeature> all_features;
    std::vector<size_t> sel_idx;
    const double ratio = gf_ratio;                            // ODOM_GF_RATIO (float) -> const double &
    est.goodFeatureMatching(kd, map, feat, all_features, sel_idx, kind, pose_pivot, pose_i, pose_ext, ratio);
    *n_sel = int(sel_idx.size());
    for (size_t i = 0; i < sel_idx.size(); ++i) sel[i] = int(sel_idx[i]);
    const Pose pose_local(pose_pivot.T_.inverse() * pose_i.T_ * pose_ext.T_);
    rel_out[0] = pose_local.t_(0); rel_out[1] = pose_local.t_(1); rel_out[2] = pose_local.t_(2);
    rel_out[3] = pose_local.q_.x(); rel_out[4] = pose_local.q_.y(); rel_out[5] = pose_local.q_.z(); rel_out[6] = pose_local.q_.w();
    return 0;
}

// clouds out: sharp, less_sharp, flat, less_flat (voxel-thinned), each n x 4 floats; counts in n_out[4]
int ref_extract_cloud(const float *xyzi, int n, const int *scan_start, const int *scan_end, int n_scans, float *out[4], int n_out[4])
{
    PointICloud in;
    in.points.resize(size_t(n));
    for (int i = 0; i < n; ++i) { in.points[i].x = xyzi[4 * i]; in.points[i].y = xyzi[4 * i + 1]; in.points[i].z = xyzi[4 * i + 2]; in.points[i].intensity = xyzi[4 * i + 3]; }
    ScanInfo si;
    si.scan_start_ind_.assign(scan_start, scan_start + n_scans);
    si.scan_end_ind_.assign(scan_end, scan_end + n_scans);
    N_SCANS = n_scans;
    cloudFeature cf;
    FeatureExtract f;
    f.extractCloud(in, si, cf);
    const char *keys[4] = {"corner_points_sharp", "corner_points_less_sharp", "surf_points_flat", "surf_points_less_flat"};
    for (int k = 0; k < 4; ++k) {
        const PointICloud &c = cf[keys[k]];
        n_out[k] = int(c.size());
        for (size_t i = 0; i < c.size(); ++i) { out[k][4 * i] = c.points[i].x; out[k][4 * i + 1] = c.points[i].y; out[k][4 * i + 2] = c.points[i].z; out[k][4 * i + 3] = c.points[i].intensity; }
    }
    return 0;
}

// kind 's': LidarMapPlaneNormFactor(point, coeff[0..3], cov), 'c': LidarMapEdgeFactor(point, coeff[0..5], cov); J: 7 doubles or NULL
int ref_map_factor_evaluate(char kind, const double point[3], const double *coeff, const double cov9[9], const double pose7[7], double *residual, double *J7)
{
    Eigen::Vector3d p(point[0], point[1], point[2]);
    Eigen::Matrix3d cov;
    for (int i = 0; i < 9; ++i) cov.d[i] = cov9[i];
    const double *params[1] = {pose7};
    double *jac[1] = {J7};
    if (kind == 's') {
        LidarMapPlaneNormFactor f(p, Eigen::Vector4d(coeff[0], coeff[1], coeff[2], coeff[3]), cov);
        f.Evaluate(params, residual, J7 ? jac : nullptr);
    } else {
        Eigen::VectorXd c(6);
        for (int i = 0; i < 6; ++i) c(i) = coeff[i];
        LidarMapEdgeFactor f(p, c, cov);
        f.Evaluate(params, residual, J7 ? jac : nullptr);
    }
    return 0;
}

// ImageSegmenter::segmentCloud. prm as orc_segment_cloud. cloud_out / outlier: n (+1) x 4 floats.
int ref_segment_cloud(const float *xyzi, int n, const double *prm, float *cloud_out, int *n_out, float *outlier, int *n_outlier, int *scan_start, int *scan_end)
{
    const int vs = int(prm[0]), hs = int(prm[1]);
    ImageSegmenter seg;
    seg.setParameter(vs, hs, int(prm[2]), int(prm[3]), int(prm[4]));
    SEGMENT_THETA = float(prm[5]); ROI_RANGE = prm[6];
    PointICloud in, out, outl;
    in.points.resize(size_t(n));
    for (int i = 0; i < n; ++i) { in.points[i].x = xyzi[4 * i]; in.points[i].y = xyzi[4 * i + 1]; in.points[i].z = xyzi[4 * i + 2]; in.points[i].intensity = xyzi[4 * i + 3]; }
    ScanInfo si;
    si.segment_flag_ = prm[7] != 0.0;
    si.scan_start_ind_.resize(vs); si.scan_end_ind_.resize(vs);
    seg.segmentCloud(in, out, outl, si);
    *n_out = int(out.size()); *n_outlier = int(outl.size());
    for (size_t i = 0; i < out.size(); ++i) { cloud_out[4 * i] = out.points[i].x; cloud_out[4 * i + 1] = out.points[i].y; cloud_out[4 * i + 2] = out.points[i].z; cloud_out[4 * i + 3] = out.points[i].intensity; }
    for (size_t i = 0; i < outl.size(); ++i) { outlier[4 * i] = outl.points[i].x; outlier[4 * i + 1] = outl.points[i].y; outlier[4 * i + 2] = outl.points[i].z; outlier[4 * i + 3] = outl.points[i].intensity; }
    for (int i = 0; i < vs; ++i) { scan_start[i] = si.scan_start_ind_[i]; scan_end[i] = si.scan_end_ind_[i]; }
    return 0;
}

// match{Surf,Corner}PointFromMap for every feature of a cloud against a map cloud (both n x 4 float rows [x y z intensity]):
// valid[i], coeffs[i * 6 ..] (plane: n, d; line: the two points)
int ref_match_points(char kind, const float *map4, int n_map, const float *feat4, int n_feat, const double pose7[7], int n_neigh, int check_fov,
                     float min_match_sq_dis, float min_plane_dis, unsigned char *valid, double *coeffs)
{
    MIN_MATCH_SQ_DIS = min_match_sq_dis; MIN_PLANE_DIS = min_plane_dis;
    PointICloud map;
    map.points.resize(size_t(n_map));
    for (int i = 0; i < n_map; ++i) { map.points[i].x = map4[4 * i]; map.points[i].y = map4[4 * i + 1]; map.points[i].z = map4[4 * i + 2]; map.points[i].intensity = map4[4 * i + 3]; }
    pcl::KdTreeFLANN<PointI>::Ptr kd(new pcl::KdTreeFLANN<PointI>());
    kd->setInputCloud(map);
    Pose pose;
    pose.t_ = Eigen::Vector3d(pose7[0], pose7[1], pose7[2]);
    pose.q_ = Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]);
    FeatureExtract f;
    for (int i = 0; i < n_feat; ++i) {
        PointI p;
        p.x = feat4[4 * i]; p.y = feat4[4 * i + 1]; p.z = feat4[4 * i + 2]; p.intensity = feat4[4 * i + 3];
        PointPlaneFeature ft;
        const bool ok = kind == 's' ? f.matchSurfPointFromMap<PointI>(kd, map, p, pose, ft, size_t(i), size_t(n_neigh), check_fov != 0)
                                    : f.matchCornerPointFromMap<PointI>(kd, map, p, pose, ft, size_t(i), size_t(n_neigh), check_fov != 0);
        valid[i] = ok ? 1 : 0;
        for (int k = 0; k < 6; ++k) coeffs[size_t(i) * 6 + k] = (ok && k < ft.coeffs_.size()) ? ft.coeffs_(k) : 0.0;
    }
    return 0;
}

// kind 's' / 'c': LidarPureOdom{PlaneNorm,Edge}Factor(point, coeff, sqrt_info) on (pivot, pose_i, ext); J21 = three 1x7 rows or NULL
int ref_pure_odom_evaluate(char kind, const double point[3], const double *coeff, double sqrt_info, const double *pivot, const double *pose_i,
                           const double *ext, double *residual, double *J21)
{
    Eigen::Vector3d p(point[0], point[1], point[2]);
    const double *params[3] = {pivot, pose_i, ext};
    double *jac[3] = {J21, J21 ? J21 + 7 : nullptr, J21 ? J21 + 14 : nullptr};
    if (kind == 's') {
        LidarPureOdomPlaneNormFactor f(p, Eigen::Vector4d(coeff[0], coeff[1], coeff[2], coeff[3]), sqrt_info);
        f.Evaluate(params, residual, J21 ? jac : nullptr);
    } else {
        Eigen::VectorXd c(6);
        for (int i = 0; i < 6; ++i) c(i) = coeff[i];
        LidarPureOdomEdgeFactor f(p, c, sqrt_info);
        f.Evaluate(params, residual, J21 ? jac : nullptr);
    }
    return 0;
}

int ref_online_calib_evaluate(char kind, const double point[3], const double *coeff, double sqrt_info, const double *ext, double *residual, double *J7)
{
    Eigen::Vector3d p(point[0], point[1], point[2]);
    const double *params[1] = {ext};
    double *jac[1] = {J7};
    if (kind == 's') {
        LidarOnlineCalibPlaneNormFactor f(p, Eigen::Vector4d(coeff[0], coeff[1], coeff[2], coeff[3]), sqrt_info);
        f.Evaluate(params, residual, J7 ? jac : nullptr);
    } else {
        Eigen::VectorXd c(6);
        for (int i = 0; i < 6; ++i) c(i) = coeff[i];
        LidarOnlineCalibEdgeFactor f(p, c, sqrt_info);
        f.Evaluate(params, residual, J7 ? jac : nullptr);
    }
    return 0;
}

// evalPointUncertainty (associate_uct.hpp:164-215): T4 = the pose's 4x4 homogeneous matrix (row-major), cov_pose 6x6, cov_meas 3x3; both
// overloads (explicit covariance / pose.cov_) are run and must agree; cov_out 3x3 row-major
int ref_eval_point_uncertainty(const float p[3], const double T4[16], const double cov_pose[36], const double cov_meas[9], double cov_out[9], double cov_out2[9])
{
    for (int i = 0; i < 9; ++i) COV_MEASUREMENT.d[i] = cov_meas[i];
    Pose pose;
    for (int i = 0; i < 16; ++i) pose.T_.d[i] = T4[i];
    Eigen::Matrix<double, 6, 6> cp;
    for (int i = 0; i < 36; ++i) { cp.d[i] = cov_pose[i]; pose.cov_.d[i] = cov_pose[i]; }
    PointI pi;
    pi.x = p[0]; pi.y = p[1]; pi.z = p[2];
    Eigen::Matrix3d c1, c2;
    evalPointUncertainty(pi, c1, pose, cp);
    evalPointUncertainty(pi, c2, pose);
    for (int i = 0; i < 9; ++i) { cov_out[i] = c1.d[i]; cov_out2[i] = c2.d[i]; }
    return 0;
}

// compoundPoseWithCov (associate_uct.hpp:31-86, method 2 = Barfoot's fourth-order terms): poses as [t, q(xyzw)], covariances 6x6 row-major
int ref_compound_pose_with_cov(const double p1[7], const double c1[36], const double p2[7], const double c2[36], double pose_cp[7], double cov_cp[36])
{
    auto mk = [](const double p[7]) {
        Pose P;
        P.t_ = Eigen::Vector3d(p[0], p[1], p[2]);
        P.q_ = Eigen::Quaterniond(p[6], p[3], p[4], p[5]);
        const Eigen::Matrix3d R = P.q_.toRotationMatrix();            // Pose::update (pose.cpp): T_ = [R t; 0 1]
        for (int i = 0; i < 16; ++i) P.T_.d[i] = 0.0;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P.T_(i, j) = R(i, j); P.T_(i, 3) = p[i]; }
        P.T_(3, 3) = 1.0;
        return P;
    };
    const Pose a = mk(p1), b = mk(p2);
    Eigen::Matrix<double, 6, 6> ca, cb, cc;
    for (int i = 0; i < 36; ++i) { ca.d[i] = c1[i]; cb.d[i] = c2[i]; }
    Pose out;
    compoundPoseWithCov(a, ca, b, cb, out, cc);
    pose_cp[0] = out.t_(0); pose_cp[1] = out.t_(1); pose_cp[2] = out.t_(2);
    pose_cp[3] = out.q_.x(); pose_cp[4] = out.q_.y(); pose_cp[5] = out.q_.z(); pose_cp[6] = out.q_.w();
    for (int i = 0; i < 36; ++i) cov_cp[i] = cc.d[i];
    return 0;
}

// ActiveFeatureSelection::goodFeatureMatching / evalFullHessian on (map, features as 11-float PointXYZIWithCov records, pose); the engine is
// re-seeded (the reference seeds it from std::random_device: its selections are not reproducible run to run, so "the same selection"
// can only mean "given the same engine state"). H36 in: the caller's seed matrix (1e-6 I in the mapper), out: sub_mat_H / mat_H.
static void afs_setup(const float *map11, int n_map, const float *feat11, int n_feat, const double pose7[7], PointICovCloud &map, PointICovCloud &feat,
                      pcl::KdTreeFLANN<PointIWithCov>::Ptr &kd, Pose &pose)
{
    auto fill = [](PointICovCloud &c, const float *a, int n) {
        c.points.resize(size_t(n));
        for (int i = 0; i < n; ++i) {
            PointIWithCov &q = c.points[size_t(i)];
            q.x = a[11 * i]; q.y = a[11 * i + 1]; q.z = a[11 * i + 2]; q.intensity = a[11 * i + 3];
            for (int k = 0; k < 6; ++k) q.cov_vec[k] = a[11 * i + 4 + k];
            q.cov_trace = a[11 * i + 10];
        }
    };
    fill(map, map11, n_map); fill(feat, feat11, n_feat);
    kd.reset(new pcl::KdTreeFLANN<PointIWithCov>());
    kd->setInputCloud(map);
    pose.t_ = Eigen::Vector3d(pose7[0], pose7[1], pose7[2]);
    pose.q_ = Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]);
}

int ref_good_feature_matching(char kind, const float *map11, int n_map, const float *feat11, int n_feat, const double pose7[7], const char *gf_method,
                              double gf_ratio, unsigned seed, float min_match_sq_dis, float min_plane_dis, double H36[36], int *sel, int *n_sel)
{
    MIN_MATCH_SQ_DIS = min_match_sq_dis; MIN_PLANE_DIS = min_plane_dis;
    PointICovCloud map, feat;
    pcl::KdTreeFLANN<PointIWithCov>::Ptr kd;
    Pose pose;
    afs_setup(map11, n_map, feat11, n_feat, pose7, map, feat, kd, pose);
    ActiveFeatureSelection afs;
    ceres::HuberLoss huber(1.0);
    afs.loss_function_ = &huber;
    afs.rgi_.m_random_engine.seed(seed);
    Eigen::Matrix<double, 6, 6> H;
    for (int i = 0; i < 36; ++i) H.d[i] = H36[i];
    std::vector<PointPlaneFeature> all_features;
    std::vector<size_t> sel_idx;
    afs.goodFeatureMatching(kd, map, feat, pose, all_features, sel_idx, kind, std::string(gf_method), gf_ratio, H);
    for (int i = 0; i < 36; ++i) H36[i] = H.d[i];
    *n_sel = int(sel_idx.size());
    for (size_t i = 0; i < sel_idx.size(); ++i) sel[i] = int(sel_idx[i]);
    return 0;
}

int ref_eval_full_hessian(char kind, const float *map11, int n_map, const float *feat11, int n_feat, const double pose7[7], float min_match_sq_dis,
                          float min_plane_dis, double H36[36], int *feat_num, double *logdet)
{
    MIN_MATCH_SQ_DIS = min_match_sq_dis; MIN_PLANE_DIS = min_plane_dis;
    PointICovCloud map, feat;
    pcl::KdTreeFLANN<PointIWithCov>::Ptr kd;
    Pose pose;
    afs_setup(map11, n_map, feat11, n_feat, pose7, map, feat, kd, pose);
    ActiveFeatureSelection afs;
    ceres::HuberLoss huber(1.0);
    afs.loss_function_ = &huber;
    Eigen::Matrix<double, 6, 6> H;
    for (int i = 0; i < 36; ++i) H.d[i] = H36[i];
    afs.evalFullHessian(kd, map, feat, pose, kind, H, *feat_num);
    for (int i = 0; i < 36; ++i) H36[i] = H.d[i];
    if (logdet) *logdet = common::logDet(H, true);             // gf_deg_factor (lidar_mapper_keyframe.cpp:463)
    return 0;
}

// LidarTracker's matching (lidar_tracker.cpp:58-59): prev4 / cur4 rows [x y z ring id]; valid[i], coeffs[i] (corner: the two line points;
// surf: plane n, d) per CURRENT feature i
int ref_track_match(char kind, const float *prev4, int n_prev, const float *cur4, int n_cur, const double pose7[7], float dist_sq_thr, float nearby_scan,
                    unsigned char *valid, double *coeffs)
{
    DISTANCE_SQ_THRESHOLD = dist_sq_thr; NEARBY_SCAN = nearby_scan;
    auto fill = [](PointICloud &c, const float *a, int n) {
        c.points.resize(size_t(n));
        for (int i = 0; i < n; ++i) { c.points[size_t(i)].x = a[4 * i]; c.points[size_t(i)].y = a[4 * i + 1]; c.points[size_t(i)].z = a[4 * i + 2]; c.points[size_t(i)].intensity = a[4 * i + 3]; }
    };
    PointICloud prev, cur;
    fill(prev, prev4, n_prev); fill(cur, cur4, n_cur);
    pcl::KdTreeFLANN<PointI>::Ptr kd(new pcl::KdTreeFLANN<PointI>());
    kd->setInputCloud(prev);
    Pose pose;
    pose.t_ = Eigen::Vector3d(pose7[0], pose7[1], pose7[2]);
    pose.q_ = Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]);
    FeatureExtract f;
    std::vector<PointPlaneFeature> feats;
    if (kind == 'c') f.matchCornerFromScan<PointI>(kd, prev, cur, pose, feats);
    else f.matchSurfFromScan<PointI>(kd, prev, cur, pose, feats);
    for (int i = 0; i < n_cur; ++i) { valid[i] = 0; for (int k = 0; k < 6; ++k) coeffs[size_t(i) * 6 + k] = 0.0; }
    for (const PointPlaneFeature &ft : feats) {
        valid[ft.idx_] = 1;
        for (int k = 0; k < 6 && k < ft.coeffs_.size(); ++k) coeffs[ft.idx_ * 6 + k] = ft.coeffs_(k);
    }
    return 0;
}

// kind 'S': LidarScanPlaneNormFactor (1 residual), 'E': LidarScanEdgeFactorVector (3 residuals); J rows x 7 or NULL
int ref_scan_factor_eval(char kind, const double point[3], const double *coeff, double s, const double pose7[7], double *r, double *J)
{
    Eigen::Vector3d p(point[0], point[1], point[2]);
    const double *params[1] = {pose7};
    double *jac[1] = {J};
    if (kind == 'S') {
        LidarScanPlaneNormFactor f(p, Eigen::Vector4d(coeff[0], coeff[1], coeff[2], coeff[3]), s);
        f.Evaluate(params, r, J ? jac : nullptr);
    } else {
        Eigen::VectorXd c(6);
        for (int i = 0; i < 6; ++i) c(i) = coeff[i];
        LidarScanEdgeFactorVector f(p, c, s);
        f.Evaluate(params, r, J ? jac : nullptr);
    }
    return 0;
}

// TransformToEnd (utility.h:79-100) over rows [x y z intensity]
int ref_transform_to_end(const float *pts4, int n, const double pose7[7], int distortion, float scan_period, float *out4)
{
    Pose pose;
    pose.t_ = Eigen::Vector3d(pose7[0], pose7[1], pose7[2]);
    pose.q_ = Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]);
    for (int i = 0; i < n; ++i) {
        PointI a, b;
        a.x = pts4[4 * i]; a.y = pts4[4 * i + 1]; a.z = pts4[4 * i + 2]; a.intensity = pts4[4 * i + 3];
        TransformToEnd(a, b, pose, distortion != 0, scan_period);
        out4[4 * i] = b.x; out4[4 * i + 1] = b.y; out4[4 * i + 2] = b.z; out4[4 * i + 3] = b.intensity;
    }
    return 0;
}

// cloudUCTAssociateToMap (lidar_mapper_keyframe.cpp:1116-1158): 11-float PointXYZIWithCov records in and out; poses [t, q(xyzw)], covariances 6x6
int ref_cloud_uct_associate_to_map(const float *in11, int n, const double pose_global[7], const double cov_global[36], const double *ext_poses, const double *ext_covs,
                                   int n_lidar, const double cov_meas[9], int with_ua, double trace_threshold, float *out11, int *n_out)
{
    NUM_OF_LASER = size_t(n_lidar); with_ua_flag = with_ua != 0; TRACE_THRESHOLD_MAPPING = trace_threshold;
    for (int i = 0; i < 9; ++i) COV_MEASUREMENT.d[i] = cov_meas[i];
    auto mk = [](const double *p, const double *c) {
        Pose P(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2]));
        for (int i = 0; i < 36; ++i) P.cov_.d[i] = c[i];
        return P;
    };
    const Pose pg = mk(pose_global, cov_global);
    std::vector<Pose> pe;
    for (int k = 0; k < n_lidar; ++k) pe.push_back(mk(ext_poses + 7 * k, ext_covs + 36 * k));
    PointICovCloud local, global;
    local.points.resize(size_t(n));
    for (int i = 0; i < n; ++i) {
        PointIWithCov &q = local.points[size_t(i)];
        q.x = in11[11 * i]; q.y = in11[11 * i + 1]; q.z = in11[11 * i + 2]; q.intensity = in11[11 * i + 3];
        for (int k = 0; k < 6; ++k) q.cov_vec[k] = in11[11 * i + 4 + k];
        q.cov_trace = in11[11 * i + 10];
    }
    cloudUCTAssociateToMap(local, global, pg, pe);
    *n_out = int(global.size());
    for (size_t i = 0; i < global.size(); ++i) {
        const PointIWithCov &q = global.points[i];
        float *o = out11 + 11 * i;
        o[0] = q.x; o[1] = q.y; o[2] = q.z; o[3] = q.intensity;
        for (int k = 0; k < 6; ++k) o[4 + k] = q.cov_vec[k];
        o[10] = q.cov_trace;
    }
    return 0;
}

// evalDegenracy (lidar_mapper_keyframe.cpp:1172-1204) on a fresh PoseLocalParameterization: is_degenerate_, V_update_ (row-major), eigenvalues
int ref_eval_degeneracy(const double H36[36], double eig_thre, int *is_deg, double V36[36], double eig[6])
{
    MAP_EIG_THRE = eig_thre;
    Eigen::Matrix<double, 6, 6> H;
    for (int i = 0; i < 36; ++i) H.d[i] = H36[i];
    PoseLocalParameterization plp;
    plp.setParameter();
    d_factor_list.clear(); d_eigvec_list.clear();
    evalDegenracy(H, &plp);
    *is_deg = plp.is_degenerate_ ? 1 : 0;
    for (int i = 0; i < 36; ++i) V36[i] = plp.V_update_.d[i];
    for (int i = 0; i < 6; ++i) eig[i] = d_factor_list.back().d[i];
    return 0;
}

// evalHessian: the mapper's J^T J from the CRS Jacobian problem.Evaluate returns (6 local columns)
int ref_eval_hessian(const int *crs_rows, const int *crs_cols, const double *crs_values, int n_rows, int n_cols, double H36[36])
{
    ceres::CRSMatrix jaco;
    jaco.num_rows = n_rows; jaco.num_cols = n_cols;
    jaco.rows.assign(crs_rows, crs_rows + n_rows + 1);
    jaco.cols.assign(crs_cols, crs_cols + crs_rows[n_rows]);
    jaco.values.assign(crs_values, crs_values + crs_rows[n_rows]);
    Eigen::Matrix<double, 6, 6> H;
    evalHessian(jaco, H);
    for (int i = 0; i < 36; ++i) H36[i] = H.d[i];
    return 0;
}

// Estimator::evalDegenracy: J as CRS (rows n_rows + 1, cols / values nnz), D = 6 * n_blocks columns; eig_thre (n_blocks) in and out; per block out:
// is_degenerate, V_update (36, row-major); d_factor_calib (NUM_OF_LASER)
int ref_estimator_eval_degeneracy(const int *crs_rows, const int *crs_cols, const double *crs_values, int n_rows, int n_cols, int opt_window_size, int num_of_laser,
                                  int estimate_extrinsic, int frame_cnt, int n_cumu_feature, double lambda_thre_calib, double *eig_thre, int *is_degenerate,
                                  double *V_update, double *d_factor_calib)
{
    OPT_WINDOW_SIZE = opt_window_size; NUM_OF_LASER = size_t(num_of_laser); ESTIMATE_EXTRINSIC = estimate_extrinsic; N_CUMU_FEATURE = n_cumu_feature;
    LAMBDA_THRE_CALIB = lambda_thre_calib;
    const int n_blocks = n_cols / 6;
    ceres::CRSMatrix jaco;
    jaco.num_rows = n_rows; jaco.num_cols = n_cols;
    jaco.rows.assign(crs_rows, crs_rows + n_rows + 1);
    jaco.cols.assign(crs_cols, crs_cols + crs_rows[n_rows]);
    jaco.values.assign(crs_values, crs_values + crs_rows[n_rows]);
    Estimator est;
    est.frame_cnt_ = frame_cnt;
    est.eig_thre_ = Eigen::VectorXd(n_blocks);
    for (int i = 0; i < n_blocks; ++i) est.eig_thre_(i) = eig_thre[i];
    est.qbl_.assign(size_t(num_of_laser), Eigen::Quaterniond());
    est.tbl_.assign(size_t(num_of_laser), Eigen::Vector3d());
    std::vector<PoseLocalParameterization> store(static_cast<size_t>(n_blocks));
    std::vector<PoseLocalParameterization *> ids;
    for (auto &p : store) { p.setParameter(); ids.push_back(&p); }     // as optimizeMap creates them (estimator.cpp:628-650)
    std::ostringstream sink;
    std::streambuf *keep = std::cout.rdbuf(sink.rdbuf());
    est.evalDegenracy(ids, jaco);
    std::cout.rdbuf(keep);
    for (int i = 0; i < n_blocks; ++i) {
        eig_thre[i] = est.eig_thre_(i);
        is_degenerate[i] = store[size_t(i)].is_degenerate_ ? 1 : 0;
        for (int k = 0; k < 36; ++k) V_update[36 * i + k] = store[size_t(i)].V_update_.d[k];
    }
    for (int i = 0; i < num_of_laser; ++i) d_factor_calib[i] = i < int(est.d_factor_calib_.size()) ? est.d_factor_calib_[size_t(i)] : 0.0;
    return 0;
}

// downsampleCurrentScan (lidar_mapper_keyframe.cpp:356-421): surf / corner fused clouds in (rows x y z lidar-id), thinned clouds with covariance out (11 floats)
int ref_downsample_current_scan(const float *surf4, int n_surf, const float *corner4, int n_corner, float leaf_surf, float leaf_corner, const double *ext_poses,
                                const double *ext_covs, int n_lidar, const double cov_meas[9], int with_ua, double trace_threshold, float *surf11, int *n_surf_out,
                                float *corner11, int *n_corner_out)
{
    with_ua_flag = with_ua != 0; TRACE_THRESHOLD_MAPPING = trace_threshold;
    for (int i = 0; i < 9; ++i) COV_MEASUREMENT.d[i] = cov_meas[i];
    pose_ext.clear();
    for (int k = 0; k < n_lidar; ++k) {
        const double *p = ext_poses + 7 * k;
        Pose P(Eigen::Quaterniond(p[6], p[3], p[4], p[5]), Eigen::Vector3d(p[0], p[1], p[2]));
        for (int i = 0; i < 36; ++i) P.cov_.d[i] = ext_covs[36 * k + i];
        pose_ext.push_back(P);
    }
    auto fill = [](PointICloud &c, const float *a, int n) {
        c.points.resize(size_t(n));
        for (int i = 0; i < n; ++i) { c.points[size_t(i)].x = a[4 * i]; c.points[size_t(i)].y = a[4 * i + 1]; c.points[size_t(i)].z = a[4 * i + 2]; c.points[size_t(i)].intensity = a[4 * i + 3]; }
    };
    fill(*laser_cloud_surf_last, surf4, n_surf); fill(*laser_cloud_corner_last, corner4, n_corner); laser_cloud_outlier->clear();
    down_size_filter_surf.setLeafSize(leaf_surf, leaf_surf, leaf_surf); down_size_filter_corner.setLeafSize(leaf_corner, leaf_corner, leaf_corner);
    down_size_filter_outlier.setLeafSize(leaf_corner, leaf_corner, leaf_corner);
    downsampleCurrentScan();
    auto dump = [](const PointICovCloud &c, float *o, int *n) {
        *n = int(c.size());
        for (size_t i = 0; i < c.size(); ++i) {
            const PointIWithCov &q = c.points[i];
            o[11 * i] = q.x; o[11 * i + 1] = q.y; o[11 * i + 2] = q.z; o[11 * i + 3] = q.intensity;
            for (int k = 0; k < 6; ++k) o[11 * i + 4 + k] = q.cov_vec[k];
            o[11 * i + 10] = q.cov_trace;
        }
    };
    dump(*laser_cloud_surf_cov, surf11, n_surf_out); dump(*laser_cloud_corner_cov, corner11, n_corner_out);
    return 0;
}

// VoxelGridCovarianceMLOAM<PointT>::filter from the reference's own lines (voxel_grid_covariance_mloam_impl.hpp:68-457). n_fields 4: PointXYZI records
// (plain branch: xyz mean, the LAST member's intensity), 11: PointXYZIWithCov records (trace gate, weights, the heaviest member's intensity);
// trace_threshold < 0 keeps the class default (2.0). out: the same record layout, voxel-index order.
int ref_voxel_filter(const float *in, int n, int n_fields, float leaf, float trace_threshold, float *out, int *n_out)
{
    if (n_fields == 4) {
        PointICloud::Ptr c(new PointICloud());
        c->points.resize(size_t(n));
        for (int i = 0; i < n; ++i) { c->points[size_t(i)].x = in[4 * i]; c->points[size_t(i)].y = in[4 * i + 1]; c->points[size_t(i)].z = in[4 * i + 2]; c->points[size_t(i)].intensity = in[4 * i + 3]; }
        pcl::VoxelGridCovarianceMLOAM<PointI> f;
        f.setInputCloud(c);
        f.setLeafSize(leaf, leaf, leaf);
        if (trace_threshold >= 0.f) f.setTraceThreshold(trace_threshold);
        PointICloud o;
        f.filter(o);
        *n_out = int(o.size());
        for (size_t i = 0; i < o.size(); ++i) { out[4 * i] = o.points[i].x; out[4 * i + 1] = o.points[i].y; out[4 * i + 2] = o.points[i].z; out[4 * i + 3] = o.points[i].intensity; }
        return 0;
    }
    if (n_fields != 11) return 1;
    PointICovCloud::Ptr c(new PointICovCloud());
    c->points.resize(size_t(n));
    for (int i = 0; i < n; ++i) {
        PointIWithCov &q = c->points[size_t(i)];
        q.x = in[11 * i]; q.y = in[11 * i + 1]; q.z = in[11 * i + 2]; q.intensity = in[11 * i + 3];
        for (int k = 0; k < 6; ++k) q.cov_vec[k] = in[11 * i + 4 + k];
        q.cov_trace = in[11 * i + 10];
    }
    pcl::VoxelGridCovarianceMLOAM<PointIWithCov> f;
    f.setInputCloud(c);
    f.setLeafSize(leaf, leaf, leaf);
    if (trace_threshold >= 0.f) f.setTraceThreshold(trace_threshold);
    PointICovCloud o;
    f.filter(o);
    *n_out = int(o.size());
    for (size_t i = 0; i < o.size(); ++i) {
        const PointIWithCov &q = o.points[i];
        float *w = out + 11 * i;
        w[0] = q.x; w[1] = q.y; w[2] = q.z; w[3] = q.intensity;
        for (int k = 0; k < 6; ++k) w[4 + k] = q.cov_vec[k];
        w[10] = q.cov_trace;
    }
    return 0;
}

// scan2MapOptimization from the reference's own lines: maps and features as 11-float PointXYZIWithCov records; gf_method / gf_ratio / seed as the
// mapper's flags; frame_cnt_in = 0 also runs the every-tenth-frame evalFullHessian + ratio policy (cpp:456-494). Out: the pose, and per ceres::Solve
// call [residual blocks, iterations, successful steps, termination, initial cost, final cost] (6 doubles each, at most max_solves of them).
static void fill_cov_cloud(PointICovCloud &c, const float *a, int n)
{
    c.points.resize(size_t(n));
    for (int i = 0; i < n; ++i) {
        PointIWithCov &q = c.points[size_t(i)];
        q.x = a[11 * i]; q.y = a[11 * i + 1]; q.z = a[11 * i + 2]; q.intensity = a[11 * i + 3];
        for (int k = 0; k < 6; ++k) q.cov_vec[k] = a[11 * i + 4 + k];
        q.cov_trace = a[11 * i + 10];
    }
}
static int dump_solve_log(double *solves, int max_solves)
{
    int n = 0;
    for (const ceres::Solver::Summary &s : ceres::g_solve_log) {
        if (n >= max_solves) break;
        double *o = solves + 6 * n++;
        o[0] = s.num_residual_blocks; o[1] = s.s.num_iterations; o[2] = s.s.num_successful_steps; o[3] = s.s.termination; o[4] = s.s.initial_cost; o[5] = s.s.final_cost;
    }
    return n;
}
int ref_scan2map_optimization(const float *surf_map11, int n_surf_map, const float *corner_map11, int n_corner_map, const float *surf11, int n_surf,
                              const float *corner11, int n_corner, const double pose7[7], int with_ua, const double cov_meas[9], double map_eig_thre,
                              const char *gf_method, double gf_ratio, unsigned seed, int frame_cnt_in, float min_match_sq_dis, float min_plane_dis,
                              double pose_out[7], double *solves, int max_solves, int *n_solves, double cov_out[36])
{
    MIN_MATCH_SQ_DIS = min_match_sq_dis; MIN_PLANE_DIS = min_plane_dis; MAP_EIG_THRE = map_eig_thre;
    with_ua_flag = with_ua != 0;
    for (int i = 0; i < 9; ++i) COV_MEASUREMENT.d[i] = cov_meas[i];
    fill_cov_cloud(*laser_cloud_surf_from_map_cov_ds, surf_map11, n_surf_map); fill_cov_cloud(*laser_cloud_corner_from_map_cov_ds, corner_map11, n_corner_map);
    fill_cov_cloud(*laser_cloud_surf_cov, surf11, n_surf); fill_cov_cloud(*laser_cloud_corner_cov, corner11, n_corner);
    FLAGS_gf_method = gf_method; FLAGS_gf_ratio_ini = gf_ratio; gf_ratio_cur = std::string(gf_method) == "wo_gf" ? 1.0 : gf_ratio;
    frame_cnt = frame_cnt_in;
    afs.rgi_.m_random_engine.seed(seed);
    pose_wmap_curr = Pose(Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]), Eigen::Vector3d(pose7[0], pose7[1], pose7[2]));
    pose_keyframes_6d.assign(20, std::make_pair(0.0, Pose()));   // more than 10 keyframes: cov_mapping = mat_H^-1 (cpp:607-610)
    d_factor_list.clear(); d_eigvec_list.clear(); gf_deg_factor_list.clear(); gf_logdet_H_list.clear(); mapping_sp_list.clear();
    ceres::g_solve_log.clear();
    std::ostringstream sink;
    std::streambuf *keep = std::cout.rdbuf(sink.rdbuf());
    scan2MapOptimization();
    std::cout.rdbuf(keep);
    pose_out[0] = pose_wmap_curr.t_(0); pose_out[1] = pose_wmap_curr.t_(1); pose_out[2] = pose_wmap_curr.t_(2);
    pose_out[3] = pose_wmap_curr.q_.x(); pose_out[4] = pose_wmap_curr.q_.y(); pose_out[5] = pose_wmap_curr.q_.z(); pose_out[6] = pose_wmap_curr.q_.w();
    *n_solves = dump_solve_log(solves, max_solves);
    if (cov_out) for (int i = 0; i < 36; ++i) cov_out[i] = pose_wmap_curr.cov_.d[i];
    return 0;
}

// the pose the mapper starts the next frame from: transformUpdate() with frame k's result and odometry, transformAssociateToMap() with frame k+1's odometry
int ref_pose_chain(const double wmap_curr_prev[7], const double wodom_prev[7], const double wodom_cur[7], double out[7])
{
    // members assigned directly, as double2Vector (cpp:247-252) and the odometry handler (cpp:1019-1026) do: no normalisation on the way in
    auto set = [](Pose &P, const double *p) { P.q_ = Eigen::Quaterniond(p[6], p[3], p[4], p[5]); P.t_ = Eigen::Vector3d(p[0], p[1], p[2]); };
    set(pose_wmap_curr, wmap_curr_prev);
    set(pose_wodom_curr, wodom_prev);
    transformUpdate();
    set(pose_wodom_curr, wodom_cur);
    transformAssociateToMap();
    out[0] = pose_wmap_curr.t_(0); out[1] = pose_wmap_curr.t_(1); out[2] = pose_wmap_curr.t_(2);
    out[3] = pose_wmap_curr.q_.x(); out[4] = pose_wmap_curr.q_.y(); out[5] = pose_wmap_curr.q_.z(); out[6] = pose_wmap_curr.q_.w();
    return 0;
}

// ---------------------------------------------------------------- Estimator::optimizeMap up to its marginalisation section (estimator.cpp:593-866) from the reference's own lines
// Problem assembly of the odometry window -- which factor ties which blocks, which blocks are constant, the calibration branch's accumulated features and its
// every-N_CUMU_FEATURE-frames gate, Huber(1.0) --, evalResidual -> Problem::Evaluate -> evalDegenracy, ceres::Solve, double2Vector. MARGINALIZATION_FACTOR = 0 and
// PRIOR_FACTOR = 0 (both out of scope): the two classes below exist so that the reference's text compiles; they are never constructed.
int WINDOW_SIZE = 4, NUM_ITERATIONS = 10, IDX_REF = 0, MARGINALIZATION_FACTOR = 0, PRIOR_FACTOR = 0;   // parameters.cpp
double SOLVER_TIME = 1.0, PRIOR_FACTOR_POS = 0.0, PRIOR_FACTOR_ROT = 0.0;
struct MarginalizationFactor : ceres::CostFunction {
    explicit MarginalizationFactor(MarginalizationInfo *) {}
    bool Evaluate(double const *const *, double *, double **) const override { return false; }
};
struct PriorFactor : ceres::CostFunction {
    PriorFactor(const Eigen::Vector3d &, const Eigen::Quaterniond &, double, double) {}
    bool Evaluate(double const *const *, double *, double **) const override { return false; }
};
#define printf(...) ((void)0)
#include "../_ref/gen/estimator_optimize_map_head.inc"          // void Estimator::optimizeMap() { ... double2Vector();        estimator.cpp:593-866
}                                                               // the marginalisation section (estimator.cpp:868-1000) is not compiled: out of scope
#undef printf
#include "../_ref/gen/estimator_vector_double.inc"              // vector2Double, double2Vector                                 estimator.cpp:1538-1576
#include "../_ref/gen/estimator_eval_residual.inc"              // evalResidual                                                 estimator.cpp:1578-1595

// Estimator::optimizeMap on a window given as arrays. poses: (OPT_WINDOW_SIZE + 1) x 7 (pivot first), exts: n_laser x 7. Features: rows
// [laser, frame (1..OPT_WINDOW_SIZE, window index i - pivot_idx), kind (0 surf / 1 corner), px, py, pz, c0..c5]; in the pure-odometry branch every row is selected
// (sel_*_feature_idx_ = all), in the calibration branch rows of the reference LiDAR become window factors and rows with frame = 0 of the other LiDARs are the
// pivot frame's features accumulated into cumu_*_map_features_. Out: poses / exts after ceres::Solve + double2Vector, the Jacobian handed to evalDegenracy as
// J^T J (D x D, D = 6 (OPT_WINDOW_SIZE + 1 + n_laser)), the evaluated cost, the number of residual blocks, is_degenerate per block.
int ref_optimize_map(int opt_window_size, int n_laser, int estimate_extrinsic, int n_cumu_feature, int frame_cnt, int num_iterations, const double *poses,
                     const double *exts, const double *feat_rows, int n_feat, const double *eig_thre, double lambda_thre_calib, double *poses_out,
                     double *exts_out, double *jtj_out, double *cost_out, int *n_blocks_out, double *solve_out)
{
    OPT_WINDOW_SIZE = opt_window_size; WINDOW_SIZE = opt_window_size; NUM_OF_LASER = size_t(n_laser); ESTIMATE_EXTRINSIC = estimate_extrinsic;
    N_CUMU_FEATURE = n_cumu_feature; NUM_ITERATIONS = num_iterations; IDX_REF = 0; LAMBDA_THRE_CALIB = lambda_thre_calib;
    POINT_PLANE_FACTOR = 1; POINT_EDGE_FACTOR = 1; MARGINALIZATION_FACTOR = 0; PRIOR_FACTOR = 0;
    Estimator est;
    est.frame_cnt_ = frame_cnt;
    const int nb = opt_window_size + 1;
    est.Qs_.resize(size_t(nb)); est.Ts_.resize(size_t(nb));
    for (int i = 0; i < nb; ++i) { const double *p = poses + i * 7; est.Ts_[size_t(i)] = Eigen::Vector3d(p[0], p[1], p[2]); est.Qs_[size_t(i)] = Eigen::Quaterniond(p[6], p[3], p[4], p[5]); }
    est.qbl_.resize(size_t(n_laser)); est.tbl_.resize(size_t(n_laser));
    for (int n = 0; n < n_laser; ++n) { const double *p = exts + n * 7; est.tbl_[size_t(n)] = Eigen::Vector3d(p[0], p[1], p[2]); est.qbl_[size_t(n)] = Eigen::Quaterniond(p[6], p[3], p[4], p[5]); }
    std::vector<std::vector<double>> pp(size_t(nb), std::vector<double>(7)), pe(size_t(n_laser), std::vector<double>(7));
    std::vector<double *> ppp, ppe;
    for (auto &v : pp) ppp.push_back(v.data());
    for (auto &v : pe) ppe.push_back(v.data());
    est.para_pose_ = ppp.data(); est.para_ex_pose_ = ppe.data();
    est.surf_map_features_.assign(size_t(n_laser), std::vector<std::vector<PointPlaneFeature>>(size_t(nb)));
    est.corner_map_features_ = est.surf_map_features_;
    est.cumu_surf_map_features_.assign(size_t(n_laser), {}); est.cumu_corner_map_features_.assign(size_t(n_laser), {});
    for (int r = 0; r < n_feat; ++r) {
        const double *f = feat_rows + size_t(r) * 12;
        const int n = int(f[0]), i = int(f[1]), kind = int(f[2]);
        PointPlaneFeature ft;
        ft.point_ = Eigen::Vector3d(f[3], f[4], f[5]);
        if (kind == 0) { ft.coeffs_ = Eigen::VectorXd(4); for (int k = 0; k < 4; ++k) ft.coeffs_(k) = f[6 + k]; ft.type_ = 's'; }
        else { ft.coeffs_ = Eigen::VectorXd(6); for (int k = 0; k < 6; ++k) ft.coeffs_(k) = f[6 + k]; ft.type_ = 'c'; }
        ft.laser_idx_ = size_t(n);
        auto &dst = (kind == 0 ? est.surf_map_features_ : est.corner_map_features_)[size_t(n)][size_t(i)];
        ft.idx_ = dst.size();
        dst.push_back(ft);
    }
    est.sel_surf_feature_idx_.assign(size_t(n_laser), std::vector<std::vector<size_t>>(size_t(nb)));
    est.sel_corner_feature_idx_ = est.sel_surf_feature_idx_;
    for (int n = 0; n < n_laser; ++n)
        for (int i = 0; i < nb; ++i) {
            for (size_t k = 0; k < est.surf_map_features_[size_t(n)][size_t(i)].size(); ++k) est.sel_surf_feature_idx_[size_t(n)][size_t(i)].push_back(k);
            for (size_t k = 0; k < est.corner_map_features_[size_t(n)][size_t(i)].size(); ++k) est.sel_corner_feature_idx_[size_t(n)][size_t(i)].push_back(k);
        }
    est.eig_thre_ = Eigen::VectorXd(nb + n_laser);
    for (int i = 0; i < nb + n_laser; ++i) est.eig_thre_(i) = eig_thre[i];
    est.d_factor_calib_.assign(size_t(n_laser), 0.0);
    est.log_lambda_.clear(); est.log_extrinsics_.clear();
    ceres::g_solve_log.clear();
    ceres::g_last_jacobian = ceres::CRSMatrix();
    std::ostringstream sink;
    std::streambuf *keep = std::cout.rdbuf(sink.rdbuf());
    est.optimizeMap();
    std::cout.rdbuf(keep);
    for (int i = 0; i < nb; ++i) {
        double *p = poses_out + i * 7;
        p[0] = est.Ts_[size_t(i)](0); p[1] = est.Ts_[size_t(i)](1); p[2] = est.Ts_[size_t(i)](2);
        p[3] = est.Qs_[size_t(i)].x(); p[4] = est.Qs_[size_t(i)].y(); p[5] = est.Qs_[size_t(i)].z(); p[6] = est.Qs_[size_t(i)].w();
    }
    for (int n = 0; n < n_laser; ++n) {
        double *p = exts_out + n * 7;
        p[0] = est.tbl_[size_t(n)](0); p[1] = est.tbl_[size_t(n)](1); p[2] = est.tbl_[size_t(n)](2);
        p[3] = est.qbl_[size_t(n)].x(); p[4] = est.qbl_[size_t(n)].y(); p[5] = est.qbl_[size_t(n)].z(); p[6] = est.qbl_[size_t(n)].w();
    }
    const size_t D = size_t(6 * (nb + n_laser));
    if (jtj_out) {                                            // J^T J of the Jacobian evalResidual handed to evalDegenracy (estimator.cpp:1592-1593)
        for (size_t k = 0; k < D * D; ++k) jtj_out[k] = 0.0;
        const ceres::CRSMatrix &J = ceres::g_last_jacobian;
        for (int r = 0; r < J.num_rows; ++r)
            for (int a = J.rows[size_t(r)]; a < J.rows[size_t(r) + 1]; ++a)
                for (int c = J.rows[size_t(r)]; c < J.rows[size_t(r) + 1]; ++c)
                    jtj_out[size_t(J.cols[size_t(a)]) * D + size_t(J.cols[size_t(c)])] += J.values[size_t(a)] * J.values[size_t(c)];
    }
    if (cost_out) *cost_out = ceres::g_last_cost;
    if (n_blocks_out) *n_blocks_out = ceres::g_solve_log.empty() ? 0 : ceres::g_solve_log.back().num_residual_blocks;
    if (solve_out && !ceres::g_solve_log.empty()) {
        const orc::SolveSummary &ss = ceres::g_solve_log.back().s;
        solve_out[0] = ss.num_iterations; solve_out[1] = ss.num_successful_steps; solve_out[2] = ss.initial_cost; solve_out[3] = ss.final_cost; solve_out[4] = ss.termination;
    }
    return 0;
}

// LidarTracker::trackCloud from the reference's own lines: the four clouds as [x y z ring id] rows
int ref_track_cloud(const float *corner_last4, int n_corner_last, const float *surf_last4, int n_surf_last, const float *corner_sharp4, int n_corner_sharp,
                    const float *surf_flat4, int n_surf_flat, const double pose7[7], float dist_sq_thr, float nearby_scan, double pose_out[7], double *solves,
                    int max_solves, int *n_solves)
{
    DISTANCE_SQ_THRESHOLD = dist_sq_thr; NEARBY_SCAN = nearby_scan;
    auto fill = [](PointICloud &c, const float *a, int n) {
        c.points.resize(size_t(n));
        for (int i = 0; i < n; ++i) { c.points[size_t(i)].x = a[4 * i]; c.points[size_t(i)].y = a[4 * i + 1]; c.points[size_t(i)].z = a[4 * i + 2]; c.points[size_t(i)].intensity = a[4 * i + 3]; }
    };
    cloudFeature prev, cur;
    fill(prev["corner_points_less_sharp"], corner_last4, n_corner_last); fill(prev["surf_points_less_flat"], surf_last4, n_surf_last);
    fill(cur["corner_points_sharp"], corner_sharp4, n_corner_sharp); fill(cur["surf_points_flat"], surf_flat4, n_surf_flat);
    const Pose ini(Eigen::Quaterniond(pose7[6], pose7[3], pose7[4], pose7[5]), Eigen::Vector3d(pose7[0], pose7[1], pose7[2]));
    ceres::g_solve_log.clear();
    LidarTracker tracker;
    const Pose out = tracker.trackCloud(prev, cur, ini);
    pose_out[0] = out.t_(0); pose_out[1] = out.t_(1); pose_out[2] = out.t_(2);
    pose_out[3] = out.q_.x(); pose_out[4] = out.q_.y(); pose_out[5] = out.q_.z(); pose_out[6] = out.q_.w();
    *n_solves = dump_solve_log(solves, max_solves);
    return 0;
}

// PoseLocalParameterization: setParameter(), optionally V_update_ <- V36 (row-major) as evalDegenracy does, then Plus
int ref_pose_plus(const double x[7], const double delta[6], const double *V36, double out[7])
{
    PoseLocalParameterization plp;
    plp.setParameter();
    if (V36) for (int i = 0; i < 36; ++i) plp.V_update_.d[i] = V36[i];
    const ceres::LocalParameterization &base = plp;
    base.Plus(x, delta, out);
    return 0;
}
}
