// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (PCL/FLANN/Ceres absent).
//
// CPU restatement of the scan-to-scan odometry (the caller side of the hot path):
//   TransformToStart                         estimator/src/utility/utility.h:55-77
//   TransformToEnd                           utility.h:79-100 (Estimator::undistortMeasurements, estimator.cpp:376-410)
//   Eigen::Quaterniond::slerp                Eigen 3.3.4 Geometry/Quaternion.h (restated)
//   FeatureExtract::matchCornerFromScan      estimator/src/featureExtract/feature_extract.hpp:132-270
//   FeatureExtract::matchSurfFromScan        feature_extract.hpp:273-376
//   LidarScanPlaneNormFactor::Evaluate       estimator/src/factor/lidar_scan_factor.hpp:24-64
//   LidarScanEdgeFactorVector::Evaluate      lidar_scan_factor.hpp:236-279
//   LidarTracker::trackCloud                 estimator/src/lidarTracker/lidar_tracker.cpp:23-129
// Clouds are PointXYZI rows [x y z intensity]; int(intensity) is the ring id (image_segmenter.hpp:128).
#pragma once
#include "mapper.hpp"
#include <cmath>
#include <vector>

namespace orc {

// Eigen 3.3.4 QuaternionBase::slerp(t, other), *this = Identity
static inline Quatd slerp_from_identity(double t, const Quatd &other)
{
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = other.w;                       // Identity.dot(other) = 0*x + 0*y + 0*z + 1*w
    const double absD = std::fabs(d);
    double scale0, scale1;
    if (absD >= one) { scale0 = 1.0 - t; scale1 = t; }
    else {
        const double theta = std::acos(absD), sinTheta = std::sin(theta);
        scale0 = std::sin((1.0 - t) * theta) / sinTheta;
        scale1 = std::sin(t * theta) / sinTheta;
    }
    if (d < 0.0) scale1 = -scale1;
    return {scale0 * 0.0 + scale1 * other.x, scale0 * 0.0 + scale1 * other.y, scale0 * 0.0 + scale1 * other.z, scale0 * 1.0 + scale1 * other.w};
}

// utility.h:55-77 (f64 math, f32 store)
static inline void transform_to_start(const float *pi /*x y z intensity*/, const Pose &pose, bool b_distortion, float scan_period, float po[3])
{
    double s = 1.0;
    if (b_distortion) s = (pi[3] - int(pi[3])) / scan_period;
    const Quatd q = slerp_from_identity(s, pose.q);
    const Vec3d t{s * pose.t.x, s * pose.t.y, s * pose.t.z};
    const Vec3d r = quat_rotate(q, {double(pi[0]), double(pi[1]), double(pi[2])});
    po[0] = float(r.x + t.x); po[1] = float(r.y + t.y); po[2] = float(r.z + t.z);
}

// utility.h:79-100 TransformToEnd: p^b = T^-1 T(s) p^c; the intermediate point goes through a float point (un_point_tmp)
static inline void transform_to_end(const float *pi /*x y z intensity*/, const Pose &pose, bool b_distortion, float scan_period, float po[3])
{
    float tmp[3];
    transform_to_start(pi, pose, b_distortion, scan_period, tmp);
    const Vec3d d{double(tmp[0]) - pose.t.x, double(tmp[1]) - pose.t.y, double(tmp[2]) - pose.t.z};
    // Eigen quaternion inverse(): conjugate / squaredNorm (4 doubles reduced pairwise: the SSE2 packet order of an x86-64 -O3 build)
    const Quatd &q = pose.q;
    const double n2 = (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
    const Quatd qi{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
    const Vec3d r = quat_rotate(qi, d);
    po[0] = float(r.x); po[1] = float(r.y); po[2] = float(r.z);
}

static inline float sqr_sum(float a, float b, float c) { return a * a + b * b + c * c; }   // common sqrSum

struct ScanCloud {              // previous frame's feature cloud + its kd-tree (pcl::KdTreeFLANN role)
    const float *pts = nullptr; // rows of `stride` floats: x y z intensity
    size_t stride = 4;
    int n = 0;
    KdTree tree;
    void set(const float *p, size_t stride_floats, int n_) { pts = p; stride = stride_floats; n = n_; tree.build(p, stride_floats, n_); }
    const float *at(int i) const { return pts + size_t(i) * stride; }
    int ring(int i) const { return int(at(i)[3]); }
};

struct TrackParams {
    float distance_sq_threshold = 25.0f;   // config distance_sq_threshold
    float nearby_scan = 2.5f;              // config nearby_scan
    float scan_period = 0.1f;
    double huber_delta = 0.1;              // lidar_tracker.cpp:45
    int max_outer = 2;                     // cpp:42
    int max_lm_iterations = 4;             // cpp:113
};

// feature_extract.hpp:132-270. Output: features in input order, type 'c', coeffs = [closest point, second point].
static inline void match_corner_from_scan(const ScanCloud &scan, const float *data, size_t dstride, int m, const Pose &pose_local,
                                          const TrackParams &tp, std::vector<Feature> &features)
{
    features.clear();
    for (int i = 0; i < m; ++i) {
        const float *pd = data + size_t(i) * dstride;
        float sel[3];
        transform_to_start(pd, pose_local, false, tp.scan_period, sel);
        int nn = -1; float d2 = 0.f;
        if (scan.tree.knn(sel, 1, &nn, &d2) < 1) continue;
        int closest = -1, ind2 = -1;
        if (d2 < tp.distance_sq_threshold) {
            closest = nn;
            const int id = scan.ring(closest);
            float best = tp.distance_sq_threshold;
            for (int j = closest + 1; j < scan.n; ++j) {
                if (scan.ring(j) <= id) continue;
                if (scan.ring(j) > (id + tp.nearby_scan)) break;
                const float *q = scan.at(j);
                const float dd = sqr_sum(q[0] - sel[0], q[1] - sel[1], q[2] - sel[2]);
                if (dd < best) { best = dd; ind2 = j; }
            }
            for (int j = closest - 1; j >= 0; --j) {
                if (scan.ring(j) >= id) continue;
                if (scan.ring(j) < (id - tp.nearby_scan)) break;
                const float *q = scan.at(j);
                const float dd = sqr_sum(q[0] - sel[0], q[1] - sel[1], q[2] - sel[2]);
                if (dd < best) { best = dd; ind2 = j; }
            }
        }
        if (ind2 >= 0) {
            Feature f;
            f.idx = size_t(i);
            f.point[0] = pd[0]; f.point[1] = pd[1]; f.point[2] = pd[2];
            const float *a = scan.at(closest), *b = scan.at(ind2);
            f.coeffs[0] = a[0]; f.coeffs[1] = a[1]; f.coeffs[2] = a[2];
            f.coeffs[3] = b[0]; f.coeffs[4] = b[1]; f.coeffs[5] = b[2];
            f.type = 'c';   // the reference leaves type_ at its default here; the tracker only reads point_ / coeffs_
            features.push_back(f);
        }
    }
}

// feature_extract.hpp:273-376. coeffs = [w (f32 normalised), negative_OA_dot_norm], type 's'.
static inline void match_surf_from_scan(const ScanCloud &scan, const float *data, size_t dstride, int m, const Pose &pose_local,
                                        const TrackParams &tp, std::vector<Feature> &features)
{
    features.clear();
    for (int i = 0; i < m; ++i) {
        const float *pd = data + size_t(i) * dstride;
        float sel[3];
        transform_to_start(pd, pose_local, false, tp.scan_period, sel);
        int nn = -1; float d2 = 0.f;
        if (scan.tree.knn(sel, 1, &nn, &d2) < 1) continue;
        if (!(d2 < tp.distance_sq_threshold)) continue;
        const int closest = nn, id = scan.ring(closest);
        int ind2 = -1, ind3 = -1;
        float best2 = tp.distance_sq_threshold, best3 = tp.distance_sq_threshold;
        for (int j = closest + 1; j < scan.n; ++j) {
            if (scan.ring(j) > (id + tp.nearby_scan)) break;
            const float *q = scan.at(j);
            const float dd = sqr_sum(q[0] - sel[0], q[1] - sel[1], q[2] - sel[2]);
            if (scan.ring(j) <= id && dd < best2) { best2 = dd; ind2 = j; }
            else if (scan.ring(j) > id && dd < best3) { best3 = dd; ind3 = j; }
        }
        for (int j = closest - 1; j >= 0; --j) {
            if (scan.ring(j) < (id - tp.nearby_scan)) break;
            const float *q = scan.at(j);
            const float dd = sqr_sum(q[0] - sel[0], q[1] - sel[1], q[2] - sel[2]);
            if (scan.ring(j) >= id && dd < best2) { best2 = dd; ind2 = j; }
            else if (scan.ring(j) < id && dd < best3) { best3 = dd; ind3 = j; }
        }
        if (ind2 >= 0 && ind3 >= 0) {
            const float *pj = scan.at(closest), *pl = scan.at(ind2), *pm = scan.at(ind3);
            const float a[3] = {pj[0] - pl[0], pj[1] - pl[1], pj[2] - pl[2]}, b[3] = {pj[0] - pm[0], pj[1] - pm[1], pj[2] - pm[2]};
            float w[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const float z = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];       // Eigen normalize(): only when squaredNorm > 0
            if (z > 0.f) { const float nrm = std::sqrt(z); w[0] /= nrm; w[1] /= nrm; w[2] /= nrm; }
            const float negative_OA_dot_norm = -(w[0] * pj[0] + w[1] * pj[1] + w[2] * pj[2]);
            Feature f;
            f.idx = size_t(i);
            f.point[0] = pd[0]; f.point[1] = pd[1]; f.point[2] = pd[2];
            f.coeffs[0] = w[0]; f.coeffs[1] = w[1]; f.coeffs[2] = w[2]; f.coeffs[3] = negative_OA_dot_norm;
            f.type = 's';
            features.push_back(f);
        }
    }
}

// lidar_scan_factor.hpp:24-64: one residual; J row-major 1x7 (7th column 0)
static inline void scan_plane_factor_evaluate(const double point[3], const double coeff[4], double s, const double *x, double *residual, double *J)
{
    const Pose P = pose_from_param(x);
    const Quatd q = slerp_from_identity(s, P.q);
    const Vec3d t{s * P.t.x, s * P.t.y, s * P.t.z};
    const Vec3d lp = quat_rotate(q, {point[0], point[1], point[2]});
    const double w[3] = {coeff[0], coeff[1], coeff[2]};
    residual[0] = (w[0] * (lp.x + t.x) + w[1] * (lp.y + t.y) + w[2] * (lp.z + t.z)) + coeff[3];
    if (!J) return;
    double R[9], S[9], RS[9];
    quat_to_rot(q, R);
    skew({point[0], point[1], point[2]}, S);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) RS[r * 3 + c] = R[r * 3 + 0] * S[0 * 3 + c] + R[r * 3 + 1] * S[1 * 3 + c] + R[r * 3 + 2] * S[2 * 3 + c];
    for (int c = 0; c < 3; ++c) {
        J[c] = w[c];
        J[3 + c] = -(w[0] * RS[0 * 3 + c] + w[1] * RS[1 * 3 + c] + w[2] * RS[2 * 3 + c]);
    }
    J[6] = 0.0;
}

// lidar_scan_factor.hpp:236-279: three residuals; J row-major 3x7
static inline void scan_edge_vector_factor_evaluate(const double point[3], const double coeff[6], double s, const double *x, double *residual, double *J)
{
    const Pose P = pose_from_param(x);
    const Quatd q = slerp_from_identity(s, P.q);
    const Vec3d t{s * P.t.x, s * P.t.y, s * P.t.z};
    const Vec3d r0 = quat_rotate(q, {point[0], point[1], point[2]});
    const Vec3d lp{r0.x + t.x, r0.y + t.y, r0.z + t.z};
    const Vec3d lpa{coeff[0], coeff[1], coeff[2]}, lpb{coeff[3], coeff[4], coeff[5]};
    const Vec3d nu = cross({lp.x - lpa.x, lp.y - lpa.y, lp.z - lpa.z}, {lp.x - lpb.x, lp.y - lpb.y, lp.z - lpb.z});
    const Vec3d de{lpa.x - lpb.x, lpa.y - lpb.y, lpa.z - lpb.z};
    const double den = std::sqrt(de.x * de.x + de.y * de.y + de.z * de.z);
    residual[0] = nu.x / den; residual[1] = nu.y / den; residual[2] = nu.z / den;
    if (!J) return;
    const double eta = 1.0 / den;
    double R[9], Sd[9], Sp[9], SdR[9], M[9];
    quat_to_rot(q, R);
    skew(de, Sd);
    skew({point[0], point[1], point[2]}, Sp);
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) SdR[r * 3 + c] = Sd[r * 3 + 0] * R[0 * 3 + c] + Sd[r * 3 + 1] * R[1 * 3 + c] + Sd[r * 3 + 2] * R[2 * 3 + c];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) M[r * 3 + c] = SdR[r * 3 + 0] * Sp[0 * 3 + c] + SdR[r * 3 + 1] * Sp[1 * 3 + c] + SdR[r * 3 + 2] * Sp[2 * 3 + c];
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) { J[r * 7 + c] = -eta * Sd[r * 3 + c]; J[r * 7 + 3 + c] = eta * M[r * 3 + c]; }
        J[r * 7 + 6] = 0.0;
    }
}

struct TrackOuterStat { int n_corner = 0, n_surf = 0; bool solved = false; SolveSummary solve; double pose_after[7]; };

// lidar_tracker.cpp:23-129
void track_cloud(const ScanCloud &corner_last, const ScanCloud &surf_last, const float *corner_sharp, size_t cs_stride, int n_corner,
                 const float *surf_flat, size_t sf_stride, int n_surf, const double pose_ini[7], const TrackParams &tp, double pose_out[7],
                 std::vector<TrackOuterStat> &stats);

}  // namespace orc
