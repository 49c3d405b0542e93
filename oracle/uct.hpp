// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (PCL absent): the voxel filters restate voxel_grid_covariance_mloam_impl.hpp including
// its unstable std::sort (literal; the member order inside a voxel is libstdc++'s -- see voxel_grid_mloam_plain).
//
// CPU restatement of
//   evalPointUncertainty / pointToFS      estimator/src/lidarMapper/associate_uct.hpp:150-215
//   adjointMatrix / covop1 / covop2 / compoundPoseWithCov (method 2)   associate_uct.hpp:9-28, 90-147
//   cloudUCTAssociateToMap                estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:1116-1158
//   VoxelGridCovarianceMLOAM::applyFilter mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp:68-457
//     (cov branch :296-333; the reference's only "known input" exercise is
//      mloam_test/src/test_pointiwithcov.cpp:23-38, which records no expected output)
#pragma once
#include "geometry.hpp"
#include <vector>
#include <algorithm>
#include <cmath>
#include <cfloat>
#include <cstdint>

namespace orc {

// compact PointXYZIWithCov: x y z intensity cov_vec[6](cxx cxy cxz cyy cyz czz) cov_trace  (11 floats)
struct PointICov { float x, y, z, intensity, cov[6], trace; };

// cov_point = [G diag(cov_pose, COV_MEASUREMENT) G^T]_{3x3}, G = [ (T p)^odot | T D ]   (row-major 3x3 out)
static inline void eval_point_uncertainty(const float pi[3], const double pose7[7], const double cov_pose[36],
                                          const double cov_meas[9], double cov_point[9])
{
    Pose P = pose_from_param(pose7);
    double R[9];
    quat_to_rot(P.q, R);
    // T * [p;1]
    double p[3] = {double(pi[0]), double(pi[1]), double(pi[2])};
    double tp[3];
    for (int r = 0; r < 3; ++r) tp[r] = R[r * 3 + 0] * p[0] + R[r * 3 + 1] * p[1] + R[r * 3 + 2] * p[2] + (&P.t.x)[r];
    // G (top 3 rows; the 4th row of pointToFS and of T*D is zero): 3x9
    double G[3][9];
    double S[9];
    skew({tp[0], tp[1], tp[2]}, S);
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) {
            G[r][c] = (r == c) ? 1.0 : 0.0;      // point(3) * I, point(3) = 1
            G[r][3 + c] = -S[r * 3 + c];
            G[r][6 + c] = R[r * 3 + c];          // T * D
        }
    }
    double C[9][9] = {{0}};
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) C[r][c] = cov_pose[r * 6 + c];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[6 + r][6 + c] = cov_meas[r * 3 + c];
    double GC[3][9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 9; ++c) {
            double s = 0.0;
            for (int k = 0; k < 9; ++k) s += G[r][k] * C[k][c];
            GC[r][c] = s;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
            for (int k = 0; k < 9; ++k) s += GC[r][k] * G[c][k];
            cov_point[r * 3 + c] = s;
        }
}

static inline void voxel_grid_covariance_mloam(const PointICov *in, int n, float leaf, float trace_threshold,
                                               std::vector<PointICov> &out)
{
    out.clear();
    if (n <= 0) return;
    const float inv = 1.0f / leaf;
    float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i) {
        const float v[3] = {in[i].x, in[i].y, in[i].z};
        if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2])) continue;
        for (int d = 0; d < 3; ++d) { min_p[d] = std::min(min_p[d], v[d]); max_p[d] = std::max(max_p[d], v[d]); }
    }
    int64_t dx = int64_t((max_p[0] - min_p[0]) * inv) + 1;
    int64_t dy = int64_t((max_p[1] - min_p[1]) * inv) + 1;
    int64_t dz = int64_t((max_p[2] - min_p[2]) * inv) + 1;
    if (dx * dy * dz > int64_t(INT32_MAX)) { out.assign(in, in + n); return; }
    int min_b[3], max_b[3], div_b[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = int(std::floor(min_p[d] * inv));
        max_b[d] = int(std::floor(max_p[d] * inv));
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    struct IdxPt {
        unsigned int idx, cloud_point_index;
        bool operator<(const IdxPt &o) const { return idx < o.idx; }
    };
    std::vector<IdxPt> iv;
    iv.reserve(n);
    for (int i = 0; i < n; ++i) {
        int ijk0 = int(std::floor(in[i].x * inv) - float(min_b[0]));
        int ijk1 = int(std::floor(in[i].y * inv) - float(min_b[1]));
        int ijk2 = int(std::floor(in[i].z * inv) - float(min_b[2]));
        iv.push_back({(unsigned)(ijk0 + ijk1 * mul1 + ijk2 * mul2), (unsigned)i});
    }
    std::sort(iv.begin(), iv.end(), std::less<IdxPt>());
    size_t index = 0;
    while (index < iv.size()) {
        size_t i2 = index + 1;
        while (i2 < iv.size() && iv[i2].idx == iv[index].idx) ++i2;
        float mu[3] = {0, 0, 0}, ity = 0, cov[7] = {0, 0, 0, 0, 0, 0, 0}, weight_total = 0, w_max = 0;
        for (size_t i = index; i < i2; ++i) {
            const PointICov &q = in[iv[i].cloud_point_index];
            float tr = q.cov[0] + q.cov[3] + q.cov[5];
            if (std::fabs(tr) >= trace_threshold) continue;
            float w = trace_threshold - tr;
            mu[0] += w * q.x; mu[1] += w * q.y; mu[2] += w * q.z;
            ity = w > w_max ? q.intensity : ity;
            w_max = w > w_max ? w : w_max;
            for (int k = 0; k < 6; ++k) cov[k] += w * w * q.cov[k];
            cov[6] += w * w * q.trace;
            weight_total += w;
        }
        if (weight_total == 0) weight_total = 1.0f;
        PointICov o;
        o.x = mu[0] / weight_total; o.y = mu[1] / weight_total; o.z = mu[2] / weight_total;
        o.intensity = ity;
        float wt2 = weight_total * weight_total;
        for (int k = 0; k < 6; ++k) o.cov[k] = cov[k] / wt2;
        o.trace = o.cov[0] + o.cov[3] + o.cov[5];
        out.push_back(o);
        index = i2;
    }
}

// The filter's other branch: a point type WITHOUT covariance fields (pcl::VoxelGridCovarianceMLOAM<PointI>, the mapper's down_size_filter_surf /
// _corner, lidar_mapper_keyframe.cpp:79, 359-364) -- voxel_grid_covariance_mloam_impl.hpp:393-431: xyz summed over the members and divided by
// their count, intensity = the LAST member's (centroid[3] = temporary[3] inside the loop, no averaging). "Last" is in the order
// std::sort leaves the members of a voxel in, and the comparator looks at the voxel index only: for a voxel whose members carry different
// intensities (a fused multi-LiDAR cloud: intensity = LiDAR id, which downsampleCurrentScan then uses to pick the extrinsic) the reference's
// result depends on libstdc++'s introsort. member_order 0 = exactly that (std::sort, comparator on idx only); 1 = members in point-index
// order (what a stable sort gives) -- the rule the HIP path implements and is pinned on.
static inline void voxel_grid_mloam_plain(const float *xyzi, int n, float leaf, int member_order, std::vector<float> &out)
{
    out.clear();
    if (n <= 0) return;
    const float inv = 1.0f / leaf;
    float min_p[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, max_p[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = 0; i < n; ++i) {
        const float *v = xyzi + 4 * size_t(i);
        if (!std::isfinite(v[0]) || !std::isfinite(v[1]) || !std::isfinite(v[2])) continue;
        for (int d = 0; d < 3; ++d) { min_p[d] = std::min(min_p[d], v[d]); max_p[d] = std::max(max_p[d], v[d]); }
    }
    int64_t dx = int64_t((max_p[0] - min_p[0]) * inv) + 1;
    int64_t dy = int64_t((max_p[1] - min_p[1]) * inv) + 1;
    int64_t dz = int64_t((max_p[2] - min_p[2]) * inv) + 1;
    if (dx * dy * dz > int64_t(INT32_MAX)) { out.assign(xyzi, xyzi + 4 * size_t(n)); return; }
    int min_b[3], max_b[3], div_b[3];
    for (int d = 0; d < 3; ++d) {
        min_b[d] = int(std::floor(min_p[d] * inv));
        max_b[d] = int(std::floor(max_p[d] * inv));
        div_b[d] = max_b[d] - min_b[d] + 1;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    struct IdxPt {
        unsigned int idx, cloud_point_index;
        bool operator<(const IdxPt &o) const { return idx < o.idx; }
    };
    std::vector<IdxPt> iv;
    iv.reserve(n);
    for (int i = 0; i < n; ++i) {
        const float *v = xyzi + 4 * size_t(i);
        int ijk0 = int(std::floor(v[0] * inv) - float(min_b[0]));
        int ijk1 = int(std::floor(v[1] * inv) - float(min_b[1]));
        int ijk2 = int(std::floor(v[2] * inv) - float(min_b[2]));
        iv.push_back({(unsigned)(ijk0 + ijk1 * mul1 + ijk2 * mul2), (unsigned)i});
    }
    if (member_order == 0) std::sort(iv.begin(), iv.end(), std::less<IdxPt>());
    else std::stable_sort(iv.begin(), iv.end(), std::less<IdxPt>());
    size_t index = 0;
    while (index < iv.size()) {
        size_t i2 = index + 1;
        while (i2 < iv.size() && iv[i2].idx == iv[index].idx) ++i2;
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        for (size_t i = index; i < i2; ++i) {
            const float *q = xyzi + 4 * size_t(iv[i].cloud_point_index);
            c[0] += q[0]; c[1] += q[1]; c[2] += q[2];
            c[3] = q[3];
        }
        const float cnt = float(i2 - index);
        out.push_back(c[0] / cnt); out.push_back(c[1] / cnt); out.push_back(c[2] / cnt); out.push_back(c[3]);
        index = i2;
    }
}

// ---- 3x3 / 6x6 helpers (row-major)
static inline void m3_mul(const double A[9], const double B[9], double C[9])
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0.0; for (int k = 0; k < 3; ++k) s += A[r * 3 + k] * B[k * 3 + c]; C[r * 3 + c] = s; }
}
static inline void m3_T(const double A[9], double B[9]) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[r * 3 + c] = A[c * 3 + r]; }
static inline void m6_mul(const double A[36], const double B[36], double C[36])
{
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0.0; for (int k = 0; k < 6; ++k) s += A[r * 6 + k] * B[k * 6 + c]; C[r * 6 + c] = s; }
}
static inline void m6_T(const double A[36], double B[36]) { for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) B[r * 6 + c] = A[c * 6 + r]; }
static inline void m6_block(const double A[36], int r0, int c0, double B[9]) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[r * 3 + c] = A[(r0 + r) * 6 + c0 + c]; }
static inline void m6_set_block(double A[36], int r0, int c0, const double B[9]) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A[(r0 + r) * 6 + c0 + c] = B[r * 3 + c]; }

// associate_uct.hpp:18-28
static inline void covop1(const double B[9], double A[9])
{
    const double tr = B[0] + B[4] + B[8];
    for (int i = 0; i < 9; ++i) A[i] = B[i];
    A[0] -= tr; A[4] -= tr; A[8] -= tr;
}
static inline void covop2(const double B[9], const double C[9], double A[9])
{
    double b1[9], c1[9], p[9], cb[9], cb1[9];
    covop1(B, b1); covop1(C, c1);
    m3_mul(b1, c1, p);
    m3_mul(C, B, cb);
    covop1(cb, cb1);
    for (int i = 0; i < 9; ++i) A[i] = p[i] + cb1[i];
}

// associate_uct.hpp:90-147, method 2 (the default every caller uses). pose = [t, q]; cov 6x6 row-major.
static inline void compound_pose_with_cov(const double pose1[7], const double cov1[36], const double pose2[7], const double cov2[36],
                                          double pose_cp[7], double cov_cp[36])
{
    const Pose P1 = pose_from_param(pose1), P2 = pose_from_param(pose2);
    const Quatd q = quat_mul(P1.q, P2.q);
    const Vec3d rt = quat_rotate(P1.q, P2.t);
    pose_cp[0] = rt.x + P1.t.x; pose_cp[1] = rt.y + P1.t.y; pose_cp[2] = rt.z + P1.t.z;
    pose_cp[3] = q.x; pose_cp[4] = q.y; pose_cp[5] = q.z; pose_cp[6] = q.w;
    // adjointMatrix(T1): [R, [t]x R; 0, R]
    double R[9], S[9], SR[9], Ad[36] = {0}, AdT[36], tmp[36], c2p[36];
    quat_to_rot(P1.q, R);
    skew(P1.t, S);
    m3_mul(S, R, SR);
    m6_set_block(Ad, 0, 0, R); m6_set_block(Ad, 0, 3, SR); m6_set_block(Ad, 3, 3, R);
    m6_T(Ad, AdT);
    m6_mul(Ad, cov2, tmp);
    m6_mul(tmp, AdT, c2p);
    double c1rr[9], c1rp[9], c1pp[9], c2rr[9], c2rp[9], c2pp[9], c1rpT[9], c2rpT[9];
    m6_block(cov1, 0, 0, c1rr); m6_block(cov1, 0, 3, c1rp); m6_block(cov1, 3, 3, c1pp);
    m6_block(c2p, 0, 0, c2rr); m6_block(c2p, 0, 3, c2rp); m6_block(c2p, 3, 3, c2pp);
    m3_T(c1rp, c1rpT); m3_T(c2rp, c2rpT);
    double A1[36] = {0}, A2[36] = {0}, t3[9], s3[9];
    covop1(c1pp, t3); m6_set_block(A1, 0, 0, t3); m6_set_block(A1, 3, 3, t3);
    for (int i = 0; i < 9; ++i) s3[i] = c1rp[i] + c1rpT[i];
    covop1(s3, t3); m6_set_block(A1, 0, 3, t3);
    covop1(c2pp, t3); m6_set_block(A2, 0, 0, t3); m6_set_block(A2, 3, 3, t3);
    for (int i = 0; i < 9; ++i) s3[i] = c2rp[i] + c2rpT[i];
    covop1(s3, t3); m6_set_block(A2, 0, 3, t3);
    double Brr[9], Brp[9], Bpp[9], u[9], BrpT[9];
    covop2(c1pp, c2rr, Brr);
    covop2(c1rpT, c2rp, u); for (int i = 0; i < 9; ++i) Brr[i] += u[i];
    covop2(c1rp, c2rpT, u); for (int i = 0; i < 9; ++i) Brr[i] += u[i];
    covop2(c1rr, c2pp, u); for (int i = 0; i < 9; ++i) Brr[i] += u[i];
    covop2(c1pp, c2rpT, Brp);
    covop2(c1rpT, c2pp, u); for (int i = 0; i < 9; ++i) Brp[i] += u[i];
    covop2(c1pp, c2pp, Bpp);
    m3_T(Brp, BrpT);
    double B[36] = {0};
    m6_set_block(B, 0, 0, Brr); m6_set_block(B, 0, 3, Brp); m6_set_block(B, 3, 0, BrpT); m6_set_block(B, 3, 3, Bpp);
    double A1T[36], A2T[36], m1[36], m2[36], m3[36], m4[36];
    m6_T(A1, A1T); m6_T(A2, A2T);
    m6_mul(A1, c2p, m1); m6_mul(c2p, A1T, m2); m6_mul(A2, cov1, m3); m6_mul(cov1, A2T, m4);
    for (int i = 0; i < 36; ++i) cov_cp[i] = cov1[i] + c2p[i] + (((m1[i] + m2[i]) + m3[i]) + m4[i]) / 12 + B[i] / 4;
}

// lidar_mapper_keyframe.cpp:1116-1158. ext: n_laser x [t,q], ext_cov: n_laser x 36.
static inline void cloud_uct_associate_to_map(const PointICov *in, int n, const double pose_global[7], const double cov_global[36],
                                              const double *ext, const double *ext_cov, int n_laser, const double cov_meas[9],
                                              bool with_ua, double trace_threshold, std::vector<PointICov> &out)
{
    std::vector<double> pc(size_t(n_laser) * 7), cc(size_t(n_laser) * 36);
    for (int l = 0; l < n_laser; ++l)
        compound_pose_with_cov(pose_global, cov_global, ext + l * 7, ext_cov + l * 36, pc.data() + l * 7, cc.data() + l * 36);
    const Pose Pg = pose_from_param(pose_global);
    out.clear();
    for (int i = 0; i < n; ++i) {
        const PointICov &po = in[i];
        const int ind = int(po.intensity);
        double cov_point[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (with_ua) {
            const Pose Pe = pose_inverse(pose_from_param(ext + ind * 7));
            const Vec3d s = quat_rotate(Pe.q, {double(po.x), double(po.y), double(po.z)});
            const float sel[3] = {float(s.x + Pe.t.x), float(s.y + Pe.t.y), float(s.z + Pe.t.z)};   // pointAssociateToMap stores f32
            eval_point_uncertainty(sel, pc.data() + ind * 7, cc.data() + ind * 36, cov_meas, cov_point);
            if (cov_point[0] + cov_point[4] + cov_point[8] > trace_threshold) continue;
        }
        PointICov o = po;
        const Vec3d g = quat_rotate(Pg.q, {double(po.x), double(po.y), double(po.z)});
        o.x = float(g.x + Pg.t.x); o.y = float(g.y + Pg.t.y); o.z = float(g.z + Pg.t.z);
        o.cov[0] = float(cov_point[0]); o.cov[1] = float(cov_point[1]); o.cov[2] = float(cov_point[2]);
        o.cov[3] = float(cov_point[4]); o.cov[4] = float(cov_point[5]); o.cov[5] = float(cov_point[8]);
        o.trace = float(cov_point[0] + cov_point[4] + cov_point[8]);
        out.push_back(o);
    }
}

}  // namespace orc
