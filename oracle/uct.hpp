// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED.
//
// CPU restatement of
//   evalPointUncertainty / pointToFS      estimator/src/lidarMapper/associate_uct.hpp:150-215
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

}  // namespace orc
