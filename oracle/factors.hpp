// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (Ceres/Eigen absent;
// pinned here by analytic-vs-numeric Jacobian assertions, the reference's own check() recipe).
//
// CPU restatement of
//   LidarMapPlaneNormFactor::{ctor,Evaluate}   estimator/src/factor/lidar_map_factor.hpp:28-71
//   LidarMapEdgeFactor::{ctor,Evaluate}        estimator/src/factor/lidar_map_factor.hpp:132-174
//   PoseLocalParameterization::{Plus,ComputeJacobian}  estimator/src/factor/pose_local_parameterization.cpp:26-55
//   ceres::HuberLoss::Evaluate + Corrector (Ceres 1.12.0 loss_function.cc / corrector.cc; the
//     rho''<=0 branch is restated in the reference at marginalization_factor.cpp:60-66)
#pragma once
#include "geometry.hpp"
#include <cmath>
#include <limits>

namespace orc {

// sqrt_info_ from the 3x3 covariance trace (lidar_map_factor.hpp:35,41 / 134,140)
static inline double sqrt_info_from_trace(double trace)
{
    double s = std::sqrt(1 / trace);
    return s >= 3.0 ? 1.0 : s / 3.0;
}

// residual[1], jacobian row-major 1x7 (7th column zero). jac may be null.
static inline void plane_norm_factor_evaluate(const double point[3], const double coeff[4], double sqrt_info,
                                              const double param[7], double *residual, double *jac)
{
    Quatd q{param[3], param[4], param[5], param[6]};
    Vec3d t{param[0], param[1], param[2]};
    Vec3d w{coeff[0], coeff[1], coeff[2]};
    double d = coeff[3];
    Vec3d p{point[0], point[1], point[2]};
    Vec3d lp = quat_rotate(q, p);
    lp = {lp.x + t.x, lp.y + t.y, lp.z + t.z};
    double a = (w.x * lp.x + w.y * lp.y + w.z * lp.z) + d;
    residual[0] = sqrt_info * a;
    if (jac) {
        double R[9], S[9];
        quat_to_rot(q, R);
        skew(p, S);
        // -w^T * R
        double wr[3];
        for (int c = 0; c < 3; ++c) wr[c] = (-w.x) * R[0 * 3 + c] + (-w.y) * R[1 * 3 + c] + (-w.z) * R[2 * 3 + c];
        double jr[3];
        for (int c = 0; c < 3; ++c) jr[c] = wr[0] * S[0 * 3 + c] + wr[1] * S[1 * 3 + c] + wr[2] * S[2 * 3 + c];
        jac[0] = sqrt_info * w.x; jac[1] = sqrt_info * w.y; jac[2] = sqrt_info * w.z;
        jac[3] = sqrt_info * jr[0]; jac[4] = sqrt_info * jr[1]; jac[5] = sqrt_info * jr[2];
        jac[6] = 0.0;
    }
}

static inline void edge_factor_evaluate(const double point[3], const double coeff[6], double sqrt_info,
                                        const double param[7], double *residual, double *jac)
{
    Quatd q{param[3], param[4], param[5], param[6]};
    Vec3d t{param[0], param[1], param[2]};
    Vec3d lpa{coeff[0], coeff[1], coeff[2]};
    Vec3d lpb{coeff[3], coeff[4], coeff[5]};
    Vec3d p{point[0], point[1], point[2]};
    Vec3d lp = quat_rotate(q, p);
    lp = {lp.x + t.x, lp.y + t.y, lp.z + t.z};
    Vec3d a{lp.x - lpa.x, lp.y - lpa.y, lp.z - lpa.z};
    Vec3d b{lp.x - lpb.x, lp.y - lpb.y, lp.z - lpb.z};
    Vec3d nu = cross(a, b);
    Vec3d de{lpa.x - lpb.x, lpa.y - lpb.y, lpa.z - lpb.z};
    double nu_n = std::sqrt(nu.x * nu.x + nu.y * nu.y + nu.z * nu.z);
    double de_n = std::sqrt(de.x * de.x + de.y * de.y + de.z * de.z);
    residual[0] = sqrt_info * nu_n / de_n;
    if (jac) {
        double R[9], S[9], D[9];
        quat_to_rot(q, R);
        skew(p, S);
        skew(de, D);
        // eta = 1/|de| * nu.normalized()^T   (NaN when nu == 0, as in the reference)
        double k = 1.0 / de_n;
        double nx = nu.x, ny = nu.y, nz = nu.z;
        double n2 = nx * nx + ny * ny + nz * nz;
        if (n2 > 0.0) { double nn = std::sqrt(n2); nx /= nn; ny /= nn; nz /= nn; }
        double eta[3] = {k * nx, k * ny, k * nz};
        double eD[3];
        for (int c = 0; c < 3; ++c) eD[c] = eta[0] * D[0 * 3 + c] + eta[1] * D[1 * 3 + c] + eta[2] * D[2 * 3 + c];
        double eDR[3];
        for (int c = 0; c < 3; ++c) eDR[c] = eD[0] * R[0 * 3 + c] + eD[1] * R[1 * 3 + c] + eD[2] * R[2 * 3 + c];
        double eDRS[3];
        for (int c = 0; c < 3; ++c) eDRS[c] = eDR[0] * S[0 * 3 + c] + eDR[1] * S[1 * 3 + c] + eDR[2] * S[2 * 3 + c];
        jac[0] = sqrt_info * (-eD[0]); jac[1] = sqrt_info * (-eD[1]); jac[2] = sqrt_info * (-eD[2]);
        jac[3] = sqrt_info * eDRS[0]; jac[4] = sqrt_info * eDRS[1]; jac[5] = sqrt_info * eDRS[2];
        jac[6] = 0.0;
    }
}

// PoseLocalParameterization::Plus with the degeneracy projection V_update_ (row-major 6x6)
static inline void pose_plus(const double x[7], const double delta[6], const double V_update[36], double x_plus_delta[7])
{
    double dx[6];
    for (int r = 0; r < 6; ++r) {
        double s = 0.0;
        for (int c = 0; c < 6; ++c) s += V_update[r * 6 + c] * delta[c];
        dx[r] = s;
    }
    Quatd q{x[3], x[4], x[5], x[6]};
    Quatd dq = delta_q({dx[3], dx[4], dx[5]});
    x_plus_delta[0] = x[0] + dx[0];
    x_plus_delta[1] = x[1] + dx[1];
    x_plus_delta[2] = x[2] + dx[2];
    Quatd qn = quat_normalized(quat_mul(q, dq));
    x_plus_delta[3] = qn.x; x_plus_delta[4] = qn.y; x_plus_delta[5] = qn.z; x_plus_delta[6] = qn.w;
}

// ceres::HuberLoss(a)::Evaluate(s, rho)
static inline void huber_evaluate(double a, double s, double rho[3])
{
    const double b = a * a;
    if (s > b) {
        const double r = std::sqrt(s);
        rho[0] = 2.0 * a * r - b;
        rho[1] = std::max(std::numeric_limits<double>::min(), a / r);
        rho[2] = -rho[1] / (2.0 * s);
    } else {
        rho[0] = s; rho[1] = 1.0; rho[2] = 0.0;
    }
}

}  // namespace orc
