// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). The six factor classes' Evaluate bodies and PoseLocalParameterization::Plus: PINNED (round 2) against the reference's OWN SOURCE LINES compiled over a shim (oracle/ref/, tests/test_oracle_ref_pin.py)
// -- which pins the formulas; Eigen's own arithmetic is supplied by oracle/ref/mini_eigen.hpp there. Otherwise PARITY UNPINNED (Ceres/Eigen absent;
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

// ---------------------------------------------------------------------------------------------------------------
// Odometry-side factors with three pose blocks (pivot, frame i, extrinsic):
//   LidarPureOdomPlaneNormFactor::Evaluate   estimator/src/factor/lidar_pure_odom_factor.hpp:38-102
//   LidarPureOdomEdgeFactor::Evaluate        estimator/src/factor/lidar_pure_odom_factor.hpp:209-282
// The transform applied to the point is T = T_pivot^-1 * T_i * T_ext. Jacobians are row-major 1x7 per block (7th column 0),
// restated term by term -- including the reference's asymmetry between the plane factor's extrinsic-rotation column
// (-w^T Rp^T Ri [Rext p]x) and the edge factor's (-eta [ba-bb]x Rp^T Ri (Rext [p]x + [t_ext]x)).
namespace orc {

static inline void mat3_mul(const double A[9], const double B[9], double C[9])
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}
static inline void mat3_T(const double A[9], double B[9])
{
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B[r * 3 + c] = A[c * 3 + r];
}
static inline void row_mat3(const double v[3], const double M[9], double o[3])
{
    for (int c = 0; c < 3; ++c) o[c] = v[0] * M[0 * 3 + c] + v[1] * M[1 * 3 + c] + v[2] * M[2 * 3 + c];
}
static inline void mat3_vec(const double M[9], const double v[3], double o[3])
{
    for (int r = 0; r < 3; ++r) o[r] = M[r * 3 + 0] * v[0] + M[r * 3 + 1] * v[1] + M[r * 3 + 2] * v[2];
}

struct OdomFrames {
    Quatd Qp, Qi, Qe;
    Vec3d tp, ti, te;
    double Rp[9], Ri[9], Re[9], RpT[9];
    Vec3d lp;   // Q_ext_pi * point + t_ext_pi
};

static inline void odom_frames(const double *pivot, const double *pose_i, const double *ext, const double point[3], OdomFrames &F)
{
    F.Qp = {pivot[3], pivot[4], pivot[5], pivot[6]}; F.tp = {pivot[0], pivot[1], pivot[2]};
    F.Qi = {pose_i[3], pose_i[4], pose_i[5], pose_i[6]}; F.ti = {pose_i[0], pose_i[1], pose_i[2]};
    F.Qe = {ext[3], ext[4], ext[5], ext[6]}; F.te = {ext[0], ext[1], ext[2]};
    Quatd Q_pi = quat_mul(quat_conj(F.Qp), F.Qi);
    Vec3d d{F.ti.x - F.tp.x, F.ti.y - F.tp.y, F.ti.z - F.tp.z};
    Vec3d t_pi = quat_rotate(quat_conj(F.Qp), d);
    Quatd Q_ext_pi = quat_mul(Q_pi, F.Qe);
    Vec3d r = quat_rotate(Q_pi, F.te);
    Vec3d t_ext_pi{r.x + t_pi.x, r.y + t_pi.y, r.z + t_pi.z};
    Vec3d lp = quat_rotate(Q_ext_pi, {point[0], point[1], point[2]});
    F.lp = {lp.x + t_ext_pi.x, lp.y + t_ext_pi.y, lp.z + t_ext_pi.z};
    quat_to_rot(F.Qp, F.Rp); quat_to_rot(F.Qi, F.Ri); quat_to_rot(F.Qe, F.Re);
    mat3_T(F.Rp, F.RpT);
}

// residual[1]; J0 (pivot), J1 (frame i), J2 (extrinsic): each 7 doubles or null
static inline void pure_odom_plane_evaluate(const double point[3], const double coeff[4], double sqrt_info, const double *pivot,
                                            const double *pose_i, const double *ext, double *residual, double *J0, double *J1, double *J2)
{
    OdomFrames F;
    odom_frames(pivot, pose_i, ext, point, F);
    const double w[3] = {coeff[0], coeff[1], coeff[2]};
    const double r = (w[0] * F.lp.x + w[1] * F.lp.y + w[2] * F.lp.z) + coeff[3];
    residual[0] = sqrt_info * r;
    double Rep[3], Rite[3], RiRep[3];
    mat3_vec(F.Re, point, Rep);                         // Rext * p
    const double te[3] = {F.te.x, F.te.y, F.te.z};
    mat3_vec(F.Ri, te, Rite);                           // Ri * t_ext
    mat3_vec(F.Ri, Rep, RiRep);                         // Ri * Rext * p
    double wRpT[3];
    row_mat3(w, F.RpT, wRpT);                           // w^T Rp^T
    if (J0) {
        double v[3] = {RiRep[0] + Rite[0] + F.ti.x - F.tp.x, RiRep[1] + Rite[1] + F.ti.y - F.tp.y, RiRep[2] + Rite[2] + F.ti.z - F.tp.z};
        double S[9], M[9], o[3];
        skew({v[0], v[1], v[2]}, S);
        mat3_mul(F.RpT, S, M);                          // Rp^T * skew(...)
        row_mat3(w, M, o);
        for (int k = 0; k < 3; ++k) { J0[k] = sqrt_info * (-wRpT[k]); J0[3 + k] = sqrt_info * o[k]; }
        J0[6] = 0.0;
    }
    double wRpTRi[3];
    row_mat3(wRpT, F.Ri, wRpTRi);                       // w^T Rp^T Ri
    if (J1) {
        double S[9], o[3];
        skew({Rep[0] + te[0], Rep[1] + te[1], Rep[2] + te[2]}, S);
        row_mat3(wRpTRi, S, o);
        for (int k = 0; k < 3; ++k) { J1[k] = sqrt_info * wRpT[k]; J1[3 + k] = sqrt_info * (-o[k]); }
        J1[6] = 0.0;
    }
    if (J2) {
        double S[9], o[3];
        skew({Rep[0], Rep[1], Rep[2]}, S);
        row_mat3(wRpTRi, S, o);
        for (int k = 0; k < 3; ++k) { J2[k] = sqrt_info * wRpTRi[k]; J2[3 + k] = sqrt_info * (-o[k]); }
        J2[6] = 0.0;
    }
}

static inline void pure_odom_edge_evaluate(const double point[3], const double coeff[6], double sqrt_info, const double *pivot,
                                           const double *pose_i, const double *ext, double *residual, double *J0, double *J1, double *J2)
{
    OdomFrames F;
    odom_frames(pivot, pose_i, ext, point, F);
    const Vec3d lpa{coeff[0], coeff[1], coeff[2]}, lpb{coeff[3], coeff[4], coeff[5]};
    const Vec3d ba{F.lp.x - lpa.x, F.lp.y - lpa.y, F.lp.z - lpa.z}, bb{F.lp.x - lpb.x, F.lp.y - lpb.y, F.lp.z - lpb.z};
    const Vec3d nu = cross(ba, bb);
    const Vec3d de{lpa.x - lpb.x, lpa.y - lpb.y, lpa.z - lpb.z};
    const double nu_n = std::sqrt(nu.x * nu.x + nu.y * nu.y + nu.z * nu.z), de_n = std::sqrt(de.x * de.x + de.y * de.y + de.z * de.z);
    residual[0] = sqrt_info * nu_n / de_n;
    if (!J0 && !J1 && !J2) return;
    double nx = nu.x, ny = nu.y, nz = nu.z;
    const double n2 = nx * nx + ny * ny + nz * nz;
    if (n2 > 0.0) { const double nn = std::sqrt(n2); nx /= nn; ny /= nn; nz /= nn; }
    const double k = 1.0 / de_n;
    const double eta[3] = {k * nx, k * ny, k * nz};
    double D[9], eD[3];
    skew({ba.x - bb.x, ba.y - bb.y, ba.z - bb.z}, D);   // skew(ba - bb)
    row_mat3(eta, D, eD);                               // eta * skew(ba - bb)
    double Rep[3], Rite[3], RiRep[3];
    mat3_vec(F.Re, point, Rep);
    const double te[3] = {F.te.x, F.te.y, F.te.z};
    mat3_vec(F.Ri, te, Rite);
    mat3_vec(F.Ri, Rep, RiRep);
    double eDRpT[3];
    row_mat3(eD, F.RpT, eDRpT);                         // eta skew(ba-bb) Rp^T
    if (J0) {
        double v[3] = {RiRep[0] + Rite[0] + F.ti.x - F.tp.x, RiRep[1] + Rite[1] + F.ti.y - F.tp.y, RiRep[2] + Rite[2] + F.ti.z - F.tp.z};
        double u[3], S[9], o[3];
        mat3_vec(F.RpT, v, u);                          // Rp^T * (...)
        skew({u[0], u[1], u[2]}, S);
        row_mat3(eD, S, o);                             // eta skew(ba-bb) skew(Rp^T(...))
        for (int q = 0; q < 3; ++q) { J0[q] = sqrt_info * (-eDRpT[q]); J0[3 + q] = sqrt_info * o[q]; }
        J0[6] = 0.0;
    }
    double eDRpTRi[3];
    row_mat3(eDRpT, F.Ri, eDRpTRi);
    if (J1) {
        double S[9], o[3];
        skew({Rep[0] + te[0], Rep[1] + te[1], Rep[2] + te[2]}, S);
        row_mat3(eDRpTRi, S, o);
        for (int q = 0; q < 3; ++q) { J1[q] = sqrt_info * eDRpT[q]; J1[3 + q] = sqrt_info * (-o[q]); }
        J1[6] = 0.0;
    }
    if (J2) {
        double Sp[9], St[9], M[9], o[3];
        skew({point[0], point[1], point[2]}, Sp);
        skew({te[0], te[1], te[2]}, St);
        mat3_mul(F.Re, Sp, M);                          // Rext * skew(p)
        for (int q = 0; q < 9; ++q) M[q] += St[q];      // + skew(t_ext)
        row_mat3(eDRpTRi, M, o);
        for (int q = 0; q < 3; ++q) { J2[q] = sqrt_info * eDRpTRi[q]; J2[3 + q] = sqrt_info * (-o[q]); }
        J2[6] = 0.0;
    }
}

}  // namespace orc
