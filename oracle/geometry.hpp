// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (Eigen absent).
// f64 quaternion / SE3 helpers restating the Eigen::Quaterniond operations the reference
// uses (q*v via the 2*cross form of QuaternionBase::_transformVector, toRotationMatrix,
// Hamilton product, normalized) and Utility::{deltaQ,skewSymmetric}
// (estimator/src/utility/utility.h:173-195).
#pragma once
#include <cmath>

namespace orc {

struct Vec3d { double x, y, z; };
struct Quatd { double x, y, z, w; };

// pose parameter block layout everywhere: [tx ty tz qx qy qz qw]  (lidar_map_factor.hpp:46-47)
struct Pose {
    Quatd q{0, 0, 0, 1};
    Vec3d t{0, 0, 0};
};

static inline Pose pose_from_param(const double *p)
{
    Pose P;
    P.t = {p[0], p[1], p[2]};
    P.q = {p[3], p[4], p[5], p[6]};
    return P;
}

static inline Vec3d cross(const Vec3d &a, const Vec3d &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Eigen: uv = q.vec x v; uv += uv; return v + w*uv + q.vec x uv
static inline Vec3d quat_rotate(const Quatd &q, const Vec3d &v)
{
    Vec3d qv{q.x, q.y, q.z};
    Vec3d uv = cross(qv, v);
    uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
    Vec3d c2 = cross(qv, uv);
    return {v.x + q.w * uv.x + c2.x, v.y + q.w * uv.y + c2.y, v.z + q.w * uv.z + c2.z};
}

static inline Quatd quat_mul(const Quatd &a, const Quatd &b)
{
    Quatd r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

static inline Quatd quat_normalized(const Quatd &q)
{
    double n2 = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (n2 > 0.0) { double n = std::sqrt(n2); return {q.x / n, q.y / n, q.z / n, q.w / n}; }
    return q;
}

static inline Quatd quat_conj(const Quatd &q) { return {-q.x, -q.y, -q.z, q.w}; }

// Eigen::Quaterniond::toRotationMatrix, row-major R[r*3+c]
static inline void quat_to_rot(const Quatd &q, double R[9])
{
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// Utility::deltaQ (utility.h:173-185): [theta/2, 1], NOT normalised
static inline Quatd delta_q(const Vec3d &theta) { return {theta.x / 2.0, theta.y / 2.0, theta.z / 2.0, 1.0}; }

// Utility::skewSymmetric (utility.h:187-195), row-major
static inline void skew(const Vec3d &v, double S[9])
{
    S[0] = 0;    S[1] = -v.z; S[2] = v.y;
    S[3] = v.z;  S[4] = 0;    S[5] = -v.x;
    S[6] = -v.y; S[7] = v.x;  S[8] = 0;
}

static inline Pose pose_inverse(const Pose &P)
{
    Pose r;
    r.q = quat_conj(P.q);
    Vec3d mt = quat_rotate(r.q, P.t);
    r.t = {-mt.x, -mt.y, -mt.z};
    return r;
}

static inline Pose pose_mul(const Pose &A, const Pose &B)
{
    Pose r;
    r.q = quat_mul(A.q, B.q);
    Vec3d rt = quat_rotate(A.q, B.t);
    r.t = {rt.x + A.t.x, rt.y + A.t.y, rt.z + A.t.z};
    return r;
}

// The pose the mapper starts frame k+1 from (lidar_mapper_keyframe.cpp:145-160): transformUpdate() after frame k, pose_wmap_wodom = pose_wmap_curr *
// pose_wodom_curr.inverse(); transformAssociateToMap() before frame k+1, pose_wmap_curr = pose_wmap_wodom * pose_wodom_curr. Pose::operator* (pose.cpp:110-113)
// and Pose::inverse (pose.cpp:99-102) both go through Pose(q, t) (pose.cpp:34-41), which normalises the quaternion; Quaterniond::inverse is conjugate / squaredNorm.
static inline Pose pose_ctor_qt(const Quatd &q, const Vec3d &t) { Pose r; r.q = quat_normalized(q); r.t = t; return r; }
static inline Pose pose_inverse_ref(const Pose &P)
{
    const double n2 = P.q.x * P.q.x + P.q.y * P.q.y + P.q.z * P.q.z + P.q.w * P.q.w;
    Quatd qi = n2 > 0.0 ? Quatd{-P.q.x / n2, -P.q.y / n2, -P.q.z / n2, P.q.w / n2} : Quatd{0, 0, 0, 0};
    Vec3d mt = quat_rotate(qi, P.t);
    return pose_ctor_qt(qi, Vec3d{-mt.x, -mt.y, -mt.z});
}
static inline Pose pose_mul_ref(const Pose &A, const Pose &B)
{
    Vec3d rt = quat_rotate(A.q, B.t);
    return pose_ctor_qt(quat_mul(A.q, B.q), Vec3d{rt.x + A.t.x, rt.y + A.t.y, rt.z + A.t.z});
}
static inline Pose pose_chain(const Pose &wmap_curr_prev, const Pose &wodom_prev, const Pose &wodom_cur)
{
    const Pose wmap_wodom = pose_mul_ref(wmap_curr_prev, pose_inverse_ref(wodom_prev));      // transformUpdate
    return pose_mul_ref(wmap_wodom, wodom_cur);                                              // transformAssociateToMap
}

}  // namespace orc
