// Device-side small math for the scan-to-map kernels (gfx950). Everything here is written for register residency:
// no dynamically indexed local arrays (those would go to scratch), loops fully unrolled, f32 paths kept un-fused
// (the translation unit is compiled with -ffp-contract=off) so that discrete decisions (line test, plane gate,
// 5th-neighbour gate) are bit-identical with the CPU reference path.
//
// Algorithms follow the libraries the reference calls at these sites:
//   3x3 f32 symmetric eigen-decomposition  = Eigen::SelfAdjointEigenSolver<Matrix3f>::compute (feature_extract.hpp:427, 688)
//   5x3 f32 least squares                  = Eigen::ColPivHouseholderQR<MatrixXf>::solve      (feature_extract.hpp:579, 823)
//   q*v, toRotationMatrix, q1*q2           = Eigen::Quaterniond                                (lidar_map_factor.hpp:46-58)
#pragma once
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

namespace mlh {

// ------------------------------------------------------------------ f64 quaternion helpers
struct d3 { double x, y, z; };
struct q4 { double x, y, z, w; };

__host__ __device__ inline d3 cross3(const d3 &a, const d3 &b)
{
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Eigen QuaternionBase::_transformVector: v + w*(2 q x v) + q x (2 q x v)
__host__ __device__ inline d3 qrot(const q4 &q, const d3 &v)
{
    d3 qv{q.x, q.y, q.z};
    d3 uv = cross3(qv, v);
    uv.x += uv.x; uv.y += uv.y; uv.z += uv.z;
    d3 c2 = cross3(qv, uv);
    return {v.x + q.w * uv.x + c2.x, v.y + q.w * uv.y + c2.y, v.z + q.w * uv.z + c2.z};
}

__host__ __device__ inline q4 qmul(const q4 &a, const q4 &b)
{
    q4 r;
    r.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
    r.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
    r.y = a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z;
    r.z = a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x;
    return r;
}

// row-major R[9]
__host__ __device__ inline void qtorot(const q4 &q, double *R)
{
    const double tx = 2.0 * q.x, ty = 2.0 * q.y, tz = 2.0 * q.z;
    const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const double txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// PoseLocalParameterization::Plus (pose_local_parameterization.cpp:26-45): dx' = V dx, t += dx'_t,
// q = (q * [dx'_theta/2, 1]).normalized()
__host__ __device__ inline void pose_plus(const double *x, const double *delta, const double *V, double *out)
{
    double dx[6];
    for (int r = 0; r < 6; ++r) {
        double s = 0.0;
        for (int c = 0; c < 6; ++c) s += (V ? V[r * 6 + c] : (r == c ? 1.0 : 0.0)) * delta[c];
        dx[r] = s;
    }
    q4 q{x[3], x[4], x[5], x[6]};
    q4 dq{dx[3] / 2.0, dx[4] / 2.0, dx[5] / 2.0, 1.0};
    q4 p = qmul(q, dq);
    double n2 = p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
    if (n2 > 0.0) { double n = sqrt(n2); p.x /= n; p.y /= n; p.z /= n; p.w /= n; }
    out[0] = x[0] + dx[0]; out[1] = x[1] + dx[1]; out[2] = x[2] + dx[2];
    out[3] = p.x; out[4] = p.y; out[5] = p.z; out[6] = p.w;
}

// ------------------------------------------------------------------ 3x3 symmetric eigen-decomposition, f32
// Eigen's JacobiRotation::makeGivens (real case), branch-free: the lanes of a wavefront disagree on |p| > |q| all the time, and with branches
// the wavefront paid for both sides (two divisions and a square root each). One quotient, one root, one reciprocal serve both cases with
// exactly the operands and roundings the taken branch of the reference would use:
//   |p| > |q|:  t = q / p, u = +-sqrt(1 + t^2) (sign of p), c = 1 / u,  s = -t * c
//   otherwise:  t = p / q, u = +-sqrt(1 + t^2) (sign of q), s = -1 / u, c = -t * s        (-1 / u == -(1 / u) bit for bit)
// q == 0 and p == 0 are patched in afterwards (their quotients, possibly NaN, are discarded).
__device__ inline void givens_f(float p, float q, float &c, float &s)
{
    const bool big_p = fabsf(p) > fabsf(q);
    const float num = big_p ? q : p, den = big_p ? p : q;
    const float t = num / den;
    float u = sqrtf(1.f + t * t);
    if (den < 0.f) u = -u;
    const float r = 1.f / u;
    const float s_b = -r;
    c = big_p ? r : -t * s_b;
    s = big_p ? -t * r : s_b;
    if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; }
    if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; }
}

__device__ inline float hypot_scaled_f(float x, float y)
{
    float ax = fabsf(x), ay = fabsf(y);
    float p = ax > ay ? ax : ay;
    float qn = ax > ay ? ay : ax;
    if (p == 0.f) return 0.f;
    float qp = qn / p;
    return p * sqrtf(1.f + qp * qp);
}

// Wilkinson shift from the trailing 2x2 {da, db; e}
__device__ inline float wilkinson_mu_f(float da, float db, float e)
{
    float td = (da - db) * 0.5f;
    float mu = db;
    if (td == 0.f) mu -= fabsf(e);
    else {
        float e2 = e * e;
        float h = hypot_scaled_f(td, e);
        if (e2 == 0.f) mu -= (e / (td + (td > 0.f ? 1.f : -1.f))) * (e / h);
        else mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    return mu;
}

// one Givens similarity on the tridiagonal entries (dk, dk1, sk) given (x, z); returns c, s
__device__ inline void tri_rot_f(float &dk, float &dk1, float &sk, float x, float z, float &c, float &s)
{
    givens_f(x, z, c, s);
    float sdk = s * dk + c * sk;
    float dkp1 = s * sk + c * dk1;
    float ndk = c * (c * dk - s * sk) - s * (c * sk - s * dk1);
    dk1 = s * sdk + c * dkp1;
    sk = c * sdk - s * dkp1;
    dk = ndk;
}

__device__ inline void rot_cols_f(float &a0, float &a1, float &a2, float &b0, float &b1, float &b2, float c, float s)
{
    float x, y;
    x = a0; y = b0; a0 = c * x - s * y; b0 = s * x + c * y;
    x = a1; y = b1; a1 = c * x - s * y; b1 = s * x + c * y;
    x = a2; y = b2; a2 = c * x - s * y; b2 = s * x + c * y;
}

// Input: lower triangle a00,a10,a11,a20,a21,a22. Output: ascending eigenvalues l0<=l1<=l2 and the eigenvector of the
// LARGEST eigenvalue (vx,vy,vz). Returns false when the iteration limit was hit.
__device__ inline bool eig3_largest_f(float a00, float a10, float a11, float a20, float a21, float a22,
                                      float &l0, float &l1, float &l2, float &vx, float &vy, float &vz)
{
    float scale = fmaxf(fmaxf(fmaxf(fabsf(a00), fabsf(a10)), fmaxf(fabsf(a11), fabsf(a20))), fmaxf(fabsf(a21), fabsf(a22)));
    if (scale == 0.f) scale = 1.f;
    a00 /= scale; a10 /= scale; a11 /= scale; a20 /= scale; a21 /= scale; a22 /= scale;
    float d0, d1, d2, s0, s1;
    // Q columns: (q00,q10,q20), (q01,q11,q21), (q02,q12,q22)
    float q00, q10, q20, q01, q11, q21, q02, q12, q22;
    d0 = a00;
    float v1norm2 = a20 * a20;
    if (v1norm2 <= FLT_MIN) {
        d1 = a11; d2 = a22; s0 = a10; s1 = a21;
        q00 = 1.f; q10 = 0.f; q20 = 0.f; q01 = 0.f; q11 = 1.f; q21 = 0.f; q02 = 0.f; q12 = 0.f; q22 = 1.f;
    } else {
        float beta = sqrtf(a10 * a10 + v1norm2);
        float invBeta = 1.f / beta;
        float m01 = a10 * invBeta;
        float m02 = a20 * invBeta;
        float q = 2.f * m01 * a21 + m02 * (a22 - a11);
        d1 = a11 + m02 * q;
        d2 = a22 - m02 * q;
        s0 = beta;
        s1 = a21 - m01 * q;
        q00 = 1.f; q10 = 0.f; q20 = 0.f;
        q01 = 0.f; q11 = m01; q21 = m02;
        q02 = 0.f; q12 = m02; q22 = -m01;
    }
    const float precision = 2.f * FLT_EPSILON;
    int iter = 0;
    bool ok = true;
    while (true) {
        if (fabsf(s0) <= (fabsf(d0) + fabsf(d1)) * precision || fabsf(s0) <= FLT_MIN) s0 = 0.f;
        if (fabsf(s1) <= (fabsf(d1) + fabsf(d2)) * precision || fabsf(s1) <= FLT_MIN) s1 = 0.f;
        int end = (s1 != 0.f) ? 2 : ((s0 != 0.f) ? 1 : 0);
        if (end == 0) break;
        iter++;
        if (iter > 90) { ok = false; break; }
        // One implicit-shift QL/QR sweep on the unreduced block. Three cases: (A) the whole 3x3 (rotations on (0,1) then
        // (1,2)), (B) the trailing 2x2, (C) the leading 2x2. Written as two predicated rotations instead of three branches,
        // so lanes of a wavefront in different cases share the instruction stream; per lane the arithmetic (operands and
        // order) is exactly that of its case.
        const bool caseC = (end == 1);
        const bool first01 = caseC || (s0 != 0.f);           // A or C: rotate (0,1) first
        const bool caseA = (end == 2) && (s0 != 0.f);
        const float mu = wilkinson_mu_f(caseC ? d0 : d1, caseC ? d1 : d2, caseC ? s0 : s1);
        float c, s, x = d1 - mu, z = s1;                      // case B's rotation input
        if (first01) {
            tri_rot_f(d0, d1, s0, d0 - mu, s0, c, s);
            rot_cols_f(q00, q10, q20, q01, q11, q21, c, s);
            if (caseA) {                                      // the bulge to chase into (1,2)
                x = s0;
                z = -s * s1;
                s1 = c * s1;
            }
        }
        if (!caseC) {
            tri_rot_f(d1, d2, s1, x, z, c, s);
            if (caseA) s0 = c * s0 - s * z;
            rot_cols_f(q01, q11, q21, q02, q12, q22, c, s);
        }
    }
    // selection sort ascending (swap eigenvalues and columns), as Eigen does when info == Success
    if (ok) {
        // i = 0: argmin over (d0,d1,d2), first minimum wins
        int k = 0; float mn = d0;
        if (d1 < mn) { mn = d1; k = 1; }
        if (d2 < mn) { mn = d2; k = 2; }
        if (k == 1) { float t = d0; d0 = d1; d1 = t;
                      t = q00; q00 = q01; q01 = t; t = q10; q10 = q11; q11 = t; t = q20; q20 = q21; q21 = t; }
        else if (k == 2) { float t = d0; d0 = d2; d2 = t;
                      t = q00; q00 = q02; q02 = t; t = q10; q10 = q12; q12 = t; t = q20; q20 = q22; q22 = t; }
        // i = 1
        if (d2 < d1) { float t = d1; d1 = d2; d2 = t;
                      t = q01; q01 = q02; q02 = t; t = q11; q11 = q12; q12 = t; t = q21; q21 = q22; q22 = t; }
    }
    l0 = d0 * scale; l1 = d1 * scale; l2 = d2 * scale;
    vx = q02; vy = q12; vz = q22;
    return ok;
}

// ------------------------------------------------------------------ 5x3 column-pivoted Householder QR solve, f32
// Solves min || A n - b ||, A rows = (ax[r], ay[r], az[r]), b[r] = -1 (the plane fit of matchSurf*FromMap).
template <int ROWS>
__device__ inline void plane_fit_qr_f(const float (&ax)[ROWS], const float (&ay)[ROWS], const float (&az)[ROWS],
                                      float &nx, float &ny, float &nz)
{
    float c0[ROWS], c1[ROWS], c2[ROWS];   // working columns
#pragma unroll
    for (int r = 0; r < ROWS; ++r) { c0[r] = ax[r]; c1[r] = ay[r]; c2[r] = az[r]; }
    float nu0, nu1, nu2, nd0, nd1, nd2;   // colNormsUpdated / colNormsDirect
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int r = 0; r < ROWS; ++r) { s0 += c0[r] * c0[r]; s1 += c1[r] * c1[r]; s2 += c2[r] * c2[r]; }
        nd0 = sqrtf(s0); nd1 = sqrtf(s1); nd2 = sqrtf(s2);
        nu0 = nd0; nu1 = nd1; nu2 = nd2;
    }
    const float maxn = fmaxf(nu0, fmaxf(nu1, nu2));
    const float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / float(ROWS);
    const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
    int nonzero_pivots = 3;
    int p0 = 0, p1 = 1, p2 = 2;           // column permutation indices
    float tau0, tau1, tau2;

    // ---------------- k = 0
    {
        int big = 0; float bn = nu0;
        if (nu1 > bn) { bn = nu1; big = 1; }
        if (nu2 > bn) { bn = nu2; big = 2; }
        if (nonzero_pivots == 3 && bn * bn < threshold_helper * float(ROWS - 0)) nonzero_pivots = 0;
        if (big == 1) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { float t = c0[r]; c0[r] = c1[r]; c1[r] = t; }
            float t = nu0; nu0 = nu1; nu1 = t; t = nd0; nd0 = nd1; nd1 = t;
            int ti = p0; p0 = p1; p1 = ti;
        } else if (big == 2) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { float t = c0[r]; c0[r] = c2[r]; c2[r] = t; }
            float t = nu0; nu0 = nu2; nu2 = t; t = nd0; nd0 = nd2; nd2 = t;
            int ti = p0; p0 = p2; p2 = ti;
        }
        float tailSq = 0.f;
#pragma unroll
        for (int r = 1; r < ROWS; ++r) tailSq += c0[r] * c0[r];
        float c00 = c0[0], beta;
        if (tailSq <= FLT_MIN) {
            tau0 = 0.f; beta = c00;
#pragma unroll
            for (int r = 1; r < ROWS; ++r) c0[r] = 0.f;
        } else {
            beta = sqrtf(c00 * c00 + tailSq);
            if (c00 >= 0.f) beta = -beta;
            float den = c00 - beta;
#pragma unroll
            for (int r = 1; r < ROWS; ++r) c0[r] = c0[r] / den;
            tau0 = (beta - c00) / beta;
        }
        c0[0] = beta;
        if (tau0 != 0.f) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int r = 1; r < ROWS; ++r) { t1 += c0[r] * c1[r]; t2 += c0[r] * c2[r]; }
            t1 += c1[0]; t2 += c2[0];
            c1[0] -= tau0 * t1; c2[0] -= tau0 * t2;
#pragma unroll
            for (int r = 1; r < ROWS; ++r) { c1[r] -= tau0 * c0[r] * t1; c2[r] -= tau0 * c0[r] * t2; }
        }
        // norm down-dating for columns 1, 2 (row 0 removed)
        if (nu1 != 0.f) {
            float temp = fabsf(c1[0]) / nu1;
            temp = (1.f + temp) * (1.f - temp);
            temp = temp < 0.f ? 0.f : temp;
            float ratio = nu1 / nd1;
            float temp2 = temp * (ratio * ratio);
            if (temp2 <= norm_downdate_threshold) {
                float s = 0.f;
#pragma unroll
                for (int r = 1; r < ROWS; ++r) s += c1[r] * c1[r];
                nd1 = sqrtf(s); nu1 = nd1;
            } else nu1 *= sqrtf(temp);
        }
        if (nu2 != 0.f) {
            float temp = fabsf(c2[0]) / nu2;
            temp = (1.f + temp) * (1.f - temp);
            temp = temp < 0.f ? 0.f : temp;
            float ratio = nu2 / nd2;
            float temp2 = temp * (ratio * ratio);
            if (temp2 <= norm_downdate_threshold) {
                float s = 0.f;
#pragma unroll
                for (int r = 1; r < ROWS; ++r) s += c2[r] * c2[r];
                nd2 = sqrtf(s); nu2 = nd2;
            } else nu2 *= sqrtf(temp);
        }
    }
    // ---------------- k = 1
    {
        int big = 1; float bn = nu1;
        if (nu2 > bn) { bn = nu2; big = 2; }
        if (nonzero_pivots == 3 && bn * bn < threshold_helper * float(ROWS - 1)) nonzero_pivots = 1;
        if (big == 2) {
#pragma unroll
            for (int r = 0; r < ROWS; ++r) { float t = c1[r]; c1[r] = c2[r]; c2[r] = t; }
            float t = nu1; nu1 = nu2; nu2 = t; t = nd1; nd1 = nd2; nd2 = t;
            int ti = p1; p1 = p2; p2 = ti;
        }
        float tailSq = 0.f;
#pragma unroll
        for (int r = 2; r < ROWS; ++r) tailSq += c1[r] * c1[r];
        float c11 = c1[1], beta;
        if (tailSq <= FLT_MIN) {
            tau1 = 0.f; beta = c11;
#pragma unroll
            for (int r = 2; r < ROWS; ++r) c1[r] = 0.f;
        } else {
            beta = sqrtf(c11 * c11 + tailSq);
            if (c11 >= 0.f) beta = -beta;
            float den = c11 - beta;
#pragma unroll
            for (int r = 2; r < ROWS; ++r) c1[r] = c1[r] / den;
            tau1 = (beta - c11) / beta;
        }
        c1[1] = beta;
        if (tau1 != 0.f) {
            float t2 = 0.f;
#pragma unroll
            for (int r = 2; r < ROWS; ++r) t2 += c1[r] * c2[r];
            t2 += c2[1];
            c2[1] -= tau1 * t2;
#pragma unroll
            for (int r = 2; r < ROWS; ++r) c2[r] -= tau1 * c1[r] * t2;
        }
        if (nu2 != 0.f) {
            float temp = fabsf(c2[1]) / nu2;
            temp = (1.f + temp) * (1.f - temp);
            temp = temp < 0.f ? 0.f : temp;
            float ratio = nu2 / nd2;
            float temp2 = temp * (ratio * ratio);
            if (temp2 <= norm_downdate_threshold) {
                float s = 0.f;
#pragma unroll
                for (int r = 2; r < ROWS; ++r) s += c2[r] * c2[r];
                nd2 = sqrtf(s); nu2 = nd2;
            } else nu2 *= sqrtf(temp);
        }
    }
    // ---------------- k = 2
    {
        if (nonzero_pivots == 3 && nu2 * nu2 < threshold_helper * float(ROWS - 2)) nonzero_pivots = 2;
        float tailSq = 0.f;
#pragma unroll
        for (int r = 3; r < ROWS; ++r) tailSq += c2[r] * c2[r];
        float c22 = c2[2], beta;
        if (tailSq <= FLT_MIN) {
            tau2 = 0.f; beta = c22;
#pragma unroll
            for (int r = 3; r < ROWS; ++r) c2[r] = 0.f;
        } else {
            beta = sqrtf(c22 * c22 + tailSq);
            if (c22 >= 0.f) beta = -beta;
            float den = c22 - beta;
#pragma unroll
            for (int r = 3; r < ROWS; ++r) c2[r] = c2[r] / den;
            tau2 = (beta - c22) / beta;
        }
        c2[2] = beta;
    }
    // ---------------- solve: c = Q^T b (first nonzero_pivots reflectors), back-substitute, un-permute
    float y[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) y[r] = -1.f;
    if (nonzero_pivots >= 1 && tau0 != 0.f) {
        float t = 0.f;
#pragma unroll
        for (int r = 1; r < ROWS; ++r) t += c0[r] * y[r];
        t += y[0];
        y[0] -= tau0 * t;
#pragma unroll
        for (int r = 1; r < ROWS; ++r) y[r] -= tau0 * c0[r] * t;
    }
    if (nonzero_pivots >= 2 && tau1 != 0.f) {
        float t = 0.f;
#pragma unroll
        for (int r = 2; r < ROWS; ++r) t += c1[r] * y[r];
        t += y[1];
        y[1] -= tau1 * t;
#pragma unroll
        for (int r = 2; r < ROWS; ++r) y[r] -= tau1 * c1[r] * t;
    }
    if (nonzero_pivots >= 3 && tau2 != 0.f) {
        float t = 0.f;
#pragma unroll
        for (int r = 3; r < ROWS; ++r) t += c2[r] * y[r];
        t += y[2];
        y[2] -= tau2 * t;
#pragma unroll
        for (int r = 3; r < ROWS; ++r) y[r] -= tau2 * c2[r] * t;
    }
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;   // solution in pivoted order
    // column-oriented back substitution (Eigen triangular_solve_vector, Upper/ColMajor): divide, then axpy upwards
    if (nonzero_pivots == 3) {
        x2 = y[2] / c2[2];
        float s1 = y[1]; s1 -= x2 * c2[1];
        float s0 = y[0]; s0 -= x2 * c2[0];
        x1 = s1 / c1[1];
        s0 -= x1 * c1[0];
        x0 = s0 / c0[0];
    } else if (nonzero_pivots == 2) {
        x1 = y[1] / c1[1];
        float s0 = y[0]; s0 -= x1 * c1[0];
        x0 = s0 / c0[0];
    } else if (nonzero_pivots == 1) {
        x0 = y[0] / c0[0];
    }
    // (p0, p1, p2) is a permutation of (0, 1, 2): selects, so the three results stay in registers (no scratch round trip)
    nx = (p0 == 0) ? x0 : ((p1 == 0) ? x1 : x2);
    ny = (p0 == 1) ? x0 : ((p1 == 1) ? x1 : x2);
    nz = (p0 == 2) ? x0 : ((p1 == 2) ? x1 : x2);
}


// The pose the mapper starts frame k + 1 from, given frame k's result xc (lidar_mapper_keyframe.cpp:145-160): transformUpdate, pose_wmap_wodom = pose_wmap_curr *
// pose_wodom_curr.inverse(), then transformAssociateToMap, pose_wmap_curr = pose_wmap_wodom * pose_wodom_curr (the next frame's), with Pose::operator* / Pose::inverse
// as pose.cpp:99-113 write them (both construct through Pose(q, t), which normalises the quaternion; Quaterniond::inverse is conjugate / squaredNorm). One lane.
// Poses as [t, q(xyzw)].
__device__ inline void pose_ctor_qt(const q4 &q, const d3 &t, q4 &qo, d3 &to)
{
    const double n = sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    qo = q4{q.x / n, q.y / n, q.z / n, q.w / n};
    to = t;
}
__device__ inline void chain_start_pose(const double *xc, const double *wodom_prev, const double *wodom_cur, double *out)
{
    const q4 qc{xc[3], xc[4], xc[5], xc[6]};
    const d3 tc{xc[0], xc[1], xc[2]};
    // pose_wodom_curr.inverse()
    const q4 qp{wodom_prev[3], wodom_prev[4], wodom_prev[5], wodom_prev[6]};
    const double n2 = qp.x * qp.x + qp.y * qp.y + qp.z * qp.z + qp.w * qp.w;
    const q4 qinv = n2 > 0.0 ? q4{-qp.x / n2, -qp.y / n2, -qp.z / n2, qp.w / n2} : q4{0.0, 0.0, 0.0, 0.0};
    const d3 mt = qrot(qinv, d3{wodom_prev[0], wodom_prev[1], wodom_prev[2]});
    q4 qi; d3 ti;
    pose_ctor_qt(qinv, d3{-mt.x, -mt.y, -mt.z}, qi, ti);
    // pose_wmap_wodom = pose_wmap_curr * inverse
    const d3 r1 = qrot(qc, ti);
    q4 qw; d3 tw;
    pose_ctor_qt(qmul(qc, qi), d3{r1.x + tc.x, r1.y + tc.y, r1.z + tc.z}, qw, tw);
    // pose_wmap_curr = pose_wmap_wodom * pose_wodom_curr (the next frame's)
    const d3 r2 = qrot(qw, d3{wodom_cur[0], wodom_cur[1], wodom_cur[2]});
    q4 qn; d3 tn;
    pose_ctor_qt(qmul(qw, q4{wodom_cur[3], wodom_cur[4], wodom_cur[5], wodom_cur[6]}), d3{r2.x + tw.x, r2.y + tw.y, r2.z + tw.z}, qn, tn);
    out[0] = tn.x; out[1] = tn.y; out[2] = tn.z; out[3] = qn.x; out[4] = qn.y; out[5] = qn.z; out[6] = qn.w;
}

}  // namespace mlh
