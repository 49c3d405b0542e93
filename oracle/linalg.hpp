// TEST INFRASTRUCTURE ONLY -- CPU oracle for the M-LOAM scan-to-map hot path.
// Nothing under oracle/ may be linked, imported or executed by the product path
// (m-loam_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
// leg may use it, and only as the checker / reported CPU baseline.
//
// Pin status (round 2): the reference's OWN lines for extractCloud, match*PointFromMap, the factor classes, Plus and ImageSegmenter are
// compiled from /root/reference over a shim and the restatements under oracle/ are held against them (oracle/ref/build_ref.py,
// tests/test_oracle_ref_pin.py). That pins the reference's logic. It does NOT pin what follows:
// PARITY UNPINNED (library arithmetic): the reference (gogojjh/M-LOAM) delegates this arithmetic to
// Eigen 3.3.4 (ROS melodic / Ubuntu 18.04 `libeigen3-dev`, docker/Dockerfile:1-5),
// which is NOT vendored under /root/reference and is absent from this image.
// The routines below restate Eigen's published algorithms:
//   * SelfAdjointEigenSolver<Matrix3f>::compute  (Eigen/src/Eigenvalues/
//     SelfAdjointEigenSolver.h: scale to [-1,1], closed-form 3x3 Householder
//     tridiagonalisation (Tridiagonalization.h, 3x3 real specialisation), implicit
//     symmetric QR steps with Wilkinson shift, selection-sort of eigenvalues)
//     -- call sites feature_extract.hpp:427, 688
//   * ColPivHouseholderQR<MatrixXf>::compute/solve (Eigen/src/QR/ColPivHouseholderQR.h,
//     Householder.h: LAPACK xGEQPF-style norm down-dating)
//     -- call sites feature_extract.hpp:579, 823
//   * SelfAdjointEigenSolver<Matrix<double,6,6>> (eigenvalues ascending + orthonormal
//     eigenvectors; restated with cyclic Jacobi, which yields the same spectrum to
//     ~1e-15 relative) -- call site lidar_mapper_keyframe.cpp:1174
//   * LLT (Cholesky) for common::logDet (math.hpp:173-202)
// All f32 accumulations are sequential left-to-right, no FMA contraction
// (compile with -ffp-contract=off).
#pragma once
#include <cmath>
#include <cfloat>
#include <cstring>
#include <algorithm>

namespace orc {

// ---------------------------------------------------------------- f32 3x3 sym eigen
struct Eig3f {
    float val[3];      // ascending
    float vec[3][3];   // vec[r][c]: column c is the eigenvector of val[c]
    bool ok;
};

// JacobiRotation<float>::makeGivens, real case (Eigen/src/Jacobi/Jacobi.h)
static inline void make_givens_f(float p, float q, float &c, float &s)
{
    if (q == 0.f) { c = p < 0.f ? -1.f : 1.f; s = 0.f; }
    else if (p == 0.f) { c = 0.f; s = q < 0.f ? 1.f : -1.f; }
    else if (std::fabs(p) > std::fabs(q)) {
        float t = q / p;
        float u = std::sqrt(1.f + t * t);
        if (p < 0.f) u = -u;
        c = 1.f / u;
        s = -t * c;
    } else {
        float t = p / q;
        float u = std::sqrt(1.f + t * t);
        if (q < 0.f) u = -u;
        s = -1.f / u;
        c = -t * s;
    }
}

static inline float hypot_eigen_f(float x, float y)
{
    float ax = std::fabs(x), ay = std::fabs(y);
    float p, qp;
    if (ax > ay) { p = ax; qp = ay / p; } else { p = ay; qp = ax / p; }
    if (p == 0.f) return 0.f;
    return p * std::sqrt(1.f + qp * qp);
}

// internal::tridiagonal_qr_step (SelfAdjointEigenSolver.h)
static inline void tridiag_qr_step_f(float *diag, float *sub, int start, int end, float Q[3][3])
{
    float td = (diag[end - 1] - diag[end]) * 0.5f;
    float e = sub[end - 1];
    float mu = diag[end];
    if (td == 0.f) mu -= std::fabs(e);
    else {
        float e2 = e * e;
        float h = hypot_eigen_f(td, e);
        if (e2 == 0.f) mu -= (e / (td + (td > 0.f ? 1.f : -1.f))) * (e / h);
        else mu -= e2 / (td + (td > 0.f ? h : -h));
    }
    float x = diag[start] - mu;
    float z = sub[start];
    for (int k = start; k < end; ++k) {
        float c, s;
        make_givens_f(x, z, c, s);
        float sdk = s * diag[k] + c * sub[k];
        float dkp1 = s * sub[k] + c * diag[k + 1];
        diag[k] = c * (c * diag[k] - s * sub[k]) - s * (c * sub[k] - s * diag[k + 1]);
        diag[k + 1] = s * sdk + c * dkp1;
        sub[k] = c * sdk - s * dkp1;
        if (k > start) sub[k - 1] = c * sub[k - 1] - s * z;
        x = sub[k];
        if (k < end - 1) {
            z = -s * sub[k + 1];
            sub[k + 1] = c * sub[k + 1];
        }
        // Q = Q * G : columns k, k+1  (applyOnTheRight(k,k+1,rot))
        for (int i = 0; i < 3; ++i) {
            float xi = Q[i][k], yi = Q[i][k + 1];
            Q[i][k] = c * xi - s * yi;
            Q[i][k + 1] = s * xi + c * yi;
        }
    }
}

// A is symmetric; only the lower triangle is read (as Eigen does).
static inline Eig3f eig3_sym_f(const float A[3][3])
{
    Eig3f out;
    float m[3][3];
    // lower triangular view, scaled by max |coeff|
    float scale = 0.f;
    for (int c = 0; c < 3; ++c)
        for (int r = c; r < 3; ++r) scale = std::max(scale, std::fabs(A[r][c]));
    if (scale == 0.f) scale = 1.f;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m[r][c] = (r >= c) ? A[r][c] / scale : 0.f;

    float diag[3], sub[2];
    float Q[3][3];
    const float tol = FLT_MIN;
    diag[0] = m[0][0];
    float v1norm2 = m[2][0] * m[2][0];
    if (v1norm2 <= tol) {
        diag[1] = m[1][1];
        diag[2] = m[2][2];
        sub[0] = m[1][0];
        sub[1] = m[2][1];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Q[r][c] = (r == c) ? 1.f : 0.f;
    } else {
        float beta = std::sqrt(m[1][0] * m[1][0] + v1norm2);
        float invBeta = 1.f / beta;
        float m01 = m[1][0] * invBeta;
        float m02 = m[2][0] * invBeta;
        float q = 2.f * m01 * m[2][1] + m02 * (m[2][2] - m[1][1]);
        diag[1] = m[1][1] + m02 * q;
        diag[2] = m[2][2] - m02 * q;
        sub[0] = beta;
        sub[1] = m[2][1] - m01 * q;
        Q[0][0] = 1.f; Q[0][1] = 0.f; Q[0][2] = 0.f;
        Q[1][0] = 0.f; Q[1][1] = m01; Q[1][2] = m02;
        Q[2][0] = 0.f; Q[2][1] = m02; Q[2][2] = -m01;
    }

    // computeFromTridiagonal_impl (Eigen 3.3.4 form of the deflation test)
    const int n = 3, maxIterations = 30;
    int end = n - 1, start = 0, iter = 0;
    const float considerAsZero = FLT_MIN;
    const float precision = 2.f * FLT_EPSILON;
    while (end > 0) {
        for (int i = start; i < end; ++i)
            if (std::fabs(sub[i]) <= (std::fabs(diag[i]) + std::fabs(diag[i + 1])) * precision ||
                std::fabs(sub[i]) <= considerAsZero)
                sub[i] = 0.f;
        while (end > 0 && sub[end - 1] == 0.f) end--;
        if (end <= 0) break;
        iter++;
        if (iter > maxIterations * n) break;
        start = end - 1;
        while (start > 0 && sub[start - 1] != 0.f) start--;
        tridiag_qr_step_f(diag, sub, start, end, Q);
    }
    out.ok = (iter <= maxIterations * n);
    if (out.ok) {
        for (int i = 0; i < n - 1; ++i) {
            int k = 0;
            float mn = diag[i];
            for (int j = 1; j < n - i; ++j) if (diag[i + j] < mn) { mn = diag[i + j]; k = j; }
            if (k > 0) {
                std::swap(diag[i], diag[k + i]);
                for (int r = 0; r < 3; ++r) std::swap(Q[r][i], Q[r][k + i]);
            }
        }
    }
    for (int i = 0; i < 3; ++i) out.val[i] = diag[i] * scale;
    std::memcpy(out.vec, Q, sizeof(Q));
    return out;
}

// ---------------------------------------------------------------- f32 col-piv Householder QR solve (rows x 3)
// Solves min ||A x - b|| for A (rows x 3, row-major a[r*3+c]), rows <= 16.
static inline void colpiv_qr_solve_f(const float *a_in, const float *b_in, int rows, float x[3])
{
    const int cols = 3;
    const int size = std::min(rows, cols);
    float qr[16][3];
    float hco[3];
    int transp[3];
    float normsUpd[3], normsDir[3];
    for (int r = 0; r < rows; ++r) for (int c = 0; c < cols; ++c) qr[r][c] = a_in[r * 3 + c];
    for (int k = 0; k < cols; ++k) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += qr[r][k] * qr[r][k];
        normsDir[k] = std::sqrt(s);
        normsUpd[k] = normsDir[k];
    }
    float maxn = std::max(normsUpd[0], std::max(normsUpd[1], normsUpd[2]));
    float th = maxn * FLT_EPSILON;
    const float threshold_helper = (th * th) / float(rows);
    const float norm_downdate_threshold = std::sqrt(FLT_EPSILON);
    int nonzero_pivots = size;
    float maxpivot = 0.f;
    for (int k = 0; k < size; ++k) {
        int big = k;
        float bn = normsUpd[k];
        for (int j = k + 1; j < cols; ++j) if (normsUpd[j] > bn) { bn = normsUpd[j]; big = j; }
        float biggest_sq = bn * bn;
        if (nonzero_pivots == size && biggest_sq < threshold_helper * float(rows - k)) nonzero_pivots = k;
        transp[k] = big;
        if (k != big) {
            for (int r = 0; r < rows; ++r) std::swap(qr[r][k], qr[r][big]);
            std::swap(normsUpd[k], normsUpd[big]);
            std::swap(normsDir[k], normsDir[big]);
        }
        // makeHouseholderInPlace on qr[k..rows-1][k]
        float tailSq = 0.f;
        for (int r = k + 1; r < rows; ++r) tailSq += qr[r][k] * qr[r][k];
        float c0 = qr[k][k];
        float tau, beta;
        if (rows - k == 1 || tailSq <= FLT_MIN) {
            tau = 0.f; beta = c0;
            for (int r = k + 1; r < rows; ++r) qr[r][k] = 0.f;
        } else {
            beta = std::sqrt(c0 * c0 + tailSq);
            if (c0 >= 0.f) beta = -beta;
            float den = c0 - beta;
            for (int r = k + 1; r < rows; ++r) qr[r][k] = qr[r][k] / den;
            tau = (beta - c0) / beta;
        }
        hco[k] = tau;
        qr[k][k] = beta;
        if (std::fabs(beta) > maxpivot) maxpivot = std::fabs(beta);
        // applyHouseholderOnTheLeft to bottomRightCorner(rows-k, cols-k-1)
        if (cols - k - 1 > 0) {
            if (rows - k == 1) {
                for (int j = k + 1; j < cols; ++j) qr[k][j] *= (1.f - tau);
            } else if (tau != 0.f) {
                for (int j = k + 1; j < cols; ++j) {
                    float tmp = 0.f;
                    for (int r = k + 1; r < rows; ++r) tmp += qr[r][k] * qr[r][j];
                    tmp += qr[k][j];
                    qr[k][j] -= tau * tmp;
                    for (int r = k + 1; r < rows; ++r) qr[r][j] -= tau * qr[r][k] * tmp;
                }
            }
        }
        for (int j = k + 1; j < cols; ++j) {
            if (normsUpd[j] != 0.f) {
                float temp = std::fabs(qr[k][j]) / normsUpd[j];
                temp = (1.f + temp) * (1.f - temp);
                temp = temp < 0.f ? 0.f : temp;
                float ratio = normsUpd[j] / normsDir[j];
                float temp2 = temp * (ratio * ratio);
                if (temp2 <= norm_downdate_threshold) {
                    float s = 0.f;
                    for (int r = k + 1; r < rows; ++r) s += qr[r][j] * qr[r][j];
                    normsDir[j] = std::sqrt(s);
                    normsUpd[j] = normsDir[j];
                } else {
                    normsUpd[j] *= std::sqrt(temp);
                }
            }
        }
    }
    // permutation indices: P = T_0 T_1 ... (setIdentity then applyTranspositionOnTheRight)
    int perm[3] = {0, 1, 2};
    for (int k = 0; k < size; ++k) std::swap(perm[k], perm[transp[k]]);

    // solve
    x[0] = x[1] = x[2] = 0.f;
    if (nonzero_pivots == 0) return;
    float c[16];
    for (int r = 0; r < rows; ++r) c[r] = b_in[r];
    // c = H_{np-1} ... H_0 c   (Q^T c, applied in order k = 0..np-1)
    for (int k = 0; k < nonzero_pivots; ++k) {
        float tau = hco[k];
        if (rows - k == 1) { c[k] *= (1.f - tau); }
        else if (tau != 0.f) {
            float tmp = 0.f;
            for (int r = k + 1; r < rows; ++r) tmp += qr[r][k] * c[r];
            tmp += c[k];
            c[k] -= tau * tmp;
            for (int r = k + 1; r < rows; ++r) c[r] -= tau * qr[r][k] * tmp;
        }
    }
    // back substitution on the leading np x np upper triangle, column-oriented as Eigen's
    // triangular_solve_vector<..., OnTheLeft, Upper, ColMajor>: x_i = c_i / U_ii, then c_0..i-1 -= x_i * U_0..i-1,i
    for (int i = nonzero_pivots - 1; i >= 0; --i) {
        c[i] = c[i] / qr[i][i];
        for (int r = 0; r < i; ++r) c[r] -= c[i] * qr[r][i];
    }
    for (int i = 0; i < nonzero_pivots; ++i) x[perm[i]] = c[i];
    for (int i = nonzero_pivots; i < cols; ++i) x[perm[i]] = 0.f;
}

// ---------------------------------------------------------------- f64 n x n helpers (row-major, n <= 32)
// cyclic Jacobi eigen-decomposition of a symmetric matrix; eigenvalues ascending, V columns.
static inline void jacobi_eig_sym_d(const double *A, int n, double *eval, double *V)
{
    double a[32 * 32];
    for (int i = 0; i < n * n; ++i) a[i] = A[i];
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < n; ++i) { dg += a[i * n + i] * a[i * n + i]; for (int j = i + 1; j < n; ++j) off += a[i * n + j] * a[i * n + j]; }
        if (off <= 1e-32 * dg || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                double apq = a[p * n + q];
                if (apq == 0.0) continue;
                double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    double akp = a[k * n + p], akq = a[k * n + q];
                    a[k * n + p] = c * akp - s * akq;
                    a[k * n + q] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    double apk = a[p * n + k], aqk = a[q * n + k];
                    a[p * n + k] = c * apk - s * aqk;
                    a[q * n + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    double vkp = V[k * n + p], vkq = V[k * n + q];
                    V[k * n + p] = c * vkp - s * vkq;
                    V[k * n + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < n; ++i) eval[i] = a[i * n + i];
    // sort ascending (selection sort like Eigen)
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        for (int j = i + 1; j < n; ++j) if (eval[j] < eval[k]) k = j;
        if (k != i) {
            std::swap(eval[i], eval[k]);
            for (int r = 0; r < n; ++r) std::swap(V[r * n + i], V[r * n + k]);
        }
    }
}

// Cholesky A = L L^T (lower, row-major). returns false if not positive definite.
static inline bool cholesky_d(const double *A, int n, double *L)
{
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        if (!(s > 0.0)) return false;
        double ljj = std::sqrt(s);
        L[j * n + j] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / ljj;
        }
    }
    return true;
}

// common::logDet(M, use_cholesky=true)  (mloam_common/libs/include/common/algos/math.hpp:173-187)
static inline double logdet_chol_d(const double *A, int n)
{
    double L[32 * 32];
    if (!cholesky_d(A, n, L)) return NAN;
    double ld = 0.0;
    for (int i = 0; i < n; ++i) ld += std::log(L[i * n + i]);
    return 2.0 * ld;
}

// solve A x = b for SPD A via Cholesky. returns false on failure.
static inline bool chol_solve_d(const double *A, const double *b, int n, double *x)
{
    double L[32 * 32], y[32];
    if (!cholesky_d(A, n, L)) return false;
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * y[k];
        y[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
    return true;
}

// general inverse by Gauss-Jordan with partial pivoting (row-major)
static inline bool inverse_d(const double *A, int n, double *Ainv)
{
    double a[32 * 64];
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) { a[i * 2 * n + j] = A[i * n + j]; a[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0; }
    }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a[r * 2 * n + c]) > std::fabs(a[piv * 2 * n + c])) piv = r;
        if (a[piv * 2 * n + c] == 0.0) return false;
        if (piv != c) for (int j = 0; j < 2 * n; ++j) std::swap(a[c * 2 * n + j], a[piv * 2 * n + j]);
        double d = a[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) a[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) if (r != c) {
            double f = a[r * 2 * n + c];
            if (f != 0.0) for (int j = 0; j < 2 * n; ++j) a[r * 2 * n + j] -= f * a[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ainv[i * n + j] = a[i * 2 * n + n + j];
    return true;
}

}  // namespace orc
