// TEST INFRASTRUCTURE ONLY. A few dozen Eigen 3.3 operations -- exactly the ones the reference's factor classes use -- so that those
// classes can be compiled FROM THE REFERENCE'S OWN SOURCE LINES (oracle/ref/build_ref.py) in an image that has no Eigen. Everything is
// evaluated eagerly, products as left-to-right sums over k (what Eigen's coefficient-based product of small fixed matrices does);
// q * v and toRotationMatrix follow Eigen's QuaternionBase formulas. This pins the reference's FORMULAS (which terms, which signs, which
// block goes where); the arithmetic inside Eigen itself stays a restatement.
#pragma once
#include <cmath>
#include <limits>
#include <cstddef>
#include <type_traits>
#include <ostream>
#include <vector>

namespace Eigen {
enum { ColMajor = 0, RowMajor = 1, Dynamic = -1 };

template <typename Derived> struct MatrixBase {
    const Derived &derived() const { return *static_cast<const Derived *>(this); }
    auto operator()(int i) const { return derived().d[i]; }
};

template <typename T, int R, int C> struct Mat;
template <typename T, int R, int C, int Opt = 0> using Matrix = Mat<T, R, C>;   // storage options accepted and ignored

template <typename T, int R, int C, int N> struct ColsRef {      // leftCols<N>() / rightCols<N>() of a row-major-irrelevant small matrix
    T *base; int col0, ld;   // element (i, j) at base[i * ld + col0 + j]   (storage below is row-major)
    ColsRef &operator=(const Mat<T, R, N> &m);
    void setZero() { for (int i = 0; i < R; ++i) for (int j = 0; j < N; ++j) base[i * ld + col0 + j] = T(0); }
};

// block<BR, BC>(i, j) / topLeftCorner / bottomRightCorner of a fixed-size matrix: assignable view, converts to a matrix
template <typename T, int BR, int BC> struct BlockRef {
    T *base; int ld;                                             // element (i, j) at base[i * ld + j]
    BlockRef &operator=(const Mat<T, BR, BC> &m);
    operator Mat<T, BR, BC>() const;
    Mat<T, BC, BR> transpose() const;
};

// a run of coefficients inside a vector -- v.head(n) / v.tail(n) with a run-time n, as VoxelGridCovarianceMLOAM::applyFilter writes them
// (voxel_grid_covariance_mloam_impl.hpp:312-336, 377, 404): `seg += w * other_seg`, `seg /= s`, `seg = fixed vector`
template <typename T> struct SmallVec { T v[16]; int n; };
template <typename T> struct SegRef {
    T *p; int n;
    SegRef &operator+=(const SmallVec<T> &o) { for (int i = 0; i < n; ++i) p[i] += o.v[i]; return *this; }
    SegRef &operator+=(const SegRef &o) { for (int i = 0; i < n; ++i) p[i] += o.p[i]; return *this; }
    SegRef &operator/=(T s) { for (int i = 0; i < n; ++i) p[i] /= s; return *this; }
    template <int R> SegRef &operator=(const Mat<T, R, 1> &m);
};
template <typename S, typename T, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
SmallVec<T> operator*(S s, const SegRef<T> &r) { SmallVec<T> o; o.n = r.n; for (int i = 0; i < r.n; ++i) o.v[i] = T(s) * r.p[i]; return o; }

template <typename T, int R, int C> struct CommaInit {
    Mat<T, R, C> &m; int k;
    CommaInit &operator,(T v) { m.d[k++] = v; return *this; }
};

template <typename T, int R, int C> struct Mat : MatrixBase<Mat<T, R, C>> {
    typedef T Scalar;
    typedef Mat Matrix;
    T d[R * C];                                                   // row-major storage
    Mat() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
    Mat(T a, T b, T c) { static_assert(R * C == 3, "3-vector"); d[0] = a; d[1] = b; d[2] = c; }
    Mat(T a, T b, T c, T e) { static_assert(R * C == 4, "4-vector"); d[0] = a; d[1] = b; d[2] = c; d[3] = e; }
    template <typename D> Mat(const MatrixBase<D> &o) { for (int i = 0; i < R * C; ++i) d[i] = o.derived().d[i]; }
    T &operator()(int i) { return d[i]; }
    const T &operator()(int i) const { return d[i]; }
    T &operator()(int i, int j) { return d[i * C + j]; }
    const T &operator()(int i, int j) const { return d[i * C + j]; }
    T &operator[](int i) { return d[i]; }
    const T &operator[](int i) const { return d[i]; }
    T x() const { return d[0]; } T y() const { return d[1]; } T z() const { return d[2]; }
    int rows() const { return R; } int cols() const { return C; }
    static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
    static Matrix Zero() { return Matrix(); }
    static Matrix Ones() { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = T(1); return m; }
    SegRef<T> head(int n) { return SegRef<T>{d, n}; }
    SegRef<T> tail(int n) { return SegRef<T>{d + R * C - n, n}; }
    Matrix &operator+=(const SmallVec<T> &o) { for (int i = 0; i < R * C; ++i) d[i] += o.v[i]; return *this; }
    void setZero() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
    template <typename U> Mat<U, R, C> cast() const { Mat<U, R, C> m; for (int i = 0; i < R * C; ++i) m.d[i] = U(d[i]); return m; }
    void setIdentity() { setZero(); for (int i = 0; i < (R < C ? R : C); ++i) (*this)(i, i) = T(1); }
    const Mat &real() const { return *this; }
    BlockRef<T, R, 1> col(int j) { return BlockRef<T, R, 1>{d + j, C}; }
    // general inverse (Gauss-Jordan with partial pivoting): library arithmetic, used for mat_V_f.transpose().inverse() only
    Mat inverse() const
    {
        static_assert(R == C, "square");
        T a[R][2 * R];
        for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) { a[i][j] = (*this)(i, j); a[i][R + j] = i == j ? T(1) : T(0); }
        for (int c = 0; c < R; ++c) {
            int piv = c;
            for (int r = c + 1; r < R; ++r) if (std::abs(a[r][c]) > std::abs(a[piv][c])) piv = r;
            if (piv != c) for (int j = 0; j < 2 * R; ++j) std::swap(a[c][j], a[piv][j]);
            const T dinv = T(1) / a[c][c];
            for (int j = 0; j < 2 * R; ++j) a[c][j] *= dinv;
            for (int r = 0; r < R; ++r) if (r != c) { const T f = a[r][c]; if (f != T(0)) for (int j = 0; j < 2 * R; ++j) a[r][j] -= f * a[c][j]; }
        }
        Mat m;
        for (int i = 0; i < R; ++i) for (int j = 0; j < R; ++j) m(i, j) = a[i][R + j];
        return m;
    }
    T trace() const { T s = d[0]; for (int i = 1; i < R; ++i) s += (*this)(i, i); return s; }
    T dot(const Matrix &o) const { T s = d[0] * o.d[0]; for (int i = 1; i < R * C; ++i) s += d[i] * o.d[i]; return s; }
    T squaredNorm() const { return dot(*this); }
    T norm() const { return std::sqrt(squaredNorm()); }
    void normalize() { *this = normalized(); }
    Mat<T, R, 1> col(int j) const { Mat<T, R, 1> m; for (int i = 0; i < R; ++i) m.d[i] = (*this)(i, j); return m; }
    Matrix normalized() const { T z2 = squaredNorm(); if (z2 > T(0)) { Matrix m = *this; T n = std::sqrt(z2); for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / n; return m; } return *this; }
    Matrix cross(const Matrix &o) const { static_assert(R * C == 3, "cross"); return Matrix(d[1] * o.d[2] - d[2] * o.d[1], d[2] * o.d[0] - d[0] * o.d[2], d[0] * o.d[1] - d[1] * o.d[0]); }
    Mat<T, C, R> transpose() const { Mat<T, C, R> m; for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) m(j, i) = (*this)(i, j); return m; }
    Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = -d[i]; return m; }
    Matrix operator+(const Matrix &o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] + o.d[i]; return m; }
    Matrix operator-(const Matrix &o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] - o.d[i]; return m; }
    Matrix operator*(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] * s; return m; }
    Matrix operator/(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / s; return m; }
    Matrix &operator+=(const Matrix &o) { for (int i = 0; i < R * C; ++i) d[i] += o.d[i]; return *this; }
    Matrix &operator/=(T s) { for (int i = 0; i < R * C; ++i) d[i] /= s; return *this; }
    template <int K> Mat<T, R, K> operator*(const Mat<T, C, K> &o) const
    {
        Mat<T, R, K> m;
        for (int i = 0; i < R; ++i) for (int j = 0; j < K; ++j) { T s = (*this)(i, 0) * o(0, j); for (int k = 1; k < C; ++k) s += (*this)(i, k) * o(k, j); m(i, j) = s; }
        return m;
    }
    CommaInit<T, R, C> operator<<(T v) { d[0] = v; return CommaInit<T, R, C>{*this, 1}; }
    template <int BR, int BC> BlockRef<T, BR, BC> block(int i, int j) { return BlockRef<T, BR, BC>{d + i * C + j, C}; }
    template <int BR, int BC> Mat<T, BR, BC> block(int i, int j) const { Mat<T, BR, BC> m; for (int r = 0; r < BR; ++r) for (int c = 0; c < BC; ++c) m(r, c) = (*this)(i + r, j + c); return m; }
    template <int BR, int BC> BlockRef<T, BR, BC> topLeftCorner() { return block<BR, BC>(0, 0); }
    template <int BR, int BC> BlockRef<T, BR, BC> bottomRightCorner() { return block<BR, BC>(R - BR, C - BC); }
    template <int BR, int BC> BlockRef<T, BR, BC> topRightCorner() { return block<BR, BC>(0, C - BC); }
    template <int BR, int BC> BlockRef<T, BR, BC> bottomLeftCorner() { return block<BR, BC>(R - BR, 0); }
    template <int BR, int BC> Mat<T, BR, BC> topLeftCorner() const { return block<BR, BC>(0, 0); }
    template <int BR, int BC> Mat<T, BR, BC> topRightCorner() const { return block<BR, BC>(0, C - BC); }
    template <int BR, int BC> Mat<T, BR, BC> bottomLeftCorner() const { return block<BR, BC>(R - BR, 0); }
    template <int BR, int BC> Mat<T, BR, BC> bottomRightCorner() const { return block<BR, BC>(R - BR, C - BC); }
    template <int N> ColsRef<T, R, C, N> leftCols() { return ColsRef<T, R, C, N>{d, 0, C}; }
    template <int N> ColsRef<T, R, C, N> rightCols() { return ColsRef<T, R, C, N>{d, C - N, C}; }
    template <int N> Mat<T, N, 1> head() const { Mat<T, N, 1> m; for (int i = 0; i < N; ++i) m.d[i] = d[i]; return m; }
    template <int N> Mat<T, N, 1> tail() const { Mat<T, N, 1> m; for (int i = 0; i < N; ++i) m.d[i] = d[R * C - N + i]; return m; }
};
template <typename S, typename T, int R, int C, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
Mat<T, R, C> operator*(S s, const Mat<T, R, C> &m) { return m * T(s); }
template <typename T> template <int R> SegRef<T> &SegRef<T>::operator=(const Mat<T, R, 1> &m) { for (int i = 0; i < n; ++i) p[i] = m.d[i]; return *this; }
template <typename T, int R, int C, int N> ColsRef<T, R, C, N> &ColsRef<T, R, C, N>::operator=(const Mat<T, R, N> &m)
{
    for (int i = 0; i < R; ++i) for (int j = 0; j < N; ++j) base[i * ld + col0 + j] = m(i, j);
    return *this;
}

template <typename T, int BR, int BC> BlockRef<T, BR, BC> &BlockRef<T, BR, BC>::operator=(const Mat<T, BR, BC> &m)
{
    for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) base[i * ld + j] = m(i, j);
    return *this;
}
template <typename T, int BR, int BC> BlockRef<T, BR, BC>::operator Mat<T, BR, BC>() const
{
    Mat<T, BR, BC> m;
    for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) m(i, j) = base[i * ld + j];
    return m;
}


// ---- a dynamic-size matrix for the few places the reference holds one (PointPlaneFeature::jaco_, common::logDet's LLT)
template <typename T> struct Mat<T, Dynamic, Dynamic> : MatrixBase<Mat<T, Dynamic, Dynamic>> {
    typedef T Scalar;
    int r = 0, c = 0;
    std::vector<T> d;                                             // row-major
    Mat() {}
    Mat(int rows_, int cols_) : r(rows_), c(cols_), d(size_t(rows_) * cols_, T(0)) {}
    template <int R, int C> Mat(const Mat<T, R, C> &m) : r(R), c(C), d(m.d, m.d + R * C) {}
    template <int R, int C> Mat &operator=(const Mat<T, R, C> &m) { r = R; c = C; d.assign(m.d, m.d + R * C); return *this; }
    int rows() const { return r; } int cols() const { return c; }
    void resize(int rows_, int cols_) { r = rows_; c = cols_; d.assign(size_t(rows_) * cols_, T(0)); }      // (Eigen leaves the coefficients uninitialised)
    T *data() { return d.data(); }
    const T *data() const { return d.data(); }
    T &operator()(int i, int j) { return d[size_t(i) * c + j]; }
    const T &operator()(int i, int j) const { return d[size_t(i) * c + j]; }
    Mat transpose() const { Mat m(c, r); for (int i = 0; i < r; ++i) for (int j = 0; j < c; ++j) m(j, i) = (*this)(i, j); return m; }
    Mat block(int r0, int c0, int nr, int nc) const { Mat m(nr, nc); for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) m(i, j) = (*this)(r0 + i, c0 + j); return m; }
    Mat operator*(const Mat &o) const
    {
        Mat m(r, o.c);
        for (int i = 0; i < r; ++i) for (int j = 0; j < o.c; ++j) { T s = (*this)(i, 0) * o(0, j); for (int k = 1; k < c; ++k) s += (*this)(i, k) * o(k, j); m(i, j) = s; }
        return m;
    }
};
typedef Mat<double, Dynamic, Dynamic> MatrixXd;
// Eigen::SparseMatrix<T, RowMajor> as Estimator::evalDegenracy uses it (resize, coeffRef, transpose, product into a dense matrix): a dense
// matrix underneath. J^T J sums over the rows in ascending order either way; the zeros a dense product adds change no sum.
template <typename T, int Options = 0> struct SparseMatrix {
    Mat<T, Dynamic, Dynamic> m;
    void resize(int r, int c) { m = Mat<T, Dynamic, Dynamic>(r, c); }
    T &coeffRef(int i, int j) { return m(i, j); }
    SparseMatrix transpose() const { SparseMatrix t; t.m = m.transpose(); return t; }
    Mat<T, Dynamic, Dynamic> operator*(const SparseMatrix &o) const { return m * o.m; }
};
template <typename T, int R, int C> std::ostream &operator<<(std::ostream &o, const Mat<T, R, C> &m)
{
    for (int i = 0; i < m.rows(); ++i) { for (int j = 0; j < m.cols(); ++j) o << (j ? " " : "") << m(i, j); if (i + 1 < m.rows()) o << "\n"; }
    return o;
}
template <typename T, int R, int C> Mat<T, R, C> operator+(const Mat<T, R, C> &a, const Mat<T, Dynamic, Dynamic> &b)
{
    Mat<T, R, C> m;
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) m(i, j) = a(i, j) + b(i, j);
    return m;
}
template <typename T, int R, int C> Mat<T, R, C> &operator+=(Mat<T, R, C> &a, const Mat<T, Dynamic, Dynamic> &b)
{
    for (int i = 0; i < R; ++i) for (int j = 0; j < C; ++j) a(i, j) += b(i, j);
    return a;
}
// Eigen::LLT of a symmetric positive definite matrix (standard left-looking Cholesky, sums in ascending k): what common::logDet reads is
// matrixL()(i, i). Library arithmetic: restated, like everything in this header.
template <typename M> struct LLT;
template <typename T> struct LLT<Mat<T, Dynamic, Dynamic>> {
    Mat<T, Dynamic, Dynamic> L;
    template <int R, int C> LLT(const Mat<T, R, C> &A) : L(R, C)
    {
        for (int j = 0; j < R; ++j) {
            T s = A(j, j);
            for (int k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
            const T ljj = std::sqrt(s);
            L(j, j) = ljj;
            for (int i = j + 1; i < R; ++i) { T t = A(i, j); for (int k = 0; k < j; ++k) t -= L(i, k) * L(j, k); L(i, j) = t / ljj; }
        }
    }
    Mat<T, Dynamic, Dynamic> &matrixL() { return L; }
};
// only instantiated, never executed (common::logDet is called with use_cholesky = true on this path)
template <typename M> struct PartialPivLU;
template <typename T> struct PartialPivLU<Mat<T, Dynamic, Dynamic>> {
    Mat<T, Dynamic, Dynamic> LU;
    struct Perm { T determinant() const { return T(1); } };
    template <int R, int C> PartialPivLU(const Mat<T, R, C> &A) : LU(A) {}
    Mat<T, Dynamic, Dynamic> &matrixLU() { return LU; }
    Perm permutationP() const { return Perm(); }
};

template <typename T, int BR, int BC> Mat<T, BC, BR> BlockRef<T, BR, BC>::transpose() const { return Mat<T, BR, BC>(*this).transpose(); }

typedef Matrix<double, 4, 4> Matrix4d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 1> Vector4f;
typedef Matrix<int, 4, 1> Vector4i;
struct VectorXf {                                                  // applyFilter's `centroid` / `temporary` (one float per point field)
    std::vector<float> v;
    static VectorXf Zero(int n) { VectorXf x; x.v.assign(size_t(n), 0.f); return x; }
    void setZero() { for (float &f : v) f = 0.f; }
    float &operator[](int i) { return v[size_t(i)]; }
    const float &operator[](int i) const { return v[size_t(i)]; }
    SegRef<float> head(int n) { return SegRef<float>{v.data(), n}; }
    SegRef<float> tail(int n) { return SegRef<float>{v.data() + v.size() - size_t(n), n}; }
    int size() const { return int(v.size()); }
};
typedef Matrix<float, 3, 3> Matrix3f;

struct VectorXd {                                                  // the edge factor keeps its six coefficients in one
    typedef double Scalar;
    std::vector<double> v;
    VectorXd() {}
    explicit VectorXd(int n) : v(size_t(n), 0.0) {}
    template <int R> VectorXd(const Mat<double, R, 1> &m) : v(m.d, m.d + R) {}
    template <int R> VectorXd &operator=(const Mat<double, R, 1> &m) { v.assign(m.d, m.d + R); return *this; }
    template <int R> operator Mat<double, R, 1>() const { Mat<double, R, 1> m; for (int i = 0; i < R; ++i) m.d[i] = v[size_t(i)]; return m; }   // Eigen: dynamic -> fixed, sizes must agree
    double &operator()(int i) { return v[size_t(i)]; }
    const double &operator()(int i) const { return v[size_t(i)]; }
    double &operator[](int i) { return v[size_t(i)]; }
    const double &operator[](int i) const { return v[size_t(i)]; }
    void resize(int n) { v.assign(size_t(n), 0.0); }
    double *data() { return v.data(); }
    const double *data() const { return v.data(); }
    int size() const { return int(v.size()); }
    const VectorXd &transpose() const { return *this; }          // only ever printed
};
inline std::ostream &operator<<(std::ostream &o, const VectorXd &x) { for (size_t i = 0; i < x.v.size(); ++i) o << (i ? " " : "") << x.v[i]; return o; }

template <typename T> struct Quaternion {
    typedef T Scalar;
    T x_, y_, z_, w_;
    Quaternion() : x_(0), y_(0), z_(0), w_(1) {}
    Quaternion(T w, T x, T y, T z) : x_(x), y_(y), z_(z), w_(w) {}
    // Quaternion(rotation matrix): Eigen 3.3 Geometry/Quaternion.h, quaternionbase_assign_impl<Other, 3, 3> (library arithmetic, restated)
    explicit Quaternion(const Matrix<T, 3, 3> &m)
    {
        T t = m(0, 0) + m(1, 1) + m(2, 2);
        T q[3];
        if (t > T(0)) {
            t = std::sqrt(t + T(1));
            w_ = T(0.5) * t;
            t = T(0.5) / t;
            x_ = (m(2, 1) - m(1, 2)) * t; y_ = (m(0, 2) - m(2, 0)) * t; z_ = (m(1, 0) - m(0, 1)) * t;
        } else {
            int i = 0;
            if (m(1, 1) > m(0, 0)) i = 1;
            if (m(2, 2) > m(i, i)) i = 2;
            const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + T(1));
            q[i] = T(0.5) * t;
            t = T(0.5) / t;
            w_ = (m(k, j) - m(j, k)) * t;
            q[j] = (m(j, i) + m(i, j)) * t;
            q[k] = (m(k, i) + m(i, k)) * t;
            x_ = q[0]; y_ = q[1]; z_ = q[2];
        }
    }
    T &x() { return x_; } T &y() { return y_; } T &z() { return z_; } T &w() { return w_; }
    T x() const { return x_; } T y() const { return y_; } T z() const { return z_; } T w() const { return w_; }
    // QuaternionBase::_transformVector: v + w * (2 q x v) + q x (2 q x v)
    Matrix<T, 3, 1> operator*(const Matrix<T, 3, 1> &v) const
    {
        const Matrix<T, 3, 1> qv(x_, y_, z_);
        Matrix<T, 3, 1> uv = qv.cross(v);
        uv += uv;
        const Matrix<T, 3, 1> c2 = qv.cross(uv);
        return Matrix<T, 3, 1>(v.d[0] + w_ * uv.d[0] + c2.d[0], v.d[1] + w_ * uv.d[1] + c2.d[1], v.d[2] + w_ * uv.d[2] + c2.d[2]);
    }
    Quaternion operator*(const Quaternion &b) const
    {
        return Quaternion(w_ * b.w_ - x_ * b.x_ - y_ * b.y_ - z_ * b.z_, w_ * b.x_ + x_ * b.w_ + y_ * b.z_ - z_ * b.y_,
                          w_ * b.y_ + y_ * b.w_ + z_ * b.x_ - x_ * b.z_, w_ * b.z_ + z_ * b.w_ + x_ * b.y_ - y_ * b.x_);
    }
    Quaternion conjugate() const { return Quaternion(w_, -x_, -y_, -z_); }
    static Quaternion Identity() { return Quaternion(T(1), T(0), T(0), T(0)); }
    // QuaternionBase::inverse: conjugate / squaredNorm (zero quaternion -> zero)
    Quaternion inverse() const { const T n2 = x_ * x_ + y_ * y_ + z_ * z_ + w_ * w_; if (n2 > T(0)) return Quaternion(w_ / n2, -x_ / n2, -y_ / n2, -z_ / n2); return Quaternion(T(0), T(0), T(0), T(0)); }
    T dot(const Quaternion &o) const { return x_ * o.x_ + y_ * o.y_ + z_ * o.z_ + w_ * o.w_; }
    // QuaternionBase::angularDistance (Eigen 3.3 Geometry/Quaternion.h): d = (*this * other.conjugate()); 2 * atan2(d.vec().norm(), |d.w()|)  (library arithmetic, restated)
    T angularDistance(const Quaternion &other) const
    {
        const Quaternion d = (*this) * other.conjugate();
        return T(2) * std::atan2(std::sqrt(d.x_ * d.x_ + d.y_ * d.y_ + d.z_ * d.z_), std::abs(d.w_));
    }
    // QuaternionBase::slerp (Eigen 3.3 Geometry/Quaternion.h): library arithmetic, restated
    Quaternion slerp(T t, const Quaternion &other) const
    {
        const T one = T(1) - std::numeric_limits<T>::epsilon();
        const T d = dot(other), absD = std::abs(d);
        T scale0, scale1;
        if (absD >= one) { scale0 = T(1) - t; scale1 = t; }
        else { const T theta = std::acos(absD), sinTheta = std::sin(theta); scale0 = std::sin((T(1) - t) * theta) / sinTheta; scale1 = std::sin(t * theta) / sinTheta; }
        if (d < T(0)) scale1 = -scale1;
        return Quaternion(scale0 * w_ + scale1 * other.w_, scale0 * x_ + scale1 * other.x_, scale0 * y_ + scale1 * other.y_, scale0 * z_ + scale1 * other.z_);
    }
    void normalize() { *this = normalized(); }
    Quaternion normalized() const { const T n = std::sqrt(x_ * x_ + y_ * y_ + z_ * z_ + w_ * w_); return Quaternion(w_ / n, x_ / n, y_ / n, z_ / n); }
    Matrix<T, 3, 3> toRotationMatrix() const
    {
        const T tx = T(2) * x_, ty = T(2) * y_, tz = T(2) * z_;
        const T twx = tx * w_, twy = ty * w_, twz = tz * w_, txx = tx * x_, txy = ty * x_, txz = tz * x_, tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
        Matrix<T, 3, 3> R;
        R(0, 0) = T(1) - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
        R(1, 0) = txy + twz; R(1, 1) = T(1) - (txx + tzz); R(1, 2) = tyz - twx;
        R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = T(1) - (txx + tyy);
        return R;
    }
};
typedef Quaternion<double> Quaterniond;

// Map<...> over a caller's array: matrices (row-major) and quaternions (coefficients x, y, z, w in memory, as Eigen stores them)
template <typename M> struct Map;
template <typename T> struct RowsRef {                           // topRows<N>() / bottomRows<N>() of a mapped row-major matrix (PoseLocalParameterization::ComputeJacobian)
    T *p; int rows, cols;
    void setZero() { for (int i = 0; i < rows * cols; ++i) p[i] = T(0); }
    void setIdentity() { setZero(); for (int i = 0; i < (rows < cols ? rows : cols); ++i) p[i * cols + i] = T(1); }
};
template <typename T, int R, int C> struct Map<Mat<T, R, C>> {
    T *p;
    explicit Map(T *q) : p(q) {}
    void setZero() { for (int i = 0; i < R * C; ++i) p[i] = T(0); }
    template <int N> RowsRef<T> topRows() { return RowsRef<T>{p, N, C}; }
    template <int N> RowsRef<T> bottomRows() { return RowsRef<T>{p + (R - N) * C, N, C}; }
    template <int N> ColsRef<T, R, C, N> leftCols() { return ColsRef<T, R, C, N>{p, 0, C}; }
    template <int N> ColsRef<T, R, C, N> rightCols() { return ColsRef<T, R, C, N>{p, C - N, C}; }
    Map &operator=(const Mat<T, R, C> &m) { for (int i = 0; i < R * C; ++i) p[i] = m.d[i]; return *this; }
    template <int BR, int BC> Mat<T, BR, BC> topLeftCorner() const { Mat<T, BR, BC> m; for (int i = 0; i < BR; ++i) for (int j = 0; j < BC; ++j) m(i, j) = p[i * C + j]; return m; }
};
template <typename T, int R, int C> struct Map<const Mat<T, R, C>> {
    const T *p;
    explicit Map(const T *q) : p(q) {}
    operator Mat<T, R, C>() const { Mat<T, R, C> m; for (int i = 0; i < R * C; ++i) m.d[i] = p[i]; return m; }
    Mat<T, R, C> operator+(const Mat<T, R, C> &o) const { return Mat<T, R, C>(*this) + o; }
};
template <typename T, int R, int C, int K> Mat<T, R, 1> operator*(const Mat<T, R, C> &a, const Map<const Mat<T, K, 1>> &b) { return a * Mat<T, K, 1>(b); }
template <typename T> struct Map<Quaternion<T>> {
    T *p;
    explicit Map(T *q) : p(q) {}
    Map &operator=(const Quaternion<T> &q) { p[0] = q.x_; p[1] = q.y_; p[2] = q.z_; p[3] = q.w_; return *this; }
};
template <typename T> struct Map<const Quaternion<T>> {
    const T *p;
    explicit Map(const T *q) : p(q) {}
    operator Quaternion<T>() const { return Quaternion<T>(p[3], p[0], p[1], p[2]); }
    Quaternion<T> operator*(const Quaternion<T> &o) const { return Quaternion<T>(*this) * o; }
};
}  // namespace Eigen

// ---- the two Eigen decompositions the match functions call: thin wrappers over the oracle's restatements (oracle/linalg.hpp)
#include "../linalg.hpp"
namespace Eigen {
template <typename M> struct SelfAdjointEigenSolver;
template <> struct SelfAdjointEigenSolver<Matrix3f> {
    orc::Eig3f e;
    explicit SelfAdjointEigenSolver(const Matrix3f &A) { float a[3][3]; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) a[i][j] = A(i, j); e = orc::eig3_sym_f(a); }
    Vector3f eigenvalues() const { return Vector3f(e.val[0], e.val[1], e.val[2]); }
    Matrix3f eigenvectors() const { Matrix3f V; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) V(i, j) = e.vec[i][j]; return V; }
};
// SelfAdjointEigenSolver<Matrix<double, 6, 6>> (evalDegenracy): eigenvalues ascending, eigenvectors in columns -- the oracle's cyclic Jacobi
// (library arithmetic restated; evalDegenracy's projector V_f^-T V_p^T does not depend on the eigenvectors' signs)
template <> struct SelfAdjointEigenSolver<Matrix<double, 6, 6>> {
    double val[6], vec[36];
    explicit SelfAdjointEigenSolver(const Matrix<double, 6, 6> &A) { orc::jacobi_eig_sym_d(A.d, 6, val, vec); }
    Matrix<double, 6, 1> eigenvalues() const { Matrix<double, 6, 1> m; for (int i = 0; i < 6; ++i) m.d[i] = val[i]; return m; }
    Matrix<double, 6, 6> eigenvectors() const { Matrix<double, 6, 6> m; for (int i = 0; i < 36; ++i) m.d[i] = vec[i]; return m; }
};
struct MatrixXf {
    int r = 0, c = 0;
    std::vector<float> v;                                          // row-major
    static MatrixXf Zero(int rr, int cc) { MatrixXf m; m.r = rr; m.c = cc; m.v.assign(size_t(rr) * cc, 0.f); return m; }
    static MatrixXf Constant(int rr, int cc, float x) { MatrixXf m; m.r = rr; m.c = cc; m.v.assign(size_t(rr) * cc, x); return m; }
    // an out-of-range element (the reference's 64-ring ground loop reads row 64: oracle/image_segmenter.hpp U3) reads as "empty" (FLT_MAX)
    // instead of whatever lies behind the buffer
    float &operator()(int i, int j) { static float outside; if (i < 0 || i >= r || j < 0 || j >= c) { outside = 3.402823466e+38f; return outside; } return v[size_t(i) * c + j]; }
    struct QR {
        const MatrixXf &A;
        Vector3f solve(const MatrixXf &b) const { float x[3]; orc::colpiv_qr_solve_f(A.v.data(), b.v.data(), A.r, x); return Vector3f(x[0], x[1], x[2]); }
    };
    QR colPivHouseholderQr() const { return QR{*this}; }
};
struct MatrixXi {
    int r = 0, c = 0;
    std::vector<int> v;
    static MatrixXi Zero(int rr, int cc) { MatrixXi m; m.r = rr; m.c = cc; m.v.assign(size_t(rr) * cc, 0); return m; }
    int &operator()(int i, int j) { static int outside; if (i < 0 || i >= r || j < 0 || j >= c) { outside = -1; return outside; } return v[size_t(i) * c + j]; }
};
}  // namespace Eigen
