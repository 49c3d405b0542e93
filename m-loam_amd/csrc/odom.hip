// Batched evaluation of the odometry-side three-block factors on gfx950:
//   LidarPureOdomPlaneNormFactor::Evaluate   estimator/src/factor/lidar_pure_odom_factor.hpp:38-102
//   LidarPureOdomEdgeFactor::Evaluate        estimator/src/factor/lidar_pure_odom_factor.hpp:209-282
// as added by Estimator::optimizeMap for every (window frame i, LiDAR n, feature) with the parameter blocks
// (para_pose_[0] = pivot, para_pose_[i - pivot], para_ex_pose_[n])  (estimator.cpp:700-780).
// The point is moved with T = T_pivot^-1 T_i T_ext; each factor yields one residual and three 1x7 row-major Jacobians
// (7th column zero). One lane per factor, f64; the factor table (point, coefficients, weight, block indices) is staged
// once per optimisation (the correspondences do not change inside ceres::Solve), each evaluation streams
// 72 B in + 176 B out per factor -- HBM streaming, no reuse.
// The Jacobian columns are the reference's formulas term by term, including the two columns that are not the true derivative
// (plane/pivot rotation: w^T Rp^T [v]x; edge/extrinsic rotation: Rext [p]x + [t_ext]x) -- Ceres is driven by what the
// reference hands it, so parity means reproducing those.
#include "ctx.hpp"
#include "dev_math.hpp"

namespace mlh {

struct OdomArgs {
    const double *tab;      // n x 10: point[3], coeff[6], sqrt_info
    const int *idx;         // n x 3: type (0 plane, 1 edge), frame index, extrinsic index
    const double *pivot;    // 7
    const double *frames;   // n_frames x 7
    const double *exts;     // n_ext x 7
    int n, n_frames, n_ext;
    double *r;              // n
    double *J;              // n x 21 (pivot | frame | extrinsic), or null
};

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 rowmul(const V3 &a, const double (&M)[9])      // a^T M
{
    return {a.x * M[0] + a.y * M[3] + a.z * M[6], a.x * M[1] + a.y * M[4] + a.z * M[7], a.x * M[2] + a.y * M[5] + a.z * M[8]};
}
__device__ __forceinline__ V3 matmul(const double (&M)[9], const V3 &v)      // M v
{
    return {M[0] * v.x + M[1] * v.y + M[2] * v.z, M[3] * v.x + M[4] * v.y + M[5] * v.z, M[6] * v.x + M[7] * v.y + M[8] * v.z};
}
__device__ __forceinline__ V3 tmatmul(const double (&M)[9], const V3 &v)     // M^T v
{
    return {M[0] * v.x + M[3] * v.y + M[6] * v.z, M[1] * v.x + M[4] * v.y + M[7] * v.z, M[2] * v.x + M[5] * v.y + M[8] * v.z};
}
__device__ __forceinline__ V3 row_skew(const V3 &a, const V3 &v)             // a^T [v]x
{
    return {a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x};
}
__device__ __forceinline__ V3 crossv(const V3 &a, const V3 &b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

__global__ __launch_bounds__(256) void pure_odom_kernel(OdomArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const double *tb = A.tab + size_t(i) * 10;
    const int type = A.idx[i * 3 + 0];
    const int fi = min(max(A.idx[i * 3 + 1], 0), A.n_frames - 1), ei = min(max(A.idx[i * 3 + 2], 0), A.n_ext - 1);
    const double *pp = A.pivot, *pi = A.frames + fi * 7, *pe = A.exts + ei * 7;
    const q4 Qp{pp[3], pp[4], pp[5], pp[6]}, Qi{pi[3], pi[4], pi[5], pi[6]}, Qe{pe[3], pe[4], pe[5], pe[6]};
    const V3 tp{pp[0], pp[1], pp[2]}, ti{pi[0], pi[1], pi[2]}, te{pe[0], pe[1], pe[2]};
    const V3 p{tb[0], tb[1], tb[2]};
    const double s = tb[9];
    // T = T_pivot^-1 * T_i * T_ext as quaternion products (hpp:47-53 / 218-224)
    const q4 Qpc{-Qp.x, -Qp.y, -Qp.z, Qp.w};
    const q4 Qpi = qmul(Qpc, Qi);
    const d3 tpi = qrot(Qpc, d3{ti.x - tp.x, ti.y - tp.y, ti.z - tp.z});
    const q4 Qx = qmul(Qpi, Qe);
    const d3 rte = qrot(Qpi, d3{te.x, te.y, te.z});
    const d3 rp = qrot(Qx, d3{p.x, p.y, p.z});
    const V3 lp{rp.x + (rte.x + tpi.x), rp.y + (rte.y + tpi.y), rp.z + (rte.z + tpi.z)};
    double Rp[9], Ri[9], Re[9];
    qtorot(Qp, Rp); qtorot(Qi, Ri); qtorot(Qe, Re);
    const V3 Rep = matmul(Re, p);                         // Rext p
    const V3 Rite = matmul(Ri, te);                       // Ri t_ext
    const V3 RiRep = matmul(Ri, Rep);                     // Ri Rext p
    const V3 v{RiRep.x + Rite.x + ti.x - tp.x, RiRep.y + Rite.y + ti.y - tp.y, RiRep.z + Rite.z + ti.z - tp.z};
    V3 a;            // the 1x3 row that multiplies d(lp): w^T (plane) or eta [ba - bb]x (edge)
    double res;
    if (type == 0) {
        a = V3{tb[3], tb[4], tb[5]};
        res = (a.x * lp.x + a.y * lp.y + a.z * lp.z) + tb[6];
    } else {
        const V3 la{tb[3], tb[4], tb[5]}, lb{tb[6], tb[7], tb[8]};
        const V3 ba{lp.x - la.x, lp.y - la.y, lp.z - la.z}, bb{lp.x - lb.x, lp.y - lb.y, lp.z - lb.z};
        const V3 nu = crossv(ba, bb);
        const V3 de{la.x - lb.x, la.y - lb.y, la.z - lb.z};
        const double nu_n = sqrt(nu.x * nu.x + nu.y * nu.y + nu.z * nu.z), de_n = sqrt(de.x * de.x + de.y * de.y + de.z * de.z);
        res = nu_n / de_n;
        V3 nh = nu;
        const double n2 = nu.x * nu.x + nu.y * nu.y + nu.z * nu.z;
        if (n2 > 0.0) { const double nn = sqrt(n2); nh = V3{nu.x / nn, nu.y / nn, nu.z / nn}; }   // Eigen normalized(): zero stays zero
        const double k = 1.0 / de_n;
        const V3 eta{k * nh.x, k * nh.y, k * nh.z};
        a = row_skew(eta, V3{ba.x - bb.x, ba.y - bb.y, ba.z - bb.z});
    }
    A.r[i] = s * res;
    if (!A.J) return;
    double *J = A.J + size_t(i) * 21;
    // row0 = a^T Rp^T: component c = sum_k a_k Rp[c][k] = (Rp a)_c
    const V3 row0 = matmul(Rp, a);
    // pivot block
    V3 rot0;
    if (type == 0) {
        // w^T (Rp^T [v]x): (w^T Rp^T) [v]x
        rot0 = row_skew(row0, v);
    } else {
        // eta[ba-bb]x [Rp^T v]x
        rot0 = row_skew(a, tmatmul(Rp, v));
    }
    J[0] = s * (-row0.x); J[1] = s * (-row0.y); J[2] = s * (-row0.z);
    J[3] = s * rot0.x; J[4] = s * rot0.y; J[5] = s * rot0.z; J[6] = 0.0;
    // frame block: [ a Rp^T | -(a Rp^T Ri) [Rext p + t_ext]x ]
    const V3 row1 = rowmul(row0, Ri);
    const V3 rot1 = row_skew(row1, V3{Rep.x + te.x, Rep.y + te.y, Rep.z + te.z});
    J[7] = s * row0.x; J[8] = s * row0.y; J[9] = s * row0.z;
    J[10] = s * (-rot1.x); J[11] = s * (-rot1.y); J[12] = s * (-rot1.z); J[13] = 0.0;
    // extrinsic block: [ a Rp^T Ri | -(a Rp^T Ri) X ],  X = [Rext p]x (plane) or Rext [p]x + [t_ext]x (edge)
    V3 rot2;
    if (type == 0) {
        rot2 = row_skew(row1, Rep);
    } else {
        const V3 t1 = row_skew(rowmul(row1, Re), p);      // row1 Rext [p]x
        const V3 t2 = row_skew(row1, te);
        rot2 = V3{t1.x + t2.x, t1.y + t2.y, t1.z + t2.z};
    }
    J[14] = s * row1.x; J[15] = s * row1.y; J[16] = s * row1.z;
    J[17] = s * (-rot2.x); J[18] = s * (-rot2.y); J[19] = s * (-rot2.z); J[20] = 0.0;
}

int pure_odom_set(mlh_ctx *ctx, int n, const int32_t *type, const double *points, const double *coeffs, const double *sqrt_info,
                  const int32_t *frame_idx, const int32_t *ext_idx)
{
    if (n <= 0 || !type || !points || !coeffs || !frame_idx || !ext_idx) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    OdomSet &O = ctx->odom;
    std::vector<double> tab(size_t(n) * 10);
    std::vector<int> idx(size_t(n) * 3);
    for (int i = 0; i < n; ++i) {
        double *t = tab.data() + size_t(i) * 10;
        for (int k = 0; k < 3; ++k) t[k] = points[size_t(i) * 3 + k];
        for (int k = 0; k < 6; ++k) t[3 + k] = coeffs[size_t(i) * 6 + k];
        t[9] = sqrt_info ? sqrt_info[i] : 1.0;              // the reference constructs these factors with s = 1.0 (estimator.cpp:735, 751)
        if (type[i] != 0 && type[i] != 1) return fail(ctx, MLH_ERR_INVALID, "factor type must be 0 (plane) or 1 (edge)");
        idx[size_t(i) * 3 + 0] = type[i]; idx[size_t(i) * 3 + 1] = frame_idx[i]; idx[size_t(i) * 3 + 2] = ext_idx[i];
    }
    MLH_HIP(ctx, O.tab.ensure(sizeof(double) * tab.size()));
    MLH_HIP(ctx, O.idx.ensure(sizeof(int) * idx.size()));
    MLH_HIP(ctx, O.r.ensure(sizeof(double) * size_t(n)));
    MLH_HIP(ctx, O.J.ensure(sizeof(double) * 21 * size_t(n)));
    MLH_HIP(ctx, hipMemcpyAsync(O.tab.p, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(O.idx.p, idx.data(), sizeof(int) * idx.size(), hipMemcpyHostToDevice, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    O.n = n;
    int mf = 0, me = 0;
    for (int i = 0; i < n; ++i) { mf = std::max(mf, frame_idx[i]); me = std::max(me, ext_idx[i]); if (frame_idx[i] < 0 || ext_idx[i] < 0) return fail(ctx, MLH_ERR_INVALID, "negative block index"); }
    O.max_frame = mf; O.max_ext = me;
    return MLH_OK;
}

int pure_odom_evaluate(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                       double *residuals, double *jacobians)
{
    OdomSet &O = ctx->odom;
    if (O.n <= 0) return fail(ctx, MLH_ERR_STATE, "mlh_pure_odom_set has not been called");
    if (!pivot || !frames || !exts || !residuals || n_frames <= O.max_frame || n_ext <= O.max_ext)
        return fail(ctx, MLH_ERR_INVALID, "pose arrays do not cover the block indices of the staged factors");
    hipStream_t st = ctx->stream;
    const size_t np = 7 * size_t(1 + n_frames + n_ext);
    MLH_HIP(ctx, O.poses.ensure(sizeof(double) * np));
    std::vector<double> h(np);
    for (int k = 0; k < 7; ++k) h[k] = pivot[k];
    for (size_t k = 0; k < 7 * size_t(n_frames); ++k) h[7 + k] = frames[k];
    for (size_t k = 0; k < 7 * size_t(n_ext); ++k) h[7 + 7 * size_t(n_frames) + k] = exts[k];
    MLH_HIP(ctx, hipMemcpyAsync(O.poses.p, h.data(), sizeof(double) * np, hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    OdomArgs A;
    A.tab = O.tab.as<double>(); A.idx = O.idx.as<int>();
    A.pivot = O.poses.as<double>(); A.frames = A.pivot + 7; A.exts = A.frames + 7 * size_t(n_frames);
    A.n = O.n; A.n_frames = n_frames; A.n_ext = n_ext; A.r = O.r.as<double>(); A.J = jacobians ? O.J.as<double>() : nullptr;
    hipLaunchKernelGGL(pure_odom_kernel, dim3((O.n + 255) / 256), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(residuals, O.r.p, sizeof(double) * size_t(O.n), hipMemcpyDeviceToHost, st));
    if (jacobians) MLH_HIP(ctx, hipMemcpyAsync(jacobians, O.J.p, sizeof(double) * 21 * size_t(O.n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    return MLH_OK;
}

}  // namespace mlh
