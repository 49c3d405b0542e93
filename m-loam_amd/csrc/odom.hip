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
#include <cfloat>

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

// one factor: residual and (J != null) its three 1x7 rows [pivot | frame | extrinsic]; tb = point[3], coeff[6], sqrt_info; pp / pi / pe = the three poses
__device__ __forceinline__ void odom_factor_core(const double *tb, int type, const double *pp, const double *pi, const double *pe, double &r_out, double *J);
__device__ __forceinline__ void odom_factor(const OdomArgs &A, int i, double &r_out, double *J)
{
    const int fi = min(max(A.idx[i * 3 + 1], 0), A.n_frames - 1), ei = min(max(A.idx[i * 3 + 2], 0), A.n_ext - 1);
    odom_factor_core(A.tab + size_t(i) * 10, A.idx[i * 3 + 0], A.pivot, A.frames + fi * 7, A.exts + ei * 7, r_out, J);
}
__device__ __forceinline__ void odom_factor_core(const double *tb, int type, const double *pp, const double *pi, const double *pe, double &r_out, double *J)
{
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
    r_out = s * res;
    if (!J) return;
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

// The row Estimator::goodFeatureMatching scores a matched feature by (Estimator::evaluateFeatJacobian, estimator.cpp:1273-1345): a surf feature's
// LidarPureOdomPlaneNormFactor(point, coeffs, 1.0) evaluated at (pivot, pose_i, ext), the first six columns of its FRAME block; a corner feature's row is
// Matrix<1, 6>::Identity() = (1 0 0 0 0 0) whatever the feature (cpp:1339-1342). For every staged feature of the kind at once, beside the validity byte.
struct OdomRowArgs {
    const float4 *feat;
    const Corr *corr;
    int m, type;
    double poses[21];       // pivot | pose_i | ext
    double *J6;             // m x 6
    uint8_t *valid;         // m
};
__global__ __launch_bounds__(256) void odom_feat_rows_kernel(OdomRowArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.m) return;
    const float4 f = A.feat[i];
    const Corr c = A.corr[i];
    const bool v = c.valid != 0 && f.w >= 0.f;
    double row[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (v) {
        if (A.type == 0) {
            double tb[10] = {double(f.x), double(f.y), double(f.z), double(c.c[0]), double(c.c[1]), double(c.c[2]), double(c.c[3]), double(c.c[4]), double(c.c[5]), 1.0};
            double r, J[21];
            odom_factor_core(tb, 0, A.poses, A.poses + 7, A.poses + 14, r, J);
#pragma unroll
            for (int k = 0; k < 6; ++k) row[k] = J[7 + k];
        } else {
            row[0] = 1.0;
        }
    }
    A.valid[i] = v ? 1 : 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) A.J6[size_t(i) * 6 + k] = row[k];
}

int pure_odom_feature_rows(mlh_ctx *ctx, int kind, const double pivot[7], const double pose_i[7], const double ext[7])
{
    FeatSet &f = ctx->feat[kind];
    if (f.m <= 0 || !f.matched || f.n_blocks != 1) return fail(ctx, MLH_ERR_STATE, "a single-block match pass must have run for this kind");
    MLH_HIP(ctx, f.J.ensure(sizeof(double) * 6 * size_t(f.m)));
    MLH_HIP(ctx, f.flag8.ensure((size_t(f.m) + 63) & ~size_t(63)));
    OdomRowArgs A;
    A.feat = f.pts.as<float4>(); A.corr = f.corr.as<Corr>(); A.m = f.m; A.type = kind;
    for (int k = 0; k < 7; ++k) { A.poses[k] = pivot[k]; A.poses[7 + k] = pose_i[k]; A.poses[14 + k] = ext[k]; }
    A.J6 = f.J.as<double>(); A.valid = f.flag8.as<uint8_t>();
    MLH_LAUNCH(odom_feat_rows_kernel, dim3((f.m + 255) / 256), dim3(256), 0, ctx->stream, A);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

__global__ __launch_bounds__(256) void pure_odom_kernel(OdomArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    double r;
    odom_factor(A, i, r, A.J ? A.J + size_t(i) * 21 : nullptr);
    A.r[i] = r;
}

// ---- normal equations of the coupled window problem (estimator.cpp:687-848): J^T J over the local parameters [pivot | frames | extrinsics]
// A factor touches three 6-wide blocks (pivot, frame f, extrinsic e): v = its 18 loss-corrected Jacobian entries. Factors are grouped by
// (f, e) when they are staged (every 256-factor tile belongs to ONE group), so a tile's contribution is one 18 x 18 symmetric block pattern:
// the tile's rows go to LDS (18 v + corrected residual + cost + count per factor), then thread t owns output entry t -- one of the 171
// upper-triangle products v_a v_b, the 18 v_a r, cost, count -- and sums it over the tile's rows in row order (fixed order: deterministic).
// odom_ne_finish_kernel adds the tiles of a group in tile order and scatters the groups into the D x D matrix (LDS-resident), again in
// fixed order. No atomics anywhere.
constexpr int NE_ROW = 22;        // LDS doubles per factor: v[18], r, cost, count, pad
constexpr int NE_OUT = 192;       // 171 + 18 + 2, padded
struct OdomNeArgs {
    OdomArgs A;
    const int *perm;        // tiles * 256 factor indices grouped by (frame, ext); -1 = padding
    double huber_delta;
    double *partial;        // tiles x NE_OUT
};

__device__ __forceinline__ void ne_entry(int t, int &a, int &b)     // t < 171 -> (a, b), a <= b < 18, row-major upper triangle
{
    int row = 0, left = t;
    while (left >= 18 - row) { left -= 18 - row; ++row; }
    a = row; b = row + left;
}

__global__ __launch_bounds__(256) void odom_ne_kernel(OdomNeArgs G)
{
    __shared__ double V[256 * NE_ROW];
    const int tile = blockIdx.x, k = threadIdx.x;
    const int i = G.perm[tile * 256 + k];
    double *row = V + k * NE_ROW;
    if (i >= 0) {
        double r, J[21];
        odom_factor(G.A, i, r, J);
        double sq = r * r, rho0 = sq, rho1 = 1.0;
        if (G.huber_delta > 0.0) {
            const double bb = G.huber_delta * G.huber_delta;
            if (sq > bb) { const double rr = sqrt(sq); rho0 = 2.0 * G.huber_delta * rr - bb; rho1 = fmax(DBL_MIN, G.huber_delta / rr); }
        }
        const double sc = sqrt(rho1);
#pragma unroll
        for (int bl = 0; bl < 3; ++bl)
#pragma unroll
            for (int c = 0; c < 6; ++c) row[bl * 6 + c] = J[bl * 7 + c] * sc;
        row[18] = r * sc; row[19] = 0.5 * rho0; row[20] = 1.0; row[21] = 0.0;
    } else {
#pragma unroll
        for (int c = 0; c < NE_ROW; ++c) row[c] = 0.0;
    }
    __syncthreads();
    if (k < NE_OUT) {
        double acc = 0.0;
        if (k < 171 + 18 + 2) {
            int a, b;
            if (k < 171) ne_entry(k, a, b);
            else if (k < 189) { a = k - 171; b = 18; }
            else { a = k - 189 + 19; b = -1; }                           // cost / count columns, summed as they are
            if (b >= 0) { for (int q = 0; q < 256; ++q) acc += V[q * NE_ROW + a] * V[q * NE_ROW + b]; }
            else { for (int q = 0; q < 256; ++q) acc += V[q * NE_ROW + a]; }
        }
        G.partial[size_t(tile) * NE_OUT + k] = acc;
    }
}

struct OdomNeFinish {
    const double *partial;
    const int *tile_group;  // per tile: frame * n_ext + ext
    int n_tiles, n_frames, n_ext;
    double *out;            // D*D + D + 2
};
__global__ __launch_bounds__(256) void odom_ne_finish_kernel(OdomNeFinish F)
{
    extern __shared__ double Hs[];            // D*D + D + 2
    const int D = 6 * (1 + F.n_frames + F.n_ext), n_out = D * D + D + 2;
    for (int q = threadIdx.x; q < n_out; q += 256) Hs[q] = 0.0;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < 191) {
        int a = 0, b = 0;
        if (t < 171) ne_entry(t, a, b);
        else if (t < 189) a = t - 171;
        int tile = 0;
        while (tile < F.n_tiles) {
            const int g = F.tile_group[tile];
            double acc = 0.0;
            while (tile < F.n_tiles && F.tile_group[tile] == g) { acc += F.partial[size_t(tile) * NE_OUT + t]; ++tile; }   // tiles of a group: in order
            const int f = g / F.n_ext, e = g % F.n_ext;
            const int off[3] = {0, 6 * (1 + f), 6 * (1 + F.n_frames + e)};
            if (t < 171) {
                const int ra = off[a / 6] + a % 6, rb = off[b / 6] + b % 6;     // ra <= rb: the blocks are ordered pivot < frames < extrinsics
                Hs[ra * D + rb] += acc;
                if (ra != rb) Hs[rb * D + ra] += acc;
            } else if (t < 189) {
                Hs[D * D + off[a / 6] + a % 6] += acc;
            } else {
                Hs[D * D + D + (t - 189)] += acc;
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < n_out; q += 256) F.out[q] = Hs[q];
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
    // factors grouped by (frame, extrinsic) for the normal-equation kernel: stable counting sort of the indices, every group padded to whole
    // 256-factor tiles (the group key uses the staged maxima; mlh_pure_odom_normal_eq re-derives (f, e) from it)
    {
        const int ne = me + 1, ng = (mf + 1) * ne;
        std::vector<int> cnt(size_t(ng) + 1, 0);
        for (int i = 0; i < n; ++i) cnt[size_t(frame_idx[i]) * ne + ext_idx[i] + 1]++;
        std::vector<int> tile_start(size_t(ng) + 1, 0);
        for (int g = 0; g < ng; ++g) tile_start[g + 1] = tile_start[g] + (cnt[g + 1] + 255) / 256;
        const int n_tiles = tile_start[ng];
        std::vector<int> perm(size_t(n_tiles) * 256, -1), fill(ng, 0), tgrp(size_t(n_tiles), 0);
        for (int i = 0; i < n; ++i) {
            const int g = frame_idx[i] * ne + ext_idx[i];
            perm[size_t(tile_start[g]) * 256 + fill[g]++] = i;
        }
        for (int g = 0; g < ng; ++g) for (int t = tile_start[g]; t < tile_start[g + 1]; ++t) tgrp[t] = g;
        O.n_tiles = n_tiles; O.group_ext = ne;
        if (n_tiles > 0) {
            MLH_HIP(ctx, O.perm.ensure(sizeof(int) * perm.size()));
            MLH_HIP(ctx, O.tile_group.ensure(sizeof(int) * tgrp.size()));
            MLH_HIP(ctx, O.partial.ensure(sizeof(double) * NE_OUT * size_t(n_tiles)));
            MLH_HIP(ctx, hipMemcpyAsync(O.perm.p, perm.data(), sizeof(int) * perm.size(), hipMemcpyHostToDevice, ctx->stream));
            MLH_HIP(ctx, hipMemcpyAsync(O.tile_group.p, tgrp.data(), sizeof(int) * tgrp.size(), hipMemcpyHostToDevice, ctx->stream));
            MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        }
        O.h_tile_group = tgrp;
        O.h_tile_frame.clear(); O.h_tile_ext.clear();
        for (int t = 0; t < n_tiles; ++t) { O.h_tile_frame.push_back(tgrp[t] / ne); O.h_tile_ext.push_back(tgrp[t] % ne); }
        O.tile_group_keyed = true;
    }
    O.device_built = false;
    return MLH_OK;
}

// ---- device-resident factor table: the correspondences a match pass left in HBM become LidarPureOdom factors without visiting the host
// (Estimator::optimizeMap builds them from the features matched against the local map, estimator.cpp:700-780: point = the feature in its
// LiDAR's frame, coefficients = the fitted plane / line in the pivot frame, s = 1.0). One call = one (frame, extrinsic) group of one kind:
// its region of the table is a whole number of 256-factor tiles reserved from the staged feature count (no host round trip for the number
// of valid ones); the valid correspondences are packed to the front IN FEATURE ORDER by a single-workgroup scan (deterministic), the rest
// of the region is padding (perm = -1), which the normal-equation kernel skips.
struct OdomAppend {
    const float4 *feat;
    const Corr *corr;
    int m, type, frame, ext, base_slot, cap_slots;
    double *tab;
    int *idx, *perm;
};
__global__ __launch_bounds__(1024) void odom_append_kernel(OdomAppend P)
{
    __shared__ int wsum[16];
    __shared__ int s_running;
    if (threadIdx.x == 0) s_running = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = 0; c0 < P.m; c0 += 1024) {
        const int i = c0 + threadIdx.x;
        const bool v = i < P.m && P.corr[i].valid != 0 && P.feat[i].w >= 0.f;
        const unsigned long long b = __ballot(v);
        const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(b);
        __syncthreads();
        int before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int x = wsum[w]; if (w < wave) before += x; total += x; }
        const int running = s_running;
        if (v) {
            const int slot = P.base_slot + running + before + in_wave;
            const float4 f = P.feat[i];
            const Corr c = P.corr[i];
            double *t = P.tab + size_t(slot) * 10;
            t[0] = double(f.x); t[1] = double(f.y); t[2] = double(f.z);
#pragma unroll
            for (int k = 0; k < 6; ++k) t[3 + k] = double(c.c[k]);
            t[9] = 1.0;
            P.idx[size_t(slot) * 3 + 0] = P.type; P.idx[size_t(slot) * 3 + 1] = P.frame; P.idx[size_t(slot) * 3 + 2] = P.ext;
            P.perm[slot] = slot;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_running = running + total;
        __syncthreads();
    }
    for (int q = s_running + threadIdx.x; q < P.cap_slots; q += 1024) P.perm[P.base_slot + q] = -1;
}

int pure_odom_begin(mlh_ctx *ctx)
{
    OdomSet &O = ctx->odom;
    O.n = 0; O.n_tiles = 0; O.max_frame = 0; O.max_ext = 0; O.group_ext = 1; O.tile_group_keyed = false; O.device_built = true;
    O.h_tile_group.clear(); O.h_tile_frame.clear(); O.h_tile_ext.clear();
    return MLH_OK;
}

int pure_odom_add_matches(mlh_ctx *ctx, int kind, int frame_idx, int ext_idx)
{
    OdomSet &O = ctx->odom;
    if (!O.device_built) return fail(ctx, MLH_ERR_STATE, "mlh_pure_odom_begin has not been called");
    if (frame_idx < 0 || ext_idx < 0) return fail(ctx, MLH_ERR_INVALID, "negative block index");
    FeatSet &f = ctx->feat[kind];
    if (f.m <= 0 || !f.matched || f.n_blocks != 1) return fail(ctx, MLH_ERR_STATE, "a single-block match pass must have run for this kind");
    hipStream_t st = ctx->stream;
    const int cap_tiles = (f.m + 255) / 256, cap_slots = cap_tiles * 256, base_slot = O.n_tiles * 256, n_new = base_slot + cap_slots;
    hipError_t e;
    if ((e = O.tab.grow(sizeof(double) * 10 * size_t(n_new), sizeof(double) * 10 * size_t(base_slot), st)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc factor table", e);
    if ((e = O.idx.grow(sizeof(int) * 3 * size_t(n_new), sizeof(int) * 3 * size_t(base_slot), st)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc factor table", e);
    if ((e = O.perm.grow(sizeof(int) * size_t(n_new), sizeof(int) * size_t(base_slot), st)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc factor table", e);
    OdomAppend P;
    P.feat = f.pts.as<float4>(); P.corr = f.corr.as<Corr>(); P.m = f.m; P.type = kind; P.frame = frame_idx; P.ext = ext_idx;
    P.base_slot = base_slot; P.cap_slots = cap_slots; P.tab = O.tab.as<double>(); P.idx = O.idx.as<int>(); P.perm = O.perm.as<int>();
    MLH_LAUNCH(odom_append_kernel, dim3(1), dim3(1024), 0, st, P);
    MLH_HIP(ctx, hipGetLastError());
    O.n = n_new; O.n_tiles += cap_tiles;
    O.max_frame = std::max(O.max_frame, frame_idx); O.max_ext = std::max(O.max_ext, ext_idx);
    // group keys are kept as (frame, ext) pairs until the window's extrinsic count is known (pure_odom_normal_eq keys and uploads them)
    for (int t = 0; t < cap_tiles; ++t) { O.h_tile_frame.push_back(frame_idx); O.h_tile_ext.push_back(ext_idx); }
    O.tile_group_keyed = false;
    return MLH_OK;
}

int pure_odom_evaluate(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                       double *residuals, double *jacobians)
{
    if (ctx->odom.device_built) return fail(ctx, MLH_ERR_STATE, "per-factor outputs need a host-staged table (mlh_pure_odom_set): a device-built one is padded");
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
    MLH_LAUNCH(pure_odom_kernel, dim3((O.n + 255) / 256), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(residuals, O.r.p, sizeof(double) * size_t(O.n), hipMemcpyDeviceToHost, st));
    if (jacobians) MLH_HIP(ctx, hipMemcpyAsync(jacobians, O.J.p, sizeof(double) * 21 * size_t(O.n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    return MLH_OK;
}

static int odom_upload_poses(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext, OdomArgs &A)
{
    OdomSet &O = ctx->odom;
    hipStream_t st = ctx->stream;
    const size_t np = 7 * size_t(1 + n_frames + n_ext);
    MLH_HIP(ctx, O.poses.ensure(sizeof(double) * np));
    std::vector<double> h(np);
    for (int k = 0; k < 7; ++k) h[k] = pivot[k];
    for (size_t k = 0; k < 7 * size_t(n_frames); ++k) h[7 + k] = frames[k];
    for (size_t k = 0; k < 7 * size_t(n_ext); ++k) h[7 + 7 * size_t(n_frames) + k] = exts[k];
    MLH_HIP(ctx, hipMemcpyAsync(O.poses.p, h.data(), sizeof(double) * np, hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));     // h is a stack-lifetime buffer
    A.tab = O.tab.as<double>(); A.idx = O.idx.as<int>();
    A.pivot = O.poses.as<double>(); A.frames = A.pivot + 7; A.exts = A.frames + 7 * size_t(n_frames);
    A.n = O.n; A.n_frames = n_frames; A.n_ext = n_ext; A.r = nullptr; A.J = nullptr;
    return MLH_OK;
}

// stages the poses, keys the tile groups for this call's extrinsic count and sizes the outputs; what both entry points below start with
static int odom_ne_prepare(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext, OdomNeArgs &G, size_t &n_out)
{
    OdomSet &O = ctx->odom;
    if (O.n <= 0) return fail(ctx, MLH_ERR_STATE, "mlh_pure_odom_set has not been called");
    if (!pivot || !frames || !exts || n_frames <= O.max_frame || n_ext <= O.max_ext)
        return fail(ctx, MLH_ERR_INVALID, "pose arrays do not cover the block indices of the staged factors");
    const int D = 6 * (1 + n_frames + n_ext);
    n_out = size_t(D) * D + D + 2;
    if (n_out * sizeof(double) > 150 * 1024) return fail(ctx, MLH_ERR_UNSUPPORTED, "window too large for the LDS-resident assembly (6 (1 + frames + extrinsics) <= 136)");
    int rc = odom_upload_poses(ctx, pivot, frames, n_frames, exts, n_ext, G.A);
    if (rc) return rc;
    hipStream_t st = ctx->stream;
    // tile -> group keys depend on this call's extrinsic count: (re)key and upload them when it changed or the table was built on the device
    if (O.group_ext != n_ext || !O.tile_group_keyed) {
        std::vector<int> tg(O.h_tile_frame.size());
        for (size_t t = 0; t < tg.size(); ++t) tg[t] = O.h_tile_frame[t] * n_ext + O.h_tile_ext[t];
        MLH_HIP(ctx, O.tile_group.ensure(sizeof(int) * std::max<size_t>(tg.size(), 1)));
        MLH_HIP(ctx, hipMemcpyAsync(O.tile_group.p, tg.data(), sizeof(int) * tg.size(), hipMemcpyHostToDevice, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        O.group_ext = n_ext; O.tile_group_keyed = true;
    }
    MLH_HIP(ctx, O.partial.ensure(sizeof(double) * NE_OUT * size_t(O.n_tiles)));
    MLH_HIP(ctx, O.ne_out.ensure(sizeof(double) * n_out));
    MLH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(odom_ne_finish_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(n_out * sizeof(double))));
    G.perm = O.perm.as<int>(); G.partial = O.partial.as<double>();
    return MLH_OK;
}

// one evaluation of the coupled normal equations at the poses resident in O.poses -> O.ne_out (two launches, nothing waited for)
static void odom_ne_enqueue(mlh_ctx *ctx, const OdomNeArgs &G, int n_frames, int n_ext, size_t n_out)
{
    OdomSet &O = ctx->odom;
    MLH_LAUNCH(odom_ne_kernel, dim3(O.n_tiles), dim3(256), 0, ctx->stream, G);
    OdomNeFinish F;
    F.partial = O.partial.as<double>(); F.tile_group = O.tile_group.as<int>(); F.n_tiles = O.n_tiles; F.n_frames = n_frames; F.n_ext = n_ext;
    F.out = O.ne_out.as<double>();
    MLH_LAUNCH(odom_ne_finish_kernel, dim3(1), dim3(256), n_out * sizeof(double), ctx->stream, F);
}

int pure_odom_normal_eq(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext, double huber_delta,
                        double *H, double *g, double *cost, int32_t *n_res)
{
    if (!H || !g) return fail(ctx, MLH_ERR_INVALID, "null output");
    OdomSet &O = ctx->odom;
    OdomNeArgs G;
    size_t n_out = 0;
    int rc = odom_ne_prepare(ctx, pivot, frames, n_frames, exts, n_ext, G, n_out);
    if (rc) return rc;
    const int D = 6 * (1 + n_frames + n_ext);
    G.huber_delta = huber_delta;
    odom_ne_enqueue(ctx, G, n_frames, n_ext, n_out);
    MLH_HIP(ctx, hipGetLastError());
    hipStream_t st = ctx->stream;
    std::vector<double> h(n_out);
    MLH_HIP(ctx, hipMemcpyAsync(h.data(), O.ne_out.p, sizeof(double) * n_out, hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    std::memcpy(H, h.data(), sizeof(double) * size_t(D) * D);
    std::memcpy(g, h.data() + size_t(D) * D, sizeof(double) * D);
    if (cost) *cost = h[size_t(D) * D + D];
    if (n_res) *n_res = int(h[size_t(D) * D + D + 1] + 0.5);
    return MLH_OK;
}

// ---- the coupled window problem solved ON THE DEVICE: Gauss-Newton on [pivot | frames | extrinsics] with some blocks held
// constant -- Estimator::optimizeMap holds para_pose_[0] and para_ex_pose_[IDX_REF] constant (estimator.cpp:636, 642) -- and a per-block V_update
// (what Estimator::evalDegenracy leaves in every PoseLocalParameterization before the solve, estimator.cpp:1598-1680: identity, a projector, or zero
// for a frozen extrinsic). One workgroup: the free rows / columns of J^T J are gathered into LDS, factorised by a right-looking Cholesky (the trailing
// update spread over the 256 threads; + 1e-6 I and a second attempt when a pivot is not positive), two column-oriented substitutions, then one thread
// per free block applies PoseLocalParameterization::Plus. An iteration is three launches (rows -> tile sums, assembly, solve); the poses never leave HBM.
struct OdomSolveArgs {
    const double *ne;       // D*D + D + 2 (odom_ne_finish_kernel's output)
    double *poses;          // 7 per block: pivot, frames, extrinsics
    const double *V;        // 36 per block (row-major V_update_) or null = identity everywhere
    int n_blocks;
    unsigned free_mask;     // bit b: block b is updated
    int *status;            // [0]: 0 ok, 1 solved with + 1e-6 I, 2 not positive definite even then (no update); [1]: iterations done
};
__global__ __launch_bounds__(256) void odom_window_solve_kernel(OdomSolveArgs S)
{
    extern __shared__ double sm[];                      // A (nf x nf, row-major) | b (nf) | x (nf)
    __shared__ int s_row[136];                           // free row -> row of the full system
    __shared__ int s_ok;
    const int t = threadIdx.x;
    const int D = 6 * S.n_blocks;
    int nf = 0;
    for (int b = 0; b < S.n_blocks; ++b) if ((S.free_mask >> b) & 1u) { if (t < 6) s_row[nf + t] = 6 * b + t; nf += 6; }
    double *A = sm, *rhs = sm + size_t(nf) * nf, *x = rhs + nf;
    __syncthreads();
    bool solved = false;
    int used_reg = 0;
    for (int attempt = 0; attempt < 2 && !solved; ++attempt) {
        for (int q = t; q < nf * nf; q += 256) {
            const int i = q / nf, j = q % nf;
            A[q] = S.ne[size_t(s_row[i]) * D + s_row[j]] + ((i == j && attempt) ? 1e-6 : 0.0);
        }
        for (int i = t; i < nf; i += 256) rhs[i] = -S.ne[size_t(D) * D + s_row[i]];
        if (t == 0) s_ok = 1;
        __syncthreads();
        for (int k = 0; k < nf; ++k) {                    // right-looking Cholesky, lower triangle in place
            if (t == 0) { const double d = A[k * nf + k]; if (!(d > 0.0)) s_ok = 0; A[k * nf + k] = sqrt(d > 0.0 ? d : 1.0); }
            __syncthreads();
            const double piv = A[k * nf + k];
            for (int i = k + 1 + t; i < nf; i += 256) A[i * nf + k] /= piv;
            __syncthreads();
            const int m = nf - k - 1;                     // trailing block: (i, j), k < j <= i < nf
            for (int q = t; q < m * m; q += 256) {
                const int i = k + 1 + q / m, j = k + 1 + q % m;
                if (j <= i) A[i * nf + j] -= A[i * nf + k] * A[j * nf + k];
            }
            __syncthreads();
        }
        if (s_ok) {
            for (int k = 0; k < nf; ++k) {                // L y = rhs
                if (t == 0) rhs[k] /= A[k * nf + k];
                __syncthreads();
                const double yk = rhs[k];
                for (int i = k + 1 + t; i < nf; i += 256) rhs[i] -= A[i * nf + k] * yk;
                __syncthreads();
            }
            for (int k = nf - 1; k >= 0; --k) {           // L^T x = y
                if (t == 0) x[k] = rhs[k] / A[k * nf + k];
                __syncthreads();
                const double xk = x[k];
                for (int i = t; i < k; i += 256) rhs[i] -= A[k * nf + i] * xk;
                __syncthreads();
            }
            solved = true;
            used_reg = attempt;
        }
        __syncthreads();
    }
    if (solved && t < S.n_blocks && ((S.free_mask >> t) & 1u)) {
        int fr = 0;
        for (int b = 0; b < t; ++b) if ((S.free_mask >> b) & 1u) fr += 6;
        double d[6], out[7];
#pragma unroll
        for (int c = 0; c < 6; ++c) d[c] = x[fr + c];
        double *pose = S.poses + 7 * t;
        pose_plus(pose, d, S.V ? S.V + 36 * t : nullptr, out);
#pragma unroll
        for (int c = 0; c < 7; ++c) pose[c] = out[c];
    }
    if (t == 0) { S.status[0] = solved ? max(S.status[0], used_reg) : 2; S.status[1] += 1; }
}

int pure_odom_gn_solve(mlh_ctx *ctx, const double pivot[7], double *frames, int n_frames, double *exts, int n_ext, double huber_delta, int n_iters,
                       uint32_t const_block_mask, const double *V_update, double *cost, int32_t *n_res, int32_t *status_out)
{
    if (n_iters <= 0 || !frames || !exts) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    OdomSet &O = ctx->odom;
    OdomNeArgs G;
    size_t n_out = 0;
    int rc = odom_ne_prepare(ctx, pivot, frames, n_frames, exts, n_ext, G, n_out);
    if (rc) return rc;
    const int nb = 1 + n_frames + n_ext, D = 6 * nb;
    if (nb > 22) return fail(ctx, MLH_ERR_UNSUPPORTED, "at most 22 parameter blocks");
    const unsigned free_mask = (~const_block_mask) & ((nb >= 32) ? 0xffffffffu : ((1u << nb) - 1u));
    int nf = 0;
    for (int b = 0; b < nb; ++b) if ((free_mask >> b) & 1u) nf += 6;
    if (nf == 0) return fail(ctx, MLH_ERR_INVALID, "every block is held constant");
    hipStream_t st = ctx->stream;
    // [V_update (36 nb doubles) | status (2 ints)] behind the poses' buffer would alias a grow: own small buffer
    const size_t v_bytes = V_update ? sizeof(double) * 36 * size_t(nb) : 0;
    MLH_HIP(ctx, O.solve_aux.ensure(v_bytes + 64));
    int *d_status = reinterpret_cast<int *>(O.solve_aux.as<char>() + v_bytes);
    if (V_update) MLH_HIP(ctx, hipMemcpyAsync(O.solve_aux.p, V_update, v_bytes, hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipMemsetAsync(d_status, 0, 2 * sizeof(int), st));
    const size_t lds = sizeof(double) * (size_t(nf) * nf + 2 * size_t(nf));
    MLH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(odom_window_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    G.huber_delta = huber_delta;
    OdomSolveArgs S;
    S.ne = O.ne_out.as<double>(); S.poses = O.poses.as<double>(); S.V = V_update ? O.solve_aux.as<double>() : nullptr; S.n_blocks = nb; S.free_mask = free_mask; S.status = d_status;
    for (int it = 0; it < n_iters; ++it) {
        odom_ne_enqueue(ctx, G, n_frames, n_ext, n_out);
        MLH_LAUNCH(odom_window_solve_kernel, dim3(1), dim3(256), lds, st, S);
    }
    MLH_HIP(ctx, hipGetLastError());
    std::vector<double> hp(7 * size_t(nb)), hne(2);
    int hs[2] = {0, 0};
    MLH_HIP(ctx, hipMemcpyAsync(hp.data(), O.poses.p, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipMemcpyAsync(hne.data(), O.ne_out.as<double>() + size_t(D) * D + D, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipMemcpyAsync(hs, d_status, sizeof(hs), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    std::memcpy(frames, hp.data() + 7, sizeof(double) * 7 * size_t(n_frames));
    std::memcpy(exts, hp.data() + 7 + 7 * size_t(n_frames), sizeof(double) * 7 * size_t(n_ext));
    if (cost) *cost = hne[0];                              // of the last linearisation point (the poses before the final update)
    if (n_res) *n_res = int(hne[1] + 0.5);
    if (status_out) *status_out = hs[0];
    return MLH_OK;
}

}  // namespace mlh
