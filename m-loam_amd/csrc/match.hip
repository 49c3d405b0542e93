// Correspondence + linearisation kernels for gfx950 (the roofline kernels of the scan-to-map path).
//
// knn_features_kernel        -- per feature: pointAssociateToMap (utility.h:103-117) -> exact 5-NN in the local-map cell grid
//   (the pcl::KdTreeFLANN::nearestKSearch role, feature_extract.hpp:666/813). 32 lanes (half a wavefront) per feature:
//   lane = (x-run of the 27-cell neighbourhood, sub-lane), all 9 runs in flight at once, consecutive sub-lanes on
//   consecutive float4 points; each lane keeps a sorted top-5 of 64-bit (distance bits, map index) keys, the group merges with
//   a 5-round shuffle tournament; the 5 winners' coordinates + squared distances go to HBM (80 B/feature).
//   HBM/L2-bound: ~(16 + 72 + 16*C + 80 + 80) bytes per feature, C = candidates in the 27 cells.
// fit_linearize_kernel<KIND> -- one lane per feature: line fit (3x3 scatter + f32 eigen-solver, hpp:669-783) or plane fit
//   (5x3 column-pivoted QR, hpp:816-878) with the reference's gates -> residual and 1x6 Jacobian of LidarMapEdgeFactor /
//   LidarMapPlaneNormFactor (lidar_map_factor.hpp:44-71, 143-174) -> Huber correction (Ceres corrector, rho''<=0 branch)
//   -> wavefront-shuffle + LDS reduction of the 21+6+2 packed normal-equation sums -> one partial record per workgroup
//   (no atomics, deterministic).
// linearize_kernel<KIND>     -- the same evaluation on the correspondences stored by the fit kernel (what ceres::Solve
//   does per LM iteration).
// blockIdx -> tile mapping is XCD-aware: consecutive tiles go to the same XCD (blockIdx % 8), so each XCD's L2 holds one
// contiguous eighth of the (spatially coherent) feature list's map neighbourhood.
#include "ctx.hpp"
#include "dev_math.hpp"
#include <cfloat>

namespace mlh {

constexpr int TPB = 256;
constexpr unsigned long long KEY_INF = 0x7f800000ffffffffull;   // (+inf, max index)

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, mask);
    hi = __shfl_xor(hi, mask);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ void key_insert(unsigned long long (&k)[5], unsigned long long key)
{
    if (key < k[4]) {
        k[4] = key;
#pragma unroll
        for (int i = 4; i > 0; --i) {
            unsigned long long a = k[i - 1], b = k[i];
            bool sw = b < a;
            k[i - 1] = sw ? b : a;
            k[i] = sw ? a : b;
        }
    }
}

__device__ __forceinline__ float clamp_cell_f(float v, float o, float inv_h, int n)
{
    float f = floorf((v - o) * inv_h);
    return fminf(fmaxf(f, -2.f), float(n + 1));   // also squashes NaN/inf before the int conversion
}

// exact 5-NN of (qx,qy,qz) by a group of 32 lanes (half a wavefront); on return every lane holds the 5 keys ascending.
// Lane gl < 27 works on x-run r = gl / 3 of the 27-cell neighbourhood (the 3 x-adjacent cells of one (dy,dz) are one
// contiguous range of the cell-sorted array) as sub-lane s = gl % 3: consecutive sub-lanes read consecutive float4 points.
// All 9 runs are in flight at once, so a query costs ~3 dependent memory round trips (cell_start, candidates, neighbours).
__device__ __forceinline__ void knn5_group32(const GridDev &g, float qx, float qy, float qz, int gl, unsigned long long (&out)[5])
{
    unsigned long long k[5] = {KEY_INF, KEY_INF, KEY_INF, KEY_INF, KEY_INF};
    const int cx = int(clamp_cell_f(qx, g.ox, g.inv_h, g.nx));
    const int cy = int(clamp_cell_f(qy, g.oy, g.inv_h, g.ny));
    const int cz = int(clamp_cell_f(qz, g.oz, g.inv_h, g.nz));
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    const int r = gl / 3, sl = gl - r * 3;
    const int y = cy + (r % 3) - 1, z = cz + (r / 3) - 1;
    const bool ok = (gl < 27) && (x0 <= x1) && (y >= 0) && (y < g.ny) && (z >= 0) && (z < g.nz);
    int b = 0, e = 0;
    if (ok) {
        const int row = (z * g.ny + y) * g.nx;
        b = g.cell_start[row + x0];
        e = g.cell_start[row + x1 + 1];
    }
    for (int j = b + sl; j < e; j += 6) {
        const float4 p0 = g.sorted[j];
        const bool h1 = (j + 3) < e;
        const float4 p1 = g.sorted[h1 ? j + 3 : j];
        {
            float dx = p0.x - qx, dy = p0.y - qy, dz = p0.z - qz;
            float d = dx * dx; d += dy * dy; d += dz * dz;
            key_insert(k, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p0.w));
        }
        if (h1) {
            float dx = p1.x - qx, dy = p1.y - qy, dz = p1.z - qz;
            float d = dx * dx; d += dy * dy; d += dz * dz;
            key_insert(k, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p1.w));
        }
    }
    // tournament merge: 5 rounds of group-min over the lanes' current heads
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        unsigned long long m = k[0];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            unsigned long long o = shfl_xor_u64(m, off);
            m = o < m ? o : m;
        }
        out[t] = m;
        if (k[0] == m && m != KEY_INF) { k[0] = k[1]; k[1] = k[2]; k[2] = k[3]; k[3] = k[4]; k[4] = KEY_INF; }
    }
}

// ---------------------------------------------------------------- per-feature evaluation (registers only)
struct Lin {
    double r;
    double J[6];
};

__device__ __forceinline__ double sqrt_info_of(double trace)
{
    double s = sqrt(1 / trace);
    return s >= 3.0 ? 1.0 : s / 3.0;
}

// LidarMapPlaneNormFactor::Evaluate (lidar_map_factor.hpp:44-71)
__device__ __forceinline__ void eval_plane(const d3 &p, const float (&c)[6], double w, const q4 &q, const d3 &t, const double *R, Lin &o)
{
    d3 n{double(c[0]), double(c[1]), double(c[2])};
    double d = double(c[3]);
    d3 lp = qrot(q, p);
    lp.x += t.x; lp.y += t.y; lp.z += t.z;
    double a = (n.x * lp.x + n.y * lp.y + n.z * lp.z) + d;
    o.r = w * a;
    // -n^T R [p]x
    double wr0 = (-n.x) * R[0] + (-n.y) * R[3] + (-n.z) * R[6];
    double wr1 = (-n.x) * R[1] + (-n.y) * R[4] + (-n.z) * R[7];
    double wr2 = (-n.x) * R[2] + (-n.y) * R[5] + (-n.z) * R[8];
    // [p]x = [0 -pz py; pz 0 -px; -py px 0]
    double j0 = wr0 * 0.0 + wr1 * p.z + wr2 * (-p.y);
    double j1 = wr0 * (-p.z) + wr1 * 0.0 + wr2 * p.x;
    double j2 = wr0 * p.y + wr1 * (-p.x) + wr2 * 0.0;
    o.J[0] = w * n.x; o.J[1] = w * n.y; o.J[2] = w * n.z;
    o.J[3] = w * j0; o.J[4] = w * j1; o.J[5] = w * j2;
}

// LidarMapEdgeFactor::Evaluate (lidar_map_factor.hpp:143-174)
__device__ __forceinline__ void eval_edge(const d3 &p, const float (&c)[6], double w, const q4 &q, const d3 &t, const double *R, Lin &o)
{
    d3 lpa{double(c[0]), double(c[1]), double(c[2])};
    d3 lpb{double(c[3]), double(c[4]), double(c[5])};
    d3 lp = qrot(q, p);
    lp.x += t.x; lp.y += t.y; lp.z += t.z;
    d3 a{lp.x - lpa.x, lp.y - lpa.y, lp.z - lpa.z};
    d3 b{lp.x - lpb.x, lp.y - lpb.y, lp.z - lpb.z};
    d3 nu = cross3(a, b);
    d3 de{lpa.x - lpb.x, lpa.y - lpb.y, lpa.z - lpb.z};
    double nu_n = sqrt(nu.x * nu.x + nu.y * nu.y + nu.z * nu.z);
    double de_n = sqrt(de.x * de.x + de.y * de.y + de.z * de.z);
    o.r = w * nu_n / de_n;
    double k = 1.0 / de_n;
    double n2 = nu.x * nu.x + nu.y * nu.y + nu.z * nu.z;
    double ex = nu.x, ey = nu.y, ez = nu.z;
    if (n2 > 0.0) { double nn = sqrt(n2); ex /= nn; ey /= nn; ez /= nn; }
    ex = k * ex; ey = k * ey; ez = k * ez;
    // eta * [de]x
    double eD0 = ex * 0.0 + ey * de.z + ez * (-de.y);
    double eD1 = ex * (-de.z) + ey * 0.0 + ez * de.x;
    double eD2 = ex * de.y + ey * (-de.x) + ez * 0.0;
    double eDR0 = eD0 * R[0] + eD1 * R[3] + eD2 * R[6];
    double eDR1 = eD0 * R[1] + eD1 * R[4] + eD2 * R[7];
    double eDR2 = eD0 * R[2] + eD1 * R[5] + eD2 * R[8];
    double j0 = eDR0 * 0.0 + eDR1 * p.z + eDR2 * (-p.y);
    double j1 = eDR0 * (-p.z) + eDR1 * 0.0 + eDR2 * p.x;
    double j2 = eDR0 * p.y + eDR1 * (-p.x) + eDR2 * 0.0;
    o.J[0] = w * (-eD0); o.J[1] = w * (-eD1); o.J[2] = w * (-eD2);
    o.J[3] = w * j0; o.J[4] = w * j1; o.J[5] = w * j2;
}

// accumulate one (possibly invalid) row and reduce the 29 sums over the workgroup -> partials[tile]
__device__ __forceinline__ void reduce_rows(bool valid, Lin L, double huber_delta, bool no_loss, double *lds_red /*4*32*/,
                                            double *__restrict__ partial_out)
{
    double acc[29];
    if (valid) {
        double s = L.r * L.r, rho0 = s, rho1 = 1.0;
        if (!no_loss && huber_delta > 0.0) {
            const double b = huber_delta * huber_delta;
            if (s > b) {
                const double rr = sqrt(s);
                rho0 = 2.0 * huber_delta * rr - b;
                rho1 = fmax(DBL_MIN, huber_delta / rr);
            }
        }
        const double sc = sqrt(rho1);
        double r = L.r * sc;
        double J[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) J[i] = L.J[i] * sc;
        int q = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) acc[q++] = J[i] * J[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[NE_G + i] = J[i] * r;
        acc[NE_COST] = 0.5 * rho0;
        acc[NE_CNT] = 1.0;
    } else {
#pragma unroll
        for (int i = 0; i < 29; ++i) acc[i] = 0.0;
    }
#pragma unroll
    for (int i = 0; i < 29; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc[i] = v;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 29; ++i) lds_red[wave * 32 + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double v = 0.0;
        if (threadIdx.x < 29) v = ((lds_red[threadIdx.x] + lds_red[32 + threadIdx.x]) + lds_red[64 + threadIdx.x]) + lds_red[96 + threadIdx.x];
        partial_out[threadIdx.x] = v;
    }
}

__device__ __forceinline__ int xcd_tile(int n_tiles)
{
    const int per = (n_tiles + 7) >> 3;
    return (blockIdx.x & 7) * per + (blockIdx.x >> 3);
}

// feature_extract.hpp:696-715
__device__ __forceinline__ bool in_laser_fov(const q4 &q, const d3 &t, float sx, float sy, float sz)
{
    d3 zt = qrot(q, d3{0.0, 0.0, 10.0});
    float zx = float(zt.x + t.x), zy = float(zt.y + t.y), zz = float(zt.z + t.z);
    double a0 = t.x - double(sx), a1 = t.y - double(sy), a2 = t.z - double(sz);
    float squared_side1 = float(a0 * a0 + a1 * a1 + a2 * a2);
    float b0 = zx - sx, b1 = zy - sy, b2 = zz - sz;
    float squared_side2 = b0 * b0 + b1 * b1 + b2 * b2;
    float check1 = 100.0f + squared_side1 - squared_side2 - 10.0f * sqrtf(3.0f) * sqrtf(squared_side1);
    float check2 = 100.0f + squared_side1 - squared_side2 + 10.0f * sqrtf(3.0f) * sqrtf(squared_side1);
    return check1 < 0 && check2 > 0;
}

struct KParams {
    GridDev grid;
    const float4 *feat;      // {x,y,z,intensity}
    const float4 *covd;      // {cxx,cyy,czz,_} or null
    float4 *nbr;             // 5 per feature: {x,y,z, sq-dist} of the k-th neighbour (w = +inf when missing)
    Corr *corr;
    double *r_out;           // nullable
    double *J_out;           // nullable
    double *partials;
    const SolverState *state;
    int m, n_tiles, pose_sel;
    uint32_t flags;
    float min_match_sq_dis, min_plane_dis;
    double huber_delta, cov_measurement_trace;
    int has_lo, has_hi;      // multi-GPU ownership half-spaces (mlh_shard_set)
    float lo[4], hi[4];
};

// pointAssociateToMap (utility.h:103-117): f64 q*p + t, stored to f32
__device__ __forceinline__ void associate_to_map(const q4 &q, const d3 &t, const float4 &fp, float &sx, float &sy, float &sz)
{
    d3 w = qrot(q, d3{double(fp.x), double(fp.y), double(fp.z)});
    sx = float(w.x + t.x); sy = float(w.y + t.y); sz = float(w.z + t.z);
}

__device__ __forceinline__ bool owns(const KParams &P, float sx, float sy, float sz)
{
    bool own = true;
    if (P.has_lo) own = own && ((((P.lo[0] * sx + P.lo[1] * sy) + P.lo[2] * sz) + P.lo[3]) >= 0.f);
    if (P.has_hi) own = own && ((((P.hi[0] * sx + P.hi[1] * sy) + P.hi[2] * sz) + P.hi[3]) < 0.f);
    return own;
}

// ---- correspondence kernel: 32 lanes per feature, 8 features per workgroup
constexpr int KNN_FPB = TPB / 32;

__global__ __launch_bounds__(TPB) void knn_features_kernel(KParams P)
{
    const int tile = xcd_tile(P.n_tiles);
    if (tile >= P.n_tiles) return;
    const int grp = threadIdx.x >> 5, gl = threadIdx.x & 31;
    const int f = tile * KNN_FPB + grp;
    if (f >= P.m) return;
    const double *pose = P.pose_sel ? P.state->cand : P.state->x;
    const q4 q{pose[3], pose[4], pose[5], pose[6]};
    const d3 t{pose[0], pose[1], pose[2]};
    const float4 fp = P.feat[f];
    float sx, sy, sz;
    associate_to_map(q, t, fp, sx, sy, sz);
    if (!owns(P, sx, sy, sz)) return;     // uniform over the 32-lane group
    unsigned long long keys[5];
    knn5_group32(P.grid, sx, sy, sz, gl, keys);
    if (gl < 5) {
        unsigned long long kk = keys[0];
        if (gl == 1) kk = keys[1]; else if (gl == 2) kk = keys[2]; else if (gl == 3) kk = keys[3]; else if (gl == 4) kk = keys[4];
        float4 o = make_float4(0.f, 0.f, 0.f, __uint_as_float(0x7f800000u));
        if (kk != KEY_INF) {
            const float4 np = P.grid.raw[(unsigned)kk];
            o = make_float4(np.x, np.y, np.z, __uint_as_float((unsigned)(kk >> 32)));
        }
        P.nbr[size_t(f) * 5 + gl] = o;
    }
}

// ---- fit + gates + residual/Jacobian + normal-equation reduction: one lane per feature
template <int KIND>
__global__ __launch_bounds__(TPB) void fit_linearize_kernel(KParams P)
{
    __shared__ double s_red[4 * 32];
    const int tile = xcd_tile(P.n_tiles);
    if (tile >= P.n_tiles) return;
    const int f = tile * TPB + threadIdx.x;
    const double *pose = P.pose_sel ? P.state->cand : P.state->x;
    const q4 q{pose[3], pose[4], pose[5], pose[6]};
    const d3 t{pose[0], pose[1], pose[2]};
    bool valid = false;
    Lin L;
    L.r = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) L.J[i] = 0.0;
    float coef[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < P.m) {
        fp = P.feat[f];
        float sx, sy, sz;
        associate_to_map(q, t, fp, sx, sy, sz);
        if (owns(P, sx, sy, sz)) {
            const float4 *nb = P.nbr + size_t(f) * 5;
            const float4 n4 = nb[4];
            if (n4.w < P.min_match_sq_dis) {     // sq_dis[k-1] < MIN_MATCH_SQ_DIS (hpp:667/814)
                float ax[5], ay[5], az[5];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float4 v = nb[j]; ax[j] = v.x; ay[j] = v.y; az[j] = v.z; }
                ax[4] = n4.x; ay[4] = n4.y; az[4] = n4.z;
                if (KIND == MLH_SURF) {
                    float nx, ny, nz;
                    plane_fit_qr_f<5>(ax, ay, az, nx, ny, nz);
                    float nn = sqrtf(nx * nx + ny * ny + nz * nz);
                    float negative_OA_dot_norm = 1 / nn;
                    float z = nx * nx + ny * ny + nz * nz;
                    if (z > 0.f) { float s = sqrtf(z); nx /= s; ny /= s; nz /= s; }
                    bool plane_valid = true;
#pragma unroll
                    for (int j = 0; j < 5; ++j)
                        if (fabsf(nx * ax[j] + ny * ay[j] + nz * az[j] + negative_OA_dot_norm) > P.min_plane_dis) plane_valid = false;
                    if (plane_valid) {
                        coef[0] = nx; coef[1] = ny; coef[2] = nz; coef[3] = negative_OA_dot_norm;
                        valid = true;
                    }
                } else {
                    float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
                    for (int j = 0; j < 5; ++j) { cx += ax[j]; cy += ay[j]; cz += az[j]; }
                    cx /= 5.0f; cy /= 5.0f; cz /= 5.0f;
                    float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        float t0 = ax[j] - cx, t1 = ay[j] - cy, t2 = az[j] - cz;
                        c00 += t0 * t0; c10 += t1 * t0; c11 += t1 * t1; c20 += t2 * t0; c21 += t2 * t1; c22 += t2 * t2;
                    }
                    float l0, l1, l2, vx, vy, vz;
                    eig3_largest_f(c00, c10, c11, c20, c21, c22, l0, l1, l2, vx, vy, vz);
                    if (l2 > 3 * l1) {
                        coef[0] = 0.1f * vx + cx; coef[1] = 0.1f * vy + cy; coef[2] = 0.1f * vz + cz;
                        coef[3] = -0.1f * vx + cx; coef[4] = -0.1f * vy + cy; coef[5] = -0.1f * vz + cz;
                        valid = true;
                    }
                }
                if (valid && (P.flags & MLH_FLAG_CHECK_FOV)) valid = in_laser_fov(q, t, sx, sy, sz);
            }
        }
        Corr c;
#pragma unroll
        for (int i = 0; i < 6; ++i) c.c[i] = coef[i];
        c.valid = valid ? 1 : 0;
        c.pad = 0;
        P.corr[f] = c;
    }
    if (valid) {
        double trace = P.cov_measurement_trace;
        if ((P.flags & MLH_FLAG_WITH_UA)) {
            trace = 0.0;
            if (P.covd) { float4 cd = P.covd[f]; trace = (double(cd.x) + double(cd.y)) + double(cd.z); }
        }
        const double w = sqrt_info_of(trace);
        double R[9];
        qtorot(q, R);
        const d3 p{double(fp.x), double(fp.y), double(fp.z)};
        if (KIND == MLH_SURF) eval_plane(p, coef, w, q, t, R, L);
        else eval_edge(p, coef, w, q, t, R, L);
    }
    if (P.r_out && f < P.m) {
        P.r_out[f] = L.r;
#pragma unroll
        for (int i = 0; i < 6; ++i) P.J_out[size_t(f) * 6 + i] = L.J[i];
    }
    reduce_rows(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, s_red, P.partials + size_t(tile) * NE_STRIDE);
}

template <int KIND>
__global__ __launch_bounds__(TPB) void linearize_kernel(KParams P)
{
    __shared__ double s_red[4 * 32];
    const int tile = xcd_tile(P.n_tiles);
    if (tile >= P.n_tiles) return;
    const int f = tile * TPB + threadIdx.x;
    if (P.state->done) {   // the device-side LM loop has terminated: keep the partials defined, do no work
        if (threadIdx.x < 32) P.partials[size_t(tile) * NE_STRIDE + threadIdx.x] = 0.0;
        return;
    }
    const double *pose = P.pose_sel ? P.state->cand : P.state->x;
    const q4 q{pose[3], pose[4], pose[5], pose[6]};
    const d3 t{pose[0], pose[1], pose[2]};
    bool valid = false;
    Lin L;
    L.r = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) L.J[i] = 0.0;
    if (f < P.m) {
        const Corr c = P.corr[f];
        if (c.valid) {
            valid = true;
            const float4 fp = P.feat[f];
            double trace = P.cov_measurement_trace;
            if ((P.flags & MLH_FLAG_WITH_UA)) {
                trace = 0.0;
                if (P.covd) { float4 cd = P.covd[f]; trace = (double(cd.x) + double(cd.y)) + double(cd.z); }
            }
            const double w = sqrt_info_of(trace);
            double R[9];
            qtorot(q, R);
            const d3 p{double(fp.x), double(fp.y), double(fp.z)};
            if (KIND == MLH_SURF) eval_plane(p, c.c, w, q, t, R, L);
            else eval_edge(p, c.c, w, q, t, R, L);
        }
    }
    if (P.r_out && f < P.m) {
        P.r_out[f] = L.r;
#pragma unroll
        for (int i = 0; i < 6; ++i) P.J_out[size_t(f) * 6 + i] = L.J[i];
    }
    reduce_rows(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, s_red, P.partials + size_t(tile) * NE_STRIDE);
}

// stand-alone exact 5-NN for mlh_knn (queries already in the map frame)
__global__ __launch_bounds__(TPB) void knn_queries_kernel(GridDev grid, const float *__restrict__ q, int nq, int *__restrict__ idx,
                                                          float *__restrict__ d2)
{
    const int grp = threadIdx.x >> 5, gl = threadIdx.x & 31;
    const int qi = blockIdx.x * KNN_FPB + grp;
    if (qi >= nq) return;
    unsigned long long keys[5];
    knn5_group32(grid, q[qi * 3 + 0], q[qi * 3 + 1], q[qi * 3 + 2], gl, keys);
    if (gl == 0) {
#pragma unroll
        for (int t = 0; t < 5; ++t) {
            const bool found = keys[t] != KEY_INF;
            idx[qi * 5 + t] = found ? int((unsigned)keys[t]) : -1;
            d2[qi * 5 + t] = __uint_as_float((unsigned)(keys[t] >> 32));
        }
    }
}

// ---------------------------------------------------------------- host launchers
static int fill_params(mlh_ctx *ctx, const MatchArgs &a, KParams &P)
{
    FeatSet &fs = ctx->feat[a.kind];
    MapGrid &mg = ctx->map[a.kind];
    if (!mg.built) return fail(ctx, MLH_ERR_STATE, "map_set has not been called for this kind");
    if (fs.m <= 0) return fail(ctx, MLH_ERR_STATE, "features_set has not been called for this kind");
    const int n_tiles = (fs.m + TPB - 1) / TPB;
    hipError_t e;
    if ((e = fs.corr.ensure(sizeof(Corr) * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc corr", e);
    if ((e = fs.nbr.ensure(sizeof(float4) * 5 * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc nbr", e);
    if ((e = fs.partials.ensure(sizeof(double) * NE_STRIDE * size_t(n_tiles))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc partials", e);
    if (a.dense) {
        if ((e = fs.r.ensure(sizeof(double) * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc r", e);
        if ((e = fs.J.ensure(sizeof(double) * 6 * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc J", e);
    }
    fs.n_blocks = n_tiles;
    P.grid = mg.dev();
    P.feat = fs.pts.as<float4>();
    P.covd = fs.has_cov ? fs.covd.as<float4>() : nullptr;
    P.corr = fs.corr.as<Corr>();
    P.nbr = fs.nbr.as<float4>();
    P.r_out = a.dense ? fs.r.as<double>() : nullptr;
    P.J_out = a.dense ? fs.J.as<double>() : nullptr;
    P.partials = fs.partials.as<double>();
    P.state = ctx->state.as<SolverState>();
    P.m = fs.m;
    P.n_tiles = n_tiles;
    P.pose_sel = a.pose_sel;
    P.flags = a.flags;
    P.min_match_sq_dis = a.min_match_sq_dis;
    P.min_plane_dis = a.min_plane_dis;
    P.huber_delta = a.huber_delta;
    P.cov_measurement_trace = a.cov_measurement_trace;
    P.has_lo = ctx->shard_lo ? 1 : 0;
    P.has_hi = ctx->shard_hi ? 1 : 0;
    for (int i = 0; i < 4; ++i) { P.lo[i] = ctx->lo_plane[i]; P.hi[i] = ctx->hi_plane[i]; }
    return MLH_OK;
}

int match_launch(mlh_ctx *ctx, const MatchArgs &a)
{
    KParams P;
    int rc = fill_params(ctx, a, P);
    if (rc) return rc;
    // the cell edge was derived from the acceptance radius given at map_set; a larger radius here would break exactness
    const MapGrid &mg = ctx->map[a.kind];
    if (std::sqrt(a.min_match_sq_dis) > mg.h) return fail(ctx, MLH_ERR_INVALID, "min_match_sq_dis exceeds the value the map grid was built for");
    // kernel A: correspondences (32 lanes per feature); kernel B: fit + linearise + reduce (one lane per feature)
    KParams PA = P;
    PA.n_tiles = (P.m + KNN_FPB - 1) / KNN_FPB;
    const int grid_a = ((PA.n_tiles + 7) / 8) * 8;
    const int grid_b = ((P.n_tiles + 7) / 8) * 8;
    prof_begin(ctx, a.kind == MLH_SURF ? MLH_K_KNN_SURF : MLH_K_KNN_CORNER);
    hipLaunchKernelGGL(knn_features_kernel, dim3(grid_a), dim3(TPB), 0, ctx->stream, PA);
    prof_end(ctx, a.kind == MLH_SURF ? MLH_K_KNN_SURF : MLH_K_KNN_CORNER);
    prof_begin(ctx, a.kind == MLH_SURF ? MLH_K_FIT_SURF : MLH_K_FIT_CORNER);
    if (a.kind == MLH_SURF) hipLaunchKernelGGL((fit_linearize_kernel<MLH_SURF>), dim3(grid_b), dim3(TPB), 0, ctx->stream, P);
    else hipLaunchKernelGGL((fit_linearize_kernel<MLH_CORNER>), dim3(grid_b), dim3(TPB), 0, ctx->stream, P);
    prof_end(ctx, a.kind == MLH_SURF ? MLH_K_FIT_SURF : MLH_K_FIT_CORNER);
    MLH_HIP(ctx, hipGetLastError());
    ctx->feat[a.kind].matched = true;
    return MLH_OK;
}

int linearize_launch(mlh_ctx *ctx, const MatchArgs &a)
{
    if (!ctx->feat[a.kind].matched) return fail(ctx, MLH_ERR_STATE, "linearize needs a previous match of this kind");
    KParams P;
    int rc = fill_params(ctx, a, P);
    if (rc) return rc;
    const int grid = ((P.n_tiles + 7) / 8) * 8;
    prof_begin(ctx, MLH_K_LINEARIZE);
    if (a.kind == MLH_SURF) hipLaunchKernelGGL((linearize_kernel<MLH_SURF>), dim3(grid), dim3(TPB), 0, ctx->stream, P);
    else hipLaunchKernelGGL((linearize_kernel<MLH_CORNER>), dim3(grid), dim3(TPB), 0, ctx->stream, P);
    prof_end(ctx, MLH_K_LINEARIZE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int knn_launch(mlh_ctx *ctx, int kind, const float *q_host, int nq, int32_t *idx, float *d2)
{
    MapGrid &mg = ctx->map[kind];
    if (!mg.built) return fail(ctx, MLH_ERR_STATE, "map_set has not been called for this kind");
    if (nq <= 0) return MLH_OK;
    MLH_HIP(ctx, ctx->knn_q.ensure(sizeof(float) * 3 * size_t(nq)));
    MLH_HIP(ctx, ctx->knn_idx.ensure(sizeof(int) * 5 * size_t(nq)));
    MLH_HIP(ctx, ctx->knn_d.ensure(sizeof(float) * 5 * size_t(nq)));
    MLH_HIP(ctx, hipMemcpyAsync(ctx->knn_q.p, q_host, sizeof(float) * 3 * size_t(nq), hipMemcpyHostToDevice, ctx->stream));
    const int grid = (nq + KNN_FPB - 1) / KNN_FPB;
    hipLaunchKernelGGL(knn_queries_kernel, dim3(grid), dim3(TPB), 0, ctx->stream, mg.dev(), ctx->knn_q.as<float>(), nq,
                       ctx->knn_idx.as<int>(), ctx->knn_d.as<float>());
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(idx, ctx->knn_idx.p, sizeof(int) * 5 * size_t(nq), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(d2, ctx->knn_d.p, sizeof(float) * 5 * size_t(nq), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLH_OK;
}

}  // namespace mlh
