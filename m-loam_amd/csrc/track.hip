// Scan-to-scan odometry on gfx950 -- the caller side of the hot path:
//   FeatureExtract::matchCornerFromScan / matchSurfFromScan   estimator/src/featureExtract/feature_extract.hpp:132-376
//   TransformToStart (no distortion: s = 1)                   estimator/src/utility/utility.h:55-77
//   LidarScanPlaneNormFactor / LidarScanEdgeFactorVector      estimator/src/factor/lidar_scan_factor.hpp:24-64, 236-279
//   LidarTracker::trackCloud                                  estimator/src/lidarTracker/lidar_tracker.cpp:23-129
//
// track_match_kernel      16 lanes per current-frame feature: transform with the pose estimate, exact 1-NN in the previous frame's
//                         cloud through the same dense cell grid as the mapper (cell edge = 1.001 * sqrt(DISTANCE_SQ_THRESHOLD), so the
//                         27-cell search is exact for every accepted neighbour), then the reference's two directional walks over
//                         the ring-ordered array. Because the cloud is ordered by ring, each walk is a contiguous index range
//                         (ring_start table); the lanes stride over it and keep 64-bit (distance bits, walk position) keys, whose
//                         minimum is exactly the element the sequential walk with its strict `<` would have kept.
// track_linearize_kernel  one lane per feature: scan-plane (1 residual) or scan-edge-vector (3 residuals) rows at SolverState::x or
//                         ::cand, Huber on the block's squared norm (Ceres corrector, rho'' <= 0 branch), packed normal equations
//                         through the shared workgroup reduction -> partial records consumed by the LM kernels of solver.hip.
// Small launches (a frame has ~2 k sharp/flat features): latency-bound by construction; nothing here is tuned beyond that.
#include "ctx.hpp"
#include "dev_math.hpp"
#include "knn_dev.hpp"
#include "solver_dev.hpp"
#include "reduce_dev.hpp"

namespace mlh {

#ifdef MLH_STAGE_CLOCK
__device__ unsigned long long g_stage_clk_track[4096 * 4];
#define MLH_TSTAGE(i)                                                                        \
    do {                                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stage_clk_track[blockIdx.x * 4 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MLH_TSTAGE(i) do { } while (0)
#endif

constexpr int TRK_G = 16;                 // lanes per feature in the nearest-neighbour search (DPP rows)
constexpr int TRK_W = 64;                 // lanes per feature in the scan-line walks: one wavefront per feature. A frame has only a few
                                          // thousand tracked features; with 16 lanes each half the chip's SIMDs would idle while every
                                          // group makes ~30 dependent trips over its 2-3 scan lines
constexpr int TRK_FPB = TPB / TRK_W;      // features per workgroup
constexpr int TRK_ROWS = TPB / TRK_G;     // 16-lane rows per workgroup (the 4 rows of a wavefront search the same feature redundantly)
constexpr unsigned BACKWARD_BIT = 0x40000000u;

struct TrackKind {
    GridDev grid;            // previous frame's cloud, indexed
    const float4 *walk;      // the same cloud in its original (ring-ordered) order: {x, y, z, ring id as float}
    const int *ring;         // ring id of every previous-frame point
    const int *ring_start;   // ring_start[r] = first index whose ring id is >= r   (0 .. max_ring + 1 valid, beyond: n)
    int n_ring_slots;        // entries of ring_start
    const float4 *cur;       // current frame's features {x, y, z, intensity}
    Corr *corr;
    int m, tiles_a, tiles_b;
};

struct TrackParamsDev {
    TrackKind k[2];          // [MLH_SURF], [MLH_CORNER]
    SolverState *state;
    double *partials;
    int pose_sel;            // 0: state->x, 1: state->cand
    int finish, lm_max_it, lm_min_blocks;   // fused Levenberg-Marquardt begin (3) / step (4) in the linearisation kernel's last workgroup
    unsigned *ticket;
    IterStatDev *stat;
    HostPublish *publish;    // this launch publishes pose + done flag to pinned host memory
    unsigned long long publish_seq;
    int use_init;
    double init_pose[7];
    float dist_sq_thr;
    float cell_h;            // cell edge of the previous-frame indices (a quarter of the acceptance radius)
    int shells;              // cube half-width (in cells) that covers the acceptance radius
    int nearby_floor;        // floor(NEARBY_SCAN): rings id - nf .. id + nf take part in the walks
    double huber_delta;
    unsigned long long loop_timeout_ticks;   // track_lm_loop_kernel: a barrier wait longer than this (100 MHz wall clock) gives the loop up (mlh_ctx::caps)
};

__device__ __forceinline__ void track_pose(const TrackParamsDev &P, q4 &q, d3 &t)
{
    if (P.use_init) {
        t = d3{P.init_pose[0], P.init_pose[1], P.init_pose[2]};
        q = q4{P.init_pose[3], P.init_pose[4], P.init_pose[5], P.init_pose[6]};
    } else {
        const double *x = P.pose_sel ? P.state->cand : P.state->x;
        t = d3{x[0], x[1], x[2]};
        q = q4{x[3], x[4], x[5], x[6]};
    }
}

__device__ __forceinline__ unsigned long long group_min16(unsigned long long m)
{
    m = dpp_min_u64<DPP_QUAD_SWAP1>(m);
    m = dpp_min_u64<DPP_QUAD_SWAP2>(m);
    m = dpp_min_u64<DPP_ROW_HALF_MIRROR>(m);
    m = dpp_min_u64<DPP_ROW_MIRROR>(m);
    return m;
}

__device__ __forceinline__ unsigned long long wave_min64(unsigned long long m)
{
    m = group_min16(m);
    unsigned long long o = shfl_xor_u64(m, 16);
    m = o < m ? o : m;
    o = shfl_xor_u64(m, 32);
    return o < m ? o : m;
}

// walk position -> array index
__device__ __forceinline__ int walk_index(unsigned rank, int closest)
{
    return (rank & BACKWARD_BIT) ? closest - 1 - int(rank & ~BACKWARD_BIT) : closest + 1 + int(rank);
}

// Exact nearest neighbour within sqrt(thr) by a group of 16 lanes. The index cells are a quarter of the acceptance radius, so the
// usual case (the neighbour of a tracked feature is decimetres away) costs one 27-cell search; a result closer than one cell edge is
// provably the global one. Otherwise the search widens cube by cube (half-width s cells covers every point within s cell edges)
// until the best candidate is inside the covered radius or the cube covers the acceptance radius.
__device__ __forceinline__ unsigned long long nearest_in_radius(const GridDev &g, float h, int shells, float qx, float qy, float qz, int gl, int *lds_run)
{
    unsigned long long nn[1];
    knn_group<1, TRK_G>(g, qx, qy, qz, gl, lds_run, nn);
    unsigned long long best = nn[0];
    if (best != KEY_INF && __uint_as_float(unsigned(best >> 32)) < h * h) return best;     // uniform over the group
    const float fx = floorf((qx - g.ox) * g.inv_h), fy = floorf((qy - g.oy) * g.inv_h), fz = floorf((qz - g.oz) * g.inv_h);
    const float lim = float(shells + 1);
    const int cx = int(fminf(fmaxf(fx, -lim), float(g.nx) + lim)), cy = int(fminf(fmaxf(fy, -lim), float(g.ny) + lim)),
              cz = int(fminf(fmaxf(fz, -lim), float(g.nz) + lim));
    for (int s = 2; s <= shells; ++s) {
        const int x0 = max(cx - s, 0), x1 = min(cx + s, g.nx - 1);
        const int side = 2 * s + 1, n_rows = side * side;
        unsigned long long mine = KEY_INF;
        // 16 x-rows of the cube at a time, exactly as knn_group treats its 9: bounds by lane, prefix scan, run table in LDS, lanes
        // striding over the flat concatenation with 4 loads in flight
        for (int chunk = 0; chunk < n_rows; chunk += TRK_G) {
            const int r = chunk + gl;
            int b = 0, e = 0;
            if (r < n_rows && x0 <= x1) {
                const int y = cy + (r % side) - s, z = cz + (r / side) - s;
                if (y >= 0 && y < g.ny && z >= 0 && z < g.nz) {
                    const int row = (z * g.ny + y) * g.nx;
                    b = g.cell_start[row + x0];
                    e = g.cell_start[row + x1 + 1];
                }
            }
            const int len = e - b;
            int incl = len;
            { const int t = dpp_row_shr<1>(incl); if (gl >= 1) incl += t; }
            { const int t = dpp_row_shr<2>(incl); if (gl >= 2) incl += t; }
            { const int t = dpp_row_shr<4>(incl); if (gl >= 4) incl += t; }
            { const int t = dpp_row_shr<8>(incl); if (gl >= 8) incl += t; }
            const int total = __shfl(incl, TRK_G - 1, TRK_G);
            if (total > 0) {                                  // uniform over the group
                __builtin_amdgcn_wave_barrier();
                lds_run[gl] = incl - len;                     // prefix[r]
                lds_run[17 + gl] = b;                         // base[r]
                if (gl == 0) lds_run[16] = total;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                int cr = 0, hi = lds_run[1], base = lds_run[17], lo = 0;
                for (int j = gl; j < total; j += TRK_G * 4) {
                    int addr[4];
                    bool v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int jj = j + TRK_G * u;
                        v[u] = jj < total;
                        addr[u] = 0;
                        if (v[u]) {
                            while (jj >= hi) { ++cr; lo = hi; hi = lds_run[cr + 1]; base = lds_run[17 + cr]; }
                            addr[u] = base + (jj - lo);
                        }
                    }
                    float4 p[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) p[u] = g.sorted[addr[u]];      // unconditional: all four in flight together
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (v[u]) {
                            const float dx = p[u].x - qx, dy2 = p[u].y - qy, dz2 = p[u].z - qz;
                            float d = dx * dx; d += dy2 * dy2; d += dz2 * dz2;
                            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w);
                            mine = key < mine ? key : mine;
                        }
                    }
                }
            }
        }
        best = group_min16(mine);
        const float r = float(s) * h;
        if (best != KEY_INF && __uint_as_float(unsigned(best >> 32)) < r * r) break;
    }
    return best;
}

__global__ __launch_bounds__(TPB) void track_match_kernel(TrackParamsDev P)
{
    __shared__ int s_run[TRK_ROWS * 36];
    const int total = P.k[0].tiles_a + P.k[1].tiles_a;
    int tile = blockIdx.x;
    if (tile >= total) return;
    const int kind = tile >= P.k[0].tiles_a ? 1 : 0;
    if (kind) tile -= P.k[0].tiles_a;
    const TrackKind &K = P.k[kind];
    const int grp = threadIdx.x / TRK_G, gl = threadIdx.x % TRK_G;     // nearest neighbour: row of 16 lanes, each row of the wavefront the same query
    const int wl = threadIdx.x % TRK_W;                                // walks: the whole wavefront
    const int f = tile * TRK_FPB + threadIdx.x / TRK_W;
    MLH_TSTAGE(0);
    if (f >= K.m) return;
    q4 q;
    d3 t;
    track_pose(P, q, t);
    const float4 fp = K.cur[f];
    // TransformToStart without distortion: s = 1, slerp(1, q) = +-q (same rotation), f64 math, f32 store
    const d3 r = qrot(q, d3{double(fp.x), double(fp.y), double(fp.z)});
    const float sx = float(r.x + t.x), sy = float(r.y + t.y), sz = float(r.z + t.z);
    unsigned long long nn[1];
    nn[0] = nearest_in_radius(K.grid, P.cell_h, P.shells, sx, sy, sz, gl, s_run + grp * 36);
    MLH_TSTAGE(1);
    bool valid = false;
    float c6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float d1 = __uint_as_float(unsigned(nn[0] >> 32));
    if (nn[0] != KEY_INF && d1 < P.dist_sq_thr) {                 // uniform over the group
        const int closest = int(unsigned(nn[0]));
        const int id = K.ring[closest];
        const int r_hi = min(id + P.nearby_floor + 1, K.n_ring_slots - 1), r_lo = max(id - P.nearby_floor, 0);
        const int fwd_end = K.ring_start[r_hi], bwd_begin = K.ring_start[r_lo];
        const float4 *pts = K.walk;
        const unsigned long long thr_key = (unsigned long long)__float_as_uint(P.dist_sq_thr) << 32;   // strict `<` against the threshold
        unsigned long long k2 = ~0ull, k3 = ~0ull;
        constexpr int WU = 8;                                      // loads in flight per lane: the walks are latency-bound otherwise
        // increasing index: closest+1 .. fwd_end-1
        for (int j0 = closest + 1 + wl; j0 < fwd_end; j0 += TRK_W * WU) {
            float4 p[WU];
#pragma unroll
            for (int u = 0; u < WU; ++u) { const int j = j0 + TRK_W * u; if (j < fwd_end) p[u] = pts[j]; }
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                const int j = j0 + TRK_W * u;
                if (j < fwd_end) {
                    const int rj = int(p[u].w);
                    const float dd = (p[u].x - sx) * (p[u].x - sx) + (p[u].y - sy) * (p[u].y - sy) + (p[u].z - sz) * (p[u].z - sz);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | unsigned(j - closest - 1);
                    if (key < thr_key) {
                        if (kind == MLH_CORNER) { if (rj > id) k2 = key < k2 ? key : k2; }
                        else if (rj <= id) k2 = key < k2 ? key : k2;
                        else k3 = key < k3 ? key : k3;
                    }
                }
            }
        }
        // decreasing index: closest-1 .. bwd_begin
        for (int j0 = closest - 1 - wl; j0 >= bwd_begin; j0 -= TRK_W * WU) {
            float4 p[WU];
#pragma unroll
            for (int u = 0; u < WU; ++u) { const int j = j0 - TRK_W * u; if (j >= bwd_begin) p[u] = pts[j]; }
#pragma unroll
            for (int u = 0; u < WU; ++u) {
                const int j = j0 - TRK_W * u;
                if (j >= bwd_begin) {
                    const int rj = int(p[u].w);
                    const float dd = (p[u].x - sx) * (p[u].x - sx) + (p[u].y - sy) * (p[u].y - sy) + (p[u].z - sz) * (p[u].z - sz);
                    const unsigned long long key = ((unsigned long long)__float_as_uint(dd) << 32) | (BACKWARD_BIT | unsigned(closest - 1 - j));
                    if (key < thr_key) {
                        if (kind == MLH_CORNER) { if (rj < id) k2 = key < k2 ? key : k2; }
                        else if (rj >= id) k2 = key < k2 ? key : k2;
                        else k3 = key < k3 ? key : k3;
                    }
                }
            }
        }
        k2 = wave_min64(k2);
        k3 = wave_min64(k3);
        MLH_TSTAGE(2);
        if (kind == MLH_CORNER) {
            if (k2 != ~0ull) {
                const float4 a = pts[closest], b = pts[walk_index(unsigned(k2), closest)];
                c6[0] = a.x; c6[1] = a.y; c6[2] = a.z; c6[3] = b.x; c6[4] = b.y; c6[5] = b.z;
                valid = true;
            }
        } else if (k2 != ~0ull && k3 != ~0ull) {
            const float4 pj = pts[closest], pl = pts[walk_index(unsigned(k2), closest)], pm = pts[walk_index(unsigned(k3), closest)];
            const float ax = pj.x - pl.x, ay = pj.y - pl.y, az = pj.z - pl.z, bx = pj.x - pm.x, by = pj.y - pm.y, bz = pj.z - pm.z;
            float wx = ay * bz - az * by, wy = az * bx - ax * bz, wz = ax * by - ay * bx;
            const float z = wx * wx + wy * wy + wz * wz;       // Eigen normalize(): only when the squared norm is positive
            if (z > 0.f) { const float nrm = sqrtf(z); wx /= nrm; wy /= nrm; wz /= nrm; }
            c6[0] = wx; c6[1] = wy; c6[2] = wz;
            c6[3] = -(wx * pj.x + wy * pj.y + wz * pj.z);
            valid = true;
        }
    }
    if (wl == 0) {
        Corr c;
#pragma unroll
        for (int i = 0; i < 6; ++i) c.c[i] = valid ? c6[i] : 0.f;
        c.valid = valid ? 1 : 0;
        c.pad = 0;
        K.corr[f] = c;
    }
    MLH_TSTAGE(3);
}

#ifdef MLH_STAGE_CLOCK
}  // namespace mlh
extern "C" int mlh_debug_stage_clock_track(unsigned long long *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk_track), sizeof(unsigned long long) * size_t(n_words));
}
namespace mlh {
#endif

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 rowmul3(const V3 &a, const double (&M)[9])      // a^T M
{
    return {a.x * M[0] + a.y * M[3] + a.z * M[6], a.x * M[1] + a.y * M[4] + a.z * M[7], a.x * M[2] + a.y * M[5] + a.z * M[8]};
}
__device__ __forceinline__ V3 row_skew3(const V3 &a, const V3 &v)             // a^T [v]x
{
    return {a.y * v.z - a.z * v.y, a.z * v.x - a.x * v.z, a.x * v.y - a.y * v.x};
}

__device__ __forceinline__ void track_publish(const TrackParamsDev &P)
{
    for (int i = 0; i < 7; ++i) P.publish->x[i] = P.state->x[i];
    P.publish->done = P.state->done;
    __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the normal-equation terms of ONE matched feature at pose (q, t): LidarScanPlaneNormFactor (1 residual) or LidarScanEdgeFactorVector (3), Huber on the block's
// squared norm -- what track_linearize_kernel and track_lm_loop_kernel accumulate (acc: zeroed by the caller)
__device__ __forceinline__ void track_eval(const TrackParamsDev &P, int kind, const Corr &c, const float4 &fp, const q4 &q, const d3 &t, double (&acc)[32])
{
    const V3 p{double(fp.x), double(fp.y), double(fp.z)};
    double R[9];
    qtorot(q, R);
    const d3 rp = qrot(q, d3{p.x, p.y, p.z});
    const V3 lp{rp.x + t.x, rp.y + t.y, rp.z + t.z};
    double res[3], J[3][6];
    int rows;
    if (kind == MLH_SURF) {
        rows = 1;
        const V3 w{double(c.c[0]), double(c.c[1]), double(c.c[2])};
        res[0] = (w.x * lp.x + w.y * lp.y + w.z * lp.z) + double(c.c[3]);
        const V3 jr = row_skew3(rowmul3(w, R), p);
        J[0][0] = w.x; J[0][1] = w.y; J[0][2] = w.z; J[0][3] = -jr.x; J[0][4] = -jr.y; J[0][5] = -jr.z;
    } else {
        rows = 3;
        const V3 la{double(c.c[0]), double(c.c[1]), double(c.c[2])}, lb{double(c.c[3]), double(c.c[4]), double(c.c[5])};
        const V3 ba{lp.x - la.x, lp.y - la.y, lp.z - la.z}, bb{lp.x - lb.x, lp.y - lb.y, lp.z - lb.z};
        const V3 nu{ba.y * bb.z - ba.z * bb.y, ba.z * bb.x - ba.x * bb.z, ba.x * bb.y - ba.y * bb.x};
        const V3 de{la.x - lb.x, la.y - lb.y, la.z - lb.z};
        const double den = sqrt(de.x * de.x + de.y * de.y + de.z * de.z);
        res[0] = nu.x / den; res[1] = nu.y / den; res[2] = nu.z / den;
        const double eta = 1.0 / den;
        // rows of [de]x
        const V3 sd[3] = {{0.0, -de.z, de.y}, {de.z, 0.0, -de.x}, {-de.y, de.x, 0.0}};
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const V3 jr = row_skew3(rowmul3(sd[r], R), p);
            J[r][0] = -eta * sd[r].x; J[r][1] = -eta * sd[r].y; J[r][2] = -eta * sd[r].z;
            J[r][3] = eta * jr.x; J[r][4] = eta * jr.y; J[r][5] = eta * jr.z;
        }
    }
    double sq = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r) if (r < rows) sq += res[r] * res[r];     // static indices: res / J stay in registers (a loop over `rows` sends them to scratch)
    double rho0 = sq, rho1 = 1.0;
    if (P.huber_delta > 0.0) {
        const double b = P.huber_delta * P.huber_delta;
        if (sq > b) {
            const double rr = sqrt(sq);
            rho0 = 2.0 * P.huber_delta * rr - b;
            rho1 = fmax(DBL_MIN, P.huber_delta / rr);
        }
    }
    const double sc = sqrt(rho1);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r >= rows) break;
        double Jr[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) Jr[i] = J[r][i] * sc;
        const double rr = res[r] * sc;
        int qd = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) acc[qd++] += Jr[i] * Jr[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[NE_G + i] += Jr[i] * rr;
    }
    acc[NE_COST] = 0.5 * rho0;
    acc[NE_CNT] = 1.0;
}

__global__ __launch_bounds__(TPB) void track_linearize_kernel(TrackParamsDev P)
{
    __shared__ double s_red[4 * 32];
    const int total = P.k[0].tiles_b + P.k[1].tiles_b;
    const int gtile = blockIdx.x;
    if (gtile >= total) return;
    if (P.pose_sel && P.state->done) {       // the LM loop has terminated: keep the partials defined, do no work
        if (threadIdx.x < 32) P.partials[size_t(gtile) * NE_STRIDE + threadIdx.x] = 0.0;
        if (P.publish && gtile == 0 && threadIdx.x == 0) track_publish(P);      // the host may be waiting for this launch's record
        return;
    }
    const int kind = gtile >= P.k[0].tiles_b ? 1 : 0;
    const int tile = kind ? gtile - P.k[0].tiles_b : gtile;
    const TrackKind &K = P.k[kind];
    const int f = tile * TPB + threadIdx.x;
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 0.0;
    if (f < K.m) {
        const Corr c = K.corr[f];
        if (c.valid) {
            q4 q;
            d3 t;
            track_pose(P, q, t);
            const float4 fp = K.cur[f];
            track_eval(P, kind, c, fp, q, t, acc);
        }
    }
    reduce_acc32(acc, kind, s_red, P.partials + size_t(gtile) * NE_STRIDE);
    if (!P.finish) return;
    // fused tail (as match.hip: fused_gn_finish): the last workgroup to arrive sums the partial records in fixed order and runs the
    // Levenberg-Marquardt begin / step -- an LM iteration of the tracker is ONE launch (lidar_tracker.cpp:66-70, 106-113: no degeneracy
    // handling, rounds with fewer than 10 correspondences are skipped)
    __shared__ int s_last;
    __shared__ double f_ne[NE_STRIDE], f_cnt2[2], f_scratch[8 * 32];
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = (atomicAdd(P.ticket, 1u) == unsigned(total - 1)) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    SumArgs sa;
    sa.p = P.partials;
    sa.lo[0] = 0; sa.hi[0] = total; sa.lo[1] = 0; sa.hi[1] = 0;
    if (P.use_init && threadIdx.x < 7) P.state->x[threadIdx.x] = P.init_pose[threadIdx.x];   // the state's pose is born here (first round of a solve); ordered
    sum_partials(sa, f_ne, f_cnt2, f_scratch);                                               // before the body by the barriers of the sum
    if (threadIdx.x < 64) {          // one wavefront runs the LM begin / step (solver_dev.hpp: rows of the 6 x 6 objects on lanes)
        double xo[7];
        int done = 0;
        if (P.finish == 3) lm_begin_body_wave(f_ne, f_cnt2, f_scratch, P.state, -1.0, P.lm_max_it, P.stat, P.lm_min_blocks, xo, done);
        else lm_step_body_wave(f_ne, P.state, P.lm_max_it, xo, done);
        if (threadIdx.x == 0) {
            *P.ticket = 0u;
            if (P.publish) {
                for (int i = 0; i < 7; ++i) P.publish->x[i] = xo[i];
                P.publish->done = done;
                __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// ---- trackCloud's Levenberg-Marquardt loop of one round in ONE launch (the mapper's lm_loop_kernel, match.hip, for the tracker's factors): every tile's workgroup
// keeps its correspondences and features in registers, evaluates at the round's pose, and then -- records through agent-scope stores / loads around a counter
// barrier, the 6 x 6 arithmetic redundantly in every workgroup, the state in LDS -- runs begin, evaluate at the candidate, step, ... until the loop ends where Ceres'
// would (<= 4 iterations, lidar_tracker.cpp:106-113; fewer than 10 correspondences: the round is skipped, :66-70). Same operations as the launches it replaces
// (track_linearize_kernel's begin / step finishes): the same pose bits. Two launches per round instead of 2 + max_lm_iterations.
__global__ __launch_bounds__(TPB) void track_lm_loop_kernel(TrackParamsDev P)
{
    __shared__ double s_red[4 * 32];
    __shared__ double f_ne[NE_STRIDE], f_scratch[(TPB / 32) * 32];
    __shared__ double s_cand[8];
    __shared__ int s_done, s_timeout;
    __shared__ LmState s_lm;
    const int total = P.k[0].tiles_b + P.k[1].tiles_b;
    const int gtile = blockIdx.x;
    if (gtile >= total) return;
    const int kind = gtile >= P.k[0].tiles_b ? 1 : 0;
    const int tile = kind ? gtile - P.k[0].tiles_b : gtile;
    const TrackKind &K = P.k[kind];
    const int f = tile * TPB + threadIdx.x;
    Corr c;
    c.valid = 0;
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < K.m) { c = K.corr[f]; fp = K.cur[f]; }
    const bool valid = f < K.m && c.valid != 0;
    const size_t set = size_t(NE_STRIDE) * size_t(total);
    // given up already -- by a workgroup of this launch that waited in vain, or by an earlier round of this call (lm_overflow == 4; the first round, whose pose
    // comes with the kernel arguments, clears it): nothing to do but leave; the failure travels on to the round that publishes
    if (threadIdx.x == 0) s_timeout = (loop_barrier_given_up(P.ticket) || (!P.use_init && P.state->lm_overflow == 4)) ? 2 : 0;
    // the round's pose: the records of the begin are taken there
    if (threadIdx.x < 7) s_cand[threadIdx.x] = P.use_init ? P.init_pose[threadIdx.x] : P.state->x[threadIdx.x];
    __syncthreads();
    int nth = 0;                                                   // barriers passed
    bool begun = false;
    while (!s_timeout) {
        const q4 q{s_cand[3], s_cand[4], s_cand[5], s_cand[6]};
        const d3 t{s_cand[0], s_cand[1], s_cand[2]};
        double acc[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = 0.0;
        if (valid) track_eval(P, kind, c, fp, q, t, acc);
        double *rec = P.partials + set * size_t(nth & 1);
        reduce_acc32<true>(acc, kind, s_red, rec + size_t(gtile) * NE_STRIDE);
        if (threadIdx.x == 0 && !loop_barrier_arrive(P.ticket, total, nth + 1, P.loop_timeout_ticks)) s_timeout = 1;
        __syncthreads();
        ++nth;
        if (s_timeout) break;
        lmc_sum_records<true>(rec, total, f_ne, f_scratch);
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            LmRegs R;
            double cand[7];
            if (!begun) {
                double x[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) x[i] = s_cand[i];
                lm_begin_wave_pp(f_ne, f_scratch, x, &s_lm, true, -1.0, P.lm_max_it, P.lm_min_blocks, R, cand);
            } else {
                lm_step_wave_pp(f_ne, &s_lm, &s_lm, true, P.lm_max_it, R, cand);
            }
            __builtin_amdgcn_wave_barrier();                       // (x was read from s_cand by every lane before any lane overwrites it)
            if (lane < 7) s_cand[lane] = pick7(cand, lane);
            if (lane == 0) s_done = R.done;
        }
        begun = true;
        __syncthreads();
        if (s_done) break;
    }
    if (gtile == 0 && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool have = nth > 0 && !s_timeout;                   // (a barrier given up on before the begin: the state's pose stays what it was)
        if (lane < 7 && (have || P.use_init)) P.state->x[lane] = have ? s_lm.x[lane] : P.init_pose[lane];
        if (lane == 0) {
            P.state->done = have ? s_lm.done : 1;
            P.state->iteration = have ? s_lm.iteration : 0;
            P.state->lm_overflow = s_timeout ? 4 : 0;              // (a later round of the call finds it and leaves; the round that publishes reports it)
            if (P.publish) {
                for (int i = 0; i < 7; ++i) P.publish->x[i] = have ? s_lm.x[i] : (P.use_init ? P.init_pose[i] : P.state->x[i]);
                P.publish->done = 1 | (s_timeout ? 4 : 0);
                __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (threadIdx.x == 0) loop_barrier_leave(P.ticket, total);
}

// ring ids + ring_start table of a previous-frame cloud; flags non-monotone / out-of-range ring ids
__global__ __launch_bounds__(256) void track_rings_kernel(const unsigned char *src, int stride, int n, int intensity_off, int *ring, int *ring_start,
                                                          int n_slots, int *bad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int r = int(*reinterpret_cast<const float *>(src + size_t(i) * stride + intensity_off));
    const int prev = i > 0 ? int(*reinterpret_cast<const float *>(src + size_t(i - 1) * stride + intensity_off)) : -1;
    ring[i] = r;
    // the previous point's id bounds the loop below: it is validated like this point's own (a malformed id there would start the
    // loop at a negative slot, or make it very long)
    if (r < 0 || r >= n_slots - 1 || r < prev || prev < -1 || prev >= n_slots - 1) { atomicOr(bad, 1); return; }
    for (int k = prev + 1; k <= r; ++k) ring_start[k] = i;
}

// p[0..n) = v, p[n] = 0 (the ring table and the flag behind it)
__global__ void fill_int_kernel(int *p, int n, int v)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
    else if (i == n) p[i] = 0;
}

// host_bad: where the "not ordered by ring" flag is copied to; valid after the next synchronisation of the stream (the caller checks it)
int track_set_prev_rings(mlh_ctx *ctx, int kind, const unsigned char *d_src, int stride, int n, int intensity_off, int *host_bad)
{
    TrackSet &T = ctx->track;
    MLH_HIP(ctx, T.ring[kind].ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, T.ring_start[kind].ensure(sizeof(int) * (TRACK_RING_SLOTS + 1)));
    int *bad = T.ring_start[kind].as<int>() + TRACK_RING_SLOTS;
    MLH_LAUNCH(fill_int_kernel, dim3((TRACK_RING_SLOTS + 1 + 255) / 256), dim3(256), 0, ctx->stream, T.ring_start[kind].as<int>(), TRACK_RING_SLOTS, n);   // (also clears the flag)
    MLH_LAUNCH(track_rings_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, d_src, stride, n, intensity_off, T.ring[kind].as<int>(),
                       T.ring_start[kind].as<int>(), TRACK_RING_SLOTS, bad);
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(host_bad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return MLH_OK;
}

static int fill_track_params(mlh_ctx *ctx, int kind_mask, const TrackArgs &a, TrackParamsDev &P)
{
    std::memset(&P, 0, sizeof(P));
    TrackSet &T = ctx->track;
    int tiles_b = 0;
    for (int k = 0; k < 2; ++k) {
        if (!(kind_mask & (1 << k))) continue;
        if (!T.grid[k].built || T.m[k] <= 0) return fail(ctx, MLH_ERR_STATE, "track_set_prev / track_set_cur have not been called for this kind");
        if (a.dist_sq_thr > 0.f && std::sqrt(a.dist_sq_thr) > T.grid[k].h * float(TRACK_SHELLS)) return fail(ctx, MLH_ERR_INVALID, "distance_sq_threshold exceeds the value the index was built for");
        P.cell_h = T.grid[k].h;
        TrackKind &K = P.k[k];
        K.grid = T.grid[k].dev();
        K.walk = T.walk[k].as<float4>(); K.ring = T.ring[k].as<int>(); K.ring_start = T.ring_start[k].as<int>(); K.n_ring_slots = TRACK_RING_SLOTS;
        K.cur = T.cur[k].as<float4>(); K.corr = T.corr[k].as<Corr>(); K.m = T.m[k];
        K.tiles_a = (K.m + TRK_FPB - 1) / TRK_FPB; K.tiles_b = (K.m + TPB - 1) / TPB;
        tiles_b += K.tiles_b;
    }
    if (!tiles_b) return fail(ctx, MLH_ERR_STATE, "no tracker features staged");
    hipError_t e;
    if ((e = ctx->partials.ensure(sizeof(double) * NE_STRIDE * size_t(tiles_b) * 2)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc partials", e);      // (two sets: the loop kernel alternates)
    ctx->n_partial_tiles = tiles_b;
    P.state = ctx->state.as<SolverState>(); P.partials = ctx->partials.as<double>();
    P.pose_sel = a.pose_sel; P.use_init = a.init_pose ? 1 : 0;
    for (int i = 0; i < 7; ++i) P.init_pose[i] = a.init_pose ? a.init_pose[i] : 0.0;
    P.shells = TRACK_SHELLS;
    P.dist_sq_thr = a.dist_sq_thr; P.nearby_floor = int(std::floor(a.nearby_scan)); P.huber_delta = a.huber_delta;
    P.finish = a.finish; P.lm_max_it = a.lm_max_it; P.lm_min_blocks = a.lm_min_blocks;
    if ((e = ensure_ticket(ctx)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc ticket", e);      // (all four words: a later scan2map's loop kernel uses the others)
    P.ticket = ctx->ticket.as<unsigned>();
    P.stat = a.stat_slot >= 0 ? ctx->stats.as<IterStatDev>() + a.stat_slot : nullptr;
    P.publish = a.publish; P.publish_seq = a.publish_seq;
    P.loop_timeout_ticks = ctx->caps.loop_timeout_ticks;
    return MLH_OK;
}

int track_match_launch(mlh_ctx *ctx, int kind_mask, const TrackArgs &a)
{
    TrackParamsDev P;
    int rc = fill_track_params(ctx, kind_mask, a, P);
    if (rc) return rc;
    MLH_LAUNCH(track_match_kernel, dim3(P.k[0].tiles_a + P.k[1].tiles_a), dim3(TPB), 0, ctx->stream, P);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int track_lm_loop_launch(mlh_ctx *ctx, int kind_mask, const TrackArgs &a)
{
    TrackParamsDev P;
    int rc = fill_track_params(ctx, kind_mask, a, P);
    if (rc) return rc;
    // every tile's workgroup has to be resident for the barrier: the host's gate (capi.hip: loop_tiles_ok) is what the device admits, asked at mlh_create
    if (P.k[0].tiles_b + P.k[1].tiles_b > ctx->caps.loop_max_tiles[2]) return fail(ctx, MLH_ERR_INVALID, "track_lm_loop_kernel: more tiles than can be resident at once on this device");
    ++ctx->caps.loop_launches;
    MLH_LAUNCH(track_lm_loop_kernel, dim3(P.k[0].tiles_b + P.k[1].tiles_b), dim3(TPB), 0, ctx->stream, P);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int track_loop_occupancy(int *blocks_per_cu)
{
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, track_lm_loop_kernel, TPB, 0) == hipSuccess ? MLH_OK : MLH_ERR_HIP;
}

int track_linearize_launch(mlh_ctx *ctx, int kind_mask, const TrackArgs &a)
{
    TrackParamsDev P;
    int rc = fill_track_params(ctx, kind_mask, a, P);
    if (rc) return rc;
    MLH_LAUNCH(track_linearize_kernel, dim3(P.k[0].tiles_b + P.k[1].tiles_b), dim3(TPB), 0, ctx->stream, P);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

}  // namespace mlh
