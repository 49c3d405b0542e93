// Correspondence + linearisation kernels for gfx950 (the roofline kernels of the scan-to-map path).
//
// knn_features_kernel<G, MB> -- per feature: pointAssociateToMap (utility.h:103-117) -> exact K-NN (K = 5, or 10 for buildCalibMap's
//   non-reference LiDARs) in the local-map cell grid (the pcl::KdTreeFLANN::nearestKSearch role, feature_extract.hpp:666/813).
//   G lanes per feature -- 16 on a frame-sized launch (latency regime), 8 on a chip-filling one (throughput regime); the group
//   search itself lives in knn_dev.hpp: the 9 x-runs of the 27-cell neighbourhood are concatenated and the lanes stride over
//   the flat candidate list (balanced, coalesced 16 B/lane); queries with fewer than K candidates leave after the 18 cell_start
//   words; each lane keeps a sorted top-K of 64-bit (distance bits, map index) keys, the group merges with a K-round
//   all-reduce-min tournament; the K winners' coordinates + squared distances go to HBM (16 K B/feature).
//   HBM/L2-bound: ~(16 + 72 + 16*C + 16*K + 16*K) bytes per feature, C = candidates in the 27 cells.
// fit_linearize_kernel<KMAX> -- one lane per feature: line fit (3x3 scatter + f32 eigen-solver, hpp:669-783) or plane fit
//   (Kx3 column-pivoted QR, hpp:816-878) with the reference's gates -> residual and 1x6 Jacobian of LidarMapEdgeFactor /
//   LidarMapPlaneNormFactor (lidar_map_factor.hpp:44-71, 143-174) -> Huber correction (Ceres corrector, rho''<=0 branch)
//   -> transposed-butterfly + LDS reduction (reduce_dev.hpp) of the 21+6+2 packed normal-equation sums -> one partial record per
//   workgroup (no atomics, deterministic) -> fused_gn_finish: the last workgroup to arrive completes the Gauss-Newton iteration
//   (sum of the records, degeneracy test, 6x6 solve, Plus) and, on the last iteration, publishes the pose to pinned host memory.
// linearize_kernel           -- the same evaluation on the correspondences stored by the fit kernel (what ceres::Solve
//   does per LM iteration).
// blockIdx -> tile mapping is XCD-aware: consecutive tiles go to the same XCD (blockIdx % 8), so each XCD's L2 holds one
// contiguous eighth of the (spatially coherent) feature list's map neighbourhood.
#include "ctx.hpp"
#include "dev_math.hpp"
#include "solver_dev.hpp"
#include "p2p_dev.hpp"
#include <hip/hip_ext.h>
#include <cfloat>
#include <cstdlib>

namespace mlh {

// Debug build only (-DMLH_STAGE_CLOCK): per-workgroup stage timestamps (100 MHz wall clock) of the fit kernel, read back
// by mlh_debug_stage_clock. Never compiled into the product library.
#ifdef MLH_STAGE_CLOCK
__device__ unsigned long long g_stage_clk[4096 * 8];
#define MLH_STAGE(tile, i)                                                                   \
    do {                                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        if (threadIdx.x == 0 && (tile) < 4096) g_stage_clk[(tile) * 8 + (i)] = wall_clock64(); \
    } while (0)
__device__ unsigned long long g_stage_clk_knn[4096 * 8];
#define MLH_KSTAGE(i)                                                                        \
    do {                                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_stage_clk_knn[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MLH_STAGE(tile, i) do { } while (0)
#define MLH_KSTAGE(i) do { } while (0)
#endif
}  // namespace mlh
#include "knn_dev.hpp"
#include "reduce_dev.hpp"
namespace mlh {
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
// (mult: how many identical residual blocks the row stands for -- 1, except for the feature a selection picked repeatedly, select.hip: apply_keep_kernel)
template <int MODE = 0>      // 0: plain stores, 1: agent-scope monotonic stores, 2: tagged words (reduce_dev.hpp: reduce_acc32)
__device__ __forceinline__ void reduce_rows(bool valid, Lin L, double huber_delta, bool no_loss, int kind, double *lds_red /*4*32*/,
                                            double *__restrict__ partial_out, int mult = 1, unsigned tag = 0u)
{
    double acc[32];
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
        if (mult > 1) {
            const double k = double(mult);
#pragma unroll
            for (int i = 0; i < 29; ++i) acc[i] *= k;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 29; ++i) acc[i] = 0.0;
    }
    acc[29] = acc[30] = acc[31] = 0.0;
    reduce_acc32<MODE>(acc, kind, lds_red, partial_out, tag);
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

constexpr int MAX_BLOCKS = 8;

struct KindP {
    GridDev grid;
    const float4 *feat;      // {x,y,z,intensity}; intensity < 0 marks a padding slot between pose blocks
    const float4 *covd;      // {cxx,cyy,czz,_} or null
    float4 *nbr;             // nbr_stride per feature: {x,y,z, sq-dist} of the k-th neighbour (w = +inf when missing)
    Corr *corr;
    double *r_out;           // nullable
    double *J_out;           // nullable
    int m;                   // feature slots (real + padding)
    int tiles_a;             // correspondence-kernel tiles (TPB / lanes features each)
    int lanes;               // lanes per query of the correspondence kernel for this kind (8 or 16)
    int tiles_b;             // fit / linearise tiles (256 features each)
    int nbr_stride;          // max K over the blocks
    int blk_start[MAX_BLOCKS + 1];   // first slot of every pose block (multiples of 256), blk_start[n_blocks] = m
};

struct KParams {
    KindP k[2];              // [MLH_SURF], [MLH_CORNER]; m = 0 when a kind is not part of the launch
    double *partials;        // tiles_b(surf) + tiles_b(corner) records
    SolverState *state;
    int pose_sel;
    uint32_t flags;
    float min_match_sq_dis, min_plane_dis;
    double huber_delta, cov_measurement_trace;
    int has_lo, has_hi;      // multi-GPU ownership half-spaces (mlh_shard_set)
    int own_mod, own_rem;    // ... or ownership by feature index: slot f belongs to this rank iff f % own_mod == own_rem (mlh_shard_set_features)
    float lo[4], hi[4];
    // pose blocks (BASELINE config 4: block 0 = body pose, block n = extrinsic of LiDAR n; 1 block otherwise)
    int n_blocks;
    int kb[MAX_BLOCKS];      // N_NEIGH per block (5 or 10)
    double thre_b[MAX_BLOCKS];   // eigen threshold per block
    int freeze_b[MAX_BLOCKS];    // 0: project the degenerate directions out (evalDegenracy); 1: do not update the block at all
    // fused Gauss-Newton finish: the last workgroup to arrive sums the partials, solves and updates the pose(s)
    int use_init;            // block 0's pose is init_pose (first iteration of a solve: no separate upload launch)
    double init_pose[7];
    HostPublish *publish;    // the finish of the last iteration hands the result to the host through pinned memory
    unsigned long long publish_seq;
    int knn_lanes;           // lanes per query of the correspondence kernel: 8 or 16 for every kind of the launch, 0 = per kind (KindP::lanes)
    int finish;              // 0: none, 1: GN (reduce + solve + Plus), 2: reduce into SolverState::ne only (multi-GPU),
                             // 3: Levenberg-Marquardt begin (fit kernel), 4: Levenberg-Marquardt step (linearize kernel)
    int lm_max_it, lm_min_blocks;
    int lm_expect_done;      // MatchArgs::lm_expect_done
    unsigned *ticket;
    IterStatDev *stat;       // n_blocks consecutive records, or null
    // sharded over several ranks with the mailbox communicator: the finishing workgroup exchanges each block's summed record with the peers (one hop, inside
    // this launch) before it solves -- a sharded Gauss-Newton iteration is the same two launches as an unsharded one. n_ranks <= 1: nothing is exchanged
    P2pDev p2p;
    // Gauss-Newton with the finish done by the consumer (MatchArgs::gn_iter): the correspondence kernel of iteration i >= 1 completes iteration i - 1 first
    int pre_finish;          // 1: sum the pre_tiles records the previous fit launch left in `partials`, solve, Plus -> this iteration's pose
    int pre_tiles;
    int pre_from_init;       // the previous iteration's pose is init_pose (kernel arguments); otherwise *x_prev
    int pre_from_state;      // ... or the state's own poses (x for block 0, xb[b] otherwise): iteration 1 of a solve over pose blocks
    // (x_prev / x_next / pose0 are the poses of block 0; block b's sit 7 doubles x b further)
    const double *x_prev;
    double *x_next;          // the workgroup that serves tile 0 stores the new pose here (the fit kernel of the same iteration reads it as pose0)
    const double *pose0;     // block 0's pose of this launch when it is neither init_pose nor the state's x / cand (iterations >= 1 of a deferred-finish solve)
    int warm;                // the neighbour records hold the previous iteration's neighbours of the same features in the same map
    // pre_finish == 2 (MatchArgs::pre_final): the records are the PREVIOUS solve's last iteration; its pose is published from here, then this frame's start pose chained from it
    HostPublish *pre_publish;
    unsigned long long pre_publish_seq;
    double pre_thre;
    int pre_freeze;
    double chain_prev[7], chain_cur[7];
    // Levenberg-Marquardt with the step done by the consumer (lm_consume_kernel): the records the previous launch left, the state its writer left, the state this
    // launch's writer leaves
    const double *partials_in;
    const LmState *lm_in;
    LmState *lm_out;
    // feature counts read on the DEVICE (mlh_downsample_scan2map: the solve is enqueued behind the thinning without the host reading what the thinning kept): the
    // launches are sized for an upper bound (KindP::m, tiles_*), the DEVM kernel variants take the counts -- and the tiles that follow from them -- from here
    const int *m_dev;        // [2]: surf, corner
    unsigned long long loop_timeout_ticks;   // lm_loop_kernel: a barrier wait longer than this (100 MHz wall clock) gives the loop up (mlh_ctx::caps)
    unsigned long long *loop_tagged;   // lm_loop_kernel: two sets of tagged records (64 words per tile), or null: records + grid barrier (MLH_LOOP_TAGGED=0)
    unsigned loop_tag_base;  // this launch's tag: (launch number << 8); the iteration goes into the low byte
    int debug_stall;         // MLH_DEBUG_LOOP_STALL=1 (tests): one workgroup of lm_loop_kernel never arrives at its second barrier -- the loop must end with the error bit, not hang
};

__device__ __forceinline__ int block_of_slot(const KindP &K, int n_blocks, int f)
{
    int b = 0;
    for (int i = 1; i < n_blocks; ++i) if (f >= K.blk_start[i]) b = i;
    return b;
}

__device__ __forceinline__ const double *block_pose(const KParams &P, int b)
{
    return b == 0 ? (P.pose_sel ? P.state->cand : P.state->x) : P.state->xb[b];
}

__device__ __forceinline__ void load_pose(const KParams &P, int b, q4 &q, d3 &t)
{
    if (P.use_init && b == 0) {            // uniform: straight from the kernel-argument segment
        t = d3{P.init_pose[0], P.init_pose[1], P.init_pose[2]};
        q = q4{P.init_pose[3], P.init_pose[4], P.init_pose[5], P.init_pose[6]};
    } else if (P.pose0) {                  // iteration >= 1 of a deferred-finish solve: the slot this iteration's correspondence launch filled (one per pose block)
        const double *pose = P.pose0 + 7 * b;
        t = d3{pose[0], pose[1], pose[2]};
        q = q4{pose[3], pose[4], pose[5], pose[6]};
    } else {
        const double *pose = block_pose(P, b);
        t = d3{pose[0], pose[1], pose[2]};
        q = q4{pose[3], pose[4], pose[5], pose[6]};
    }
}

// pointAssociateToMap (utility.h:103-117): f64 q*p + t, stored to f32
__device__ __forceinline__ void associate_to_map(const q4 &q, const d3 &t, const float4 &fp, float &sx, float &sy, float &sz)
{
    d3 w = qrot(q, d3{double(fp.x), double(fp.y), double(fp.z)});
    sx = float(w.x + t.x); sy = float(w.y + t.y); sz = float(w.z + t.z);
}

__device__ __forceinline__ bool owns(const KParams &P, int f, float sx, float sy, float sz)
{
    bool own = P.own_mod <= 1 || (f % P.own_mod) == P.own_rem;
    if (P.has_lo) own = own && ((((P.lo[0] * sx + P.lo[1] * sy) + P.lo[2] * sz) + P.lo[3]) >= 0.f);
    if (P.has_hi) own = own && ((((P.hi[0] * sx + P.hi[1] * sy) + P.hi[2] * sz) + P.hi[3]) < 0.f);
    return own;
}

// ---- correspondence kernel: 8 lanes per feature, 32 features per workgroup, both feature kinds in one launch

// the K winners' coordinates + squared distances -> the feature's neighbour records. A macro, expanded in the function that owns `keys`: as a function taking the
// array by reference, the lane's pick (a select chain over keys[t]) is folded into ONE load at a selected offset before the array is scalarised, and the
// keys then live in scratch / LDS instead of registers (seen in the ISA: 48 bytes of scratch, +10 KB of LDS per workgroup).
// lane t (< K <= G) fetches winner t: one load per lane, all in flight together
#define MLH_STORE_WINNERS(K_, G_, Kd_, f_, gl_, keys_)                                                          \
    do {                                                                                                        \
        unsigned long long kk_ = keys_[0];                                                                      \
        _Pragma("unroll") for (int t_ = 1; t_ < (K_); ++t_) kk_ = ((gl_) == t_ % (G_)) ? keys_[t_] : kk_;      \
        if ((K_) <= (G_)) {                                                                                     \
            if ((gl_) < (K_)) {                                                                                 \
                float4 o_ = make_float4(0.f, 0.f, 0.f, __uint_as_float(0x7f800000u));                           \
                if (kk_ != KEY_INF) {                                                                           \
                    const float4 np_ = (Kd_).grid.raw[(unsigned)kk_];                                           \
                    o_ = make_float4(np_.x, np_.y, np_.z, __uint_as_float((unsigned)(kk_ >> 32)));              \
                }                                                                                               \
                (Kd_).nbr[size_t(f_) * (Kd_).nbr_stride + (gl_)] = o_;                                          \
            }                                                                                                   \
        } else {                                                                                                \
            _Pragma("unroll") for (int t_ = 0; t_ < (K_); ++t_) {                                               \
                if ((t_ % (G_)) == (gl_)) {                                                                     \
                    const unsigned long long k2_ = keys_[t_];                                                   \
                    float4 o_ = make_float4(0.f, 0.f, 0.f, __uint_as_float(0x7f800000u));                       \
                    if (k2_ != KEY_INF) {                                                                       \
                        const float4 np_ = (Kd_).grid.raw[(unsigned)k2_];                                       \
                        o_ = make_float4(np_.x, np_.y, np_.z, __uint_as_float((unsigned)(k2_ >> 32)));          \
                    }                                                                                           \
                    (Kd_).nbr[size_t(f_) * (Kd_).nbr_stride + t_] = o_;                                         \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)

template <int K, int G>
__device__ __forceinline__ void knn_feature(const KParams &P, const KindP &Kd, int f, float sx, float sy, float sz, int gl, int *lds_run)
{
    unsigned long long keys[K];
    if constexpr (G == 32) knn_group_bounded<K, 32, true>(Kd.grid, sx, sy, sz, gl, lds_run, 0x7f800000u, keys);
    else if constexpr (G == 16) knn_group16_pruned<K>(Kd.grid, sx, sy, sz, gl, lds_run, keys);
    else knn_group8_pruned<K>(Kd.grid, sx, sy, sz, gl, lds_run, keys);
    MLH_KSTAGE(4);
    MLH_STORE_WINNERS(K, G, Kd, f, gl, keys);
    MLH_KSTAGE(5);
}

// The same feature one Gauss-Newton iteration later: `old` is this lane's share (lane t: records t, t + G, ... below K) of the neighbour records the previous
// iteration left -- K points of the SAME map. Their largest squared distance from the query's new position bounds the K-th neighbour's from above, so the search
// is a single walk over the cells within that bound (knn_group_bounded). A feature that had fewer than K neighbours (w = +inf) searches as before.
template <int K, int G, int NREC>
__device__ __forceinline__ void knn_feature_warm(const KParams &P, const KindP &Kd, int f, float sx, float sy, float sz, int gl, int *lds_run, const float4 (&old)[NREC])
{
    unsigned bits = 0u;
#pragma unroll
    for (int r = 0; r < NREC; ++r) {
        if (gl + r * G < K) {
            const float dx = old[r].x - sx, dy = old[r].y - sy, dz = old[r].z - sz;
            const float d = knn_sqdist(dx, dy, dz);
            const unsigned bb = (old[r].w < __uint_as_float(0x7f800000u) && d < __uint_as_float(0x7f800000u)) ? __float_as_uint(d) : 0x7f800000u;
            bits = bb > bits ? bb : bits;
        }
    }
    bits = group_max_u32<G>(bits);
    if (bits >= 0x7f800000u) { knn_feature<K, G>(P, Kd, f, sx, sy, sz, gl, lds_run); return; }     // uniform over the group
    unsigned long long keys[K];
    knn_group_bounded<K, G>(Kd.grid, sx, sy, sz, gl, lds_run, bits, keys);
    MLH_KSTAGE(4);
    MLH_STORE_WINNERS(K, G, Kd, f, gl, keys);
    MLH_KSTAGE(5);
}

// MB = more than one pose block in the launch (config 4); without it the block bookkeeping (a per-lane block index and the
// per-block K lookup it drags along) compiles away
template <int G, bool MB, bool K10, bool PRE, bool WARM>
__device__ __forceinline__ void knn_features_body(const KParams &P, const KindP &K, int tile, int *s_run, const double *s_pose, int m_feat)
{
    constexpr int FPB = TPB / G;          // queries per workgroup
    constexpr int RUNW = 2 * KNN_RUN_WORDS;
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int f = tile * FPB + grp;
    MLH_KSTAGE(0);
    if (f >= m_feat) return;
    const float4 fp = K.feat[f];
    if (fp.w < 0.f) return;               // padding slot
    const int b = MB ? block_of_slot(K, P.n_blocks, f) : 0;
    constexpr int NREC = WARM ? ((K10 ? 10 : 5) + G - 1) / G : 1;
    float4 old[NREC];
    if constexpr (WARM) {                  // requested together with the feature, before the pose is known
        const int kq = K10 ? (MB ? P.kb[b] : P.kb[0]) : 5;
#pragma unroll
        for (int r = 0; r < NREC; ++r) {
            old[r] = make_float4(0.f, 0.f, 0.f, __uint_as_float(0x7f800000u));
            if (gl + r * G < kq) old[r] = K.nbr[size_t(f) * K.nbr_stride + gl + r * G];
        }
    }
    q4 q;
    d3 t;
    if constexpr (PRE) {
        t = d3{s_pose[0], s_pose[1], s_pose[2]};
        q = q4{s_pose[3], s_pose[4], s_pose[5], s_pose[6]};
    } else {
        load_pose(P, b, q, t);
    }
    float sx, sy, sz;
    associate_to_map(q, t, fp, sx, sy, sz);
    MLH_KSTAGE(1);
    if (!owns(P, f, sx, sy, sz)) return;     // uniform over the lane group
    // K10: some pose block of the launch asks for 10 neighbours (buildCalibMap's non-reference LiDARs); without it the K = 10 search is not
    // even compiled in, so the ordinary frame's kernel keeps the K = 5 register footprint
    if constexpr (WARM) {
        if (K10 && (MB ? P.kb[b] : P.kb[0]) == 10) knn_feature_warm<10, G, NREC>(P, K, f, sx, sy, sz, gl, s_run + grp * RUNW, old);
        else knn_feature_warm<5, G, NREC>(P, K, f, sx, sy, sz, gl, s_run + grp * RUNW, old);
    } else {
        if (K10 && (MB ? P.kb[b] : P.kb[0]) == 10) knn_feature<10, G>(P, K, f, sx, sy, sz, gl, s_run + grp * RUNW);
        else knn_feature<5, G>(P, K, f, sx, sy, sz, gl, s_run + grp * RUNW);
    }
}

// The prologue of a correspondence launch that completes a Gauss-Newton iteration first (all TPB threads of the workgroup, converged):
//   s_pose <- the pose that iteration linearised at; the records its fit launch left are summed (sum_partials<TPB, 12>'s arithmetic -- same slices, same four chains,
//   same association: the same bits --, inlined: the record loads leave at once instead of behind an argument block's trip through scratch, and nobody waits for the
//   per-kind counts, which only statistics read); gn_finish_wave solves and applies Plus on the first wavefront; s_pose holds the updated pose when this returns.
// MODE 1: an iteration of the running solve. MODE 2: the LAST iteration of the PREVIOUS solve (thresholds from the launch arguments).
// b: the pose block this workgroup's features belong to (0 without blocks): the records summed are that block's tiles -- its surf tiles, then its corner tiles, as the
// classic finish walks them -- and the thresholds that block's.
template <int MODE, bool MB = false>
__device__ __forceinline__ void gn_prologue(const KParams &P, int b, double *s_pose, double *f_ne, double *f_cnt2, double *f_scratch)
{
    if (threadIdx.x < 7) {
        double v;
        if (P.pre_from_init) v = P.init_pose[threadIdx.x];
        else if (MB && P.pre_from_state) v = (b == 0 ? P.state->x : P.state->xb[b])[threadIdx.x];
        else v = P.x_prev[7 * b + threadIdx.x];
        s_pose[threadIdx.x] = v;
    }
    {
        constexpr int NS = TPB / 32, U = 12;
        const int c = threadIdx.x & 31, sl = threadIdx.x >> 5;
        int lo0 = 0, n0 = P.pre_tiles, lo1 = 0, n1 = 0;
        if constexpr (MB) {
            lo0 = P.k[0].m > 0 ? P.k[0].blk_start[b] / TPB : 0;
            n0 = P.k[0].m > 0 ? (P.k[0].blk_start[b + 1] + TPB - 1) / TPB - lo0 : 0;
            lo1 = P.k[0].tiles_b + (P.k[1].m > 0 ? P.k[1].blk_start[b] / TPB : 0);
            n1 = P.k[1].m > 0 ? (P.k[0].tiles_b + (P.k[1].blk_start[b + 1] + TPB - 1) / TPB) - lo1 : 0;
            n0 = max(n0, 0); n1 = max(n1, 0);
        }
        const int ntot = n0 + n1;
        const double *__restrict__ rec = P.partials;
        double ch[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = sl; j < ntot; j += U * NS) {
            double tv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + NS * u;
                const int tile = jj < n0 ? lo0 + jj : lo1 + (jj - n0);
                tv[u] = jj < ntot ? rec[size_t(tile) * NE_STRIDE + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) ch[u & 3] += tv[u];
        }
        f_scratch[sl * 32 + c] = (ch[0] + ch[1]) + (ch[2] + ch[3]);
        __syncthreads();
        if (threadIdx.x < 32) {
            double tsum = 0.0;
#pragma unroll
            for (int q = 0; q < NS; ++q) tsum += f_scratch[q * 32 + c];
            f_ne[c] = tsum;
        }
        __syncthreads();
    }
    if (threadIdx.x < 64) {
        double xo[7];
        gn_finish_wave(f_ne, f_cnt2, s_pose, nullptr, MODE == 2 ? P.pre_thre : P.thre_b[b], MODE == 2 ? P.pre_freeze : P.freeze_b[b], nullptr, f_scratch, xo);
    }
    __syncthreads();
}

// the previous solve's pose, final now: to its host record (seq stored last, system-scope release) and to the state. One thread of ONE workgroup.
__device__ __forceinline__ void publish_final_pose(const KParams &P, const double *s_pose)
{
    for (int i = 0; i < 7; ++i) P.state->x[i] = s_pose[i];
    if (P.pre_publish) {
        for (int i = 0; i < 7; ++i) P.pre_publish->x[i] = s_pose[i];
        __hip_atomic_store(&P.pre_publish->seq, P.pre_publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// G = lanes per query for both kinds, or 0: per kind (KindP::lanes -- a workgroup serves one kind, so the choice is uniform over it)
// PRE 1: the launch completes the previous Gauss-Newton iteration first -- EVERY workgroup sums the records the previous fit launch's tiles left (same order,
//   same code: the same bits everywhere), runs the 6 x 6 solve + Plus on one wavefront and goes on with the pose in LDS; the kernel boundary behind the fit
//   launch is the only synchronisation (no ticket, no fence, no last workgroup whose serial tail the other 255 compute units wait for). The workgroup of
//   tile 0 leaves the pose in HBM for the fit kernel of this iteration.
// PRE 2: iteration 0 of a chained solve whose predecessor left its LAST iteration as records: the same prologue completes that solve (tile 0's workgroup publishes its
//   pose to the host and stores it as the state's), then every workgroup computes this frame's chained start pose from it (chain_start_pose, one lane) -- the
//   predecessor's serial finish and the chain launch between the two frames are gone.
// WARM: the search is bounded by the previous iteration's neighbours (knn_feature_warm). All three only in the single-block, K = 5 launches of mlh_gn_solve*.
// DEVM: the feature counts come from P.m_dev (G = 0 only: lanes per kind); the grid was sized for a bound, the workgroups beyond the real tiles leave
template <int G, bool MB, bool K10, int PRE = 0, bool WARM = false, bool DEVM = false>
__global__ __launch_bounds__(TPB) void knn_features_kernel(KParams P)
{
    __shared__ int s_run[(G == 16) ? (TPB / 16) * 2 * KNN_RUN_WORDS : (TPB / 8) * 2 * KNN_RUN_WORDS];
    __shared__ double s_pose[PRE ? 8 : 1];
    int m0 = P.k[0].m, m1 = P.k[1].m, ta0 = P.k[0].tiles_a;
    int total = P.k[0].tiles_a + P.k[1].tiles_a;
    if constexpr (DEVM) {
        static_assert(G == 0 && !MB && PRE == 0, "device-side counts: per-kind lanes, one block, no prologue");
        m0 = P.m_dev[0]; m1 = P.m_dev[1];
        const int f0 = TPB / P.k[0].lanes, f1 = TPB / P.k[1].lanes;
        ta0 = (m0 + f0 - 1) / f0;
        total = ta0 + (m1 + f1 - 1) / f1;
        if ((int(blockIdx.x) >> 3) >= ((total + 7) >> 3)) return;     // (xcd_tile is a bijection only on the first 8 * ceil(total / 8) workgroups)
    }
    int tile = xcd_tile(total);
    if (tile >= total) return;
#ifdef MLH_KNN_HEAVY_FIRST
    tile = total - 1 - tile;       // A/B build only (scripts/build_variant.py): the corner tiles -- the queries with the most candidates -- are dispatched first
#endif
    if constexpr (PRE != 0) {
        __shared__ double f_ne[NE_STRIDE], f_cnt2[2], f_scratch[(TPB / 32) * 32];
        int pb = 0;
        bool writer = tile == 0;
        if constexpr (MB) {
            // pose blocks start on 256-slot boundaries and a workgroup serves 16 or 32 consecutive slots of ONE kind: all of them in one block. The block's pose is
            // written by the workgroup of its first surf tile (the host defers a solve over blocks only when every block has surf features)
            const int kk = tile >= P.k[0].tiles_a ? 1 : 0;
            const int tk = kk ? tile - P.k[0].tiles_a : tile;
            const int lanes = (G == 0) ? P.k[kk].lanes : G;
            const int f0 = tk * (TPB / lanes);
            pb = block_of_slot(P.k[kk], P.n_blocks, f0);
            writer = kk == 0 && f0 == P.k[0].blk_start[pb];
        }
        gn_prologue<PRE, MB>(P, pb, s_pose, f_ne, f_cnt2, f_scratch);
        if constexpr (PRE == 2) {
            __shared__ double s_chain[8];
            if (threadIdx.x == 0) {
                if (tile == 0) publish_final_pose(P, s_pose);
                double xc[7], out[7];
                for (int i = 0; i < 7; ++i) xc[i] = s_pose[i];
                double cp[7], cc[7];
                for (int i = 0; i < 7; ++i) { cp[i] = P.chain_prev[i]; cc[i] = P.chain_cur[i]; }
                chain_start_pose(xc, cp, cc, out);
                for (int i = 0; i < 7; ++i) s_chain[i] = out[i];
            }
            __syncthreads();
            if (threadIdx.x < 7) s_pose[threadIdx.x] = s_chain[threadIdx.x];
            __syncthreads();
        }
        if (writer && threadIdx.x < 7) P.x_next[7 * pb + threadIdx.x] = s_pose[threadIdx.x];
    }
    const int kind = tile >= ta0 ? 1 : 0;
    if (kind) tile -= ta0;
    const KindP &K = P.k[kind];
    const int m_feat = kind ? m1 : m0;
    if constexpr (G == 0) {
        if (K.lanes == 8) knn_features_body<8, MB, K10, PRE != 0, WARM>(P, K, tile, s_run, s_pose, m_feat);
        else if (K.lanes == 32) knn_features_body<32, MB, K10, PRE != 0, WARM>(P, K, tile, s_run, s_pose, m_feat);
        else knn_features_body<16, MB, K10, PRE != 0, WARM>(P, K, tile, s_run, s_pose, m_feat);
    } else {
        knn_features_body<G, MB, K10, PRE != 0, WARM>(P, K, tile, s_run, s_pose, m_feat);
    }
}

// a pending last iteration completed by itself (no chained successor took it): one workgroup, the same prologue, the same publication
__global__ __launch_bounds__(TPB) void gn_final_kernel(KParams P)
{
    __shared__ double s_pose[8], f_ne[NE_STRIDE], f_cnt2[2], f_scratch[(TPB / 32) * 32];
    gn_prologue<2>(P, 0, s_pose, f_ne, f_cnt2, f_scratch);
    if (threadIdx.x == 0) publish_final_pose(P, s_pose);
}

// the reference's plane fit + gate (feature_extract.hpp:816-840)
template <int K>
__device__ __forceinline__ bool fit_plane(const float (&ax)[K], const float (&ay)[K], const float (&az)[K], float min_plane_dis, float (&coef)[6])
{
    float nx, ny, nz;
    plane_fit_qr_f<K>(ax, ay, az, nx, ny, nz);
    float nn = sqrtf(nx * nx + ny * ny + nz * nz);
    float negative_OA_dot_norm = 1 / nn;
    float z = nx * nx + ny * ny + nz * nz;
    if (z > 0.f) { float s = sqrtf(z); nx /= s; ny /= s; nz /= s; }
    bool plane_valid = true;
#pragma unroll
    for (int j = 0; j < K; ++j)
        if (fabsf(nx * ax[j] + ny * ay[j] + nz * az[j] + negative_OA_dot_norm) > min_plane_dis) plane_valid = false;
    coef[0] = nx; coef[1] = ny; coef[2] = nz; coef[3] = negative_OA_dot_norm;
    return plane_valid;
}

// the reference's line fit + test (feature_extract.hpp:669-693, 767-777)
template <int K>
__device__ __forceinline__ bool fit_line(const float (&ax)[K], const float (&ay)[K], const float (&az)[K], float (&coef)[6])
{
    float cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) { cx += ax[j]; cy += ay[j]; cz += az[j]; }
    const float kf = float(K);
    cx /= kf; cy /= kf; cz /= kf;
    float c00 = 0.f, c10 = 0.f, c11 = 0.f, c20 = 0.f, c21 = 0.f, c22 = 0.f;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        float t0 = ax[j] - cx, t1 = ay[j] - cy, t2 = az[j] - cz;
        c00 += t0 * t0; c10 += t1 * t0; c11 += t1 * t1; c20 += t2 * t0; c21 += t2 * t1; c22 += t2 * t2;
    }
    float l0, l1, l2, vx, vy, vz;
    eig3_largest_f(c00, c10, c11, c20, c21, c22, l0, l1, l2, vx, vy, vz);
    coef[0] = 0.1f * vx + cx; coef[1] = 0.1f * vy + cy; coef[2] = 0.1f * vz + cz;
    coef[3] = -0.1f * vx + cx; coef[4] = -0.1f * vy + cy; coef[5] = -0.1f * vz + cz;
    return l2 > 3 * l1;
}

__device__ __forceinline__ double feature_weight(const KParams &P, const KindP &K, int f)
{
    double trace = P.cov_measurement_trace;
    if ((P.flags & MLH_FLAG_WITH_UA)) {
        trace = 0.0;
        if (K.covd) { float4 cd = K.covd[f]; trace = (double(cd.x) + double(cd.y)) + double(cd.z); }
    }
    return sqrt_info_of(trace);
}

// the same from a covariance diagonal the caller already holds in registers (requested together with the feature: no dependent trip)
__device__ __forceinline__ double feature_weight_pref(const KParams &P, const KindP &K, const float4 &cd)
{
    double trace = P.cov_measurement_trace;
    if ((P.flags & MLH_FLAG_WITH_UA)) {
        trace = 0.0;
        if (K.covd) trace = (double(cd.x) + double(cd.y)) + double(cd.z);
    }
    return sqrt_info_of(trace);
}

// Fused tail: the last workgroup to arrive (agent-scope release/acquire around an atomic ticket) sums the partial records
// in fixed order, runs the degeneracy test + the 6x6 solve + Plus for every pose block and re-arms the ticket: a GN iteration
// costs two launches.
template <bool LM, int NT = TPB>
__device__ __forceinline__ void fused_gn_finish(const KParams &P, int total_tiles)
{
    __shared__ int s_last;
    __shared__ double f_ne[NE_STRIDE], f_cnt2[2], f_scratch[(NT / 32) * 32];   // f_scratch doubles as the Jacobi work area (DEG_WORK <= 256)
    __syncthreads();                               // this workgroup's partial record is written
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = atomicAdd(P.ticket, 1u);
        s_last = (tk == unsigned(total_tiles - 1)) ? 1 : 0;
    }
    __syncthreads();
    if (!s_last) return;
    MLH_STAGE(4095, 0);
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (P.use_init && threadIdx.x < 7) P.state->x[threadIdx.x] = P.init_pose[threadIdx.x];   // the state's pose is born here
    __syncthreads();
    if constexpr (LM) {
        // scan2map on one GPU: the Levenberg-Marquardt begin / step of solver_dev.hpp runs right here, so an LM iteration is ONE launch
        SumArgs sa;
        sa.p = P.partials;
        sa.lo[0] = 0; sa.hi[0] = total_tiles; sa.lo[1] = 0; sa.hi[1] = 0;
        sum_partials<NT, (NT > 256 ? 21 : 12)>(sa, f_ne, f_cnt2, f_scratch);
        bool exchanged = true;
        if (P.p2p.n_ranks > 1) {     // sharded over several ranks: everybody's sums before anybody's LM decision (same bits, same decision on every rank)
            exchanged = p2p_exchange<NT>(P.p2p, f_ne, NE_STRIDE);
            if (threadIdx.x == 0) { f_cnt2[0] = f_ne[NE_CNT + 1]; f_cnt2[1] = f_ne[NE_CNT + 2]; }
            __syncthreads();
        }
        MLH_STAGE(4095, 1);
        if (threadIdx.x < 64) {      // one wavefront runs the LM begin / step (solver_dev.hpp: rows of the 6 x 6 objects on lanes)
            double xo[7];
            int done = 0;
            if (!exchanged) {        // a peer timed out: no decision on partial sums -- the loop ends here, the pose stays, the host reports the error word
#pragma unroll
                for (int i = 0; i < 7; ++i) xo[i] = P.state->x[i];
                done = 1;
                if (threadIdx.x == 0) P.state->done = 1;
            }
            else if (P.finish == 3) {
                // split submission: this outer iteration was enqueued without anybody having seen the previous LM loop end. If it has not, the frame's result is
                // not the reference's (<= max_num_iterations per outer iteration): flagged, published with the pose, and the host re-solves or reports
                if (threadIdx.x == 0 && P.lm_expect_done) {
                    P.state->lm_overflow = (P.lm_expect_done > 0 && (P.state->lm_overflow || !P.state->done)) ? 1 : 0;
                    P.state->lm_used_max = P.lm_expect_done > 0 ? fmax(P.state->lm_used_max, double(P.state->iteration)) : 0.0;      // the loop that just ended
                }
                lm_begin_body_wave(f_ne, f_cnt2, f_scratch, P.state, P.thre_b[0], P.lm_max_it, P.stat, P.lm_min_blocks, xo, done);
            }
            else lm_step_body_wave(f_ne, P.state, P.lm_max_it, xo, done);
            if (threadIdx.x == 0) {
                *P.ticket = 0u;
                if (P.publish) {     // last launch of a chunk of LM steps: the pose and the `done` flag go to the host from here
                    for (int i = 0; i < 7; ++i) P.publish->x[i] = xo[i];
                    P.publish->done = done | (P.state->lm_overflow ? 2 : 0);
                    P.publish->xb[2][0] = fmax(P.state->lm_used_max, double(P.state->iteration));
                    __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        MLH_STAGE(4095, 2);
        return;
    }
    if (P.finish == 2) {
        // multi-GPU: only the local reduction happens here; the all-reduce and the (redundant, identical) solve follow
        if (P.n_blocks == 1) {
            SumArgs sa;
            sa.p = P.partials;
            sa.lo[0] = 0; sa.hi[0] = total_tiles; sa.lo[1] = 0; sa.hi[1] = 0;
            sum_partials<NT, (NT > 256 ? 21 : 12)>(sa, f_ne, f_cnt2, f_scratch);
            if (threadIdx.x < NE_STRIDE) P.state->ne[threadIdx.x] = f_ne[threadIdx.x];
        } else {
            for (int b = 0; b < P.n_blocks; ++b) {         // one record per pose block, all-reduced in one message
                SumArgs sa;
                sa.p = P.partials;
                sa.lo[0] = P.k[0].m > 0 ? P.k[0].blk_start[b] / TPB : 0;
                sa.hi[0] = P.k[0].m > 0 ? (P.k[0].blk_start[b + 1] + TPB - 1) / TPB : 0;
                sa.lo[1] = P.k[0].tiles_b + (P.k[1].m > 0 ? P.k[1].blk_start[b] / TPB : 0);
                sa.hi[1] = P.k[0].tiles_b + (P.k[1].m > 0 ? (P.k[1].blk_start[b + 1] + TPB - 1) / TPB : 0);
                sum_partials<NT, (NT > 256 ? 21 : 12)>(sa, f_ne, f_cnt2, f_scratch);
                if (threadIdx.x < NE_STRIDE) P.state->neb[b][threadIdx.x] = f_ne[threadIdx.x];
                __syncthreads();
            }
        }
        if (threadIdx.x == 0) *P.ticket = 0u;
        return;
    }
    for (int b = 0; b < P.n_blocks; ++b) {
        SumArgs sa;
        sa.p = P.partials;
        sa.lo[0] = P.k[0].m > 0 ? P.k[0].blk_start[b] / TPB : 0;
        sa.hi[0] = P.k[0].m > 0 ? (P.k[0].blk_start[b + 1] + TPB - 1) / TPB : 0;
        sa.lo[1] = P.k[0].tiles_b + (P.k[1].m > 0 ? P.k[1].blk_start[b] / TPB : 0);
        sa.hi[1] = P.k[0].tiles_b + (P.k[1].m > 0 ? (P.k[1].blk_start[b + 1] + TPB - 1) / TPB : 0);
        if (P.n_blocks == 1) { sa.lo[0] = 0; sa.hi[0] = total_tiles; sa.lo[1] = 0; sa.hi[1] = 0; }   // one block: every record (any tile size)
        sum_partials<NT, (NT > 256 ? 21 : 12)>(sa, f_ne, f_cnt2, f_scratch);
        bool exchanged = true;
        if (P.p2p.n_ranks > 1) {                     // this rank's sums -> everybody's sums (rank order: the same bits on every rank)
            exchanged = p2p_exchange<NT>(P.p2p, f_ne, NE_STRIDE);
            if (threadIdx.x == 0) { f_cnt2[0] = f_ne[NE_CNT + 1]; f_cnt2[1] = f_ne[NE_CNT + 2]; }
            __syncthreads();
        }
        MLH_STAGE(4095, 1);
        if (threadIdx.x < 64 && !exchanged) {        // a peer timed out: nothing is solved on partial sums; the pose stays and is published as it is (the host reports the error word)
            if (P.publish && threadIdx.x == 0) for (int i = 0; i < 7; ++i) (b == 0 ? P.publish->x : P.publish->xb[b])[i] = (b == 0 ? P.state->x : P.state->xb[b])[i];
        } else if (threadIdx.x < 64) {
            double xo[7];
            const double *x_in = P.pose0 ? P.pose0 + 7 * b : nullptr;      // last iteration of a deferred-finish solve: the pose comes from its iteration slot
            gn_finish_wave(f_ne, f_cnt2, b == 0 ? P.state->x : P.state->xb[b], nullptr /* nothing downstream reads a mirror of ne / V_update in GN mode */, P.thre_b[b], P.freeze_b[b],
                           P.stat ? P.stat + b : nullptr, f_scratch, xo, x_in);
            if (x_in && threadIdx.x < 7) {           // ... and goes to the state whatever the solve decided (an unchanged pose included)
                const int l = threadIdx.x;
                (b == 0 ? P.state->x : P.state->xb[b])[l] = l == 0 ? xo[0] : (l == 1 ? xo[1] : (l == 2 ? xo[2] : (l == 3 ? xo[3] : (l == 4 ? xo[4] : (l == 5 ? xo[5] : xo[6])))));
            }
            // the solve's last launch hands the pose(s) to the host: straight from the finish's registers (reading the state back would be one more round trip)
            if (P.publish && threadIdx.x == 0) for (int i = 0; i < 7; ++i) (b == 0 ? P.publish->x : P.publish->xb[b])[i] = xo[i];
        }
        __syncthreads();
        MLH_STAGE(4095, 2);
    }
    if (threadIdx.x == 0) {
        *P.ticket = 0u;
#if defined(MLH_EXP) && MLH_EXP == 5
        if (P.publish) __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#elif defined(MLH_EXP) && MLH_EXP == 6
        (void)0;
#else
        if (P.publish) __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
    }
}

// v: the feature's neighbour records {x, y, z, sq-dist}, already in registers (the kernel requests them together with the feature itself)
template <int K, int KV>
__device__ __forceinline__ bool fit_feature(const KParams &P, int kind, const float4 (&v)[KV], float (&coef)[6])
{
    static_assert(K <= KV, "neighbour buffer too small");
    if (!(v[K - 1].w < P.min_match_sq_dis)) return false;     // sq_dis[k-1] < MIN_MATCH_SQ_DIS (hpp:667/814)
    float ax[K], ay[K], az[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { ax[j] = v[j].x; ay[j] = v[j].y; az[j] = v[j].z; }
    return kind == MLH_SURF ? fit_plane<K>(ax, ay, az, P.min_plane_dis, coef) : fit_line<K>(ax, ay, az, coef);
}

// ---- fit + gates + residual/Jacobian + normal-equation reduction: one lane per feature, both kinds in one launch
// KMAX = the largest N_NEIGH of the launch: with 5 (every mapper launch) the 10-neighbour fits are not compiled in, which halves
// the kernel's register footprint (occupancy matters once a launch has more workgroups than the chip holds at once)
// FIN = false: the launch only leaves its tiles' partial records (a deferred-finish Gauss-Newton iteration, or a caller that reduces elsewhere): no ticket, no
// finishing workgroup, and none of that code in the kernel
template <int KMAX, bool LM, bool FIN = true, bool DEVM = false>
__global__ __launch_bounds__(TPB) void fit_linearize_kernel(KParams P)
{
    __shared__ double s_red[4 * 32];
    int m0 = P.k[0].m, m1 = P.k[1].m, tb0 = P.k[0].tiles_b;
    int total = P.k[0].tiles_b + P.k[1].tiles_b;
    if constexpr (DEVM) {            // (the counts from the device, knn_features_kernel<.., DEVM>)
        m0 = P.m_dev[0]; m1 = P.m_dev[1];
        tb0 = (m0 + TPB - 1) / TPB;
        total = tb0 + (m1 + TPB - 1) / TPB;
        if ((int(blockIdx.x) >> 3) >= ((total + 7) >> 3)) return;
    }
    const int gtile = xcd_tile(total);
    if (gtile >= total) return;
    const int kind = gtile >= tb0 ? 1 : 0;
    const int tile = kind ? gtile - tb0 : gtile;
    const KindP &K = P.k[kind];
    const int m_feat = kind ? m1 : m0;
    MLH_STAGE(gtile, 0);
    const int f = tile * TPB + threadIdx.x;
    const int b = block_of_slot(K, P.n_blocks, tile * TPB);       // uniform over the workgroup (blocks start on tile boundaries)
    q4 q;
    d3 t;
    load_pose(P, b, q, t);
    bool valid = false;
    Lin L;
    L.r = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) L.J[i] = 0.0;
    float coef[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 fp = make_float4(0.f, 0.f, 0.f, -1.f), cdv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < m_feat) {
        // ONE memory round trip before the fit: the feature and all of its neighbour records are requested together (the records sit at
        // f * stride whatever the feature turns out to be; what is fetched for a padding slot or a feature another rank owns is discarded).
        // The map-frame position is only needed by the ownership planes of a sharded map and by the field-of-view gate: an ordinary
        // single-GPU launch skips the f64 transform here altogether (uniform branch).
        fp = K.feat[f];
        float4 nbv[KMAX];
        const float4 *nb = K.nbr + size_t(f) * K.nbr_stride;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) nbv[j] = nb[j];
        if ((P.flags & MLH_FLAG_WITH_UA) && K.covd) cdv = K.covd[f];       // the weight's covariance diagonal rides in the same trip
        MLH_STAGE(gtile, 5);
        const bool need_pos = P.has_lo || P.has_hi || (P.flags & MLH_FLAG_CHECK_FOV);
        float sx = 0.f, sy = 0.f, sz = 0.f;
        if (need_pos) associate_to_map(q, t, fp, sx, sy, sz);
        if (fp.w >= 0.f && owns(P, f, sx, sy, sz)) {
            valid = (KMAX == 10 && P.kb[b] == 10) ? fit_feature<KMAX, KMAX>(P, kind, nbv, coef) : fit_feature<5, KMAX>(P, kind, nbv, coef);
            if (valid && (P.flags & MLH_FLAG_CHECK_FOV)) valid = in_laser_fov(q, t, sx, sy, sz);
        }
        MLH_STAGE(gtile, 6);
        Corr c;
#pragma unroll
        for (int i = 0; i < 6; ++i) c.c[i] = valid ? coef[i] : 0.f;
        c.valid = valid ? 1 : 0;
        c.pad = 0;
        K.corr[f] = c;
    }
    MLH_STAGE(gtile, 1);
    if (valid) {
        const double w = feature_weight_pref(P, K, cdv);
        double R[9];
        qtorot(q, R);
        const d3 p{double(fp.x), double(fp.y), double(fp.z)};
        if (kind == MLH_SURF) eval_plane(p, coef, w, q, t, R, L);
        else eval_edge(p, coef, w, q, t, R, L);
    }
    if (K.r_out && f < m_feat) {
        K.r_out[f] = L.r;
#pragma unroll
        for (int i = 0; i < 6; ++i) K.J_out[size_t(f) * 6 + i] = L.J[i];
    }
    MLH_STAGE(gtile, 2);
    reduce_rows(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, kind, s_red, P.partials + size_t(gtile) * NE_STRIDE);
    MLH_STAGE(gtile, 3);
    if constexpr (FIN) { if (P.finish) fused_gn_finish<LM>(P, total); }
    MLH_STAGE(gtile, 4);
}

#ifdef MLH_STAGE_CLOCK
}  // namespace mlh
extern "C" int mlh_debug_stage_clock(unsigned long long *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk), sizeof(unsigned long long) * size_t(n_words));
}
extern "C" int mlh_debug_stage_clock_step(unsigned long long *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_step_clk), sizeof(unsigned long long) * size_t(n_words));
}
extern "C" int mlh_debug_stage_clock_knn(unsigned long long *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk_knn), sizeof(unsigned long long) * size_t(n_words));
}
namespace mlh {
#endif

template <bool LM>
__global__ __launch_bounds__(TPB) void linearize_kernel(KParams P)
{
    __shared__ double s_red[4 * 32];
    const int total = P.k[0].tiles_b + P.k[1].tiles_b;
    const int gtile = xcd_tile(total);
    if (gtile >= total) return;
    if (P.pose_sel && P.state->done) {   // candidate evaluation after the device-side LM loop has terminated: keep the partials defined, do no work
        if (threadIdx.x < 32) P.partials[size_t(gtile) * NE_STRIDE + threadIdx.x] = 0.0;
        if (LM && P.publish && gtile == 0 && threadIdx.x == 0) {      // ... but the host may be waiting for this launch's publication
            for (int i = 0; i < 7; ++i) P.publish->x[i] = P.state->x[i];
            P.publish->done = P.state->done | (P.state->lm_overflow ? 2 : 0);
            P.publish->xb[2][0] = fmax(P.state->lm_used_max, double(P.state->iteration));
            __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    const int kind = gtile >= P.k[0].tiles_b ? 1 : 0;
    const int tile = kind ? gtile - P.k[0].tiles_b : gtile;
    const KindP &K = P.k[kind];
    MLH_STAGE(gtile, 0);
    const int f = tile * TPB + threadIdx.x;
    const double *pose = block_pose(P, block_of_slot(K, P.n_blocks, tile * TPB));
    const q4 q{pose[3], pose[4], pose[5], pose[6]};
    const d3 t{pose[0], pose[1], pose[2]};
    bool valid = false;
    int mult = 1;
    Lin L;
    L.r = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) L.J[i] = 0.0;
    if (f < K.m) {
        // correspondence, feature and (with uncertainty weighting) its covariance diagonal in ONE round trip, whatever `valid` turns out to be
        const Corr c = K.corr[f];
        const float4 fp = K.feat[f];
        float4 cdv = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((P.flags & MLH_FLAG_WITH_UA) && K.covd) cdv = K.covd[f];
        if (c.valid) {
            valid = true;
            mult = c.valid;
            const double w = feature_weight_pref(P, K, cdv);
            double R[9];
            qtorot(q, R);
            const d3 p{double(fp.x), double(fp.y), double(fp.z)};
            if (kind == MLH_SURF) eval_plane(p, c.c, w, q, t, R, L);
            else eval_edge(p, c.c, w, q, t, R, L);
        }
    }
    if (K.r_out && f < K.m) {
        K.r_out[f] = L.r;
#pragma unroll
        for (int i = 0; i < 6; ++i) K.J_out[size_t(f) * 6 + i] = L.J[i];
    }
    MLH_STAGE(gtile, 2);
    reduce_rows(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, kind, s_red, P.partials + size_t(gtile) * NE_STRIDE, mult);
    MLH_STAGE(gtile, 3);
    if constexpr (LM) { if (P.finish == 3 || P.finish == 4) fused_gn_finish<true>(P, total); }   // 3: the LM begin on rows a selection kept (scan2map with good-feature selection)
    MLH_STAGE(gtile, 4);
}

// ---- Levenberg-Marquardt, the step done by the CONSUMER of the records (the Gauss-Newton path's round-4 move, for the loop scan2MapOptimization really runs).
// linearize_kernel<true> evaluates at the candidate, leaves its tile's record, takes a ticket, and the last workgroup to arrive sums the records and runs the LM
// step while 87 others have left: fence + ticket + acquire, then sum + step, all behind the slowest tile. Here a launch ends at its records; the NEXT launch's every
// workgroup sums them (sum_partials<TPB, 12>'s slices, chains and association: the same bits in every workgroup), runs the LM begin / step on its first wavefront
// (lm_begin_wave_pp / lm_step_wave_pp: the accept / reject decision and the next candidate are a function of the records and the state, so all workgroups hold the
// same candidate), and evaluates its own tile there -- with the tile's correspondences, features and covariances requested before the sum, so that their trip
// runs under the step. The state lives in two LmState records: launch g reads [(g - 1) & 1], its tile-0 workgroup writes [g & 1] (and mirrors the accepted pose
// into SolverState::x, which only launches behind this one read); the records alternate between two buffers the same way. A launch that finds the loop terminated
// copies the state forward and leaves (the host enqueues a look-ahead of launches without reading the verdict in between, as before).
// FIRST: the launch behind a match launch whose fit kernel ran with finish 0 -- the records are at the state's pose (or init_pose), the LM loop begins here.
__device__ __forceinline__ void lmc_publish(const KParams &P, const double (&x)[7], int done, int overflow, double used_max, int iteration)
{
    for (int i = 0; i < 7; ++i) P.publish->x[i] = x[i];
    P.publish->done = done | (overflow ? 2 : 0);
    P.publish->xb[2][0] = fmax(used_max, double(iteration));
    __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <bool FIRST>
__global__ __launch_bounds__(TPB) void lm_consume_kernel(KParams P)
{
    __shared__ double s_red[4 * 32];
    __shared__ double f_ne[NE_STRIDE], f_scratch[(TPB / 32) * 32];
    __shared__ double s_cand[8];
    __shared__ int s_done;
    const int total = P.k[0].tiles_b + P.k[1].tiles_b;
    const int gtile = xcd_tile(total);
    if (gtile >= total) return;
    const LmState *Si = P.lm_in;
    LmState *So = P.lm_out;
    const bool writer = gtile == 0;
    if constexpr (!FIRST) {
        if (Si->done) {                    // the loop ended in an earlier launch (uniform over the grid): the state goes forward as it is
            if (writer && threadIdx.x < 64) {
                const int lane = threadIdx.x;
                constexpr int NW = int(sizeof(LmState) / sizeof(double));
                static_assert(sizeof(LmState) % sizeof(double) == 0, "LmState is copied in doubles");
                const double *src = reinterpret_cast<const double *>(Si);
                double *dst = reinterpret_cast<double *>(So);
                for (int i = lane; i < NW; i += 64) dst[i] = src[i];
                if (P.publish && lane == 0) {
                    double x[7];
                    for (int i = 0; i < 7; ++i) x[i] = Si->x[i];
                    lmc_publish(P, x, Si->done, Si->lm_overflow, Si->lm_used_max, Si->iteration);
                }
            }
            return;
        }
    }
    // this tile's inputs: requested now, used after the step
    const int kind = gtile >= P.k[0].tiles_b ? 1 : 0;
    const int tile = kind ? gtile - P.k[0].tiles_b : gtile;
    const KindP &K = P.k[kind];
    const int f = tile * TPB + threadIdx.x;
    Corr c;
    c.valid = 0;
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f), cdv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < K.m) {
        c = K.corr[f];
        fp = K.feat[f];
        if ((P.flags & MLH_FLAG_WITH_UA) && K.covd) cdv = K.covd[f];
    }
    lmc_sum_records(P.partials_in, total, f_ne, f_scratch);
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        LmRegs R;
        double cand[7];
        int overflow;
        double used_max;
        if constexpr (FIRST) {
            double x[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i] = P.use_init ? P.init_pose[i] : P.state->x[i];
            // (split submission: this outer iteration was enqueued without anybody having seen the previous LM loop end -- fused_gn_finish's bookkeeping)
            if (P.lm_expect_done) {
                overflow = (P.lm_expect_done > 0 && (Si->lm_overflow || !Si->done)) ? 1 : 0;
                used_max = P.lm_expect_done > 0 ? fmax(Si->lm_used_max, double(Si->iteration)) : 0.0;
            } else { overflow = Si->lm_overflow; used_max = Si->lm_used_max; }
            lm_begin_wave_pp(f_ne, f_scratch, x, So, writer, P.thre_b[0], P.lm_max_it, P.lm_min_blocks, R, cand);
            if (writer && P.use_init && lane < 7) P.state->x[lane] = pick7(x, lane);       // the state's pose is born here
        } else {
            overflow = Si->lm_overflow; used_max = Si->lm_used_max;
            lm_step_wave_pp(f_ne, Si, So, writer, P.lm_max_it, R, cand);
            if (writer && lane < 7) P.state->x[lane] = pick7(R.x, lane);                   // read by the launches BEHIND this one only
        }
        if (lane < 7) s_cand[lane] = pick7(cand, lane);
        if (lane == 0) s_done = R.done;
        if (writer && lane == 0) {
            So->lm_overflow = overflow; So->lm_used_max = used_max;
            if (P.publish) lmc_publish(P, R.x, R.done, overflow, used_max, R.iteration);
        }
    }
    __syncthreads();
    if (s_done) return;                    // terminated by this launch: nothing to evaluate (the next launch finds `done` and reads no records)
    const q4 q{s_cand[3], s_cand[4], s_cand[5], s_cand[6]};
    const d3 t{s_cand[0], s_cand[1], s_cand[2]};
    bool valid = false;
    int mult = 1;
    Lin L;
    L.r = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) L.J[i] = 0.0;
    if (f < K.m && c.valid) {
        valid = true;
        mult = c.valid;
        const double w = feature_weight_pref(P, K, cdv);
        double R9[9];
        qtorot(q, R9);
        const d3 p{double(fp.x), double(fp.y), double(fp.z)};
        if (kind == MLH_SURF) eval_plane(p, c.c, w, q, t, R9, L);
        else eval_edge(p, c.c, w, q, t, R9, L);
    }
    reduce_rows(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, kind, s_red, P.partials + size_t(gtile) * NE_STRIDE, mult);
}

// ---- The whole Levenberg-Marquardt loop of one outer iteration in ONE launch. The consumer-side launches above still pay a launch boundary per LM iteration
// (~4.4 us of an ~8.9 us launch on the 88-tile mapper frame) and a look-ahead of launches behind the loop's end; the tiles of a frame that qualifies for the
// consumer-side schedule whose workgroups are all resident at once (the host asks the device: mlh_ctx::caps, capi.hip: set_loop_gates) can synchronise among themselves instead:
//   every workgroup: tile inputs -> registers (once); sum the fit launch's records; LM begin; then, until the loop terminates:
//     evaluate the tile at the candidate -> record (buffer (it + 1) & 1) -> grid barrier (release; one atomic arrival; spin on the counter; acquire)
//     -> sum all records -> LM step (the state stays in LDS: every workgroup runs the identical arithmetic on identical inputs, so they agree on accept / reject,
//     on the next candidate and on the iteration the loop ends at, without exchanging anything but the records).
// The loop ends on the device when Ceres' loop would: no look-ahead budget, no launches that find `done`, nothing for the host to poll between LM iterations,
// and the split submission cannot overflow. Same operations in the same order as the launches it replaces: the same bits.
// Barrier: P.ticket[1] counts arrivals (monotonic over the launch: iteration `it` waits for total * (it + 1)), P.ticket[2] counts workgroups that have left; the
// last one to leave zeroes both for the next launch. Residency is the HOST's business (capi.hip: loop_tiles_ok -- the occupancy query x the compute units the
// solver's stream may use, asked at mlh_create); should a barrier nevertheless not complete within P.loop_timeout_ticks of the 100 MHz wall clock (a workgroup that
// never became resident beside another context's kernels, a fault), the loop is given up: P.ticket[3] tells every workgroup still to come or still polling, bit 2 of
// the published `done` word tells the host, which solves the frame again through the launch-per-iteration form (lm_consume_kernel: no residency requirement) --
// a slow frame, not a lost one. The later loop launches of such a frame (lm_overflow == 4 in the state) leave at once.
// MLH_LOOP_COH 1: the records cross the barrier as agent-scope monotonic stores / loads (write-through, read past the L2 of the reader's XCD) -- no L2 write-back
// and invalidate around the barrier; 0: plain stores + release fence / acquire fence + plain loads
#ifndef MLH_LOOP_COH
#define MLH_LOOP_COH 1
#endif
#ifndef MLH_LOOP_SPLIT_STEP
#define MLH_LOOP_SPLIT_STEP 0
#endif
#ifndef MLH_LOOP_SLEEP
#define MLH_LOOP_SLEEP 1
#endif
// MLH_LOOP_KEEP_REGS 1 (round 6): the first wavefront keeps the LM state's registers from step to step (lm_step_wave_keep) instead of loading them from and storing
// them to the LDS state around every step: scan2map 0.184 -> 0.176 ms (three alternations: 0.1836 / 0.1844 / 0.1840 against 0.1761 / 0.1768 / 0.1755), the same bits
#ifndef MLH_LOOP_KEEP_REGS
#define MLH_LOOP_KEEP_REGS 1
#endif
// FIT (round 6): the launch begins with the outer iteration's FIT -- fit_linearize_kernel's body for this tile: neighbour records -> line / plane fit + gates -> the
// correspondence record (stored: later launches and callers read it) -> residual + Jacobian at the start pose -> the tile's record, tagged with iteration 0 and
// summed by polling like every other record of the loop -- instead of reading what a fit launch in front of it left. One launch boundary fewer per outer iteration;
// the same operations on the same values: the same bits. Tagged records only (P.loop_tagged: the host's loop_fit_fusable()).
template <bool DEVM = false, bool FIT = false>
__global__ __launch_bounds__(TPB, 2) void lm_loop_kernel(KParams P)      // (2: two workgroups per compute unit -- 256 registers in all; the residency gates count on them)
{
    __shared__ double s_red[4 * 32];
    __shared__ double f_ne[NE_STRIDE], f_scratch[(TPB / 32) * 32];
    __shared__ double s_cand[8];
    __shared__ int s_done, s_timeout;
#if MLH_LOOP_SPLIT_STEP
    __shared__ double s_gmax;
#endif
    __shared__ LmState s_lm;
    int m0 = P.k[0].m, m1 = P.k[1].m, tb0 = P.k[0].tiles_b;
    int total = P.k[0].tiles_b + P.k[1].tiles_b;
    if constexpr (DEVM) {            // (the counts from the device: the tiles that exist are the barrier's participants)
        m0 = P.m_dev[0]; m1 = P.m_dev[1];
        tb0 = (m0 + TPB - 1) / TPB;
        total = tb0 + (m1 + TPB - 1) / TPB;
        if (total == 0) {
            // the thinning kept nothing of either kind: no tile exists, nobody would publish -- the first workgroup of the launch the host waits for does (bit 3:
            // "no features"; the host reports what mlh_scan2map reports for an empty feature set, include/mloam_hip.h)
            if (blockIdx.x == 0 && threadIdx.x == 0 && P.publish) {
                for (int i = 0; i < 7; ++i) P.publish->x[i] = P.use_init ? P.init_pose[i] : P.state->x[i];
                P.publish->done = 1 | 8;
                P.publish->xb[2][0] = 0.0;
                __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            return;
        }
        if ((int(blockIdx.x) >> 3) >= ((total + 7) >> 3)) return;
    }
    const int gtile = xcd_tile(total);
    if (gtile >= total) return;            // (the grid is rounded up to a multiple of 8: the padding workgroups take no part in the barrier)
    const bool writer = gtile == 0;
    const int kind = gtile >= tb0 ? 1 : 0;
    const int tile = kind ? gtile - tb0 : gtile;
    const KindP &K = P.k[kind];
    const int m_feat = kind ? m1 : m0;
    const int f = tile * TPB + threadIdx.x;
    Corr c;
    c.valid = 0;
    float4 fp = make_float4(0.f, 0.f, 0.f, 0.f), cdv = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FIT) {
        // fit_linearize_kernel<5, .., FIN = false>'s body (single block, N_NEIGH 5, no ownership planes, no field-of-view gate: what scan2MapOptimization launches)
#pragma unroll
        for (int i = 0; i < 6; ++i) c.c[i] = 0.f;
        c.pad = 0;
        if (f < m_feat) {
            fp = K.feat[f];
            float4 nbv[5];
            const float4 *nb = K.nbr + size_t(f) * K.nbr_stride;
#pragma unroll
            for (int j = 0; j < 5; ++j) nbv[j] = nb[j];
            if ((P.flags & MLH_FLAG_WITH_UA) && K.covd) cdv = K.covd[f];
            float coef[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            bool ok = false;
            if (fp.w >= 0.f) ok = fit_feature<5, 5>(P, kind, nbv, coef);
#pragma unroll
            for (int i = 0; i < 6; ++i) c.c[i] = ok ? coef[i] : 0.f;
            c.valid = ok ? 1 : 0;
            K.corr[f] = c;
        }
    } else if (f < m_feat) {
        c = K.corr[f];
        fp = K.feat[f];
        if ((P.flags & MLH_FLAG_WITH_UA) && K.covd) cdv = K.covd[f];
    }
    const bool valid = f < m_feat && c.valid != 0;
    const int mult = valid ? c.valid : 1;
    const double w = feature_weight_pref(P, K, cdv);
    const d3 p{double(fp.x), double(fp.y), double(fp.z)};
    const size_t set = size_t(NE_STRIDE) * size_t(total);
    // given up already -- by a workgroup of this launch that waited in vain, or by an earlier loop of this frame: nothing to do but leave
    if (threadIdx.x == 0)
        s_timeout = (__hip_atomic_load(P.ticket + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || (P.lm_expect_done >= 0 && P.state->lm_overflow == 4)) ? 2 : 0;
    if constexpr (FIT) {
        __syncthreads();                               // s_timeout
        if (s_timeout == 0) {                          // (uniform)
            // the linearisation at the start pose, as the fit launch evaluates it (load_pose: the launch's pose argument or the state's)
            double xs[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) xs[i] = P.use_init ? P.init_pose[i] : P.state->x[i];
            const q4 q0{xs[3], xs[4], xs[5], xs[6]};
            const d3 t0{xs[0], xs[1], xs[2]};
            Lin L0;
            L0.r = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) L0.J[i] = 0.0;
            if (valid) {
                double R9[9];
                qtorot(q0, R9);
                if (kind == MLH_SURF) eval_plane(p, c.c, w, q0, t0, R9, L0);
                else eval_edge(p, c.c, w, q0, t0, R9, L0);
            }
            const unsigned tag0 = P.loop_tag_base;     // iteration byte 0: the fit's record (the loop's iterations carry 1, 2, ...)
            unsigned long long *trec0 = P.loop_tagged;  // set 0 (iteration `it` writes set (it + 1) & 1: 1, 0, 1, ...; set 0 is written again by iteration 1, which
                                                        // no workgroup reaches before every workgroup has stored its iteration-0 record -- behind its read of this one)
            reduce_rows<2>(valid, L0, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, kind, s_red, reinterpret_cast<double *>(trec0 + size_t(gtile) * 64), 1, tag0);
            lmc_sum_records_tagged(trec0, total, tag0, f_ne, f_scratch, P.ticket, P.loop_timeout_ticks, &s_timeout);
        }
    } else {
        lmc_sum_records(P.partials_in, total, f_ne, f_scratch);
    }
    const bool skipped = s_timeout != 0;           // (uniform: written before the sum's barriers)
    if (threadIdx.x == 0 && skipped) s_done = 1;
#if MLH_LOOP_KEEP_REGS
    LmRegs Rk;                     // (first wavefront only: the LM state's registers, kept from step to step; stored to s_lm once, behind the loop)
    double candk[7], x_cost_k = 0.0;
#endif
    if (threadIdx.x < 64 && !skipped) {
        const int lane = threadIdx.x;
        LmRegs R;
        double cand[7], x[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) x[i] = P.use_init ? P.init_pose[i] : P.state->x[i];
        lm_begin_wave_pp(f_ne, f_scratch, x, &s_lm, true, P.thre_b[0], P.lm_max_it, P.lm_min_blocks, R, cand);
        if (lane < 7) s_cand[lane] = pick7(cand, lane);
        if (lane == 0) s_done = R.done;
#if MLH_LOOP_KEEP_REGS
        Rk = R; x_cost_k = f_ne[NE_COST];
#pragma unroll
        for (int i = 0; i < 7; ++i) candk[i] = cand[i];
#endif
    }
    __syncthreads();
    int it = 0;
    while (!s_done) {
        if (it == 2) MLH_STAGE(gtile, 0);                 // (debug build only: the third iteration's stages, scripts/stageclock_loop.py)
        const q4 q{s_cand[3], s_cand[4], s_cand[5], s_cand[6]};
        const d3 t{s_cand[0], s_cand[1], s_cand[2]};
        Lin L;
        L.r = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) L.J[i] = 0.0;
        if (valid) {
            double R9[9];
            qtorot(q, R9);
            if (kind == MLH_SURF) eval_plane(p, c.c, w, q, t, R9, L);
            else eval_edge(p, c.c, w, q, t, R9, L);
        }
        if (it == 2) MLH_STAGE(gtile, 1);
        if (P.loop_tagged) {
            // (round 6) the record leaves as tagged words and is summed by polling for the tag: no arrival atomic, no counter poll between the store and the loads
            const unsigned tag = P.loop_tag_base | unsigned(it + 1);
            unsigned long long *trec = P.loop_tagged + size_t(total) * 64 * size_t((it + 1) & 1);
            const bool stall = P.debug_stall && it == 1 && gtile == (total > 1 ? 1 : 0);
            if (stall) {                                               // (tests: this workgroup's record never arrives)
                if (threadIdx.x == 0) { s_timeout = 1; __hip_atomic_store(P.ticket + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                __syncthreads();
            } else {
                reduce_rows<2>(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, kind, s_red, reinterpret_cast<double *>(trec + size_t(gtile) * 64), mult, tag);
                if (it == 2) MLH_STAGE(gtile, 2);
                if (it == 2) MLH_STAGE(gtile, 3);
                lmc_sum_records_tagged(trec, total, tag, f_ne, f_scratch, P.ticket, P.loop_timeout_ticks, &s_timeout);
            }
            if (s_timeout) break;
        } else {
            double *rec = P.partials + set * size_t((it + 1) & 1);
            reduce_rows<MLH_LOOP_COH != 0>(valid, L, P.huber_delta, (P.flags & MLH_FLAG_NO_LOSS) != 0, kind, s_red, rec + size_t(gtile) * NE_STRIDE, mult);
            // (the record's 32 words are stored by the first 32 lanes of the wavefront thread 0 belongs to: its s_waitcnt covers them)
            if (!MLH_LOOP_COH) __syncthreads();
            if (it == 2) MLH_STAGE(gtile, 2);
            if (threadIdx.x == 0) {
                if (!MLH_LOOP_COH) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned target = unsigned(total) * unsigned(it + 1);
                const bool stall = P.debug_stall && it == 1 && gtile == (total > 1 ? 1 : 0);
                if (stall) { s_timeout = 1; __hip_atomic_store(P.ticket + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                else {
                    __hip_atomic_fetch_add(P.ticket + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    unsigned spins = 0;
                    long long t0 = 0;
                    while (__hip_atomic_load(P.ticket + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                        if (MLH_LOOP_SLEEP) __builtin_amdgcn_s_sleep(MLH_LOOP_SLEEP);
                        if ((++spins & 63u) == 0u) {      // (a completed barrier takes ~2 us = a few polls: the clock and the word below are only read by a wait that is already long)
                            const long long now = wall_clock64();
                            if (t0 == 0) t0 = now;
                            const bool late = (unsigned long long)(now - t0) > P.loop_timeout_ticks;
                            if (late) __hip_atomic_store(P.ticket + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (late || __hip_atomic_load(P.ticket + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { s_timeout = 1; break; }
                        }
                    }
                }
                if (!MLH_LOOP_COH) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("" ::: "memory");
            }
            __syncthreads();
            if (it == 2) MLH_STAGE(gtile, 3);
            if (s_timeout) break;
            lmc_sum_records<MLH_LOOP_COH != 0>(rec, total, f_ne, f_scratch);
        }
        if (it == 2) MLH_STAGE(gtile, 4);
#if MLH_LOOP_SPLIT_STEP
        {
            // the LM step on the first wavefront, the gradient max-norm an ACCEPTED step needs on the second, side by side (lm_step_wave_spec1 / 2)
            LmRegs R;
            double cand[7];
            LmStepSpec sp;
            if (threadIdx.x < 64) lm_step_wave_spec1(f_ne, &s_lm, P.lm_max_it, R, cand, sp);
            else if (threadIdx.x < 128) {
                LmRegs Q;
#pragma unroll
                for (int i = 0; i < 7; ++i) Q.x[i] = s_lm.cand[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) Q.g[i] = f_ne[NE_G + i];
                const double gm = gradient_max_norm_wave(Q, s_lm.V, threadIdx.x & 63);
                if (threadIdx.x == 64) s_gmax = gm;
            }
            __syncthreads();
            if (threadIdx.x < 64) {
                const int lane = threadIdx.x;
                lm_step_wave_spec2(&s_lm, &s_lm, true, R, cand, sp, s_gmax);
                if (lane < 7) s_cand[lane] = pick7(cand, lane);
                if (lane == 0) s_done = R.done;
            }
        }
#elif MLH_LOOP_KEEP_REGS
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            lm_step_wave_keep(f_ne, &s_lm, P.lm_max_it, Rk, candk, x_cost_k);
            if (lane < 7) s_cand[lane] = pick7(candk, lane);
            if (lane == 0) s_done = Rk.done;
        }
#else
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            LmRegs R;
            double cand[7];
            lm_step_wave_pp(f_ne, &s_lm, &s_lm, true, P.lm_max_it, R, cand);
            if (lane < 7) s_cand[lane] = pick7(cand, lane);
            if (lane == 0) s_done = R.done;
        }
#endif
        __syncthreads();
        if (it == 2) MLH_STAGE(gtile, 5);
        ++it;
    }
#if MLH_LOOP_KEEP_REGS
    if (threadIdx.x < 64 && !skipped) {            // the state the publication below (and nothing else) reads
        lm_state_store_pp(Rk, candk, s_lm.ne, s_lm.V, &s_lm, threadIdx.x);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
#endif
    if (writer && threadIdx.x < 64 && skipped) {
        // the loop never began here: the pose in the state is what the last loop that ran left; the failure travels on to the launch that publishes
        if (threadIdx.x == 0) {
            P.state->lm_overflow = 4;
            if (P.publish) {
                for (int i = 0; i < 7; ++i) P.publish->x[i] = P.state->x[i];
                P.publish->done = 4;
                P.publish->xb[2][0] = P.state->lm_used_max;
                __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    else if (writer && threadIdx.x < 64) {
        const int lane = threadIdx.x;
        if (lane < 7) P.state->x[lane] = s_lm.x[lane];          // read by the launches BEHIND this one only (the other workgroups took their start pose long ago)
        if (lane == 0) {
            const double used = P.lm_expect_done < 0 ? double(s_lm.iteration) : fmax(P.state->lm_used_max, double(s_lm.iteration));
            // (a barrier given up on in an EARLIER outer iteration of the frame must not be lost: lm_overflow -- unused otherwise by this schedule -- carries it to the
            // launch that publishes)
            const int failed = (s_timeout ? 1 : 0) | ((P.lm_expect_done >= 0 && P.state->lm_overflow == 4) ? 1 : 0);
            P.state->lm_used_max = used;
            P.state->lm_overflow = failed ? 4 : 0;
            P.state->done = s_lm.done;
            P.state->iteration = s_lm.iteration;
            if (P.publish) {
                for (int i = 0; i < 7; ++i) P.publish->x[i] = s_lm.x[i];
                P.publish->done = (s_lm.done ? 1 : 0) | (failed ? 4 : 0);
                P.publish->xb[2][0] = used;
                __hip_atomic_store(&P.publish->seq, P.publish_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (threadIdx.x == 0) {                // the last workgroup to leave re-arms the barrier for the next launch
        const unsigned left = atomicAdd(P.ticket + 2, 1u);
        if (left == unsigned(total - 1)) {
            __hip_atomic_store(P.ticket + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.ticket + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(P.ticket + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// stand-alone exact 5-NN for mlh_knn (queries already in the map frame)
__global__ __launch_bounds__(TPB) void knn_queries_kernel(GridDev grid, const float *__restrict__ q, int nq, int *__restrict__ idx,
                                                          float *__restrict__ d2)
{
    constexpr int G = 8, FPB = TPB / G;
    __shared__ int s_run[FPB * 20];
    const int grp = threadIdx.x / G, gl = threadIdx.x % G;
    const int qi = blockIdx.x * FPB + grp;
    if (qi >= nq) return;
    unsigned long long keys[5];
    knn_group<5, G, true>(grid, q[qi * 3 + 0], q[qi * 3 + 1], q[qi * 3 + 2], gl, s_run + grp * 20, keys);
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
// lanes per query, per kind. A small launch (a 16-ring frame) leaves most SIMDs idle and is bound by the latency of its trips: 16 lanes.
// A full frame keeps ~5 wavefronts per SIMD busy and is bound by VALU issue, where the per-query instruction count is what matters:
// 8 lanes (twice the queries per wavefront) for a kind whose map is sparse enough that a query's candidates fit one or two 8-lane trips,
// 16 lanes (pruned, near-cells-first) for a dense map, whose heavy queries would otherwise set the launch's duration. Measured up to 229 k
// queries per launch (un-thinned features) and on config 4's 52 k: the density rule beats "8 lanes for every kind" there too (44.5 vs 48.6 us,
// 0.325 vs 0.356 ms), so there is no upper query count at which it is switched off.
void knn_lanes_for(const mlh_ctx *ctx, int kind_mask, int lanes[2])
{
    long long queries = 0;
    for (int k = 0; k < 2; ++k) if (kind_mask & (1 << k)) queries += ctx->feat[k].m;
    for (int k = 0; k < 2; ++k) {
        const MapGrid &mg = ctx->map[k];
        // ~9 of the 27 cells around a query on a surface are occupied, each about as full as the cell an average map point lives in
        // (pop_sq / n, size-biased: the clusters of a dense edge map count by their points, not by their cells)
        const double est27 = (mg.occupied > 0 && mg.n > 0) ? 9.0 * double(mg.pop_sq) / double(mg.n) : 1e9;
        lanes[k] = queries <= KNN_LATENCY_LIMIT ? 16 : (est27 < double(KNN_TWO_PHASE_MIN) ? 8 : 16);
        if (ctx->knn_lanes_override == 8 || ctx->knn_lanes_override == 16 || ctx->knn_lanes_override == 32) lanes[k] = ctx->knn_lanes_override;
        if (ctx->knn_lanes_override > 99) {          // "SSCC": surf lanes, corner lanes (816, 832, 1632, ...)
            const int v = k == 0 ? ctx->knn_lanes_override / 100 : ctx->knn_lanes_override % 100;
            if (v == 8 || v == 16 || v == 32) lanes[k] = v;
        }
    }
}

static int fill_params(mlh_ctx *ctx, const MatchArgs &a, KParams &P)
{
    // a solve whose last iteration is still a set of tile records (mlh_ctx::gn_pending): any launch that writes records -- other than the chained successor's first,
    // which consumes them -- completes that solve first
    if (ctx->gn_pending.active && !a.pre_final) { const int frc = gn_flush_pending(ctx); if (frc) return frc; }
    std::memset(&P, 0, sizeof(P));
    int tiles_b_total = 0;
    P.n_blocks = a.n_blocks > 0 ? a.n_blocks : 1;
    int kmax = 5;
    for (int b = 0; b < P.n_blocks; ++b) {
        P.kb[b] = a.k_neigh[b] == 10 ? 10 : 5;
        P.thre_b[b] = a.eig_thre[b];
        P.freeze_b[b] = a.freeze[b];
        kmax = std::max(kmax, P.kb[b]);
    }
    int lanes[2];
    knn_lanes_for(ctx, a.kind_mask, lanes);
    {
        const bool both = (a.kind_mask & 3) == 3;
        P.knn_lanes = both ? (lanes[0] == lanes[1] ? lanes[0] : 0) : lanes[(a.kind_mask & 1) ? 0 : 1];
        if (P.knn_lanes == 32) P.knn_lanes = 0;      // (32 lanes per query exist in the per-kind kernels only)
    }
    for (int k = 0; k < 2; ++k) {
        KindP &K = P.k[k];
        if (!(a.kind_mask & (1 << k))) continue;
        FeatSet &fs = ctx->feat[k];
        MapGrid &mg = ctx->map[k];
        if (!mg.built) return fail(ctx, MLH_ERR_STATE, "map_set has not been called for this kind");
        if (fs.m <= 0) return fail(ctx, MLH_ERR_STATE, "features_set has not been called for this kind");
        if (fs.n_blocks != P.n_blocks) return fail(ctx, MLH_ERR_STATE, "the staged features hold a different number of pose blocks than requested");
        // the cell edge was derived from the acceptance radius given at map_set; a larger radius here would break exactness
        if (a.min_match_sq_dis > 0.f && std::sqrt(a.min_match_sq_dis) > mg.h)
            return fail(ctx, MLH_ERR_INVALID, "min_match_sq_dis exceeds the value the map grid was built for");
        hipError_t e;
        if ((e = fs.corr.ensure(sizeof(Corr) * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc corr", e);
        if (fs.nbr_stride < kmax) { fs.nbr.release(); fs.nbr_stride = kmax; }
        if ((e = fs.nbr.ensure(sizeof(float4) * size_t(fs.nbr_stride) * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc nbr", e);
        if (a.dense) {
            if ((e = fs.r.ensure(sizeof(double) * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc r", e);
            if ((e = fs.J.ensure(sizeof(double) * 6 * size_t(fs.m))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc J", e);
        }
        K.grid = mg.dev();
        K.feat = fs.pts.as<float4>();
        K.covd = fs.has_cov ? fs.covd.as<float4>() : nullptr;
        K.nbr = fs.nbr.as<float4>();
        K.corr = fs.corr.as<Corr>();
        K.r_out = a.dense ? fs.r.as<double>() : nullptr;
        K.J_out = a.dense ? fs.J.as<double>() : nullptr;
        K.m = fs.m;
        K.lanes = lanes[k];
        K.tiles_a = (fs.m + TPB / K.lanes - 1) / (TPB / K.lanes);
        K.tiles_b = (fs.m + TPB - 1) / TPB;
        K.nbr_stride = fs.nbr_stride;
        for (int b = 0; b <= MAX_BLOCKS; ++b) K.blk_start[b] = fs.blk_start[std::min(b, fs.n_blocks)];
        tiles_b_total += K.tiles_b;
    }
    if (tiles_b_total == 0) return fail(ctx, MLH_ERR_STATE, "no map/features staged for the requested kinds");
    hipError_t e;
    // (two sets of records: the consumer-side Levenberg-Marquardt launches alternate between them)
    if ((e = ctx->partials.ensure(sizeof(double) * NE_STRIDE * size_t(tiles_b_total) * 2)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc partials", e);
    if ((e = ensure_ticket(ctx)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc ticket", e);
    ctx->n_partial_tiles = tiles_b_total;
    P.partials = ctx->partials.as<double>();
    P.state = ctx->state.as<SolverState>();
    P.pose_sel = a.pose_sel;
    P.flags = a.flags;
    P.min_match_sq_dis = a.min_match_sq_dis;
    P.min_plane_dis = a.min_plane_dis;
    P.huber_delta = a.huber_delta;
    P.cov_measurement_trace = a.cov_measurement_trace;
    P.has_lo = ctx->shard_lo ? 1 : 0;
    P.has_hi = ctx->shard_hi ? 1 : 0;
    P.own_mod = ctx->own_mod; P.own_rem = ctx->own_rem;
    for (int i = 0; i < 4; ++i) { P.lo[i] = ctx->lo_plane[i]; P.hi[i] = ctx->hi_plane[i]; }
    P.finish = a.finish;
    P.lm_max_it = a.lm_max_it; P.lm_min_blocks = a.lm_min_blocks; P.lm_expect_done = a.lm_expect_done;
    // the mailbox communicator rides in the finishing workgroup of a Gauss-Newton launch (finish == 1) and of an LM begin / step launch (3 / 4); other launches
    // exchange nothing
    if ((a.finish == 1 || a.finish == 3 || a.finish == 4) && ctx->p2p.active) p2p_fill(ctx, P.p2p);
    else { P2pDev none{}; none.n_ranks = 1; P.p2p = none; }
    P.use_init = a.init_pose ? 1 : 0;
    for (int i = 0; i < 7; ++i) P.init_pose[i] = a.init_pose ? a.init_pose[i] : 0.0;
    P.warm = a.warm ? 1 : 0;
    if (a.gn_iter >= 1 && a.gn_blocks) {
        // iteration i >= 1 of a deferred-finish solve over pose blocks: every block's pose of iteration i in SolverState::xib[i & 1][b]; iteration 1 updates the poses
        // the state holds (x, xb[b])
        SolverState *S = ctx->state.as<SolverState>();
        P.pre_finish = 1;
        P.pre_tiles = tiles_b_total;
        P.pre_from_init = 0;
        P.pre_from_state = a.gn_iter == 1 ? 1 : 0;
        P.x_prev = &S->xib[(a.gn_iter - 1) & 1][0][0];
        P.x_next = &S->xib[a.gn_iter & 1][0][0];
        P.pose0 = &S->xib[a.gn_iter & 1][0][0];
        P.use_init = 0;
    } else if (a.gn_iter >= 1) {
        // iteration i >= 1 of a deferred-finish solve: the correspondence kernel turns iteration i - 1's records and pose into pose i (SolverState::xi[base + (i & 1)]);
        // iteration 1 finds pose 0 where iteration 0 found it -- the kernel arguments, the state's x (a chained solve behind a chain launch), or iteration 0's slot
        // (a chained solve whose first launch computed the start pose itself, MatchArgs::pre_final)
        SolverState *S = ctx->state.as<SolverState>();
        P.pre_finish = 1;
        P.pre_tiles = tiles_b_total;
        P.pre_from_init = (a.gn_iter == 1 && a.init_pose) ? 1 : 0;
        P.x_prev = (a.gn_iter == 1 && !a.pre_final) ? S->x : S->xi[a.gn_slot_base + ((a.gn_iter - 1) & 1)];
        P.x_next = S->xi[a.gn_slot_base + (a.gn_iter & 1)];
        P.pose0 = S->xi[a.gn_slot_base + (a.gn_iter & 1)];
        P.use_init = 0;
    } else if (a.gn_iter == 0 && a.pre_final) {
        // iteration 0 of a chained solve that completes its predecessor first (knn_features_kernel<.., PRE = 2>): the predecessor's records, thresholds, last pose slot
        // and host record; this frame's start pose goes to xi[base] (the fit kernel of this iteration and iteration 1's prologue read it there)
        SolverState *S = ctx->state.as<SolverState>();
        P.pre_finish = 2;
        P.pre_tiles = a.pre_final_tiles;
        P.pre_from_init = 0;
        P.x_prev = S->xi[a.pre_final_slot];
        P.x_next = S->xi[a.gn_slot_base];
        P.pose0 = S->xi[a.gn_slot_base];
        P.use_init = 0;
        P.pre_publish = a.pre_final_publish;
        P.pre_publish_seq = a.pre_final_seq;
        P.pre_thre = a.pre_final_thre;
        P.pre_freeze = a.pre_final_freeze;
        for (int i = 0; i < 7; ++i) { P.chain_prev[i] = a.chain_prev[i]; P.chain_cur[i] = a.chain_cur[i]; }
    }
    P.m_dev = a.m_dev;
    P.publish = (a.finish == 1 || a.finish == 4 || a.lmc) ? a.publish : nullptr;
    if (a.lmc) {
        if (!ctx->lm_pp.p) {
            if ((e = ctx->lm_pp.ensure(2 * sizeof(LmState))) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "alloc lm state", e);
            if ((e = hipMemsetAsync(ctx->lm_pp.p, 0, 2 * sizeof(LmState), ctx->stream)) != hipSuccess) return fail(ctx, MLH_ERR_HIP, "memset lm state", e);
        }
        const unsigned long long g = ++ctx->lmc_count;
        P.lm_in = ctx->lm_pp.as<LmState>() + ((g - 1) & 1);
        P.lm_out = ctx->lm_pp.as<LmState>() + (g & 1);
        const size_t set = size_t(NE_STRIDE) * size_t(tiles_b_total);
        P.partials_in = ctx->partials.as<double>() + set * size_t((a.lmc_j - 1) & 1);
        P.partials = ctx->partials.as<double>() + set * size_t(a.lmc_j & 1);
        if (a.lmc == 3) { P.partials_in = ctx->partials.as<double>(); P.partials = ctx->partials.as<double>(); }     // (the loop kernel alternates by itself)
    }
    P.publish_seq = a.publish_seq;
    P.ticket = ctx->ticket.as<unsigned>();
    P.stat = (a.stat_slot >= 0) ? ctx->stats.as<IterStatDev>() + a.stat_slot : nullptr;
    return MLH_OK;
}

template <typename Kern>
static void launch_timed(mlh_ctx *ctx, int kid, Kern kern, int grid, const KParams &P)
{
    hipEvent_t a = nullptr, b = nullptr;
    if (prof_kernel_events(ctx, kid, &a, &b)) {
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(TPB), 0, ctx->stream, a, b, 0, P);   // start/stop = the dispatch's own timestamps
        if (launch_check_enabled()) launch_check("correspondence / fit / linearise kernel (event-bracketed launch)");
    } else
        MLH_LAUNCH(kern, dim3(grid), dim3(TPB), 0, ctx->stream, P);
}

int gn_flush_pending(mlh_ctx *ctx)
{
    if (!ctx || !ctx->gn_pending.active) return MLH_OK;
    KParams P;
    std::memset(&P, 0, sizeof(P));
    SolverState *S = ctx->state.as<SolverState>();
    P.partials = ctx->partials.as<double>();
    P.state = S;
    P.pre_finish = 2;
    P.pre_tiles = ctx->gn_pending.tiles;
    P.x_prev = S->xi[ctx->gn_pending.slot];
    P.pre_publish = static_cast<HostPublish *>(ctx->gn_pending.rec);
    P.pre_publish_seq = ctx->gn_pending.seq;
    P.pre_thre = ctx->gn_pending.thre;
    P.pre_freeze = ctx->gn_pending.freeze;
    ctx->gn_pending.active = false;
    launch_timed(ctx, MLH_K_SOLVE, gn_final_kernel, 1, P);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int match_launch(mlh_ctx *ctx, const MatchArgs &a)
{
    KParams P;
    int rc = fill_params(ctx, a, P);
    if (rc) return rc;
    // kernel A: correspondences (8 or 16 lanes per feature); kernel B: fit + linearise + reduce (one lane per feature)
    const int grid_a = ((P.k[0].tiles_a + P.k[1].tiles_a + 7) / 8) * 8;
    const int grid_b = ((P.k[0].tiles_b + P.k[1].tiles_b + 7) / 8) * 8;
    const bool mb = P.n_blocks > 1;
    bool k10 = false;
    for (int b = 0; b < P.n_blocks; ++b) k10 = k10 || P.kb[b] == 10;
    if (P.m_dev) {
        // the feature counts are on the device only (mlh_downsample_scan2map): per-kind lanes, both kinds, one block, K = 5, records only
        if (mb || k10 || P.finish != 0 || P.pre_finish || (a.kind_mask & 3) != 3) return fail(ctx, MLH_ERR_UNSUPPORTED, "device-side feature counts: one block, N_NEIGH 5, both kinds, records only");
        if (P.warm) launch_timed(ctx, MLH_K_KNN, knn_features_kernel<0, false, false, 0, true, true>, grid_a, P);
        else launch_timed(ctx, MLH_K_KNN, knn_features_kernel<0, false, false, 0, false, true>, grid_a, P);
        if (!a.no_fit) launch_timed(ctx, MLH_K_FIT, fit_linearize_kernel<5, false, false, true>, grid_b, P);
        MLH_HIP(ctx, hipGetLastError());
        for (int k = 0; k < 2; ++k) ctx->feat[k].matched = !a.no_fit;       // (no_fit: the loop launch behind this one writes the correspondences)
        return MLH_OK;
    }
    {
        // <lanes, pose blocks, any block with K = 10>
#define MLH_KNN_LAUNCH(G_) do { \
            if (mb) { if (k10) launch_timed(ctx, MLH_K_KNN, knn_features_kernel<G_, true, true>, grid_a, P); else launch_timed(ctx, MLH_K_KNN, knn_features_kernel<G_, true, false>, grid_a, P); } \
            else { if (k10) launch_timed(ctx, MLH_K_KNN, knn_features_kernel<G_, false, true>, grid_a, P); else launch_timed(ctx, MLH_K_KNN, knn_features_kernel<G_, false, false>, grid_a, P); } \
        } while (0)
#define MLH_KNN_LAUNCH_GN(G_) do { \
            if (P.pre_finish == 2) launch_timed(ctx, MLH_K_KNN_FIRST, knn_features_kernel<G_, false, false, 2, false>, grid_a, P); \
            else if (P.pre_finish && P.warm) launch_timed(ctx, MLH_K_KNN_PRE, knn_features_kernel<G_, false, false, 1, true>, grid_a, P); \
            else if (P.pre_finish) launch_timed(ctx, MLH_K_KNN_PRE, knn_features_kernel<G_, false, false, 1, false>, grid_a, P); \
            else launch_timed(ctx, MLH_K_KNN, knn_features_kernel<G_, false, false, 0, true>, grid_a, P); \
        } while (0)
#define MLH_KNN_LAUNCH_GN_MB(G_) do { \
            if (k10) launch_timed(ctx, MLH_K_KNN_PRE, knn_features_kernel<G_, true, true, 1, true>, grid_a, P); \
            else launch_timed(ctx, MLH_K_KNN_PRE, knn_features_kernel<G_, true, false, 1, true>, grid_a, P); \
        } while (0)
        if ((P.pre_finish || P.warm) && (mb || k10 || a.gn_blocks)) {     // (a.gn_blocks: a blocks solve over ONE block keeps the blocks' pose slots)
            // a solve over pose blocks (or with N_NEIGH = 10): the finish in the consumer and the bounded search come together or not at all
            if (!(P.pre_finish == 1 && P.warm)) return fail(ctx, MLH_ERR_UNSUPPORTED, "pose blocks: the deferred finish and the bounded search are one schedule");
            if (P.knn_lanes == 0) MLH_KNN_LAUNCH_GN_MB(0);
            else if (P.knn_lanes == 16) MLH_KNN_LAUNCH_GN_MB(16);
            else MLH_KNN_LAUNCH_GN_MB(8);
        }
        else if (P.pre_finish || P.warm) {
            if (P.knn_lanes == 0) MLH_KNN_LAUNCH_GN(0);
            else if (P.knn_lanes == 16) MLH_KNN_LAUNCH_GN(16);
            else MLH_KNN_LAUNCH_GN(8);
        }
        else if (P.knn_lanes == 0) MLH_KNN_LAUNCH(0);
        else if (P.knn_lanes == 16) MLH_KNN_LAUNCH(16);
        else MLH_KNN_LAUNCH(8);
#undef MLH_KNN_LAUNCH_GN
#undef MLH_KNN_LAUNCH_GN_MB
#undef MLH_KNN_LAUNCH
    }
    if (a.no_fit) {
        // the fit rides in the loop launch behind this one (lm_consume_launch, fit_in_loop): single block, N_NEIGH 5, both kinds, records only
        if (k10 || P.n_blocks != 1 || P.finish != 0 || (a.kind_mask & 3) != 3 || a.dense) return fail(ctx, MLH_ERR_UNSUPPORTED, "match_launch without its fit: single block, N_NEIGH 5, both kinds, records only");
        MLH_HIP(ctx, hipGetLastError());
        for (int k = 0; k < 2; ++k) ctx->feat[k].matched = false;
        return MLH_OK;
    }
    if (P.finish == 3) {
        if (k10 || P.n_blocks != 1) return fail(ctx, MLH_ERR_UNSUPPORTED, "the fused Levenberg-Marquardt begin is single-block, N_NEIGH = 5");
        launch_timed(ctx, MLH_K_FIT, fit_linearize_kernel<5, true>, grid_b, P);
    } else if (k10 && P.finish == 0) launch_timed(ctx, MLH_K_FIT, fit_linearize_kernel<10, false, false>, grid_b, P);
    else if (k10) launch_timed(ctx, MLH_K_FIT, fit_linearize_kernel<10, false>, grid_b, P);
    else if (P.finish == 0) launch_timed(ctx, MLH_K_FIT, fit_linearize_kernel<5, false, false>, grid_b, P);
    else launch_timed(ctx, MLH_K_FIT, fit_linearize_kernel<5, false>, grid_b, P);
    MLH_HIP(ctx, hipGetLastError());
    for (int k = 0; k < 2; ++k) if (a.kind_mask & (1 << k)) ctx->feat[k].matched = true;
    return MLH_OK;
}

int linearize_launch(mlh_ctx *ctx, const MatchArgs &a)
{
    for (int k = 0; k < 2; ++k)
        if ((a.kind_mask & (1 << k)) && !ctx->feat[k].matched) return fail(ctx, MLH_ERR_STATE, "linearize needs a previous match of this kind");
    KParams P;
    int rc = fill_params(ctx, a, P);
    if (rc) return rc;
    const int grid_b = ((P.k[0].tiles_b + P.k[1].tiles_b + 7) / 8) * 8;
    if (P.finish == 3 || P.finish == 4) launch_timed(ctx, MLH_K_LINEARIZE, linearize_kernel<true>, grid_b, P);
    else launch_timed(ctx, MLH_K_LINEARIZE, linearize_kernel<false>, grid_b, P);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_consume_launch(mlh_ctx *ctx, const MatchArgs &a)
{
    for (int k = 0; k < 2; ++k)
        if ((a.kind_mask & (1 << k)) && !ctx->feat[k].matched && !a.fit_in_loop) return fail(ctx, MLH_ERR_STATE, "the Levenberg-Marquardt launches need a previous match of this kind");
    if (a.lmc < 1 || a.lmc > 3 || a.lmc_j < 1 || a.n_blocks > 1 || a.dense) return fail(ctx, MLH_ERR_INVALID, "lm_consume_launch: single block, no dense rows");
    if (a.fit_in_loop && (a.lmc != 3 || (a.kind_mask & 3) != 3 || !loop_fit_fusable(a))) return fail(ctx, MLH_ERR_INVALID, "the fit rides in the one-launch loop with tagged records only");
    KParams P;
    int rc = fill_params(ctx, a, P);
    if (rc) return rc;
    if (P.p2p.n_ranks > 1) return fail(ctx, MLH_ERR_UNSUPPORTED, "the consumer-side Levenberg-Marquardt schedule is single-GPU");
    const int grid_b = ((P.k[0].tiles_b + P.k[1].tiles_b + 7) / 8) * 8;
    if (a.lmc == 3) {
        { const char *e = std::getenv("MLH_DEBUG_LOOP_STALL"); P.debug_stall = (e && std::atoi(e) != 0) ? 1 : 0; }
        // every tile's workgroup has to be resident for the barrier: the host's gate (capi.hip: loop_tiles_ok) is what the device admits, asked at mlh_create
        if (P.k[0].tiles_b + P.k[1].tiles_b > ctx->caps.loop_max_tiles[P.m_dev ? 1 : 0]) return fail(ctx, MLH_ERR_INVALID, "lm_loop_kernel: more tiles than can be resident at once on this device");
        P.loop_timeout_ticks = ctx->caps.loop_timeout_ticks;
        if (a.lm_expect_done < 0) ++ctx->caps.loop_launches;      // (the first loop of a frame)
        // the iterations' records as tagged words summed by polling (MLH_LOOP_TAGGED=0: plain records behind a grid barrier, as through round 5)
        { hipError_t e = loop_tagged_arm(ctx, size_t(P.k[0].tiles_b + P.k[1].tiles_b), a.lm_max_it, &P.loop_tagged, &P.loop_tag_base); if (e != hipSuccess) return fail(ctx, MLH_ERR_HIP, "tagged records", e); }
        if (a.fit_in_loop) {
            if (!P.loop_tagged) return fail(ctx, MLH_ERR_INVALID, "the fit rides in the one-launch loop with tagged records only");
            if (P.m_dev) launch_timed(ctx, MLH_K_LINEARIZE, lm_loop_kernel<true, true>, grid_b, P);
            else launch_timed(ctx, MLH_K_LINEARIZE, lm_loop_kernel<false, true>, grid_b, P);
            for (int k = 0; k < 2; ++k) ctx->feat[k].matched = true;
        }
        else if (P.m_dev) launch_timed(ctx, MLH_K_LINEARIZE, lm_loop_kernel<true>, grid_b, P);
        else launch_timed(ctx, MLH_K_LINEARIZE, lm_loop_kernel<false>, grid_b, P);
    }
    else if (a.lmc == 1) launch_timed(ctx, MLH_K_LINEARIZE, lm_consume_kernel<true>, grid_b, P);
    else launch_timed(ctx, MLH_K_LINEARIZE, lm_consume_kernel<false>, grid_b, P);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

bool loop_fit_fusable(const MatchArgs &loop_args)
{
    const char *env = std::getenv("MLH_LOOP_TAGGED"), *env_fit = std::getenv("MLH_LOOP_FIT");      // (A/B switches, read at every call: MLH_LOOP_FIT=0 keeps the fit launch)
    if ((env && std::atoi(env) == 0) || (env_fit && std::atoi(env_fit) == 0)) return false;
    return loop_args.lm_max_it <= 200 && loop_args.n_blocks == 1 && !loop_args.dense;
}

int lm_loop_occupancy(int blocks_per_cu[2])
{
    // (the forms with and without the fit in front: the smaller number gates both)
    int with_fit[2] = {0, 0};
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu[0], lm_loop_kernel<false>, TPB, 0) != hipSuccess) return MLH_ERR_HIP;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu[1], lm_loop_kernel<true>, TPB, 0) != hipSuccess) return MLH_ERR_HIP;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&with_fit[0], lm_loop_kernel<false, true>, TPB, 0) != hipSuccess) return MLH_ERR_HIP;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&with_fit[1], lm_loop_kernel<true, true>, TPB, 0) != hipSuccess) return MLH_ERR_HIP;
    for (int i = 0; i < 2; ++i) blocks_per_cu[i] = std::min(blocks_per_cu[i], with_fit[i]);
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
    const int grid = (nq + TPB / 8 - 1) / (TPB / 8);
    MLH_LAUNCH(knn_queries_kernel, dim3(grid), dim3(TPB), 0, ctx->stream, mg.dev(), ctx->knn_q.as<float>(), nq,
                       ctx->knn_idx.as<int>(), ctx->knn_d.as<float>());
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(idx, ctx->knn_idx.p, sizeof(int) * 5 * size_t(nq), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(d2, ctx->knn_d.p, sizeof(float) * 5 * size_t(nq), hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return MLH_OK;
}

}  // namespace mlh
