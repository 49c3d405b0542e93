// Internal definitions shared by the translation units of libmloam_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <string>
#include <vector>
#include <cstdlib>
#include <sched.h>
#include "../../include/mloam_hip.h"

namespace mlh {

// ---------------------------------------------------------------- device-side records
// local-map cell grid (dense, x fastest). Cell edge h is a hair above sqrt(min_match_sq_dis) so that every map point
// within the acceptance radius of a query lies in the query's 27-cell neighbourhood.
struct GridDev {
    const float4 *sorted;     // cell-sorted points {x,y,z, original index (int bits)}
    const float4 *raw;        // the same points in original order
    const int *cell_start;    // ncell + 1 exclusive prefix of per-cell counts
    float ox, oy, oz, inv_h;
    int nx, ny, nz, n;
};

// what the match kernel keeps per feature for later re-linearisation (LM iterations on fixed correspondences)
struct __attribute__((aligned(16))) Corr {
    float c[6];        // surf: n_hat(3), d, 0, 0 ; corner: X1(3), X2(3)   (all exactly f32-valued in the reference)
    int valid;
    int pad;           // the f64 weight is recomputed from the feature's covariance diagonal on every evaluation
};

// packed normal equations: 21 upper-triangular J^T J entries, 6 J^T r, cost, count; padded to 32
constexpr int NE_H = 0, NE_G = 21, NE_COST = 27, NE_CNT = 28, NE_STRIDE = 32;

// device-resident optimiser state (one per context)
struct SolverState {
    double x[7];             // current pose [t, q(xyzw)]
    double cand[7];          // candidate pose (LM)
    double xb[8][7];         // poses of the additional blocks (config 4: extrinsics); xb[0] unused (block 0 is x)
    double V[36];            // PoseLocalParameterization::V_update_
    double ne[NE_STRIDE];    // normal equations at x
    double ce[NE_STRIDE];    // normal equations at cand (multi-GPU: all-reduced before lm_step consumes them)
    double neb[8][NE_STRIDE];   // multi-GPU pose-block mode: one record per block, all-reduced together
    double diag[6];          // LM diagonal (Jacobi-scaled)
    double S[6];             // Jacobi scaling 1/(1+sqrt(H_ii)) from iteration zero
    double radius, decrease_factor;
    double model_cost_change;
    double gmax;
    int reuse_diagonal;
    int iteration;
    int done;
    int termination;
    int num_successful;
    int num_invalid;
    int evaluations;
    int lm_overflow;         // split-submission scan2map: an outer iteration was begun while the previous one's LM loop had not terminated inside its look-ahead budget
    // Gauss-Newton with the finish deferred to the consumer (match.hip): iteration i's pose of a solve lives in xi[base + (i & 1)] -- written by ONE workgroup of
    // iteration i's correspondence launch while the others may still be reading xi[base + ((i - 1) & 1)]; consecutive solves alternate base 0 / 2 (a solve's first
    // launch may still be reading the previous solve's last slot while it fills its own first one)
    double xi[4][7];
    double xib[2][8][7];     // the same for a solve over several pose blocks (mlh_gn_solve_blocks): iteration i's pose of block b in xib[i & 1][b]
    double lm_used_max;      // split-submission scan2map: the largest LM iteration count of the solve's outer iterations so far (the host sizes the next frame's look-ahead by it)
    double pad2[1];
};

// The Levenberg-Marquardt part of the solver state, in the form the consumer-side schedule keeps it (match.hip: lm_consume_kernel): two of these, launch g reads
// [(g - 1) & 1] and ONE workgroup of it writes [g & 1] -- the other workgroups of launch g may still be reading the first while that one stores the second
struct LmState {
    double x[7], cand[7];
    double V[36];
    double ne[NE_STRIDE];
    double diag[6], S[6];
    double radius, decrease_factor, model_cost_change, gmax, lm_used_max;
    int reuse_diagonal, iteration, done, termination, num_successful, num_invalid, evaluations, lm_overflow;
};

// pinned host record the device writes the result pose(s) into; seq is stored last with system-scope release
struct HostPublish {
    double x[7];
    double xb[8][7];
    long long done;          // SolverState::done at publication (the LM driver polls it)
    unsigned long long seq;
};

struct IterStatDev {        // mirrors mlh_iter_stat, written by the device-side update kernels
    int n_surf, n_corner, is_degenerate, lm_iterations, successful_steps, termination;
    double cost, final_cost;
    double eigval[6];
    double H[36];
    double g[6];
    double pose_after[7];
};

// ---------------------------------------------------------------- host-side containers
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    // owns its allocation: a buffer that is a member of the context (or of one of its sub-structures) is freed with it, so a new member
    // cannot be forgotten in mlh_destroy (which sets the device before `delete ctx`)
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    // grow to `bytes`, preserving the first `keep` bytes (device-to-device copy on `st`)
    hipError_t grow(size_t bytes, size_t keep, hipStream_t st)
    {
        if (bytes <= cap) return hipSuccess;
        void *np = nullptr;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&np, want);
        if (e != hipSuccess) return e;
        if (p && keep) {
            e = hipMemcpyAsync(np, p, keep, hipMemcpyDeviceToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) { (void)hipFree(np); return e; }
        }
        if (p) (void)hipFree(p);
        p = np; cap = want;
        return hipSuccess;
    }
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = bytes + bytes / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct MapGrid {
    DevBuf raw, sorted, cell_id, cell_start, cell_fill, block_sums, bounds, occ;
    int n = 0;
    bool want_occ = false;     // the index build also collects the occupancy statistics below (the mapper's two maps)
    int occ_parts = 0;         // per-workgroup partials in `occ`
    long long *occ_host = nullptr;   // pinned mirror of the totals (owned by the context)
    int occupied = 0;          // non-empty cells of the current index (0: not measured)
    long long pop_sq = 0;      // sum over the cells of population^2: pop_sq / n = the population of the cell an average map POINT lives in,
                               // which is what a query near the map sees (the density the query kernels tune to)
    float ox = 0, oy = 0, oz = 0, h = 1.f, inv_h = 1.f;
    int nx = 0, ny = 0, nz = 0;
    long long ncell = 0;
    float min_match_sq_dis = 1.0f;
    bool built = false;
    bool geom_valid = false;   // ox.. nz describe a box (with margin) laid out by a bounds pass; reusable while the clouds keep fitting
    float geom_sq_dis = 0.f;   // the acceptance radius that geometry was derived from
    int cur = 0;               // which of the two cell arrays (cell_start / cell_fill) holds the current index
    bool twin_clean = false;   // the other one has been cleared for the next build
    int *cells(int which) const { return (which ? cell_fill : cell_start).as<int>() + 3; }   // see grid.hip: cells + 1 is 16-byte aligned
    GridDev dev() const
    {
        GridDev g;
        g.sorted = sorted.as<float4>(); g.raw = raw.as<float4>(); g.cell_start = cells(cur);
        g.ox = ox; g.oy = oy; g.oz = oz; g.inv_h = inv_h; g.nx = nx; g.ny = ny; g.nz = nz; g.n = n;
        return g;
    }
};

struct FeatSet {
    DevBuf pts;        // float4 {x,y,z,intensity}
    DevBuf covd;       // float4 {cxx, cyy, czz, 0}  (diagonal of the f32 cov_vec)
    DevBuf corr;       // Corr per feature
    DevBuf nbr;        // 5 float4 per feature: the 5 nearest map points + squared distances
    DevBuf r, J;       // dense residual / Jacobian (double, double[6]) when requested
    DevBuf fps_order;  // 'fps' selection: [count][visiting order] (select.hip: fps_order_kernel)
    DevBuf fps_work;   // 'fps' selection: Morton keys, ranks, permutation of the pruned loop (select.hip: fps_order_pruned_kernel)
    DevBuf flag8;      // one byte per feature: Corr::valid on the way to the host, the selection's verdict on the way back (select.hip)
    int m = 0;             // feature slots (real + padding between pose blocks)
    int n_blocks = 1;
    int blk_start[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int blk_real[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // real features per block
    int nbr_stride = 5;
    bool has_cov = false;
    bool matched = false;
};

struct ScanBuf {
    DevBuf pts;            // float4 {x,y,z,intensity}
    DevBuf start, end;     // per ring
    int *end_alias = nullptr;   // mlh_scan_upload sends both tables in one copy: the end table then sits behind the start table in `start`
    int *end_ptr() { return end_alias ? end_alias : end.as<int>(); }
    DevBuf curvature, label, picked;
    DevBuf stage;          // per-ring staged picks
    DevBuf ring_counts;    // 4 counts per ring
    DevBuf ring_offsets;   // 4 exclusive offsets per ring (+ totals)
    DevBuf lists[4];
    DevBuf totals;         // 4 ints
    DevBuf vox_stage, vox_out, ring_vox;   // per-ring VoxelGrid of the less-flat points
    DevBuf tie_scratch;                    // label kernel: std::sort scratch of sectors with equal curvatures when it does not fit LDS
    DevBuf vox_keys, vox_perm;             // its voxel indices in list order and the order std::sort leaves them in (reference member order)
    bool voxelised = false;
    int n = 0, n_rings = 0;
    int max_ring_len = 0;  // max over rings of (scan_end - scan_start)
    bool extracted = false;
    int h_totals[5] = {0, 0, 0, 0, 0};   // host copy of the four list sizes + the thinned less-flat count, fetched once per scan
    bool h_lists_valid = false, h_vox_valid = false;
};

struct VoxBuf {   // scratch of mlh_voxel_filter
    DevBuf in, bounds, cell, wpre, cnt, vox_of, word_of, sorted_idx, members, leader, out, sums, total;
};

struct SegBuf {    // ImageSegmenter scratch (segment.hip)
    DevBuf raw, pix, owner, range, ground, keep;
    DevBuf edge;               // the cluster search's angle verdicts per pixel (seg_edge_kernel)
    DevBuf outmask, row_cnt;   // device row assembly: the outlier pixels' bit mask (from the host's cluster search), per-row counts of kept points (+ first kept index)
    void *h_rows = nullptr;    // pinned: [vs + 2] ints the row kernels leave for the host (kept points per row, total, first kept point index)
    size_t h_rows_cap = 0;
    DevBuf unc;            // points / ground pairs whose bin the device cannot decide (an angle within an ulp-scale margin of a bin edge): [counters 2 x int][records]
    DevBuf fix;            // the host's verdicts for the undecided points: {point index, pixel}
    void *h_unc = nullptr; // pinned mirror of `unc`
    size_t h_unc_cap = 0;
    void *h_img = nullptr;     // pinned: range / owner / ground images as the cluster search reads them, and the outlier mask it writes
    size_t h_img_cap = 0;
    void *h_bfs = nullptr;     // plain: labels and the cluster search's queue / pushed-pixel arrays
    size_t h_bfs_cap = 0;
    ~SegBuf() { if (h_unc) (void)hipHostFree(h_unc); if (h_rows) (void)hipHostFree(h_rows); if (h_img) (void)hipHostFree(h_img); std::free(h_bfs); }
};

struct OdomSet {   // staged LidarPureOdom factor table (odom.hip)
    DevBuf tab, idx, poses, r, J;
    int n = 0, max_frame = 0, max_ext = 0;
    // normal equations of the coupled window problem: factor indices grouped by (frame, extrinsic) in 256-factor tiles
    DevBuf perm, tile_group, partial, ne_out, solve_aux;
    int n_tiles = 0, group_ext = 1;
    bool tile_group_keyed = true;
    std::vector<int> h_tile_group, h_tile_frame, h_tile_ext;
    bool device_built = false;   // table appended from match passes on the device (padded regions): normal equations only
};

constexpr int FUSE_BLOCKS = 64;           // workgroups per kind of the fusion kernel: each leaves one partial bounding box of what it appended (frontend.hip)
constexpr int TRACK_SHELLS = 4;          // the tracker's index cells are 1/4 of its acceptance radius (track.hip: nearest_in_radius)
constexpr int TRACK_RING_SLOTS = 258;   // ring ids 0..255 (+ the slots the walks' upper bound can reach)
struct TrackSet {   // scan-to-scan odometry (track.hip): previous frame's clouds + indices, current frame's features
    MapGrid grid[2];
    DevBuf ring[2], ring_start[2], walk[2], cur[2], corr[2];
    int m[2] = {0, 0};
};
struct TrackArgs {
    int pose_sel = 0;
    const double *init_pose = nullptr;
    float dist_sq_thr = 25.f, nearby_scan = 2.5f;
    double huber_delta = 0.1;
    int finish = 0;          // track_linearize_launch: 3 / 4 = its last workgroup runs the Levenberg-Marquardt begin / step (solver_dev.hpp)
    int lm_max_it = 4, lm_min_blocks = 10, stat_slot = -1;
    HostPublish *publish = nullptr;          // finish 3 / 4: this launch hands pose + done flag to the host (pinned memory)
    unsigned long long publish_seq = 0;
};

struct Profile {
    unsigned mask = 0;     // bit k: bracket launches of kernel id k
    int every = 1;         // bracket every n-th launch of a kernel id only (event pairs cost ~6 us of host/queue time each)
    long long seen[MLH_K_COUNT] = {0};
    double total_ms[MLH_K_COUNT] = {0};
    long long launches[MLH_K_COUNT] = {0};
    struct Pending { int id; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

}  // namespace mlh

struct mlh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // What the device (and the part of it the solver's stream may use) admits, asked once at mlh_create (capi.hip: query_device_caps) -- the kernels whose
    // workgroups synchronise among themselves inside one launch (match.hip: lm_loop_kernel, track.hip: track_lm_loop_kernel) need EVERY workgroup resident at
    // once, and a plain launch checks nothing: the hosts gates them on loop_max_tiles[] instead of on a literal sized for a whole 256-CU part.
    struct DeviceCaps {
        int cu_count = 0;                 // hipDeviceProp_t::multiProcessorCount
        int cu_solver = 0;                // compute units the solver's stream may use (MLH_SOLVER_CU_MASK; otherwise all of them)
        bool solver_masked = false;
        bool staging_masked = false;      // the staging stream (mlh_map_set_pair_overlapped) exists and is confined to a part of the compute units (informational)
        int loop_demoted[3] = {-1, -1, -1};   // >= 0: a barrier was given up on -- the gate this context keeps below from then on
        int blocks_per_cu[3] = {0, 0, 0}; // hipOccupancyMaxActiveBlocksPerMultiprocessor: lm_loop_kernel<false>, lm_loop_kernel<true>, track_lm_loop_kernel
        int loop_max_tiles[3] = {0, 0, 0};// workgroups of those kernels the host will put behind one in-kernel barrier (0: never -- the launch-per-iteration forms)
        unsigned long long loop_timeout_ticks = 0;   // a barrier wait longer than this many 100 MHz ticks gives the loop up (MLH_LOOP_TIMEOUT_US, default 20 ms)
        unsigned long long loop_launches = 0;        // frames / rounds solved through the one-launch loop
        unsigned long long loop_timeouts = 0;        // ... that came back with the barrier given up on
        unsigned long long loop_fallbacks = 0;       // ... and were solved again through the launch-per-iteration form (the caller got a pose, later)
    } caps;
    // The local maps are double-buffered: `map` points at the set the solvers read; mlh_map_set_pair_overlapped stages the NEXT frame's maps into the
    // other set on a second stream while a submitted solve still reads this one, then switches `map` (launches capture a set's device pointers when they
    // are enqueued, so solves already in flight keep theirs).
    mlh::MapGrid map_sets[2][2];
    mlh::MapGrid *map = map_sets[0];
    int map_set_cur = 0;
    hipStream_t stream2 = nullptr;            // staging stream of the overlapped path (created on first use)
    hipEvent_t ev_set_built[2] = {nullptr, nullptr};   // recorded on the staging stream when set s has been built (the host waits on it)

    mlh::FeatSet feat[2];
    mlh::ScanBuf scan;
    mlh::DevBuf state;       // SolverState
    mlh::DevBuf partials;    // NE_STRIDE doubles per fit/linearise tile (surf tiles, then corner tiles)
    int n_partial_tiles = 0;
    mlh::DevBuf oob_flag;    // map staging: bit k set = the cloud of kind k has points outside its grid box
    bool oob_init = false;
    mlh::DevBuf ticket;      // arrival counter of the fused GN finish
    mlh::DevBuf loop_tagged; // lm_loop_kernel: the iterations' records as tagged words (two sets of 64 words per tile; match.hip: lm_consume_launch)
    unsigned loop_launch_seq = 0;   // ... and the launch number their tags carry (24 bits)
    mlh::DevBuf lm_pp;       // two LmState records of the consumer-side Levenberg-Marquardt schedule
    unsigned long long lmc_count = 0;     // consumer launches so far (its parity picks the record a launch writes)
    mlh::DevBuf stats;       // IterStatDev[...]
    mlh::DevBuf knn_q, knn_idx, knn_d;
    mlh::DevBuf tmp;         // H2D staging of caller records before packing
    mlh::DevBuf tmp_stage;   // the same for mlh_map_set_pair_overlapped, whose copies and pack kernels run on the staging stream beside the main stream's
    void *h_pts = nullptr;   // pinned landing place (two halves) of a caller's PAGEABLE scan points (mlh_scan_upload)
    size_t h_pts_cap = 0;    // bytes per half
    hipEvent_t ev_pts[2] = {nullptr, nullptr};
    bool ev_pts_used[2] = {false, false};
    unsigned pts_turn = 0;
    void *h_solve = nullptr; // pinned HostPublish record of a solve submitted with mlh_gn_solve_begin (collected by mlh_gn_solve_end)
    unsigned long long solve_seq = 0, solve_collected = 0;   // submitted / collected solves (at most two apart)
    unsigned long long set_reader_seq[2] = {0, 0};            // the youngest submitted solve that reads map set 0 / 1 (mlh_map_set_pair_overlapped: a set is rewritten only behind its readers)
    struct SolveSlot {                 // what mlh_scan2map_end needs to know about the solve whose record is h_solve[seq & 1]
        int kind = 0;                  // 0: Gauss-Newton (mlh_gn_solve_begin*), 1: scan2map (mlh_scan2map_begin*), 2: scan2map on maps too small to optimise against (the start pose comes back)
        bool chained = false;
        double start[7] = {0, 0, 0, 0, 0, 0, 1};
        mlh_solver_opts opts;
        unsigned long long epoch = 0;  // stage_epoch at submission
        bool tainted = false;          // chained behind a frame whose LM loop outgrew its look-ahead: began from an unfinished pose (mlh_scan2map_end status 3)
        int loop_tiles = 0;            // > 0: the frame's LM loops were submitted as one launch each over this many workgroups (lm_loop_kernel)
    } solve_slot[2];
    int lm_lookahead_auto = 10;           // mlh_scan2map_begin(lm_lookahead = 0): the previous frame's largest LM iteration count + 2 (10 until a frame has been collected)
    unsigned long long stage_epoch = 0;   // bumped by every call that restages a map or a feature set: a re-solve of an in-flight frame is only sound on unchanged inputs
    bool solve_pending = false;
    bool map_read_unsynced = false;   // a launch that reads the current map set was enqueued and its call did not wait for it (mlh_pure_odom_add_matches)
    // mlh_scan_upload_ahead: the NEXT scan's points copied to the device on a stream of their own (the copy engine beside this frame's kernels); the mlh_scan_upload
    // that names the same host buffer packs from `buf` instead of copying
    struct ScanAhead {
        hipStream_t cs = nullptr;
        mlh::DevBuf buf;
        hipEvent_t ev_arrived = nullptr, ev_consumed = nullptr;
        const void *src = nullptr;
        int n = 0, stride = 0;
        bool valid = false, consumed_recorded = false, src_pinned = false;
        unsigned long long issued = 0, used = 0;     // (tests)
    } ahead;
    void *h_state = nullptr; // pinned HostPublish record the device writes the result pose(s) into (capi.hip)
    void *h_occ = nullptr;   // pinned mirror of the two maps' occupancy totals (grid.hip): {cells, squares} per kind, written behind every index build
    unsigned long long publish_seq = 0;
    mlh::DevBuf uct_buf;     // point-uncertainty scratch
    mlh::VoxBuf vox;
    mlh::OdomSet odom;
    mlh::SegBuf seg;
    mlh::TrackSet track;
    mlh::DevBuf fused[2];    // body-frame union of the LiDARs' mapping features (mlh_fuse_*): float4 {x,y,z,lidar index}
    int fused_n[2] = {0, 0};   // valid when !fused_dirty
    int *h_dev_err = nullptr;   // one pinned int a kernel sets when it has to give up (device std::sort: a wait that was never released); see device_error_check
    void *h_rings = nullptr;                // pinned ring tables of mlh_scan_upload (two halves) ...
    size_t h_rings_cap = 0;                 // ... bytes per half
    hipEvent_t ev_rings[2] = {nullptr, nullptr};
    bool ev_rings_used[2] = {false, false};
    unsigned rings_turn = 0;
    hipEvent_t ev_handover = nullptr;      // mlh_features_copy: recorded on the source context's stream, waited for on this one's
    // mlh_fuse_add_scan_from(dst, this): a launch on ANOTHER context's stream reads this context's scan buffers; recorded there behind it, waited for on this
    // context's stream by whatever rewrites the scan next (scan_wait_readers). Set by the thread that drives dst while this context is idle.
    hipEvent_t ev_scan_reader = nullptr;
    std::atomic<bool> scan_reader_pending{false};
    unsigned long long *h_sync = nullptr;   // pinned word stream_wait_spin's launch stores into
    unsigned long long sync_seq = 0;
    unsigned long long counts_seq = 0;      // publications of the thinned feature counts straight from a kernel (voxel.hip)
    // downsample_current_scan_pair_run(.., defer = true): the thinning was enqueued and NOT waited for -- where its two counts will be (device; pinned host + the
    // sequence number their publication carries)
    const int *thin_counts_dev = nullptr;
    const int *thin_counts_host = nullptr;
    const unsigned long long *thin_seq_host = nullptr;
    unsigned long long thin_seq = 0;
    void *h_scratch = nullptr;  // 256 pinned bytes: the landing place of the few-int read-backs (record counts) that end a staging call
    void *fused_host = nullptr; // pinned record mlh_fused_cloud's publication launch fills: [2 counts (padded to 4 ints)][2 x 6 bounds][sequence word at byte 64]
    unsigned long long fused_seq = 0;
    size_t fused_host_cap = 0;
    mlh::DevBuf fused_cnt;   // the two record counts, device side (appends never wait for the host)
    size_t fused_bound[2] = {0, 0};   // host-side upper bounds of the counts (capacity)
    bool fused_dirty = false;
    mlh::DevBuf fused_part;  // per-append, per-kind, per-workgroup partial bounds of the appended points
    int fused_parts = 0;
    float fused_minmax[2][6];   // folded by mlh_fused_cloud: the voxel filter of a fused cloud needs no bounds pass of its own
    int knn_lanes_override = 0;   // MLH_KNN_LANES=8|16|32, or SSCC (816, 832, 1632: surf lanes, corner lanes), in the environment at mlh_create: pins the correspondence kernel's lanes per query (tests, tuning)
    int gn_final_defer = 1;       // MLH_GN_FINAL_DEFER=0: a solve submitted with mlh_gn_solve_begin* finishes its LAST iteration in its own fit launch (classic); 1: that
                                  // iteration, too, only leaves its records -- the next mlh_gn_solve_begin_chained completes it in its first launch (and publishes the pose
                                  // from there), mlh_gn_solve_end or any other solver call completes it with a one-workgroup launch if no successor did
    struct GnPending {            // the last iteration of the newest submitted solve is still a set of tile records
        bool active = false;
        int tiles = 0, slot = 0, freeze = 0;
        double thre = 100.0;
        void *rec = nullptr;      // HostPublish of that solve
        unsigned long long seq = 0;
    } gn_pending;
    int gn_slot_base = 0;         // xi slots of the next solve
    int gn_defer = 1;             // MLH_GN_DEFER=0: Gauss-Newton solves keep the classic finish (the fit kernel's last-arriving workgroup) in every iteration (A/B, tests)
    int knn_warm = 1;             // MLH_KNN_WARM=0: iterations >= 1 of a solve search without the previous iteration's neighbours as a bound (A/B, tests)
    // multi-GPU
    bool shard_lo = false, shard_hi = false;
    float lo_plane[4] = {0, 0, 0, 0}, hi_plane[4] = {0, 0, 0, 0};
    int own_mod = 1, own_rem = 0;   // feature-index ownership (replicated map): mlh_shard_set_features
    void *comm = nullptr;    // ncclComm_t
    // the mailbox communicator (comm.hip): every rank's mailbox mapped into this process; one kernel per all-reduce, no library in between
    struct P2p {
        bool active = false;
        void *mailbox = nullptr;                      // this rank's own (device memory, exported through hipIpc)
        void *peer[16] = {};                          // rank r's mailbox as this process sees it (peer[rank] == mailbox)
        void *counter = nullptr;                      // device word: exchanges completed (its parity picks the half of the mailboxes in use)
    } p2p;
    mlh::DevBuf allreduce_buf;   // staging of mlh_allreduce_f64
    int extract_tie_ref = 1;           // extractCloud, equal curvatures inside a sector: 1 = the order the reference's std::sort call leaves (default), 0 = (curvature, index)
    int vox_member_order = 1;          // voxel filters, members of a voxel: 1 = in the order libstdc++'s std::sort leaves them (the reference's), produced on the device
                                       // (stdsort.hip); 2 = the same through a host pass that calls the platform's own std::sort; 0 = in point-index order
    mlh::DevBuf stdsort;               // scratch of device_std_sort_by_key
    void *vox_order_host = nullptr; // pinned staging of that host pass (voxelgrid.hip): [slot n][members n]
    size_t vox_order_host_cap = 0;
    void *select_host[2] = {nullptr, nullptr}; // pinned staging of the good-feature selection, per feature kind (select.hip)
    size_t select_host_cap[2] = {0, 0};
    std::vector<char> select_rows[2];   // the same rows in ordinary (CPU-cached) memory: what the selection loops read
    unsigned long long select_seq[2] = {0, 0};  // stream_flag_post after each kind's copies to the host
    bool select_staged[2] = {false, false};
    long select_fps_start[2] = {-1, -1};        // 'fps': the starting point drawn at staging time
    struct { int active = 0, m = 0, n_use = 0, cur0 = 0; void *host_dst = nullptr; } fps_pending[2];   // 'fps': a kind's loop staged but not yet launched (select.hip: good_feature_fps_flush)
    int n_ranks = 1, rank = 0;
    mlh::Profile prof;
};

namespace mlh {

int fail(mlh_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess);

// MLH_CHECK_LAUNCH=1 (debug runs): hipGetLastError() right behind EVERY kernel launch, so that a bad launch configuration is reported under the name of the kernel
// that caused it instead of surfacing at the next synchronisation under another call's name. The error is sticky for the calling thread: the next MLH_HIP check
// (every entry point runs several) returns MLH_ERR_HIP with "launch of <kernel>: <hip error>". Off (the default) the launches are followed by nothing.
bool launch_check_enabled();
void launch_check(const char *kernel);
hipError_t launch_check_take(const char **kernel);      // this thread's sticky launch error (hipSuccess when none); cleared by the call
int fail_launch(mlh_ctx *ctx, const char *kernel, hipError_t e);
#define MLH_LAUNCH(kern, grid, block, lds, st, ...)                                \
    do {                                                                           \
        hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);               \
        if (::mlh::launch_check_enabled()) ::mlh::launch_check(#kern);             \
    } while (0)

#define MLH_HIP(ctx, expr)                                                         \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) return ::mlh::fail((ctx), MLH_ERR_HIP, #expr, _e);   \
        if (::mlh::launch_check_enabled()) {                                       \
            const char *_k = nullptr;                                              \
            const hipError_t _le = ::mlh::launch_check_take(&_k);                  \
            if (_le != hipSuccess) return ::mlh::fail_launch((ctx), _k, _le);      \
        }                                                                          \
    } while (0)

// profiling brackets (HIP events on the context's stream)
void prof_begin(mlh_ctx *ctx, int id);
void prof_end(mlh_ctx *ctx, int id);
void prof_collect(mlh_ctx *ctx);
// events whose timestamps come from the dispatch itself (hipExtLaunchKernelGGL); false when kernel id is not profiled
bool prof_kernel_events(mlh_ctx *ctx, int id, hipEvent_t *start, hipEvent_t *stop);

// extract.hip
int extract_run(mlh_ctx *ctx);
// voxel.hip
int ring_voxel_run(mlh_ctx *ctx, float leaf);
int point_uncertainty_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int mem, const double *ext_poses,
                          const double *ext_covs, int n_lidar, const double cov_meas[9], double trace_thr, float *cov6_host, int *keep_host);
// track.hip
int track_set_prev_rings(mlh_ctx *ctx, int kind, const unsigned char *d_src, int stride, int n, int intensity_off, int *host_bad);
int track_match_launch(mlh_ctx *ctx, int kind_mask, const mlh::TrackArgs &a);
int track_linearize_launch(mlh_ctx *ctx, int kind_mask, const mlh::TrackArgs &a);
int track_lm_loop_launch(mlh_ctx *ctx, int kind_mask, const mlh::TrackArgs &a);      // one round's whole LM loop (begin at the round's pose .. termination) in one launch
// segment.hip
int segment_cloud_run(mlh_ctx *ctx, const void *points, int stride, int intensity_off, int n, int mem, const mlh_segment_params &prm,
                      float *cloud_out, int32_t *n_out, int32_t *scan_start, int32_t *scan_end, float *outlier_out, int32_t outlier_capacity, int32_t *n_outlier);
// odom.hip
int pure_odom_set(mlh_ctx *ctx, int n, const int32_t *type, const double *points, const double *coeffs, const double *sqrt_info,
                  const int32_t *frame_idx, const int32_t *ext_idx);
int pure_odom_evaluate(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext,
                       double *residuals, double *jacobians);
int pure_odom_begin(mlh_ctx *ctx);
int pure_odom_add_matches(mlh_ctx *ctx, int kind, int frame_idx, int ext_idx);
int pure_odom_feature_rows(mlh_ctx *ctx, int kind, const double pivot[7], const double pose_i[7], const double ext[7]);      // odom.hip: validity + scored rows of the staged features -> FeatSet::flag8 / J
int pure_odom_normal_eq(mlh_ctx *ctx, const double pivot[7], const double *frames, int n_frames, const double *exts, int n_ext, double huber_delta,
                        double *H, double *g, double *cost, int32_t *n_res);
int pure_odom_gn_solve(mlh_ctx *ctx, const double pivot[7], double *frames, int n_frames, double *exts, int n_ext, double huber_delta, int n_iters,
                       uint32_t const_block_mask, const double *V_update, double *cost, int32_t *n_res, int32_t *status_out);
// voxelgrid.hip
int device_exclusive_scan(mlh_ctx *ctx, int *data, long long n, mlh::DevBuf &sums, int *grand_total);
// A kernel that cannot honour its contract (today: the device std::sort when a queue wait runs out, or a range it was never told about) sets
// the context's pinned error word instead of leaving a wrong order behind silently; every call that waits for the stream afterwards reports it.
inline int *device_error_word(mlh_ctx *ctx)
{
    if (!ctx->h_dev_err) {
        void *p = nullptr;
        if (hipHostMalloc(&p, sizeof(int), hipHostMallocDefault) == hipSuccess) { ctx->h_dev_err = static_cast<int *>(p); *ctx->h_dev_err = 0; }
    }
    return ctx->h_dev_err;
}
inline int device_error_check(mlh_ctx *ctx)
{
    if (ctx->h_dev_err && *static_cast<volatile int *>(ctx->h_dev_err) != 0) {
        const int code = *ctx->h_dev_err;
        *ctx->h_dev_err = 0;
        if (code == 2)
            return fail(ctx, MLH_ERR_STATE, "a peer rank did not arrive at a mailbox exchange within 5 s: the normal equations were NOT summed over the job, no update was applied from them, "
                                            "and the ranks may no longer hold the same pose -- the result of this call is not valid");
        return fail(ctx, MLH_ERR_STATE, "a device kernel gave up (std::sort emulation: unreleased wait or an unannounced range); the results of this call are not valid");
    }
    return MLH_OK;
}
// One turn of a host-side wait on a pinned word (the publications of the solves, the staging hand-shakes, the few-int read-backs): those waits are microseconds
// long, so the default is to spin (`pause`) -- a sleeping thread's wake-up costs tens. A process with one thread per LiDAR + the mapper + the tracker, each in
// such a wait, burns that many cores; MLH_HOST_WAIT=yield (read once per process) spins the first 64 turns (~2 us: the common case still pays nothing) and then
// gives the core to whoever is runnable between two looks. Either way a wait falls back to the blocking hipStreamSynchronize after 200 ms.
inline void host_wait_relax(unsigned spins)
{
    static const bool yield_mode = [] { const char *e = std::getenv("MLH_HOST_WAIT"); return e && std::strcmp(e, "yield") == 0; }();
    if (yield_mode && spins > 64u) { sched_yield(); return; }
#if defined(__x86_64__)
    __builtin_ia32_pause();
#endif
}
// 64 pinned ints owned by the context (lazily allocated); nullptr on allocation failure
inline int *pinned_ints(mlh_ctx *ctx)
{
    if (!ctx->h_scratch && hipHostMalloc(&ctx->h_scratch, 256, hipHostMallocDefault) != hipSuccess) ctx->h_scratch = nullptr;
    return static_cast<int *>(ctx->h_scratch);
}
// Everything enqueued on the context's stream so far (kernels, and copies into PINNED host memory) has completed when this returns. A one-thread launch stores a
// sequence number into pinned host memory (system-scope release) and the host spins on that word: a few microseconds, where hipStreamSynchronize's wake-up
// costs tens -- a mapper frame has three such read-backs (fused cloud sizes, thinned record counts, feature counts). Falls back to the blocking call after
// 200 ms (a profiler, a fault). capi.hip has the definition.
hipError_t stream_wait_spin(mlh_ctx *ctx);
// the two halves of it: post a marker behind what is enqueued now; wait for a posted marker later
hipError_t stream_flag_post(mlh_ctx *ctx, unsigned long long *seq_out);
hipError_t stream_flag_wait(mlh_ctx *ctx, unsigned long long seq);
// *out <- one device int, through the pinned block (a pageable landing place costs a staging hop); waits for the stream
inline hipError_t read_back_int(mlh_ctx *ctx, const void *dev, int *out)
{
    int *h = pinned_ints(ctx);
    int *dst = h ? h + 8 : out;
    hipError_t e = hipMemcpyAsync(dst, dev, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = h ? stream_wait_spin(ctx) : hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && h) *out = *dst;
    return e;
}
// stdsort.hip: vals_out <- the permutation of 0..n-1 that std::sort (libstdc++, comparator on the key only) leaves for keys[0..n0) and keys[n0..n)
// Voxel keys computed INSIDE the sort's init launch (the frame's thinning pipeline, voxel.hip: downsample_current_scan_pair_run): point i of cloud 0 (i < n0) or
// cloud 1 -> its voxel index in the cloud's own dense grid, the second grid numbered behind the first (the arithmetic of vox_mark_kernel / PCL's VoxelGrid).
struct VoxKeyGen {
    const unsigned char *src0, *src1;
    int stride, n0;
    float inv_leaf0, inv_leaf1;
    int min_b0[3], mul1_0, mul2_0;
    int min_b1[3], mul1_1, mul2_1, cell_off1;
};
int device_std_sort_by_key(mlh_ctx *ctx, const int *src_keys, int n0, int n, int *vals_out, const VoxKeyGen *gen = nullptr);
const int *device_std_sort_keys(mlh_ctx *ctx, int n);       // the keys of the last device_std_sort_by_key over n elements, sorted, once its launches have run
int device_std_sort_segments(mlh_ctx *ctx, const int *src_keys, const int *counts, const int *offsets, int stride, int field, int n_segments, int n, int longest,
                             int *vals_out, bool counters_cleared = false);
int *device_std_sort_counters(mlh_ctx *ctx, int n, int *n_counters);
void host_std_sort_permutation(const int *slot, int lo, int hi, int *members);   // voxelgrid.hip: the platform's own std::sort
// voxel.hip
void compound_pose_with_cov(const double p1[7], const double c1[36], const double p2[7], const double c2[36], double pc[7], double cc[36]);
int downsample_current_scan_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int mem, float leaf, const double *ext_poses,
                                const double *ext_covs, int n_lidar, const double cov_meas[9], int with_ua, double trace_thr, mlh::DevBuf &pts_out,
                                mlh::DevBuf &covd_out, float *out11_dev, int *n_out, const float *known_bounds = nullptr);
int cloud_uct_associate_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int cov_off, int trace_off,
                            const double pose_global[7], const double cov_global[36], const double *ext_poses, const double *ext_covs,
                            int n_lidar, const double cov_meas[9], int with_ua, double trace_thr, void *out, int *n_out, int mem);
int voxel_filter_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int cov_off, int trace_off, float leaf,
                     float trace_thr, void *out_host, int *n_out, int mem, const float *known_bounds = nullptr, bool sync_total = true,
                     bool centroid_all = false);
// frontend.hip
void gather_points_launch(mlh_ctx *ctx, const float4 *pts, const int *list, int n, float4 *out);
int transform_cloud_launch(mlh_ctx *ctx, void *dev, int stride, int n, const double pose[7]);
int transform_to_end_launch(mlh_ctx *ctx, void *dev, int stride, int n, int intensity_off, const double pose[7], int b_distortion, float scan_period);
int fuse_append_launch(mlh_ctx *ctx, ScanBuf &sb, int ring_begin, int ring_end, int lidar_idx, const double ext_pose[7]);   // sb: this context's scan or another's (mlh_fuse_add_scan_from)
int voxel_filter_run2(mlh_ctx *ctx, const void *src0, int n0, const float bounds0[6], float leaf0, const void *src1, int n1, const float bounds1[6],
                      float leaf1, int stride, int intensity_off, int *first_voxels_word);
int downsample_current_scan_pair_run(mlh_ctx *ctx, const void *surf, int n_surf, const float bounds_surf[6], float leaf_surf, const void *corner, int n_corner,
                                     const float bounds_corner[6], float leaf_corner, int stride, int intensity_off, const double *ext_poses, const double *ext_covs,
                                     int n_lidar, const double cov_meas[9], int with_ua, double trace_thr, int *n_surf_out, int *n_corner_out, bool defer = false);
// grid.hip
int grid_build(mlh_ctx *ctx, int kind_mask, bool recompute_bounds);
int grid_build_grids(mlh_ctx *ctx, mlh::MapGrid **grids, int n_grids, bool recompute_bounds, int *pub_oob = nullptr, mlh::HostPublish *pub = nullptr,
                     unsigned long long pub_seq = 0);
void knn_lanes_for(const mlh_ctx *ctx, int kind_mask, int lanes[2]);
int map_stage_and_build(mlh_ctx *ctx, int n_maps, const int *kinds, const unsigned char *const *src, const int *n, int stride, const float *sq_dis,
                        mlh::HostPublish *pub, unsigned long long seq);
// match.hip
struct MatchArgs {
    int kind_mask = 3;   // bit MLH_SURF, bit MLH_CORNER: which feature kinds take part in the launch
    uint32_t flags = 0;
    float min_match_sq_dis = 1.f, min_plane_dis = 0.2f;
    double huber_delta = 0.1, cov_measurement_trace = 0.0075;
    bool dense = false;  // also write r / J per feature
    int pose_sel = 0;    // 0: SolverState::x, 1: SolverState::cand
    int finish = 0;      // 1: the fit kernel's last workgroup completes the GN iteration (reduce + solve + Plus); 2: local reduce only;
                         // 3 / 4: it runs the Levenberg-Marquardt begin (match_launch) / step (linearize_launch)
    int lm_max_it = 30, lm_min_blocks = 0;
    int lm_expect_done = 0;   // finish == 3 (LM begin): -1 = first outer iteration of a solve (clears SolverState::lm_overflow); 1 = a later outer iteration submitted without the
                              // host having seen the previous LM loop's verdict: if that loop has not terminated, lm_overflow is raised (the launches go on; the host discards)
    int stat_slot = -1;
    int n_blocks = 1;    // pose blocks
    int k_neigh[8] = {5, 5, 5, 5, 5, 5, 5, 5};
    double eig_thre[8] = {100, 100, 100, 100, 100, 100, 100, 100};
    int freeze[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double *init_pose = nullptr;       // host: block 0's pose for this launch comes from the kernel arguments and is written to the state by the finish
    // Gauss-Newton with the finish done by the consumer: this launch pair is iteration `gn_iter` of `gn_iters` (>= 2). The fit kernel of every iteration but
    // the last only leaves its tiles' partial records; the correspondence kernel of iteration i >= 1 starts by summing them (every workgroup, same order, same
    // bits) and running the 6 x 6 solve + Plus itself. gn_iter < 0: the classic form (the fit kernel's last-arriving workgroup finishes).
    int gn_iter = -1, gn_iters = 0;
    int gn_slot_base = 0;     // this solve's pair of SolverState::xi slots (0 or 2)
    bool gn_blocks = false;   // the solve runs over pose blocks: per-block iteration poses in SolverState::xib
    bool warm = false;   // the neighbour records of the previous iteration (same features, same map) bound this iteration's search
    // iteration 0 of a CHAINED solve that also completes the PREVIOUS solve, whose last iteration left only its tiles' records (mlh_ctx::gn_pending): every workgroup of
    // this correspondence launch sums them, solves, applies Plus -> the previous frame's final pose; tile 0's workgroup publishes it to that solve's host record and
    // stores it as the state's pose; then the chained start pose of THIS frame (transformUpdate + transformAssociateToMap) is computed from it, in every workgroup
    bool pre_final = false;
    int pre_final_tiles = 0, pre_final_slot = 0, pre_final_freeze = 0;
    double pre_final_thre = 100.0;
    HostPublish *pre_final_publish = nullptr;
    unsigned long long pre_final_seq = 0;
    const double *chain_prev = nullptr, *chain_cur = nullptr;   // host: the two odometry poses of the chain (7 doubles each)
    // Levenberg-Marquardt with the step done by the consumer (lm_consume_launch): 1 = the first launch behind a match launch whose fit kernel only left its records
    // (finish 0) -- sums them, runs the LM begin, evaluates at the first candidate; 2 = a later launch -- sums the records at the candidate, runs the LM step,
    // evaluates at the next candidate. lmc_j: the launch's number within its loop (1, 2, ...: record buffer (j - 1) & 1 is read, j & 1 written)
    int lmc = 0, lmc_j = 0;
    // round 6: the fit of an outer iteration inside the loop launch that follows it (lm_loop_kernel<.., FIT>): match_launch with `no_fit` ends behind the
    // correspondence kernel, lm_consume_launch (lmc == 3) with `fit_in_loop` begins with the fit -- one launch boundary fewer per outer iteration. Only together,
    // and only where loop_fit_fusable() says so (tagged records: the fit's record leaves as the loop's own do).
    bool no_fit = false, fit_in_loop = false;
    const int *m_dev = nullptr;   // device: the two feature counts (surf, corner) when the host has not read them (mlh_downsample_scan2map); FeatSet::m then holds upper bounds
    HostPublish *publish = nullptr;          // pinned host record the finish writes the pose(s) to (finish == 1 only)
    unsigned long long publish_seq = 0;
};
int match_launch(mlh_ctx *ctx, const MatchArgs &a);
int gn_flush_pending(mlh_ctx *ctx);      // completes a pending last iteration with a one-workgroup launch (no-op when nothing is pending)
int linearize_launch(mlh_ctx *ctx, const MatchArgs &a);
int lm_consume_launch(mlh_ctx *ctx, const MatchArgs &a);
bool loop_fit_fusable(const MatchArgs &loop_args);   // the loop launch described by these arguments would exchange tagged records (ctx.hpp: loop_tagged_arm's conditions)
int lm_loop_occupancy(int blocks_per_cu[2]);      // hipOccupancyMaxActiveBlocksPerMultiprocessor of lm_loop_kernel<false> / <true>
int track_loop_occupancy(int *blocks_per_cu);     // ... of track_lm_loop_kernel (track.hip)
// the arrival counters of the fused finishes and of lm_loop_kernel's barrier: four zeroed words, whoever asks first ([0]: the finish tickets of match.hip and
// track.hip; [1] arrivals, [2] departures, [3] release flag of the loop kernel). One place, so that no caller can leave the others' words unallocated or unzeroed.
// The tagged record sets of the one-launch LM loops (match.hip: lm_loop_kernel, track.hip: track_lm_loop_kernel; reduce_dev.hpp: lmc_sum_records_tagged): two sets of
// 64 words per tile, and the number this launch's tags carry. *buf stays null where the loop keeps its grid barrier (MLH_LOOP_TAGGED=0; more iterations than the tag's
// iteration byte counts). Zero is never a tag: a fresh allocation and a wrap of the 24-bit launch number clear the words.
inline hipError_t loop_tagged_arm(mlh_ctx *ctx, size_t tiles, int lm_max_it, unsigned long long **buf, unsigned *tag_base)
{
    const char *env = std::getenv("MLH_LOOP_TAGGED");           // (read at every call, as the other schedule switches: tests run both forms in one process)
    const bool off = env && std::atoi(env) == 0;
    *buf = nullptr; *tag_base = 0u;
    if (off || lm_max_it > 200 || tiles == 0) return hipSuccess;
    const size_t bytes = sizeof(unsigned long long) * 64 * tiles * 2;
    bool clear = bytes > ctx->loop_tagged.cap;
    hipError_t e = ctx->loop_tagged.ensure(bytes);
    if (e != hipSuccess) return e;
    ctx->loop_launch_seq = (ctx->loop_launch_seq + 1u) & 0xffffffu;
    if (ctx->loop_launch_seq == 0u) { ctx->loop_launch_seq = 1u; clear = true; }
    if (clear && (e = hipMemsetAsync(ctx->loop_tagged.p, 0, ctx->loop_tagged.cap, ctx->stream)) != hipSuccess) return e;
    *buf = ctx->loop_tagged.as<unsigned long long>();
    *tag_base = ctx->loop_launch_seq << 8;
    return hipSuccess;
}

inline hipError_t ensure_ticket(mlh_ctx *ctx)
{
    if (ctx->ticket.p) return hipSuccess;
    hipError_t e = ctx->ticket.ensure(4 * sizeof(unsigned));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(ctx->ticket.p, 0, 4 * sizeof(unsigned), ctx->stream);
}
int knn_launch(mlh_ctx *ctx, int kind, const float *q_host, int nq, int32_t *idx, float *d2);
// select.hip
}  // namespace mlh
#include <random>
namespace mlh {
int good_feature_select(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, float min_match_sq_dis,
                        float min_plane_dis, std::vector<int32_t> &sel_out, double H[36], uint8_t *matched_out);
// its two halves: the dense pass + copies to the host, enqueued (no wait); the selection loop on the copied rows, flags sent back (no wait)
int good_feature_stage(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, float min_match_sq_dis, float min_plane_dis, bool defer_fps = false);
int good_feature_fps_flush(mlh_ctx *ctx);          // launches what the stages called with defer_fps left pending (both kinds' 'fps' loops in one launch)
int odom_good_feature_select(mlh_ctx *ctx, int kind, float gf_ratio, std::mt19937 &rng, std::vector<int32_t> &sel_out);     // select.hip: Estimator::goodFeatureMatching's loop over those rows
int good_feature_finish(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, std::vector<int32_t> &sel_out, double H[36],
                        uint8_t *matched_out);
// solver.hip
int reduce_only_launch(mlh_ctx *ctx, int to_ce);
int gn_update_prereduced_launch(mlh_ctx *ctx, double map_eig_thre, int stat_slot);
int gn_update_blocks_prereduced_launch(mlh_ctx *ctx, int n_blocks, const double *eig_thre, const int *freeze, int stat_slot);
// comm.hip
inline bool distributed(const mlh_ctx *ctx) { return ctx->comm != nullptr || ctx->p2p.active; }
int comm_allreduce_state(mlh_ctx *ctx, int to_ce);
int comm_allreduce_blocks(mlh_ctx *ctx, int n_blocks);
void comm_destroy(mlh_ctx *ctx);   // in-place ncclAllReduce of SolverState::ne / ::ce on the stream
int lm_begin_launch(mlh_ctx *ctx, double map_eig_thre, int max_iterations, int stat_slot, int min_blocks = 0, const double *init_pose = nullptr);
int lm_step_launch(mlh_ctx *ctx, int max_iterations, int stat_slot);
int lm_finish_launch(mlh_ctx *ctx, int stat_slot);

}  // namespace mlh
