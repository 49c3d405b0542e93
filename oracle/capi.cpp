// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). Pin status per file: feature_extract.cpp, factors.hpp, image_segmenter.cpp are held against the
// reference's own source lines (oracle/ref/); linalg.hpp, kdtree.hpp, mapper.cpp, uct.hpp, tracker.hpp restate library arithmetic: PARITY UNPINNED there.
// extern "C" surface of the CPU oracle, loaded with ctypes by tests/, bench.py's cpu_baseline leg and
// __graft_entry__.smoke(). The product library (m-loam_amd/) never links or loads this.
#include "image_segmenter.hpp"
#include "feature_extract.hpp"
#include "mapper.hpp"
#include "uct.hpp"
#include "tracker.hpp"
#include "linalg.hpp"
#include <cstring>
#include <chrono>

using namespace orc;

extern "C" {

int orc_extract(const float *xyzi, int n, const int *scan_start, const int *scan_end, int n_scans,
                float *curvature, int *label, int *picked,
                int *sharp, int *n_sharp, int *less_sharp, int *n_less_sharp, int *flat, int *n_flat,
                int *less_flat_raw, int *n_less_flat_raw, float *less_flat_ds, int *n_less_flat_ds, long *n_ties)
{
    ExtractResult r;
    extract_cloud(reinterpret_cast<const PointI *>(xyzi), n, scan_start, scan_end, n_scans, r);
    if (curvature) std::memcpy(curvature, r.curvature.data(), sizeof(float) * n);
    if (label) std::memcpy(label, r.label.data(), sizeof(int) * n);
    if (picked) std::memcpy(picked, r.picked.data(), sizeof(int) * n);
    auto put = [](const std::vector<int> &v, int *dst, int *cnt) {
        if (dst) std::memcpy(dst, v.data(), sizeof(int) * v.size());
        if (cnt) *cnt = (int)v.size();
    };
    put(r.sharp, sharp, n_sharp);
    put(r.less_sharp, less_sharp, n_less_sharp);
    put(r.flat, flat, n_flat);
    put(r.less_flat_raw, less_flat_raw, n_less_flat_raw);
    if (less_flat_ds) std::memcpy(less_flat_ds, r.less_flat_ds.data(), sizeof(PointI) * r.less_flat_ds.size());
    if (n_less_flat_ds) *n_less_flat_ds = (int)r.less_flat_ds.size();
    if (n_ties) *n_ties = r.n_ties;
    return 0;
}

// 0: the reference's comparator (curvature only; tie order = libstdc++'s); 1: (curvature, index) total order, NaN last
void orc_set_tie_rule(int rule) { set_tie_rule(rule); }

int orc_voxel_grid(const float *xyzi, int n, float leaf, float *out, int *n_out)
{
    std::vector<PointI> o;
    voxel_grid_xyzi(reinterpret_cast<const PointI *>(xyzi), n, leaf, o);
    std::memcpy(out, o.data(), sizeof(PointI) * o.size());
    *n_out = (int)o.size();
    return 0;
}

struct OrcMap {
    std::vector<float> pts;   // owned copy, stride floats
    int stride;
    MapCloud mc;
};

void *orc_map_create(const float *pts, int stride_floats, int n)
{
    OrcMap *m = new OrcMap;
    m->pts.assign(pts, pts + size_t(n) * stride_floats);
    m->stride = stride_floats;
    m->mc.set(m->pts.data(), stride_floats, n);
    return m;
}
void orc_map_destroy(void *h) { delete static_cast<OrcMap *>(h); }

// timed kd-tree (re)build, for the reference-faithful baseline row (setInputCloud every frame, cpp:433-434)
double orc_map_rebuild_seconds(void *h)
{
    OrcMap *m = static_cast<OrcMap *>(h);
    auto t0 = std::chrono::steady_clock::now();
    m->mc.set(m->pts.data(), m->stride, m->mc.n);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

int orc_knn(void *h, const float *queries, int nq, int k, int *idx, float *d2)
{
    OrcMap *m = static_cast<OrcMap *>(h);
    for (int i = 0; i < nq; ++i) {
        int f = m->mc.tree.knn(queries + size_t(i) * 3, k, idx + size_t(i) * k, d2 + size_t(i) * k);
        for (int j = f; j < k; ++j) { idx[size_t(i) * k + j] = -1; d2[size_t(i) * k + j] = INFINITY; }
    }
    return 0;
}

// dense per-feature match: valid[n], coeffs[n*6]
int orc_match(void *h, char type, const float *feats, int stride, int n, const double *pose7, int n_neigh, int check_fov,
              float min_match_sq_dis, float min_plane_dis, unsigned char *valid, double *coeffs)
{
    OrcMap *m = static_cast<OrcMap *>(h);
    MatchParams mp{min_match_sq_dis, min_plane_dis};
    Pose pose = pose_from_param(pose7);
    for (int i = 0; i < n; ++i) {
        Feature f;
        bool ok = (type == 's')
                      ? match_surf_point_from_map(m->mc, feats + size_t(i) * stride, pose, f, i, n_neigh, check_fov != 0, mp)
                      : match_corner_point_from_map(m->mc, feats + size_t(i) * stride, pose, f, i, n_neigh, check_fov != 0, mp);
        valid[i] = ok ? 1 : 0;
        for (int k = 0; k < 6; ++k) coeffs[size_t(i) * 6 + k] = ok ? f.coeffs[k] : 0.0;
    }
    return 0;
}

int orc_factor_eval(char type, const double *point, const double *coeff, double cov_trace, const double *pose7,
                    double *residual, double *jac7)
{
    double si = sqrt_info_from_trace(cov_trace);
    if (type == 's') plane_norm_factor_evaluate(point, coeff, si, pose7, residual, jac7);
    else edge_factor_evaluate(point, coeff, si, pose7, residual, jac7);
    return 0;
}

int orc_pose_plus(const double *x, const double *delta, const double *V_update, double *out)
{
    pose_plus(x, delta, V_update, out);
    return 0;
}

// transformUpdate + transformAssociateToMap (lidar_mapper_keyframe.cpp:145-160): the mapper's start pose of the next frame
int orc_pose_chain(const double *wmap_curr_prev, const double *wodom_prev, const double *wodom_cur, double *out)
{
    const Pose r = pose_chain(pose_from_param(wmap_curr_prev), pose_from_param(wodom_prev), pose_from_param(wodom_cur));
    out[0] = r.t.x; out[1] = r.t.y; out[2] = r.t.z; out[3] = r.q.x; out[4] = r.q.y; out[5] = r.q.z; out[6] = r.q.w;
    return 0;
}

int orc_huber(double a, double s, double *rho3) { huber_evaluate(a, s, rho3); return 0; }

// per-feature r, J (weighted, not loss-corrected) + loss-corrected reduction over valid features
int orc_linearize(char type, const float *feats, int stride, int n, const double *cov_trace /*n or null*/,
                  const double *pose7, const unsigned char *valid, const double *coeffs, double huber_delta,
                  double *r_out, double *J_out /*n*6*/, double *H36, double *g6, double *cost, int *count)
{
    std::vector<ResidualBlock> blocks;
    for (int i = 0; i < n; ++i) {
        if (r_out) r_out[i] = 0.0;
        if (J_out) for (int k = 0; k < 6; ++k) J_out[size_t(i) * 6 + k] = 0.0;
        if (!valid[i]) continue;
        ResidualBlock b;
        b.type = type;
        const float *p = feats + size_t(i) * stride;
        b.point[0] = p[0]; b.point[1] = p[1]; b.point[2] = p[2];
        for (int k = 0; k < 6; ++k) b.coeffs[k] = coeffs[size_t(i) * 6 + k];
        b.sqrt_info = sqrt_info_from_trace(cov_trace ? cov_trace[i] : 0.0);
        blocks.push_back(b);
        double r, J[7];
        if (type == 's') plane_norm_factor_evaluate(b.point, b.coeffs, b.sqrt_info, pose7, &r, J);
        else edge_factor_evaluate(b.point, b.coeffs, b.sqrt_info, pose7, &r, J);
        if (r_out) r_out[i] = r;
        if (J_out) for (int k = 0; k < 6; ++k) J_out[size_t(i) * 6 + k] = J[k];
    }
    NormalEq ne;
    evaluate_problem(blocks, pose7, huber_delta, ne, true);
    if (H36) std::memcpy(H36, ne.H, sizeof(ne.H));
    if (g6) std::memcpy(g6, ne.g, sizeof(ne.g));
    if (cost) *cost = ne.cost;
    if (count) *count = ne.n;
    return 0;
}

int orc_eval_degeneracy(const double *H36, double eig_thre, double *eigval6, double *eigvec36, double *V_update36, int *is_deg)
{
    Degeneracy d;
    eval_degeneracy(H36, eig_thre, d);
    if (eigval6) std::memcpy(eigval6, d.eigval, sizeof(d.eigval));
    if (eigvec36) std::memcpy(eigvec36, d.eigvec, sizeof(d.eigvec));
    if (V_update36) std::memcpy(V_update36, d.V_update, sizeof(d.V_update));
    if (is_deg) *is_deg = d.is_degenerate ? 1 : 0;
    return 0;
}

int orc_window_eval_degeneracy(const double *JtJ, int D, int n_pose_blocks, double *eig_thre, int estimate_extrinsic, long frame_cnt, int n_cumu_feature,
                               double lambda_thre_calib, int *is_degenerate, double *V_update, double *eigval, double *d_factor_calib)
{
    window_eval_degeneracy(JtJ, D, n_pose_blocks, eig_thre, estimate_extrinsic != 0, frame_cnt, n_cumu_feature, lambda_thre_calib, is_degenerate, V_update, eigval,
                           d_factor_calib);
    return 0;
}

static FeatureCloud make_fc(const float *feats, int stride, int n, int cov_off)
{
    FeatureCloud fc;
    fc.pts = feats; fc.stride = stride; fc.n = n; fc.cov_off = cov_off;
    return fc;
}

static MapperParams make_params(const double *prm)
{
    // prm: [min_match_sq_dis, min_plane_dis, huber_delta, map_eig_thre, with_ua, cov_measurement_trace,
    //       max_outer, max_lm_iterations, gf_method(0 wo_gf,1 rnd,2 fps,3 gd_fix,4 gd_float), gf_ratio, seed,
    //       n_neigh, check_fov, freeze_when_degenerate]
    MapperParams p;
    p.mp.min_match_sq_dis = (float)prm[0];
    p.mp.min_plane_dis = (float)prm[1];
    p.huber_delta = prm[2];
    p.map_eig_thre = prm[3];
    p.with_ua = prm[4] != 0.0;
    p.cov_measurement_trace = prm[5];
    p.max_outer = (int)prm[6];
    p.max_lm_iterations = (int)prm[7];
    static const char *names[] = {"wo_gf", "rnd", "fps", "gd_fix", "gd_float"};
    p.sel.gf_method = names[(int)prm[8]];
    p.sel.gf_ratio = prm[9];
    p.sel.seed = (uint64_t)prm[10];
    p.n_neigh = (int)prm[11];
    p.check_fov = prm[12] != 0.0;
    p.freeze_when_degenerate = prm[13] != 0.0;
    return p;
}

// outer_stats: per outer iteration 64 doubles:
//  [0] n_surf_sel [1] n_corner_sel [2] lm iterations [3] successful steps [4] initial cost [5] final cost [6] termination
//  [7] is_degenerate [8..13] eigvals [14..20] pose_after [21..56] H0 [57] evaluations
int orc_scan2map(void *surf_map, void *corner_map, const float *surf, int surf_stride, int n_surf, int surf_cov_off,
                 const float *corner, int corner_stride, int n_corner, int corner_cov_off,
                 const double *pose_init, const double *prm, double *pose_out, double *outer_stats, int *n_outer, double *H_final)
{
    OrcMap *ms = static_cast<OrcMap *>(surf_map), *mc = static_cast<OrcMap *>(corner_map);
    MapperParams p = make_params(prm);
    Scan2MapResult res;
    scan2map_optimization(ms->mc, mc->mc, make_fc(surf, surf_stride, n_surf, surf_cov_off),
                          make_fc(corner, corner_stride, n_corner, corner_cov_off), pose_init, p, res);
    std::memcpy(pose_out, res.pose, sizeof(double) * 7);
    *n_outer = (int)res.outer.size();
    for (size_t i = 0; i < res.outer.size(); ++i) {
        double *o = outer_stats + i * 64;
        const OuterStat &s = res.outer[i];
        std::memset(o, 0, sizeof(double) * 64);
        o[0] = s.n_surf_sel; o[1] = s.n_corner_sel; o[2] = s.solve.num_iterations; o[3] = s.solve.num_successful_steps;
        o[4] = s.solve.initial_cost; o[5] = s.solve.final_cost; o[6] = s.solve.termination; o[7] = s.deg.is_degenerate;
        for (int k = 0; k < 6; ++k) o[8 + k] = s.deg.eigval[k];
        for (int k = 0; k < 7; ++k) o[14 + k] = s.pose_after[k];
        for (int k = 0; k < 36; ++k) o[21 + k] = s.H0[k];
        o[57] = s.solve.num_evaluations;
    }
    if (H_final) std::memcpy(H_final, res.H_final, sizeof(res.H_final));
    return 0;
}

// good-feature selection only (one call of goodFeatureMatching), for config-5 parity
int orc_good_feature_matching(void *map, char type, const float *feats, int stride, int n, int cov_off, const double *pose7,
                              const double *prm, int *sel_idx, int *n_sel, double *H36, unsigned char *matched, double *jaco /*n*6*/)
{
    OrcMap *m = static_cast<OrcMap *>(map);
    MapperParams p = make_params(prm);
    std::mt19937 rng((uint32_t)p.sel.seed);
    std::vector<Feature> all;
    std::vector<size_t> sel;
    double H[36];
    for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1e-6 : 0.0;
    good_feature_matching(m->mc, make_fc(feats, stride, n, cov_off), pose_from_param(pose7), all, sel, type, p.sel, H, p.mp, rng);
    *n_sel = (int)sel.size();
    for (size_t i = 0; i < sel.size(); ++i) sel_idx[i] = (int)sel[i];
    if (H36) std::memcpy(H36, H, sizeof(H));
    for (int i = 0; i < n; ++i) {
        if (matched) matched[i] = all[i].type != 'n';
        if (jaco) for (int k = 0; k < 6; ++k) jaco[size_t(i) * 6 + k] = all[i].jaco[k];
    }
    return 0;
}

int orc_eval_full_hessian(void *map, char type, const float *feats, int stride, int n, int cov_off, const double *pose7,
                          float min_match_sq_dis, float min_plane_dis, double *H36_inout, int *feat_num_inout)
{
    OrcMap *m = static_cast<OrcMap *>(map);
    MatchParams mp{min_match_sq_dis, min_plane_dis};
    eval_full_hessian(m->mc, make_fc(feats, stride, n, cov_off), pose_from_param(pose7), type, H36_inout, *feat_num_inout, mp);
    return 0;
}

// n_iters GN iterations (re-match every iteration). iter_stats: per iteration 48 doubles:
//  [0] n_surf [1] n_corner [2] cost [3] is_degenerate [4..10] pose_after [11..16] g [17..37] H upper (21)
// returns wall seconds in *seconds (matching+linearise+solve only; kd-trees already built)
int orc_gn_iterations(void *surf_map, void *corner_map, const float *surf, int surf_stride, int n_surf, int surf_cov_off,
                      const float *corner, int corner_stride, int n_corner, int corner_cov_off,
                      double *pose_inout, const double *prm, int n_iters, int n_threads, double *iter_stats, double *seconds)
{
    OrcMap *ms = static_cast<OrcMap *>(surf_map), *mc = static_cast<OrcMap *>(corner_map);
    MapperParams p = make_params(prm);
    FeatureCloud fs = make_fc(surf, surf_stride, n_surf, surf_cov_off), fc = make_fc(corner, corner_stride, n_corner, corner_cov_off);
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < n_iters; ++it) {
        GnIterStat st;
        gn_iteration(ms->mc, mc->mc, fs, fc, pose_inout, p, st, n_threads);
        if (iter_stats) {
            double *o = iter_stats + size_t(it) * 48;
            std::memset(o, 0, sizeof(double) * 48);
            o[0] = st.n_surf; o[1] = st.n_corner; o[2] = st.ne.cost; o[3] = st.deg.is_degenerate;
            for (int k = 0; k < 7; ++k) o[4 + k] = st.pose_after[k];
            for (int k = 0; k < 6; ++k) o[11 + k] = st.ne.g[k];
            int q = 17;
            for (int r = 0; r < 6; ++r) for (int c = r; c < 6; ++c) o[q++] = st.ne.H[r * 6 + c];
        }
    }
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return 0;
}

// the odometry's selection (Estimator::goodFeatureMatching, estimator.cpp:1347-1517): rel = the Pose the reference builds from T_pivot^-1 T_i T_ext
int orc_odom_good_feature_matching(void *map, char type, const float *feats, int stride, int n, const double *rel7, const double *pivot, const double *pose_i,
                                   const double *ext, float gf_ratio, unsigned seed, float min_match_sq_dis, float min_plane_dis, int *sel_idx, int *n_sel,
                                   unsigned char *matched, double *jaco /*n*6*/)
{
    OrcMap *m = static_cast<OrcMap *>(map);
    std::mt19937 rng(seed);
    std::vector<Feature> all;
    std::vector<size_t> sel;
    MatchParams mp;
    mp.min_match_sq_dis = min_match_sq_dis; mp.min_plane_dis = min_plane_dis;
    odom_good_feature_matching(m->mc, make_fc(feats, stride, n, -1), pose_from_param(rel7), pivot, pose_i, ext, all, sel, type, double(gf_ratio), mp, rng);
    *n_sel = (int)sel.size();
    for (size_t i = 0; i < sel.size(); ++i) sel_idx[i] = (int)sel[i];
    for (int i = 0; i < n; ++i) {
        if (matched) matched[i] = all[i].type != 'n';
        if (jaco) for (int k = 0; k < 6; ++k) jaco[size_t(i) * 6 + k] = all[i].jaco[k];
    }
    return 0;
}

// pure-odometry factor (3 pose blocks); J: 3 x 7
int orc_pure_odom_eval(char type, const double *point, const double *coeff, double sqrt_info, const double *pivot, const double *pose_i,
                       const double *ext, double *residual, double *J21)
{
    if (type == 's') pure_odom_plane_evaluate(point, coeff, sqrt_info, pivot, pose_i, ext, residual, J21, J21 + 7, J21 + 14);
    else pure_odom_edge_evaluate(point, coeff, sqrt_info, pivot, pose_i, ext, residual, J21, J21 + 7, J21 + 14);
    return 0;
}

// whole window in one call (timing baseline of the batched GPU evaluation): types 0 = plane, 1 = edge; J: n x 21
int orc_pure_odom_eval_batch(int n, const int *types, const double *points, const double *coeffs6, const double *sqrt_info, const int *frame_idx,
                             const int *ext_idx, const double *pivot, const double *frames, const double *exts, double *residuals, double *J)
{
    for (int i = 0; i < n; ++i) {
        double *Ji = J + size_t(i) * 21;
        const double si = sqrt_info ? sqrt_info[i] : 1.0;
        if (types[i] == 0) pure_odom_plane_evaluate(points + 3 * i, coeffs6 + 6 * i, si, pivot, frames + 7 * frame_idx[i], exts + 7 * ext_idx[i], residuals + i, Ji, Ji + 7, Ji + 14);
        else pure_odom_edge_evaluate(points + 3 * i, coeffs6 + 6 * i, si, pivot, frames + 7 * frame_idx[i], exts + 7 * ext_idx[i], residuals + i, Ji, Ji + 7, Ji + 14);
    }
    return 0;
}

// The normal equations of the coupled window problem Estimator::optimizeMap hands to Ceres (estimator.cpp:687-848): parameter blocks in
// para_ids order -- window poses [pivot, frame 1 .. n_frames] then extrinsics [0 .. n_ext) -- local size 6 each (PoseLocalParameterization:
// ComputeJacobian = [I6; 0], so the first 6 columns of every 1x7 Jacobian ARE the local Jacobian). What Estimator::evalResidual evaluates:
// problem.Evaluate on the LidarPureOdom residual blocks with every listed block variable (estimator.cpp:1577-1595), i.e. loss-corrected
// rows (HuberLoss(1.0), estimator.cpp:602; Corrector with rho'' <= 0: row and residual scaled by sqrt(rho')); J^T J is what
// evalDegenracy (estimator.cpp:1598-1680) takes its diagonal 6x6 blocks from. H: D x D row-major, D = 6 (1 + n_frames + n_ext); g = J^T r.
int orc_pure_odom_normal_eq(int n, const int *types, const double *points, const double *coeffs6, const double *sqrt_info, const int *frame_idx,
                            const int *ext_idx, const double *pivot, const double *frames, int n_frames, const double *exts, int n_ext,
                            double huber_delta, double *H, double *g, double *cost, int *n_res)
{
    const int D = 6 * (1 + n_frames + n_ext);
    std::memset(H, 0, sizeof(double) * size_t(D) * D);
    std::memset(g, 0, sizeof(double) * D);
    *cost = 0.0;
    *n_res = 0;
    for (int i = 0; i < n; ++i) {
        double r, J[21];
        const double si = sqrt_info ? sqrt_info[i] : 1.0;
        if (types[i] == 0) pure_odom_plane_evaluate(points + 3 * i, coeffs6 + 6 * i, si, pivot, frames + 7 * frame_idx[i], exts + 7 * ext_idx[i], &r, J, J + 7, J + 14);
        else pure_odom_edge_evaluate(points + 3 * i, coeffs6 + 6 * i, si, pivot, frames + 7 * frame_idx[i], exts + 7 * ext_idx[i], &r, J, J + 7, J + 14);
        double rho[3];
        huber_evaluate(huber_delta, r * r, rho);
        *cost += 0.5 * rho[0];
        (*n_res)++;
        const double s = std::sqrt(rho[1]);
        const int off[3] = {0, 6 * (1 + frame_idx[i]), 6 * (1 + n_frames + ext_idx[i])};
        double v[18];
        for (int b = 0; b < 3; ++b) for (int k = 0; k < 6; ++k) v[b * 6 + k] = J[b * 7 + k] * s;
        const double rs = r * s;
        for (int a = 0; a < 18; ++a) {
            const int ra = off[a / 6] + a % 6;
            g[ra] += v[a] * rs;
            for (int b = 0; b < 18; ++b) H[size_t(ra) * D + off[b / 6] + b % 6] += v[a] * v[b];
        }
    }
    return 0;
}

// ImageSegmenter::segmentCloud (image_segmenter.hpp:139-393). prm: [vertical_scans, horizon_scans, min_cluster_size, segment_valid_point_num,
// segment_valid_line_num, segment_theta, roi_range, segment_flag]. Buffers sized for n points (+1 outlier row); label / range images vs x hs.
int orc_segment_cloud(const float *xyzi, int n, const double *prm, float *cloud_out, int *n_out, float *outlier, int *n_outlier, int *scan_start,
                      int *scan_end, float *range_mat, int *label_mat, int *pixel_of_point)
{
    SegParams p;
    p.vertical_scans = int(prm[0]); p.horizon_scans = int(prm[1]); p.min_cluster_size = int(prm[2]); p.segment_valid_point_num = int(prm[3]);
    p.segment_valid_line_num = int(prm[4]); p.segment_theta = float(prm[5]); p.roi_range = prm[6]; p.segment_flag = prm[7] != 0.0;
    SegResult r;
    segment_cloud(xyzi, n, p, r);
    std::memcpy(cloud_out, r.cloud_out.data(), sizeof(float) * r.cloud_out.size());
    *n_out = int(r.cloud_out.size() / 4);
    std::memcpy(outlier, r.cloud_outlier.data(), sizeof(float) * r.cloud_outlier.size());
    *n_outlier = int(r.cloud_outlier.size() / 4);
    std::memcpy(scan_start, r.scan_start.data(), sizeof(int) * r.scan_start.size());
    std::memcpy(scan_end, r.scan_end.data(), sizeof(int) * r.scan_end.size());
    if (range_mat) std::memcpy(range_mat, r.range_mat.data(), sizeof(float) * r.range_mat.size());
    if (label_mat) std::memcpy(label_mat, r.label_mat.data(), sizeof(int) * r.label_mat.size());
    if (pixel_of_point) std::memcpy(pixel_of_point, r.pixel_of_point.data(), sizeof(int) * r.pixel_of_point.size());
    return 0;
}

// ---- small numeric kernels exposed for cross-checks against numpy
int orc_eig3f(const float *A9, float *val3, float *vec9)
{
    float A[3][3];
    std::memcpy(A, A9, sizeof(A));
    Eig3f e = eig3_sym_f(A);
    std::memcpy(val3, e.val, sizeof(e.val));
    std::memcpy(vec9, e.vec, sizeof(e.vec));
    return e.ok ? 0 : 1;
}
int orc_qr_solve(const float *A, const float *b, int rows, float *x3) { colpiv_qr_solve_f(A, b, rows, x3); return 0; }
double orc_logdet(const double *A, int n) { return logdet_chol_d(A, n); }

// ---- uncertainty / covariance voxel grid (next rows)
int orc_eval_point_uncertainty(const float *xyz, int n, int stride, const double *pose7, const double *cov_pose36,
                               const double *cov_meas9, double *cov_out /*n*9*/)
{
    for (int i = 0; i < n; ++i) eval_point_uncertainty(xyz + size_t(i) * stride, pose7, cov_pose36, cov_meas9, cov_out + size_t(i) * 9);
    return 0;
}

int orc_voxel_grid_cov(const float *pts11, int n, float leaf, float trace_threshold, float *out11, int *n_out)
{
    std::vector<PointICov> o;
    voxel_grid_covariance_mloam(reinterpret_cast<const PointICov *>(pts11), n, leaf, trace_threshold, o);
    std::memcpy(out11, o.data(), sizeof(PointICov) * o.size());
    *n_out = (int)o.size();
    return 0;
}

int orc_voxel_grid_mloam_plain(const float *xyzi, int n, float leaf, int member_order, float *out, int *n_out)
{
    std::vector<float> o;
    voxel_grid_mloam_plain(xyzi, n, leaf, member_order, o);
    std::memcpy(out, o.data(), sizeof(float) * o.size());
    *n_out = int(o.size() / 4);
    return 0;
}

// The permutation std::sort leaves when its comparator sees the key only -- pcl's cloud_point_index_idx::operator< (voxel_grid.h), as the
// reference's voxel filters call it (voxel_grid_covariance_mloam_impl.hpp:227). Plain std::sort, one thread: the checker of stdsort.hip.
int orc_std_sort_permutation(const int *keys, int n, int *perm)
{
    struct IdxPt {
        unsigned idx, cloud_point_index;
        bool operator<(const IdxPt &o) const { return idx < o.idx; }
    };
    std::vector<IdxPt> iv(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) iv[size_t(i)] = IdxPt{unsigned(keys[i]), unsigned(i)};
    std::sort(iv.begin(), iv.end(), std::less<IdxPt>());
    for (int i = 0; i < n; ++i) perm[i] = int(iv[size_t(i)].cloud_point_index);
    return 0;
}

int orc_compound_pose_with_cov(const double *pose1, const double *cov1, const double *pose2, const double *cov2, double *pose_cp, double *cov_cp)
{
    compound_pose_with_cov(pose1, cov1, pose2, cov2, pose_cp, cov_cp);
    return 0;
}

int orc_cloud_uct_associate_to_map(const float *pts11, int n, const double *pose_global, const double *cov_global, const double *ext,
                                   const double *ext_cov, int n_laser, const double *cov_meas9, int with_ua, double trace_threshold,
                                   float *out11, int *n_out)
{
    std::vector<PointICov> o;
    cloud_uct_associate_to_map(reinterpret_cast<const PointICov *>(pts11), n, pose_global, cov_global, ext, ext_cov, n_laser, cov_meas9,
                               with_ua != 0, trace_threshold, o);
    std::memcpy(out11, o.data(), sizeof(PointICov) * o.size());
    *n_out = (int)o.size();
    return 0;
}

// ---- scan-to-scan tracker (next row 4)
static TrackParams make_track_params(const double *p)
{
    TrackParams t;
    t.distance_sq_threshold = float(p[0]); t.nearby_scan = float(p[1]); t.scan_period = float(p[2]); t.huber_delta = p[3];
    t.max_outer = int(p[4]); t.max_lm_iterations = int(p[5]);
    return t;
}

// kind 'c' / 's'; prev, cur: n x 4 [x y z ring]; outputs: valid[m], coeffs[m*6]
int orc_track_match(char kind, const float *prev, int n_prev, const float *cur, int m, const double *pose7, const double *tprm,
                    unsigned char *valid, double *coeffs)
{
    ScanCloud sc;
    sc.set(prev, 4, n_prev);
    const TrackParams tp = make_track_params(tprm);
    std::vector<Feature> f;
    if (kind == 'c') match_corner_from_scan(sc, cur, 4, m, pose_from_param(pose7), tp, f);
    else match_surf_from_scan(sc, cur, 4, m, pose_from_param(pose7), tp, f);
    std::memset(valid, 0, size_t(m));
    std::memset(coeffs, 0, sizeof(double) * 6 * size_t(m));
    for (const Feature &ft : f) { valid[ft.idx] = 1; std::memcpy(coeffs + ft.idx * 6, ft.coeffs, sizeof(double) * 6); }
    return (int)f.size();
}

// type 'S' (scan plane, 1 residual) / 'E' (scan edge vector, 3 residuals); J: rows x 7
int orc_scan_factor_eval(char type, const double *point, const double *coeff, double s, const double *pose7, double *residual, double *J)
{
    if (type == 'S') scan_plane_factor_evaluate(point, coeff, s, pose7, residual, J);
    else scan_edge_vector_factor_evaluate(point, coeff, s, pose7, residual, J);
    return 0;
}

// stats: per outer iteration 16 doubles: [0] n_corner [1] n_surf [2] solved [3] lm iterations [4] initial cost [5] final cost
// [6] termination [7..13] pose_after
int orc_track_cloud(const float *corner_last, int n_cl, const float *surf_last, int n_sl, const float *corner_sharp, int n_cs,
                    const float *surf_flat, int n_sf, const double *pose_ini, const double *tprm, double *pose_out, double *stats, int *n_outer)
{
    ScanCloud cl, sl;
    cl.set(corner_last, 4, n_cl);
    sl.set(surf_last, 4, n_sl);
    const TrackParams tp = make_track_params(tprm);
    std::vector<TrackOuterStat> st;
    track_cloud(cl, sl, corner_sharp, 4, n_cs, surf_flat, 4, n_sf, pose_ini, tp, pose_out, st);
    *n_outer = (int)st.size();
    for (size_t i = 0; i < st.size(); ++i) {
        double *o = stats + i * 16;
        std::memset(o, 0, sizeof(double) * 16);
        o[0] = st[i].n_corner; o[1] = st[i].n_surf; o[2] = st[i].solved; o[3] = st[i].solve.num_iterations;
        o[4] = st[i].solve.initial_cost; o[5] = st[i].solve.final_cost; o[6] = st[i].solve.termination;
        for (int k = 0; k < 7; ++k) o[7 + k] = st[i].pose_after[k];
    }
    return 0;
}

// TransformToEnd over n rows [x y z intensity] -> out rows (intensity copied)
int orc_transform_to_end(const float *pts, int n, const double *pose7, int b_distortion, float scan_period, float *out)
{
    const Pose pose = pose_from_param(pose7);
    for (int i = 0; i < n; ++i) {
        transform_to_end(pts + size_t(i) * 4, pose, b_distortion != 0, scan_period, out + size_t(i) * 4);
        out[size_t(i) * 4 + 3] = pts[size_t(i) * 4 + 3];
    }
    return 0;
}

}  // extern "C"
