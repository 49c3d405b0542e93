// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (Ceres' trust-region LM, the selection loops and evalDegenracy are restated; the
// factors they evaluate are pinned through factors.hpp, the matches through feature_extract.cpp).
// See mapper.hpp for the list of reference functions restated here.
#include "mapper.hpp"
#include "tracker.hpp"
#include "linalg.hpp"
#include <queue>
#include <numeric>
#include <cstring>
#include <cstdio>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

static inline void add_outer(double H[36], const double j[6])
{
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[r * 6 + c] += j[r] * j[c];
}

void evaluate_problem(const std::vector<ResidualBlock> &blocks, const double x[7], double huber_delta,
                      NormalEq &ne, bool with_jacobian)
{
    std::memset(&ne, 0, sizeof(ne));
    for (const ResidualBlock &b : blocks) {
        // 's' / 'c': map factors (1 residual); 'S' / 'E': scan-to-scan factors of the tracker (1 / 3 residuals, sqrt_info carries s)
        double r[3], J[21];
        int rows = 1;
        if (b.type == 's') plane_norm_factor_evaluate(b.point, b.coeffs, b.sqrt_info, x, r, with_jacobian ? J : nullptr);
        else if (b.type == 'c') edge_factor_evaluate(b.point, b.coeffs, b.sqrt_info, x, r, with_jacobian ? J : nullptr);
        else if (b.type == 'S') scan_plane_factor_evaluate(b.point, b.coeffs, b.sqrt_info, x, r, with_jacobian ? J : nullptr);
        else { scan_edge_vector_factor_evaluate(b.point, b.coeffs, b.sqrt_info, x, r, with_jacobian ? J : nullptr); rows = 3; }
        double sq = 0.0;
        for (int q = 0; q < rows; ++q) sq += r[q] * r[q];
        double rho[3];
        huber_evaluate(huber_delta, sq, rho);
        ne.cost += 0.5 * rho[0];
        ne.n++;
        if (with_jacobian) {
            // Corrector with rho'' <= 0: residuals and Jacobian rows of the block scaled by sqrt(rho')
            const double s = std::sqrt(rho[1]);
            for (int q = 0; q < rows; ++q) {
                double *Jq = J + q * 7;
                const double rq = r[q] * s;
                for (int k = 0; k < 6; ++k) Jq[k] *= s;
                add_outer(ne.H, Jq);
                for (int k = 0; k < 6; ++k) ne.g[k] += Jq[k] * rq;
            }
        }
    }
}

void eval_degeneracy(const double H[36], double eig_thre, Degeneracy &out)
{
    jacobi_eig_sym_d(H, 6, out.eigval, out.eigvec);
    double Vp[36];
    std::memcpy(Vp, out.eigvec, sizeof(Vp));
    out.is_degenerate = false;
    for (int j = 0; j < 6; ++j) {
        if (out.eigval[j] < eig_thre) {
            for (int r = 0; r < 6; ++r) Vp[r * 6 + j] = 0.0;
            out.is_degenerate = true;
        } else break;
    }
    // mat_P = (V_f^T)^-1 * V_p^T   (lidar_mapper_keyframe.cpp:1192)
    double Vft[36], Vft_inv[36];
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Vft[r * 6 + c] = out.eigvec[c * 6 + r];
    inverse_d(Vft, 6, Vft_inv);
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            for (int k = 0; k < 6; ++k) s += Vft_inv[r * 6 + k] * Vp[c * 6 + k];   // V_p^T(k,c) = Vp(c,k)
            out.V_update[r * 6 + c] = s;
        }
    if (!out.is_degenerate) {   // V_update_ stays Identity (setParameter) unless degenerate
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) out.V_update[r * 6 + c] = (r == c) ? 1.0 : 0.0;
    }
}

void window_eval_degeneracy(const double *JtJ, int D, int n_pose_blocks, double *eig_thre, bool estimate_extrinsic, long frame_cnt,
                            int n_cumu_feature, double lambda_thre_calib, int *is_degenerate, double *V_update, double *eigval, double *d_factor_calib)
{
    const int n_blocks = D / 6;
    auto block = [&](int i, double H[36]) {
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[r * 6 + c] = JtJ[size_t(6 * i + r) * D + 6 * i + c];
    };
    auto identity = [](double *V) { for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) V[r * 6 + c] = (r == c) ? 1.0 : 0.0; };
    // poses (cpp:1610-1636): the mapper's rule per diagonal block, with the block's own threshold
    for (int i = 0; i < n_pose_blocks && i < n_blocks; ++i) {
        double H[36];
        block(i, H);
        Degeneracy d;
        eval_degeneracy(H, eig_thre[i], d);
        is_degenerate[i] = d.is_degenerate ? 1 : 0;
        std::memcpy(V_update + 36 * i, d.V_update, sizeof(d.V_update));
        std::memcpy(eigval + 6 * i, d.eigval, sizeof(d.eigval));
    }
    // extrinsics (cpp:1638-1678)
    for (int i = n_pose_blocks; i < n_blocks; ++i) {
        is_degenerate[i] = 0;
        identity(V_update + 36 * i);
        for (int k = 0; k < 6; ++k) eigval[6 * i + k] = 0.0;
        d_factor_calib[i - n_pose_blocks] = 0.0;
        if (!estimate_extrinsic) continue;
        bool freeze = false;
        if (frame_cnt % n_cumu_feature == 0) {          // "need to optimize the extrinsics"
            double H[36], vec[36];
            block(i, H);
            jacobi_eig_sym_d(H, 6, eigval + 6 * i, vec);
            const double lambda = eigval[6 * i] / n_cumu_feature;
            if (lambda >= lambda_thre_calib) { eig_thre[i] = lambda_thre_calib; d_factor_calib[i - n_pose_blocks] = lambda; }
            else if (lambda > eig_thre[i]) eig_thre[i] = lambda;
            else freeze = true;                          // degenerate for calibration: the extrinsic is not updated
        } else {
            freeze = true;                               // not enough accumulated features
        }
        if (freeze) { is_degenerate[i] = 1; std::memset(V_update + 36 * i, 0, sizeof(double) * 36); }
    }
}

void ceres_like_solve(const std::vector<ResidualBlock> &blocks, double x[7], const double V_update[36],
                      double huber_delta, int max_num_iterations, SolveSummary &sum)
{
    ceres_like_solve_generic([&](const double *at, NormalEq &ne) { evaluate_problem(blocks, at, huber_delta, ne, true); },
                             [&](const double *at, const double *delta, double *out) { pose_plus(at, delta, V_update, out); }, x, max_num_iterations, sum);
}

// lidar_mapper.h:130-174: weighted, NOT loss-corrected 1x6 Jacobian at pose_local
static void evaluate_feat_jacobian_matching(const Pose &pose_local, Feature &feature, double cov_trace)
{
    double param[7] = {pose_local.t.x, pose_local.t.y, pose_local.t.z, pose_local.q.x, pose_local.q.y, pose_local.q.z, pose_local.q.w};
    double res, jaco[7];
    double sqrt_info = sqrt_info_from_trace(cov_trace);
    if (feature.type == 's') plane_norm_factor_evaluate(feature.point, feature.coeffs, sqrt_info, param, &res, jaco);
    else if (feature.type == 'c') edge_factor_evaluate(feature.point, feature.coeffs, sqrt_info, param, &res, jaco);
    else return;
    for (int k = 0; k < 6; ++k) feature.jaco[k] = jaco[k];
}

static inline bool match_one(const MapCloud &map, const FeatureCloud &cloud, size_t que_idx, const Pose &pose_local,
                             Feature &f, char feature_type, const MatchParams &mp)
{
    if (feature_type == 's') return match_surf_point_from_map(map, cloud.at(que_idx), pose_local, f, que_idx, 5, false, mp);
    if (feature_type == 'c') return match_corner_point_from_map(map, cloud.at(que_idx), pose_local, f, que_idx, 5, false, mp);
    return false;
}

void eval_full_hessian(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose_local, char feature_type,
                       double mat_H[36], int &feat_num, const MatchParams &mp)
{
    for (size_t i = 0; i < (size_t)cloud.n; i++) {
        Feature f;
        if (!match_one(map, cloud, i, pose_local, f, feature_type, mp)) continue;
        evaluate_feat_jacobian_matching(pose_local, f, cloud.cov_trace(i));
        add_outer(mat_H, f.jaco);
        feat_num++;
    }
}

namespace {
struct FeatureWithScore {   // parameters.h:177-191
    size_t idx; double score; double jaco[6];
    bool operator<(const FeatureWithScore &o) const { return score < o.score; }
};
inline size_t rand_uniform(std::mt19937 &rng, size_t lo, size_t hi)
{
    std::uniform_int_distribution<size_t> d(lo, hi);   // RandomGeneratorInt::geneRandUniform
    return d(rng);
}
}  // namespace

void good_feature_matching(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose_local,
                           std::vector<Feature> &all_features, std::vector<size_t> &sel_feature_idx,
                           char feature_type, const SelectParams &sp, double sub_mat_H[36],
                           const MatchParams &mp, std::mt19937 &rng)
{
    const size_t num_all_features = cloud.n;
    all_features.assign(num_all_features, Feature());
    std::vector<size_t> all_feature_idx(num_all_features);
    std::vector<int> feature_visited(num_all_features, -1);
    std::iota(all_feature_idx.begin(), all_feature_idx.end(), 0);
    size_t num_use_features = static_cast<size_t>(num_all_features * sp.gf_ratio);
    sel_feature_idx.assign(num_use_features, 0);
    size_t num_sel_features = 0;
    const std::string &gf_method = sp.gf_method;
    const size_t MAX_RANDOM_QUEUE_TIME = 20;

    auto match_and_jac = [&](size_t que_idx) -> bool {
        bool b = match_one(map, cloud, que_idx, pose_local, all_features[que_idx], feature_type, mp);
        if (b) evaluate_feat_jacobian_matching(pose_local, all_features[que_idx], cloud.cov_trace(que_idx));
        return b;
    };

    if (gf_method == "wo_gf") {
        for (size_t j = 0; j < all_feature_idx.size(); j++) {
            size_t que_idx = all_feature_idx[j];
            if (match_and_jac(que_idx)) {
                add_outer(sub_mat_H, all_features[que_idx].jaco);
                if (num_sel_features >= sel_feature_idx.size()) sel_feature_idx.resize(num_sel_features + 1);   // gf_ratio < 1 with wo_gf overruns in the reference
                sel_feature_idx[num_sel_features++] = que_idx;
            }
        }
    } else if (gf_method == "rnd") {
        while (true) {
            if (num_sel_features >= num_use_features || all_feature_idx.size() == 0) break;
            size_t j = rand_uniform(rng, 0, all_feature_idx.size() - 1);
            size_t que_idx = all_feature_idx[j];
            if (match_and_jac(que_idx)) {
                add_outer(sub_mat_H, all_features[que_idx].jaco);
                sel_feature_idx[num_sel_features++] = que_idx;
            }
            all_feature_idx.erase(all_feature_idx.begin() + j);
        }
    } else if (gf_method == "fps") {
        if (num_all_features > 0) {
            size_t k = rand_uniform(rng, 0, all_feature_idx.size() - 1);
            feature_visited[k] = 1;
            size_t cnt_visited = 1;
            const float *point_old = cloud.at(k);
            // the starting point is matched but its Jacobian is neither evaluated nor accumulated (lidar_mapper.h:356-386)
            if (match_one(map, cloud, k, pose_local, all_features[k], feature_type, mp) && num_use_features > 0)
                sel_feature_idx[num_sel_features++] = k;
            std::vector<float> dist(num_all_features, 1e5);
            while (true) {
                // The reference's loop has no "everything visited" exit (lidar_mapper.h:391-399: that test is commented out). Once every point has been visited the
                // scan below skips them all and leaves best_j at its initial value 1, so feature 1 is matched again on every further round: if it matches it is
                // appended (and its J^T J added) AGAIN, round after round, until the count is reached; if it does not, the loop spins until its 20 ms wall-clock
                // cut-off and returns what it has. Both outcomes are deterministic and restated here (the spin as an immediate exit). One feature only:
                // points[1] does not exist, the reference reads past its cloud -- stop.
                if (num_sel_features >= num_use_features) break;
                if (cnt_visited >= num_all_features) {
                    if (num_all_features < 2 || !match_and_jac(1)) break;
                    while (num_sel_features < num_use_features) {
                        add_outer(sub_mat_H, all_features[1].jaco);
                        sel_feature_idx[num_sel_features++] = 1;
                    }
                    break;
                }
                float best_d = -1;
                size_t best_j = 1;
                for (size_t j = 0; j < num_all_features; j++) {
                    if (feature_visited[j] == 1) continue;
                    const float *pn = cloud.at(j);
                    float ddx = point_old[0] - pn[0], ddy = point_old[1] - pn[1], ddz = point_old[2] - pn[2];
                    float d = std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
                    float d2 = std::min(d, dist[j]);
                    dist[j] = d2;
                    best_j = d2 > best_d ? j : best_j;
                    best_d = d2 > best_d ? d2 : best_d;
                }
                size_t que_idx = best_j;
                point_old = cloud.at(que_idx);
                feature_visited[que_idx] = 1;
                cnt_visited++;
                if (match_and_jac(que_idx)) {
                    add_outer(sub_mat_H, all_features[que_idx].jaco);
                    sel_feature_idx[num_sel_features++] = que_idx;
                }
            }
        }
    } else if (gf_method == "gd_fix" || gf_method == "gd_float") {
        size_t num_rnd_que = 0;
        while (true) {
            if (num_sel_features >= num_use_features || all_feature_idx.size() == 0) break;
            size_t size_rnd_subset = static_cast<size_t>(1.0 * num_all_features / num_use_features);
            std::priority_queue<FeatureWithScore, std::vector<FeatureWithScore>, std::less<FeatureWithScore>> heap_subset;
            bool lost = false;
            while (true) {
                if (all_feature_idx.size() == 0) break;
                num_rnd_que = 0;
                size_t j = 0;
                while (num_rnd_que < MAX_RANDOM_QUEUE_TIME) {
                    j = rand_uniform(rng, 0, all_feature_idx.size() - 1);
                    if (feature_visited[j] < int(num_sel_features)) {
                        feature_visited[j] = int(num_sel_features);
                        break;
                    }
                    num_rnd_que++;
                }
                if (num_rnd_que >= MAX_RANDOM_QUEUE_TIME) break;
                size_t que_idx = all_feature_idx[j];
                if (all_features[que_idx].type == 'n') {
                    if (!match_and_jac(que_idx)) {
                        all_feature_idx.erase(all_feature_idx.begin() + j);
                        feature_visited.erase(feature_visited.begin() + j);
                        continue;
                    }
                }
                const double *jaco = all_features[que_idx].jaco;
                double Ht[36];
                std::memcpy(Ht, sub_mat_H, sizeof(Ht));
                add_outer(Ht, jaco);
                FeatureWithScore fws;
                fws.idx = que_idx; fws.score = logdet_chol_d(Ht, 6);
                std::memcpy(fws.jaco, jaco, sizeof(fws.jaco));
                heap_subset.push(fws);
                if (heap_subset.size() >= size_rnd_subset) {
                    const FeatureWithScore &top = heap_subset.top();
                    auto iter = std::find(all_feature_idx.begin(), all_feature_idx.end(), top.idx);
                    if (iter == all_feature_idx.end()) { lost = true; break; }
                    add_outer(sub_mat_H, top.jaco);
                    size_t position = iter - all_feature_idx.begin();
                    all_feature_idx.erase(all_feature_idx.begin() + position);
                    feature_visited.erase(feature_visited.begin() + position);
                    sel_feature_idx[num_sel_features++] = top.idx;
                    break;
                }
            }
            if (num_rnd_que >= MAX_RANDOM_QUEUE_TIME || lost) break;
        }
    }
    sel_feature_idx.resize(num_sel_features);
}

// Estimator::goodFeatureMatching (estimator.cpp:1347-1517) with Estimator::evaluateFeatJacobian (:1273-1345): the ODOMETRY's selection in front of the window's
// residual blocks (buildLocalMap, estimator.cpp:1241-1263, gf_ratio = ODOM_GF_RATIO: 0.8 in every shipped configuration). gf_ratio == 1.0: every feature is
// matched, the matched ones are kept in feature order. Otherwise the stochastic-greedy loop of the mapper's gd_fix (same draws, same stamps, same heap), with
//   - the row that is scored: a surf feature's LidarPureOdomPlaneNormFactor(point, coeffs, 1.0) evaluated at (pivot, pose_i, ext), its FRAME block's first six
//     columns; a corner feature's row is Matrix<1, 6>::Identity() -- (1 0 0 0 0 0), whatever the feature (cpp:1339-1342);
//   - no uncertainty weights, sub_mat_H seeded with 1e-6 I inside the function (cpp:1372);
//   - ten failed draws (MAX_RANDOM_QUEUE_TIME, estimator.h:63) do NOT end the selection here (the `break` behind the message is missing, cpp:1505-1509): the outer loop starts over with an empty heap
//     -- whatever the heap held is forgotten, its stamps stay -- and only the 7 ms wall-clock cut-off (MAX_FEATURE_SELECT_TIME, estimator.h:62) ends a loop in
//     which no draw can succeed any more. Restated without the clock: the loop ends when no pool entry is left that the current round may still draw.
// gf_ratio arrives as the float ODOM_GF_RATIO widened to double (parameters.cpp:85): num_use = size_t(n * double(float(ratio))).
void odom_good_feature_matching(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose_local, const double pivot[7], const double pose_i[7],
                                const double ext[7], std::vector<Feature> &all_features, std::vector<size_t> &sel_feature_idx, char feature_type,
                                double gf_ratio, const MatchParams &mp, std::mt19937 &rng)
{
    const size_t num_all_features = cloud.n;
    all_features.assign(num_all_features, Feature());
    std::vector<size_t> all_feature_idx(num_all_features);
    std::vector<int> feature_visited(num_all_features, -1);
    std::iota(all_feature_idx.begin(), all_feature_idx.end(), 0);
    const size_t num_use_features = static_cast<size_t>(num_all_features * gf_ratio);
    sel_feature_idx.assign(num_use_features, 0);
    const size_t size_rnd_subset = num_use_features ? static_cast<size_t>(1.0 * num_all_features / num_use_features) : 0;
    double sub_mat_H[36];
    for (int i = 0; i < 36; ++i) sub_mat_H[i] = (i % 7 == 0) ? 1e-6 : 0.0;
    size_t num_sel_features = 0;
    const size_t MAX_RANDOM_QUEUE_TIME = 10;      // estimator.h:63 (the mapper's is 20, lidar_mapper.h:83)
    auto jac = [&](Feature &f) {
        if (f.type == 's') {
            double r, J0[7], J1[7], J2[7];
            pure_odom_plane_evaluate(f.point, f.coeffs, 1.0, pivot, pose_i, ext, &r, J0, J1, J2);
            for (int k = 0; k < 6; ++k) f.jaco[k] = J1[k];
        } else if (f.type == 'c') {
            for (int k = 0; k < 6; ++k) f.jaco[k] = k == 0 ? 1.0 : 0.0;
        }
    };
    if (gf_ratio == 1.0) {
        for (size_t j = 0; j < all_feature_idx.size(); j++) {
            const size_t que_idx = all_feature_idx[j];
            if (match_one(map, cloud, que_idx, pose_local, all_features[que_idx], feature_type, mp)) {
                if (num_sel_features >= sel_feature_idx.size()) sel_feature_idx.resize(num_sel_features + 1);
                sel_feature_idx[num_sel_features++] = que_idx;
            }
        }
    } else {
        size_t num_rnd_que = 0;
        while (true) {
            if (num_sel_features >= num_use_features || all_feature_idx.size() == 0) break;
            std::priority_queue<FeatureWithScore, std::vector<FeatureWithScore>, std::less<FeatureWithScore>> heap_subset;
            bool lost = false;
            while (true) {
                if (all_feature_idx.size() == 0) break;
                num_rnd_que = 0;
                size_t j = 0;
                while (num_rnd_que < MAX_RANDOM_QUEUE_TIME) {
                    j = rand_uniform(rng, 0, all_feature_idx.size() - 1);
                    if (feature_visited[j] < int(num_sel_features)) {
                        feature_visited[j] = int(num_sel_features);
                        break;
                    }
                    num_rnd_que++;
                }
                if (num_rnd_que >= MAX_RANDOM_QUEUE_TIME) break;
                const size_t que_idx = all_feature_idx[j];
                if (all_features[que_idx].type == 'n') {
                    if (match_one(map, cloud, que_idx, pose_local, all_features[que_idx], feature_type, mp)) jac(all_features[que_idx]);
                    else {
                        all_feature_idx.erase(all_feature_idx.begin() + j);
                        feature_visited.erase(feature_visited.begin() + j);
                        continue;
                    }
                }
                const double *jaco = all_features[que_idx].jaco;
                double Ht[36];
                std::memcpy(Ht, sub_mat_H, sizeof(Ht));
                add_outer(Ht, jaco);
                FeatureWithScore fws;
                fws.idx = que_idx; fws.score = logdet_chol_d(Ht, 6);
                std::memcpy(fws.jaco, jaco, sizeof(fws.jaco));
                heap_subset.push(fws);
                if (heap_subset.size() >= size_rnd_subset) {
                    const FeatureWithScore &top = heap_subset.top();
                    auto iter = std::find(all_feature_idx.begin(), all_feature_idx.end(), top.idx);
                    if (iter == all_feature_idx.end()) { lost = true; break; }     // "not exist feature idx": leaves the inner loop only (cpp:1487-1491)
                    add_outer(sub_mat_H, top.jaco);
                    const size_t position = iter - all_feature_idx.begin();
                    all_feature_idx.erase(all_feature_idx.begin() + position);
                    feature_visited.erase(feature_visited.begin() + position);
                    sel_feature_idx[num_sel_features++] = top.idx;
                    break;
                }
            }
            (void)lost;
            if (num_rnd_que >= MAX_RANDOM_QUEUE_TIME) {
                // the reference prints "early termination" and goes on; it can only leave through its clock when nothing is left to draw
                bool drawable = false;
                for (int v : feature_visited) if (v < int(num_sel_features)) { drawable = true; break; }
                if (!drawable) break;
            }
        }
    }
    sel_feature_idx.resize(num_sel_features);
}

static void build_blocks(const std::vector<Feature> &all_surf, const std::vector<size_t> &sel_surf, const FeatureCloud &surf,
                         const std::vector<Feature> &all_corner, const std::vector<size_t> &sel_corner, const FeatureCloud &corner,
                         const MapperParams &prm, std::vector<ResidualBlock> &blocks)
{
    blocks.clear();
    blocks.reserve(sel_surf.size() + sel_corner.size());
    for (size_t fid : sel_surf) {   // cpp:537-548
        const Feature &f = all_surf[fid];
        ResidualBlock b;
        b.type = 's';
        std::memcpy(b.point, f.point, sizeof(b.point));
        std::memcpy(b.coeffs, f.coeffs, sizeof(b.coeffs));
        b.sqrt_info = sqrt_info_from_trace(prm.with_ua ? surf.cov_trace(f.idx) : prm.cov_measurement_trace);
        blocks.push_back(b);
    }
    for (size_t fid : sel_corner) {   // cpp:551-571
        const Feature &f = all_corner[fid];
        ResidualBlock b;
        b.type = 'c';
        std::memcpy(b.point, f.point, sizeof(b.point));
        std::memcpy(b.coeffs, f.coeffs, sizeof(b.coeffs));
        b.sqrt_info = sqrt_info_from_trace(prm.with_ua ? corner.cov_trace(f.idx) : prm.cov_measurement_trace);
        blocks.push_back(b);
    }
}

void scan2map_optimization(const MapCloud &surf_map, const MapCloud &corner_map,
                           const FeatureCloud &surf, const FeatureCloud &corner,
                           const double pose_init[7], const MapperParams &prm, Scan2MapResult &res)
{
    res = Scan2MapResult();
    std::memcpy(res.pose, pose_init, sizeof(double) * 7);
    std::memset(res.H_final, 0, sizeof(res.H_final));
    if (!(surf_map.n > 50 && corner_map.n > 10)) return;   // cpp:429
    std::mt19937 rng((uint32_t)prm.sel.seed);
    for (int iter_cnt = 0; iter_cnt < prm.max_outer; iter_cnt++) {
        OuterStat st;
        Pose pose_wmap_curr = pose_from_param(res.pose);
        std::vector<Feature> all_surf, all_corner;
        std::vector<size_t> sel_surf, sel_corner;
        double sub_H[36];
        for (int i = 0; i < 36; ++i) sub_H[i] = (i % 7 == 0) ? 1e-6 : 0.0;
        good_feature_matching(corner_map, corner, pose_wmap_curr, all_corner, sel_corner, 'c', prm.sel, sub_H, prm.mp, rng);
        for (int i = 0; i < 36; ++i) sub_H[i] = (i % 7 == 0) ? 1e-6 : 0.0;
        good_feature_matching(surf_map, surf, pose_wmap_curr, all_surf, sel_surf, 's', prm.sel, sub_H, prm.mp, rng);
        st.n_surf_sel = (int)sel_surf.size();
        st.n_corner_sel = (int)sel_corner.size();
        std::vector<ResidualBlock> blocks;
        build_blocks(all_surf, sel_surf, surf, all_corner, sel_corner, corner, prm, blocks);

        NormalEq ne;
        evaluate_problem(blocks, res.pose, prm.huber_delta, ne, true);   // problem.Evaluate -> evalHessian (cpp:575-581)
        std::memcpy(st.H0, ne.H, sizeof(st.H0));
        if (ne.n == 0) std::memset(st.H0, 0, sizeof(st.H0));
        eval_degeneracy(st.H0, prm.map_eig_thre, st.deg);
        ceres_like_solve(blocks, res.pose, st.deg.V_update, prm.huber_delta, prm.max_lm_iterations, st.solve);
        if (iter_cnt == prm.max_outer - 1 && prm.with_ua) {
            NormalEq nf;
            evaluate_problem(blocks, res.pose, prm.huber_delta, nf, true);
            std::memcpy(res.H_final, nf.H, sizeof(res.H_final));
        }
        std::memcpy(st.pose_after, res.pose, sizeof(st.pose_after));
        res.outer.push_back(st);
    }
}

static void match_all_parallel(const MapCloud &map, const FeatureCloud &cloud, const Pose &pose, char type,
                               const MapperParams &prm, std::vector<Feature> &feats, std::vector<uint8_t> &ok, int n_threads)
{
    feats.assign(cloud.n, Feature());
    ok.assign(cloud.n, 0);
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 256) if (n_threads > 1)
    for (int i = 0; i < cloud.n; ++i) {
        bool b = (type == 's') ? match_surf_point_from_map(map, cloud.at(i), pose, feats[i], i, prm.n_neigh, prm.check_fov, prm.mp)
                               : match_corner_point_from_map(map, cloud.at(i), pose, feats[i], i, prm.n_neigh, prm.check_fov, prm.mp);
        ok[i] = b ? 1 : 0;
    }
}

void gn_iteration(const MapCloud &surf_map, const MapCloud &corner_map, const FeatureCloud &surf, const FeatureCloud &corner,
                  double x[7], const MapperParams &prm, GnIterStat &st, int n_threads)
{
    Pose pose = pose_from_param(x);
    std::vector<Feature> fs, fc;
    std::vector<uint8_t> oks, okc;
    match_all_parallel(surf_map, surf, pose, 's', prm, fs, oks, n_threads);
    match_all_parallel(corner_map, corner, pose, 'c', prm, fc, okc, n_threads);
    std::vector<size_t> sel_s, sel_c;
    for (int i = 0; i < surf.n; ++i) if (oks[i]) sel_s.push_back(i);
    for (int i = 0; i < corner.n; ++i) if (okc[i]) sel_c.push_back(i);
    st.n_surf = (int)sel_s.size();
    st.n_corner = (int)sel_c.size();
    std::vector<ResidualBlock> blocks;
    build_blocks(fs, sel_s, surf, fc, sel_c, corner, prm, blocks);
    evaluate_problem(blocks, x, prm.huber_delta, st.ne, true);
    eval_degeneracy(st.ne.H, prm.map_eig_thre, st.deg);
    if (prm.freeze_when_degenerate && st.deg.is_degenerate) std::memset(st.deg.V_update, 0, sizeof(st.deg.V_update));
    double rhs[6], d[6];
    for (int i = 0; i < 6; ++i) rhs[i] = -st.ne.g[i];
    bool ok = chol_solve_d(st.ne.H, rhs, 6, d);
    if (!ok) {
        double Hd[36];
        std::memcpy(Hd, st.ne.H, sizeof(Hd));
        for (int i = 0; i < 6; ++i) Hd[i * 6 + i] += 1e-6;
        ok = chol_solve_d(Hd, rhs, 6, d);
    }
    if (ok) {
        double xn[7];
        pose_plus(x, d, st.deg.V_update, xn);
        std::memcpy(x, xn, sizeof(xn));
    }
    std::memcpy(st.pose_after, x, sizeof(st.pose_after));
}

// lidar_tracker.cpp:23-129: two rounds of { match sharp corners / flat surfs against the previous frame's less-sharp / less-flat
// clouds at the current estimate, Ceres (Huber 0.1, <= 4 iterations, identity V_update) on the fixed correspondences }
void track_cloud(const ScanCloud &corner_last, const ScanCloud &surf_last, const float *corner_sharp, size_t cs_stride, int n_corner,
                 const float *surf_flat, size_t sf_stride, int n_surf, const double pose_ini[7], const TrackParams &tp, double pose_out[7],
                 std::vector<TrackOuterStat> &stats)
{
    double x[7];
    std::memcpy(x, pose_ini, sizeof(x));
    double V[36];
    for (int i = 0; i < 36; ++i) V[i] = (i % 7 == 0) ? 1.0 : 0.0;
    stats.clear();
    for (int iter = 0; iter < tp.max_outer; ++iter) {
        TrackOuterStat st;
        const Pose pose_local = pose_from_param(x);
        std::vector<Feature> cf, sf;
        match_corner_from_scan(corner_last, corner_sharp, cs_stride, n_corner, pose_local, tp, cf);
        match_surf_from_scan(surf_last, surf_flat, sf_stride, n_surf, pose_local, tp, sf);
        st.n_corner = (int)cf.size(); st.n_surf = (int)sf.size();
        if (cf.size() + sf.size() >= 10) {      // cpp:66-70: "less correspondence" -> continue
            std::vector<ResidualBlock> blocks;
            blocks.reserve(cf.size() + sf.size());
            for (const Feature &f : sf) { ResidualBlock b; b.type = 'S'; std::memcpy(b.point, f.point, sizeof(b.point)); std::memcpy(b.coeffs, f.coeffs, sizeof(b.coeffs)); b.sqrt_info = 1.0; blocks.push_back(b); }
            for (const Feature &f : cf) { ResidualBlock b; b.type = 'E'; std::memcpy(b.point, f.point, sizeof(b.point)); std::memcpy(b.coeffs, f.coeffs, sizeof(b.coeffs)); b.sqrt_info = 1.0; blocks.push_back(b); }
            ceres_like_solve(blocks, x, V, tp.huber_delta, tp.max_lm_iterations, st.solve);
            st.solved = true;
        }
        std::memcpy(st.pose_after, x, sizeof(x));
        stats.push_back(st);
    }
    std::memcpy(pose_out, x, sizeof(x));
}

}  // namespace orc
