// Good-feature selection (ActiveFeatureSelection::goodFeatureMatching, estimator/src/lidarMapper/lidar_mapper.h:229-573).
// The GPU matches every feature and evaluates every 1x6 Jacobian in one pass; this file holds the host-side selection loops
// that consume those rows. The loops are sequential by construction (each greedy pick changes the information matrix the
// next score is computed against), which is why the reference gives them a 20 ms wall-clock budget (lidar_mapper.h:82).
#include "ctx.hpp"
#include <algorithm>
#include <cmath>
#include <numeric>
#include <queue>
#include <random>

namespace mlh {

namespace {

struct Rows {               // per-feature results of the GPU pass
    std::vector<Corr> corr;
    std::vector<double> J;  // m x 6
    std::vector<float4> pts;
    bool matched(size_t i) const { return corr[i].valid != 0; }
    const double *jaco(size_t i) const { return &J[i * 6]; }
};

inline void rank1_update(double H[36], const double *j)
{
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) H[r * 6 + c] += j[r] * j[c];
}

// common::logDet(M, use_cholesky = true)  (mloam_common/libs/include/common/algos/math.hpp:173-187)
double logdet_cholesky6(const double A[36])
{
    double L[36] = {0};
    double ld = 0.0;
    for (int j = 0; j < 6; ++j) {
        double s = A[j * 6 + j];
        for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
        if (!(s > 0.0)) return NAN;
        const double ljj = std::sqrt(s);
        L[j * 6 + j] = ljj;
        ld += std::log(ljj);
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i * 6 + j];
            for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = t / ljj;
        }
    }
    return 2.0 * ld;
}

inline size_t draw(std::mt19937 &rng, size_t lo, size_t hi)
{
    std::uniform_int_distribution<size_t> d(lo, hi);   // RandomGeneratorInt<size_t>::geneRandUniform re-creates the distribution per draw
    return d(rng);
}

// The reference keeps the not-yet-consumed feature slots in a std::vector it erases from (all_feature_idx, lidar_mapper.h:350,
// 531-553): position j of that vector is always the (j+1)-th surviving ORIGINAL index, because it starts as 0..M-1 and only ever
// loses elements. A Fenwick tree over "alive" flags answers the same three questions -- element at position j, position of an
// element (the reference's std::find), erase -- in O(log M) instead of O(M) per operation, with identical results.
class AlivePool {
public:
    explicit AlivePool(size_t n) : n_(n), alive_(n), t_(n + 1, 0), alive_flag_(n, 1)
    {
        for (size_t i = 1; i <= n; ++i) { t_[i] += 1; const size_t j = i + (i & (~i + 1)); if (j <= n) t_[j] += t_[i]; }
        log_ = 1;
        while ((size_t(1) << log_) <= n) ++log_;
    }
    size_t size() const { return alive_; }
    bool empty() const { return alive_ == 0; }
    // original index of the element at position j (0-based) among the survivors
    size_t at(size_t j) const
    {
        size_t pos = 0, k = j + 1;
        for (int b = log_; b >= 0; --b) {
            const size_t nxt = pos + (size_t(1) << b);
            if (nxt <= n_ && t_[nxt] < int(k)) { pos = nxt; k -= size_t(t_[nxt]); }
        }
        return pos;   // 0-based original index
    }
    bool contains(size_t idx) const { return alive_flag_[idx] != 0; }
    void erase_index(size_t idx)
    {
        alive_flag_[idx] = 0;
        --alive_;
        for (size_t i = idx + 1; i <= n_; i += i & (~i + 1)) t_[i] -= 1;
    }
private:
    size_t n_, alive_;
    std::vector<int> t_;
    std::vector<char> alive_flag_;
    int log_;
};

struct Scored {   // FeatureWithScore (parameters.h:177-191): max-heap on the logdet score
    size_t idx;
    double score;
    bool operator<(const Scored &o) const { return score < o.score; }
};

void select_wo_gf(const Rows &R, std::vector<size_t> &sel, double H[36])
{
    for (size_t i = 0; i < R.corr.size(); ++i)
        if (R.matched(i)) { rank1_update(H, R.jaco(i)); sel.push_back(i); }
}

void select_rnd(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    AlivePool pool(R.corr.size());
    while (sel.size() < n_use && !pool.empty()) {
        const size_t j = draw(rng, 0, pool.size() - 1);
        const size_t q = pool.at(j);
        if (R.matched(q)) { rank1_update(H, R.jaco(q)); sel.push_back(q); }
        pool.erase_index(q);
    }
}

void select_fps(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    const size_t n = R.corr.size();
    if (n == 0) return;
    std::vector<char> visited(n, 0);
    size_t cur = draw(rng, 0, n - 1);
    visited[cur] = 1;
    size_t n_visited = 1;
    // the starting point is kept when matched, but its Jacobian is not accumulated (lidar_mapper.h:356-386)
    if (R.matched(cur) && n_use > 0) sel.push_back(cur);
    std::vector<float> dist(n, 1e5f);
    while (sel.size() < n_use && n_visited < n) {   // the reference leaves through its wall-clock cut-off once all are visited
        float best_d = -1.f;
        size_t best_j = 1;
        const float4 po = R.pts[cur];
        for (size_t j = 0; j < n; ++j) {
            if (visited[j]) continue;
            const float4 pn = R.pts[j];
            const float ddx = po.x - pn.x, ddy = po.y - pn.y, ddz = po.z - pn.z;
            const float d = std::sqrt(ddx * ddx + ddy * ddy + ddz * ddz);
            const float d2 = std::min(d, dist[j]);
            dist[j] = d2;
            if (d2 > best_d) { best_j = j; best_d = d2; }
        }
        cur = best_j;
        visited[cur] = 1;
        ++n_visited;
        if (R.matched(cur)) { rank1_update(H, R.jaco(cur)); sel.push_back(cur); }
    }
}

// stochastic-greedy logdet maximisation (lidar_mapper.h:458-563): draw a random subset of size M / M_use, score every
// member by logdet(H + j^T j), keep the best, repeat. Unmatched draws are dropped from the pool.
void select_greedy(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    const size_t n_all = R.corr.size();
    AlivePool pool(n_all);
    std::vector<int> stamp(n_all, -1);        // feature_visited, kept per ORIGINAL index (the reference erases it in lockstep with the pool)
    const size_t max_retry = 20;              // MAX_RANDOM_QUEUE_TIME
    size_t retries = 0;
    while (sel.size() < n_use && !pool.empty()) {
        const size_t subset = static_cast<size_t>(1.0 * n_all / n_use);
        std::priority_queue<Scored> heap;
        bool lost = false;
        while (!pool.empty()) {
            retries = 0;
            size_t q = 0;
            while (retries < max_retry) {
                const size_t j = draw(rng, 0, pool.size() - 1);
                q = pool.at(j);
                if (stamp[q] < int(sel.size())) { stamp[q] = int(sel.size()); break; }
                ++retries;
            }
            if (retries >= max_retry) break;
            if (!R.matched(q)) {              // "not found constraints or outlier constraints": forget the slot
                pool.erase_index(q);
                continue;
            }
            double Ht[36];
            std::copy(H, H + 36, Ht);
            rank1_update(Ht, R.jaco(q));
            heap.push(Scored{q, logdet_cholesky6(Ht)});
            if (heap.size() >= subset) {
                const Scored top = heap.top();
                if (!pool.contains(top.idx)) { lost = true; break; }     // the reference's std::find miss
                rank1_update(H, R.jaco(top.idx));
                pool.erase_index(top.idx);
                sel.push_back(top.idx);
                break;
            }
        }
        if (retries >= max_retry || lost) break;
    }
}

}  // namespace

// One goodFeatureMatching call. The device-side solver state must already hold the pose (SolverState::x).
int good_feature_select(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, float min_match_sq_dis,
                        float min_plane_dis, std::vector<int32_t> &sel_out, double H[36], uint8_t *matched_out)
{
    FeatSet &f = ctx->feat[kind];
    MatchArgs a;
    a.kind_mask = 1 << kind;
    a.flags = MLH_FLAG_WITH_UA | MLH_FLAG_NO_LOSS;   // extractCov(point) weight, rows not loss-corrected (lidar_mapper.h:162-164)
    a.min_match_sq_dis = min_match_sq_dis; a.min_plane_dis = min_plane_dis;
    a.huber_delta = 0.0; a.dense = true; a.pose_sel = 0;
    int rc = match_launch(ctx, a);
    if (rc) return rc;
    Rows R;
    const size_t m = size_t(f.m);
    R.corr.resize(m); R.J.resize(m * 6);
    MLH_HIP(ctx, hipMemcpyAsync(R.corr.data(), f.corr.p, sizeof(Corr) * m, hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(R.J.data(), f.J.p, sizeof(double) * 6 * m, hipMemcpyDeviceToHost, ctx->stream));
    if (method == MLH_GF_FPS) {
        R.pts.resize(m);
        MLH_HIP(ctx, hipMemcpyAsync(R.pts.data(), f.pts.p, sizeof(float4) * m, hipMemcpyDeviceToHost, ctx->stream));
    }
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    if (matched_out) for (size_t i = 0; i < m; ++i) matched_out[i] = R.matched(i) ? 1 : 0;

    const size_t n_use = static_cast<size_t>(m * ratio);   // num_use_features (lidar_mapper.h:247)
    std::vector<size_t> sel;
    sel.reserve(method == MLH_GF_WO ? m : n_use);
    switch (method) {
        case MLH_GF_WO: select_wo_gf(R, sel, H); break;
        case MLH_GF_RND: select_rnd(R, n_use, rng, sel, H); break;
        case MLH_GF_FPS: select_fps(R, n_use, rng, sel, H); break;
        case MLH_GF_GD_FIX:
        case MLH_GF_GD_FLOAT: select_greedy(R, n_use, rng, sel, H); break;
        default: return fail(ctx, MLH_ERR_INVALID, "unknown gf_method");
    }
    // keep only the selected correspondences valid on the device
    if (method != MLH_GF_WO) {
        for (auto &c : R.corr) c.valid = 0;
        for (size_t i : sel) R.corr[i].valid = 1;
        MLH_HIP(ctx, hipMemcpyAsync(f.corr.p, R.corr.data(), sizeof(Corr) * m, hipMemcpyHostToDevice, ctx->stream));
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    sel_out.assign(sel.begin(), sel.end());
    return MLH_OK;
}

}  // namespace mlh
