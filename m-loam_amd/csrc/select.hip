// Good-feature selection (ActiveFeatureSelection::goodFeatureMatching, estimator/src/lidarMapper/lidar_mapper.h:229-573).
// The GPU matches every feature and evaluates every 1x6 Jacobian in one pass; this file holds the host-side selection loops
// that consume those rows. The loops are sequential by construction (each greedy pick changes the information matrix the
// next score is computed against), which is why the reference gives them a 20 ms wall-clock budget (lidar_mapper.h:82).
#include "ctx.hpp"
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <numeric>
#include <queue>
#include <random>

namespace mlh {

namespace {

struct Rows {               // per-feature results of the GPU pass, in the context's pinned staging block
    Corr *corr = nullptr;
    const double *J = nullptr;  // m x 6
    const float4 *pts = nullptr;
    size_t m = 0;
    size_t size() const { return m; }
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
// loses elements. The same three questions -- element at position j, position of an element (the reference's std::find), erase --
// are answered here from one "alive" bit per slot plus a Fenwick tree over the 64-bit words' populations: a branch-free descent over
// log2(M/64) levels and a six-step rank search inside the word, instead of O(M) per operation, with identical results. (The draw loops
// are one dependent chain per draw -- draw, look up, erase -- so the lookup's latency is the loop's speed: a Fenwick tree over single
// slots with a data-dependent branch per level ran at ~170 ns per draw, this runs at ~30.)
class AlivePool {
public:
    explicit AlivePool(size_t n) : alive_(n), nw_((n + 63) / 64)
    {
        log_ = 0;
        while ((size_t(2) << log_) <= nw_) ++log_;                  // largest power of two <= nw_ is 1 << log_
        bits_.assign(nw_ + 1, 0);
        for (size_t w = 0; w < nw_; ++w) bits_[w] = (w * 64 + 64 <= n) ? ~uint64_t(0) : ((uint64_t(1) << (n - w * 64)) - 1);
        t_.assign((size_t(2) << log_) + 1, kNever);                  // slots past nw_ are never taken by the descent
        for (size_t i = 1; i <= nw_; ++i) t_[i] = 0;
        for (size_t i = 1; i <= nw_; ++i) {
            t_[i] += int32_t(popcount64(bits_[i - 1]));
            const size_t j = i + (i & (~i + 1));
            if (j <= nw_) t_[j] += t_[i];
        }
    }
    size_t size() const { return alive_; }
    bool empty() const { return alive_ == 0; }
    // original index of the element at position j (0-based) among the survivors
    size_t at(size_t j) const
    {
        size_t pos = 0;
        int32_t k = int32_t(j);
        for (int b = log_; b >= 0; --b) {
            const int32_t v = t_[pos + (size_t(1) << b)];
            const int32_t take = -int32_t(v <= k);                  // all-ones / zero: the comparison's outcome is a coin flip, keep it out of the branch predictor
            pos += (size_t(1) << b) & size_t(int64_t(take));
            k -= v & take;
        }
        // rank search inside the word: per-byte populations (SWAR), their running sums by one multiply, the first byte whose running sum
        // exceeds k by a carry-free byte-wise compare, the bit inside that byte from a 2 KB table
        const uint64_t w = bits_[pos];
        uint64_t c = w - ((w >> 1) & 0x5555555555555555ull);
        c = (c & 0x3333333333333333ull) + ((c >> 2) & 0x3333333333333333ull);
        c = (c + (c >> 4)) & 0x0f0f0f0f0f0f0f0full;
        const uint64_t run = c * 0x0101010101010101ull;               // byte i: population of bytes 0..i (<= 64)
        const uint64_t over = ((run | 0x8080808080808080ull) - (uint64_t(k) + 1) * 0x0101010101010101ull) & 0x8080808080808080ull;
        const unsigned byte = unsigned(__builtin_ctzll(over)) >> 3;   // first byte with run > k (exists: k < the word's population)
        const unsigned before = unsigned(((run << 8) >> (8 * byte)) & 0xff);
        const size_t bit = 8 * byte + kSelect8.at[(w >> (8 * byte)) & 0xff][unsigned(k) - before];
        return pos * 64 + bit;
    }
    bool contains(size_t idx) const { return (bits_[idx >> 6] >> (idx & 63)) & 1; }
    void erase_index(size_t idx)
    {
        bits_[idx >> 6] &= ~(uint64_t(1) << (idx & 63));
        --alive_;
        for (size_t i = (idx >> 6) + 1; i <= nw_; i += i & (~i + 1)) t_[i] -= 1;
    }
private:
    struct Select8 {                                               // at[v][r] = position of the r-th (0-based) set bit of the byte v
        uint8_t at[256][8];
        Select8()
        {
            for (int v = 0; v < 256; ++v) {
                int r = 0;
                for (int b = 0; b < 8; ++b) if (v >> b & 1) at[v][r++] = uint8_t(b);
                for (; r < 8; ++r) at[v][r] = 0;
            }
        }
    };
    static const Select8 kSelect8;
    static constexpr int32_t kNever = 0x3fffffff;
    static inline unsigned popcount64(uint64_t x)                  // SWAR: the baseline x86-64 target has no popcnt instruction
    {
        x = x - ((x >> 1) & 0x5555555555555555ull);
        x = (x & 0x3333333333333333ull) + ((x >> 2) & 0x3333333333333333ull);
        x = (x + (x >> 4)) & 0x0f0f0f0f0f0f0f0full;
        return unsigned((x * 0x0101010101010101ull) >> 56);
    }
    size_t alive_, nw_;
    std::vector<uint64_t> bits_;
    std::vector<int32_t> t_;
    int log_;
};

const AlivePool::Select8 AlivePool::kSelect8;

struct Scored {   // FeatureWithScore (parameters.h:177-191): max-heap on the logdet score
    size_t idx;
    double score;
    bool operator<(const Scored &o) const { return score < o.score; }
};

void select_wo_gf(const Rows &R, std::vector<size_t> &sel, double H[36])
{
    for (size_t i = 0; i < R.size(); ++i)
        if (R.matched(i)) { rank1_update(H, R.jaco(i)); sel.push_back(i); }
}

void select_rnd(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    AlivePool pool(R.size());
    while (sel.size() < n_use && !pool.empty()) {
        const size_t j = draw(rng, 0, pool.size() - 1);
        const size_t q = pool.at(j);
        if (R.matched(q)) { rank1_update(H, R.jaco(q)); sel.push_back(q); }
        pool.erase_index(q);
    }
}

void select_fps(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    const size_t n = R.size();
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

// 6x6 inverse of a symmetric positive definite matrix through its Cholesky factor (once per selection; rank-1 updated afterwards)
bool spd_inverse6(const double A[36], double Ainv[36])
{
    double L[36] = {0};
    for (int j = 0; j < 6; ++j) {
        double s = A[j * 6 + j];
        for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
        if (!(s > 0.0)) return false;
        const double ljj = std::sqrt(s);
        L[j * 6 + j] = ljj;
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i * 6 + j];
            for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = t / ljj;
        }
    }
    double Li[36] = {0};                               // L^-1 (lower triangular)
    for (int c = 0; c < 6; ++c) {
        Li[c * 6 + c] = 1.0 / L[c * 6 + c];
        for (int r = c + 1; r < 6; ++r) {
            double t = 0.0;
            for (int k = c; k < r; ++k) t -= L[r * 6 + k] * Li[k * 6 + c];
            Li[r * 6 + c] = t / L[r * 6 + r];
        }
    }
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double t = 0.0; for (int k = std::max(r, c); k < 6; ++k) t += Li[k * 6 + r] * Li[k * 6 + c]; Ainv[r * 6 + c] = t; }
    return true;
}

// stochastic-greedy logdet maximisation (lidar_mapper.h:458-563): draw a random subset of size M / M_use, score every
// member by logdet(H + j^T j), keep the best, repeat. Unmatched draws are dropped from the pool.
// Scoring: logdet(H + j^T j) = logdet(H) + log(1 + j H^-1 j^T) (matrix determinant lemma), so inside one subset -- all members are scored
// against the same H -- the best member is the one with the largest q = j H^-1 j^T: 42 multiply-adds against a maintained inverse
// (Sherman-Morrison per pick) instead of a Cholesky factorisation and six logarithms per candidate. The reference compares the logdet
// VALUES (magnitude <~ 100, so resolved to ~1e-13 by the Cholesky sum): whenever the two best members' q are closer than the error the
// maintained inverse can carry (64 eps cond(H), at least 1e-10, relative to 1 + q) the whole subset is re-scored with the reference's arithmetic (logdet_cholesky6) and pushed through the same heap -- selections stay
// identical to the literal loop's (MLH_SELECT_EXACT=1 in the environment runs the literal scoring; tests compare the two).
void select_greedy(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel, double H[36])
{
    const size_t n_all = R.size();
    AlivePool pool(n_all);
    std::vector<int> stamp(n_all, -1);        // feature_visited, kept per ORIGINAL index (the reference erases it in lockstep with the pool)
    const size_t max_retry = 20;              // MAX_RANDOM_QUEUE_TIME
    size_t retries = 0;
    double Hinv[36];
    bool have_inv = std::getenv("MLH_SELECT_EXACT") == nullptr && spd_inverse6(H, Hinv);
    double replay_tol = 1e-10;
    if (have_inv) {
        double nh = 0.0, ni = 0.0;
        for (int k = 0; k < 36; ++k) { nh += H[k] * H[k]; ni += Hinv[k] * Hinv[k]; }
        replay_tol = std::max(1e-10, 64.0 * 2.220446049250313e-16 * std::sqrt(nh) * std::sqrt(ni));
    }
    auto exact_score = [&](size_t q) { double Ht[36]; std::copy(H, H + 36, Ht); rank1_update(Ht, R.jaco(q)); return logdet_cholesky6(Ht); };
    auto quad = [&](const double *j, double *Hj) {
        double q = 0.0;
        for (int r = 0; r < 6; ++r) { double t = 0.0; for (int c = 0; c < 6; ++c) t += Hinv[r * 6 + c] * j[c]; Hj[r] = t; q += j[r] * t; }
        return q;
    };
    struct Cand { size_t idx; double q; };
    std::vector<Cand> subset_c;
    while (sel.size() < n_use && !pool.empty()) {
        const size_t subset = static_cast<size_t>(1.0 * n_all / n_use);
        subset_c.clear();
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
            double Hj[6];
            subset_c.push_back(Cand{q, have_inv ? quad(R.jaco(q), Hj) : exact_score(q)});
            if (subset_c.size() >= subset) {
                // the heap's top: the largest score; among equal scores std::priority_queue keeps ... whichever its sift leaves on top, so
                // near-ties (and exact ties) are settled by the reference's own numbers, pushed through the reference's own container
                size_t best = 0, second = size_t(-1);
                for (size_t k = 1; k < subset_c.size(); ++k) {
                    if (subset_c[k].q > subset_c[best].q) { second = best; best = k; }
                    else if (second == size_t(-1) || subset_c[k].q > subset_c[second].q) second = k;
                }
                size_t top_idx = subset_c[best].idx;
                if (have_inv && second != size_t(-1) && !(subset_c[best].q - subset_c[second].q > replay_tol * (1.0 + subset_c[best].q))) {
                    std::priority_queue<Scored> heap;                      // too close to call on q: replay the subset literally
                    for (const Cand &c : subset_c) heap.push(Scored{c.idx, exact_score(c.idx)});
                    top_idx = heap.top().idx;
                }
                if (!pool.contains(top_idx)) { lost = true; break; }     // the reference's std::find miss
                const double *jt = R.jaco(top_idx);
                if (have_inv) {                                            // Sherman-Morrison: (H + j^T j)^-1 = H^-1 - (H^-1 j^T)(j H^-1) / (1 + j H^-1 j^T)
                    double Hj2[6];
                    const double qq = quad(jt, Hj2);
                    const double inv = 1.0 / (1.0 + qq);
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) Hinv[r * 6 + c] -= Hj2[r] * Hj2[c] * inv;
                }
                rank1_update(H, jt);
                pool.erase_index(top_idx);
                sel.push_back(top_idx);
                if (have_inv) {
                    // How far q = j H^-1 j^T can be off: ~ eps * cond(H) per use of the maintained inverse. The reference starts from
                    // H = 1e-6 I, so the first picks see cond ~ 1e6..1e8 -- there the inverse is rebuilt from H after every pick instead of
                    // rank-1 updated, and the "too close to call" band is widened to that error, so that a near-tie inside it is always settled
                    // by the reference's own logdet arithmetic. Frobenius norms: cond_2 <= ||H||_F ||H^-1||_F.
                    double nh = 0.0, ni = 0.0;
                    for (int k = 0; k < 36; ++k) { nh += H[k] * H[k]; ni += Hinv[k] * Hinv[k]; }
                    const double kappa = std::sqrt(nh) * std::sqrt(ni);
                    if (kappa > 1e5 || (sel.size() & 255) == 0) {
                        have_inv = spd_inverse6(H, Hinv);       // refresh: keeps the update's rounding from accumulating
                        if (have_inv) { ni = 0.0; for (int k = 0; k < 36; ++k) ni += Hinv[k] * Hinv[k]; }
                    }
                    replay_tol = std::max(1e-10, 64.0 * 2.220446049250313e-16 * std::sqrt(nh) * std::sqrt(ni));
                }
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
    // pinned staging (grow-only, owned by the context): [Corr m][J 6m][pts m]
    const size_t off_j = sizeof(Corr) * m, off_p = off_j + sizeof(double) * 6 * m, need = off_p + sizeof(float4) * m;
    if (need > ctx->select_host_cap) {
        if (ctx->select_host) (void)hipHostFree(ctx->select_host);
        ctx->select_host = nullptr; ctx->select_host_cap = 0;
        MLH_HIP(ctx, hipHostMalloc(&ctx->select_host, need + need / 4, hipHostMallocDefault));
        ctx->select_host_cap = need + need / 4;
    }
    char *hb = static_cast<char *>(ctx->select_host);
    R.corr = reinterpret_cast<Corr *>(hb); R.J = reinterpret_cast<const double *>(hb + off_j); R.pts = reinterpret_cast<const float4 *>(hb + off_p);
    R.m = m;
    MLH_HIP(ctx, hipMemcpyAsync(hb, f.corr.p, sizeof(Corr) * m, hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(hb + off_j, f.J.p, sizeof(double) * 6 * m, hipMemcpyDeviceToHost, ctx->stream));
    if (method == MLH_GF_FPS) MLH_HIP(ctx, hipMemcpyAsync(hb + off_p, f.pts.p, sizeof(float4) * m, hipMemcpyDeviceToHost, ctx->stream));
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
        for (size_t i = 0; i < m; ++i) R.corr[i].valid = 0;
        for (size_t i : sel) R.corr[i].valid = 1;
        MLH_HIP(ctx, hipMemcpyAsync(f.corr.p, R.corr, sizeof(Corr) * m, hipMemcpyHostToDevice, ctx->stream));
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    sel_out.assign(sel.begin(), sel.end());
    return MLH_OK;
}

}  // namespace mlh
