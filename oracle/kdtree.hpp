// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (PCL/FLANN absent).
// Exact k-nearest-neighbour search standing in for pcl::KdTreeFLANN<PointT>::nearestKSearch
// (PCL 1.8.0 -> FLANN KDTreeSingleIndex, L2_Simple<float>, eps = 0, sorted results), called at
// feature_extract.hpp:406, 570, 666, 813 and built at lidar_mapper_keyframe.cpp:433-434.
// Semantics restated: squared distance accumulated in f32 as ((dx*dx + dy*dy) + dz*dz)
// (flann/algorithms/dist.h L2_Simple: result += diff*diff over the dimensions in order);
// the k results are the exact k smallest, ascending. Distance ties are implementation-defined in
// FLANN; here they are broken by the smaller point index (the HIP path uses the same rule).
#pragma once
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace orc {

class KdTree {
public:
    void build(const float *pts, size_t stride_floats, int n)
    {
        n_ = n;
        xyz_.resize(size_t(n) * 3);
        idx_.resize(n);
        for (int i = 0; i < n; ++i) idx_[i] = i;
        src_ = pts; stride_ = stride_floats;
        nodes_.clear();
        nodes_.reserve(size_t(n) / 4 + 16);
        if (n > 0) build_rec(0, n);
        for (int i = 0; i < n; ++i) {
            const float *p = pts + size_t(idx_[i]) * stride_floats;
            xyz_[size_t(i) * 3 + 0] = p[0]; xyz_[size_t(i) * 3 + 1] = p[1]; xyz_[size_t(i) * 3 + 2] = p[2];
        }
        src_ = nullptr;
    }
    int size() const { return n_; }

    // returns number found (min(k, n)); out arrays sorted ascending by (d2, index)
    int knn(const float q[3], int k, int *out_idx, float *out_d2) const
    {
        int found = 0;
        if (n_ == 0) return 0;
        search(0, q, k, out_idx, out_d2, found);
        return found;
    }

private:
    static constexpr int LEAF = 12;
    struct Node {
        float bmin[3], bmax[3];
        int lo, hi;       // point range
        int left, right;  // children (-1 for leaf)
    };
    std::vector<Node> nodes_;
    std::vector<float> xyz_;
    std::vector<int> idx_;
    const float *src_ = nullptr;
    size_t stride_ = 0;
    int n_ = 0;

    int build_rec(int lo, int hi)
    {
        Node nd;
        nd.lo = lo; nd.hi = hi; nd.left = nd.right = -1;
        for (int d = 0; d < 3; ++d) { nd.bmin[d] = INFINITY; nd.bmax[d] = -INFINITY; }
        for (int i = lo; i < hi; ++i) {
            const float *p = src_ + size_t(idx_[i]) * stride_;
            for (int d = 0; d < 3; ++d) { nd.bmin[d] = std::min(nd.bmin[d], p[d]); nd.bmax[d] = std::max(nd.bmax[d], p[d]); }
        }
        int me = int(nodes_.size());
        nodes_.push_back(nd);
        if (hi - lo > LEAF) {
            int dim = 0;
            float ext = nd.bmax[0] - nd.bmin[0];
            for (int d = 1; d < 3; ++d) if (nd.bmax[d] - nd.bmin[d] > ext) { ext = nd.bmax[d] - nd.bmin[d]; dim = d; }
            int mid = (lo + hi) / 2;
            const float *src = src_; size_t st = stride_;
            std::nth_element(idx_.begin() + lo, idx_.begin() + mid, idx_.begin() + hi,
                             [src, st, dim](int a, int b) {
                                 float va = src[size_t(a) * st + dim], vb = src[size_t(b) * st + dim];
                                 return va < vb || (va == vb && a < b);
                             });
            int l = build_rec(lo, mid);
            int r = build_rec(mid, hi);
            nodes_[me].left = l; nodes_[me].right = r;
        }
        return me;
    }

    static inline double box_lb2(const Node &nd, const float q[3])
    {
        double s = 0.0;
        for (int d = 0; d < 3; ++d) {
            double e = 0.0;
            if (q[d] < nd.bmin[d]) e = double(nd.bmin[d]) - double(q[d]);
            else if (q[d] > nd.bmax[d]) e = double(q[d]) - double(nd.bmax[d]);
            s += e * e;
        }
        return s;
    }

    void search(int ni, const float q[3], int k, int *oi, float *od, int &found) const
    {
        const Node &nd = nodes_[ni];
        if (nd.left < 0) {
            for (int i = nd.lo; i < nd.hi; ++i) {
                const float *p = &xyz_[size_t(i) * 3];
                float dx = p[0] - q[0], dy = p[1] - q[1], dz = p[2] - q[2];
                float d2 = 0.f;
                d2 += dx * dx; d2 += dy * dy; d2 += dz * dz;
                int id = idx_[i];
                if (found == k) {
                    if (!(d2 < od[k - 1] || (d2 == od[k - 1] && id < oi[k - 1]))) continue;
                }
                int pos = (found < k) ? found : k - 1;
                while (pos > 0 && (d2 < od[pos - 1] || (d2 == od[pos - 1] && id < oi[pos - 1]))) {
                    od[pos] = od[pos - 1]; oi[pos] = oi[pos - 1]; --pos;
                }
                od[pos] = d2; oi[pos] = id;
                if (found < k) ++found;
            }
            return;
        }
        double lbl = box_lb2(nodes_[nd.left], q), lbr = box_lb2(nodes_[nd.right], q);
        int first = nd.left, second = nd.right;
        double lb1 = lbl, lb2 = lbr;
        if (lbr < lbl) { std::swap(first, second); std::swap(lb1, lb2); }
        // conservative pruning: the f32-rounded distance of a point may undershoot the exact bound
        if (found < k || lb1 * (1.0 - 1e-5) <= double(od[k - 1])) search(first, q, k, oi, od, found);
        if (found < k || lb2 * (1.0 - 1e-5) <= double(od[k - 1])) search(second, q, k, oi, od, found);
    }
};

}  // namespace orc
