// Host logic check (no GPU): mlh::std_sort_mt must leave every sequence exactly as std::sort leaves it -- equal keys included, which is the
// whole point (m-loam_amd/csrc/std_sort_mt.hpp). Compiled and run by tests/test_abi.py::test_std_sort_mt_equals_std_sort.
#include "std_sort_mt.hpp"
#include <cstdio>
#include <cstring>
#include <random>

struct IdxPt {   // cloud_point_index_idx (pcl/filters/voxel_grid.h): the comparator sees idx only
    unsigned idx, cloud_point_index;
    bool operator<(const IdxPt &o) const { return idx < o.idx; }
};

int main()
{
    std::mt19937 g(1);
    int bad = 0, cases = 0;
    for (int n : {0, 1, 2, 15, 16, 17, 33, 100, 1025, 5000, 30000, 40000, 200000})
        for (int nv : {1, 2, 7, n / 4 + 1, n + 1})
            for (int levels : {0, 1, 2, 3})
                for (int pattern = 0; pattern < 4; ++pattern) {
                    std::vector<IdxPt> a(static_cast<size_t>(n));
                    for (int i = 0; i < n; ++i) {
                        unsigned k = 0;
                        switch (pattern) {
                            case 0: k = unsigned(g() % unsigned(nv)); break;                 // random voxel per point
                            case 1: k = unsigned(i % nv); break;                             // interleaved
                            case 2: k = unsigned((n - i) / (n / nv + 1)); break;             // descending runs
                            default: k = unsigned(i < n / 2 ? i : n - i) % unsigned(nv);     // organ pipe
                        }
                        a[size_t(i)] = IdxPt{k, unsigned(i)};
                    }
                    std::vector<IdxPt> b = a, c = a;
                    std::sort(b.begin(), b.end(), std::less<IdxPt>());
                    mlh::std_sort_mt(c.data(), c.data() + c.size(), levels);
                    ++cases;
                    if (n && std::memcmp(b.data(), c.data(), sizeof(IdxPt) * size_t(n)) != 0) {
                        ++bad;
                        std::printf("MISMATCH n %d voxels %d fork levels %d pattern %d\n", n, nv, levels, pattern);
                    }
                }
    std::printf("%d cases, %d mismatches\n", cases, bad);
    return bad ? 1 : 0;
}
