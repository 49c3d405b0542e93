// AlivePool (m-loam_amd/csrc/alive_pool.hpp) against the container it stands in for: a std::vector<size_t> holding 0..n-1 that only loses
// elements (all_feature_idx, lidar_mapper.h:350, 531-553; cloud_scan[row], image_segmenter.hpp:366-376). Position look-up, membership and
// erase must answer exactly as the vector does, for every size around the tree's group boundaries (64 slots per word, 16 children per node).
#include "alive_pool.hpp"
#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

int main()
{
    const size_t sizes[] = {1, 2, 63, 64, 65, 127, 128, 1023, 1024, 1025, 1600, 4097, 11732, 16384, 16385, 70000};
    for (size_t n : sizes) {
        for (int mode = 0; mode < 3; ++mode) {              // erase everything drawn / two of three / only from the front
            mlh::AlivePool pool(n);
            std::vector<size_t> v(n);
            for (size_t i = 0; i < n; ++i) v[i] = i;
            std::mt19937 rng(unsigned(n) * 3u + unsigned(mode));
            const size_t steps = std::min<size_t>(n < 5000 ? 2 * n : 4000, 20000);
            for (size_t it = 0; it < steps && !pool.empty(); ++it) {
                if (pool.size() != v.size()) { std::printf("size mismatch n %zu\n", n); return 1; }
                size_t j = mode == 2 ? 0 : std::uniform_int_distribution<size_t>(0, v.size() - 1)(rng);
                if (it % 97 == 0) j = v.size() - 1;         // the last survivor too
                const size_t got = pool.at(j);
                if (got != v[j]) { std::printf("at mismatch n %zu mode %d it %zu j %zu got %zu want %zu\n", n, mode, it, j, got, v[j]); return 1; }
                if (!pool.contains(got)) { std::printf("contains(alive) false n %zu\n", n); return 1; }
                if (mode != 1 || it % 3 != 2) {
                    pool.erase_index(got);
                    v.erase(v.begin() + long(j));
                    if (pool.contains(got)) { std::printf("contains(erased) true n %zu\n", n); return 1; }
                }
            }
            // what is left, in order
            for (size_t j = 0; j < v.size(); j += std::max<size_t>(1, v.size() / 257))
                if (pool.at(j) != v[j]) { std::printf("sweep mismatch n %zu mode %d j %zu\n", n, mode, j); return 1; }
        }
    }
    std::puts("ok");
    return 0;
}
