// libstdc++'s std::sort on several threads, with the result std::sort itself gives -- element for element, equal keys included.
//
// Why this exists: the reference's voxel filters order a cloud's points with std::sort and a comparator that sees the voxel index only
// (voxel_grid_covariance_mloam_impl.hpp:215-236), so the order of a voxel's members is whatever libstdc++'s introsort leaves, and the
// reference's results depend on it (voxelgrid.hip: members_in_std_sort_order). Reproducing that order means running that algorithm on
// that sequence; it does not mean running it on one thread. std::sort is
//     __introsort_loop(first, last, 2 * floor(log2(n)));  __final_insertion_sort(first, last);
// and the loop is a quicksort whose two sides, once a partition is done, never look at each other again: the right side is the
// recursive call, the left side the loop's next trip. Handing the recursive call of the first few partitions to another thread changes
// when the two sides are sorted, not what is done to them -- same pivots (median of three at the same positions), same partition
// (libstdc++'s own __unguarded_partition_pivot is called, not restated), same depth budget and heap-sort fallback, and the same single
// final insertion pass over the whole range once every thread has joined. Host-only header; no HIP in here.
#pragma once
#include <algorithm>
#include <cstddef>
#include <system_error>
#include <thread>
#include <vector>

namespace mlh {

#if defined(__GLIBCXX__)
namespace detail {
template <class It>
void introsort_forked(It first, It last, long depth_limit, int fork_levels)
{
    auto comp = __gnu_cxx::__ops::__iter_less_iter();
    std::vector<std::thread> forks;
    while (last - first > 16) {                                   // _S_threshold
        if (depth_limit == 0) { std::__partial_sort(first, last, last, comp); break; }
        --depth_limit;
        It cut = std::__unguarded_partition_pivot(first, last, comp);
        if (fork_levels > 0 && last - cut > 1024) {
            --fork_levels;
            const long d = depth_limit;
            const int f = fork_levels;
            bool forked = false;
            try { forks.emplace_back([cut, last, d, f] { introsort_forked(cut, last, d, f); }); forked = true; }
            catch (const std::system_error &) {}                  // no thread to be had: this one does the work, same result
            if (!forked) std::__introsort_loop(cut, last, depth_limit, comp);
        } else {
            std::__introsort_loop(cut, last, depth_limit, comp);
        }
        last = cut;
    }
    for (std::thread &t : forks) t.join();
}
}  // namespace detail

// fork_levels = k: up to 2^k threads
template <class It>
void std_sort_mt(It first, It last, int fork_levels)
{
    if (first == last) return;
    long lg = 0;
    for (std::size_t n = std::size_t(last - first); n > 1; n >>= 1) ++lg;   // std::__lg
    detail::introsort_forked(first, last, lg * 2, fork_levels);
    std::__final_insertion_sort(first, last, __gnu_cxx::__ops::__iter_less_iter());
}
#else
template <class It>
void std_sort_mt(It first, It last, int) { std::sort(first, last); }          // another standard library: its own sort, one thread
#endif

}  // namespace mlh
