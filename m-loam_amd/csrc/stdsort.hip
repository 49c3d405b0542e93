// libstdc++'s std::sort, on the device, with the result std::sort itself gives -- element for element, equal keys included.
//
// Why: the reference's voxel filters order a cloud's points with std::sort and a comparator that sees the voxel index only
// (voxel_grid_covariance_mloam_impl.hpp:215-236; pcl/filters/voxel_grid.h: cloud_point_index_idx::operator<), so the order of a voxel's
// members -- and with it the LiDAR id a mixed voxel keeps, the first-heaviest member on equal weights and the association of every f32
// sum -- is whatever libstdc++'s introsort leaves. Reproducing the reference's numbers means reproducing that permutation. A host pass
// that runs std::sort on the slots costs a round trip plus 1.2 ms for a frame's 78 k points; this file produces the same permutation
// without leaving HBM.
//
// std::sort(first, last) is (bits/stl_algo.h)
//     __introsort_loop(first, last, 2 * floor(log2(n)));   __final_insertion_sort(first, last);
// The loop is a quicksort: while a range is longer than 16, move the median of (first+1, mid, last-1) to `first`, run the unguarded
// Hoare partition of (first+1, last) around it, recurse into [cut, last), continue with [first, cut); a range whose depth budget is
// used up is heap-sorted instead (__partial_sort(first, last, last)). What it leaves is a sequence of ranges of at most 16 elements,
// every element of a range >= every element of the ranges to its left, and the final pass is an insertion sort that moves an element
// left only past strictly greater ones: it never crosses a range boundary and is stable inside a range.
//
// None of that needs one thread:
//  * the ranges of one recursion depth are disjoint and independent;
//  * the Hoare partition of a range is a function of two lists of the ORIGINAL arrangement: L = positions (ascending) where the left
//    scan stops (!(a[p] < pivot)), R = positions (descending) where the right scan stops (!(pivot < a[p]), with `first` itself as the
//    last entry, which is what makes the library's right scan "unguarded"). The sequential loop swaps L[k] with R[k] for k = 0, 1, ...
//    as long as L[k] < R[k] -- a swapped position is never visited again --, and returns min(L[K], R[K-1]) where K is the number of
//    swaps (after the last swap the left scan runs into either the next original stop or the element it has just moved to R[K-1]).
//    Ranks from wavefront ballots, the pairs through rank-indexed tables, the swaps in parallel;
//  * the final pass is an insertion sort per range of <= 16 (one thread each), the heap sort of an exhausted range (adversarial inputs
//    only) one thread running libstdc++'s __make_heap / __sort_heap as written.
//
// Launches: the first SS_BIG_LEVELS recursion depths one launch each -- a 1024-thread workgroup per range longer than SS_LEAF, its 16
// wavefronts streaming contiguous sixteenths of the range in coalesced 64-wide tiles --, then ONE launch in which a workgroup takes a
// range of at most SS_LEAF elements into LDS and runs the rest of its recursion there (every wavefront partitioning sub-ranges
// for itself, a ticket queue between them), final insertion pass included, and writes it back. A range that is still longer than SS_LEAF after the
// big levels (adversarial inputs only) goes through the same code on global memory: slower, same result.
// tests: tests/test_gpu_parity.py::test_device_std_sort_equals_std_sort (against the library's own std::sort: duplicates, the patterns that
// exhaust the depth budget, sizes around every threshold); the voxel-filter parity tests run on top of it.
#include "ctx.hpp"
#include "stdsort_dev.hpp"
#include <climits>
#include <cstdlib>
#include <mutex>

namespace mlh {

#ifdef MLH_STAGE_CLOCK
__device__ unsigned long long g_stage_clk_sort[1024 * 8];
#define MLH_SSTAGE(i)                                                                                  \
    do {                                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                  \
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_stage_clk_sort[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
// per-workgroup totals over all wavefronts (10 ns ticks): [0..2] partitions of <= 64 / <= 256 / longer, [3..5] their ticks, [6] bookkeeping ticks, [7] waiting ticks
__device__ unsigned long long g_stage_clk_sort2[1024 * 16];
#define MLH_SACC(i, v) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) atomicAdd(&g_stage_clk_sort2[blockIdx.x * 16 + (i)], (unsigned long long)(v)); } while (0)
#define MLH_SCLK() wall_clock64()
__device__ unsigned long long g_stage_clk_mid[1024 * 16];
#define MLH_MACC(i, v) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_stage_clk_mid[blockIdx.x * 16 + (i)] += (unsigned long long)(v); } while (0)
#else
#define MLH_MACC(i, v) do { } while (0)
#define MLH_SSTAGE(i) do { } while (0)
#define MLH_SACC(i, v) do { } while (0)
#define MLH_SCLK() 0ull
#endif

namespace {

// (the MLH_SS_* macros: A/B builds of scripts/build_variant.py; the defaults are what ships)
#ifndef MLH_SS_LEAF
#define MLH_SS_LEAF 2048
#endif
#ifndef MLH_SS_BIG_U
#define MLH_SS_BIG_U 4
#endif
#ifndef MLH_SS_BIG_LEVELS
#define MLH_SS_BIG_LEVELS 12
#endif
#ifndef MLH_SS_SWAP_U
#define MLH_SS_SWAP_U 4
#endif
constexpr int SS_LEAF = MLH_SS_LEAF;  // ranges up to this many elements are finished in LDS by one workgroup
constexpr int SS_BIG_WG = 1024;
constexpr int SS_BIG_WAVES = SS_BIG_WG / 64;
constexpr int SS_BIG_U = MLH_SS_BIG_U;           // 64-wide tiles a wavefront of a big level keeps in flight
constexpr int SS_SWAP_U = MLH_SS_SWAP_U;         // swaps a thread of a big level keeps in flight
constexpr int SS_BIG_LEVELS = MLH_SS_BIG_LEVELS;
// Wide ranges (round 5): a range longer than SS_WIDE_MIN is partitioned by SEVERAL workgroups, SS_WIDE_CHUNK elements each, in two launches per level -- the stop
// lists (stdsort_wide_stops_kernel), then the pairing, the swaps and the children (the wide half of stdsort_big_level_kernel) -- instead of streaming through the
// 16 wavefronts of ONE compute unit (28 / 25 / 24 / 19 us for the first four levels of a frame's 62 k points). Same pairs, same cut: the partition is a function
// of the two stop lists of the arrangement it starts from (header), however many workgroups write them down.
#ifndef MLH_SS_WIDE_MIN
#define MLH_SS_WIDE_MIN 8192
#endif
constexpr int SS_WIDE_MIN = MLH_SS_WIDE_MIN;
#ifndef MLH_SS_WIDE_CHUNK
#define MLH_SS_WIDE_CHUNK 2048
#endif
#ifndef MLH_SS_WIDE_EXTRA
#define MLH_SS_WIDE_EXTRA 2
#endif
constexpr int SS_WIDE_CHUNK = MLH_SS_WIDE_CHUNK;                 // elements per workgroup of a wide range: 16 wavefronts x 2 tiles (A/B: 4096 -> 0.2170, 2048 -> 0.2130 ms per thinning call)
constexpr int SS_WIDE_WAVE = SS_WIDE_CHUNK / SS_BIG_WAVES;       // 256 elements per wavefront
constexpr int SS_WIDE_MAXW = 1024;                               // wavefront chunks per range the pairing phase indexes: ranges of up to 262 144 elements
constexpr int SS_WIDE_INFO = 16;                                 // ints per wide range in the info block
constexpr int SS_LEAF_WG = 1024;
// The mid launch (round 6): what the wide levels leave longer than a leaf is finished -- down to pieces of at most SS_LEAF elements -- by ONE launch in which a
// workgroup takes a range of up to SS_MID elements into LDS (keys, vals and the two stop tables -- 16-bit: positions local to the range --: 144 KB) and partitions it
// workgroup-wide there, piece by piece,
// until every piece fits a leaf. Through round 5 these were seven more big levels: a launch each (~9 us: eight dependent trips to global memory for a range of a
// few thousand elements) for a chain of 2-4 partitions per range. Same partition function (wg_partition), same cuts.
#ifndef MLH_SS_MID
#define MLH_SS_MID 12288
#endif
constexpr int SS_MID = MLH_SS_MID;
constexpr int SS_MID_STACK = 64;                                      // ranges a workgroup of the mid launch still owes (depth first)
constexpr int SS_LOCAL_LIST = 2 * SS_LEAF / (SS_THRESHOLD + 1) + 8;   // queue records of a leaf in LDS: the ranges the workgroup-wide phase hands over + one per partition with two children > 16 inside each
// A leaf's ranges longer than this are partitioned by the WHOLE workgroup (16 wavefronts, a sixteenth of the range each), one after the other, before the wavefronts
// go their own ways. With wg_partition (written for global memory: per-wavefront prefix tables, a four-step search per look-up) it did NOT pay for ranges that live in
// LDS (profiles/r05_knockout_experiments.txt item 14: 512: 0.2077, 256: 0.2253, 1024: 0.2008 against 0.2027 ms per thinning call without). With the LDS partition
// function of the mid launch (wg_partition_lds: stops at range-wide ranks, one table entry per look-up) it does (round 6, three alternations): 768: 0.1872-0.1876,
// 512: 0.1884-0.1885, 1024: 0.1904-0.1911, 1536: 0.1899-0.1927 against 0.1924-0.1960 without. Same cuts either way; an oversize leaf (longer than SS_LEAF, worked
// on in GLOBAL memory) always starts workgroup-wide through wg_partition.
#ifndef MLH_SS_LEAF_WIDE
#define MLH_SS_LEAF_WIDE 768
#endif
constexpr int SS_LEAF_WIDE = MLH_SS_LEAF_WIDE;
constexpr int SS_LEAF_STACK = 32;                                     // ranges the workgroup-wide phase still owes (depth first: a handful)

struct SortSeg { int first, last, depth, pad; };

struct StdSortArgs {
    int *keys;          // n: sorted in place (the comparator sees these only)
    int *vals;          // n: carried along
    int *lt, *rt;       // n each: the L / R position tables of a range live at [first, ...) of these
    int *gfin, *glist;  // n each: final-range list / queue records of a range finished on GLOBAL memory (oversize leaves only)
    SortSeg *seg[2];    // ranges longer than SS_LEAF of the current / the next big level
    SortSeg *leaf;      // ranges of 2 .. SS_LEAF elements
    int *cnt;           // [0 .. SS_BIG_LEVELS]: ranges per big level; [SS_CNT_LEAF]: leaves
    int n;
    int over_level;     // the level whose list holds the ranges longer than SS_LEAF the leaf launch still has to take: SS_BIG_LEVELS after the big
                        // levels ran, 0 when they were skipped (then nothing should be there -- `longest` said so -- but if something is, it is sorted)
    int *err;           // pinned host word: set when a wait is given up on (never observed; the result would be a wrong order)
    int *wcl, *wcr;     // wide ranges: left / right stops per wavefront chunk, all wide ranges of the level back to back (they live in glist: unused until the leaf launch)
    int *winfo;         // wide ranges: SS_WIDE_INFO ints per range (in gfin)
    int wide_on;        // this level's wide ranges were prepared by stdsort_wide_stops_kernel
};
constexpr int SS_CNT_LEAF = SS_BIG_LEVELS + 1;
constexpr int SS_CNT = SS_BIG_LEVELS + 2;

__device__ inline int floor_log2(int n) { return ss_floor_log2(n); }

__device__ inline void emit_global(const StdSortArgs &A, int first, int last, int depth, SortSeg *next, int *next_cnt)
{
    const int m = last - first;
    if (m > SS_LEAF) next[atomicAdd(next_cnt, 1)] = SortSeg{first, last, depth, 0};
    else if (m > 1) A.leaf[atomicAdd(&A.cnt[SS_CNT_LEAF], 1)] = SortSeg{first, last, depth, 0};
}

// keys <- the points' slots, vals <- the point indices, one range per cloud (std::sort is called once per cloud)
template <bool GEN>
__global__ __launch_bounds__(256) void stdsort_init_kernel(StdSortArgs A, const int *src_keys, int n0, VoxKeyGen G)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < A.n) {
        int key;
        if constexpr (GEN) {
            // the voxel of point i (voxelgrid.hip: vox_mark_kernel -- floor(x * inv_leaf) - min_b per axis, x fastest; the second cloud's grid behind the first's)
            const bool first = i < G.n0;
            const float *p = reinterpret_cast<const float *>(first ? G.src0 + size_t(i) * G.stride : G.src1 + size_t(i - G.n0) * G.stride);
            const float il = first ? G.inv_leaf0 : G.inv_leaf1;
            const int *mb = first ? G.min_b0 : G.min_b1;
            const int ijk0 = int(floorf(p[0] * il) - float(mb[0]));
            const int ijk1 = int(floorf(p[1] * il) - float(mb[1]));
            const int ijk2 = int(floorf(p[2] * il) - float(mb[2]));
            key = first ? ijk0 + ijk1 * G.mul1_0 + ijk2 * G.mul2_0 : G.cell_off1 + ijk0 + ijk1 * G.mul1_1 + ijk2 * G.mul2_1;
        } else {
            key = src_keys[i];
        }
        A.keys[i] = key; A.vals[i] = i;
    }
    if (i == 0) {
        for (int k = 0; k < SS_CNT; ++k) A.cnt[k] = 0;
        const int lo[2] = {0, n0}, hi[2] = {n0, A.n};
        for (int c = 0; c < 2; ++c)
            if (hi[c] - lo[c] > 1) emit_global(A, lo[c], hi[c], 2 * floor_log2(hi[c] - lo[c]), A.seg[0], &A.cnt[0]);
    }
}

// the same for many ranges (one std::sort call per segment): segment s covers [offsets[s * stride + field], + counts[s * stride + field])
// of the key array; vals <- the GLOBAL element index. Keys outside every segment are never looked at.
__global__ __launch_bounds__(256) void stdsort_init_segments_kernel(StdSortArgs A, const int *src_keys, const int *counts, const int *offsets, int stride, int field,
                                                                    int n_segments)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < A.n) { A.keys[i] = src_keys[i]; A.vals[i] = i; }
    if (i < n_segments) {                                            // the counters were cleared by the launch before this one
        const int c = counts[i * stride + field], o = offsets[i * stride + field];
        if (c > 1 && o >= 0 && o + c <= A.n) emit_global(A, o, o + c, 2 * floor_log2(c), A.seg[0], &A.cnt[0]);
    }
}

// the comparator of the voxel filters: int voxel slots (stdsort_dev.hpp holds the pieces shared with extract.hip's per-sector sort)
__device__ inline void median_to_first(int *keys, int *vals, int f, int l) { ss_median_to_first(keys, vals, f, l, IntLess()); }
__device__ inline void heap_sort_range(int *k, int *v, int len) { ss_heap_sort_range(k, v, len, IntLess()); }
// __unguarded_partition_pivot of [f, l) by a WHOLE 1024-thread workgroup (all threads converged): its 16 wavefronts stream contiguous sixteenths of the
// range in coalesced 64-wide tiles, ONCE -- each leaves its chunk's left and right stops, ascending, in its own stretch of the tables; a prefix over the
// wavefronts' counts (LDS) turns a rank of the whole range into (chunk, entry) by a four-step search; the number of crossing pairs comes from a 64-ary search
// over that predicate, and the pairs are swapped in parallel. (Rounds 2-3 streamed the keys twice, a count pass in front of the table pass: the launch is
// bound by instruction issue on its ONE compute unit, and the count pass was a third of it -- thinning 0.52 -> 0.48 ms per frame without it.)
// keys / vals / lt / rt: global memory; w_left / w_right (17 ints each) and sh_k (TWO ints: the number of crossing pairs, the cut): LDS. Returns the cut.
template <typename Tab>
__device__ __forceinline__ int wg_partition(int *keys, int *vals, Tab *lt, Tab *rt, int f, int l, int *w_left, int *w_right, int *sh_k)
{
    // w_left / w_right: 17 ints each. In: scratch. During the call: per-wavefront stop counts, then w_left[w] = left stops in the chunks before w (w = 0 .. 16),
    // w_right[w] = right stops in chunks w .. 15 (w_right[16] = 0).
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = l - f;
    if (t == 0) { median_to_first(keys, vals, f, l); *sh_k = 0; }
    __syncthreads();
    const int piv = keys[f];
    const int chunk = ((m + SS_BIG_WAVES - 1) / SS_BIG_WAVES + 63) & ~63;      // per wavefront, a multiple of the tile
    const int lo = min(f + wave * chunk, l), hi = min(lo + chunk, l);
    int cl, cr;
    // ONE pass over the keys (the launch is bound by instruction issue on its one compute unit: a big level streams a 62 k range through 16 wavefronts):
    // every wavefront leaves its chunk's stops, ascending, in its own stretch of the tables (lt[lo ..], rt[lo ..])
    ss_wave_stop_lists<SS_BIG_U>(keys, lt, rt, f, lo, hi, piv, cl, cr, IntLess());
    if (lane == 0) { w_left[wave + 1] = cl; w_right[wave] = cr; }
    __syncthreads();
    if (t < 64) {
        // prefix of the left counts, suffix of the right counts over the 16 wavefronts, on 16 lanes (a thread walking the two tables one LDS round trip at a time
        // was a third of a partition's fixed cost: 2 of ~7 us for a range of a few thousand elements)
        static_assert(SS_BIG_WAVES == 16, "one lane per wavefront of the workgroup");
        const int w = lane & 15;
        int il = w_left[w + 1], ir = w_right[15 - w];                 // ir: counts from the LAST wavefront backwards
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int a = __shfl_up(il, off, 16), b = __shfl_up(ir, off, 16); if (w >= off) { il += a; ir += b; } }
        if (lane < 16) { w_left[w + 1] = il; w_right[15 - w] = ir; }
        if (lane == 0) { w_left[0] = 0; w_right[SS_BIG_WAVES] = 0; }
    }
    __syncthreads();
    const int nL = w_left[SS_BIG_WAVES], nR = w_right[0];
    // the k-th left stop of the RANGE: chunk w = the largest one with w_left[w] <= k, entry k - w_left[w] of its list; the k-th right stop counted from the right:
    // chunk w = the largest one with w_right[w] > k, entry (its count) - 1 - (k - w_right[w + 1]) of its ascending list. Four LDS look-ups each.
    auto left_at = [&](int k) {
        int w = 0;
#pragma unroll
        for (int step = SS_BIG_WAVES / 2; step > 0; step >>= 1) w += (w_left[w + step] <= k) ? step : 0;
        return int(lt[min(f + w * chunk, l) + (k - w_left[w])]);
    };
    auto right_at = [&](int k) {
        int w = 0;
#pragma unroll
        for (int step = SS_BIG_WAVES / 2; step > 0; step >>= 1) w += (w_right[w + step] > k) ? step : 0;
        const int after = w_right[w + 1], cnt = w_right[w] - after;
        return int(rt[min(f + w * chunk, l) + (cnt - 1 - (k - after))]);
    };
    // K = how many pairs cross (L[k] < R[k] holds for a prefix of k): a 64-ary search by the first wavefront instead of testing every pair
    const int npair = min(nL, nR);
    if (wave == 0) {
        int lo_k = 0, hi_k = npair;
        while (hi_k > lo_k) {                                        // uniform
            const int step = (hi_k - lo_k + 63) >> 6;
            const int k = lo_k + lane * step;
            const bool p = k < hi_k && left_at(k) < right_at(k);
            const int c = __popcll(__ballot(p));
            const int new_hi = min(hi_k, lo_k + c * step);
            lo_k = c > 0 ? lo_k + (c - 1) * step + 1 : lo_k;
            hi_k = c > 0 ? max(new_hi, lo_k) : lo_k;
        }
        if (lane == 0) *sh_k = lo_k;
    }
    __syncthreads();
    const int K = *sh_k;
    for (int k0 = t; k0 < K; k0 += SS_SWAP_U * SS_BIG_WG) {      // SS_SWAP_U swaps in flight: positions, then the elements, then the stores
        int p[SS_SWAP_U], q[SS_SWAP_U], kp[SS_SWAP_U], kq[SS_SWAP_U], vp[SS_SWAP_U], vq[SS_SWAP_U];
#pragma unroll
        for (int u = 0; u < SS_SWAP_U; ++u) { const int k = k0 + u * SS_BIG_WG; const bool on = k < K; p[u] = on ? left_at(k) : -1; q[u] = on ? right_at(k) : -1; }
#pragma unroll
        for (int u = 0; u < SS_SWAP_U; ++u) if (p[u] >= 0) { kp[u] = keys[p[u]]; kq[u] = keys[q[u]]; vp[u] = vals[p[u]]; vq[u] = vals[q[u]]; }
#pragma unroll
        for (int u = 0; u < SS_SWAP_U; ++u) if (p[u] >= 0) { keys[p[u]] = kq[u]; keys[q[u]] = kp[u]; vals[p[u]] = vq[u]; vals[q[u]] = vp[u]; }
    }
    int cut = INT_MAX;
    if (t == 0) {
        if (K < nL) cut = min(cut, left_at(K));
        if (K > 0) cut = min(cut, right_at(K - 1));
        sh_k[1] = cut;                                              // (NOT the word K is read from: a wavefront that is late -- behind another context's kernels on
                                                                     //  this compute unit -- may not have read K yet; through round 6 both lived in sh_k[0], and four
                                                                     //  contexts sorting at once turned that into wrong swaps: scripts/soak_api.py, 4 threads)
    }
    __syncthreads();
    cut = sh_k[1];
    __syncthreads();                                             // the tables and sh_k / w_* are reused by the next range
    return cut;
}

// ------------------------------------------------------------------ wide ranges: several workgroups per range
__device__ __forceinline__ bool ss_is_wide(const SortSeg &s)
{
    const int m = s.last - s.first;
    return s.depth > 0 && m > SS_WIDE_MIN && m <= SS_WIDE_MAXW * SS_WIDE_WAVE;
}

// Which wide range and which of its chunks does workgroup `wg` serve? The level's ranges are walked 64 at a time by the first wavefront (a scan of the wide ranges'
// chunk counts); sh[0] = index of the range in the level's list (-1: none), sh[1] = chunk, sh[2] = chunks of the range, sh[3] = chunks of the wide ranges before it,
// sh[4] = its ordinal among the level's wide ranges. All threads of the workgroup, converged; returns after a barrier.
__device__ __forceinline__ void ss_wide_locate(const SortSeg *cur, int count, int wg, int *sh)
{
    const int t = threadIdx.x, lane = t & 63;
    if (t == 0) sh[0] = -1;
    __syncthreads();
    if (t < 64) {
        int base = 0, ord_base = 0;
        for (int b0 = 0; b0 < count; b0 += 64) {                      // uniform
            const int si = b0 + lane;
            SortSeg s{0, 0, 0, 0};
            if (si < count) s = cur[si];
            const bool wide = si < count && ss_is_wide(s);
            const int nch = wide ? (s.last - s.first + SS_WIDE_CHUNK - 1) / SS_WIDE_CHUNK : 0;
            int incl = nch, oin = wide ? 1 : 0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int v = __shfl_up(incl, off), o = __shfl_up(oin, off);
                if (lane >= off) { incl += v; oin += o; }
            }
            const int excl = incl - nch;
            if (wide && wg >= base + excl && wg < base + incl) { sh[0] = si; sh[1] = wg - (base + excl); sh[2] = nch; sh[3] = base + excl; sh[4] = ord_base + oin - 1; }
            base += __shfl(incl, 63);
            ord_base += __shfl(oin, 63);
            if (base > wg) break;                                     // uniform: found (or passed)
        }
    }
    __syncthreads();
}

// the median of (first + 1, mid, last - 1) WITHOUT moving it (ss_median_to_first's choice): a wide range's workgroups all read the arrangement the level started
// from; position `first` is treated as holding the median's element and the median's position as holding first's ("virtual" move, made real by the range's
// leading workgroup in the second launch)
__device__ __forceinline__ int ss_median_pos(int f, int l, int ka, int kb, int kc)
{
    const int ia = f + 1, ib = f + (l - f) / 2, ic = l - 1;
    IntLess less;
    if (less(ka, kb)) return less(kb, kc) ? ib : (less(ka, kc) ? ic : ia);
    return less(ka, kc) ? ia : (less(kb, kc) ? ic : ib);
}

// first launch of a level with wide ranges: every workgroup writes down the left and right stops of its SS_WIDE_CHUNK elements (per wavefront, ascending, at the
// wavefront chunk's own stretch of lt / rt) and their counts
__global__ __launch_bounds__(SS_BIG_WG) void stdsort_wide_stops_kernel(StdSortArgs A, int level)
{
    __shared__ int sh[8];
    const SortSeg *cur = A.seg[level & 1];
    const int count = A.cnt[level];
    ss_wide_locate(cur, count, blockIdx.x, sh);
    if (sh[0] < 0) return;
    const SortSeg s = cur[sh[0]];
    const int chunk = sh[1], cbase = sh[3], ord = sh[4];
    const int f = s.first, l = s.last, m = l - f;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int ka = A.keys[f + 1], kb = A.keys[f + m / 2], kc = A.keys[l - 1], k0 = A.keys[f];
    const int med = ss_median_pos(f, l, ka, kb, kc);
    const int piv = med == f + 1 ? ka : (med == l - 1 ? kc : kb);
    const int lo = min(f + chunk * SS_WIDE_CHUNK + wave * SS_WIDE_WAVE, l), hi = min(lo + SS_WIDE_WAVE, l);
    const unsigned long long below = ss_lanes_below();
    IntLess less;
    int run_l = 0, run_r = 0, med_l = -1, med_r = -1;
    int k[SS_WIDE_WAVE / 64];
#pragma unroll
    for (int u = 0; u < SS_WIDE_WAVE / 64; ++u) { const int p = lo + 64 * u + lane; k[u] = p < hi ? A.keys[p] : 0; }
#pragma unroll
    for (int u = 0; u < SS_WIDE_WAVE / 64; ++u) {
        const int p = lo + 64 * u + lane;
        const bool in = p < hi;
        const int key = p == f ? piv : (p == med ? k0 : k[u]);
        const bool is_l = in && p > f && !less(key, piv);
        const bool is_r = in && (p == f || !less(piv, key));
        const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
        const int rl = run_l + __popcll(ml & below), rr = run_r + __popcll(mr & below);
        if (is_l) A.lt[lo + rl] = p;
        if (is_r) A.rt[lo + rr] = p;
        if (in && p == med) { med_l = is_l ? rl : -1; med_r = is_r ? rr : -1; }
        run_l += __popcll(ml);
        run_r += __popcll(mr);
    }
    if (lane == 0) { A.wcl[(cbase + chunk) * SS_BIG_WAVES + wave] = run_l; A.wcr[(cbase + chunk) * SS_BIG_WAVES + wave] = run_r; }
    // what the second launch needs to know about the median: its position, both elements, and -- from the lane that met it -- its entries in its wavefront's lists
    int *info = A.winfo + ord * SS_WIDE_INFO;
    if (chunk == 0 && t == 0) { info[0] = med; info[1] = k0; info[2] = A.vals[f]; info[3] = piv; info[4] = A.vals[med]; }
    if (med >= lo && med < hi && lane == ((med - lo) & 63)) { info[5] = med_l; info[6] = med_r; info[7] = (med - f) / SS_WIDE_WAVE; }
}

// second launch, the wide half: pairing, swaps, the median's move, the children -- by all workgroups of the range (every one rebuilds the prefix tables and finds K
// for itself; the swaps are dealt round-robin; the range's first workgroup moves the median and emits the children)
__device__ __forceinline__ void ss_wide_pairs(const StdSortArgs &A, const SortSeg &s, int chunk, int nchunks, int cbase, int ord, int level, int *w_left, int *w_right, int *sh)
{
    const int f = s.first, l = s.last, t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int NW = nchunks * SS_BIG_WAVES;
    const int *info = A.winfo + ord * SS_WIDE_INFO;
    const int med = info[0], k0 = info[1], v0 = info[2], piv = info[3], vmed = info[4];
    // prefix of the left counts, suffix of the right counts over the range's wavefront chunks (one per thread; SS_WIDE_MAXW = the workgroup's size)
    const int cl = t < NW ? A.wcl[cbase * SS_BIG_WAVES + t] : 0, cr = t < NW ? A.wcr[cbase * SS_BIG_WAVES + t] : 0;
    int il = cl, ir = cr;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int a = __shfl_up(il, off), b = __shfl_up(ir, off); if (lane >= off) { il += a; ir += b; } }
    __shared__ int s_wl[SS_BIG_WAVES], s_wr[SS_BIG_WAVES];
    if (lane == 63) { s_wl[wave] = il; s_wr[wave] = ir; }
    __syncthreads();
    int bl = 0, br = 0, nL = 0, nR = 0;
#pragma unroll
    for (int w = 0; w < SS_BIG_WAVES; ++w) { bl += w < wave ? s_wl[w] : 0; br += w < wave ? s_wr[w] : 0; nL += s_wl[w]; nR += s_wr[w]; }
    w_left[t + 1] = bl + il;                                       // left stops in the chunks 0 .. t
    w_right[t] = nR - (br + ir - cr);                              // right stops in the chunks t .. NW - 1
    if (t == 0) { w_left[0] = 0; w_right[SS_WIDE_MAXW] = 0; sh[5] = 0; }
    __syncthreads();
    auto wave_lo = [&](int w) { return min(f + w * SS_WIDE_WAVE, l); };
    auto left_at = [&](int k) {
        int w = 0;
#pragma unroll
        for (int step = SS_WIDE_MAXW / 2; step > 0; step >>= 1) w += (w_left[w + step] <= k) ? step : 0;
        return A.lt[wave_lo(w) + (k - w_left[w])];
    };
    auto right_at = [&](int k) {
        int w = 0;
#pragma unroll
        for (int step = SS_WIDE_MAXW / 2; step > 0; step >>= 1) w += (w_right[w + step] > k) ? step : 0;
        const int after = w_right[w + 1], cnt = w_right[w] - after;
        return A.rt[wave_lo(w) + (cnt - 1 - (k - after))];
    };
    const int npair = min(nL, nR);
    if (wave == 0) {
        int lo_k = 0, hi_k = npair;
        while (hi_k > lo_k) {                                        // uniform
            const int step = (hi_k - lo_k + 63) >> 6;
            const int k = lo_k + lane * step;
            const bool p = k < hi_k && left_at(k) < right_at(k);
            const int c = __popcll(__ballot(p));
            const int new_hi = min(hi_k, lo_k + c * step);
            lo_k = c > 0 ? lo_k + (c - 1) * step + 1 : lo_k;
            hi_k = c > 0 ? max(new_hi, lo_k) : lo_k;
        }
        if (lane == 0) sh[5] = lo_k;
    }
    __syncthreads();
    const int K = sh[5];
    for (int k0_ = chunk * SS_BIG_WG + t; k0_ < K; k0_ += SS_SWAP_U * nchunks * SS_BIG_WG) {
        int p[SS_SWAP_U], q[SS_SWAP_U], kp[SS_SWAP_U], kq[SS_SWAP_U], vp[SS_SWAP_U], vq[SS_SWAP_U];
#pragma unroll
        for (int u = 0; u < SS_SWAP_U; ++u) { const int k = k0_ + u * nchunks * SS_BIG_WG; const bool on = k < K; p[u] = on ? left_at(k) : -1; q[u] = on ? right_at(k) : -1; }
#pragma unroll
        for (int u = 0; u < SS_SWAP_U; ++u) if (p[u] >= 0) {
            // (the median's position still holds its old element: it reads as first's; `first` itself is never part of a pair)
            kp[u] = p[u] == med ? k0 : A.keys[p[u]]; vp[u] = p[u] == med ? v0 : A.vals[p[u]];
            kq[u] = q[u] == med ? k0 : A.keys[q[u]]; vq[u] = q[u] == med ? v0 : A.vals[q[u]];
        }
#pragma unroll
        for (int u = 0; u < SS_SWAP_U; ++u) if (p[u] >= 0) { A.keys[p[u]] = kq[u]; A.keys[q[u]] = kp[u]; A.vals[p[u]] = vq[u]; A.vals[q[u]] = vp[u]; }
    }
    if (chunk == 0 && t == 0) {
        // the median's move, made real: `first` takes the median's element; the median's position takes first's unless a pair has (or will have) written there
        A.keys[f] = piv; A.vals[f] = vmed;
        const int wm = info[7], e_l = info[5], e_r = info[6];
        bool swapped = false;
        if (e_l >= 0) swapped = swapped || (w_left[wm] + e_l) < K;
        if (e_r >= 0) { const int after = w_right[wm + 1], cnt = w_right[wm] - after; swapped = swapped || (after + (cnt - 1 - e_r)) < K; }
        if (!swapped) { A.keys[med] = k0; A.vals[med] = v0; }
        int cut = INT_MAX;
        if (K < nL) cut = min(cut, left_at(K));
        if (K > 0) cut = min(cut, right_at(K - 1));
        SortSeg *next = A.seg[(level + 1) & 1];
        emit_global(A, cut, l, s.depth - 1, next, &A.cnt[level + 1]);     // the recursive call
        emit_global(A, f, cut, s.depth - 1, next, &A.cnt[level + 1]);     // the loop's next trip
    }
}

// ------------------------------------------------------------------ big levels: one 1024-thread workgroup per range longer than SS_LEAF (workgroups gridDim.x - n_wide_wg ..),
// several per wide range (workgroups 0 .. n_wide_wg - 1, when stdsort_wide_stops_kernel has run for this level)
__global__ __launch_bounds__(SS_BIG_WG) void stdsort_big_level_kernel(StdSortArgs A, int level, int n_wide_wg)
{
    __shared__ int w_left[SS_WIDE_MAXW + 1], w_right[SS_WIDE_MAXW + 1];
    __shared__ int sh_k[2];
    __shared__ int sh[8];
    const SortSeg *cur = A.seg[level & 1];
    SortSeg *next = A.seg[(level + 1) & 1];
    const int count = A.cnt[level];
    const int t = threadIdx.x;
    if (int(blockIdx.x) < n_wide_wg) {
        ss_wide_locate(cur, count, blockIdx.x, sh);
        if (sh[0] < 0) return;
        const SortSeg s = cur[sh[0]];
        ss_wide_pairs(A, s, sh[1], sh[2], sh[3], sh[4], level, w_left, w_right, sh);
        return;
    }
    const int n_classic = gridDim.x - n_wide_wg;
    for (int si = blockIdx.x - n_wide_wg; si < count; si += n_classic) {
        const SortSeg s = cur[si];
        if (n_wide_wg > 0 && ss_is_wide(s)) continue;                 // served by the wide workgroups
        const int f = s.first, l = s.last, m = l - f;
        if (s.depth == 0) {                                          // __partial_sort(first, last, last): sorted for good, no children
            if (t == 0) heap_sort_range(A.keys + f, A.vals + f, m);
            continue;
        }
        const int cut = wg_partition(A.keys, A.vals, A.lt, A.rt, f, l, w_left, w_right, sh_k);
        if (t == 0) {
            emit_global(A, cut, l, s.depth - 1, next, &A.cnt[level + 1]);     // the recursive call
            emit_global(A, f, cut, s.depth - 1, next, &A.cnt[level + 1]);     // the loop's next trip
        }
    }
}

// ------------------------------------------------------------------ the mid launch: ranges of SS_LEAF < m <= SS_MID in LDS, workgroup-wide, down to leaves
// [f, l) of the global arrays, SS_LEAF < l - f <= SS_MID, into LDS; partitioned there until every piece is at most SS_LEAF long (those go to the leaf list, in
// global coordinates: the leaf launch finds them in HBM); written back. All threads of the workgroup, converged.
typedef unsigned short MidTab;                                        // a position inside a range of at most SS_MID (< 65 536) elements
// wg_partition for a range that lives in LDS: the same stop lists, pairs and cut, with the tables written at range-wide ranks. Counting the stops first (one more
// pass over keys that are a ds_read away) puts every stop at its final index -- L[k] = lt[f + k], R[k] = rt[f + nR - 1 - k] -- so the 64-ary search for the number of
// crossing pairs and every swap read ONE table entry instead of walking the per-wavefront prefix tables (a four-step search through LDS per look-up: a third of a
// partition's 4 us on these sizes). On global memory the second pass is what costs (stdsort.hip header); here it is what saves.
template <typename Tab>
__device__ __forceinline__ int wg_partition_lds(int *keys, int *vals, Tab *lt, Tab *rt, int f, int l, int *w_left, int *w_right, int *sh_k)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, m = l - f;
    if (t == 0) { median_to_first(keys, vals, f, l); *sh_k = 0; }
    __syncthreads();
    const int piv = keys[f];
    const int chunk = ((m + SS_BIG_WAVES - 1) / SS_BIG_WAVES + 63) & ~63;      // per wavefront, a multiple of the tile
    const int lo = min(f + wave * chunk, l), hi = min(lo + chunk, l);
    const unsigned long long below = ss_lanes_below();
    IntLess less;
    int cl = 0, cr = 0;
    for (int base = lo; base < hi; base += 64) {
        const int p = base + lane;
        const bool in = p < hi;
        const int k = in ? keys[p] : 0;
        cl += __popcll(__ballot(in && p > f && !less(k, piv)));
        cr += __popcll(__ballot(in && (p == f || !less(piv, k))));
    }
    if (lane == 0) { w_left[wave + 1] = cl; w_right[wave] = cr; }
    __syncthreads();
    if (t < 64) {                                                    // (the prefix / suffix over the 16 wavefronts, as in wg_partition)
        const int w = lane & 15;
        int il = w_left[w + 1], ir = w_right[15 - w];
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const int a = __shfl_up(il, off, 16), b = __shfl_up(ir, off, 16); if (w >= off) { il += a; ir += b; } }
        if (lane < 16) { w_left[w + 1] = il; w_right[15 - w] = ir; }
        if (lane == 0) { w_left[0] = 0; w_right[SS_BIG_WAVES] = 0; }
    }
    __syncthreads();
    const int nL = w_left[SS_BIG_WAVES], nR = w_right[0];
    int run_l = w_left[wave], run_r = nR - w_right[wave];            // left / right stops in the chunks before this one (both lists ascending by position)
    for (int base = lo; base < hi; base += 64) {
        const int p = base + lane;
        const bool in = p < hi;
        const int k = in ? keys[p] : 0;
        const bool is_l = in && p > f && !less(k, piv);
        const bool is_r = in && (p == f || !less(piv, k));
        const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
        if (is_l) lt[f + run_l + __popcll(ml & below)] = Tab(p);
        if (is_r) rt[f + run_r + __popcll(mr & below)] = Tab(p);
        run_l += __popcll(ml);
        run_r += __popcll(mr);
    }
    __syncthreads();
    const int npair = min(nL, nR), rlast = f + nR - 1;               // the k-th right stop from the right: rt[rlast - k]
    if (wave == 0) {
        int lo_k = 0, hi_k = npair;
        while (hi_k > lo_k) {                                        // uniform
            const int step = (hi_k - lo_k + 63) >> 6;
            const int k = lo_k + lane * step;
            const bool pr = k < hi_k && int(lt[f + k]) < int(rt[rlast - k]);
            const int c = __popcll(__ballot(pr));
            const int new_hi = min(hi_k, lo_k + c * step);
            lo_k = c > 0 ? lo_k + (c - 1) * step + 1 : lo_k;
            hi_k = c > 0 ? max(new_hi, lo_k) : lo_k;
        }
        if (lane == 0) *sh_k = lo_k;
    }
    __syncthreads();
    const int K = *sh_k;
    for (int k = t; k < K; k += SS_BIG_WG) {
        const int p = int(lt[f + k]), q = int(rt[rlast - k]);
        const int kp = keys[p], kq = keys[q], vp = vals[p], vq = vals[q];
        keys[p] = kq; keys[q] = kp; vals[p] = vq; vals[q] = vp;
    }
    int cut = INT_MAX;
    if (t == 0) {
        if (K < nL) cut = min(cut, int(lt[f + K]));
        if (K > 0) cut = min(cut, int(rt[rlast - (K - 1)]));
        sh_k[1] = cut;                                              // (NOT the word K is read from: a wavefront that is late -- behind another context's kernels on
                                                                     //  this compute unit -- may not have read K yet; through round 6 both lived in sh_k[0], and four
                                                                     //  contexts sorting at once turned that into wrong swaps: scripts/soak_api.py, 4 threads)
    }
    __syncthreads();
    cut = sh_k[1];
    __syncthreads();                                                 // the tables and sh_k / w_* are reused by the next range
    return cut;
}
constexpr size_t SS_MID_LDS = size_t(SS_MID) * (2 * sizeof(int) + 2 * sizeof(MidTab));
static_assert(SS_MID < 65536 && SS_MID_LDS <= 150 * 1024, "the mid launch's range has to fit 16-bit positions and the compute unit's LDS");
__device__ __forceinline__ void mid_lds_subtree(const StdSortArgs &A, int f, int l, int depth0, int *sk, int *sv, MidTab *slt, MidTab *srt, int *wl, int *wr, int *sh_k, int *stk)
{
    const int t = threadIdx.x, m = l - f;
    for (int i = t; i < m; i += SS_BIG_WG) { sk[i] = A.keys[f + i]; sv[i] = A.vals[f + i]; }
    int top = 0;
    auto route = [&](int lo, int hi, int d) {
        const int size = hi - lo;
        if (size > SS_LEAF && top < SS_MID_STACK) {
            if (t == 0) { stk[3 * top] = lo; stk[3 * top + 1] = hi; stk[3 * top + 2] = d; }
            ++top;                                                   // (uniform: every thread keeps the count)
        } else if (size > 1 && t == 0) {
            A.leaf[atomicAdd(&A.cnt[SS_CNT_LEAF], 1)] = SortSeg{f + lo, f + hi, d, 0};      // (longer than a leaf only if the stack is full: the leaf launch's global path)
        }
    };
    route(0, m, depth0);
    while (top > 0) {                                                // uniform
        __syncthreads();                                             // thread 0's stack writes (first trip: the loads above)
        --top;
        const int a = stk[3 * top], b = stk[3 * top + 1], d = stk[3 * top + 2];
        __syncthreads();                                             // read by everybody before the slot is written again
        if (d == 0) {                                                // __partial_sort(first, last, last): sorted for good, no children
            [[maybe_unused]] const unsigned long long ch = MLH_SCLK();
            if (t == 0) heap_sort_range(sk + a, sv + a, b - a);
            MLH_MACC(3, 1); MLH_MACC(4, MLH_SCLK() - ch); MLH_MACC(5, b - a);
            continue;
        }
        [[maybe_unused]] const unsigned long long cp = MLH_SCLK();
        const int cut = wg_partition_lds(sk, sv, slt, srt, a, b, wl, wr, sh_k);
        MLH_MACC(0, 1); MLH_MACC(1, MLH_SCLK() - cp); MLH_MACC(2, b - a);
        route(cut, b, d - 1);                                        // the library's recursive call
        route(a, cut, d - 1);                                        // its loop's next trip
    }
    __syncthreads();
    for (int i = t; i < m; i += SS_BIG_WG) { A.keys[f + i] = sk[i]; A.vals[f + i] = sv[i]; }
    __syncthreads();
}

__global__ __launch_bounds__(SS_BIG_WG) void stdsort_mid_kernel(StdSortArgs A, int level)
{
    extern __shared__ int s_mid[];                                   // keys | vals (SS_MID ints each) | left stops | right stops (SS_MID 16-bit positions each)
    __shared__ int w_left[SS_BIG_WAVES + 1], w_right[SS_BIG_WAVES + 1], sh_k[2];
    __shared__ int s_stk[3 * SS_MID_STACK], s_gstk[3 * SS_MID_STACK];
    int *sk = s_mid, *sv = s_mid + SS_MID;
    MidTab *slt = reinterpret_cast<MidTab *>(s_mid + 2 * SS_MID), *srt = slt + SS_MID;
    const SortSeg *cur = A.seg[level & 1];
    const int count = A.cnt[level];
    const int t = threadIdx.x;
    for (int si = blockIdx.x; si < count; si += gridDim.x) {
        const SortSeg s = cur[si];
        [[maybe_unused]] const unsigned long long c_all = MLH_SCLK();
        MLH_MACC(7, s.last - s.first);
        if (s.last - s.first <= SS_MID) { mid_lds_subtree(A, s.first, s.last, s.depth, sk, sv, slt, srt, w_left, w_right, sh_k, s_stk); MLH_MACC(6, MLH_SCLK() - c_all); continue; }
        // still longer than SS_MID after the wide levels (unbalanced partitions; a range too long to be "wide"): partitioned on global memory until its pieces fit
        int top = 0;
        if (t == 0) { s_gstk[0] = s.first; s_gstk[1] = s.last; s_gstk[2] = s.depth; }
        ++top;
        while (top > 0) {                                            // uniform
            __syncthreads();
            --top;
            const int a = s_gstk[3 * top], b = s_gstk[3 * top + 1], d = s_gstk[3 * top + 2];
            __syncthreads();
            if (d == 0) {
                if (t == 0) heap_sort_range(A.keys + a, A.vals + a, b - a);
                continue;
            }
            const int cut = wg_partition(A.keys, A.vals, A.lt, A.rt, a, b, w_left, w_right, sh_k);
            const int c0[2] = {cut, a}, c1[2] = {b, cut};
            for (int c = 0; c < 2; ++c) {                            // the library's recursive call, then its loop's next trip
                const int size = c1[c] - c0[c];
                if (size > SS_MID && top < SS_MID_STACK) {
                    if (t == 0) { s_gstk[3 * top] = c0[c]; s_gstk[3 * top + 1] = c1[c]; s_gstk[3 * top + 2] = d - 1; }
                    ++top;
                } else if (size > SS_LEAF && size <= SS_MID) {
                    mid_lds_subtree(A, c0[c], c1[c], d - 1, sk, sv, slt, srt, w_left, w_right, sh_k, s_stk);
                } else if (size > 1 && t == 0) {
                    A.leaf[atomicAdd(&A.cnt[SS_CNT_LEAF], 1)] = SortSeg{c0[c], c1[c], d - 1, 0};
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ leaves: the rest of a range's recursion inside one workgroup
// Arrays in LOCAL coordinates [0, m). Every wavefront works on its own: it partitions a sub-range, keeps one child longer than 16 for
// itself and hands the other to a queue in (first, last, depth budget, ready) records; a wavefront without work takes the next ticket
// of that queue and waits for the record to appear. `remaining` counts the elements not yet in a range of <= 16 (or heap-sorted):
// at zero everybody leaves. At most one record per partition whose children are BOTH longer than 16: < m / 17 of them, plus the root.
struct LeafMem {
    int *keys, *vals, *lt, *rt;
    int *fin;           // (first, last) pairs of ranges of 2..16: the final insertion pass
    int *q;             // queue records, 4 ints each
    int qcap;
    int *scr;           // LDS, 128 ints per wavefront: the stop tables of a partition that fits one 64-wide tile
};
enum { LQ_TAIL = 0, LQ_HEAD = 1, LQ_REMAINING = 2, LQ_NFIN = 3 };

__device__ inline int wg_load(int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void wg_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ inline void wg_fence() { ss_wg_fence(); }

// wl / wr (17 ints each), sk, stk (3 * SS_LEAF_STACK ints): LDS scratch of the workgroup-wide phase
template <bool LDS>      // LDS: M's arrays live in LDS (the workgroup-wide partitions take wg_partition_lds)
__device__ __forceinline__ void leaf_sort(const LeafMem &M, int m, int depth0, int *sh, int *err, int *wl, int *wr, int *sk, int *stk)
{
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 4 * M.qcap; i += SS_LEAF_WG) M.q[i] = 0;
    if (t == 0) { sh[LQ_TAIL] = 0; sh[LQ_HEAD] = 0; sh[LQ_REMAINING] = m; sh[LQ_NFIN] = 0; }
    __syncthreads();
    // where a range goes: the workgroup-wide stack, the wavefronts' queue, the final-insertion list (thread 0; the workgroup-wide phase is single-file)
    auto route = [&](int lo, int hi, int d, int &top) {
        const int size = hi - lo;
        if (size > (LDS ? SS_LEAF_WIDE : SS_LEAF) && top < SS_LEAF_STACK) {
            if (t == 0) { stk[3 * top] = lo; stk[3 * top + 1] = hi; stk[3 * top + 2] = d; }
            ++top;                                                   // (uniform: every thread keeps the count)
        } else if (t == 0) {
            if (size > SS_THRESHOLD) { const int e = sh[LQ_TAIL]++; M.q[4 * e] = lo; M.q[4 * e + 1] = hi; M.q[4 * e + 2] = d; M.q[4 * e + 3] = 1; }
            else { if (size > 1) { const int e = sh[LQ_NFIN]++; M.fin[2 * e] = lo; M.fin[2 * e + 1] = hi; } sh[LQ_REMAINING] -= size; }
        }
    };
    {
        int top = 0;
        route(0, m, depth0, top);
        while (top > 0) {                                            // uniform
            __syncthreads();                                         // thread 0's stack writes
            --top;
            const int f = stk[3 * top], l = stk[3 * top + 1], d = stk[3 * top + 2];
            __syncthreads();                                         // read by everybody before the slot is written again
            if (d == 0) {                                            // __partial_sort(first, last, last): sorted for good, no children
                if (t == 0) { heap_sort_range(M.keys + f, M.vals + f, l - f); sh[LQ_REMAINING] -= l - f; }
                continue;
            }
            const int cut = LDS ? wg_partition_lds(M.keys, M.vals, M.lt, M.rt, f, l, wl, wr, sk) : wg_partition(M.keys, M.vals, M.lt, M.rt, f, l, wl, wr, sk);
            route(cut, l, d - 1, top);                               // the library's recursive call
            route(f, cut, d - 1, top);                               // its loop's next trip
        }
    }
    __syncthreads();
    int f = 0, l = 0, d = 0;
    bool have = false;
    MLH_SSTAGE(1);
    while (true) {
        [[maybe_unused]] const unsigned long long c_wait = MLH_SCLK();
        if (!have) {
            if (wg_load(&sh[LQ_REMAINING]) == 0) break;
            int ticket = 0;
            if (lane == 0) ticket = atomicAdd(&sh[LQ_HEAD], 1);
            ticket = __shfl(ticket, 0);
            bool got = false;
            for (int spins = 0; spins < (1 << 24); ++spins) {        // bounded: a waiting wavefront can only be released by progress elsewhere
                if (ticket < M.qcap && wg_load(&M.q[4 * ticket + 3]) != 0) { got = true; break; }
                if (wg_load(&sh[LQ_REMAINING]) == 0) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (!got) {
                // released by `remaining == 0` (all done) or never: the latter leaves elements unsorted -- say so instead of returning a wrong order
                if (wg_load(&sh[LQ_REMAINING]) != 0 && lane == 0 && err) __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            f = M.q[4 * ticket]; l = M.q[4 * ticket + 1]; d = M.q[4 * ticket + 2];
            have = true;
        }
        const int size = l - f;
        [[maybe_unused]] const unsigned long long c_part = MLH_SCLK();
        MLH_SACC(7, c_part - c_wait);
        if (d == 0) {                                                 // __partial_sort(first, last, last): sorted for good, no children
            if (size <= 64) ss_heap_sort_wave64(M.keys + f, M.vals + f, size, IntLess());     // in registers (the leftovers of an exhausted budget are short)
            else if (lane == 0) heap_sort_range(M.keys + f, M.vals + f, size);
            MLH_SACC(size <= 64 ? 8 : 10, 1); MLH_SACC(size <= 64 ? 9 : 11, MLH_SCLK() - c_part); MLH_SACC(12, size);
            wg_fence();
            if (lane == 0) atomicSub(&sh[LQ_REMAINING], size);
            have = false;
            continue;
        }
        const int cut = ss_wave_partition(M.keys, M.vals, M.lt, M.rt, M.scr + (t >> 6) * 128, f, l, IntLess());
        [[maybe_unused]] const unsigned long long c_book = MLH_SCLK();
        MLH_SACC(size <= 64 ? 0 : (size <= 256 ? 1 : 2), 1);
        MLH_SACC(size <= 64 ? 3 : (size <= 256 ? 4 : 5), c_book - c_part);
        // children: [cut, l) is the library's recursive call, [f, cut) its loop's next trip; both get d - 1
        const int size_a = cut - f, size_b = l - cut;
        const bool big_a = size_a > SS_THRESHOLD, big_b = size_b > SS_THRESHOLD;
        if (lane == 0) {
            int done = 0;
            if (!big_a) { if (size_a > 1) { const int e = atomicAdd(&sh[LQ_NFIN], 1); M.fin[2 * e] = f; M.fin[2 * e + 1] = cut; } done += size_a; }
            if (!big_b) { if (size_b > 1) { const int e = atomicAdd(&sh[LQ_NFIN], 1); M.fin[2 * e] = cut; M.fin[2 * e + 1] = l; } done += size_b; }
            if (big_a && big_b) {                                     // the SHORTER child goes to the queue: the longer one is the likelier critical path,
                const int e = atomicAdd(&sh[LQ_TAIL], 1);             // and a hand-over costs a poll interval (the ranges are disjoint: any order is the same sort)
                const bool push_a = size_a < size_b;
                M.q[4 * e] = push_a ? f : cut; M.q[4 * e + 1] = push_a ? cut : l; M.q[4 * e + 2] = d - 1;
                wg_store(&M.q[4 * e + 3], 1);
            }
            if (done) { wg_fence(); atomicSub(&sh[LQ_REMAINING], done); }
        }
        if (big_a && (!big_b || size_a >= size_b)) { l = cut; d -= 1; }
        else if (big_b) { f = cut; d -= 1; }
        else have = false;
        MLH_SACC(6, MLH_SCLK() - c_book);
    }
    MLH_SSTAGE(2);
    __syncthreads();
    MLH_SSTAGE(3);
    // __final_insertion_sort, range by range. Insertion sort is stable, and with a consistent comparator no element crosses the bounds of the ranges the loop
    // left (<= 16 elements each): what it leaves in a range is the range's STABLE sort. Sixteen lanes per range place every element directly -- its slot is the
    // number of elements that are smaller, or equal and earlier -- instead of one thread shifting elements one LDS round trip at a time (~10 us of a leaf launch).
    const int n_fin = sh[LQ_NFIN];
    static_assert(SS_THRESHOLD == 16, "one 16-lane group per final range");
    for (int i0 = 0; i0 < n_fin; i0 += SS_LEAF_WG / 16) {
        const int i = i0 + (t >> 4), e = t & 15;
        const bool on = i < n_fin;
        const int a = on ? M.fin[2 * i] : 0, b = on ? M.fin[2 * i + 1] : 0, size = b - a;
        const bool mine = on && e < size;
        const int key = mine ? M.keys[a + e] : 0, val = mine ? M.vals[a + e] : 0;
        int slot = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int kj = __shfl(key, (lane & 48) | j);                      // element j of this group's range
            slot += (j < size && (kj < key || (kj == key && j < e))) ? 1 : 0;
        }
        if (mine) { M.keys[a + slot] = key; M.vals[a + slot] = val; }     // all of the group's reads are in registers by now (one wavefront, in order)
    }
    MLH_SSTAGE(4);
    __syncthreads();
    MLH_SSTAGE(5);
}

__global__ __launch_bounds__(SS_LEAF_WG) void stdsort_leaf_kernel(StdSortArgs A)
{
    __shared__ int s_keys[SS_LEAF], s_vals[SS_LEAF], s_lt[SS_LEAF], s_rt[SS_LEAF], s_fin[SS_LEAF];
    __shared__ int s_q[4 * SS_LOCAL_LIST];
    __shared__ int s_scr[(SS_LEAF_WG / 64) * 128];
    __shared__ int sh[4];
    __shared__ int s_wl[SS_BIG_WAVES + 1], s_wr[SS_BIG_WAVES + 1], s_k[2], s_stk[3 * SS_LEAF_STACK];
    static_assert(SS_LEAF_WG == SS_BIG_WG, "wg_partition is written for the big levels' workgroup");
    const int n_leaf = A.cnt[SS_CNT_LEAF], n_left_over = A.cnt[A.over_level];
    const SortSeg *over = A.seg[A.over_level & 1];
    const int t = threadIdx.x;
    MLH_SSTAGE(0);
    for (int si = blockIdx.x; si < n_leaf + n_left_over; si += gridDim.x) {
        const SortSeg s = si < n_leaf ? A.leaf[si] : over[si - n_leaf];
#ifdef MLH_STAGE_CLOCK
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_stage_clk_sort[blockIdx.x * 8 + 7] = (unsigned long long)(s.last - s.first);
#endif
        const int f = s.first, m = s.last - s.first;
        LeafMem M;
        M.scr = s_scr;
        if (m <= SS_LEAF) {
            for (int i = t; i < m; i += SS_LEAF_WG) { s_keys[i] = A.keys[f + i]; s_vals[i] = A.vals[f + i]; }
            M.keys = s_keys; M.vals = s_vals; M.lt = s_lt; M.rt = s_rt; M.fin = s_fin; M.q = s_q; M.qcap = SS_LOCAL_LIST;
            __syncthreads();
            leaf_sort<true>(M, m, s.depth, sh, A.err, s_wl, s_wr, s_k, s_stk);
            for (int i = t; i < m; i += SS_LEAF_WG) { A.keys[f + i] = s_keys[i]; A.vals[f + i] = s_vals[i]; }
            MLH_SSTAGE(6);
            __syncthreads();
        } else {                                                       // still longer than a leaf after the big levels: the same code on global memory
            M.keys = A.keys + f; M.vals = A.vals + f; M.lt = A.lt + f; M.rt = A.rt + f; M.fin = A.gfin + f;
            M.q = A.glist + f; M.qcap = min(2 * (m / (SS_THRESHOLD + 1)) + 2, m / 4);
            leaf_sort<false>(M, m, s.depth, sh, A.err, s_wl, s_wr, s_k, s_stk);
        }
    }
}

}  // namespace

// hipFuncAttributeMaxDynamicSharedMemorySize of stdsort_mid_kernel, once per device; false: the device does not grant it (the big levels run instead)
static bool stdsort_mid_lds_granted(int device)
{
    static std::mutex mu;
    static signed char state[64] = {};          // 0: not asked, 1: granted, -1: refused
    if (device < 0 || device >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (state[device] == 0) {
        const bool ok = hipFuncSetAttribute(reinterpret_cast<const void *>(stdsort_mid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(SS_MID_LDS)) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        state[device] = ok ? 1 : -1;
    }
    return state[device] > 0;
}

static int stdsort_setup(mlh_ctx *ctx, int n, int *vals_out, StdSortArgs &A, size_t &nbig, size_t &nleaf)
{
    DevBuf &S = ctx->stdsort;
    const size_t ni = size_t(n);
    nbig = ni / SS_LEAF + 4; nleaf = ni / 2 + 4;
    // [keys n][lt n][rt n][gfin n][glist n][cnt (padded to 64)][seg0][seg1][leaf]
    const size_t off_lt = ni, off_rt = 2 * ni, off_gfin = 3 * ni, off_glist = 4 * ni, off_cnt = 5 * ni, off_seg0 = off_cnt + 64;
    const size_t seg_ints = nbig * 4, off_seg1 = off_seg0 + seg_ints, off_leaf = off_seg1 + seg_ints, total = off_leaf + nleaf * 4;
    MLH_HIP(ctx, S.ensure(sizeof(int) * total));
    int *base = S.as<int>();
    A.keys = base; A.vals = vals_out; A.lt = base + off_lt; A.rt = base + off_rt; A.gfin = base + off_gfin; A.glist = base + off_glist;
    A.cnt = base + off_cnt;
    A.seg[0] = reinterpret_cast<SortSeg *>(base + off_seg0); A.seg[1] = reinterpret_cast<SortSeg *>(base + off_seg1);
    A.leaf = reinterpret_cast<SortSeg *>(base + off_leaf); A.n = n;
    // wide ranges' per-level scratch lives where the leaf launch's global-memory lists go (nothing else touches them before that launch): counts of at most
    // n / 256 + 16 per wavefront-chunk twice, SS_WIDE_INFO ints per wide range
    A.wcl = A.glist; A.wcr = A.glist + ni / 2; A.winfo = A.gfin; A.wide_on = 0;
    A.over_level = SS_BIG_LEVELS;
    A.err = device_error_word(ctx);
    return MLH_OK;
}

static int stdsort_levels(mlh_ctx *ctx, StdSortArgs A, int longest, size_t nbig, size_t nleaf)
{
    hipStream_t st = ctx->stream;
    // Big levels: always all SS_BIG_LEVELS of them when a range can be longer than a leaf. (Round 3 tried ceil(log2(longest / SS_LEAF)) + 4 launches: the
    // median-of-three partitions of a frame's 62 k voxel slots are unbalanced enough that ranges longer than a leaf survive nine levels and fall into the leaf
    // launch's global-memory path -- the thinning step went 0.56 -> 0.66 ms. An empty level costs 3-4 us; the slow path costs 70.)
    int n_levels = longest > SS_LEAF ? SS_BIG_LEVELS : 0;
    A.over_level = n_levels;                                  // no big level: whatever the init kernel routed to level 0 anyway is finished by the leaf launch
    if (n_levels > 0) {
        const int grid_big = int(std::min<size_t>(nbig, 64));
        // wide levels: as long as a range can still be longer than SS_WIDE_MIN -- the sizes roughly halve per level; two more levels for the unbalanced
        // partitions. A wide range met later than that is partitioned by one workgroup, as all were until round 5 (same result).
        static const bool wide_off = std::getenv("MLH_SS_WIDE_OFF") != nullptr;      // (A/B runs)
        int n_wide_levels = 0;
        const bool fits = size_t(A.n) / SS_WIDE_WAVE + 64 <= size_t(A.n) / 2 && size_t(A.n) / SS_WIDE_MIN * SS_WIDE_INFO + 64 <= size_t(A.n);
        if (!wide_off && fits && longest > SS_WIDE_MIN) {
            int lg = 0;
            while ((SS_WIDE_MIN << lg) < longest) ++lg;
            static const int extra = std::getenv("MLH_SS_WIDE_EXTRA") ? std::atoi(std::getenv("MLH_SS_WIDE_EXTRA")) : MLH_SS_WIDE_EXTRA;      // (A/B runs)
            n_wide_levels = std::min(n_levels, lg + extra);
        }
        // the mid launch behind the wide levels instead of the remaining big levels (MLH_SS_MID_OFF: A/B runs; a device that does not grant its 144 KB of LDS)
        static const bool mid_off = std::getenv("MLH_SS_MID_OFF") != nullptr;
        const bool mid = !mid_off && n_wide_levels < SS_BIG_LEVELS && stdsort_mid_lds_granted(ctx->device);
        if (mid) n_levels = n_wide_levels;
        const int n_wide_wg = A.n / SS_WIDE_CHUNK + A.n / SS_WIDE_MIN + 4;     // >= the chunks of all wide ranges of a level
        for (int level = 0; level < n_levels; ++level) {
            const bool wide = level < n_wide_levels;
            if (wide) MLH_LAUNCH(stdsort_wide_stops_kernel, dim3(n_wide_wg), dim3(SS_BIG_WG), 0, st, A, level);
            MLH_LAUNCH(stdsort_big_level_kernel, dim3(grid_big + (wide ? n_wide_wg : 0)), dim3(SS_BIG_WG), 0, st, A, level, wide ? n_wide_wg : 0);
        }
        if (mid) {
            // one workgroup per range the wide levels left longer than a leaf (at most n / SS_LEAF of them); it leaves nothing for a next level
            const int grid_mid = int(std::min<size_t>(nbig, 256));
            MLH_LAUNCH(stdsort_mid_kernel, dim3(grid_mid), dim3(SS_BIG_WG), SS_MID_LDS, st, A, n_levels);
            A.over_level = n_levels + 1;                          // (an empty list: the init launch cleared every level's counter)
        } else A.over_level = n_levels;
    }
    const int grid_leaf = int(std::min<size_t>(nleaf, 1024));
    MLH_LAUNCH(stdsort_leaf_kernel, dim3(grid_leaf), dim3(SS_LEAF_WG), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// Sorts (keys = src_keys[0..n), vals = 0..n-1) as two std::sort calls would -- [0, n0) and [n0, n) -- and leaves the permuted vals in
// vals_out (device, n ints). Everything is enqueued on the context's stream; nothing is waited for.
int device_std_sort_by_key(mlh_ctx *ctx, const int *src_keys, int n0, int n, int *vals_out, const VoxKeyGen *gen)
{
    if (n <= 0) return MLH_OK;
    StdSortArgs A;
    size_t nbig, nleaf;
    int rc = stdsort_setup(ctx, n, vals_out, A, nbig, nleaf);
    if (rc) return rc;
    if (gen) MLH_LAUNCH(stdsort_init_kernel<true>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A, src_keys, n0, *gen);
    else MLH_LAUNCH(stdsort_init_kernel<false>, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A, src_keys, n0, VoxKeyGen{});
    return stdsort_levels(ctx, A, std::max(n0, n - n0), nbig, nleaf);
}

const int *device_std_sort_keys(mlh_ctx *ctx, int n)
{
    StdSortArgs A;
    size_t nbig, nleaf;
    if (n <= 0 || stdsort_setup(ctx, n, nullptr, A, nbig, nleaf)) return nullptr;
    return A.keys;
}

// One std::sort call per segment (device-side tables of counts and offsets, `stride` ints per segment, entry `field`): vals_out[i] <- the
// global index of the element std::sort leaves at position i of its segment. longest = a host-side upper bound of a segment's length
// (it only decides whether the big-level launches are needed).
// The per-level range counters of a sort over n elements (SS_CNT ints): a caller that has a launch of its own in front of device_std_sort_segments can clear
// them there and pass counters_cleared = true -- hipMemsetAsync of these 56 unaligned bytes is THREE fill launches (~15 us in the frame's kernel trace).
int *device_std_sort_counters(mlh_ctx *ctx, int n, int *n_counters)
{
    StdSortArgs A;
    size_t nbig, nleaf;
    if (n <= 0 || stdsort_setup(ctx, n, nullptr, A, nbig, nleaf)) return nullptr;
    if (n_counters) *n_counters = SS_CNT;
    return A.cnt;
}

__global__ void stdsort_clear_counters_kernel(int *cnt)
{
    if (threadIdx.x < SS_CNT) cnt[threadIdx.x] = 0;
}

int device_std_sort_segments(mlh_ctx *ctx, const int *src_keys, const int *counts, const int *offsets, int stride, int field, int n_segments, int n, int longest,
                             int *vals_out, bool counters_cleared)
{
    if (n <= 0 || n_segments <= 0) return MLH_OK;
    StdSortArgs A;
    size_t nbig, nleaf;
    int rc = stdsort_setup(ctx, n, vals_out, A, nbig, nleaf);
    if (rc) return rc;
    // one thread per segment appends its range: the counters start at zero
    if (!counters_cleared) MLH_LAUNCH(stdsort_clear_counters_kernel, dim3(1), dim3(64), 0, ctx->stream, A.cnt);
    MLH_LAUNCH(stdsort_init_segments_kernel, dim3((std::max(n, n_segments) + 255) / 256), dim3(256), 0, ctx->stream, A, src_keys, counts, offsets, stride, field,
                       n_segments);
    return stdsort_levels(ctx, A, longest, nbig, nleaf);
}

#ifdef MLH_STAGE_CLOCK
}  // namespace mlh
extern "C" int mlh_debug_stage_clock_sort(unsigned long long *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk_sort), sizeof(unsigned long long) * size_t(n_words));
}
extern "C" int mlh_debug_stage_clock_mid(unsigned long long *out, int n_words, int clear)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk_mid), sizeof(unsigned long long) * size_t(n_words));
    if (clear) { static unsigned long long z[1024 * 16]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(mlh::g_stage_clk_mid), z, sizeof(z)); }
    return rc;
}
extern "C" int mlh_debug_stage_clock_sort2(unsigned long long *out, int n_words, int clear)
{
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk_sort2), sizeof(unsigned long long) * size_t(n_words));
    if (clear) { static unsigned long long z[1024 * 16]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(mlh::g_stage_clk_sort2), z, sizeof(z)); }
    return rc;
}
namespace mlh {
#endif
}  // namespace mlh
