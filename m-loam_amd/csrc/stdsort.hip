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
//  * the ranges of one recursion depth are disjoint and independent: one launch per depth, one workgroup per range;
//  * the Hoare partition of a range is a function of two lists of the ORIGINAL arrangement: L = positions (ascending) where the left
//    scan stops (!(a[p] < pivot)), R = positions (descending) where the right scan stops (!(pivot < a[p]), with `first` itself as the
//    last entry, which is what makes the library's right scan "unguarded"). The sequential loop swaps L[k] with R[k] for k = 0, 1, ...
//    as long as L[k] < R[k] -- a swapped position is never visited again --, and returns min(L[K], R[K-1]) where K is the number of
//    swaps (after the last swap the left scan runs into either the next original stop or the element it has just moved to R[K-1]).
//    Ranks by prefix sums, the pairs by a rank-indexed table, the swaps in parallel;
//  * the final pass is an insertion sort per range of <= 16 (one thread each), the heap sort of an exhausted range (adversarial inputs
//    only) one thread running libstdc++'s __make_heap / __sort_heap as written.
// tests: tests/test_gpu_parity.py::test_device_std_sort_equals_std_sort (against std::sort through the oracle library, duplicates and the
// patterns that exhaust the depth budget included); the voxel-filter parity tests run on top of it.
#include "ctx.hpp"
#include <climits>

namespace mlh {

namespace {

constexpr int SS_WG = 256;
constexpr int SS_THRESHOLD = 16;      // std::_S_threshold

struct SortSeg { int first, last, depth, pad; };

struct StdSortArgs {
    int *keys;          // n: sorted in place (the comparator sees these only)
    int *vals;          // n: carried along
    int *lt, *rt;       // n each: the L / R position tables of a range live at [first, ...) of these
    SortSeg *seg[2];    // ranges longer than 16 of the current / the next depth
    SortSeg *fin;       // ranges of 2..16 elements: the final insertion pass
    int *cnt;           // [0 .. SS_MAX_LEVELS]: ranges per depth; [SS_FIN]: final ranges
    int n;
};
constexpr int SS_MAX_LEVELS = 64;     // 2 * floor(log2(n)) <= 62
constexpr int SS_FIN = SS_MAX_LEVELS + 1;
constexpr int SS_CNT = SS_MAX_LEVELS + 2;

__device__ inline int floor_log2(int n) { return 31 - __clz(n); }

__device__ inline void emit_range(const StdSortArgs &A, int first, int last, int depth, SortSeg *next, int *next_cnt)
{
    const int m = last - first;
    if (m > SS_THRESHOLD) next[atomicAdd(next_cnt, 1)] = SortSeg{first, last, depth, 0};
    else if (m > 1) A.fin[atomicAdd(&A.cnt[SS_FIN], 1)] = SortSeg{first, last, 0, 0};
}

// keys <- the points' slots, vals <- the point indices, one range per cloud (std::sort is called once per cloud)
__global__ __launch_bounds__(256) void stdsort_init_kernel(StdSortArgs A, const int *src_keys, int n0)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < A.n) { A.keys[i] = src_keys[i]; A.vals[i] = i; }
    if (i == 0) {
        for (int k = 1; k < SS_CNT; ++k) A.cnt[k] = 0;
        A.cnt[0] = 0;
        const int lo[2] = {0, n0}, hi[2] = {n0, A.n};
        for (int c = 0; c < 2; ++c)
            if (hi[c] - lo[c] > 1) emit_range(A, lo[c], hi[c], 2 * floor_log2(hi[c] - lo[c]), A.seg[0], &A.cnt[0]);
    }
}

__device__ inline void swap_elem(const StdSortArgs &A, int p, int q)
{
    const int kp = A.keys[p], kq = A.keys[q], vp = A.vals[p], vq = A.vals[q];
    A.keys[p] = kq; A.keys[q] = kp; A.vals[p] = vq; A.vals[q] = vp;
}

// bits/stl_heap.h on (keys, vals) + base: __adjust_heap with its trailing __push_heap, __make_heap, __sort_heap -- one thread, as written
__device__ void heap_adjust(int *k, int *v, int hole, int len, int key, int val)
{
    const int top = hole;
    int c = hole;
    while (c < (len - 1) / 2) {
        c = 2 * (c + 1);
        if (k[c] < k[c - 1]) c--;
        k[hole] = k[c]; v[hole] = v[c];
        hole = c;
    }
    if ((len & 1) == 0 && c == (len - 2) / 2) {
        c = 2 * (c + 1);
        k[hole] = k[c - 1]; v[hole] = v[c - 1];
        hole = c - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && k[parent] < key) {
        k[hole] = k[parent]; v[hole] = v[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    k[hole] = key; v[hole] = val;
}

__device__ void heap_sort_range(int *k, int *v, int len)
{
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            heap_adjust(k, v, parent, len, k[parent], v[parent]);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        const int key = k[last], val = v[last];
        k[last] = k[0]; v[last] = v[0];
        heap_adjust(k, v, 0, last, key, val);
    }
}

// inclusive scan of one int per thread over the workgroup; *total = the sum. LDS: wave_sums[SS_WG / 64]
__device__ inline int block_inclusive_scan(int x, int *wave_sums, int *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int y = __shfl_up(x, off);
        if (lane >= off) x += y;
    }
    __syncthreads();                       // wave_sums may still be read from the previous use
    if (lane == 63) wave_sums[wave] = x;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SS_WG / 64; ++w) { const int s = wave_sums[w]; if (w < wave) base += s; tot += s; }
    *total = tot;
    return x + base;
}

// one recursion depth: every range longer than 16 is partitioned once (or heap-sorted when its depth budget is used up)
__global__ __launch_bounds__(SS_WG) void stdsort_level_kernel(StdSortArgs A, int level)
{
    __shared__ int wave_sums[SS_WG / 64];
    __shared__ int sh_k;
    const SortSeg *cur = A.seg[level & 1];
    SortSeg *next = A.seg[(level + 1) & 1];
    const int count = A.cnt[level];
    const int t = threadIdx.x;
    for (int si = blockIdx.x; si < count; si += gridDim.x) {
        const SortSeg s = cur[si];
        const int f = s.first, l = s.last, m = l - f;
        if (s.depth == 0) {                                          // __partial_sort(first, last, last): sorted for good, no children
            if (t == 0) heap_sort_range(A.keys + f, A.vals + f, m);
            continue;
        }
        if (t == 0) {                                                // __move_median_to_first(first, first + 1, mid, last - 1)
            const int ia = f + 1, ib = f + m / 2, ic = l - 1;
            const int ka = A.keys[ia], kb = A.keys[ib], kc = A.keys[ic];
            int med;
            if (ka < kb) med = (kb < kc) ? ib : ((ka < kc) ? ic : ia);
            else med = (ka < kc) ? ia : ((kb < kc) ? ic : ib);
            swap_elem(A, f, med);
            sh_k = 0;
        }
        __syncthreads();
        const int piv = A.keys[f];
        // thread t owns positions [f + t * per, f + (t + 1) * per) of [f, l); position f is a stop of the right scan only
        const int per = (m + SS_WG - 1) / SS_WG;
        const int p0 = f + t * per, p1 = min(p0 + per, l);
        int cl = 0, cr = 0;
        for (int p = p0; p < p1; ++p) {
            const int k = A.keys[p];
            cl += (p > f && !(k < piv)) ? 1 : 0;
            cr += (p == f || !(piv < k)) ? 1 : 0;
        }
        int nL, nR;
        const int inc_l = block_inclusive_scan(cl, wave_sums, &nL);
        const int inc_r = block_inclusive_scan(cr, wave_sums, &nR);
        int rl = inc_l - cl;                                         // rank of this thread's first left stop
        int rr = nR - inc_r;                                         // rank of this thread's LAST right stop (ranks grow leftwards)
        for (int p = p0; p < p1; ++p) {
            const int k = A.keys[p];
            if (p > f && !(k < piv)) A.lt[f + rl++] = p;
        }
        for (int p = p1 - 1; p >= p0; --p) {
            const int k = A.keys[p];
            if (p == f || !(piv < k)) A.rt[f + rr++] = p;
        }
        __syncthreads();
        // K = number of swaps: L[k] < R[k] holds for a prefix of k
        const int npair = min(nL, nR);
        int mine = 0;
        for (int k = t; k < npair; k += SS_WG) mine += (A.lt[f + k] < A.rt[f + k]) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
        if ((t & 63) == 0 && mine) atomicAdd(&sh_k, mine);
        __syncthreads();
        const int K = sh_k;
        for (int k = t; k < K; k += SS_WG) swap_elem(A, A.lt[f + k], A.rt[f + k]);
        if (t == 0) {
            int cut = INT_MAX;
            if (K < nL) cut = min(cut, A.lt[f + K]);
            if (K > 0) cut = min(cut, A.rt[f + K - 1]);
            emit_range(A, cut, l, s.depth - 1, next, &A.cnt[level + 1]);     // the recursive call
            emit_range(A, f, cut, s.depth - 1, next, &A.cnt[level + 1]);     // the loop's next trip
        }
        __syncthreads();                                             // sh_k and the tables are reused by the next range of this workgroup
    }
}

// __final_insertion_sort restricted to a range of <= 16 (it never moves an element across a range boundary): one thread per range
__global__ __launch_bounds__(256) void stdsort_final_kernel(StdSortArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.cnt[SS_FIN]) return;
    const SortSeg s = A.fin[i];
    int *k = A.keys, *v = A.vals;
    for (int p = s.first + 1; p < s.last; ++p) {
        const int key = k[p], val = v[p];
        int j = p;
        while (j > s.first && key < k[j - 1]) { k[j] = k[j - 1]; v[j] = v[j - 1]; --j; }
        k[j] = key; v[j] = val;
    }
}

}  // namespace

// Sorts (keys = src_keys[0..n), vals = 0..n-1) as two std::sort calls would -- [0, n0) and [n0, n) -- and leaves the permuted vals in
// vals_out (device, n ints). Everything is enqueued on the context's stream; nothing is waited for.
int device_std_sort_by_key(mlh_ctx *ctx, const int *src_keys, int n0, int n, int *vals_out)
{
    if (n <= 0) return MLH_OK;
    hipStream_t st = ctx->stream;
    DevBuf &S = ctx->stdsort;
    const size_t ni = size_t(n), nseg = ni / (SS_THRESHOLD + 1) + 4, nfin = ni / 2 + 4;
    // [keys n][lt n][rt n][cnt SS_CNT (padded to 128)][seg0][seg1][fin]
    const size_t off_lt = ni, off_rt = 2 * ni, off_cnt = 3 * ni, off_seg0 = off_cnt + 128;
    const size_t seg_ints = nseg * 4, off_seg1 = off_seg0 + seg_ints, off_fin = off_seg1 + seg_ints, total = off_fin + nfin * 4;
    MLH_HIP(ctx, S.ensure(sizeof(int) * total));
    int *base = S.as<int>();
    StdSortArgs A;
    A.keys = base; A.vals = vals_out; A.lt = base + off_lt; A.rt = base + off_rt; A.cnt = base + off_cnt;
    A.seg[0] = reinterpret_cast<SortSeg *>(base + off_seg0); A.seg[1] = reinterpret_cast<SortSeg *>(base + off_seg1);
    A.fin = reinterpret_cast<SortSeg *>(base + off_fin); A.n = n;
    hipLaunchKernelGGL(stdsort_init_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A, src_keys, n0);
    int big = std::max(n0, n - n0), lg = 0;
    while ((1 << (lg + 1)) <= big) ++lg;
    const int levels = 2 * lg + 1;                                   // depth budgets 2*lg .. 0
    const int grid = int(std::min<size_t>(nseg, 2048));
    for (int level = 0; level < levels; ++level) hipLaunchKernelGGL(stdsort_level_kernel, dim3(grid), dim3(SS_WG), 0, st, A, level);
    hipLaunchKernelGGL(stdsort_final_kernel, dim3(int((nfin + 255) / 256)), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

}  // namespace mlh
