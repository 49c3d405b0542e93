// Building blocks of libstdc++'s std::sort restated for wavefronts (bits/stl_algo.h, bits/stl_heap.h), shared by stdsort.hip (the voxel
// filters' member order: whole clouds, many workgroups) and extract.hip (extractCloud's per-sector sort, feature_extract.cpp:162: one
// wavefront per sector inside the label kernel). Everything is a template over the comparator, because the permutation std::sort leaves
// among EQUIVALENT elements is a property of the comparison sequence: the voxel filters compare int voxel indices, extractCloud compares
// f32 curvatures (where a NaN compares false against everything -- reproduced, not avoided).
//
// The argument why a Hoare partition can be done by a wavefront at once is in stdsort.hip's header: it is a function of the two stop
// lists of the ORIGINAL arrangement (left-scan stops ascending, right-scan stops descending with `first` as the sentinel), the k-th of
// one swapped with the k-th of the other while they have not crossed. That holds for any comparator under which the sequential scans
// stay inside the range, consistent or not.
#pragma once
#include <hip/hip_runtime.h>
#include <climits>

namespace mlh {

constexpr int SS_THRESHOLD = 16;      // std::_S_threshold
// the wave-level partition of a range longer than one tile (ss_wave_partition): tiles / swaps a lane keeps in flight, the pair count by search (A/B builds)
#ifndef MLH_SS_HEAP_PAR
#define MLH_SS_HEAP_PAR 0
#endif
#ifndef MLH_SS_WAVE_U
#define MLH_SS_WAVE_U 4
#endif
#ifndef MLH_SS_WAVE_SWAP_U
#define MLH_SS_WAVE_SWAP_U 2
#endif
#ifndef MLH_SS_WAVE_SEARCH
#define MLH_SS_WAVE_SEARCH 1
#endif

struct IntLess { __device__ __forceinline__ bool operator()(int a, int b) const { return a < b; } };
// keys are f32 bit patterns compared AS FLOATS (CompObject: cloudCurvature[i] < cloudCurvature[j], feature_extract.hpp)
struct FloatBitsLess { __device__ __forceinline__ bool operator()(int a, int b) const { return __int_as_float(a) < __int_as_float(b); } };

__device__ inline int ss_floor_log2(int n) { return 31 - __clz(n); }
__device__ inline unsigned long long ss_lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }
__device__ inline void ss_wg_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); }

__device__ inline void ss_swap_elem(int *keys, int *vals, int p, int q)
{
    const int kp = keys[p], kq = keys[q], vp = vals[p], vq = vals[q];
    keys[p] = kq; keys[q] = kp; vals[p] = vq; vals[q] = vp;
}

// __move_median_to_first(first, first + 1, mid, last - 1) on the range [f, l)
template <typename Less>
__device__ inline void ss_median_to_first(int *keys, int *vals, int f, int l, Less less)
{
    const int ia = f + 1, ib = f + (l - f) / 2, ic = l - 1;
    const int ka = keys[ia], kb = keys[ib], kc = keys[ic];
    int med;
    if (less(ka, kb)) med = less(kb, kc) ? ib : (less(ka, kc) ? ic : ia);
    else med = less(ka, kc) ? ia : (less(kb, kc) ? ic : ib);
    ss_swap_elem(keys, vals, f, med);
}

// bits/stl_heap.h on (keys, vals): __adjust_heap with its trailing __push_heap, __make_heap, __sort_heap -- one thread, as written
template <typename Less>
__device__ void ss_heap_adjust(int *k, int *v, int hole, int len, int key, int val, Less less)
{
    const int top = hole;
    int c = hole;
    while (c < (len - 1) / 2) {
        c = 2 * (c + 1);
        if (less(k[c], k[c - 1])) c--;
        k[hole] = k[c]; v[hole] = v[c];
        hole = c;
    }
    if ((len & 1) == 0 && c == (len - 2) / 2) {
        c = 2 * (c + 1);
        k[hole] = k[c - 1]; v[hole] = v[c - 1];
        hole = c - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && less(k[parent], key)) {
        k[hole] = k[parent]; v[hole] = v[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    k[hole] = key; v[hole] = val;
}

template <typename Less>
__device__ void ss_heap_sort_range(int *k, int *v, int len, Less less)
{
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            ss_heap_adjust(k, v, parent, len, k[parent], v[parent], less);
            if (parent == 0) break;
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        const int key = k[last], val = v[last];
        k[last] = k[0]; v[last] = v[0];
        ss_heap_adjust(k, v, 0, last, key, val, less);
    }
}

// The same heap sort for a range of at most 64 elements, by a whole wavefront (all lanes converged): element i lives in lane i's registers, the sequential
// algorithm's one moving index is uniform, so every k[i] / v[i] above becomes a v_readlane with a scalar lane select or a one-lane select -- no LDS round trip per
// sift level. Depth-exhausted ranges are what lands here: median-of-three on the piecewise-monotone key sequences of a scan ring runs out of its 2 log2(n)
// budget on one ring in five, the leftovers are 17-50 elements long, and heap-sorting one of them through LDS from a single lane took longer than the rest of
// the ring's sort (the per-ring leaf launch: slowest ring 81 us against a median of 35).
template <typename Less>
__device__ __forceinline__ void ss_heap_sort_wave64(int *k, int *v, int len_, Less less)
{
    const int lane = threadIdx.x & 63, len = __builtin_amdgcn_readfirstlane(len_);
    int key_r = lane < len ? k[lane] : 0, val_r = lane < len ? v[lane] : 0;
    auto rd = [&](int reg, int i) { return __builtin_amdgcn_readlane(reg, i); };
#if MLH_SS_HEAP_PAR
    // __adjust_heap (+ its trailing __push_heap) for ALL levels at once. The library first walks the hole down to a leaf, always to the larger child (the right one
    // on a tie; the only child of the last inner node of an even-sized heap), moving that child up -- a path that depends on the heap's content alone -- and then walks
    // the new value up that same path while the element above it is smaller. With one element per lane: every lane compares its two children (two shuffles), two
    // ballots make "which child does node p prefer" a uniform mask, each lane tests its own ancestors against the mask (is it on the path from the hole, and how
    // deep), a third ballot finds where the upward walk stops -- the deepest path node that is the hole's start or holds an element not smaller than the value --,
    // path nodes above that point take their preferred child's element, the point takes the value. Built, equal to libstdc++ on every test and soak -- and SLOWER than
    // the level-by-level form below (23.8 against 16.8 us per heap sort of 31-94 elements: four ds_bpermute, three ballots and the 64-bit mask tests cost more than six
    // levels of scalar-indexed v_readlane + select): not the default (profiles/r05_knockout_experiments.txt item 19).
    auto adjust = [&](int hole, int n, int key, int val) {
        const int lc = 2 * lane + 1, rc = 2 * lane + 2;
        const int kl = __shfl(key_r, lc & 63), kr = __shfl(key_r, rc & 63), vl = __shfl(val_r, lc & 63), vr = __shfl(val_r, rc & 63);
        const bool two = rc < n, has = lc < n;                      // (lc == n - 1 without a right child: n even, lane == (n - 2) / 2)
        const bool pick_l = two ? less(kr, kl) : true;
        const int kc = pick_l ? kl : kr, vc = pick_l ? vl : vr;     // the preferred child's element
        const unsigned long long HAS = __ballot(has), PL = __ballot(has && pick_l);
        bool on = lane < n;
        int x = lane, depth = 0;
        while (x > hole) {                                          // (at most six trips; lanes outside the hole's subtree fall below it)
            const int p = (x - 1) >> 1;
            on = on && ((HAS >> p) & 1ull) && ((((PL >> p) & 1ull) != 0ull) == ((x & 1) != 0));
            x = p;
            ++depth;
        }
        on = on && x == hole;
        const bool f = on && (depth == 0 || !less(key_r, key));
        const unsigned long long F = __ballot(f);                  // never empty: the hole itself
        const int jstar = 63 - __clzll((long long)F);
        if (on && lane < jstar) { key_r = kc; val_r = vc; }
        else if (lane == jstar) { key_r = key; val_r = val; }
    };
#else
    // every index and every travelling element is wavefront-uniform and is KEPT in scalar registers (readfirstlane where the compiler cannot see it): a sift level
    // is two v_readlane (the children's keys), a scalar compare and two selects (which child, its key -- already read), one v_readlane (its value), two
    // v_writelane into the hole's lane. (Round 6; through round 5 the make-heap phase ran with its indices in vector registers -- v_readfirstlane + wait states
    // in front of every v_readlane, four of them per level, the moves as compare + select: 17.8 us per heap sort of 31-94 elements.)
    // (this toolchain has no writelane builtin; two scalar operands exceed the constant bus, so the lane select goes through M0; the s_nop covers the wait
    // states behind a write of M0 / a v_readfirstlane -- hazards inside inline asm are not the compiler's)
    auto wr = [&](int &reg, int value, int i) { asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(value), "s"(i) : "m0"); };
    auto adjust = [&](int hole_, int n_, int key_, int val_) {
        int hole = __builtin_amdgcn_readfirstlane(hole_);
        const int n = __builtin_amdgcn_readfirstlane(n_), key = __builtin_amdgcn_readfirstlane(key_), val = __builtin_amdgcn_readfirstlane(val_);
        const int top = hole, lim = (n - 1) / 2;
        int c = hole;
        while (c < lim) {
            const int r = 2 * c + 2, l = r - 1;
            const int kr = rd(key_r, r), kl = rd(key_r, l);
            const bool left = less(kr, kl);
            c = left ? l : r;
            const int kc = left ? kl : kr, vc = rd(val_r, c);
            wr(key_r, kc, hole); wr(val_r, vc, hole);
            hole = c;
        }
        if ((n & 1) == 0 && c == (n - 2) / 2) {
            c = 2 * c + 1;
            const int kc = rd(key_r, c), vc = rd(val_r, c);
            wr(key_r, kc, hole); wr(val_r, vc, hole);
            hole = c;
        }
        while (hole > top) {
            const int parent = (hole - 1) / 2;
            const int kp = rd(key_r, parent);
            if (!less(kp, key)) break;
            const int vp = rd(val_r, parent);
            wr(key_r, kp, hole); wr(val_r, vp, hole);
            hole = parent;
        }
        wr(key_r, key, hole); wr(val_r, val, hole);
    };
#endif
    if (len >= 2) {
        int parent = (len - 2) / 2;
        while (true) {
            adjust(parent, len, rd(key_r, parent), rd(val_r, parent));
            if (parent == 0) break;
            parent--;
        }
    }
    int last = len;
    while (last > 1) {
        --last;
        const int key = rd(key_r, last), val = rd(val_r, last);
        const int k0 = rd(key_r, 0), v0 = rd(val_r, 0);
#if MLH_SS_HEAP_PAR
        key_r = (lane == last) ? k0 : key_r; val_r = (lane == last) ? v0 : val_r;
#else
        wr(key_r, k0, last); wr(val_r, v0, last);
#endif
        adjust(0, last, key, val);
    }
    if (lane < len) { k[lane] = key_r; v[lane] = val_r; }
}

// One wavefront streams [lo, hi) of the range [f, l) in 64-wide tiles, SS_U tiles per trip with the loads issued first, ONCE: the positions of its left stops go to
// lt[lo], lt[lo + 1], ... and of its right stops to rt[lo], rt[lo + 1], ..., both in ASCENDING position (a chunk has at most hi - lo stops of either kind, so the
// chunks' sub-tables cannot overlap); n_left / n_right = how many. The k-th right stop counted from the RIGHT end -- the one the partition loop pairs with the
// k-th left stop -- is entry (n_right - 1 - k) of that list; a count pass in front (rounds 2 and 3 had one, to rank the right stops from the right while
// writing) is not needed. Used by the wave-level partitions below and, chunk by chunk, by the 1024-thread partitions of stdsort.hip's big levels.
template <int SS_U, typename Less, typename Tab>      // Tab: int, or unsigned short where the positions are local to a range of < 65 536 elements (stdsort.hip: the mid launch's LDS tables)
__device__ inline void ss_wave_stop_lists(const int *keys, Tab *lt, Tab *rt, int f, int lo, int hi, int piv, int &n_left, int &n_right, Less less)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long below = ss_lanes_below();
    int run_l = 0, run_r = 0;
    for (int base = lo; base < hi; base += 64 * SS_U) {
        int k[SS_U];
#pragma unroll
        for (int u = 0; u < SS_U; ++u) { const int p = base + 64 * u + lane; k[u] = p < hi ? keys[p] : 0; }
#pragma unroll
        for (int u = 0; u < SS_U; ++u) {
            const int p = base + 64 * u + lane;
            const bool in = p < hi;
            const bool is_l = in && p > f && !less(k[u], piv);
            const bool is_r = in && (p == f || !less(piv, k[u]));
            const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
            if (is_l) lt[lo + run_l + __popcll(ml & below)] = Tab(p);
            if (is_r) rt[lo + run_r + __popcll(mr & below)] = Tab(p);
            run_l += __popcll(ml);
            run_r += __popcll(mr);
        }
    }
    n_left = run_l; n_right = run_r;
}

// __unguarded_partition_pivot of the range [f, l), 16 < l - f, by ONE wavefront (all 64 lanes converged); returns the cut.
// size <= 64: the whole range in one tile -- element f + lane in lane's registers, median and pivot through lane shuffles, the two stop tables
// (64 entries each) in `scr` (128 ints of LDS owned by this wavefront), the swaps as one shuffle; otherwise the streaming passes above with the
// tables at [f, ...) of lt / rt. Same pairs, same cut as the sequential loop.
template <typename Less>
__device__ __forceinline__ int ss_wave_partition(int *keys, int *vals, int *lt, int *rt, int *scr, int f, int l, Less less)
{
    const int lane = threadIdx.x & 63;
    const int size = l - f;
    int cut;
    if (size <= 64) {
        int *sl = scr, *sr = scr + 64;
        const bool in = lane < size;
        int key = in ? keys[f + lane] : 0, val = in ? vals[f + lane] : 0;
        const int ia = 1, ib = size / 2, ic = size - 1;
        const int ka = __shfl(key, ia), kb = __shfl(key, ib), kc = __shfl(key, ic);
        int med;
        if (less(ka, kb)) med = less(kb, kc) ? ib : (less(ka, kc) ? ic : ia);
        else med = less(ka, kc) ? ia : (less(kb, kc) ? ic : ib);
        const int k0 = __shfl(key, 0), v0 = __shfl(val, 0), km = __shfl(key, med), vm = __shfl(val, med);
        if (lane == 0) { key = km; val = vm; } else if (lane == med) { key = k0; val = v0; }
        const int piv = km;
        const bool is_l = in && lane > 0 && !less(key, piv);
        const bool is_r = in && (lane == 0 || !less(piv, key));
        const unsigned long long ml = __ballot(is_l), mr = __ballot(is_r);
        const int nL = __popcll(ml), nR = __popcll(mr);
        const int rank_l = __popcll(ml & ss_lanes_below()), rank_r = __popcll((mr >> lane) >> 1);
        if (is_l) sl[rank_l] = lane;
        if (is_r) sr[rank_r] = lane;
        ss_wg_fence();
        const int npair = min(nL, nR);
        const int partner = (is_l && rank_l < npair) ? sr[rank_l] : -1;
        const int K = __popcll(__ballot(lane < partner));          // true for a prefix of the left ranks
        int src = lane;
        if (is_l && rank_l < K) src = partner;
        else if (is_r && rank_r < K) src = sl[rank_r];
        const int nk = __shfl(key, src), nv = __shfl(val, src);
        int c = INT_MAX;
        if (K < nL) c = min(c, sl[K]);
        if (K > 0) c = min(c, sr[K - 1]);
        if (in) { keys[f + lane] = nk; vals[f + lane] = nv; }
        cut = f + c;
        ss_wg_fence();
    } else {
        if (lane == 0) ss_median_to_first(keys, vals, f, l, less);
        ss_wg_fence();
        const int piv = keys[f];
        int nL, nR;
        ss_wave_stop_lists<MLH_SS_WAVE_U>(keys, lt, rt, f, f, l, piv, nL, nR, less);      // left stops ascending at lt[f ..], right stops ascending at rt[f ..]
        ss_wg_fence();
        const int npair = min(nL, nR), rlast = f + nR - 1;                    // the k-th right stop from the right: rt[rlast - k]
        // K = how many pairs cross: L[k] < R[k] holds for a PREFIX of k (the left stops ascend, the right stops counted from the right descend), so a 64-ary search
        // finds it in one or two rounds where testing every pair took npair / 64 (13 for a ring's 1 670 voxel keys)
        int K;
#if MLH_SS_WAVE_SEARCH
        {
            int lo_k = 0, hi_k = npair;
            while (hi_k > lo_k) {                                            // uniform
                const int step = (hi_k - lo_k + 63) >> 6;
                const int k = lo_k + lane * step;
                const bool p = k < hi_k && lt[f + k] < rt[rlast - k];
                const int c = __popcll(__ballot(p));
                const int new_hi = min(hi_k, lo_k + c * step);
                lo_k = c > 0 ? lo_k + (c - 1) * step + 1 : lo_k;
                hi_k = c > 0 ? max(new_hi, lo_k) : lo_k;
            }
            K = lo_k;
        }
#else
        K = 0;
        for (int base = 0; base < npair; base += 64) {
            const int k = base + lane;
            K += __popcll(__ballot(k < npair && lt[f + k] < rt[rlast - k]));
        }
#endif
        // the swaps, MLH_SS_WAVE_SWAP_U pairs per lane in flight: positions, then the elements, then the stores (the pairs are disjoint)
        for (int k0 = lane; k0 < K; k0 += 64 * MLH_SS_WAVE_SWAP_U) {
            int pp[MLH_SS_WAVE_SWAP_U], qq[MLH_SS_WAVE_SWAP_U], kp[MLH_SS_WAVE_SWAP_U], kq[MLH_SS_WAVE_SWAP_U], vp[MLH_SS_WAVE_SWAP_U], vq[MLH_SS_WAVE_SWAP_U];
#pragma unroll
            for (int u = 0; u < MLH_SS_WAVE_SWAP_U; ++u) { const int k = k0 + 64 * u; const bool on = k < K; pp[u] = on ? lt[f + k] : -1; qq[u] = on ? rt[rlast - k] : -1; }
#pragma unroll
            for (int u = 0; u < MLH_SS_WAVE_SWAP_U; ++u) if (pp[u] >= 0) { kp[u] = keys[pp[u]]; kq[u] = keys[qq[u]]; vp[u] = vals[pp[u]]; vq[u] = vals[qq[u]]; }
#pragma unroll
            for (int u = 0; u < MLH_SS_WAVE_SWAP_U; ++u) if (pp[u] >= 0) { keys[pp[u]] = kq[u]; keys[qq[u]] = kp[u]; vals[pp[u]] = vq[u]; vals[qq[u]] = vp[u]; }
        }
        cut = INT_MAX;
        if (K < nL) cut = min(cut, lt[f + K]);
        if (K > 0) cut = min(cut, rt[rlast - (K - 1)]);
        ss_wg_fence();
    }
    return cut;
}

// std::sort(keys, keys + n, less) with `vals` carried along, by ONE wavefront (all 64 lanes converged, n >= 0).
//   keys, vals, lt, rt : n ints each (LDS or global);  stk : 3 * (n / 17 + 2) ints -- the ranges the recursion still owes;
//   bits : (n + 31) / 32 + 1 words -- one bit per position, set where a range the loop has finished with starts;  scr : 128 ints of LDS.
// The introsort loop runs range by range (the ranges are disjoint, so their order does not matter): partition, push the right part (the
// library's recursive call) if it is longer than 16, go on with the left part (the loop's next trip), both with the depth budget minus one;
// a range whose budget is used up is heap-sorted by one lane. What the loop leaves is finished by the insertion pass:
//   consistent (a strict weak order on these keys: no NaN among them): an element never moves across the bounds of the ranges the loop
//     left, and the pass is stable inside one -- every range is insertion-sorted by a lane of its own;
//   otherwise: the library's own sequence, by one lane -- __insertion_sort on the first 16, __unguarded_insertion_sort on the rest (with NaNs
//     in play elements DO cross those bounds). The unguarded scan is given a guard at position 0, where the library would walk out of the range.
template <typename Less>
__device__ __forceinline__ void ss_wave_std_sort(int *keys, int *vals, int *lt, int *rt, int *stk, unsigned *bits, int *scr, int n, bool consistent, Less less)
{
    const int lane = threadIdx.x & 63;
    if (n < 2) return;
    for (int w = lane; w < (n + 31) / 32 + 1; w += 64) bits[w] = 0u;
    ss_wg_fence();
    int f = 0, l = n, d = 2 * ss_floor_log2(n), top = 0;
    bool have = true;
    while (true) {
        if (!have) {
            if (top == 0) break;
            --top;
            f = stk[3 * top]; l = stk[3 * top + 1]; d = stk[3 * top + 2];
            have = true;
        }
        if (l - f <= SS_THRESHOLD) {                                   // finished by the loop: a range of the insertion pass
            if (lane == 0 && l > f) atomicOr(&bits[f >> 5], 1u << (f & 31));
            have = false;
            continue;
        }
        if (d == 0) {                                                  // __partial_sort(first, last, last): sorted for good
            if (l - f <= 64) ss_heap_sort_wave64(keys + f, vals + f, l - f, less);
            else if (lane == 0) ss_heap_sort_range(keys + f, vals + f, l - f, less);
            if (lane == 0) atomicOr(&bits[f >> 5], 1u << (f & 31));
            ss_wg_fence();
            have = false;
            continue;
        }
        const int cut = ss_wave_partition(keys, vals, lt, rt, scr, f, l, less);
        --d;
        if (l - cut > SS_THRESHOLD) {
            if (lane == 0) { stk[3 * top] = cut; stk[3 * top + 1] = l; stk[3 * top + 2] = d; }
            ++top;
            ss_wg_fence();
        } else if (lane == 0 && l > cut) atomicOr(&bits[cut >> 5], 1u << (cut & 31));
        l = cut;
    }
    ss_wg_fence();
    if (consistent) {
        for (int p = lane; p < n; p += 64) {
            if (!((bits[p >> 5] >> (p & 31)) & 1u)) continue;
            int e = p + 1;
            while (e < n && !((bits[e >> 5] >> (e & 31)) & 1u)) ++e;
            for (int q = p + 1; q < e; ++q) {
                const int key = keys[q], val = vals[q];
                int j = q;
                while (j > p && less(key, keys[j - 1])) { keys[j] = keys[j - 1]; vals[j] = vals[j - 1]; --j; }
                keys[j] = key; vals[j] = val;
            }
        }
    } else if (lane == 0) {
        const int head = min(n, SS_THRESHOLD);
        for (int q = 1; q < n; ++q) {
            const int key = keys[q], val = vals[q];
            int j = q;
            if (q < head && less(key, keys[0])) {                      // __insertion_sort: move_backward + put in front
                for (; j > 0; --j) { keys[j] = keys[j - 1]; vals[j] = vals[j - 1]; }
            } else {
                while (j > 0 && less(key, keys[j - 1])) { keys[j] = keys[j - 1]; vals[j] = vals[j - 1]; --j; }
            }
            keys[j] = key; vals[j] = val;
        }
    }
    ss_wg_fence();
}

}  // namespace mlh
