// LOAM feature extraction on gfx950: FeatureExtract::extractCloud
// (estimator/src/featureExtract/feature_extract.cpp:118-297).
//
//   curvature_kernel  cpp:133-142  one thread per point, neighbours staged in LDS (21 B/point algorithmic HBM traffic);
//                     f32 sum in the reference's exact left-to-right order, no FMA (-ffp-contract=off)
//   label_kernel      cpp:152-265  one workgroup (six wavefronts) per ring: ring xyz + curvature + picked flags resident in LDS;
//                     wavefront j sorts sector j in registers with a wave-local bitonic network on 64-bit (curvature, index)
//                     keys (lane-xor shuffles, no barriers); then the greedy edge / flat walks: 64 sorted candidates are tested
//                     per batch with a ballot, the first eligible one is taken, its +-5 neighbour suppression extent comes from a
//                     table built from gap bits, and the batch's other candidates are retired in registers. The reference walks
//                     the sectors in order because suppression marks cross sector bounds (cpp:201, 212); here the six sectors
//                     walk speculatively at once and a sector is re-walked only when a predecessor's marks hit one of its picks.
//   emit_kernel       writes the four index lists (emission order = ring asc, sector asc, pick order; every ring's workgroup sums
//                     the list sizes of the rings before it); less-flat = positions with label <= 0 (cpp:258-264), stream-compacted.
#include "ctx.hpp"
#include "sort_dev.hpp"
#include "stdsort_dev.hpp"
#include <algorithm>

namespace mlh {

#ifdef MLH_STAGE_CLOCK
__device__ unsigned long long g_stage_clk_label[256 * 8];
#define MLH_LSTAGE(i)                                                                        \
    do {                                                                                     \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                        \
        if (threadIdx.x == 0 && blockIdx.x < 256) g_stage_clk_label[blockIdx.x * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define MLH_LSTAGE(i) do { } while (0)
#endif

__global__ __launch_bounds__(256) void curvature_kernel(const float4 *__restrict__ pts, int n, float *__restrict__ curv,
                                                        int *__restrict__ label, int *__restrict__ picked)
{
    __shared__ float sx[256 + 10], sy[256 + 10], sz[256 + 10];
    const int base = blockIdx.x * 256;
    for (int t = threadIdx.x; t < 266; t += 256) {
        int gi = base - 5 + t;
        float4 p = (gi >= 0 && gi < n) ? pts[gi] : make_float4(0.f, 0.f, 0.f, 0.f);
        sx[t] = p.x; sy[t] = p.y; sz[t] = p.z;
    }
    __syncthreads();
    const int i = base + threadIdx.x;
    if (i >= n) return;
    float c = 0.f;
    if (i >= 5 && i < n - 5) {
        const int t = threadIdx.x + 5;
        float dx = sx[t - 5] + sx[t - 4] + sx[t - 3] + sx[t - 2] + sx[t - 1] - 10 * sx[t] + sx[t + 1] + sx[t + 2] + sx[t + 3] + sx[t + 4] + sx[t + 5];
        float dy = sy[t - 5] + sy[t - 4] + sy[t - 3] + sy[t - 2] + sy[t - 1] - 10 * sy[t] + sy[t + 1] + sy[t + 2] + sy[t + 3] + sy[t + 4] + sy[t + 5];
        float dz = sz[t - 5] + sz[t - 4] + sz[t - 3] + sz[t - 2] + sz[t - 1] - 10 * sz[t] + sz[t + 1] + sz[t + 2] + sz[t + 3] + sz[t + 4] + sz[t + 5];
        c = dx * dx + dy * dy + dz * dz;
    }
    curv[i] = c;
    label[i] = 0;
    picked[i] = 0;
}

constexpr int STAGE_SHARP = 12, STAGE_LESS = 120, STAGE_FLAT = 24;
constexpr int STAGE_STRIDE = STAGE_SHARP + STAGE_LESS + STAGE_FLAT;

struct LabelArgs {
    const float4 *pts;
    const float *curv;
    const int *start, *end;
    int *label, *picked;
    int *stage;         // [ring][STAGE_STRIDE]
    int *ring_counts;   // [ring][4]
    int n, max_span, sort_p;
    int tie_ref;        // 1: a sector whose curvatures are not all distinct is ordered as the reference's std::sort call leaves it (feature_extract.cpp:162)
    int tie_words;      // ints of std::sort scratch per sector: 5 * sort_p + sort_p / 32 + 2 + 128
    int *tie_scratch;   // [ring][6][tie_words] in HBM when the scratch does not fit the LDS budget (4000-column rings), else null
};

// gap test of the suppression loops (cpp:192-213): squared distance between consecutive points > 0.05 (double literal)
__device__ __forceinline__ bool gap_exceeds(const float *sx, const float *sy, const float *sz, int a, int b)
{
    float dx = sx[a] - sx[b], dy = sy[a] - sy[b], dz = sz[a] - sz[b];
    return double(dx * dx + dy * dy + dz * dz) > 0.05;
}

constexpr int LTPB = 384;   // label kernel: six wavefronts, one per sector while sorting

// one sector: build the keys in registers, sort, store the sorted list to LDS for the walks
template <int KPL>
__device__ __forceinline__ void sort_sector(const float *sc, int sp, int len, unsigned long long *kj, int lane)
{
    unsigned long long v[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int k = lane * KPL + r;
        unsigned long long key = ~0ull;                          // padding sorts to the end
        if (k < len) {
            const int li = sp + k;
            const float c = sc[li];
            // (curvature, index) ascending; a NaN curvature (non-finite input point) takes one canonical pattern above every number, so
            // NaNs order by index among themselves whatever payload the arithmetic left in them
            key = ((unsigned long long)((c != c) ? 0x7fc00000u : __float_as_uint(c)) << 32) | (unsigned)li;
        }
        v[r] = key;
    }
    wave_bitonic_sort<KPL>(v, lane);
#pragma unroll
    for (int r = 0; r < KPL; ++r) kj[lane * KPL + r] = v[r];
}

// The reference sorts a sector's indices with std::sort and a comparator that sees the curvature only (feature_extract.cpp:152-162,
// feature_extract.hpp: compObject): among EQUAL curvatures the order is whatever libstdc++'s introsort leaves, and the greedy walks consume
// that order. A sector whose curvatures are all distinct has one sorted order, which the bitonic network above has produced; only when the
// sorted keys show two equal curvatures (or a NaN, which compares false against everything and makes the comparator inconsistent) is the
// library's algorithm run on the sector's ORIGINAL arrangement (indices ascending) by this wavefront -- stdsort_dev.hpp: same comparison
// sequence, same permutation -- and its result replaces the list. Returns true when the list may not be sorted any more (NaN present: the
// walks then must not stop at the first candidate that fails the curvature test, exactly as the reference's loops do not).
__device__ __forceinline__ bool reference_tie_order(const float *sc, int sp, int len, unsigned long long *kj, int *tie, int P, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    bool tie_here = false, nan_here = false;
    for (int k = lane; k < len; k += 64) {
        const unsigned hi = unsigned(kj[k] >> 32);
        if (k + 1 < len && hi == unsigned(kj[k + 1] >> 32)) tie_here = true;
        if (hi == 0x7fc00000u) nan_here = true;
    }
    const bool has_nan = __ballot(nan_here) != 0ull;
    if (__ballot(tie_here) == 0ull && !has_nan) return false;
    int *keys = tie, *vals = tie + P, *lt = tie + 2 * P, *rt = tie + 3 * P, *stk = tie + 4 * P;
    unsigned *bits = reinterpret_cast<unsigned *>(tie + 5 * P);
    int *scr = tie + 5 * P + P / 32 + 2;                      // the stop tables of a partition that fits one 64-wide tile
    for (int k = lane; k < len; k += 64) { keys[k] = __float_as_int(sc[sp + k]); vals[k] = sp + k; }
    ss_wg_fence();
    ss_wave_std_sort(keys, vals, lt, rt, stk, bits, scr, len, !has_nan, FloatBitsLess());
    for (int k = lane; k < len; k += 64) kj[k] = ((unsigned long long)(unsigned)keys[k] << 32) | (unsigned)vals[k];
    ss_wg_fence();
    return has_nan;
}

// Suppression of a pick's +-5 neighbours (cpp:192-213, 233-254): how far the marks reach (nf forwards, nb backwards: up to the first
// consecutive-point gap > 0.05, at most 5) depends only on the geometry, so it is tabulated for every position of the ring before
// the walks (sext[t] = nf | nb << 4); a candidate lane brings its entry along and a pick costs no LDS read at all.

// The two greedy walks of ONE sector by one wavefront (cpp:166-256). Per pick the loop does register work plus one predicated LDS
// store (the suppression marks); the picks themselves are parked in lane registers (lane p keeps the p-th pick) and are labelled /
// staged by the caller. Marks on positions inside [lo, hi] go to spicked; marks that reach past those bounds (at most 5 positions
// on either side) go to spill[0..4] (hi+1 .. hi+5) and spill[5..9] (lo-1 .. lo-5) -- the caller decides when they take effect.
struct SectorPicks { int npick, my_pick, nfl, my_flat; };
__device__ __forceinline__ SectorPicks walk_sector(const unsigned long long *kj, int len, const float *sc, const unsigned char *sext,
                                                   int *spicked, int lo, int hi, unsigned char *spill, int lane, bool sorted)
{
    const int off = lane - 5;                                  // lanes 0..10 cover the positions sel-5 .. sel+5
    SectorPicks R;
    // ---- edge walk, descending curvature (cpp:166-215): picks 1-2 sharp, 3-20 less sharp, the 21st candidate ends it unlabelled
    int npick = 0, my_pick = 0;
    bool stop = false;
    for (int base = len - 1; base >= 0 && !stop; base -= 64) {
        const int k = base - lane;
        const int li = (k >= 0) ? int(unsigned(kj[k])) : 0;
        const bool c_ok = (k >= 0) && (double(sc[li]) > 0.1);
        const int ext = sext[li];
        bool elig = c_ok && (spicked[li] == 0);            // one LDS look per batch; picks inside the batch retire lanes in registers
        unsigned long long m = __ballot(elig);
        while (m) {
            if (npick == 20) { stop = true; break; }
            const int l = __ffsll((long long)m) - 1;
            const int sel = __builtin_amdgcn_readlane(li, l);
            const int xe = __builtin_amdgcn_readlane(ext, l);
            const int nf = xe & 15, nb = xe >> 4;
            my_pick = (lane == npick) ? sel : my_pick;
            npick++;
            if (lane < 11 && off >= -nb && off <= nf) {        // the pick and its neighbours up to the first gap
                const int x = sel + off;
                if (x > hi) spill[x - hi - 1] = 1;
                else if (x < lo) spill[5 + (lo - 1 - x)] = 1;
                else spicked[x] = 1;
            }
            elig = elig && !(li >= sel - nb && li <= sel + nf);
            m = __ballot(elig);
        }
        __builtin_amdgcn_wave_barrier();
        // sorted descending: once a NUMBER fails c > 0.1 every later candidate fails too. NaN curvatures (non-finite input points) sit
        // at the top of the order and fail the test without saying anything about what follows: they are walked past, as the reference's
        // loop walks past every candidate that does not qualify (cpp:166-215 has no early exit but the 21st pick)
        const float cv = sc[li];
        if (sorted && __ballot((k >= 0) && !c_ok && !(cv != cv)) != 0ull) stop = true;
    }
    R.npick = npick; R.my_pick = my_pick;
    // ---- flat walk, ascending curvature (cpp:219-256): 4 picks; the 4th is labelled but neither marked nor suppressing
    int nfl = 0, my_flat = 0;
    stop = false;
    for (int base = 0; base < len && !stop; base += 64) {
        const int k = base + lane;
        const int li = (k < len) ? int(unsigned(kj[k])) : 0;
        const bool c_ok = (k < len) && (double(sc[li]) < 0.1);
        const int ext = sext[li];
        bool elig = c_ok && (spicked[li] == 0);
        unsigned long long m = __ballot(elig);
        while (m) {
            const int l = __ffsll((long long)m) - 1;
            const int sel = __builtin_amdgcn_readlane(li, l);
            my_flat = (lane == nfl) ? sel : my_flat;
            nfl++;
            if (nfl >= 4) { stop = true; break; }
            const int xe = __builtin_amdgcn_readlane(ext, l);
            const int nf = xe & 15, nb = xe >> 4;
            if (lane < 11 && off >= -nb && off <= nf) {
                const int x = sel + off;
                if (x > hi) spill[x - hi - 1] = 1;
                else if (x < lo) spill[5 + (lo - 1 - x)] = 1;
                else spicked[x] = 1;
            }
            elig = elig && !(li >= sel - nb && li <= sel + nf);
            m = __ballot(elig);
        }
        __builtin_amdgcn_wave_barrier();
        const unsigned long long inb = __ballot(k < len);
        if (sorted && __ballot(c_ok) != inb) stop = true;
    }
    R.nfl = nfl; R.my_flat = my_flat;
    return R;
}

__global__ __launch_bounds__(LTPB) void label_kernel(LabelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int ring = blockIdx.x;
    const int s = A.start[ring], e = A.end[ring];
    int *rc = A.ring_counts + ring * 4;
    if (e - s < 6 || s < 5 || e + 5 > A.n) {   // cpp:155 (and a guard against ring tables that are not inset)
        if (threadIdx.x < 4) rc[threadIdx.x] = 0;
        return;
    }
    MLH_LSTAGE(0);
    const int span = e - s + 11;           // [s-5, e+5]
    const int g0 = s - 5;                  // global index of local 0
    const int P = A.sort_p;
    float *sx = reinterpret_cast<float *>(smem);
    float *sy = sx + A.max_span;
    float *sz = sy + A.max_span;
    float *sc = sz + A.max_span;
    int *spicked = reinterpret_cast<int *>(sc + A.max_span);
    int *slabel = spicked + A.max_span;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(slabel + A.max_span + ((A.max_span & 1) ? 1 : 0));
    unsigned *sgap = reinterpret_cast<unsigned *>(keys + 6 * size_t(P));     // gap bits, (max_span + 64 + 383) / 32 + 4 words
    unsigned char *sext = reinterpret_cast<unsigned char *>(sgap + (A.max_span + 64 + 383) / 32 + 4);   // suppression extents, max_span bytes
    int *s_tie = reinterpret_cast<int *>(sext + ((A.max_span + 15) & ~15));      // std::sort scratch of the six sectors (when it fits: tie_scratch == null)
    __shared__ int s_stage[STAGE_STRIDE];
    __shared__ int s_cnt[4];
    __shared__ int s_unsorted[6];

    for (int t = threadIdx.x; t < span; t += LTPB) {
        float4 p = A.pts[g0 + t];
        sx[t] = p.x; sy[t] = p.y; sz[t] = p.z;
        sc[t] = A.curv[g0 + t];
        spicked[t] = 0;
        slabel[t] = 0;
    }
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    // sector bounds (cpp:160-161), local coordinates (local = global - g0)
    int sp[6], ep[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        sp[j] = s + (e - s) * j / 6 - g0;
        ep[j] = s + (e - s) * (j + 1) / 6 - 1 - g0;
    }
    __syncthreads();
    MLH_LSTAGE(1);
    // gap bits: bit t = squared distance between local t and t + 1 exceeds 0.05 (cpp:196, 206, 237, 247; the literal is a double)
    for (int base = 0; base < span + 64; base += LTPB) {      // uniform trip count: every wavefront ballots 64 positions per trip
        const int t = base + threadIdx.x;
        const bool gap = (t + 1 < span) && gap_exceeds(sx, sy, sz, t + 1, t);
        const unsigned long long m = __ballot(gap);
        if ((threadIdx.x & 63) == 0) { sgap[t >> 5] = unsigned(m); sgap[(t >> 5) + 1] = unsigned(m >> 32); }
    }
    __syncthreads();
    // suppression extents of every position that can be picked (local 5 .. span-6)
    for (int t = threadIdx.x; t < span; t += LTPB) {
        unsigned char ext = 0;
        if (t >= 5 && t + 5 < span) {
            const int b0 = t - 5;                                 // window: gap bits [t-5, t+4]
            const unsigned long long w = (unsigned long long)sgap[b0 >> 5] | ((unsigned long long)sgap[(b0 >> 5) + 1] << 32);
            const unsigned bits = unsigned(w >> (b0 & 31)) & 0x3ffu;
            const unsigned fwd = (bits >> 5) & 31u, bwd = bits & 31u;   // fwd bit l-1: gap(t+l, t+l-1); bwd bit 5-l: gap(t-l, t-l+1)
            const int nf = fwd ? (__ffs(fwd) - 1) : 5;
            const int nb = bwd ? (4 - (31 - __clz(bwd))) : 5;
            ext = (unsigned char)(nf | (nb << 4));
        }
        sext[t] = ext;
    }
    MLH_LSTAGE(2);
    // sort: wavefront j sorts sector j in registers (keys: (curvature bits << 32) | local index), result to keys[j * P ..]
    {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int len = ep[wave] - sp[wave] + 1;
        unsigned long long *kj = keys + wave * P;
        switch (P >> 6) {
        case 1: sort_sector<1>(sc, sp[wave], len, kj, lane); break;
        case 2: sort_sector<2>(sc, sp[wave], len, kj, lane); break;
        case 4: sort_sector<4>(sc, sp[wave], len, kj, lane); break;
        case 8: sort_sector<8>(sc, sp[wave], len, kj, lane); break;
        case 16: sort_sector<16>(sc, sp[wave], len, kj, lane); break;
        default: sort_sector<32>(sc, sp[wave], len, kj, lane); break;
        }
        bool unsorted = false;
        if (A.tie_ref) {
            int *tie = A.tie_scratch ? A.tie_scratch + (size_t(ring) * 6 + wave) * A.tie_words : s_tie + wave * A.tie_words;
            unsorted = reference_tie_order(sc, sp[wave], len, kj, tie, P, lane);
        }
        if (lane == 0) s_unsorted[wave] = unsorted ? 1 : 0;
    }
    __syncthreads();
    MLH_LSTAGE(3);
    __shared__ unsigned char s_spill[6 * 10 + 4];
    __shared__ int s_np[6], s_nf[6];
    if (threadIdx.x < 64) s_spill[threadIdx.x] = 0;
    const bool sectors_in_parallel = (e - s) >= 30 && LTPB >= 6 * 64;   // every sector >= 5 points: a pick's +-5 marks reach at most into the adjacent sector
    if (sectors_in_parallel) {
        // The reference walks the sectors one after the other, and the only thing sector j+1 sees of sector j is the picked-marks
        // that j's last picks leave on j+1's first (at most 5) positions. So all six sectors walk SPECULATIVELY at once, one wavefront
        // each, with marks that cross a sector bound set aside (s_spill); then, in sector order, a sector whose speculative picks
        // include a position its predecessor really marked is walked again with those marks in place (its own spill changes with
        // it, so its successor is checked against the final one). Marks on positions that were not picked change nothing: they can only
        // remove candidates, and a candidate that was not picked either was never reached or was already suppressed.
        __syncthreads();                                   // s_spill cleared
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        const int lo = sp[wave], hi = ep[wave];
        const unsigned long long *kj = keys + wave * P;
        unsigned char *spill = s_spill + wave * 10;
        SectorPicks pk = walk_sector(kj, hi - lo + 1, sc, sext, spicked, lo, hi, spill, lane, !s_unsorted[wave]);
        for (int round = 1; round < 6; ++round) {
            __syncthreads();                               // sector round-1 is final
            if (wave == round) {
                int inc = 0;
#pragma unroll
                for (int i = 0; i < 5; ++i) inc |= s_spill[(round - 1) * 10 + i] ? (1 << i) : 0;
                const int de = pk.my_pick - lo, df = pk.my_flat - lo;
                const bool hit = (lane < pk.npick && de < 5 && ((inc >> de) & 1)) || (lane < pk.nfl && df < 5 && ((inc >> df) & 1));
                if (__ballot(hit)) {
                    for (int x = lo + lane; x <= hi; x += 64) spicked[x] = 0;
                    if (lane < 10) spill[lane] = 0;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    if (lane < 5 && ((inc >> lane) & 1)) spicked[lo + lane] = 1;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    pk = walk_sector(kj, hi - lo + 1, sc, sext, spicked, lo, hi, spill, lane, !s_unsorted[wave]);
                }
            }
        }
        if (lane == 0) { s_np[wave] = pk.npick; s_nf[wave] = pk.nfl; }
        __syncthreads();
        int n_sharp = 0, n_less = 0, n_flat = 0;            // picks of the sectors before this one (output order: sector, then pick)
        for (int j = 0; j < wave; ++j) { n_sharp += min(s_np[j], 2); n_less += s_np[j]; n_flat += s_nf[j]; }
        if (lane < pk.npick) {
            slabel[pk.my_pick] = (lane < 2) ? 2 : 1;
            if (lane < 2) s_stage[n_sharp + lane] = pk.my_pick + g0;
            s_stage[STAGE_SHARP + n_less + lane] = pk.my_pick + g0;
        }
        if (lane < pk.nfl) {
            slabel[pk.my_flat] = -1;
            s_stage[STAGE_SHARP + STAGE_LESS + n_flat + lane] = pk.my_flat + g0;
        }
        // the marks that crossed a sector bound now take effect for the picked[] output
        if (lane < 10 && spill[lane]) spicked[lane < 5 ? hi + 1 + lane : lo - 1 - (lane - 5)] = 1;
        if (wave == 5 && lane == 0) { s_cnt[0] = n_sharp + min(pk.npick, 2); s_cnt[1] = n_less + pk.npick; s_cnt[2] = n_flat + pk.nfl; }
        MLH_LSTAGE(4);
    } else if (threadIdx.x < 64) {
        // short rings: wave 0 walks the sectors in order, every mark taking effect at once
        const int lane = threadIdx.x;
        int n_sharp = 0, n_less = 0, n_flat = 0;
        for (int j = 0; j < 6; ++j) {
            const SectorPicks pk = walk_sector(keys + j * P, ep[j] - sp[j] + 1, sc, sext, spicked, 0, span - 1, s_spill, lane, !s_unsorted[j]);
            if (lane < pk.npick) {
                slabel[pk.my_pick] = (lane < 2) ? 2 : 1;
                if (lane < 2) s_stage[n_sharp + lane] = pk.my_pick + g0;
                s_stage[STAGE_SHARP + n_less + lane] = pk.my_pick + g0;
            }
            n_sharp += pk.npick < 2 ? pk.npick : 2;
            n_less += pk.npick;
            if (lane < pk.nfl) {
                slabel[pk.my_flat] = -1;
                s_stage[STAGE_SHARP + STAGE_LESS + n_flat + lane] = pk.my_flat + g0;
            }
            n_flat += pk.nfl;
        }
        if (lane == 0) { s_cnt[0] = n_sharp; s_cnt[1] = n_less; s_cnt[2] = n_flat; }
        MLH_LSTAGE(4);
    }
    __syncthreads();
    // write back labels / picked, count less-flat (label <= 0 over [s, e-1])
    int my_lf = 0;
    for (int t = threadIdx.x; t < span; t += LTPB) {
        const int gi = g0 + t;
        if (spicked[t]) A.picked[gi] = 1;
        if (gi >= s && gi <= e - 1) {
            const int lb = slabel[t];
            if (lb != 0) A.label[gi] = lb;
            if (lb <= 0) my_lf++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_lf += __shfl_xor(my_lf, off);
    if ((threadIdx.x & 63) == 0 && my_lf) atomicAdd(&s_cnt[3], my_lf);
    for (int t = threadIdx.x; t < STAGE_STRIDE; t += LTPB) A.stage[ring * STAGE_STRIDE + t] = s_stage[t];
    __syncthreads();
    if (threadIdx.x < 4) rc[threadIdx.x] = s_cnt[threadIdx.x];
    MLH_LSTAGE(5);
}

#ifdef MLH_STAGE_CLOCK
}  // namespace mlh
extern "C" int mlh_debug_stage_clock_label(unsigned long long *out, int n_words)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(mlh::g_stage_clk_label), sizeof(unsigned long long) * size_t(n_words));
}
namespace mlh {
#endif

struct EmitArgs {
    const int *start, *end, *label, *stage, *ring_counts;
    int *ring_offsets, *totals;       // written here: [n_rings + 1][4] exclusive offsets (last row = totals) and the 4 totals
    int n_rings;
    int *list0, *list1, *list2, *list3;
};

__global__ __launch_bounds__(256) void emit_kernel(EmitArgs A)
{
    __shared__ int s_w[4];
    __shared__ int s_base;
    __shared__ int s_off[4];
    const int ring = blockIdx.x;
    // the ring's offsets in the four output lists = the counts of the rings before it, summed right here by one wavefront (a few
    // dozen words): no separate scan launch between the label kernel and this one
    if (threadIdx.x < 64) {
        int acc[4] = {0, 0, 0, 0};
        for (int r = threadIdx.x; r < ring; r += 64) {
            const int4 c = reinterpret_cast<const int4 *>(A.ring_counts)[r];
            acc[0] += c.x; acc[1] += c.y; acc[2] += c.z; acc[3] += c.w;
        }
#pragma unroll
        for (int l = 0; l < 4; ++l) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) acc[l] += __shfl_xor(acc[l], off);
        }
        if (threadIdx.x < 4) {
            const int v = threadIdx.x == 0 ? acc[0] : (threadIdx.x == 1 ? acc[1] : (threadIdx.x == 2 ? acc[2] : acc[3]));
            s_off[threadIdx.x] = v;
            A.ring_offsets[ring * 4 + threadIdx.x] = v;
            if (ring == A.n_rings - 1) {
                const int tot = v + A.ring_counts[ring * 4 + threadIdx.x];
                A.ring_offsets[A.n_rings * 4 + threadIdx.x] = tot;
                A.totals[threadIdx.x] = tot;
            }
        }
    }
    __syncthreads();
    const int *rc = A.ring_counts + ring * 4, *ro = s_off;
    const int *st = A.stage + ring * STAGE_STRIDE;
    for (int t = threadIdx.x; t < rc[0]; t += 256) A.list0[ro[0] + t] = st[t];
    for (int t = threadIdx.x; t < rc[1]; t += 256) A.list1[ro[1] + t] = st[STAGE_SHARP + t];
    for (int t = threadIdx.x; t < rc[2]; t += 256) A.list2[ro[2] + t] = st[STAGE_SHARP + STAGE_LESS + t];
    if (rc[3] == 0) return;
    const int s = A.start[ring], e = A.end[ring];
    if (threadIdx.x == 0) s_base = ro[3];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = s; c0 <= e - 1; c0 += 256) {
        const int gi = c0 + threadIdx.x;
        const bool take = (gi <= e - 1) && (A.label[gi] <= 0);
        const unsigned long long m = __ballot(take);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_w[w];
        const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (take) A.list3[s_base + wbase + before] = gi;
        __syncthreads();
        if (threadIdx.x == 0) s_base += total;
        __syncthreads();
    }
}

int extract_run(mlh_ctx *ctx)
{
    ScanBuf &sb = ctx->scan;
    if (sb.n <= 0 || sb.n_rings <= 0) return fail(ctx, MLH_ERR_STATE, "scan_upload has not been called");
    hipStream_t st = ctx->stream;
    const int n = sb.n, R = sb.n_rings;
    MLH_HIP(ctx, sb.curvature.ensure(sizeof(float) * size_t(n)));
    MLH_HIP(ctx, sb.label.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, sb.picked.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, sb.stage.ensure(sizeof(int) * STAGE_STRIDE * size_t(R)));
    MLH_HIP(ctx, sb.ring_counts.ensure(sizeof(int) * 4 * size_t(R)));
    MLH_HIP(ctx, sb.ring_offsets.ensure(sizeof(int) * 4 * size_t(R + 1)));
    MLH_HIP(ctx, sb.totals.ensure(sizeof(int) * 4));
    MLH_HIP(ctx, sb.lists[0].ensure(sizeof(int) * STAGE_SHARP * size_t(R)));
    MLH_HIP(ctx, sb.lists[1].ensure(sizeof(int) * STAGE_LESS * size_t(R)));
    MLH_HIP(ctx, sb.lists[2].ensure(sizeof(int) * STAGE_FLAT * size_t(R)));
    MLH_HIP(ctx, sb.lists[3].ensure(sizeof(int) * size_t(n)));

    // LDS budget of the label kernel from the widest ring
    const int max_span = ((sb.max_ring_len + 11 + 3) / 4) * 4;
    int max_sector = (sb.max_ring_len + 5) / 6 + 1;
    int P = 64;
    while (P < max_sector) P <<= 1;
    size_t lds = sizeof(float) * 4 * size_t(max_span) + sizeof(int) * 2 * size_t(max_span) + 8 + sizeof(unsigned long long) * 6 * size_t(P) +
                       sizeof(unsigned) * (size_t(max_span + 64 + 383) / 32 + 4) + size_t(max_span) + 16;
    if (lds > 160 * 1024 - 1024) return fail(ctx, MLH_ERR_UNSUPPORTED, "ring too long for the LDS-resident label kernel");
    if (P > 2048) return fail(ctx, MLH_ERR_UNSUPPORTED, "sector longer than 2048 points");      // before anything is enqueued or bracketed
    // scratch of the reference-order pass (only touched by sectors with equal curvatures): in LDS behind the extents when it fits, else in HBM
    const int tie_words = 5 * P + P / 32 + 2 + 128;
    int *tie_scratch = nullptr;
    if (ctx->extract_tie_ref) {
        const size_t lds_tie = ((lds + 15) & ~size_t(15)) + 16 + sizeof(int) * 6 * size_t(tie_words);
        if (lds_tie <= 160 * 1024 - 1024) lds = lds_tie;
        else {
            MLH_HIP(ctx, sb.tie_scratch.ensure(sizeof(int) * 6 * size_t(tie_words) * size_t(R)));
            tie_scratch = sb.tie_scratch.as<int>();
        }
    }
    MLH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(label_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));

    prof_begin(ctx, MLH_K_EXTRACT);
    MLH_LAUNCH(curvature_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sb.pts.as<float4>(), n, sb.curvature.as<float>(),
                       sb.label.as<int>(), sb.picked.as<int>());
    LabelArgs la;
    la.pts = sb.pts.as<float4>(); la.curv = sb.curvature.as<float>(); la.start = sb.start.as<int>(); la.end = sb.end_ptr();
    la.label = sb.label.as<int>(); la.picked = sb.picked.as<int>(); la.stage = sb.stage.as<int>(); la.ring_counts = sb.ring_counts.as<int>();
    la.n = n; la.max_span = max_span; la.sort_p = P;
    la.tie_ref = ctx->extract_tie_ref ? 1 : 0; la.tie_words = tie_words; la.tie_scratch = tie_scratch;
    MLH_LAUNCH(label_kernel, dim3(R), dim3(LTPB), lds, st, la);
    EmitArgs ea;
    ea.start = sb.start.as<int>(); ea.end = sb.end_ptr(); ea.label = sb.label.as<int>(); ea.stage = sb.stage.as<int>();
    ea.ring_counts = sb.ring_counts.as<int>(); ea.ring_offsets = sb.ring_offsets.as<int>(); ea.totals = sb.totals.as<int>(); ea.n_rings = R;
    ea.list0 = sb.lists[0].as<int>(); ea.list1 = sb.lists[1].as<int>(); ea.list2 = sb.lists[2].as<int>(); ea.list3 = sb.lists[3].as<int>();
    MLH_LAUNCH(emit_kernel, dim3(R), dim3(256), 0, st, ea);
    prof_end(ctx, MLH_K_EXTRACT);
    MLH_HIP(ctx, hipGetLastError());
    sb.extracted = true;
    sb.h_lists_valid = false;
    return MLH_OK;
}

}  // namespace mlh
