// Device-side building blocks shared by the correspondence kernels (match.hip: scan-to-map; track.hip: scan-to-scan):
// 64-bit (distance bits, index) keys, DPP lane exchanges, the exact K-NN of one query by a group of 8 or 16 lanes over the
// dense cell grid built by grid.hip.
#pragma once
#include "ctx.hpp"
#include <cfloat>

#ifndef MLH_KSTAGE
#define MLH_KSTAGE(i) do { } while (0)
#endif

namespace mlh {

// FLANN's L2 accumulation of one candidate (dx*dx + dy*dy + dz*dz, left to right, f32, no contraction) -- the ONE place every search and the bound of the bounded
// search take a squared distance from, with explicitly rounded operations: two inlined copies of an expression could be contracted differently by a compiler that
// ignored -ffp-contract=off, and the bounded search compares the bits of one against the bits of the other.
__device__ __forceinline__ float knn_sqdist(float dx, float dy, float dz)
{
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

constexpr int TPB = 256;

// lanes per query in the correspondence kernel: 8 when the launch fills the chip on its own (throughput: one wavefront serves 8
// queries), 16 when it does not (latency: a frame's ~20k thinned features leave most SIMDs idle with 8, and twice the lanes halve
// the candidate trips of the queries in dense cells, which set the kernel's duration)
constexpr int KNN_LATENCY_LIMIT = 8192;   // total queries up to which every kind uses 16 lanes (the launch cannot fill the chip either way)
#ifndef MLH_KNN_U
#define MLH_KNN_U 8
#endif
// candidate loads in flight per lane. A/B on the bench frame inside one gpurun call, two alternations (profiles/r05_knockout_experiments.txt): 2: 0.1286 / 0.1286,
// 4 (rounds 1-4): 0.1300 / 0.1268, 8: 0.1246 / 0.1232, 12: 0.1283 / 0.1281, 16: 0.1304 / 0.1304 ms per step -- the queries in dense corner cells set the launch's
// duration by their dependent candidate trips; eight loads halve those trips, more cost registers (occupancy) without shortening anything
constexpr int KNN_U = MLH_KNN_U;
constexpr unsigned long long KEY_INF = 0x7f800000ffffffffull;   // (+inf, max index)

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int mask)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, mask);
    hi = __shfl_xor(hi, mask);
    return ((unsigned long long)hi << 32) | lo;
}

// DPP lane permutations (VALU latency, no LDS crossbar trip): the 3 (4) exchange partners that all-reduce a group of 8 (16) lanes
constexpr int DPP_QUAD_SWAP1 = 0xB1;      // quad_perm:[1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;      // quad_perm:[2,3,0,1]
constexpr int DPP_ROW_HALF_MIRROR = 0x141; // lane i <-> 7 - i inside each 8-lane half row
constexpr int DPP_ROW_MIRROR = 0x140;      // lane i <-> 15 - i inside each 16-lane row
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v)
{
    int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long m)
{
    const unsigned long long o = dpp_u64<CTRL>(m);
    return o < m ? o : m;
}
// row_shr:OFF with out-of-row lanes reading 0 (bound_ctrl)
template <int OFF>
__device__ __forceinline__ int dpp_row_shr(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + OFF, 0xF, 0xF, true); }

// 64-bit unsigned minimum / maximum of two KEYS in ONE instruction each. gfx950 has no v_min_u64, but it has full-rate v_min_f64 / v_max_f64, and a key
// read as an IEEE double orders exactly as it does as an unsigned integer: its upper word is the bit pattern of a NON-NEGATIVE f32 (a squared distance, or
// +inf for "no candidate"), so the double is non-negative, and its exponent field -- the f32's exponent and top three mantissa bits -- is all ones only for
// f32 patterns 0x7ff00000 and above, which no distance takes (a NaN distance, 0x7fc00000, reads as an ordinary double). Non-negative finite doubles compare
// like their bit patterns, denormals included (f64 denormals are not flushed). Round 2 built the same results from a 64-bit compare and four v_cndmask
// per slot: the sorted insertion below went from ~24 to 9 instructions, a compare-exchange from ~9 to 2.
__device__ __forceinline__ unsigned long long key_min(unsigned long long a, unsigned long long b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (unsigned long long)__double_as_longlong(r);
}
__device__ __forceinline__ unsigned long long key_max(unsigned long long a, unsigned long long b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(__longlong_as_double((long long)a)), "v"(__longlong_as_double((long long)b)));
    return (unsigned long long)__double_as_longlong(r);
}

// sorted insertion: slot i becomes max(old k[i-1], min(old k[i], key)) -- the key where it belongs, the larger entries moved up by one, the largest
// dropped; no branch, no compare chain
template <int K>
__device__ __forceinline__ void key_insert(unsigned long long (&k)[K], unsigned long long key)
{
#pragma unroll
    for (int i = K - 1; i > 0; --i) k[i] = key_max(k[i - 1], key_min(k[i], key));
    k[0] = key_min(k[0], key);
}

__device__ __forceinline__ float clamp_cell_f(float v, float o, float inv_h, int n)
{
    float f = floorf((v - o) * inv_h);
    return fminf(fmaxf(f, -2.f), float(n + 1));   // also squashes NaN/inf before the int conversion
}

// exact K-NN (K = 5, or 10 for buildCalibMap's non-reference LiDARs) of (qx,qy,qz) by a group of 8 lanes (8 queries per
// wavefront); on return every lane holds the K keys ascending.
// The 27-cell neighbourhood is 9 x-runs (the 3 x-adjacent cells of one (dy,dz) are one contiguous range of the cell-sorted
// array). Lane r fetches the bounds of run r (lane 0 also run 8); a 3-step shuffle scan gives the run offsets, which go to
// LDS; the lanes then stride over the FLAT concatenation of the 9 runs (lane l takes candidates l, l+8, ...), so the work
// is balanced whatever the per-run occupancy, consecutive lanes read consecutive float4 points, and a query with fewer than
// K candidates (most corner features far from any edge) is rejected right after the 18 cell_start words.
// lds_run: 20 ints per group: [0..9] prefix offsets of the runs (10 entries), [10..18] base index of each run.
// SHORT_OK: also answer a query whose 27 cells hold fewer than K points (the raw query API, mlh_knn: every neighbour inside the acceptance radius is reported,
// like a kd-tree's; the matching kernels need K neighbours inside the radius or nothing, and leave right after the cell bounds)
template <int K, int G, bool SHORT_OK = false>
__device__ __forceinline__ void knn_group(const GridDev &g, float qx, float qy, float qz, int gl, int *lds_run, unsigned long long (&out)[K])
{
    static_assert(G == 8 || G == 16, "group width");
    unsigned long long k[K];
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = KEY_INF;
    const int cx = int(clamp_cell_f(qx, g.ox, g.inv_h, g.nx));
    const int cy = int(clamp_cell_f(qy, g.oy, g.inv_h, g.ny));
    const int cz = int(clamp_cell_f(qz, g.oz, g.inv_h, g.nz));
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    int b = 0, e = 0, b8 = 0, e8 = 0;
    {
        const int y = cy + (gl % 3) - 1, z = cz + (gl / 3) - 1;
        if ((gl < 9) && (x0 <= x1) && (y >= 0) && (y < g.ny) && (z >= 0) && (z < g.nz)) {
            const int row = (z * g.ny + y) * g.nx;
            b = g.cell_start[row + x0];
            e = g.cell_start[row + x1 + 1];
        }
        const int y8 = cy + 1, z8 = cz + 1;       // run 8 = (dy, dz) = (+1, +1): lane 0's second run when there are only 8 lanes
        if (G == 8 && gl == 0 && (x0 <= x1) && (y8 < g.ny) && (z8 < g.nz) && (y8 >= 0) && (z8 >= 0)) {
            const int row = (z8 * g.ny + y8) * g.nx;
            b8 = g.cell_start[row + x0];
            e8 = g.cell_start[row + x1 + 1];
        }
    }
    MLH_KSTAGE(2);
    const int len = e - b;
    int incl = len;
    if (G == 16) {
        // small launches are latency-bound: DPP row shifts / lane swaps (VALU latency) instead of ds_bpermute round trips
        { const int t = dpp_row_shr<1>(incl); if (gl >= 1) incl += t; }
        { const int t = dpp_row_shr<2>(incl); if (gl >= 2) incl += t; }
        { const int t = dpp_row_shr<4>(incl); if (gl >= 4) incl += t; }
        { const int t = dpp_row_shr<8>(incl); if (gl >= 8) incl += t; }
    } else {
        // chip-filling launches are issue-bound on the VALU: leave the exchanges to the LDS crossbar
#pragma unroll
        for (int off = 1; off < G; off <<= 1) {
            const int t = __shfl_up(incl, off, G);
            if (gl >= off) incl += t;
        }
    }
    const int len8 = (G == 8) ? __shfl(e8 - b8, 0, G) : 0;
    const int total = __shfl(incl, G - 1, G) + len8;
    if (SHORT_OK ? total > 0 : total >= K) {           // uniform over the group
        if (G == 8) {
            lds_run[gl] = incl - len;                  // prefix[r]
            lds_run[10 + gl] = b;                      // base[r]
            if (gl == 0) { lds_run[8] = total - len8; lds_run[9] = total; lds_run[18] = b8; }
        } else {
            if (gl < 10) lds_run[gl] = incl - len;     // lane 9 holds no run: incl - len = total
            if (gl < 9) lds_run[10 + gl] = b;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        int cr = 0, hi = lds_run[1], base = lds_run[10], lo = 0;
        // KNN_U candidates per lane per trip. The addresses depend only on the run table, so they are resolved first (LDS look-ups,
        // divergent control flow) and the KNN_U loads are then issued UNCONDITIONALLY, back to back (lanes past the end re-read
        // element 0): with the loads inside the divergent address code the compiler drains vmcnt after every one of them and a trip
        // costs KNN_U memory round trips instead of one.
        for (int j = gl; j < total; j += G * KNN_U) {
            int addr[KNN_U];
            bool v[KNN_U];
#pragma unroll
            for (int u = 0; u < KNN_U; ++u) {
                const int jj = j + G * u;
                v[u] = jj < total;
                addr[u] = 0;
                if (v[u]) {
                    while (jj >= hi) { ++cr; lo = hi; hi = lds_run[cr + 1]; base = lds_run[10 + cr]; }
                    addr[u] = base + (jj - lo);
                }
            }
            float4 p[KNN_U];
#pragma unroll
            for (int u = 0; u < KNN_U; ++u) p[u] = g.sorted[addr[u]];
#pragma unroll
            for (int u = 0; u < KNN_U; ++u) {
                if (v[u]) {
                    float dx = p[u].x - qx, dy = p[u].y - qy, dz = p[u].z - qz;
                    const float d = knn_sqdist(dx, dy, dz);
                    key_insert<K>(k, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w));
                }
            }
        }
    }
    MLH_KSTAGE(3);
    // tournament merge: K rounds of group-min over the lanes' current heads
#pragma unroll
    for (int t = 0; t < K; ++t) {
        unsigned long long m = k[0];
        if (G == 16) {
            m = dpp_min_u64<DPP_QUAD_SWAP1>(m);
            m = dpp_min_u64<DPP_QUAD_SWAP2>(m);
            m = dpp_min_u64<DPP_ROW_HALF_MIRROR>(m);
            m = dpp_min_u64<DPP_ROW_MIRROR>(m);
        } else {
#pragma unroll
            for (int off = 1; off < G; off <<= 1) {
                const unsigned long long o = shfl_xor_u64(m, off);
                m = o < m ? o : m;
            }
        }
        out[t] = m;
        if (k[0] == m && m != KEY_INF) {
#pragma unroll
            for (int i = 0; i < K - 1; ++i) k[i] = k[i + 1];
            k[K - 1] = KEY_INF;
        }
    }
}

// ---------------------------------------------------------------- pruned 16-lane search (the fused match kernel, match.hip)
// u32 group-min over the G (8 or 16) lanes of a group, DPP only
template <int G = 16>
__device__ __forceinline__ unsigned dpp_row_min_u32(unsigned m)
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_QUAD_SWAP1, 0xF, 0xF, false); m = o < m ? o : m;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_QUAD_SWAP2, 0xF, 0xF, false); m = o < m ? o : m;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_ROW_HALF_MIRROR, 0xF, 0xF, false); m = o < m ? o : m;
    if (G == 16) { o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_ROW_MIRROR, 0xF, 0xF, false); m = o < m ? o : m; }
    return m;
}

// K rounds of group-min over the 16 lanes' sorted lists, on 32-bit words: the smallest head distance of the group, then the smallest
// index among the lanes whose head has that distance -- the lexicographic (distance, index) minimum, ties included -- and the owner pops
template <int K, int G = 16>
__device__ __forceinline__ void knn_tournament16(unsigned long long (&k)[K], unsigned long long (&out)[K])
{
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const unsigned hd = (unsigned)(k[0] >> 32), hi = (unsigned)k[0];
        const unsigned md = dpp_row_min_u32<G>(hd);
        const unsigned mi = dpp_row_min_u32<G>(hd == md ? hi : 0xffffffffu);
        out[t] = ((unsigned long long)md << 32) | mi;
        if (hd == md && hi == mi && k[0] != KEY_INF) {
#pragma unroll
            for (int i = 0; i < K - 1; ++i) k[i] = k[i + 1];
            k[K - 1] = KEY_INF;
        }
    }
}

constexpr int KNN_RUN_WORDS = 40;       // LDS ints per query group: [0..18] segment prefix offsets (+ end), [19..36] segment bases
constexpr int KNN_SEG_BASE = 19;
constexpr int KNN_TWO_PHASE_MIN = 128;  // candidates in the 27 cells from which the search goes near-cells-first and prunes the rest
constexpr float KNN_PRUNE_SLACK = 1.0e-3f;   // metres taken off every face distance before a cell is pruned (covers the f32 rounding of
                                             // the cell assignment; a pruned cell is farther than the bound by at least this much)

// flat, balanced walk over the segments of the run table: candidate j of the concatenation goes to lane j % G
// ascending compare-exchange of two keys (v_min_f64 / v_max_f64 on the key bits, see key_min)
__device__ __forceinline__ void key_cx(unsigned long long &a, unsigned long long &b)
{
    const unsigned long long lo = key_min(a, b), hi = key_max(a, b);
    a = lo;
    b = hi;
}

// EMPTY: the lane's list is empty on entry (the first walk of a search): its first trip's candidates are sorted by a 5-exchange network and
// become the list, instead of going through KNN_U sorted insertions into a list of +inf
template <int K, int G = 16, bool EMPTY = false>
__device__ __forceinline__ void knn_walk16(const GridDev &g, float qx, float qy, float qz, int gl, const int *lds_run, int total,
                                           unsigned bound_bits, unsigned long long (&k)[K])
{
    int cr = 0, hi = lds_run[1], base = lds_run[KNN_SEG_BASE], lo = 0;
    bool first = true;
    for (int j = gl; j < total; j += G * KNN_U) {
        int addr[KNN_U];
        bool v[KNN_U];
#pragma unroll
        for (int u = 0; u < KNN_U; ++u) {
            const int jj = j + G * u;
            v[u] = jj < total;
            addr[u] = 0;
            if (v[u]) {
                while (jj >= hi) { ++cr; lo = hi; hi = lds_run[cr + 1]; base = lds_run[KNN_SEG_BASE + cr]; }
                addr[u] = base + (jj - lo);
            }
        }
        float4 p[KNN_U];
#pragma unroll
        for (int u = 0; u < KNN_U; ++u) p[u] = g.sorted[addr[u]];     // unconditional, back to back: one round trip per trip
#pragma unroll
        for (int u = 0; u < KNN_U; ++u) asm volatile("" : "+v"(p[u].w));   // keep the index word in the 16-byte load (the compiler would
                                                                           // otherwise fetch it again, dependently, inside the insertion branch)
        if (EMPTY && KNN_U == 4 && K >= 4 && first) {
            first = false;
            unsigned long long c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float dx = p[u].x - qx, dy = p[u].y - qy, dz = p[u].z - qz;
                const float d = knn_sqdist(dx, dy, dz);
                const bool take = v[u] && (__float_as_uint(d) <= bound_bits);
                c[u] = take ? (((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w)) : KEY_INF;
            }
            key_cx(c[0], c[1]); key_cx(c[2], c[3]); key_cx(c[0], c[2]); key_cx(c[1], c[3]); key_cx(c[1], c[2]);
#pragma unroll
            for (int u = 0; u < 4; ++u) k[u] = c[u];
            continue;
        }
#pragma unroll
        for (int u = 0; u < KNN_U; ++u) {
            if (v[u]) {
                float dx = p[u].x - qx, dy = p[u].y - qy, dz = p[u].z - qz;
                const float d = knn_sqdist(dx, dy, dz);
                if (__float_as_uint(d) <= bound_bits)
                    key_insert<K>(k, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(p[u].w));
            }
        }
    }
}

// inclusive sum over the G (8 or 16) lanes of a group (DPP row shifts; a lane never adds what came from the neighbouring group)
template <int G = 16>
__device__ __forceinline__ int row_scan16(int v, int gl)
{
    { const int t = dpp_row_shr<1>(v); if (gl >= 1) v += t; }
    { const int t = dpp_row_shr<2>(v); if (gl >= 2) v += t; }
    { const int t = dpp_row_shr<4>(v); if (gl >= 4) v += t; }
    if (G == 16) { const int t = dpp_row_shr<8>(v); if (gl >= 8) v += t; }
    return v;
}

// run table of a group of G lanes: every lane offers NP pieces [b[p], b[p] + len[p]); only non-empty pieces get a slot, in lane order.
// Returns the total number of candidates (uniform over the group).
template <int G, int NP>
__device__ __forceinline__ int knn_fill_table(int *lds_run, int gl, const int (&b)[NP], const int (&len)[NP])
{
    int lsum = 0, np = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) { lsum += len[p]; np += len[p] > 0 ? 1 : 0; }
    // (a 32-lane group: the pieces come from its first 16-lane row only -- nine (dy, dz) rows, one per lane -- so that row's scan is the group's)
    constexpr int GS = G > 16 ? 16 : G, LASTL = GS - 1;
    const int incl = row_scan16<GS>(lsum, gl), slot_incl = row_scan16<GS>(np, gl);
    int slot = slot_incl - np, off = incl - lsum;
#pragma unroll
    for (int p = 0; p < NP; ++p)
        if (len[p] > 0) { lds_run[slot] = off; lds_run[KNN_SEG_BASE + slot] = b[p]; ++slot; off += len[p]; }
    if (gl == LASTL) lds_run[slot_incl] = incl;                        // end sentinel behind the last slot
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    return __shfl(incl, LASTL, G);
}

// run table of the group: every lane offers up to two pieces [bL, bL + lenL), [bR, bR + lenR); only non-empty pieces get a slot.
// Returns the total number of candidates (uniform over the group).
__device__ __forceinline__ int knn_fill_table16(int *lds_run, int gl, int bL, int lenL, int bR, int lenR)
{
    const int len = lenL + lenR, np = (lenL > 0 ? 1 : 0) + (lenR > 0 ? 1 : 0);
    const int incl = row_scan16(len, gl), slot_incl = row_scan16(np, gl);
    int slot = slot_incl - np;
    if (lenL > 0) { lds_run[slot] = incl - len; lds_run[KNN_SEG_BASE + slot] = bL; ++slot; }
    if (lenR > 0) { lds_run[slot] = incl - lenR; lds_run[KNN_SEG_BASE + slot] = bR; }
    if (gl == 15) lds_run[slot_incl] = incl;                           // end sentinel behind the last slot
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    return __shfl(incl, 15, 16);
}

// u32 group-max over the G (8 or 16) lanes of a group, DPP only
template <int G = 16>
__device__ __forceinline__ unsigned dpp_row_max_u32(unsigned m)
{
    unsigned o;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_QUAD_SWAP1, 0xF, 0xF, false); m = o > m ? o : m;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_QUAD_SWAP2, 0xF, 0xF, false); m = o > m ? o : m;
    o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_ROW_HALF_MIRROR, 0xF, 0xF, false); m = o > m ? o : m;
    if (G == 16) { o = (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, DPP_ROW_MIRROR, 0xF, 0xF, false); m = o > m ? o : m; }
    return m;
}

// ---- 32 lanes per query (two 16-lane DPP rows): the dense kinds of a frame-sized launch, where a query's candidate trips set the launch's duration
// lane i <-> lane i ^ 16 inside each 32-lane half of the wavefront (ds_swizzle, bit mode: and 0x1f, or 0, xor 0x10)
__device__ __forceinline__ unsigned swz_xor16_u32(unsigned v) { return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F); }
__device__ __forceinline__ unsigned long long swz_xor16_u64(unsigned long long v)
{
    return ((unsigned long long)swz_xor16_u32((unsigned)(v >> 32)) << 32) | swz_xor16_u32((unsigned)v);
}
template <int G>
__device__ __forceinline__ unsigned group_max_u32(unsigned m)
{
    if constexpr (G == 32) {
        m = dpp_row_max_u32<16>(m);
        const unsigned o = swz_xor16_u32(m);
        return o > m ? o : m;
    } else return dpp_row_max_u32<G>(m);
}
// every 16-lane row merges its own lists (knn_tournament16), the rows trade their K winners (all exchanges in flight together) and each lane inserts the other
// row's into its own: the K smallest (distance, index) keys of the 32 lanes' lists, ascending -- no candidate is on two lanes, so the keys are distinct
template <int K>
__device__ __forceinline__ void knn_tournament32(unsigned long long (&k)[K], unsigned long long (&out)[K])
{
    knn_tournament16<K, 16>(k, out);
    unsigned long long other[K];
#pragma unroll
    for (int i = 0; i < K; ++i) other[i] = swz_xor16_u64(out[i]);
#pragma unroll
    for (int i = 0; i < K; ++i) key_insert<K>(out, other[i]);
}

// Exact K-NN of (qx,qy,qz) when an upper bound of the K-th neighbour's squared distance is known BEFORE the search (bound_bits: the f32 bit pattern; the
// caller got it from K map points it already knows -- the previous Gauss-Newton iteration's neighbours of the same feature, re-measured from the query's new
// position: K points within the bound exist, so nothing farther can be among the K nearest). One walk, over the cells whose box is not farther than the
// bound -- what phase 2 of the pruned search does after phase 1 has found its bound, without a phase 1, a merge and a second run table. Same result as
// knn_group16_pruned / knn_group8_pruned wherever the K-th distance is below the cell edge; where it is not, both report a K-th distance of at least the
// cell edge (the acceptance test rejects the feature either way). lds_run: KNN_RUN_WORDS ints.
// COLD: no bound is known (bound_bits = +inf: all 27 cells in one flat walk -- what a 32-lane group does where the 16-lane search goes near-cells-first); a query
// with fewer than K points in its 27 cells finds nothing, as in knn_group16_pruned.
template <int K, int G, bool COLD = false>
__device__ __forceinline__ void knn_group_bounded(const GridDev &g, float qx, float qy, float qz, int gl, int *lds_run, unsigned bound_bits,
                                                  unsigned long long (&out)[K])
{
    static_assert(G == 8 || G == 16 || G == 32, "group width");
    constexpr int NR = (G >= 16) ? 1 : 2;                               // (dy, dz) rows per lane: lanes 0..8 one each, or lanes 0..4 two each
    unsigned long long k[K];
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = KEY_INF;
    const int cx = int(clamp_cell_f(qx, g.ox, g.inv_h, g.nx));
    const int cy = int(clamp_cell_f(qy, g.oy, g.inv_h, g.ny));
    const int cz = int(clamp_cell_f(qz, g.oz, g.inv_h, g.nz));
    int w[NR][4], dyr[NR], dzr[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
        const int r = NR * gl + s;
        dyr[s] = (r % 3) - 1; dzr[s] = ((r / 3) % 3) - 1;
        w[s][0] = w[s][1] = w[s][2] = w[s][3] = 0;
        const int y = cy + dyr[s], z = cz + dzr[s];
        if ((r < 9) && (y >= 0) && (y < g.ny) && (z >= 0) && (z < g.nz)) {
            const int row = (z * g.ny + y) * g.nx;
            const int xa = min(max(cx - 1, 0), g.nx), xb = min(max(cx, 0), g.nx), xc = min(max(cx + 1, 0), g.nx), xd = min(max(cx + 2, 0), g.nx);
            w[s][0] = g.cell_start[row + xa]; w[s][1] = g.cell_start[row + xb]; w[s][2] = g.cell_start[row + xc]; w[s][3] = g.cell_start[row + xd];
        }
    }
    MLH_KSTAGE(2);
    const float h = 1.f / g.inv_h;
    const float fx0 = g.ox + float(cx) * h, fy0 = g.oy + float(cy) * h, fz0 = g.oz + float(cz) * h;
    const float gxl = fmaxf((qx - fx0) - KNN_PRUNE_SLACK, 0.f), gxr = fmaxf(((fx0 + h) - qx) - KNN_PRUNE_SLACK, 0.f);
    const float Bm = __uint_as_float(bound_bits) * 1.0001f;
    int pb[NR], pl[NR];
#pragma unroll
    for (int s = 0; s < NR; ++s) {
        const float gy = fmaxf((dyr[s] < 0 ? qy - fy0 : (dyr[s] > 0 ? (fy0 + h) - qy : 0.f)) - (dyr[s] ? KNN_PRUNE_SLACK : 0.f), 0.f);
        const float gz = fmaxf((dzr[s] < 0 ? qz - fz0 : (dzr[s] > 0 ? (fz0 + h) - qz : 0.f)) - (dzr[s] ? KNN_PRUNE_SLACK : 0.f), 0.f);
        const float dM = gy * gy + gz * gz, dL = dM + gxl * gxl, dR = dM + gxr * gxr;
        const bool keepM = dM <= Bm;
        const int kb = keepM ? (dL <= Bm ? w[s][0] : w[s][1]) : w[s][3], ke = keepM ? (dR <= Bm ? w[s][3] : w[s][2]) : w[s][3];
        pb[s] = kb; pl[s] = max(ke - kb, 0);
    }
    const int total = knn_fill_table<G, NR>(lds_run, gl, pb, pl);
    if (total >= (COLD ? K : 1)) knn_walk16<K, G, true>(g, qx, qy, qz, gl, lds_run, total, bound_bits, k);
    MLH_KSTAGE(3);
    if constexpr (G == 32) knn_tournament32<K>(k, out);
    else knn_tournament16<K, G>(k, out);
}

// Exact K-NN of (qx,qy,qz) by a group of 16 lanes; every lane returns the K keys ascending. Same result as knn_group<K, 16>, fewer
// candidates read where the map is dense:
//   lanes 0..8 fetch the FOUR cell_start words of their row (dy, dz) -> the three x-cells of the row separately, and the squared
//   distance from the query to each cell's box (less a 1 mm slack). With fewer than KNN_TWO_PHASE_MIN points in the 27 cells the
//   search is one flat walk over everything, as before (a second dependent round trip would not pay). Otherwise it goes
//   near-cells-first: phase 1 walks the cells within tau of the query, tau ~ 1.3 x the K-th-neighbour distance a planar map of this
//   density would give; the K-th smallest distance found there (merged over the lanes on the 32-bit distance words) is an upper bound
//   B of the true K-th distance, and phase 2 walks only the cells that are not farther than sqrt(B) and were not read yet -- none,
//   when tau was large enough. On the 0.2 m corner map this reads ~1/4 of the 27 cells' points.
// lds_run: 2 * KNN_RUN_WORDS ints for this group.
template <int K>
__device__ __forceinline__ void knn_group16_pruned(const GridDev &g, float qx, float qy, float qz, int gl, int *lds_run,
                                                   unsigned long long (&out)[K])
{
    unsigned long long k[K];
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = KEY_INF;
    const int cx = int(clamp_cell_f(qx, g.ox, g.inv_h, g.nx));
    const int cy = int(clamp_cell_f(qy, g.oy, g.inv_h, g.ny));
    const int cz = int(clamp_cell_f(qz, g.oz, g.inv_h, g.nz));
    const int dyr = (gl % 3) - 1, dzr = ((gl / 3) % 3) - 1;
    int w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    {
        const int y = cy + dyr, z = cz + dzr;
        if ((gl < 9) && (y >= 0) && (y < g.ny) && (z >= 0) && (z < g.nz)) {
            const int row = (z * g.ny + y) * g.nx;
            const int xa = min(max(cx - 1, 0), g.nx), xb = min(max(cx, 0), g.nx), xc = min(max(cx + 1, 0), g.nx), xd = min(max(cx + 2, 0), g.nx);
            w0 = g.cell_start[row + xa]; w1 = g.cell_start[row + xb]; w2 = g.cell_start[row + xc]; w3 = g.cell_start[row + xd];
        }
    }
    MLH_KSTAGE(2);
    const int n27 = __shfl(row_scan16(w3 - w0, gl), 15, 16);
    if (n27 >= K) {                                                     // uniform over the group
        bool one_pass = n27 < KNN_TWO_PHASE_MIN;
        float dL = 0.f, dM = 0.f, dR = 0.f;
        int b1 = w0, e1 = w3;                                           // phase-1 range of this lane's row
        if (!one_pass) {
            const float h = 1.f / g.inv_h;
            const float fx0 = g.ox + float(cx) * h, fy0 = g.oy + float(cy) * h, fz0 = g.oz + float(cz) * h;
            const float gy = fmaxf((dyr < 0 ? qy - fy0 : (dyr > 0 ? (fy0 + h) - qy : 0.f)) - (dyr ? KNN_PRUNE_SLACK : 0.f), 0.f);
            const float gz = fmaxf((dzr < 0 ? qz - fz0 : (dzr > 0 ? (fz0 + h) - qz : 0.f)) - (dzr ? KNN_PRUNE_SLACK : 0.f), 0.f);
            const float gxl = fmaxf((qx - fx0) - KNN_PRUNE_SLACK, 0.f), gxr = fmaxf(((fx0 + h) - qx) - KNN_PRUNE_SLACK, 0.f);
            dM = gy * gy + gz * gz; dL = dM + gxl * gxl; dR = dM + gxr * gxr;
            // K-th neighbour of a planar map with n27 / 9 points per cell: r^2 = 9 K / (pi n27); tau^2 = 1.69 r^2, within [0.15^2, 0.6^2]
            float tau2 = fminf(fmaxf((1.69f * 9.f * float(K) / 3.14159265f) / float(n27), 0.0225f), 0.36f);
            int n1 = 0;
            // a query off the surface (first iteration: the pose is still wrong) has nothing that close: widen until K points are in
#pragma unroll 1
            for (int widen = 0; widen < 3; ++widen) {
                const bool inM = dM <= tau2;
                b1 = inM ? (dL <= tau2 ? w0 : w1) : w3;
                e1 = inM ? (dR <= tau2 ? w3 : w2) : w3;
                n1 = __shfl(row_scan16(e1 - b1, gl), 15, 16);
                if (n1 >= K) break;
                tau2 *= 4.f;
            }
            if (n1 < K) { one_pass = true; b1 = w0; e1 = w3; }          // still too few points near the query: read everything at once
        }
        const int total1 = knn_fill_table16(lds_run, gl, b1, e1 - b1, 0, 0);
        knn_walk16<K, 16, true>(g, qx, qy, qz, gl, lds_run, total1, 0x7f800000u, k);
        if (!one_pass) {
            // K-th smallest distance of phase 1 (merge of the lanes' sorted lists on the distance words; ties pop together, which can
            // only enlarge the bound)
            unsigned hd[K];
#pragma unroll
            for (int i = 0; i < K; ++i) hd[i] = (unsigned)(k[i] >> 32);
            unsigned kth = 0x7f800000u;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                kth = dpp_row_min_u32(hd[0]);
                if (hd[0] == kth) {
#pragma unroll
                    for (int i = 0; i < K - 1; ++i) hd[i] = hd[i + 1];
                    hd[K - 1] = 0x7f800000u;
                }
            }
            const float Bm = __uint_as_float(kth) * 1.0001f;
            // cells not farther than the bound and not read in phase 1: at most one piece on either side of the phase-1 range
            const bool keepM = dM <= Bm;
            const int kb = keepM ? (dL <= Bm ? w0 : w1) : w3, ke = keepM ? (dR <= Bm ? w3 : w2) : w3;
            const int lenL = max(min(ke, b1) - kb, 0), bR = max(kb, e1), lenR = max(ke - bR, 0);
            const int total2 = knn_fill_table16(lds_run + KNN_RUN_WORDS, gl, kb, lenL, bR, lenR);
            if (total2 > 0) knn_walk16<K>(g, qx, qy, qz, gl, lds_run + KNN_RUN_WORDS, total2, kth, k);
        }
    }
    MLH_KSTAGE(3);
    knn_tournament16<K>(k, out);
}

// The same search by a group of 8 lanes (twice the queries per wavefront: the correspondence kernel of a full frame is bound by VALU
// issue, not by latency, and per-query instruction count is what an 8-lane group halves). Lanes 0..4 hold the rows 2l and 2l + 1 of the
// 9 (dy, dz) rows; everything else as in knn_group16_pruned.
template <int K>
__device__ __forceinline__ void knn_group8_pruned(const GridDev &g, float qx, float qy, float qz, int gl, int *lds_run,
                                                  unsigned long long (&out)[K])
{
    unsigned long long k[K];
#pragma unroll
    for (int i = 0; i < K; ++i) k[i] = KEY_INF;
    const int cx = int(clamp_cell_f(qx, g.ox, g.inv_h, g.nx));
    const int cy = int(clamp_cell_f(qy, g.oy, g.inv_h, g.ny));
    const int cz = int(clamp_cell_f(qz, g.oz, g.inv_h, g.nz));
    int w[2][4];
    int dyr[2], dzr[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int r = 2 * gl + s;
        dyr[s] = (r % 3) - 1; dzr[s] = (r / 3) - 1;
        w[s][0] = w[s][1] = w[s][2] = w[s][3] = 0;
        const int y = cy + dyr[s], z = cz + dzr[s];
        if ((r < 9) && (y >= 0) && (y < g.ny) && (z >= 0) && (z < g.nz)) {
            const int row = (z * g.ny + y) * g.nx;
            const int xa = min(max(cx - 1, 0), g.nx), xb = min(max(cx, 0), g.nx), xc = min(max(cx + 1, 0), g.nx), xd = min(max(cx + 2, 0), g.nx);
            w[s][0] = g.cell_start[row + xa]; w[s][1] = g.cell_start[row + xb]; w[s][2] = g.cell_start[row + xc]; w[s][3] = g.cell_start[row + xd];
        }
    }
    MLH_KSTAGE(2);
    const int n27 = __shfl(row_scan16<8>((w[0][3] - w[0][0]) + (w[1][3] - w[1][0]), gl), 7, 8);
    if (n27 >= K) {                                                     // uniform over the group
        bool one_pass = n27 < KNN_TWO_PHASE_MIN;
        float dL[2] = {0.f, 0.f}, dM[2] = {0.f, 0.f}, dR[2] = {0.f, 0.f};
        int b1[2] = {w[0][0], w[1][0]}, e1[2] = {w[0][3], w[1][3]};
        if (!one_pass) {
            const float h = 1.f / g.inv_h;
            const float fx0 = g.ox + float(cx) * h, fy0 = g.oy + float(cy) * h, fz0 = g.oz + float(cz) * h;
            const float gxl = fmaxf((qx - fx0) - KNN_PRUNE_SLACK, 0.f), gxr = fmaxf(((fx0 + h) - qx) - KNN_PRUNE_SLACK, 0.f);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float gy = fmaxf((dyr[s] < 0 ? qy - fy0 : (dyr[s] > 0 ? (fy0 + h) - qy : 0.f)) - (dyr[s] ? KNN_PRUNE_SLACK : 0.f), 0.f);
                const float gz = fmaxf((dzr[s] < 0 ? qz - fz0 : (dzr[s] > 0 ? (fz0 + h) - qz : 0.f)) - (dzr[s] ? KNN_PRUNE_SLACK : 0.f), 0.f);
                dM[s] = gy * gy + gz * gz; dL[s] = dM[s] + gxl * gxl; dR[s] = dM[s] + gxr * gxr;
            }
            float tau2 = fminf(fmaxf((1.69f * 9.f * float(K) / 3.14159265f) / float(n27), 0.0225f), 0.36f);
            int n1 = 0;
#pragma unroll 1
            for (int widen = 0; widen < 3; ++widen) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bool inM = dM[s] <= tau2;
                    b1[s] = inM ? (dL[s] <= tau2 ? w[s][0] : w[s][1]) : w[s][3];
                    e1[s] = inM ? (dR[s] <= tau2 ? w[s][3] : w[s][2]) : w[s][3];
                }
                n1 = __shfl(row_scan16<8>((e1[0] - b1[0]) + (e1[1] - b1[1]), gl), 7, 8);
                if (n1 >= K) break;
                tau2 *= 4.f;
            }
            if (n1 < K) { one_pass = true; b1[0] = w[0][0]; e1[0] = w[0][3]; b1[1] = w[1][0]; e1[1] = w[1][3]; }
        }
        {
            const int len1[2] = {e1[0] - b1[0], e1[1] - b1[1]};
            const int total1 = knn_fill_table<8, 2>(lds_run, gl, b1, len1);
            knn_walk16<K, 8, true>(g, qx, qy, qz, gl, lds_run, total1, 0x7f800000u, k);
        }
        if (!one_pass) {
            unsigned hd[K];
#pragma unroll
            for (int i = 0; i < K; ++i) hd[i] = (unsigned)(k[i] >> 32);
            unsigned kth = 0x7f800000u;
#pragma unroll
            for (int t = 0; t < K; ++t) {
                kth = dpp_row_min_u32<8>(hd[0]);
                if (hd[0] == kth) {
#pragma unroll
                    for (int i = 0; i < K - 1; ++i) hd[i] = hd[i + 1];
                    hd[K - 1] = 0x7f800000u;
                }
            }
            const float Bm = __uint_as_float(kth) * 1.0001f;
            int pb[4], pl[4];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bool keepM = dM[s] <= Bm;
                const int kb = keepM ? (dL[s] <= Bm ? w[s][0] : w[s][1]) : w[s][3], ke = keepM ? (dR[s] <= Bm ? w[s][3] : w[s][2]) : w[s][3];
                pb[2 * s] = kb; pl[2 * s] = max(min(ke, b1[s]) - kb, 0);
                pb[2 * s + 1] = max(kb, e1[s]); pl[2 * s + 1] = max(ke - pb[2 * s + 1], 0);
            }
            const int total2 = knn_fill_table<8, 4>(lds_run + KNN_RUN_WORDS, gl, pb, pl);
            if (total2 > 0) knn_walk16<K, 8>(g, qx, qy, qz, gl, lds_run + KNN_RUN_WORDS, total2, kth, k);
        }
    }
    MLH_KSTAGE(3);
    knn_tournament16<K, 8>(k, out);
}

}  // namespace mlh
