// Good-feature selection (ActiveFeatureSelection::goodFeatureMatching, estimator/src/lidarMapper/lidar_mapper.h:229-573).
// The GPU matches every feature and evaluates every 1x6 Jacobian in one pass; this file holds the host-side selection loops
// that consume those rows. The loops are sequential by construction (each greedy pick changes the information matrix the
// next score is computed against), which is why the reference gives them a 20 ms wall-clock budget (lidar_mapper.h:82).
#include "ctx.hpp"
#include "alive_pool.hpp"
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <cmath>
#include <numeric>
#include <queue>
#include <random>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace mlh {

namespace {

// What the selection loops need from a correspondence record is its `valid` word, and what they send back is one verdict per feature: both cross the
// bus as bytes (a Corr is 32 B; at 11-12 k features and four selections per frame the records were a third of the traffic and two host-side block copies).
__global__ void pack_valid_kernel(const Corr *__restrict__ corr, int m, uint8_t *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) out[i] = corr[i].valid != 0;
}
// Corr::valid <- how many residual blocks the selection built on the feature: 0 or 1 -- or `dup_count` for the one feature (`dup_index`) the reference's fps
// loop keeps appending after it has visited everything (fps_after_exhaustion); linearize_kernel weighs the feature's row by that count
__global__ void apply_keep_kernel(Corr *__restrict__ corr, int m, const uint8_t *__restrict__ keep, int dup_index, int dup_count)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) corr[i].valid = (i == dup_index) ? dup_count : int(keep[i]);
}

// Farthest-point sampling (goodFeatureMatching, 'fps': lidar_mapper.h:352-408) as ONE workgroup that keeps every point, its running minimum distance and its
// visited bit in registers (thread t owns points t, t + 1024, ...): an iteration is "distance of every unvisited point to the current one, running minimum,
// arg-max" -- per point exactly the host loop's f32 arithmetic (un-fused products, correctly rounded square root, std::min's operand order), and the arg-max
// takes the LOWEST index among equal distances, as the host loop's strict `>` over ascending j does. The key that is maximised is (distance bits + 1, ~index);
// zero means "no candidate" (NaN coordinates only), and the call is handed to the host loop (count -1). The winner's owner publishes its coordinates and its validity
// through LDS, so the next iteration starts without a global read. Output: the visiting order (start point excluded) until n_use matched points have been
// kept or every point has been visited; the host replays its bookkeeping (selection list, information matrix) along that order.
// wavefront-wide unsigned max / min through the DPP network (row shifts inside the four 16-lane rows, then the two row broadcasts gfx9 has for this):
// the result is lane 63's; six dependent VALU instructions instead of six LDS-crossbar permutes
template <int CTRL, int ROW_MASK> __device__ __forceinline__ unsigned fps_dpp(unsigned identity, unsigned v)
{
    return unsigned(__builtin_amdgcn_update_dpp(int(identity), int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ unsigned fps_row_umax(unsigned v)      // lane 15 of every row: the row's maximum
{
    v = max(v, fps_dpp<0x111, 0xf>(0u, v)); v = max(v, fps_dpp<0x112, 0xf>(0u, v));       // row_shr:1, :2
    v = max(v, fps_dpp<0x114, 0xf>(0u, v)); v = max(v, fps_dpp<0x118, 0xf>(0u, v));       // row_shr:4, :8
    return v;
}
__device__ __forceinline__ unsigned fps_row_umin(unsigned v)
{
    v = min(v, fps_dpp<0x111, 0xf>(~0u, v)); v = min(v, fps_dpp<0x112, 0xf>(~0u, v));
    v = min(v, fps_dpp<0x114, 0xf>(~0u, v)); v = min(v, fps_dpp<0x118, 0xf>(~0u, v));
    return v;
}
__device__ __forceinline__ unsigned fps_wave_umax(unsigned v)
{
    v = fps_row_umax(v);
    v = max(v, fps_dpp<0x142, 0xa>(0u, v));                        // row_bcast:15 into rows 1 and 3
    v = max(v, fps_dpp<0x143, 0xc>(0u, v));                        // row_bcast:31 into rows 2 and 3
    return unsigned(__builtin_amdgcn_readlane(int(v), 63));
}
__device__ __forceinline__ unsigned fps_wave_umin(unsigned v)
{
    v = fps_row_umin(v);
    v = min(v, fps_dpp<0x142, 0xa>(~0u, v));
    v = min(v, fps_dpp<0x143, 0xc>(~0u, v));
    return unsigned(__builtin_amdgcn_readlane(int(v), 63));
}

constexpr int FPS_THREADS = 1024, FPS_PMAX = 16;      // up to 16384 points on the device; longer clouds take the host loop
// The square root is spelled sqrtf: with this toolchain that is the correctly rounded one (bit-equal to the host's on 26 M values incl. every float in
// [1, 4) and denormals), while __fsqrt_rn compiles to the bare v_sqrt_f32 and is one ulp off on 15 % of them (scripts/exp/fp_rounding_check.hip).
// (the per-thread points are sixteen sets of NAMED scalars, expanded by macro: as arrays the compiler kept them in 16-register tuples and moved whole tuples
// around every conditional element update -- 1173 spilled registers)
#define FPS_FOR16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
__global__ __launch_bounds__(FPS_THREADS) void fps_order_kernel(const float4 *__restrict__ pts, const uint8_t *__restrict__ valid, int n, int n_use, int cur0,
                                                                 int *__restrict__ order, int *__restrict__ n_order, const int *__restrict__ gate)
{
    if (gate && gate[0] == 0) return;                               // (behind fps_order_pruned_kernel: only the clouds that kernel leaves alone)
    __shared__ unsigned s_d[FPS_THREADS / 64], s_j[FPS_THREADS / 64];   // per wavefront: (distance bits + 1, 0 = no candidate) and the lowest index that has it
    __shared__ float s_cur[4];
    const int t = threadIdx.x;
    unsigned vis = 0, val = 0;                                      // bit k: point t + 1024 k has been visited (or does not exist) / is a matched feature
#define FPS_LOAD(k)                                                                                                  \
    float px##k, py##k, pz##k, dist##k = 1e5f;                                                                       \
    {                                                                                                                \
        const int j = t + k * FPS_THREADS, jj = j < n ? j : n - 1;                                                   \
        const float4 p = pts[jj];                                                                                    \
        px##k = p.x; py##k = p.y; pz##k = p.z;                                                                       \
        vis |= (j < n ? 0u : 1u) << k;                                                                               \
        val |= ((j < n && valid[jj]) ? 1u : 0u) << k;                                                                \
    }
    FPS_FOR16(FPS_LOAD)
#undef FPS_LOAD
    if ((cur0 & (FPS_THREADS - 1)) == t) vis |= 1u << (cur0 / FPS_THREADS);
    if (t == 0) {
        const float4 p = pts[cur0];
        s_cur[0] = p.x; s_cur[1] = p.y; s_cur[2] = p.z; s_cur[3] = valid[cur0] ? 1.f : 0.f;
    }
    __syncthreads();
    int n_sel = (s_cur[3] != 0.f && n_use > 0) ? 1 : 0, n_visited = 1, n_out = 0;
    while (n_sel < n_use && n_visited < n) {
        const float ox = s_cur[0], oy = s_cur[1], oz = s_cur[2];
        float best_d = -1.f, bx = 0.f, by = 0.f, bz = 0.f;
        int best_k = -1;
#define FPS_STEP(k)                                                                                                  \
        if (!(vis >> k & 1u)) {                                                                                      \
            const float ddx = ox - px##k, ddy = oy - py##k, ddz = oz - pz##k;                                        \
            const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz)));    /* sqrtf, NOT __fsqrt_rn: see below */ \
            const float d2 = (dist##k < d) ? dist##k : d;              /* std::min(d, dist[j]) */                     \
            dist##k = d2;                                                                                            \
            if (d2 > best_d) { best_d = d2; best_k = k; bx = px##k; by = py##k; bz = pz##k; }                          \
        }
        FPS_FOR16(FPS_STEP)
#undef FPS_STEP
        const int best_j = best_k >= 0 ? t + best_k * FPS_THREADS : -1, best_v = best_k >= 0 ? int(val >> best_k & 1u) : 0;
        // arg-max in two keys: the largest distance first, then the lowest index among the points that have it -- inside the wavefront, then over the sixteen
        // wavefronts (one row of lanes reads their pairs back from LDS)
        const unsigned my_d = best_k >= 0 ? __float_as_uint(best_d) + 1u : 0u;
        const unsigned wave_d = fps_wave_umax(my_d);
        const unsigned wave_j = fps_wave_umin((my_d == wave_d && best_k >= 0) ? unsigned(best_j) : ~0u);
        if ((t & 63) == 0) { s_d[t >> 6] = wave_d; s_j[t >> 6] = wave_j; }     // (last round's reads are behind that round's closing barrier)
        __syncthreads();
        const unsigned od = s_d[t & 15], oj = s_j[t & 15];
        const unsigned all_d = unsigned(__builtin_amdgcn_readlane(int(fps_row_umax(od)), 15));
        if (!all_d) { if (t == 0) *n_order = -1; return; }          // only with NaN coordinates (every comparison false): the host loop takes the call
        const int cur = int(__builtin_amdgcn_readlane(int(fps_row_umin(od == all_d ? oj : ~0u)), 15));
        if (cur == best_j) { s_cur[0] = bx; s_cur[1] = by; s_cur[2] = bz; s_cur[3] = best_v ? 1.f : 0.f; }
        if ((cur & (FPS_THREADS - 1)) == t) vis |= 1u << (cur / FPS_THREADS);
        if (t == 0) order[n_out] = cur;
        ++n_out;
        __syncthreads();
        ++n_visited;
        if (s_cur[3] != 0.f) ++n_sel;
    }
    if (t == 0) *n_order = n_out;
}
#undef FPS_FOR16

// ---- fps with exact pruning (round 6). fps_order_kernel above pays 2.8 us per visit whatever the visit changes: every unvisited point's distance to the new pick,
// every round. But `dist[j] = min(d, dist[j])` (lidar_mapper.h:403-404) only CHANGES points closer to the pick than their running minimum, and after k visits that
// is a neighbourhood of ~N / k points. So: the points are put in Morton order (fps_keys / fps_rank / fps_perm kernels: 30-bit keys, rank by counting), a BUCKET is
// the 64 points one wavefront holds in one register slot (lane l of wavefront w, slot k <-> Morton position ((k * 16 + w) * 64 + l: neighbouring buckets -- the ones a
// pick re-measures together -- belong to different wavefronts), and lane k of the wavefront keeps
// slot k's bounding box and its cached arg-max (largest running minimum among its unvisited points, lowest original index among equals, that point's coordinates).
// A round: 16 lanes test their slot's box against the pick (a slot whose box is farther from the pick than its largest running minimum cannot change: for every
// point in it the host's `d2 = min(d, dist[j])` returns dist[j]), the slots that pass are re-measured -- per point exactly the host loop's f32 arithmetic, as above --
// and their arg-max re-cached; the wavefront's best slot, then the best of the 16 wavefronts (one barrier per round: the candidates alternate between two LDS sets).
// The same picks in the same order as fps_order_kernel and the host loop (tests: goodFeatureMatching 'fps' pick for pick); 2.8 -> 1.06 us per visit (5.1-5.5 slots
// re-measured per round on the config-5 frame; what is left is the round's chain: box test -> slot -> wavefront's best -> LDS -> barrier -> decode), and a frame's
// two kinds run side by side (one workgroup each in one launch: good_feature_fps_flush): 49 -> 13.2 ms per config-5 frame.
// The box test is conservative by 1e-4 relative against ~3e-7 of rounding in either distance, and only applied to a squared box distance in the normal range;
// a cloud with ANY non-finite coordinate (fps_keys_kernel's flag) is left to fps_order_kernel, launched behind this one and gated on the same word: its
// comparisons treat NaN as the host's do.
// (A/B, ms per config-5 frame, two alternations: 8 wavefronts x 32 slots 14.05 / 14.08, 16 x 16 13.25 / 13.20 -- a pick's five or so buckets collide less often on
// one wavefront)
#ifndef MLH_FPP_WAVES
#define MLH_FPP_WAVES 16
#endif
constexpr int FPP_WAVES = MLH_FPP_WAVES, FPP_SLOTS = 256 / FPP_WAVES, FPP_THREADS = FPP_WAVES * 64;      // 16 384 points, as the dense kernel
static_assert(FPP_WAVES == 8 || FPP_WAVES == 16, "8 wavefronts x 32 slots or 16 x 16");
static_assert(FPP_WAVES * FPP_SLOTS * 64 == FPS_THREADS * FPS_PMAX, "both fps kernels take the same clouds");
constexpr int FPR_TPB = 256, FPR_PARTS = 8, FPR_TILE = 1024;
#if MLH_FPP_WAVES == 8
#define FPP_FOR32(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) \
                     M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#else
#define FPP_FOR32(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)
#endif

__device__ __forceinline__ unsigned fpp_ord(float f) { const unsigned b = __float_as_uint(f); return b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u); }   // float order -> unsigned order
__device__ __forceinline__ float fpp_unord(unsigned u) { return __uint_as_float(u ^ ((u >> 31) ? 0x80000000u : 0xffffffffu)); }
__device__ __forceinline__ unsigned fpp_spread3(unsigned v)       // 10 bits -> every third bit
{
    v = (v | (v << 16)) & 0x030000ffu; v = (v | (v << 8)) & 0x0300f00fu; v = (v | (v << 4)) & 0x030c30c3u; v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__device__ __forceinline__ float fpp_readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// one workgroup: the cloud's bounding box, then a 30-bit Morton key per point (cubic cells: 1024 along the longest side); aux[0] = "a coordinate is not finite"
__global__ __launch_bounds__(1024) void fps_keys_kernel(const float4 *__restrict__ pts, int n, unsigned *__restrict__ keys, int *__restrict__ rank, int *__restrict__ aux)
{
    __shared__ unsigned s_lo[3][16], s_hi[3][16];
    __shared__ int s_bad;
    const int t = threadIdx.x;
    if (t == 0) s_bad = 0;
    __syncthreads();
    unsigned lo[3] = {~0u, ~0u, ~0u}, hi[3] = {0u, 0u, 0u};
    bool bad = false;
    for (int i = t; i < n; i += 1024) {
        const float4 p = pts[i];
        const float c[3] = {p.x, p.y, p.z};
        if (!(isfinite(p.x) && isfinite(p.y) && isfinite(p.z))) { bad = true; continue; }
        for (int a = 0; a < 3; ++a) { lo[a] = min(lo[a], fpp_ord(c[a])); hi[a] = max(hi[a], fpp_ord(c[a])); }
    }
    if (bad) s_bad = 1;
    for (int a = 0; a < 3; ++a) {
        const unsigned l = fps_wave_umin(lo[a]), h = fps_wave_umax(hi[a]);
        if ((t & 63) == 0) { s_lo[a][t >> 6] = l; s_hi[a][t >> 6] = h; }
    }
    __syncthreads();
    float blo[3], scale = 0.f;
    {
        float ext = 0.f;
        for (int a = 0; a < 3; ++a) {
            unsigned l = ~0u, h = 0u;
            for (int w = 0; w < 16; ++w) { l = min(l, s_lo[a][w]); h = max(h, s_hi[a][w]); }
            blo[a] = fpp_unord(l);
            const float e = fpp_unord(h) - blo[a];
            if (l <= h && e > ext) ext = e;
        }
        if (ext > 0.f && isfinite(ext)) scale = 1023.f / ext;
    }
    for (int i = t; i < n; i += 1024) {
        const float4 p = pts[i];
        unsigned key = 0x3fffffffu;
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            const unsigned qx = unsigned(fminf(fmaxf((p.x - blo[0]) * scale, 0.f), 1023.f)), qy = unsigned(fminf(fmaxf((p.y - blo[1]) * scale, 0.f), 1023.f)),
                           qz = unsigned(fminf(fmaxf((p.z - blo[2]) * scale, 0.f), 1023.f));
            key = fpp_spread3(qx) | (fpp_spread3(qy) << 1) | (fpp_spread3(qz) << 2);
        }
        keys[i] = key;
        rank[i] = 0;
    }
    if (t < 16) aux[t] = t == 0 ? s_bad : 0;
}

// rank[i] += the number of points of this launch row's share that sort before point i (key, then index): grid (ceil(n / 256), FPR_PARTS)
__global__ __launch_bounds__(FPR_TPB) void fps_rank_kernel(const unsigned *__restrict__ keys, int n, int *__restrict__ rank)
{
    __shared__ unsigned s_k[FPR_TILE];
    const int i = blockIdx.x * FPR_TPB + threadIdx.x;
    const unsigned ki = i < n ? keys[i] : 0u;
    const int share = (n + FPR_PARTS - 1) / FPR_PARTS, j0 = int(blockIdx.y) * share, j1 = min(n, j0 + share);
    int cnt = 0;
    for (int base = j0; base < j1; base += FPR_TILE) {
        const int len = min(FPR_TILE, j1 - base);
        __syncthreads();
        for (int q = threadIdx.x; q < len; q += FPR_TPB) s_k[q] = keys[base + q];
        __syncthreads();
        for (int q = 0; q < len; ++q) cnt += (s_k[q] < ki + ((base + q) < i ? 1u : 0u)) ? 1 : 0;      // (key_j, j) < (key_i, i); keys are 30 bits
    }
    if (i < n && cnt) atomicAdd(rank + i, cnt);
}
__global__ void fps_perm_kernel(const int *__restrict__ rank, int n, int *__restrict__ perm)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) perm[rank[i]] = i;
}

// signed max / min over a wavefront (the keys below are the BITS of a non-negative float, or of -1.0f for "not a candidate": ordered as signed integers)
__device__ __forceinline__ int fpp_wave_imax(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false)); v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false)); v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int fpp_row_imax(int v)                   // lane 15 of every row
{
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x111, 0xf, 0xf, false)); v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x114, 0xf, 0xf, false)); v = max(v, __builtin_amdgcn_update_dpp(INT_MIN, v, 0x118, 0xf, 0xf, false));
    return v;
}
constexpr int FPP_NOT = int(0xbf800000u);                            // the bits of -1.0f

struct FpsJob {
    const float4 *pts;      // the kind's features, original order
    const uint8_t *valid;   // 1 = matched
    const int *perm;        // Morton position -> original index
    const int *aux;         // aux[0] != 0: a coordinate is not finite -- this kernel leaves the cloud to fps_order_kernel (launched behind it, gated on the same word)
    int *order, *n_order;   // out: visiting order (start point excluded), its length (-1: the host loop takes the call)
    int n, n_use, cur0;
};
struct FpsJobs { FpsJob j[2]; };        // one workgroup per job: the two kinds of a frame side by side (n == 0: no job)

// A slot's arg-max after its points were re-measured -> its owner lane. key: the lane's running minimum as float bits (FPP_NOT: visited or no such point).
// Equal keys (features on a lattice, repeated features) are decided by the lowest ORIGINAL index, as the host loop's strict `>` over ascending j decides them; the
// indices live in LDS and are only read on that path.
__device__ __forceinline__ void fpp_slot_finish(int key, const unsigned short *id_p, float px, float py, float pz, unsigned v, int k, int lane, int &bkey, float &bx,
                                                float &by, float &bz, unsigned &bloc)
{
    const int mk = fpp_wave_imax(key);
    unsigned long long wb = __ballot(key == mk);
    if (mk >= 0 && __popcll(wb) != 1) {
        const unsigned id = *id_p;
        const unsigned mj = fps_wave_umin(key == mk ? id : ~0u);
        wb = __ballot(key == mk && id == mj);
    }
    const int wl = mk >= 0 ? __ffsll(wb) - 1 : 0;
    const float wx = fpp_readlane(px, wl), wy = fpp_readlane(py, wl), wz = fpp_readlane(pz, wl);
    const unsigned wv = unsigned(__builtin_amdgcn_readlane(int(v), wl));
    if (lane == k) { bkey = mk; bx = wx; by = wy; bz = wz; bloc = unsigned(k << 6 | wl) | (wv << 31); }
}

__global__ __launch_bounds__(FPP_THREADS) void fps_order_pruned_kernel(FpsJobs J)
{
    __shared__ uint4 s_c[2][FPP_WAVES][2];                           // a wavefront's candidate: {key, location | matched << 31, -, -}, {x, y, z, -}; two sets
    __shared__ unsigned short s_id[FPP_WAVES * FPP_SLOTS * 64];      // original index by Morton position    } 96 KB of LDS for 64 registers a lane does not have
    __shared__ float s_dist[FPP_WAVES * FPP_SLOTS * 64];             // running minimum by Morton position   } (-1: visited, or no such point)
    const FpsJob &P = J.j[blockIdx.x];
    const int n = P.n, n_use = P.n_use, cur0 = P.cur0;
    if (n <= 0 || P.aux[0] != 0) return;
    const float4 *__restrict__ pts = P.pts;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const unsigned short *my_id = s_id + w * 64 + lane;             // slot k: + k * FPP_WAVES * 64
    float *my_dist = s_dist + w * 64 + lane;
    unsigned val = 0;                   // bit k: the point of slot k is a matched feature
#define FPP_LOAD(k)                                                                                                  \
    float px##k, py##k, pz##k;                                                                                       \
    {                                                                                                                \
        const int pos = (k * FPP_WAVES + w) * 64 + lane, j = pos < n ? P.perm[pos] : -1, jj = j >= 0 ? j : cur0;     \
        const float4 p = pts[jj];                                                                                    \
        px##k = p.x; py##k = p.y; pz##k = p.z; s_id[pos] = (unsigned short)(j);                                      \
        s_dist[pos] = (j >= 0 && j != cur0) ? 1e5f : -1.f;                                                           \
        val |= ((j >= 0 && P.valid[jj]) ? 1u : 0u) << k;                                                             \
    }
    FPP_FOR32(FPP_LOAD)
#undef FPP_LOAD
    // lane k: slot k's box and cached arg-max (bkey: the largest running minimum as float bits, FPP_NOT = no candidate left; before the first round: +inf, "measure me")
    float lox = 0.f, loy = 0.f, loz = 0.f, hix = 0.f, hiy = 0.f, hiz = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    int bkey = FPP_NOT;
    unsigned bloc = 0u;
#define FPP_BOX(k)                                                                                                   \
    {                                                                                                                \
        const int pos = (k * FPP_WAVES + w) * 64 + lane;                                                             \
        const bool ex = pos < n;                                                                                     \
        const unsigned ax = fps_wave_umin(ex ? fpp_ord(px##k) : ~0u), bxx = fps_wave_umax(ex ? fpp_ord(px##k) : 0u); \
        const unsigned ay = fps_wave_umin(ex ? fpp_ord(py##k) : ~0u), byy = fps_wave_umax(ex ? fpp_ord(py##k) : 0u); \
        const unsigned az = fps_wave_umin(ex ? fpp_ord(pz##k) : ~0u), bzz = fps_wave_umax(ex ? fpp_ord(pz##k) : 0u); \
        const bool any = __ballot(ex && s_id[pos] != (unsigned short)(cur0)) != 0ull;                                \
        if (lane == k) {                                                                                             \
            lox = fpp_unord(ax); hix = fpp_unord(bxx); loy = fpp_unord(ay); hiy = fpp_unord(byy); loz = fpp_unord(az); hiz = fpp_unord(bzz); \
            bkey = any ? 0x7f800000 : FPP_NOT;                                                                       \
        }                                                                                                            \
    }
    FPP_FOR32(FPP_BOX)
#undef FPP_BOX
    float ox, oy, oz;
    int n_sel;
    {
        const float4 p = pts[cur0];
        ox = p.x; oy = p.y; oz = p.z;
        n_sel = (P.valid[cur0] && n_use > 0) ? 1 : 0;
    }
    int n_visited = 1, n_out = 0, par = 0;
#ifdef MLH_FPS_STATS
    unsigned long long ck[6] = {0, 0, 0, 0, 0, 0}, c_prev = __builtin_readcyclecounter();
#define FPP_CK(i) do { const unsigned long long c_now = __builtin_readcyclecounter(); ck[i] += c_now - c_prev; c_prev = c_now; } while (0)
#else
#define FPP_CK(i) do { } while (0)
#endif
    while (n_sel < n_use && n_visited < n) {
        unsigned act;
        {
            const float ax = fmaxf(fmaxf(lox - ox, ox - hix), 0.f), ay = fmaxf(fmaxf(loy - oy, oy - hiy), 0.f), az = fmaxf(fmaxf(loz - oz, oz - hiz), 0.f);
            const float lb2 = ax * ax + ay * ay + az * az;
            const float md = __int_as_float(bkey);
            const bool pruned = lb2 > 1e-30f && lb2 * 0.9999f > md * md;
            act = unsigned(__ballot(lane < FPP_SLOTS && bkey >= 0 && !pruned));
        }
        FPP_CK(0);
#ifdef MLH_FPS_STATS
        if (lane == 0) { atomicAdd(const_cast<int *>(P.aux) + 1, __popc(act)); atomicMax(const_cast<int *>(P.aux) + 3 + w, __popc(act)); if (t == 0) atomicAdd(const_cast<int *>(P.aux) + 2, 1); }
#endif
        while (act) {                                                // (uniform)
            const int k = __ffs(act) - 1;
            act &= act - 1u;
            switch (k) {
#define FPP_CASE(k)                                                                                                  \
            case k: {                                                                                                \
                /* (the empty asm keeps the 32 slots' square roots INSIDE their cases: hoisted out of this loop as loop-invariant code they cost */ \
                /*  3 us per round -- every slot measured every round, the work this kernel exists to skip) */       \
                float qx = ox, qy = oy, qz = oz;                                                                     \
                asm volatile("" : "+v"(qx), "+v"(qy), "+v"(qz));                                                     \
                const float dk = my_dist[k * FPP_WAVES * 64];                                                        \
                const float ddx = qx - px##k, ddy = qy - py##k, ddz = qz - pz##k;                                    \
                const float d = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(ddx, ddx), __fmul_rn(ddy, ddy)), __fmul_rn(ddz, ddz)));    /* sqrtf: see fps_order_kernel */ \
                const float d2 = (dk < d) ? dk : d;                        /* std::min(d, dist[j]); a visited point keeps its -1 */ \
                my_dist[k * FPP_WAVES * 64] = d2;                                                                    \
                fpp_slot_finish(__float_as_int(d2), my_id + k * FPP_WAVES * 64, px##k, py##k, pz##k, val >> k & 1u, k, lane, bkey, bx, by, bz, bloc); \
            } break;
            FPP_FOR32(FPP_CASE)
#undef FPP_CASE
            default: break;
            }
        }
        FPP_CK(1);
        // the wavefront's best slot -> LDS; then the best of the wavefronts (every wavefront works it out for itself)
        const int k1 = lane < FPP_SLOTS ? bkey : FPP_NOT;
        const int wk = fpp_wave_imax(k1);
        unsigned long long sb = __ballot(k1 == wk);
        if (wk >= 0 && __popcll(sb) != 1) {
            const unsigned id = (k1 == wk) ? s_id[((lane & (FPP_SLOTS - 1)) * FPP_WAVES + w) * 64 + (bloc & 63u)] : ~0u;
            const unsigned wj = fps_wave_umin(id);
            sb = __ballot(k1 == wk && id == wj);
        }
        const int sl = wk >= 0 ? __ffsll(sb) - 1 : 0;
        if (lane == sl) {
            s_c[par][w][0] = make_uint4(unsigned(wk), bloc, 0u, 0u);
            s_c[par][w][1] = make_uint4(__float_as_uint(bx), __float_as_uint(by), __float_as_uint(bz), 0u);
        }
        FPP_CK(2);
        __syncthreads();
        FPP_CK(3);
        const bool rd = lane < FPP_WAVES;
        const uint4 c0 = s_c[par][rd ? lane : 0][0];
        uint4 c1 = s_c[par][rd ? lane : 0][1];
        asm volatile("" : "+v"(c1.x), "+v"(c1.y), "+v"(c1.z));      // (both reads in flight together: left alone, the second is issued behind the key's reduction)
        const int gk_l = rd ? int(c0.x) : FPP_NOT;
        const int gk = __builtin_amdgcn_readlane(fpp_row_imax(gk_l), 15);
        if (gk < 0) { if (t == 0) *P.n_order = -1; return; }        // nothing left to visit although the loop goes on: the host loop takes the call (as fps_order_kernel)
        unsigned long long gb = __ballot(rd && gk_l == gk);
        if (__popcll(gb) != 1) {
            const unsigned id = (rd && gk_l == gk) ? s_id[(((c0.y >> 6) & 31u) * FPP_WAVES + lane) * 64 + (c0.y & 63u)] : ~0u;
            const unsigned gj2 = unsigned(__builtin_amdgcn_readlane(int(fps_row_umin(id)), 15));
            gb = __ballot(rd && gk_l == gk && id == gj2);
        }
        const int ww = __ffsll(gb) - 1;
        const unsigned gloc = unsigned(__builtin_amdgcn_readlane(int(c0.y), ww));
        ox = fpp_readlane(__uint_as_float(c1.x), ww); oy = fpp_readlane(__uint_as_float(c1.y), ww); oz = fpp_readlane(__uint_as_float(c1.z), ww);
        const int gpos = (int((gloc >> 6) & 31u) * FPP_WAVES + ww) * 64 + int(gloc & 63u);
        if (t == 0) P.order[n_out] = int(s_id[gpos]);               // (nobody waits for this store)
        if (w == ww && lane == int(gloc & 63u)) s_dist[gpos] = -1.f;      // visited (read again by this wavefront only)
        ++n_out;
        ++n_visited;
        if (gloc >> 31) ++n_sel;
        par ^= 1;
        FPP_CK(4);
    }
#ifdef MLH_FPS_STATS
    if (lane == 0) for (int i = 0; i < 5; ++i) const_cast<int *>(P.aux)[16 + 8 * w + i] = int(ck[i] / (unsigned long long)(n_out > 0 ? n_out : 1));
#endif
    if (t == 0) *P.n_order = n_out;
}
#undef FPP_CK
#undef FPP_FOR32

struct Rows {               // per-feature results of the GPU pass, copied out of the context's pinned staging block
    const uint8_t *valid = nullptr;   // Corr::valid != 0
    const double *J = nullptr;  // m x 6
    const float4 *pts = nullptr;
    size_t m = 0;
    size_t size() const { return m; }
    bool matched(size_t i) const { return valid[i] != 0; }
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

__attribute__((always_inline)) inline size_t draw(std::mt19937 &rng, size_t lo, size_t hi)
{
    std::uniform_int_distribution<size_t> d(lo, hi);   // RandomGeneratorInt<size_t>::geneRandUniform re-creates the distribution per draw
    return d(rng);
}


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

// The reference's loop has no "every point visited" exit (lidar_mapper.h:391-399: that test is commented out). With everything visited its scan leaves best_j at the
// initial value 1, so feature 1 is matched again on every further round: matched, it is appended -- and its J^T J added -- again and again until the count is
// reached; unmatched, the loop spins until the 20 ms cut-off and returns what it has. Restated (the spin as an immediate return); a single feature (points[1]
// does not exist) stops.
void fps_after_exhaustion(const Rows &R, size_t n_use, std::vector<size_t> &sel, double H[36])
{
    if (sel.size() >= n_use || R.size() < 2 || !R.matched(1)) return;
    while (sel.size() < n_use) { rank1_update(H, R.jaco(1)); sel.push_back(1); }
}

void select_fps(const Rows &R, size_t n_use, size_t cur, std::vector<size_t> &sel, double H[36])
{
    const size_t n = R.size();
    if (n == 0) return;
    std::vector<char> visited(n, 0);
    visited[cur] = 1;                              // `cur` = the starting point, drawn by the caller (rgi_.geneRandUniform(0, size - 1), lidar_mapper.h:356)
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
    fps_after_exhaustion(R, n_use, sel, H);
}

// the same bookkeeping along a visiting order the device produced (fps_order_kernel): who is kept, and the information matrix in pick order
void select_fps_replay(const Rows &R, size_t n_use, size_t start, const int *order, size_t n_order, std::vector<size_t> &sel, double H[36])
{
    if (R.matched(start) && n_use > 0) sel.push_back(start);
    for (size_t i = 0; i < n_order && sel.size() < n_use; ++i) {
        const size_t cur = size_t(order[i]);
        if (R.matched(cur)) { rank1_update(H, R.jaco(cur)); sel.push_back(cur); }
    }
    fps_after_exhaustion(R, n_use, sel, H);      // (the order ends early only when every point has been visited)
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
    auto exact_score = [&](size_t q) __attribute__((noinline)) { double Ht[36]; std::copy(H, H + 36, Ht); rank1_update(Ht, R.jaco(q)); return logdet_cholesky6(Ht); };
    // q = j H^-1 j^T, and H^-1 j^T on the side. H^-1 is symmetric to the bit (spd_inverse6 and the rank-1 update below both are), so with SSE2 the product is
    // accumulated a COLUMN at a time, two rows per register: every row's sum still adds its six terms in the order c = 0..5 (same bits as the scalar loop)
    auto quad = [&](const double *j, double *Hj) __attribute__((always_inline)) {   // (left out of line, the loop ran ~15 % slower)
#if defined(__SSE2__)
        __m128d jc = _mm_set1_pd(j[0]);
        __m128d a0 = _mm_mul_pd(_mm_loadu_pd(Hinv), jc), a1 = _mm_mul_pd(_mm_loadu_pd(Hinv + 2), jc), a2 = _mm_mul_pd(_mm_loadu_pd(Hinv + 4), jc);
        for (int c = 1; c < 6; ++c) {
            jc = _mm_set1_pd(j[c]);
            a0 = _mm_add_pd(a0, _mm_mul_pd(_mm_loadu_pd(Hinv + 6 * c), jc));
            a1 = _mm_add_pd(a1, _mm_mul_pd(_mm_loadu_pd(Hinv + 6 * c + 2), jc));
            a2 = _mm_add_pd(a2, _mm_mul_pd(_mm_loadu_pd(Hinv + 6 * c + 4), jc));
        }
        _mm_storeu_pd(Hj, a0); _mm_storeu_pd(Hj + 2, a1); _mm_storeu_pd(Hj + 4, a2);
        double q = 0.0;
        for (int r = 0; r < 6; ++r) q += j[r] * Hj[r];
        return q;
#else
        double q = 0.0;
        for (int r = 0; r < 6; ++r) { double t = 0.0; for (int c = 0; c < 6; ++c) t += Hinv[r * 6 + c] * j[c]; Hj[r] = t; q += j[r] * t; }
        return q;
#endif
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
                    if (nh * ni > 1e10 || (sel.size() & 255) == 0) {     // kappa = sqrt(nh ni) > 1e5
                        have_inv = spd_inverse6(H, Hinv);       // refresh: keeps the update's rounding from accumulating
                        if (have_inv) { ni = 0.0; for (int k = 0; k < 36; ++k) ni += Hinv[k] * Hinv[k]; }
                    }
                    replay_tol = std::max(1e-10, 64.0 * 2.220446049250313e-16 * std::sqrt(nh * ni));
                }
                break;
            }
        }
        if (retries >= max_retry || lost) break;
    }
}

}  // namespace

// The farthest-point loops of the kinds staged with `defer_fps` (or of the one kind staged without), one workgroup each in ONE launch -- a loop is a chain of
// dependent rounds on one compute unit, so two of them side by side cost the longer one --, then per kind: the dense kernel behind a gate (it runs only where the
// pruned one declined: a cloud with a non-finite coordinate), the copy of the visiting order, the marker.
int good_feature_fps_flush(mlh_ctx *ctx)
{
    FpsJobs J;
    std::memset(&J, 0, sizeof(J));
    int nj = 0, kinds[2] = {-1, -1};
    const bool fps_dense = std::getenv("MLH_FPS_DENSE") != nullptr;        // (A/B / tests: the round-4 kernel -- every point re-measured every round -- on every cloud)
    for (int kind : {MLH_CORNER, MLH_SURF}) {
        if (!ctx->fps_pending[kind].active) continue;
        FeatSet &f = ctx->feat[kind];
        const int m = ctx->fps_pending[kind].m;
        FpsJob &j = J.j[nj];
        j.pts = f.pts.as<float4>(); j.valid = f.flag8.as<uint8_t>(); j.perm = f.fps_work.as<int>() + 2 * size_t(m); j.aux = j.perm + m;
        j.order = f.fps_order.as<int>() + 1; j.n_order = f.fps_order.as<int>();
        j.n = m; j.n_use = ctx->fps_pending[kind].n_use; j.cur0 = ctx->fps_pending[kind].cur0;
        kinds[nj++] = kind;
    }
    if (!nj) return MLH_OK;
    if (!fps_dense) MLH_LAUNCH(fps_order_pruned_kernel, dim3(unsigned(nj)), dim3(FPP_THREADS), 0, ctx->stream, J);
    for (int q = 0; q < nj; ++q) {
        const FpsJob &j = J.j[q];
        MLH_LAUNCH(fps_order_kernel, dim3(1), dim3(FPS_THREADS), 0, ctx->stream, j.pts, j.valid, j.n, j.n_use, j.cur0, j.order, j.n_order, fps_dense ? nullptr : j.aux);
    }
    MLH_HIP(ctx, hipGetLastError());
#ifdef MLH_FPS_STATS
    for (int q = 0; q < nj; ++q) {
        int h[16] = {0}, hc[64];
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(h, J.j[q].aux, sizeof(h), hipMemcpyDeviceToHost);
        (void)hipMemcpy(hc, J.j[q].aux + 16, sizeof(hc), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "fps stats: m %d nonfinite %d rounds %d slots re-measured %d (%.2f per round, all waves); per-wave max in a round:", J.j[q].n, h[0], h[2], h[1], h[2] ? double(h[1]) / h[2] : 0.0);
        for (int u = 0; u < FPP_WAVES; ++u) std::fprintf(stderr, " %d", h[3 + u]);
        std::fprintf(stderr, " | cycles per round (test | slots | wave best | barrier | decode) by wavefront:");
        for (int u = 0; u < FPP_WAVES; ++u) std::fprintf(stderr, "  %d %d %d %d %d", hc[8 * u], hc[8 * u + 1], hc[8 * u + 2], hc[8 * u + 3], hc[8 * u + 4]);
        std::fprintf(stderr, "\n");
    }
#endif
    for (int q = 0; q < nj; ++q) {
        const int kind = kinds[q];
        FeatSet &f = ctx->feat[kind];
        MLH_HIP(ctx, hipMemcpyAsync(ctx->fps_pending[kind].host_dst, f.fps_order.p, sizeof(int) * (size_t(J.j[q].n) + 1), hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, stream_flag_post(ctx, &ctx->select_seq[kind]));
        ctx->select_staged[kind] = true;
        ctx->fps_pending[kind].active = 0;
    }
    return MLH_OK;
}

// One goodFeatureMatching call, in two halves. The device-side solver state must already hold the pose (SolverState::x).
// Stage: the dense pass of one kind and the copies of its rows into that kind's pinned block, enqueued; a marker behind them.
int good_feature_stage(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, float min_match_sq_dis, float min_plane_dis, bool defer_fps)
{
    FeatSet &f = ctx->feat[kind];
    MatchArgs a;
    a.kind_mask = 1 << kind;
    a.flags = MLH_FLAG_WITH_UA | MLH_FLAG_NO_LOSS;   // extractCov(point) weight, rows not loss-corrected (lidar_mapper.h:162-164)
    a.min_match_sq_dis = min_match_sq_dis; a.min_plane_dis = min_plane_dis;
    a.huber_delta = 0.0; a.dense = true; a.pose_sel = 0;
    int rc = match_launch(ctx, a);
    if (rc) return rc;
    const size_t m = size_t(f.m);
    // pinned staging (grow-only, owned by the context, one block per kind): [J 6m][pts m][valid m bytes, padded][keep m bytes, padded][fps: count, order m]
    const size_t mp = (m + 63) & ~size_t(63);
    const size_t off_p = sizeof(double) * 6 * m, off_v = off_p + sizeof(float4) * m, off_k = off_v + mp, off_o = off_k + mp, need = off_o + sizeof(int) * (m + 1);
    if (need > ctx->select_host_cap[kind]) {
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));          // a copy of an earlier call may still be using the old block
        if (ctx->select_host[kind]) (void)hipHostFree(ctx->select_host[kind]);
        ctx->select_host[kind] = nullptr; ctx->select_host_cap[kind] = 0;
        MLH_HIP(ctx, hipHostMalloc(&ctx->select_host[kind], need + need / 4, hipHostMallocDefault));
        ctx->select_host_cap[kind] = need + need / 4;
    }
    MLH_HIP(ctx, f.flag8.ensure(mp));
    MLH_LAUNCH(pack_valid_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, ctx->stream, f.corr.as<Corr>(), int(m), f.flag8.as<uint8_t>());
    MLH_HIP(ctx, hipGetLastError());
    char *hb = static_cast<char *>(ctx->select_host[kind]);
    MLH_HIP(ctx, hipMemcpyAsync(hb + off_v, f.flag8.p, m, hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(hb, f.J.p, sizeof(double) * 6 * m, hipMemcpyDeviceToHost, ctx->stream));
    ctx->select_fps_start[kind] = -1;
    ctx->fps_pending[kind].active = 0;               // (a loop an earlier call staged and never launched -- it failed in between -- is void)
    if (method == MLH_GF_FPS && m > 0) {
        // the one number this method draws (the starting point, lidar_mapper.h:356) is drawn here, in call order: corner before surf, as the finishes will run
        const size_t cur0 = draw(rng, 0, m - 1);
        ctx->select_fps_start[kind] = long(cur0);
        const bool fps_on_host = std::getenv("MLH_FPS_HOST") != nullptr;          // (measurement / tests only: the host loop on every call)
        if (m <= size_t(FPS_THREADS) * FPS_PMAX && !fps_on_host) {
            MLH_HIP(ctx, f.fps_order.ensure(sizeof(int) * (m + 1)));
            // Morton order of the cloud (keys, rank by counting, permutation) now; the loop itself now or -- `defer_fps`: a frame's two kinds -- side by side in
            // one launch (good_feature_fps_flush): [keys m][rank m][perm m][aux 96]
            MLH_HIP(ctx, f.fps_work.ensure(sizeof(int) * (3 * m + 96)));
            unsigned *keys = f.fps_work.as<unsigned>();
            int *rank = f.fps_work.as<int>() + m, *perm = rank + m, *aux = perm + m;
            MLH_LAUNCH(fps_keys_kernel, dim3(1), dim3(1024), 0, ctx->stream, f.pts.as<float4>(), int(m), keys, rank, aux);
            MLH_LAUNCH(fps_rank_kernel, dim3(unsigned((m + FPR_TPB - 1) / FPR_TPB), FPR_PARTS), dim3(FPR_TPB), 0, ctx->stream, keys, int(m), rank);
            MLH_LAUNCH(fps_perm_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, ctx->stream, rank, int(m), perm);
            MLH_HIP(ctx, hipGetLastError());
            ctx->fps_pending[kind].active = 1; ctx->fps_pending[kind].m = int(m); ctx->fps_pending[kind].n_use = int(static_cast<size_t>(m * ratio));
            ctx->fps_pending[kind].cur0 = int(cur0); ctx->fps_pending[kind].host_dst = hb + off_o;
            if (defer_fps) return MLH_OK;                            // (copy of the order, marker, `staged`: good_feature_fps_flush)
            return good_feature_fps_flush(ctx);
        } else {
            MLH_HIP(ctx, hipMemcpyAsync(hb + off_p, f.pts.p, sizeof(float4) * m, hipMemcpyDeviceToHost, ctx->stream));   // too long for one workgroup's registers: host loop
            *reinterpret_cast<int *>(hb + off_o) = -1;
        }
    }
    MLH_HIP(ctx, stream_flag_post(ctx, &ctx->select_seq[kind]));
    ctx->select_staged[kind] = true;
    return MLH_OK;
}

// Finish: wait for that kind's marker, run the selection loop, send the flags back (enqueued: whatever reads them is behind on the stream).
int good_feature_finish(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, std::vector<int32_t> &sel_out, double H[36],
                        uint8_t *matched_out)
{
    FeatSet &f = ctx->feat[kind];
    if (!ctx->select_staged[kind]) return fail(ctx, MLH_ERR_STATE, "good_feature_finish without good_feature_stage");
    ctx->select_staged[kind] = false;
    static const bool timing = std::getenv("MLH_SEL_TIMING") != nullptr;
    const auto tc0 = std::chrono::steady_clock::now();
    MLH_HIP(ctx, stream_flag_wait(ctx, ctx->select_seq[kind]));
    Rows R;
    const size_t m = size_t(f.m);
    const size_t mp = (m + 63) & ~size_t(63);
    const size_t off_p = sizeof(double) * 6 * m, off_v = off_p + sizeof(float4) * m, off_k = off_v + mp, off_o = off_k + mp;
    char *hb = static_cast<char *>(ctx->select_host[kind]);
    R.m = m;
    const int *fps_dev = reinterpret_cast<const int *>(hb + off_o);                     // [count or -1][visiting order]
    const bool fps_host = method == MLH_GF_FPS && m > 0 && fps_dev[0] < 0;
    if (fps_host && m <= size_t(FPS_THREADS) * FPS_PMAX && !std::getenv("MLH_FPS_HOST")) {                              // the kernel gave up (NaN coordinates): the points were not sent along
        MLH_HIP(ctx, hipMemcpyAsync(hb + off_p, f.pts.p, sizeof(float4) * m, hipMemcpyDeviceToHost, ctx->stream));
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    // The selection loops jump around in these rows (a pool look-up decides which one comes next). Pinned host memory is mapped so that the CPU does not
    // cache it: read in place, every access is a trip to DRAM -- the `rnd` loop, which scores nothing, took 0.6 / 1.0 ms per call that way, 0.12 / 0.28 ms on an
    // ordinary copy; a bulk copy out of the pinned block runs at ~30 GB/s. So: DMA into the pinned block, one memcpy into the context's
    // cacheable block, loops on that (config 5, gd_fix: 7.2 -> 4.05 ms per frame; profiles/r03_gfbench.txt).
    ctx->select_rows[kind].resize(off_k);
    char *cb = ctx->select_rows[kind].data();
    std::memcpy(cb, hb, fps_host ? off_v : off_p);
    std::memcpy(cb + off_v, hb + off_v, m);
    R.valid = reinterpret_cast<const uint8_t *>(cb + off_v); R.J = reinterpret_cast<const double *>(cb); R.pts = reinterpret_cast<const float4 *>(cb + off_p);
    if (matched_out) for (size_t i = 0; i < m; ++i) matched_out[i] = R.matched(i) ? 1 : 0;

    const auto tc1 = std::chrono::steady_clock::now();
    const size_t n_use = static_cast<size_t>(m * ratio);   // num_use_features (lidar_mapper.h:247)
    std::vector<size_t> sel;
    sel.reserve(method == MLH_GF_WO ? m : n_use);
    switch (method) {
        case MLH_GF_WO: select_wo_gf(R, sel, H); break;
        case MLH_GF_RND: select_rnd(R, n_use, rng, sel, H); break;
        case MLH_GF_FPS:
            if (m == 0) break;
            if (fps_host) select_fps(R, n_use, size_t(ctx->select_fps_start[kind]), sel, H);
            else select_fps_replay(R, n_use, size_t(ctx->select_fps_start[kind]), fps_dev + 1, size_t(fps_dev[0]), sel, H);
            break;
        case MLH_GF_GD_FIX:
        case MLH_GF_GD_FLOAT: select_greedy(R, n_use, rng, sel, H); break;
        default: return fail(ctx, MLH_ERR_INVALID, "unknown gf_method");
    }
    const auto tc2 = std::chrono::steady_clock::now();
    // keep only the selected correspondences valid on the device: one verdict byte per feature goes back, a launch writes them into the records
    if (method != MLH_GF_WO) {
        uint8_t *keep = reinterpret_cast<uint8_t *>(hb + off_k);     // (sequential writes into the pinned block: the mapping is fine for those)
        std::memset(keep, 0, m);
        int dup_count = 0;                                            // only feature 1 can be picked more than once (fps_after_exhaustion)
        for (size_t i : sel) { keep[i] = 1; dup_count += (i == 1) ? 1 : 0; }
        MLH_HIP(ctx, hipMemcpyAsync(f.flag8.p, keep, m, hipMemcpyHostToDevice, ctx->stream));
        MLH_LAUNCH(apply_keep_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, ctx->stream, f.corr.as<Corr>(), int(m), f.flag8.as<uint8_t>(),
                           dup_count > 1 ? 1 : -1, dup_count);
        MLH_HIP(ctx, hipGetLastError());
    }
    sel_out.assign(sel.begin(), sel.end());
    if (timing) {
        const auto tc3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::micro>(y - x).count(); };
        std::fprintf(stderr, "[good_feature_finish] kind %d m %zu: wait for the rows + copy out of the pinned block %.0f us | selection loop %.0f us (%zu picks) | flags enqueued %.0f us\n", kind, m,
                     us(tc0, tc1), us(tc1, tc2), sel.size(), us(tc2, tc3));
    }
    return MLH_OK;
}

// Estimator::goodFeatureMatching's selection loop (estimator.cpp:1414-1512) over rows the device evaluated (odom.hip: pure_odom_feature_rows): the mapper's
// stochastic-greedy loop with the estimator's limits and its one structural difference --
//   - MAX_RANDOM_QUEUE_TIME is 10 here (estimator.h:63), and ten failed draws do NOT end the selection (the `break` behind the "early termination" message is missing,
//     cpp:1505-1509): the outer loop starts over with an empty heap; whatever the heap held keeps its stamp and cannot be drawn again before the next pick;
//   - only the 7 ms wall-clock cut-off ends a loop in which no draw can succeed any more; restated without the clock: it ends when no pool entry is left that
//     this round may still draw (n_fresh == 0);
//   - sub_mat_H starts at 1e-6 I inside the function and is not returned; the scores are the reference's own arithmetic (logDet through Cholesky), pushed through
//     the reference's own container (std::priority_queue with std::less on the score).
void select_greedy_odom(const Rows &R, size_t n_use, std::mt19937 &rng, std::vector<size_t> &sel)
{
    const size_t n_all = R.size();
    if (n_use == 0 || n_all == 0) return;
    AlivePool pool(n_all);
    std::vector<int> stamp(n_all, -1);
    const size_t max_retry = 10;
    const size_t subset = static_cast<size_t>(1.0 * n_all / n_use);
    double H[36];
    for (int i = 0; i < 36; ++i) H[i] = (i % 7 == 0) ? 1e-6 : 0.0;
    size_t n_fresh = n_all;                        // alive entries whose stamp is below the current number of picks
    while (sel.size() < n_use && !pool.empty()) {
        std::priority_queue<Scored> heap;
        size_t retries = 0;
        while (!pool.empty()) {
            retries = 0;
            size_t q = 0;
            while (retries < max_retry) {
                const size_t j = draw(rng, 0, pool.size() - 1);
                q = pool.at(j);
                if (stamp[q] < int(sel.size())) { stamp[q] = int(sel.size()); --n_fresh; break; }
                ++retries;
            }
            if (retries >= max_retry) break;
            if (!R.matched(q)) { pool.erase_index(q); continue; }
            if (subset <= 1) {
                // subsets of one (every ratio above 0.5, ODOM_GF_RATIO = 0.8 among them): the heap's top is the feature just pushed whatever its score, and
                // sub_mat_H is not returned -- the pick is the draw; neither the score nor the rows are needed
                pool.erase_index(q);
                sel.push_back(q);
                n_fresh = pool.size();
                break;
            }
            double Ht[36];
            std::copy(H, H + 36, Ht);
            rank1_update(Ht, R.jaco(q));
            heap.push(Scored{q, logdet_cholesky6(Ht)});
            if (heap.size() >= subset) {
                const size_t top = heap.top().idx;
                if (!pool.contains(top)) break;
                rank1_update(H, R.jaco(top));
                pool.erase_index(top);
                sel.push_back(top);
                n_fresh = pool.size();
                break;
            }
        }
        if (retries >= max_retry && n_fresh == 0) break;
    }
}

// The odometry's selection in front of mlh_pure_odom_add_matches' append: rows and validity come back from the device (pure_odom_feature_rows has been enqueued), the
// loop runs on the host, the verdicts go back into Corr::valid -- the append then packs exactly the selected correspondences. gf_ratio is the reference's float
// ODOM_GF_RATIO widened to double (num_use = size_t(m * double(ratio)), estimator.cpp:1366-1367).
int odom_good_feature_select(mlh_ctx *ctx, int kind, float gf_ratio, std::mt19937 &rng, std::vector<int32_t> &sel_out)
{
    FeatSet &f = ctx->feat[kind];
    const size_t m = size_t(f.m);
    const size_t mp = (m + 63) & ~size_t(63);
    const size_t off_v = sizeof(double) * 6 * m, off_k = off_v + mp, need = off_k + mp;
    if (need > ctx->select_host_cap[kind]) {
        MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->select_host[kind]) (void)hipHostFree(ctx->select_host[kind]);
        ctx->select_host[kind] = nullptr; ctx->select_host_cap[kind] = 0;
        MLH_HIP(ctx, hipHostMalloc(&ctx->select_host[kind], need + need / 4, hipHostMallocDefault));
        ctx->select_host_cap[kind] = need + need / 4;
    }
    char *hb = static_cast<char *>(ctx->select_host[kind]);
    const size_t n_use = static_cast<size_t>(m * double(gf_ratio));
    const bool need_rows = n_use > 0 && static_cast<size_t>(1.0 * m / n_use) > 1;      // subsets of one never look at a score (select_greedy_odom)
    if (need_rows) MLH_HIP(ctx, hipMemcpyAsync(hb, f.J.p, sizeof(double) * 6 * m, hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipMemcpyAsync(hb + off_v, f.flag8.p, m, hipMemcpyDeviceToHost, ctx->stream));
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->select_rows[kind].resize(off_k);
    char *cb = ctx->select_rows[kind].data();
    if (need_rows) std::memcpy(cb, hb, off_v);      // (the loop jumps around in these rows: ordinary memory, see good_feature_finish)
    std::memcpy(cb + off_v, hb + off_v, m);
    Rows R;
    R.m = m; R.J = reinterpret_cast<const double *>(cb); R.valid = reinterpret_cast<const uint8_t *>(cb + off_v); R.pts = nullptr;
    std::vector<size_t> sel;
    sel.reserve(n_use);
    select_greedy_odom(R, n_use, rng, sel);
    uint8_t *keep = reinterpret_cast<uint8_t *>(hb + off_k);
    std::memset(keep, 0, m);
    for (size_t i : sel) keep[i] = 1;
    MLH_HIP(ctx, hipMemcpyAsync(f.flag8.p, keep, m, hipMemcpyHostToDevice, ctx->stream));
    MLH_LAUNCH(apply_keep_kernel, dim3(unsigned((m + 255) / 256)), dim3(256), 0, ctx->stream, f.corr.as<Corr>(), int(m), f.flag8.as<uint8_t>(), -1, 0);
    MLH_HIP(ctx, hipGetLastError());
    sel_out.assign(sel.begin(), sel.end());
    return MLH_OK;
}

// Both halves back to back, and the stream drained: what mlh_good_feature_matching (one kind, caller reads the selection) uses.
int good_feature_select(mlh_ctx *ctx, int kind, int method, double ratio, std::mt19937 &rng, float min_match_sq_dis,
                        float min_plane_dis, std::vector<int32_t> &sel_out, double H[36], uint8_t *matched_out)
{
    int rc = good_feature_stage(ctx, kind, method, ratio, rng, min_match_sq_dis, min_plane_dis);
    if (rc) return rc;
    if ((rc = good_feature_finish(ctx, kind, method, ratio, rng, sel_out, H, matched_out))) return rc;
    MLH_HIP(ctx, hipStreamSynchronize(ctx->stream));
    prof_collect(ctx);
    return MLH_OK;
}

}  // namespace mlh
