// ImageSegmenter::segmentCloud for gfx950 (estimator/src/imageSegmenter/image_segmenter.hpp:88-393, image_segmenter.cpp:18-63):
// the producer of the ring-major cloud + ScanInfo that extractCloud consumes. Raw, unordered cloud in -> the context's scan staged in HBM
// (as mlh_scan_upload leaves it), so a frame goes driver cloud -> segmentCloud -> extractCloud without the ring-major cloud ever being
// assembled on the host.
//
// What runs where, and why:
//   device  projectCloud (hpp:88-136): per point range / row / column in the reference's float-double mix; "the first point to claim a pixel
//           keeps it" is an atomicMin on the point index. Ground labelling (hpp:179-227): a stencil over pixel pairs. The final gather of the
//           kept points into the ring-major float4 cloud (intensity += row id, hpp:128) and its ring table.
//   host    the cluster search (hpp:229-359) and the outlier erasure (hpp:362-383). Both are DEFINED by their sequential order: whether a
//           "same beam" neighbour joins a cluster depends on the record of the NEXT queue entry at that moment (hpp:297-299), the distance
//           uses the previous neighbour's alpha (hpp:285-286), erasures use positions that earlier erasures have shifted (hpp:374). A parallel
//           labelling would be a different segmenter; one wavefront stepping through the queue out of HBM would be ~20x slower than a CPU
//           core (every step is a dependent memory round trip). The range image (115 KB for 16 x 1800) goes down, a list of kept point
//           indices comes back.
// Undefined behaviour in the reference, and what happens here instead (INTEGRATION.md has the same list for the maintainer):
//   (U1) hpp:285-286 computes `dist` with the alpha of the PREVIOUS neighbour, uninitialised on the first one: one variable for the whole
//        call, starting at 0;  (U2) hpp:374 erases with positions that earlier erasures have shifted, possibly past the end: the stale
//        position is used as it is, one outside the row erases nothing;  (U3) the 64-ring ground loop reads row 64 (hpp:183-185) and
//        segment_alphay_ is never set: the loop is clipped to existing rows, alphay takes the value the cluster search assigns.
// atan / atan2 run in f32 on the device (ocml) and in glibc on the reference's CPU: the last ulp may differ, which matters only for a point
// within one ulp of a row / column bin edge or a ground pair within one ulp of 10 degrees.
#include "ctx.hpp"
#include <mutex>
#include "alive_pool.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>

namespace mlh {

struct SegSetup {   // ImageSegmenter::setParameter
    int vs, hs, ground_scan_id, is64;
    float ang_res_x, ang_res_y, ang_bottom, alphax, alphay;
};

static bool seg_setup(const mlh_segment_params &p, SegSetup &s)
{
    s.vs = p.vertical_scans; s.hs = p.horizon_scans; s.is64 = 0; s.alphay = 0.f; s.ang_bottom = 0.f;
    if (p.vertical_scans == 16) {
        s.ang_res_x = 360.0 / p.horizon_scans; s.ang_res_y = 2.0; s.ang_bottom = 15.0 + 0.1; s.ground_scan_id = 7;
        s.alphax = s.ang_res_x / 180.0 * M_PI; s.alphay = s.ang_res_y / 180.0 * M_PI;
    } else if (p.vertical_scans == 32) {
        s.ang_res_x = 360.0 / p.horizon_scans; s.ang_res_y = 41.33 / float(p.vertical_scans - 1); s.ang_bottom = 30.0 + 0.67; s.ground_scan_id = 20;
        s.alphax = s.ang_res_x / 180.0 * M_PI; s.alphay = s.ang_res_y / 180.0 * M_PI;
    } else if (p.vertical_scans == 64) {
        s.ang_res_x = 360.0 / p.horizon_scans; s.ang_res_y = FLT_MAX; s.ground_scan_id = 63; s.is64 = 1;
        s.alphax = s.ang_res_x / 180.0 * M_PI;
    } else return false;
    return true;
}

struct SegDev {
    const unsigned char *src; int stride, intensity_off, n;
    SegSetup S; double roi_range;
    int *pix;            // n: row * hs + col, or -1
    int *owner;          // vs * hs: smallest point index that claimed the pixel (INT_MAX: empty)
    float *range_mat;    // vs * hs
    unsigned char *ground;   // vs * hs
    // Bins are decided by f32 atan / atan2, where the device's libm and the host's (glibc: what the reference runs) may differ in the last ulp. A point whose row or
    // column value sits within a margin of a bin edge (a thousand times that ulp), or a ground pair whose angle sits that close to 10 degrees, is NOT decided here:
    // it is handed to the host, which evaluates the reference's expression with its own libm (segment_cloud_run). A few dozen per scan.
    int *unc_count;      // [0]: undecided points, [1]: undecided ground pairs
    float4 *unc_pts;     // {x, y, z, point index as int bits}, capacity n
    float4 *unc_gnd;     // 2 records per pair: {x1, y1, z1, pixel as int bits}, {x2, y2, z2, 0}, capacity 2 * vs * hs
};

constexpr float SEG_EDGE_MARGIN_BINS = 4.0e-4f;      // in units of one bin (row or column); an f32 ulp of a 180-degree angle is 1.5e-5 degrees = 7.6e-5 columns at 0.2 degrees: five of them
constexpr float SEG_EDGE_MARGIN_DEG = 1.0e-4f;       // for the comparisons of an angle with a constant (2, -8.83, -24.33, 10 degrees): ~50 ulps of the angle

// projectCloud, one lane per point (hpp:96-135)
__global__ __launch_bounds__(256) void seg_project_kernel(SegDev D)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= D.n) return;
    const float *rec = reinterpret_cast<const float *>(D.src + size_t(i) * D.stride);
    const float x = rec[0], y = rec[1], z = rec[2];
    int pix = -1;
    bool undecided = false;
    const float range = sqrtf(x * x + y * y + z * z);
    if (!(double(range) < D.roi_range)) {
        const float vertical_angle = float(double(atanf(z / sqrtf(x * x + y * y)) * 180) / M_PI);
        int row_id;
        bool ok = true;
        if (D.S.is64) {
            double rv;
            if (double(vertical_angle) >= -8.83) { rv = double(2 - vertical_angle) * 3.0 + 0.5; row_id = int(rv); }          // (2 - angle) is a float subtraction in the reference
            else { rv = (-8.83 - double(vertical_angle)) * 2.0 + 0.5; row_id = D.S.vs / 2 + int(rv); }
            if (vertical_angle > 2 || double(vertical_angle) < -24.33 || row_id > 50 || row_id < 0) ok = false;
            undecided = fabs(rv - rint(rv)) < double(SEG_EDGE_MARGIN_BINS) || fabsf(vertical_angle - 2.f) < SEG_EDGE_MARGIN_DEG || fabs(double(vertical_angle) + 8.83) < double(SEG_EDGE_MARGIN_DEG) ||
                        fabs(double(vertical_angle) + 24.33) < double(SEG_EDGE_MARGIN_DEG);
        } else {
            const float rv = (vertical_angle + D.S.ang_bottom) / D.S.ang_res_y;
            row_id = int(rv);
            if (row_id < 0 || row_id >= D.S.vs) ok = false;
            undecided = fabsf(rv - rintf(rv)) < SEG_EDGE_MARGIN_BINS;
        }
        if (ok || undecided) {
            const float horizon_angle = float(double(atan2f(x, y) * 180) / M_PI);
            const double cv = (double(horizon_angle) - 90.0) / double(D.S.ang_res_x);
            int column_id = int(-round(cv) + double(D.S.hs / 2));
            if (column_id >= D.S.hs) column_id -= D.S.hs;
            undecided = undecided || fabs(fabs(cv - floor(cv)) - 0.5) < double(SEG_EDGE_MARGIN_BINS);
            if (!undecided && column_id >= 0 && column_id < D.S.hs) {
                pix = column_id + row_id * D.S.hs;
                atomicMin(D.owner + pix, i);          // range_mat(row, col) != FLT_MAX -> continue: the first point in input order keeps the pixel
            }
        }
        if (undecided) {                              // the host decides with the reference's libm; its verdict arrives through seg_apply_fix_kernel
            const int k = atomicAdd(D.unc_count, 1);
            D.unc_pts[k] = make_float4(x, y, z, __int_as_float(i));
        }
    }
    D.pix[i] = pix;
}

// the host's verdicts for the undecided points: fix[k] = {point index, pixel or -1}
__global__ __launch_bounds__(256) void seg_apply_fix_kernel(const int2 *fix, int n_fix, int *pix, int *owner)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_fix) return;
    const int2 f = fix[k];
    pix[f.x] = f.y;
    if (f.y >= 0) atomicMin(owner + f.y, f.x);
}

// the memset pattern 0x7f7f7f7f stands for "no point yet"; normalise to INT_MAX for everything that follows
__global__ __launch_bounds__(256) void seg_owner_fix_kernel(int *owner, int npx)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p < npx && owner[p] == 0x7f7f7f7f) owner[p] = INT_MAX;
}

__device__ __forceinline__ void seg_point(const SegDev &D, int i, float &x, float &y, float &z)
{
    const float *rec = reinterpret_cast<const float *>(D.src + size_t(i) * D.stride);
    x = rec[0]; y = rec[1]; z = rec[2];
}

// range image + ground pairs (hpp:166-227), one lane per pixel
__global__ __launch_bounds__(256) void seg_image_kernel(SegDev D)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    const int npx = D.S.vs * D.S.hs;
    if (p >= npx) return;
    const int o = D.owner[p];
    float range = FLT_MAX, x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (o != INT_MAX) { seg_point(D, o, x1, y1, z1); range = sqrtf(x1 * x1 + y1 * y1 + z1 * z1); }
    D.range_mat[p] = range;
    const int i = p / D.S.hs;
    const int lo = D.S.is64 ? D.S.ground_scan_id : 0, hi = D.S.is64 ? D.S.vs : D.S.ground_scan_id;
    if (i >= lo && i < hi && i + 1 < D.S.vs && o != INT_MAX) {
        const int o2 = D.owner[p + D.S.hs];
        if (o2 != INT_MAX) {
            float x2, y2, z2;
            seg_point(D, o2, x2, y2, z2);
            const float dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;
            const float vertical_angle = float(double(atan2f(dz, sqrtf(dx * dx + dy * dy)) * 180) / M_PI);
            if (fabsf(fabsf(vertical_angle) - 10.f) < SEG_EDGE_MARGIN_DEG) {                     // too close to the threshold for this libm to speak for the host's
                const int k = atomicAdd(D.unc_count + 1, 1);
                D.unc_gnd[2 * k] = make_float4(x1, y1, z1, __int_as_float(p));
                D.unc_gnd[2 * k + 1] = make_float4(x2, y2, z2, 0.f);
            } else if (fabsf(vertical_angle) <= 10) { D.ground[p] = 1; D.ground[p + D.S.hs] = 1; }      // both lanes that touch a pixel write the same 1
        }
    }
}

// The cluster search's angle test (hpp:281-289) for every pixel and each of its four neighbours, ahead of the search: which pairs are joined by the angle rule does
// not depend on the search's order -- only its fall-back rule (hpp:297-300) and the labels do. Two bits per direction (the search's neighbour order: up, right,
// left, down): 2 = joined, 0 = not, 1 = within 1e-4 (relative) of the threshold, or a theta outside the range the tangent form covers: the host evaluates
// std::atan2 there, as it did for every such pair before. The arithmetic is the host loop's, operation for operation, on the same floats (the sin / cos / tan
// table entries are the host's libm results, passed in): the same booleans.
struct SegEdge {
    const float *range_mat;
    unsigned char *edge;
    int vs, hs, is64, theta_simple;
    float t_sin[4], t_cos[4], tan_theta;
};
__global__ __launch_bounds__(256) void seg_edge_kernel(SegEdge E)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= E.vs * E.hs) return;
    const int fx = p / E.hs, fy = p - fx * E.hs;
    const float rf = E.range_mat[p];
    unsigned bits = 0u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int dx = q == 0 ? -1 : (q == 3 ? 1 : 0), dy = q == 1 ? 1 : (q == 2 ? -1 : 0);
        const int tx = fx + dx;
        int ty = fy + dy;
        if (tx < 0 || tx >= E.vs) continue;
        if (ty < 0) ty = E.hs - 1;
        if (ty >= E.hs) ty = 0;
        const float rt = E.range_mat[tx * E.hs + ty];
        const float d1 = (rf < rt) ? rt : rf, d2 = (rt < rf) ? rt : rf;      // std::max(rf, rt) / std::min(rf, rt) as the host loop spells them (a NaN range stays where std:: leaves it)
        const int a = dx == 0 ? 1 : (E.is64 ? (tx <= 32 ? 2 : 3) : 2);
        const float ay = d2 * E.t_sin[a], ax = d1 - d2 * E.t_cos[a];
        const float lim = ax * E.tan_theta;
        unsigned code = 1u;
        if (E.theta_simple && ax > 0.f && ay > lim * 1.0001f) code = 2u;
        else if (E.theta_simple && ax > 0.f && ay < lim * 0.9999f) code = 0u;
        bits |= code << (2 * q);
    }
    E.edge[p] = (unsigned char)bits;
}

// the kept points, ring-major: out[k] = {x, y, z, intensity + row}
__global__ __launch_bounds__(256) void seg_gather_kernel(SegDev D, const int *keep, int n_keep, float4 *out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_keep) return;
    const int i = keep[k];
    const float *rec = reinterpret_cast<const float *>(D.src + size_t(i) * D.stride);
    float inten = D.intensity_off >= 0 ? *reinterpret_cast<const float *>(D.src + size_t(i) * D.stride + D.intensity_off) : 0.f;
    inten += float(D.pix[i] / D.S.hs);
    out[k] = make_float4(rec[0], rec[1], rec[2], inten);
}

// ---- the rows of the output cloud, on the device (round 5; until then the host assembled them: ~0.85 ms of a 2.6 ms call on a 64-ring scan)
// After the cluster search only its verdict is order-dependent no more: which pixels are outliers. Everything behind it is data movement with a fixed rule
// (image_segmenter.hpp:359-383), one row at a time and the rows independent of each other:
//   * cloud_scan[row] = the row's pixel owners in INPUT order (the reference pushes them while it walks the cloud) = the owners sorted by point index;
//     cloud_scan_order(row, col) = the owner's position in that list at fill time;
//   * for every outlier pixel of the row, columns ascending: cloud_scan[row].erase(begin() + cloud_scan_order(row, col)) -- "erase what is NOW at the position
//     recorded at fill time", stale positions included, a position at or past the current size erasing nothing (U2);
//   * the rows concatenated, scan_start = offset + 5, scan_end = offset + size - 6.
// One workgroup per row: a bitonic sort of (owner, column) keys in LDS gives list and positions; the erasures run in their own order on ONE wavefront that holds
// the row's alive flags as 64 x 64 bits in registers (lane l: positions 64 l .. 64 l + 63) -- "the element now at position p" is a popcount prefix over the lanes,
// a ballot and a bit select inside one word, ~0.1 us per erasure with no memory access, exact for any order of positions (the host needed an order-statistic tree
// for clouds that are not in firing order); the survivors are compacted by a block scan. A second launch adds up the rows' sizes and gathers the points.
struct SegRows {
    const int *owner;           // vs * hs (INT_MAX: empty)
    const unsigned *outmask;    // one bit per pixel: label 999999
    int *kept;                  // vs * hs: the surviving point indices of row r at [r * hs, ...)
    int *row_cnt;               // [vs]: survivors per row
    int vs, hs, hs2;            // hs2 = hs rounded up to a power of two (<= SEG_ROW_MAX)
    int segment_flag;
};
constexpr int SEG_ROW_MAX = 4096;
constexpr int SEG_ROW_TPB = 1024;

__global__ __launch_bounds__(SEG_ROW_TPB) void seg_rows_kernel(SegRows A)
{
    extern __shared__ unsigned long long s_dyn[];
    unsigned long long *s_key = s_dyn;                                // hs2 keys: (owner << 32) | column
    int *s_order = reinterpret_cast<int *>(s_key + A.hs2);            // hs2: position of column c's owner in the sorted list
    int *s_epos = s_order + A.hs2;                                    // hs2: the positions to erase, in erasure order (+ hs2 more behind it: s_idx)
    __shared__ unsigned long long s_alive[64];
    __shared__ int s_wave_tot[SEG_ROW_TPB / 64], s_misc[4];
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int hs = A.hs, hs2 = A.hs2;
    for (int c = t; c < hs2; c += SEG_ROW_TPB) {
        const unsigned o = c < hs ? (unsigned)A.owner[size_t(r) * hs + c] : (unsigned)INT_MAX;
        s_key[c] = (static_cast<unsigned long long>(o) << 32) | unsigned(c);
    }
    __syncthreads();
    for (int k = 2; k <= hs2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = t; i < hs2; i += SEG_ROW_TPB) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_key[i], b = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { s_key[i] = b; s_key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    // size0 = the non-empty pixels (sorted in front); positions by column
    int mine = 0;
    for (int i = t; i < hs2; i += SEG_ROW_TPB) {
        const unsigned long long k = s_key[i];
        const bool full = (unsigned)(k >> 32) != (unsigned)INT_MAX;
        mine += full ? 1 : 0;
        const int col = int(unsigned(k));
        if (col < hs) s_order[col] = full ? i : -1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off);
    if (lane == 0) s_wave_tot[wave] = mine;
    __syncthreads();
    int size0 = 0;
#pragma unroll
    for (int w = 0; w < SEG_ROW_TPB / 64; ++w) size0 += s_wave_tot[w];
    __syncthreads();
    // the erasure list: outlier pixels, columns ascending -> their fill-time positions (a block-wide compaction: thread t owns columns [cpt t, cpt (t + 1)))
    int n_erase = 0;
    if (A.segment_flag) {
        const int cpt = hs2 / SEG_ROW_TPB > 0 ? hs2 / SEG_ROW_TPB : 1;
        const int c0 = t * cpt;
        int flags = 0, cnt = 0;
        for (int u = 0; u < cpt; ++u) {
            const int c = c0 + u;
            if (c < hs) {
                const size_t px = size_t(r) * hs + c;
                if ((A.outmask[px >> 5] >> (px & 31)) & 1u) { flags |= 1 << u; ++cnt; }
            }
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += s_wave_tot[w];
        for (int w = 0; w < SEG_ROW_TPB / 64; ++w) n_erase += s_wave_tot[w];
        int dst = base + incl - cnt;
        for (int u = 0; u < cpt; ++u) if (flags & (1 << u)) s_epos[dst++] = s_order[c0 + u];
        __syncthreads();
    }
    // The erasures. Sequential by definition -- "erase what is NOW at position p", one after the other -- but a RUN of strictly monotone positions can be applied at
    // once. Ascending: the j-th erasure removes the element whose rank in the list AS IT WAS AT THE RUN'S START is p_j + j (the j earlier ones all sat in front of
    // it), and it erases nothing from the first j on with p_j + j >= size (U2: a position at or past the current size). Descending (round 6): the earlier ones all
    // sat BEHIND it and moved nothing in front of it -- the j-th erasure removes run-start rank p_j, and it erases something iff p_j < the run-start size (the
    // positions strictly fall, so once one is inside the list every later one is, by at least as much as the list has shrunk). A driver's cloud gives one or two
    // runs per row either way -- a pixel's fill position grows with its column when the sensor turns counter-clockwise in the image's sense, falls when it turns
    // clockwise (a ring-major cloud in firing order: until round 6 every erasure of such a row was a run of its own, 1.7 ms of a 64-ring call instead of 0.17), with
    // one step where the sweep starts; an unordered cloud gives many short runs, each still exact.
    // Per run: every thread turns its ranks into list indices (a search over the 64 words' popcount prefix, a bit select inside the word), the bits are cleared,
    // the prefix is rebuilt. (First version: one wavefront erasing one element at a time, 0.15 us each -- 0.5-0.8 ms for a 64-ring scan's rows.)
    int *s_runs = s_order;                                            // (the positions by column are not needed any more)
    int *s_idx = s_epos + hs2;                                        // hs2: the list index an erasure removes
    __shared__ int s_pref[65];
    int n_runs = 0;
    {
        const int ept = hs2 / SEG_ROW_TPB > 0 ? hs2 / SEG_ROW_TPB : 1;
        const int e0 = t * ept;
        int flags = 0, cnt = 0;
        // a run starts where the step into the element is not the step before it (the first element; an equal position; a change of direction): the steps inside a
        // run then all have the direction of the step into its second element
        for (int u = 0; u < ept; ++u) {
            const int e = e0 + u;
            if (e >= n_erase) continue;
            bool start = e == 0;
            if (e >= 1) {
                const int d1 = (s_epos[e] > s_epos[e - 1]) - (s_epos[e] < s_epos[e - 1]);
                start = d1 == 0;
                if (e >= 2 && !start) { const int d0 = (s_epos[e - 1] > s_epos[e - 2]) - (s_epos[e - 1] < s_epos[e - 2]); start = d0 != d1; }
            }
            if (start) { flags |= 1 << u; ++cnt; }
        }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();                                               // (also: every read of s_order as positions lies before this barrier)
        int base = 0;
        for (int w = 0; w < wave; ++w) base += s_wave_tot[w];
        for (int w = 0; w < SEG_ROW_TPB / 64; ++w) n_runs += s_wave_tot[w];
        int dst = base + incl - cnt;
        for (int u = 0; u < ept; ++u) if (flags & (1 << u)) s_runs[dst++] = e0 + u;
    }
    if (wave == 0) {
        unsigned long long w = 0ull;
        const int lo = lane * 64;
        if (size0 >= lo + 64) w = ~0ull;
        else if (size0 > lo) w = (1ull << (size0 - lo)) - 1ull;
        s_alive[lane] = w;
        const int pc = __popcll(w);
        int incl = pc;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        s_pref[lane] = incl - pc;
        if (lane == 63) s_pref[64] = incl;
        if (lane == 0) { s_misc[0] = size0; s_misc[1] = 0; }
    }
    __syncthreads();
    // A row whose erasures come in runs shorter than ~8 (an unordered cloud: nothing may be assumed about a driver's order) is cheaper one erasure at a time on ONE
    // wavefront that keeps the row's alive bits (lane l: list indices 64 l .. 64 l + 63) and their popcount prefix in registers -- "the element now at position p":
    // a ballot finds the word, a popcount search the bit, the lanes behind it count one less; ~0.15 us per erasure and no barrier, against three barriers (~1.2 us)
    // per run below (a shuffled 64-ring scan's rows: 2.45 -> 0.3 ms per call).
    const bool one_by_one = n_runs * 8 > n_erase;                       // uniform
    if (one_by_one) {
        if (wave == 0) {
            unsigned long long w = s_alive[lane];
            int pref = s_pref[lane], size = s_misc[0];
            for (int e = 0; e < n_erase; ++e) {
                const int p = s_epos[e];
                if (p < 0 || p >= size) continue;                       // (U2) a position at or past the current size erases nothing
                const unsigned long long holds = __ballot(pref <= p);   // lane 0's prefix is 0: never empty
                const int wd = __builtin_amdgcn_readfirstlane(63 - __clzll((long long)holds));
                const int rr = p - __builtin_amdgcn_readlane(pref, wd);
                if (lane == wd) {
                    unsigned long long m = w;
                    int r2 = rr, bit = 0;
#pragma unroll
                    for (int sh = 32; sh > 0; sh >>= 1) {
                        const int c = __popcll(m & ((1ull << sh) - 1ull));
                        if (r2 >= c) { r2 -= c; m >>= sh; bit += sh; }
                    }
                    w &= ~(1ull << bit);
                }
                pref -= lane > wd ? 1 : 0;
                --size;
            }
            s_alive[lane] = w;
            if (lane == 0) s_misc[0] = size;
        }
        __syncthreads();
    }
    for (int rn = 0; rn < (one_by_one ? 0 : n_runs); ++rn) {            // uniform
        const int ra = s_runs[rn], rb = rn + 1 < n_runs ? s_runs[rn + 1] : n_erase;
        const int cnt0 = s_misc[0];
        const bool ascending = rb - ra < 2 || s_epos[ra + 1] > s_epos[ra];
        int valid = 0;
        for (int j = ra + t; j < rb; j += SEG_ROW_TPB) {
            const int tt = s_epos[j] + (ascending ? j - ra : 0);
            int idx = -1;
            if (s_epos[j] >= 0 && tt < cnt0) {
                int wd = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) wd += (s_pref[wd + step] <= tt) ? step : 0;      // the last word whose prefix is <= tt: it holds rank tt
                int rr = tt - s_pref[wd], bit = 0;
                unsigned long long m = s_alive[wd];
#pragma unroll
                for (int sh = 32; sh > 0; sh >>= 1) {
                    const int c = __popcll(m & ((1ull << sh) - 1ull));
                    if (rr >= c) { rr -= c; m >>= sh; bit += sh; }
                }
                idx = wd * 64 + bit;
                ++valid;
            }
            s_idx[j] = idx;
        }
        if (valid) atomicAdd(&s_misc[1], valid);
        __syncthreads();
        for (int j = ra + t; j < rb; j += SEG_ROW_TPB) {
            const int idx = s_idx[j];
            if (idx >= 0) atomicAnd(&s_alive[idx >> 6], ~(1ull << (idx & 63)));
        }
        __syncthreads();
        if (wave == 0) {
            const int pc = __popcll(s_alive[lane]);
            int incl = pc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
            s_pref[lane] = incl - pc;
            if (lane == 63) s_pref[64] = incl;
            if (lane == 0) { s_misc[0] = cnt0 - s_misc[1]; s_misc[1] = 0; }
        }
        __syncthreads();
    }
    // survivors, in list order -> kept[r * hs ...]
    {
        const int ipt = hs2 / SEG_ROW_TPB > 0 ? hs2 / SEG_ROW_TPB : 1;
        const int i0 = t * ipt;
        int cnt = 0;
        for (int u = 0; u < ipt; ++u) { const int i = i0 + u; if (i < size0 && ((s_alive[i >> 6] >> (i & 63)) & 1ull)) ++cnt; }
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
        if (lane == 63) s_wave_tot[wave] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < wave; ++w) base += s_wave_tot[w];
        int dst = base + incl - cnt;
        for (int u = 0; u < ipt; ++u) {
            const int i = i0 + u;
            if (i < size0 && ((s_alive[i >> 6] >> (i & 63)) & 1ull)) A.kept[size_t(r) * hs + dst++] = int(unsigned(s_key[i] >> 32));
        }
        if (t == 0) A.row_cnt[r] = s_misc[0];
    }
}

// the rows concatenated: out[k] = {x, y, z, intensity + row} of the k-th kept point, the ring tables as ScanInfo has them, and what the host needs to know
// (sizes per row, total, first kept point) in pinned memory
__global__ __launch_bounds__(256) void seg_rows_gather_kernel(SegDev D, const int *kept, const int *row_cnt, float4 *out, int *start, int *end, int *host_rows)
{
    const int r = blockIdx.y, vs = D.S.vs, hs = D.S.hs;
    int off = 0, total = 0;
    for (int q = 0; q < vs; ++q) { const int c = row_cnt[q]; off += q < r ? c : 0; total += c; }
    const int cnt = row_cnt[r];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        start[r] = off + 5; end[r] = off + cnt - 6;
        host_rows[r] = cnt;
        if (r == 0) {
            host_rows[vs] = total;
            int first = -1;
            for (int q = 0; q < vs && first < 0; ++q) if (row_cnt[q] > 0) first = kept[size_t(q) * hs];
            host_rows[vs + 1] = first;
        }
    }
    for (int k = blockIdx.x * 256 + threadIdx.x; k < cnt; k += gridDim.x * 256) {
        const int i = kept[size_t(r) * hs + k];
        const float *rec = reinterpret_cast<const float *>(D.src + size_t(i) * D.stride);
        float inten = D.intensity_off >= 0 ? *reinterpret_cast<const float *>(D.src + size_t(i) * D.stride + D.intensity_off) : 0.f;
        inten += float(r);
        out[off + k] = make_float4(rec[0], rec[1], rec[2], inten);
    }
}

// ---- host: cluster search + outlier erasure on the images (sequential by definition, see the file comment)
// The host side of a call, carved out of two blocks the context keeps (SegBuf::h_img pinned -- the three images land there straight from the copy engine and the
// mask leaves from there; SegBuf::h_bfs plain): a fresh set of vectors per call cost ~2.5 MB of allocation, page faults and fills per 64-ring scan.
struct SegHost {
    float *range;
    int *owner;
    unsigned char *ground;
    unsigned *outmask;      // one bit per pixel: label 999999 (filled where the cluster search marks an infeasible cluster)
    size_t outmask_words;
    unsigned char *edge;    // seg_edge_kernel's verdicts, 2 bits per neighbour
    int *label;
    uint16_t *qx, *qy;      // the queue: every pushed pixel enters it exactly once, so it is the cluster's pixel list too
    int8_t *q_last_dy;      // queue_last_dy of the entry
    // queue_last_dis of the entry, kept as its ingredients: dist = sqrt(d1^2 + d2^2 - 2 d1 d2 cos(alpha)) is read back only by the same-beam rule (hpp:297-300), for a
    // few per cent of the entries. d1 / d2 are the larger / smaller range of the entry's pixel and of the pixel it was reached from (q_parent); the alpha in force
    // when it was pushed is an index into the four-entry table (q_alpha; 255: the literal 0 a cluster's seed is pushed with, hpp:245). The square root is taken
    // where the rule reads it (same expression, same operands: the same float)
    uint32_t *q_parent;
    unsigned char *q_alpha;
};

static int seg_host_carve(mlh_ctx *ctx, SegBuf &B, int npx, SegHost &H)
{
    const size_t words = (size_t(npx) + 31) / 32;
    const size_t img = sizeof(float) * size_t(npx) + sizeof(int) * size_t(npx) + sizeof(unsigned) * words + 2 * size_t(npx) + 64;
    if (B.h_img_cap < img) {
        if (B.h_img) (void)hipHostFree(B.h_img);
        B.h_img = nullptr; B.h_img_cap = 0;
        MLH_HIP(ctx, hipHostMalloc(&B.h_img, img, hipHostMallocDefault));
        B.h_img_cap = img;
    }
    unsigned char *p = static_cast<unsigned char *>(B.h_img);
    H.range = reinterpret_cast<float *>(p); p += sizeof(float) * size_t(npx);
    H.owner = reinterpret_cast<int *>(p); p += sizeof(int) * size_t(npx);
    H.outmask = reinterpret_cast<unsigned *>(p); p += sizeof(unsigned) * words;
    H.outmask_words = words;
    H.ground = p; p += size_t(npx);
    H.edge = p;
    const size_t bfs = size_t(npx) * (sizeof(int) + 2 * sizeof(uint16_t) + 2 + sizeof(uint32_t)) + 64;
    if (B.h_bfs_cap < bfs) {
        std::free(B.h_bfs);
        B.h_bfs = std::malloc(bfs);
        B.h_bfs_cap = B.h_bfs ? bfs : 0;
        if (!B.h_bfs) return fail(ctx, MLH_ERR_INVALID, "mlh_segment_cloud: out of host memory");
    }
    unsigned char *q = static_cast<unsigned char *>(B.h_bfs);
    H.label = reinterpret_cast<int *>(q); q += sizeof(int) * size_t(npx);
    H.q_parent = reinterpret_cast<uint32_t *>(q); q += sizeof(uint32_t) * size_t(npx);
    H.qx = reinterpret_cast<uint16_t *>(q); q += sizeof(uint16_t) * size_t(npx);
    H.qy = reinterpret_cast<uint16_t *>(q); q += sizeof(uint16_t) * size_t(npx);
    H.q_last_dy = reinterpret_cast<int8_t *>(q); q += size_t(npx);
    H.q_alpha = q;
    std::memset(H.outmask, 0, sizeof(unsigned) * words);
    // the queue's records persist from cluster to cluster WITHIN a call (an entry behind the queue's end is read by the same-beam rule, hpp:297) and start a call
    // as "seed" records: dy 0, the literal 0 distance
    std::memset(H.q_last_dy, 0, size_t(npx));
    std::memset(H.q_alpha, 255, size_t(npx));
    return MLH_OK;
}

static void seg_clusters(const SegSetup &S0, const mlh_segment_params &prm, SegHost &H, const float (&t_cos)[4], const float (&t_sin)[4])
{
    SegSetup S = S0;
    const int vs = S.vs, hs = S.hs;
    uint16_t *qx = H.qx, *qy = H.qy;
    int8_t *q_last_dy = H.q_last_dy;
    uint32_t *q_parent = H.q_parent;
    unsigned char *q_alpha = H.q_alpha;
    const float *range = H.range;
    int *label = H.label;
    const unsigned char *edge = H.edge;
    int label_count = 2;
    static const int8_t nb[4][2] = {{-1, 0}, {0, 1}, {0, -1}, {1, 0}};
    // alpha -- ONE variable for the whole call, starting at 0 (U1) -- only ever holds 0, alphax or one of the two alphay values: it is carried as an index into
    // the table of the std::cos / std::sin results (computed once by the caller: the same floats the reference's calls return)
    int alpha_idx = 0;
    auto dist_of = [](float d1, float d2, float c) { return std::sqrt(d1 * d1 + d2 * d2 - 2 * d1 * d2 * c); };
    // queue_last_dis of queue entry k (see SegHost)
    auto dist_last_of = [&](int k) -> float {
        if (q_alpha[k] == 255) return 0.f;
        const float ra = range[size_t(qx[k]) * hs + qy[k]], rb = range[q_parent[k]];
        return dist_of(std::max(ra, rb), std::min(ra, rb), t_cos[q_alpha[k]]);
    };
    std::vector<char> line_flag(vs);
    for (int i = 0; i < vs; i++) {
        for (int j = 0; j < hs; j++) {
            if (label[size_t(i) * hs + j] != 0) continue;
            std::fill(line_flag.begin(), line_flag.end(), 0);
            qx[0] = uint16_t(i); qy[0] = uint16_t(j); q_last_dy[0] = 0; q_alpha[0] = 255;
            int q_start = 0, q_end = 1;
            while (q_start < q_end) {
                const int fx = qx[q_start], fy = qy[q_start];
                ++q_start;
                const size_t fp = size_t(fx) * hs + fy;
                label[fp] = label_count;
                const unsigned eb = edge[fp];
                for (int q = 0; q < 4; ++q) {
                    int tx = fx + nb[q][0], ty = fy + nb[q][1];
                    if (tx < 0 || tx >= vs) continue;
                    if (ty < 0) ty = hs - 1;
                    if (ty >= hs) ty = 0;
                    const size_t tp = size_t(tx) * hs + ty;
                    if (label[tp] != 0) continue;
                    const int alpha_prev = alpha_idx;                           // dist is computed with the alpha of the PREVIOUS evaluated neighbour (U1)
                    alpha_idx = nb[q][0] == 0 ? 1 : (S.is64 ? (tx <= 32 ? 2 : 3) : 2);      // (64 rings: segment_alphay_ follows the neighbour's row, hpp:266-272)
                    // angle > theta: seg_edge_kernel's verdict; within 1e-4 (relative) of the threshold std::atan2 itself, as the reference calls it
                    const unsigned code = (eb >> (2 * q)) & 3u;
                    bool push = code == 2u;
                    if (code == 1u) {
                        const float rf = range[fp], rt = range[tp];
                        const float d1 = std::max(rf, rt), d2 = std::min(rf, rt);
                        const float ay = d2 * t_sin[alpha_idx], ax = d1 - d2 * t_cos[alpha_idx];
                        push = std::atan2(ay, ax) > prm.segment_theta;
                    }
                    if (!push && nb[q][1] == 0 && q_last_dy[q_start] == 0) {          // the record of the NEXT queue entry (hpp:297)
                        const float rf = range[fp], rt = range[tp];
                        const float dist_last = dist_last_of(q_start);
                        const float dist = dist_of(std::max(rf, rt), std::min(rf, rt), t_cos[alpha_prev]);
                        push = (dist_last / dist <= 1.2) && (dist_last / dist >= 0.8);
                    }
                    if (push) {
                        qx[q_end] = uint16_t(tx); qy[q_end] = uint16_t(ty); q_last_dy[q_end] = nb[q][1];
                        q_parent[q_end] = uint32_t(fp); q_alpha[q_end] = (unsigned char)alpha_prev;
                        ++q_end;
                        label[tp] = label_count;
                        line_flag[tx] = 1;
                    }
                }
            }
            const int n_pushed = q_end;
            bool feasible = false;
            if (n_pushed >= prm.min_cluster_size) feasible = true;
            else if (n_pushed >= prm.segment_valid_point_num) {
                int lines = 0;
                for (int r = 0; r < vs; ++r) lines += line_flag[r] ? 1 : 0;
                feasible = lines >= prm.segment_valid_line_num;
            }
            if (feasible) ++label_count;
            else for (int k = 0; k < n_pushed; ++k) {
                const size_t px = size_t(qx[k]) * hs + qy[k];
                label[px] = 999999;
                H.outmask[px >> 5] |= 1u << (px & 31);
            }
        }
    }
}

// projectCloud's bin of one point with the HOST's libm (hpp:96-135; the expressions of seg_project_kernel, std::atan / std::atan2 in place of the device's): the
// verdict for a point the device found too close to a bin edge to decide. -1: the point claims no pixel.
static int seg_pixel_host(float x, float y, float z, const SegSetup &S, double roi_range)
{
    const float range = std::sqrt(x * x + y * y + z * z);
    if (double(range) < roi_range) return -1;
    const float vertical_angle = float(double(std::atan(z / std::sqrt(x * x + y * y)) * 180) / M_PI);
    int row_id;
    if (S.is64) {
        if (double(vertical_angle) >= -8.83) row_id = int(double(2 - vertical_angle) * 3.0 + 0.5);
        else row_id = S.vs / 2 + int((-8.83 - double(vertical_angle)) * 2.0 + 0.5);
        if (vertical_angle > 2 || double(vertical_angle) < -24.33 || row_id > 50 || row_id < 0) return -1;
    } else {
        row_id = int((vertical_angle + S.ang_bottom) / S.ang_res_y);
        if (row_id < 0 || row_id >= S.vs) return -1;
    }
    const float horizon_angle = float(double(std::atan2(x, y) * 180) / M_PI);
    int column_id = int(-std::round((double(horizon_angle) - 90.0) / double(S.ang_res_x)) + double(S.hs / 2));
    if (column_id >= S.hs) column_id -= S.hs;
    if (column_id < 0 || column_id >= S.hs) return -1;
    return column_id + row_id * S.hs;
}

// the ground test of one pixel pair with the host's libm (hpp:196-205)
static bool seg_ground_host(const float4 &a, const float4 &b)
{
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    const float vertical_angle = float(double(std::atan2(dz, std::sqrt(dx * dx + dy * dy)) * 180) / M_PI);
    return std::fabs(vertical_angle) <= 10;
}

constexpr int SEG_UNC_FIRST = 2048;     // undecided records fetched with the counters (a scan has a few dozen); more than that: one more copy

// hipFuncAttributeMaxDynamicSharedMemorySize of seg_rows_kernel, per device and monotone (two contexts -- two threads, two GPUs of a multi-rank process -- share the
// function object of their device; a smaller request never lowers what a launch in flight was granted). false: this device does not grant `bytes`.
static bool seg_rows_lds_granted(int device, size_t bytes)
{
    static std::mutex mu;
    static size_t granted[64] = {};
    static size_t refused[64] = {};
    if (device < 0 || device >= 64) return false;
    std::lock_guard<std::mutex> lock(mu);
    if (bytes <= granted[device]) return true;
    if (refused[device] && bytes >= refused[device]) return false;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(seg_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)) != hipSuccess) {
        (void)hipGetLastError();
        refused[device] = bytes;
        return false;
    }
    granted[device] = bytes;
    return true;
}

int segment_cloud_run(mlh_ctx *ctx, const void *points, int stride, int intensity_off, int n, int mem, const mlh_segment_params &prm,
                      float *cloud_out, int32_t *n_out, int32_t *scan_start, int32_t *scan_end, float *outlier_out, int32_t outlier_capacity, int32_t *n_outlier)
{
    SegSetup S;
    if (!seg_setup(prm, S)) return fail(ctx, MLH_ERR_UNSUPPORTED, "ImageSegmenter is set up for 16, 32 or 64 vertical scans (image_segmenter.cpp:18-61)");
    if (!points || n <= 0 || stride < 12 || (stride & 3)) return fail(ctx, MLH_ERR_INVALID, "bad point buffer");
    if (prm.horizon_scans <= 0 || prm.horizon_scans > 65535) return fail(ctx, MLH_ERR_INVALID, "horizon_scans out of range");
    hipStream_t st = ctx->stream;
    const int vs = S.vs, hs = S.hs, npx = vs * hs;
    SegBuf &B = ctx->seg;
    const unsigned char *src = static_cast<const unsigned char *>(points);
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, B.raw.ensure(size_t(n) * stride));
        MLH_HIP(ctx, hipMemcpyAsync(B.raw.p, points, size_t(n) * stride, hipMemcpyHostToDevice, st));
        src = B.raw.as<unsigned char>();
    }
    MLH_HIP(ctx, B.pix.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, B.owner.ensure(sizeof(int) * size_t(npx)));
    MLH_HIP(ctx, B.range.ensure(sizeof(float) * size_t(npx)));
    MLH_HIP(ctx, B.ground.ensure(size_t(npx)));
    MLH_HIP(ctx, hipMemsetAsync(B.owner.p, 0x7f, sizeof(int) * size_t(npx), st));      // 0x7f7f7f7f > any index; read back as "empty" below
    MLH_HIP(ctx, hipMemsetAsync(B.ground.p, 0, size_t(npx), st));
    // undecided points / ground pairs: [4 ints: two counters][n point records][2 x npx pair records]
    const size_t unc_bytes = 16 + sizeof(float4) * (size_t(n) + 2 * size_t(npx));
    MLH_HIP(ctx, B.unc.ensure(unc_bytes));
    MLH_HIP(ctx, hipMemsetAsync(B.unc.p, 0, 16, st));
    const size_t h_need = 16 + sizeof(float4) * size_t(std::max(n, 2 * npx));
    if (B.h_unc_cap < h_need) {
        if (B.h_unc) (void)hipHostFree(B.h_unc);
        B.h_unc = nullptr; B.h_unc_cap = 0;
        MLH_HIP(ctx, hipHostMalloc(&B.h_unc, h_need, hipHostMallocDefault));
        B.h_unc_cap = h_need;
    }
    SegDev D;
    D.src = src; D.stride = stride; D.intensity_off = intensity_off; D.n = n; D.S = S; D.roi_range = prm.roi_range;
    D.pix = B.pix.as<int>(); D.owner = B.owner.as<int>(); D.range_mat = B.range.as<float>(); D.ground = B.ground.as<unsigned char>();
    D.unc_count = B.unc.as<int>();
    D.unc_pts = reinterpret_cast<float4 *>(B.unc.as<unsigned char>() + 16);
    D.unc_gnd = D.unc_pts + n;
    MLH_LAUNCH(seg_project_kernel, dim3((n + 255) / 256), dim3(256), 0, st, D);
    MLH_HIP(ctx, hipGetLastError());
    int n_undecided_pts = 0;
    {
        // the points whose bin the device left open: fetched (counters + the first records in one copy), decided with the host's libm, sent back, claimed
        int *hc = static_cast<int *>(B.h_unc);
        float4 *hp = reinterpret_cast<float4 *>(static_cast<unsigned char *>(B.h_unc) + 16);
        MLH_HIP(ctx, hipMemcpyAsync(B.h_unc, B.unc.p, 16 + sizeof(float4) * size_t(std::min(n, SEG_UNC_FIRST)), hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, stream_wait_spin(ctx));
        n_undecided_pts = std::min(hc[0], n);
        if (n_undecided_pts > SEG_UNC_FIRST) {
            MLH_HIP(ctx, hipMemcpyAsync(hp + SEG_UNC_FIRST, D.unc_pts + SEG_UNC_FIRST, sizeof(float4) * size_t(n_undecided_pts - SEG_UNC_FIRST), hipMemcpyDeviceToHost, st));
            MLH_HIP(ctx, stream_wait_spin(ctx));
        }
        if (n_undecided_pts > 0) {
            // the verdicts overwrite the records they answer, in place in the pinned block (an int2 per float4 slot: never ahead of the record being read), and go
            // back from there: no pageable staging, nothing to wait for
            int2 *fix = reinterpret_cast<int2 *>(hp);
            for (int k = 0; k < n_undecided_pts; ++k) {
                const float4 rec = hp[k];
                int idx;
                std::memcpy(&idx, &rec.w, sizeof(int));
                fix[k] = make_int2(idx, seg_pixel_host(rec.x, rec.y, rec.z, S, prm.roi_range));
            }
            MLH_HIP(ctx, B.fix.ensure(sizeof(int2) * size_t(n_undecided_pts)));
            MLH_HIP(ctx, hipMemcpyAsync(B.fix.p, fix, sizeof(int2) * size_t(n_undecided_pts), hipMemcpyHostToDevice, st));
            MLH_LAUNCH(seg_apply_fix_kernel, dim3((n_undecided_pts + 255) / 256), dim3(256), 0, st, (const int2 *)B.fix.as<int2>(), n_undecided_pts, D.pix, D.owner);
        }
    }
    MLH_LAUNCH(seg_owner_fix_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, D.owner, npx);
    MLH_LAUNCH(seg_image_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, D);
    MLH_HIP(ctx, hipGetLastError());
    // MLH_SEG_TIMING=1: one line per call on stderr with the wall time of the call's phases (how much the host hop of the cluster search costs)
    static const bool seg_timing = std::getenv("MLH_SEG_TIMING") != nullptr;
    const auto tp0 = std::chrono::steady_clock::now();
    SegHost H;
    { const int hrc = seg_host_carve(ctx, B, npx, H); if (hrc) return hrc; }
    MLH_HIP(ctx, hipMemcpyAsync(H.range, B.range.p, sizeof(float) * size_t(npx), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipMemcpyAsync(H.owner, B.owner.p, sizeof(int) * size_t(npx), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipMemcpyAsync(H.ground, B.ground.p, size_t(npx), hipMemcpyDeviceToHost, st));
    // the cluster search's angle verdicts, made on the device from the range image (seg_edge_kernel) with the host's own sin / cos / tan table
    float t_cos[4], t_sin[4];
    {
        const float ay64[2] = {float(0.333 / 180.0 * M_PI), float(0.5 / 180.0 * M_PI)};
        const float t_alpha[4] = {0.f, S.alphax, S.is64 ? ay64[0] : S.alphay, S.is64 ? ay64[1] : S.alphay};
        for (int k = 0; k < 4; ++k) { t_cos[k] = std::cos(t_alpha[k]); t_sin[k] = std::sin(t_alpha[k]); }
        SegEdge E;
        MLH_HIP(ctx, B.edge.ensure(size_t(npx)));
        E.range_mat = B.range.as<float>(); E.edge = B.edge.as<unsigned char>(); E.vs = vs; E.hs = hs; E.is64 = S.is64;
        E.theta_simple = (prm.segment_theta > 0.01f && prm.segment_theta < 1.5f) ? 1 : 0;
        E.tan_theta = E.theta_simple ? std::tan(prm.segment_theta) : 0.f;
        for (int k = 0; k < 4; ++k) { E.t_cos[k] = t_cos[k]; E.t_sin[k] = t_sin[k]; }
        MLH_LAUNCH(seg_edge_kernel, dim3((npx + 255) / 256), dim3(256), 0, st, E);
        MLH_HIP(ctx, hipMemcpyAsync(H.edge, B.edge.p, size_t(npx), hipMemcpyDeviceToHost, st));
    }
    int n_undecided_gnd = 0;
    {
        // the ground pairs within the margin of 10 degrees: decided here, with the host's libm, straight into the ground image the cluster search reads
        int *hc = static_cast<int *>(B.h_unc);
        float4 *hg = reinterpret_cast<float4 *>(static_cast<unsigned char *>(B.h_unc) + 16);
        MLH_HIP(ctx, hipMemcpyAsync(hc, B.unc.p, 16, hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, hipMemcpyAsync(hg, D.unc_gnd, sizeof(float4) * 2 * size_t(std::min(npx, SEG_UNC_FIRST)), hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, stream_wait_spin(ctx));
        n_undecided_gnd = std::min(hc[1], npx);
        if (n_undecided_gnd > SEG_UNC_FIRST) {
            MLH_HIP(ctx, hipMemcpyAsync(hg + 2 * SEG_UNC_FIRST, D.unc_gnd + 2 * SEG_UNC_FIRST, sizeof(float4) * 2 * size_t(n_undecided_gnd - SEG_UNC_FIRST), hipMemcpyDeviceToHost, st));
            MLH_HIP(ctx, hipStreamSynchronize(st));
        }
        for (int k = 0; k < n_undecided_gnd; ++k) {
            int pxl;
            std::memcpy(&pxl, &hg[2 * k].w, sizeof(int));
            if (pxl >= 0 && pxl + hs < npx && seg_ground_host(hg[2 * k], hg[2 * k + 1])) { H.ground[size_t(pxl)] = 1; H.ground[size_t(pxl) + hs] = 1; }
        }
    }
    const auto tp1 = std::chrono::steady_clock::now();
    for (int p = 0; p < npx; ++p) H.label[p] = (H.owner[p] == INT_MAX) ? -1 : (H.ground[p] ? 1 : 0);
    seg_clusters(S, prm, H, t_cos, t_sin);
    const auto tp2 = std::chrono::steady_clock::now();
    int hs2 = 1;
    while (hs2 < hs) hs2 <<= 1;
    static const bool host_rows = std::getenv("MLH_SEG_HOST_ROWS") != nullptr;       // (A/B runs: the round-4 host assembly)
    // seg_rows_kernel's dynamic LDS (20 bytes per padded row pixel: 80 KB at 4 096 columns) has to be granted per DEVICE, once, before a launch that needs it; a
    // device that does not grant it (a part with less LDS per compute unit) takes the host assembly below instead of failing the call
    bool device_rows = hs2 <= SEG_ROW_MAX && !host_rows;
    if (device_rows) device_rows = seg_rows_lds_granted(ctx->device, size_t(20) * size_t(hs2));
    if (device_rows) {
        // ---- rows, erasure, concatenation and gather on the device (seg_rows_kernel / seg_rows_gather_kernel); the host keeps the outlier list only
        std::vector<int> outlier_idx, outlier_row;
        if (prm.segment_flag)
            for (size_t wd = 0; wd < H.outmask_words; ++wd) {
                unsigned m = H.outmask[wd];
                while (m) {
                    const size_t px = wd * 32 + size_t(__builtin_ctz(m));
                    m &= m - 1;
                    if ((px % size_t(hs)) % 5 == 0) { outlier_idx.push_back(H.owner[px]); outlier_row.push_back(int(px / size_t(hs))); }
                }
            }
        const size_t mask_bytes = sizeof(unsigned) * H.outmask_words;
        MLH_HIP(ctx, B.outmask.ensure(mask_bytes));
        MLH_HIP(ctx, B.row_cnt.ensure(sizeof(int) * size_t(vs)));
        MLH_HIP(ctx, B.keep.ensure(sizeof(int) * size_t(npx)));
        if (B.h_rows_cap < sizeof(int) * size_t(vs + 2)) {
            if (B.h_rows) (void)hipHostFree(B.h_rows);
            B.h_rows = nullptr; B.h_rows_cap = 0;
            MLH_HIP(ctx, hipHostMalloc(&B.h_rows, sizeof(int) * size_t(vs + 2) * 2, hipHostMallocDefault));
            B.h_rows_cap = sizeof(int) * size_t(vs + 2) * 2;
        }
        ScanBuf &sb = ctx->scan;
        sb.extracted = false; sb.voxelised = false; sb.h_lists_valid = sb.h_vox_valid = false;
        MLH_HIP(ctx, sb.pts.ensure(sizeof(float4) * size_t(std::max(std::min(n, npx), 1))));
        MLH_HIP(ctx, sb.start.ensure(sizeof(int) * size_t(vs)));
        MLH_HIP(ctx, sb.end.ensure(sizeof(int) * size_t(vs)));
        sb.end_alias = nullptr;
        MLH_HIP(ctx, hipMemcpyAsync(B.outmask.p, H.outmask, mask_bytes, hipMemcpyHostToDevice, st));           // (pinned: the next call rewrites it behind this call's waits)
        SegRows R;
        R.owner = B.owner.as<int>(); R.outmask = B.outmask.as<unsigned>(); R.kept = B.keep.as<int>(); R.row_cnt = B.row_cnt.as<int>();
        R.vs = vs; R.hs = hs; R.hs2 = hs2; R.segment_flag = prm.segment_flag ? 1 : 0;
        const size_t lds = size_t(20) * size_t(hs2);
        MLH_LAUNCH(seg_rows_kernel, dim3(vs), dim3(SEG_ROW_TPB), lds, st, R);
        int *h_rows = static_cast<int *>(B.h_rows);
        MLH_LAUNCH(seg_rows_gather_kernel, dim3(4, vs), dim3(256), 0, st, D, (const int *)B.keep.as<int>(), (const int *)B.row_cnt.as<int>(), sb.pts.as<float4>(),
                   sb.start.as<int>(), sb.end.as<int>(), h_rows);
        MLH_HIP(ctx, hipGetLastError());
        MLH_HIP(ctx, stream_wait_spin(ctx));
        const auto tq3 = std::chrono::steady_clock::now();
        const int n_keep = h_rows[vs], first_kept = h_rows[vs + 1];
        std::vector<int> hstart(vs), hend(vs);
        int off = 0, max_len = 0, row0 = -1;
        for (int r = 0; r < vs; ++r) {
            hstart[r] = off + 5; off += h_rows[r]; hend[r] = off - 6;
            if (row0 < 0 && h_rows[r] > 0) row0 = r;
            if (hend[r] - hstart[r] >= 6) max_len = std::max(max_len, hend[r] - hstart[r]);
        }
        if (cloud_out && n_keep > 0) {
            MLH_HIP(ctx, hipMemcpyAsync(cloud_out, sb.pts.p, sizeof(float4) * size_t(n_keep), hipMemcpyDeviceToHost, st));
            MLH_HIP(ctx, hipStreamSynchronize(st));
        }
        if (seg_timing) {
            const auto tp3 = std::chrono::steady_clock::now();
            auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
            std::fprintf(stderr, "[mlh_segment_cloud] undecided on the device, decided with the host's libm: %d points, %d ground pairs\n", n_undecided_pts, n_undecided_gnd);
            std::fprintf(stderr, "[mlh_segment_cloud] n %d: kernels + 3 image copies to the host %.1f us | host cluster search (BFS, queue order) %.1f us | rows on the device: mask up + "
                                 "sort / erase / compact / gather + wait %.1f us | cloud to the host %.1f us\n", n, us(tp0, tp1), us(tp1, tp2), us(tp2, tq3), us(tq3, tp3));
        }
        sb.n = n_keep; sb.n_rings = vs; sb.max_ring_len = max_len;
        if (n_out) *n_out = n_keep;
        if (scan_start) std::memcpy(scan_start, hstart.data(), sizeof(int) * size_t(vs));
        if (scan_end) std::memcpy(scan_end, hend.data(), sizeof(int) * size_t(vs));
        if (n_outlier) *n_outlier = int(outlier_idx.size()) + (n_keep > 0 ? 1 : 0);
        if (outlier_out) {
            size_t k = 0;
            auto put = [&](int i, int row) -> int {
                if (k >= size_t(outlier_capacity)) return 0;       // the caller's buffer is full: *n_outlier tells it how many rows there are
                float rec[4] = {0, 0, 0, 0};
                if (mem == MLH_MEM_HOST) {
                    const unsigned char *q = static_cast<const unsigned char *>(points) + size_t(i) * stride;
                    std::memcpy(rec, q, 12);
                    if (intensity_off >= 0) std::memcpy(rec + 3, q + intensity_off, 4);
                } else {
                    if (hipMemcpy(rec, src + size_t(i) * stride, 12, hipMemcpyDeviceToHost) != hipSuccess) return 1;
                    if (intensity_off >= 0 && hipMemcpy(rec + 3, src + size_t(i) * stride + intensity_off, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
                }
                rec[3] += float(row);
                std::memcpy(outlier_out + 4 * k, rec, 16);
                ++k;
                return 0;
            };
            for (size_t q = 0; q < outlier_idx.size(); ++q) if (put(outlier_idx[q], outlier_row[q])) return fail(ctx, MLH_ERR_HIP, "outlier fetch");
            if (n_keep > 0 && first_kept >= 0 && put(first_kept, row0)) return fail(ctx, MLH_ERR_HIP, "outlier fetch");
        }
        return MLH_OK;
    }
    // ---- host assembly (rows longer than SEG_ROW_MAX pixels; MLH_SEG_HOST_ROWS=1)
    // the rows as the reference fills them: every pixel owner, in input order; cloud_scan_order = its position at fill time. One pass over the input
    // indices (a pixel's owner is an input index: bucket the pixels by owner, walk the indices upwards) gives every row already sorted and every
    // pixel its rank; the outlier erasure -- "erase what is NOW at the position recorded at fill time", stale positions included (U2) -- runs on an
    // order-statistic structure per row instead of vector::erase (round 2: 5 ms of sort / lower_bound / memmove per 64-ring scan)
    std::vector<int> pixel_of(size_t(n), -1);
    for (int p = 0; p < npx; ++p) { const int o = H.owner[p]; if (o != INT_MAX && o >= 0 && o < n) pixel_of[size_t(o)] = p; }
    std::vector<std::vector<int>> rows(vs);
    std::vector<int> order(npx, 0);
    for (int r = 0; r < vs; ++r) rows[r].reserve(size_t(hs));
    for (int o = 0; o < n; ++o) {
        const int p = pixel_of[size_t(o)];
        if (p < 0) continue;
        std::vector<int> &v = rows[p / hs];
        order[p] = int(v.size());
        v.push_back(o);
    }
    const auto tq1 = std::chrono::steady_clock::now();
    std::vector<int> outlier_idx;
    int n_linear_rows = 0, n_pool_rows = 0, n_erased = 0;
    if (prm.segment_flag) {
        std::vector<int> epos;
        std::vector<char> dead;
        std::vector<size_t> first_run;
        for (int r = 0; r < vs; ++r) {
            // the positions to erase, in the order the reference erases (columns ascending)
            epos.clear();
            int descents = 0;
            for (int c = 0; c < hs; ++c)
                if (H.label[size_t(r) * hs + c] == 999999) {
                    const int pos = order[size_t(r) * hs + c];
                    if (!epos.empty() && pos <= epos.back()) ++descents;
                    epos.push_back(pos);
                    if (c % 5 == 0) outlier_idx.push_back(H.owner[size_t(r) * hs + c]);
                }
            if (epos.empty()) continue;
            // A driver delivers a ring's points in firing order: a pixel's fill position grows with its column, with ONE step down where the sweep's first
            // column sits. The erased positions then come as one or two strictly ascending runs, and "erase what is NOW at position p" resolves in a linear
            // pass. First run: the k-th erasure removes the ORIGINAL element p + k (the k elements the run already removed were all in front of it).
            // Second run (it restarts at the front of the row): the j-th erasure removes the original element t with t - (removed originals below t) = q, i.e.
            // t = q + j + c, c = how many of the FIRST run's originals lie at or below t -- both lists ascend, so c only ever grows: a merge. A position at or
            // past the current size erases nothing (U2). More than two runs (an unordered cloud) go to the order-statistic structure below.
            if (descents <= 1) {
                std::vector<int> &row = rows[r];
                const size_t size0 = row.size();
                dead.assign(size0, 0);
                first_run.clear();
                size_t erased = 0, j = 0, c = 0;
                bool second = false;
                for (size_t e = 0; e < epos.size(); ++e) {
                    if (e > 0 && epos[e] <= epos[e - 1]) second = true;
                    const int pos = epos[e];
                    if (pos < 0 || size_t(pos) >= size0 - erased) continue;
                    size_t t;
                    if (!second) {
                        t = size_t(pos) + first_run.size();
                        first_run.push_back(t);
                    } else {
                        t = size_t(pos) + j + c;
                        while (c < first_run.size() && first_run[c] <= t) { ++c; ++t; }
                        ++j;
                    }
                    dead[t] = 1;
                    ++erased;
                }
                ++n_linear_rows; n_erased += int(erased);
                if (erased) {
                    size_t w = 0;
                    for (size_t i = 0; i < size0; ++i) if (!dead[i]) row[w++] = row[i];
                    row.resize(w);
                }
                continue;
            }
            ++n_pool_rows;
            AlivePool alive(rows[r].size());
            bool any = false;
            for (const int pos : epos)
                if (pos >= 0 && size_t(pos) < alive.size()) { alive.erase_index(alive.at(size_t(pos))); any = true; }       // stale position, as it is (U2)
            if (any) {
                std::vector<int> kept;
                kept.reserve(alive.size());
                for (size_t k = 0; k < rows[r].size(); ++k) if (alive.contains(k)) kept.push_back(rows[r][k]);
                rows[r].swap(kept);
            }
        }
    }
    const auto tq2 = std::chrono::steady_clock::now();
    std::vector<int> keep, hstart(vs), hend(vs);
    for (int r = 0; r < vs; ++r) {
        hstart[r] = int(keep.size()) + 5;
        keep.insert(keep.end(), rows[r].begin(), rows[r].end());
        hend[r] = int(keep.size()) - 6;
    }
    const int n_keep = int(keep.size());
    const auto tq3 = std::chrono::steady_clock::now();
    // stage the scan exactly as mlh_scan_upload would
    ScanBuf &sb = ctx->scan;
    sb.extracted = false; sb.voxelised = false; sb.h_lists_valid = sb.h_vox_valid = false;
    MLH_HIP(ctx, sb.pts.ensure(sizeof(float4) * size_t(std::max(n_keep, 1))));
    MLH_HIP(ctx, sb.start.ensure(sizeof(int) * size_t(vs)));
    MLH_HIP(ctx, sb.end.ensure(sizeof(int) * size_t(vs)));
    sb.end_alias = nullptr;                                   // this path fills the two tables separately
    MLH_HIP(ctx, B.keep.ensure(sizeof(int) * size_t(std::max(n_keep, 1))));
    if (n_keep > 0) {
        MLH_HIP(ctx, hipMemcpyAsync(B.keep.p, keep.data(), sizeof(int) * size_t(n_keep), hipMemcpyHostToDevice, st));
        MLH_LAUNCH(seg_gather_kernel, dim3((n_keep + 255) / 256), dim3(256), 0, st, D, (const int *)B.keep.as<int>(), n_keep, sb.pts.as<float4>());
        MLH_HIP(ctx, hipGetLastError());
    }
    MLH_HIP(ctx, hipMemcpyAsync(sb.start.p, hstart.data(), sizeof(int) * size_t(vs), hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipMemcpyAsync(sb.end.p, hend.data(), sizeof(int) * size_t(vs), hipMemcpyHostToDevice, st));
    if (cloud_out && n_keep > 0) MLH_HIP(ctx, hipMemcpyAsync(cloud_out, sb.pts.p, sizeof(float4) * size_t(n_keep), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    if (seg_timing) {
        const auto tp3 = std::chrono::steady_clock::now();
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        std::fprintf(stderr, "[mlh_segment_cloud] undecided on the device, decided with the host's libm: %d points, %d ground pairs\n", n_undecided_pts, n_undecided_gnd);
        std::fprintf(stderr, "[mlh_segment_cloud] n %d: kernels + 3 image copies to the host %.1f us | host cluster search (BFS, queue order) %.1f us | row assembly + keep list + gather + sync %.1f us\n",
                     n, us(tp0, tp1), us(tp1, tp2), us(tp2, tp3));
        std::fprintf(stderr, "    rows %.1f | erase %.1f (rows resolved in one linear pass %d, through the order-statistic pool %d; erased %d) | keep %.1f | upload + gather + sync %.1f us\n",
                     us(tp2, tq1), us(tq1, tq2), n_linear_rows, n_pool_rows, n_erased, us(tq2, tq3), us(tq3, tp3));
    }
    int max_len = 0;
    for (int r = 0; r < vs; ++r) if (hend[r] - hstart[r] >= 6) max_len = std::max(max_len, hend[r] - hstart[r]);
    sb.n = n_keep; sb.n_rings = vs; sb.max_ring_len = max_len;
    if (n_out) *n_out = n_keep;
    if (scan_start) std::memcpy(scan_start, hstart.data(), sizeof(int) * size_t(vs));
    if (scan_end) std::memcpy(scan_end, hend.data(), sizeof(int) * size_t(vs));
    // laser_cloud_outlier: every fifth-column outlier pixel's point, then the first point of the output cloud (hpp:378, 391)
    if (n_outlier) *n_outlier = int(outlier_idx.size()) + (n_keep > 0 ? 1 : 0);
    if (outlier_out) {
        // the few outlier records are assembled from the caller's cloud when it is on the host, else fetched point by point
        size_t k = 0;
        auto put = [&](int i, int row) -> int {
            if (k >= size_t(outlier_capacity)) return 0;       // the caller's buffer is full: *n_outlier tells it how many rows there are
            float rec[4] = {0, 0, 0, 0};
            if (mem == MLH_MEM_HOST) {
                const unsigned char *q = static_cast<const unsigned char *>(points) + size_t(i) * stride;
                std::memcpy(rec, q, 12);
                if (intensity_off >= 0) std::memcpy(rec + 3, q + intensity_off, 4);
            } else {
                if (hipMemcpy(rec, src + size_t(i) * stride, 12, hipMemcpyDeviceToHost) != hipSuccess) return 1;
                if (intensity_off >= 0 && hipMemcpy(rec + 3, src + size_t(i) * stride + intensity_off, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            }
            rec[3] += float(row);
            std::memcpy(outlier_out + 4 * k, rec, 16);
            ++k;
            return 0;
        };
        std::vector<int> pix_of(outlier_idx.size());
        for (int r = 0, q = 0; r < vs && q < int(outlier_idx.size()); ++r)
            for (int c = 0; c < hs && q < int(outlier_idx.size()); ++c)
                if (prm.segment_flag && H.label[size_t(r) * hs + c] == 999999 && c % 5 == 0) { if (put(outlier_idx[q], r)) return fail(ctx, MLH_ERR_HIP, "outlier fetch"); ++q; }
        if (n_keep > 0) {
            int row0 = 0;
            while (row0 < vs && rows[row0].empty()) ++row0;
            if (put(keep[0], row0)) return fail(ctx, MLH_ERR_HIP, "outlier fetch");
        }
    }
    return MLH_OK;
}

}  // namespace mlh
