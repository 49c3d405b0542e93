// Local-map index build for gfx950: a dense cell grid over the map cloud, points counting-sorted by cell.
// Replaces the FLANN kd-tree build the reference performs every frame
// (pcl::KdTreeFLANN::setInputCloud, estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:433-434).
//
// Layout in HBM: raw[n] and sorted[n] as float4 {x, y, z, original index}; cell_start[ncell+1] (int32, exclusive prefix),
// cells ordered x fastest so the 3 x-adjacent cells of a query form ONE contiguous run of `sorted` (9 runs per query).
// One steady-state build = 4 launches for BOTH maps together (count, chunk-local scan, add, scatter); all streaming:
//   count   16 B/pt read + 1 returning atomic + 4 B rank      scan+add   16 B/cell + 4 B/cell (twin clear)
//   scatter 16 B + 4 B read, 16 B written per point, no atomics
// The count lands in cell_start[c+1] and the value the atomic returns is the point's rank inside its cell; the in-place
// INCLUSIVE scan turns slot c+1 into start[c+1], so with cell_start[0] = 0 the array is the exclusive prefix the queries read
// and the scatter writes to start[c] + rank. Two cell arrays alternate: the add pass of one build clears the other array,
// so the next build starts from zeros without a clear launch; with few chunks (<= 4096) the add pass also sums the chunk
// totals before it itself instead of a separate single-workgroup scan launch.
#include "ctx.hpp"
#include <chrono>
#include <cmath>
#include <climits>

namespace mlh {

__device__ __forceinline__ int float_order_key(float f)
{
    int k = __float_as_int(f);
    return k >= 0 ? k : k ^ 0x7fffffff;
}
static inline float key_to_float(int k)
{
    int b = k >= 0 ? k : k ^ 0x7fffffff;
    float f;
    std::memcpy(&f, &b, sizeof(f));
    return f;
}

// bounding box as order-preserving integer keys: one partial record per workgroup (no atomics: a few thousand wavefronts hammering six
// words cost 0.5 ms on a 500 k map), folded by the host, which needs the box for the grid dimensions anyway
constexpr int BOUNDS_BLOCKS = 128;
constexpr float GRID_MARGIN_XY = 2.f, GRID_MARGIN_Z = 1.f;
__global__ __launch_bounds__(256) void bounds_kernel(const float4 *__restrict__ pts, int n, int *__restrict__ partial)
{
    __shared__ int lds[4][6];
    int mn[3] = {INT_MAX, INT_MAX, INT_MAX}, mx[3] = {INT_MIN, INT_MIN, INT_MIN};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
        int kx = float_order_key(p.x), ky = float_order_key(p.y), kz = float_order_key(p.z);
        mn[0] = min(mn[0], kx); mx[0] = max(mx[0], kx);
        mn[1] = min(mn[1], ky); mx[1] = max(mx[1], ky);
        mn[2] = min(mn[2], kz); mx[2] = max(mx[2], kz);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mn[d] = min(mn[d], __shfl_xor(mn[d], off));
            mx[d] = max(mx[d], __shfl_xor(mx[d], off));
        }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { lds[threadIdx.x >> 6][d] = mn[d]; lds[threadIdx.x >> 6][3 + d] = mx[d]; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        int r = lds[0][d];
        for (int w = 1; w < 4; ++w) r = d < 3 ? min(r, lds[w][d]) : max(r, lds[w][d]);
        partial[blockIdx.x * 6 + d] = r;
    }
}

// ---- one job per map; kernels take both jobs and split their grid between them
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = 256 * SCAN_ITEMS;
constexpr int FUSED_SUMS_MAX = 4096;   // chunk count up to which scan_add sums the chunk totals itself (no scan_sums launch)

struct GridJob {
    const float4 *raw;
    float4 *sorted;
    int *cell_start;       // ncell + 1: the array being built
    int *cell_next;        // the twin array, cleared here for the NEXT build
    int *rank;             // n: arrival rank of each point inside its cell
    int *block_sums;
    long long *occ_host;   // pinned host mirror of occ[0..1] (or null)
    int occ_parts;         // partials behind occ[2]
    long long *occ;        // occupancy statistics of this build (null: not wanted): [0] non-empty cells, [1] sum of squared cell populations,
                           // then one (cells, squares) partial per scanning workgroup
    int n;
    int ncell;
    float ox, oy, oz, inv_h;
    int nx, ny, nz;
    int nb_pts;            // workgroups streaming the points
    int nb_scan;           // workgroups scanning the cells
    int need_zero;         // the array being built was not cleared by the previous build
    int sums_scanned;      // block_sums already hold exclusive prefixes (large grids: scan_sums_kernel ran)
};
struct GridJobs {
    GridJob j[2];
    // optional: the staging pass's fit flags leave for the host in the first launch of the build (one thread), instead of a launch of their own
    int *pub_oob; HostPublish *pub; unsigned long long pub_seq;
    int pub_stage;         // 0: cell_count_kernel publishes (the clouds were packed by the launch before it); 1: scan_local_kernel does (pack_count_kernel packed them)
};

// a frame's local map arrives as a cloud of caller records: packed to float4 {x,y,z,index} and, in the same pass, checked against the grid box already set up for
// this kind (flag word, OR-ed)
struct PackJob { const unsigned char *src; float4 *out; int n, stride; float lo[3], hi[3]; int nb; };
struct PackJobs { PackJob j[2]; int *oob; };

__device__ __forceinline__ int cell_of(const GridJob &J, const float4 &p)
{
    float fx = fminf(fmaxf(floorf((p.x - J.ox) * J.inv_h), 0.f), float(J.nx - 1));
    float fy = fminf(fmaxf(floorf((p.y - J.oy) * J.inv_h), 0.f), float(J.ny - 1));
    float fz = fminf(fmaxf(floorf((p.z - J.oz) * J.inv_h), 0.f), float(J.nz - 1));
    return (int(fz) * J.ny + int(fy)) * J.nx + int(fx);
}

// cell_start lives 3 ints into its allocation, so A = cell_start + 1 (the array the scan works on) is 16-byte aligned
__device__ __forceinline__ void zero_chunk(int *cells, int ncell, int b)
{
    int4 *A4 = reinterpret_cast<int4 *>(cells + 1);
    const int base4 = b * (SCAN_CHUNK / 4) + threadIdx.x * 2;           // in int4 units
    const int n4 = (ncell + 3) / 4;                                      // allocation is padded to a multiple of 4 (+4)
    const int4 z = make_int4(0, 0, 0, 0);
    if (base4 < n4) A4[base4] = z;
    if (base4 + 1 < n4) A4[base4 + 1] = z;
    if (b == 0 && threadIdx.x == 0) cells[0] = 0;
}

__global__ __launch_bounds__(256) void zero_cells_kernel(GridJobs G)
{
    const int job = blockIdx.x >= G.j[0].nb_scan ? 1 : 0;
    const GridJob &J = G.j[job];
    if (!J.need_zero) return;
    zero_chunk(J.cell_start, J.ncell, job ? blockIdx.x - G.j[0].nb_scan : blockIdx.x);
}

// per-cell counts land in cell_start[c + 1]; the value the atomic returns is the point's rank inside its cell.
// Device-scope atomics execute at the memory side on a multi-XCD part and bound this kernel, so lanes are merged first: the
// map arrives ordered by voxel index, x fastest (it is the output of the voxel filter), i.e. neighbouring lanes usually fall
// into the same cell -- every run of equal cells inside a wavefront issues ONE atomic for the whole run (a cloud in random
// order degrades to one atomic per lane, as before).
__global__ __launch_bounds__(256) void cell_count_kernel(GridJobs G)
{
    const int job = blockIdx.x >= G.j[0].nb_pts ? 1 : 0;
    const GridJob &J = G.j[job];
    const int b = job ? blockIdx.x - G.j[0].nb_pts : blockIdx.x;
    const int lane = threadIdx.x & 63;
    if (G.pub && G.pub_stage == 0 && blockIdx.x == 0 && threadIdx.x == 0) {   // the packed clouds' fit flags: final before this launch started
        G.pub->done = *G.pub_oob;
        *G.pub_oob = 0;
        __hip_atomic_store(&G.pub->seq, G.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (int base = b * 256; base < J.n; base += J.nb_pts * 256) {       // uniform over the workgroup
        const int i = base + threadIdx.x;
        const bool valid = i < J.n;
        const int c = valid ? cell_of(J, J.raw[i]) : -1 - lane;           // distinct negatives: never merged
        const int prev = __shfl_up(c, 1);
        const bool head = (lane == 0) || (c != prev);
        const unsigned long long H = __ballot(head);
        const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const int my_head = 63 - __clzll(H & upto);
        const unsigned long long above = (my_head == 63) ? 0ull : (H & ~((2ull << my_head) - 1ull));
        const int next = above ? (__ffsll((long long)above) - 1) : 64;
        int first = 0;
        if (head && valid) first = atomicAdd(&J.cell_start[c + 1], next - my_head);
        first = __shfl(first, my_head);
        if (valid) J.rank[i] = first + (lane - my_head);
    }
}

// pack + fit check + count in ONE pass over the caller's records (every staged kind keeps its grid geometry: the steady state of a mapper between keyframes): the
// record is read once, the packed point goes to raw[] and straight into cell_of() -- pack_check_kernel and cell_count_kernel as one launch (~6 us of a frame's
// index build and a kernel boundary). A point outside the box is counted in the clamped cell like any other; the flag it raises makes the host lay out a new box
// and build again, as before. The flags leave for the host from the NEXT launch (scan_local_kernel, pub_stage 1).
__global__ __launch_bounds__(256) void pack_count_kernel(GridJobs G, PackJobs P)
{
    const int job = blockIdx.x >= G.j[0].nb_pts ? 1 : 0;
    const GridJob &J = G.j[job];
    const PackJob &Q = P.j[job];
    const int b = job ? blockIdx.x - G.j[0].nb_pts : blockIdx.x;
    const int lane = threadIdx.x & 63;
    bool bad = false;
    for (int base = b * 256; base < J.n; base += J.nb_pts * 256) {       // uniform over the workgroup
        const int i = base + threadIdx.x;
        const bool valid = i < J.n;
        int c = -1 - lane;                                               // distinct negatives: never merged
        if (valid) {
            const float *rec = reinterpret_cast<const float *>(Q.src + size_t(i) * Q.stride);
            const float x = rec[0], y = rec[1], z = rec[2];
            const float4 p = make_float4(x, y, z, __int_as_float(i));
            Q.out[i] = p;
            bad = bad || !(x >= Q.lo[0] && x < Q.hi[0] && y >= Q.lo[1] && y < Q.hi[1] && z >= Q.lo[2] && z < Q.hi[2]);   // NaN -> bad
            c = cell_of(J, p);
        }
        const int prev = __shfl_up(c, 1);
        const bool head = (lane == 0) || (c != prev);
        const unsigned long long H = __ballot(head);
        const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
        const int my_head = 63 - __clzll(H & upto);
        const unsigned long long above = (my_head == 63) ? 0ull : (H & ~((2ull << my_head) - 1ull));
        const int next = above ? (__ffsll((long long)above) - 1) : 64;
        int first = 0;
        if (head && valid) first = atomicAdd(&J.cell_start[c + 1], next - my_head);
        first = __shfl(first, my_head);
        if (valid) J.rank[i] = first + (lane - my_head);
    }
    if (__ballot(bad) != 0ull && lane == 0) atomicOr(P.oob, 1 << job);
}

__device__ __forceinline__ int block_exclusive_scan_256(int v, int *lds, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wave) base += lds[w];
    total = lds[0] + lds[1] + lds[2] + lds[3];
    __syncthreads();
    return base + incl - v;
}

// in-place INCLUSIVE scan of A[i] = cell_start[i + 1], i in [0, ncell): chunk-local part (8 ints = 2 x int4 per thread).
// Inclusive, so that cell_start[c + 1] = start of cell c + 1 and (with cell_start[0] = 0) the array is the exclusive prefix.
__global__ __launch_bounds__(256) void scan_local_kernel(GridJobs G)
{
    __shared__ int lds[4];
    __shared__ long long lds_occ[8];
    const int job = blockIdx.x >= G.j[0].nb_scan ? 1 : 0;
    const GridJob &J = G.j[job];
    const int b = job ? blockIdx.x - G.j[0].nb_scan : blockIdx.x;
    if (G.pub && G.pub_stage == 1 && blockIdx.x == 0 && threadIdx.x == 0) {   // pack_count_kernel's fit flags: final before this launch started
        G.pub->done = *G.pub_oob;
        *G.pub_oob = 0;
        __hip_atomic_store(&G.pub->seq, G.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    int4 *A4 = reinterpret_cast<int4 *>(J.cell_start + 1);
    const int base = b * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    const int n_pad = ((J.ncell + 3) / 4) * 4;                           // padded tail holds zeros
    int4 v0 = make_int4(0, 0, 0, 0), v1 = v0;
    if (base < n_pad) v0 = A4[base / 4];
    if (base + 4 < n_pad) v1 = A4[base / 4 + 1];
    const int s = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
    long long occ_nz = 0, occ_sq = 0;                                    // occupancy statistics: one partial per workgroup, summed by occ_total()
    if (J.occ) {
        occ_nz = (v0.x > 0) + (v0.y > 0) + (v0.z > 0) + (v0.w > 0) + (v1.x > 0) + (v1.y > 0) + (v1.z > 0) + (v1.w > 0);
        occ_sq = (long long)v0.x * v0.x + (long long)v0.y * v0.y + (long long)v0.z * v0.z + (long long)v0.w * v0.w +
                 (long long)v1.x * v1.x + (long long)v1.y * v1.y + (long long)v1.z * v1.z + (long long)v1.w * v1.w;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { occ_nz += __shfl_xor(occ_nz, off); occ_sq += __shfl_xor(occ_sq, off); }
        if ((threadIdx.x & 63) == 0) { lds_occ[2 * (threadIdx.x >> 6)] = occ_nz; lds_occ[2 * (threadIdx.x >> 6) + 1] = occ_sq; }
    }
    int total;
    int ex = block_exclusive_scan_256(s, lds, total);
    int4 o0, o1;
    ex += v0.x; o0.x = ex; ex += v0.y; o0.y = ex; ex += v0.z; o0.z = ex; ex += v0.w; o0.w = ex;
    ex += v1.x; o1.x = ex; ex += v1.y; o1.y = ex; ex += v1.z; o1.z = ex; ex += v1.w; o1.w = ex;
    if (base < n_pad) A4[base / 4] = o0;
    if (base + 4 < n_pad) A4[base / 4 + 1] = o1;
    if (threadIdx.x == 0) {
        J.block_sums[b] = total;
        if (J.occ) {                                                     // the block scan above synchronised after the LDS writes
            J.occ[2 + 2 * b] = lds_occ[0] + lds_occ[2] + lds_occ[4] + lds_occ[6];
            J.occ[3 + 2 * b] = lds_occ[1] + lds_occ[3] + lds_occ[5] + lds_occ[7];
        }
    }
}

// sums the per-workgroup occupancy partials of one index build into occ[0..1] (256 threads, all of them)
__device__ __forceinline__ void occ_total(long long *occ, int n_part, long long *lds)
{
    long long nz = 0, sq = 0;
    for (int i = threadIdx.x; i < n_part; i += 256) { nz += occ[2 + 2 * i]; sq += occ[3 + 2 * i]; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { nz += __shfl_xor(nz, off); sq += __shfl_xor(sq, off); }
    if ((threadIdx.x & 63) == 0) { lds[2 * (threadIdx.x >> 6)] = nz; lds[2 * (threadIdx.x >> 6) + 1] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) { occ[0] = lds[0] + lds[2] + lds[4] + lds[6]; occ[1] = lds[1] + lds[3] + lds[5] + lds[7]; }
    __syncthreads();
}

// large grids only: exclusive scan of the chunk totals
__global__ __launch_bounds__(256) void scan_sums_kernel(GridJobs G)
{
    __shared__ int lds[4];
    const GridJob &J = G.j[blockIdx.x];
    if (!J.sums_scanned) return;
    int carry = 0;
    for (int start = 0; start < J.nb_scan; start += 256) {
        int i = start + threadIdx.x;
        int v = (i < J.nb_scan) ? J.block_sums[i] : 0;
        int total;
        int ex = block_exclusive_scan_256(v, lds, total);
        if (i < J.nb_scan) J.block_sums[i] = carry + ex;
        carry += total;
    }
}

// adds the totals of the preceding chunks (summed here when there are few chunks) and clears the twin array for the next build
__global__ __launch_bounds__(256) void scan_add_kernel(GridJobs G)
{
    __shared__ int lds[4];
    const int job = blockIdx.x >= G.j[0].nb_scan ? 1 : 0;
    const GridJob &J = G.j[job];
    const int b = job ? blockIdx.x - G.j[0].nb_scan : blockIdx.x;
    zero_chunk(J.cell_next, J.ncell, b);
    int add;
    if (J.sums_scanned) {
        add = J.block_sums[b];
    } else {
        int s = 0;
        for (int i = threadIdx.x; i < b; i += 256) s += J.block_sums[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
        __syncthreads();
        add = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    }
    if (b == 0) {                                                        // first chunk: offset 0 -- the idle workgroup totals the occupancy statistics
        if (J.occ) {
            __shared__ long long lds_occ[8];
            occ_total(J.occ, J.occ_parts, lds_occ);
            if (threadIdx.x == 0 && J.occ_host) { J.occ_host[0] = J.occ[0]; J.occ_host[1] = J.occ[1]; }   // pinned mirror, read by the NEXT staging call
        }
        return;
    }
    int4 *A4 = reinterpret_cast<int4 *>(J.cell_start + 1);
    const int base = b * SCAN_CHUNK + threadIdx.x * SCAN_ITEMS;
    const int n_pad = ((J.ncell + 3) / 4) * 4;
    if (base < n_pad) { int4 v = A4[base / 4]; v.x += add; v.y += add; v.z += add; v.w += add; A4[base / 4] = v; }
    if (base + 4 < n_pad) { int4 v = A4[base / 4 + 1]; v.x += add; v.y += add; v.z += add; v.w += add; A4[base / 4 + 1] = v; }
}

// no atomics: position = start of the cell + the rank taken while counting
__global__ __launch_bounds__(256) void scatter_kernel(GridJobs G)
{
    const int job = blockIdx.x >= G.j[0].nb_pts ? 1 : 0;
    const GridJob &J = G.j[job];
    const int b = job ? blockIdx.x - G.j[0].nb_pts : blockIdx.x;
    for (int i = b * 256 + threadIdx.x; i < J.n; i += J.nb_pts * 256) {
        const float4 p = J.raw[i];
        const int r = J.rank[i];
        J.sorted[J.cell_start[cell_of(J, p)] + r] = p;
    }
}

// bounds of one point set: the kernel + the copy of its partial records into hp (6 * BOUNDS_BLOCKS ints), nothing waited for
static int bounds_launch(mlh_ctx *ctx, MapGrid &g, int *hp)
{
    hipStream_t st = ctx->stream;
    const int grid_pts = std::min((g.n + 255) / 256, BOUNDS_BLOCKS);
    MLH_HIP(ctx, g.bounds.ensure(6 * BOUNDS_BLOCKS * sizeof(int)));
    MLH_LAUNCH(bounds_kernel, dim3(grid_pts), dim3(256), 0, st, g.raw.as<float4>(), g.n, g.bounds.as<int>());
    MLH_HIP(ctx, hipMemcpyAsync(hp, g.bounds.p, sizeof(int) * 6 * size_t(grid_pts), hipMemcpyDeviceToHost, st));
    return MLH_OK;
}

// ... and, once the stream has been waited for, the grid geometry and buffers that follow from them
static int bounds_finish(mlh_ctx *ctx, MapGrid &g, const int *hp, float min_match_sq_dis)
{
    const int n = g.n;
    const int grid_pts = std::min((n + 255) / 256, BOUNDS_BLOCKS);
    int hb[6] = {INT_MAX, INT_MAX, INT_MAX, INT_MIN, INT_MIN, INT_MIN};
    for (int blk = 0; blk < grid_pts; ++blk)
        for (int d = 0; d < 3; ++d) { hb[d] = std::min(hb[d], hp[blk * 6 + d]); hb[3 + d] = std::max(hb[3 + d], hp[blk * 6 + 3 + d]); }
    float mn[3], mx[3];
    for (int d = 0; d < 3; ++d) { mn[d] = key_to_float(hb[d]); mx[d] = key_to_float(hb[3 + d]); }
    for (int d = 0; d < 3; ++d)
        if (!std::isfinite(mn[d]) || !std::isfinite(mx[d])) return fail(ctx, MLH_ERR_INVALID, "map cloud has non-finite coordinates");
    g.h = std::sqrt(min_match_sq_dis) * 1.001f;
    g.inv_h = 1.0f / g.h;
    // the box is laid GRID_MARGIN cells wider than the cloud (one cell in z): the next frames' local maps -- the same keyframe window
    // moved a little -- then fit the geometry already set up, and mlh_map_set skips this host round trip (map_stage_and_build)
    g.ox = mn[0] - GRID_MARGIN_XY * g.h; g.oy = mn[1] - GRID_MARGIN_XY * g.h; g.oz = mn[2] - GRID_MARGIN_Z * g.h;
    g.nx = int(std::floor((mx[0] + GRID_MARGIN_XY * g.h - g.ox) * g.inv_h)) + 1;
    g.ny = int(std::floor((mx[1] + GRID_MARGIN_XY * g.h - g.oy) * g.inv_h)) + 1;
    g.nz = int(std::floor((mx[2] + GRID_MARGIN_Z * g.h - g.oz) * g.inv_h)) + 1;
    g.geom_sq_dis = min_match_sq_dis;
    g.geom_valid = true;
    g.ncell = (long long)g.nx * g.ny * g.nz;
    if (g.ncell >= (1ll << 31) - 2 * SCAN_CHUNK) return fail(ctx, MLH_ERR_UNSUPPORTED, "map extent needs more than 2^31 cells");
    const int nb = int((g.ncell + SCAN_CHUNK) / SCAN_CHUNK);   // covers ncell + 1 entries
    MLH_HIP(ctx, g.sorted.ensure(sizeof(float4) * size_t(n)));
    MLH_HIP(ctx, g.cell_start.ensure(sizeof(int) * size_t(g.ncell + 16)));   // 3 ints of lead-in (alignment of cell_start + 1) + padded tail
    MLH_HIP(ctx, g.cell_fill.ensure(sizeof(int) * size_t(g.ncell + 16)));    // the twin (cleared by one build for the next)
    MLH_HIP(ctx, g.cell_id.ensure(sizeof(int) * size_t(n)));                 // per-point rank inside its cell
    g.cur = 0;
    g.twin_clean = false;
    MLH_HIP(ctx, g.block_sums.ensure(sizeof(int) * size_t(nb + 1)));
    return MLH_OK;
}

// the build goes into the twin of the array the queries currently read when that twin is clean, else into array 0 after a clear
static GridJob make_job(MapGrid &g)
{
    GridJob J;
    std::memset(&J, 0, sizeof(J));
    const int dst = g.twin_clean ? 1 - g.cur : 0;
    J.need_zero = g.twin_clean ? 0 : 1;
    g.cur = dst;
    g.twin_clean = true;
    J.cell_start = g.cells(dst); J.cell_next = g.cells(1 - dst); J.rank = g.cell_id.as<int>();
    J.raw = g.raw.as<float4>(); J.sorted = g.sorted.as<float4>(); J.block_sums = g.block_sums.as<int>(); J.occ = g.want_occ ? g.occ.as<long long>() : nullptr; J.occ_parts = g.occ_parts; J.occ_host = g.want_occ ? g.occ_host : nullptr;
    J.n = g.n; J.ncell = int(g.ncell); J.ox = g.ox; J.oy = g.oy; J.oz = g.oz; J.inv_h = g.inv_h; J.nx = g.nx; J.ny = g.ny; J.nz = g.nz;
    J.nb_pts = std::min((g.n + 255) / 256, 4096);
    J.nb_scan = int((g.ncell + SCAN_CHUNK) / SCAN_CHUNK);
    J.sums_scanned = J.nb_scan > FUSED_SUMS_MAX ? 1 : 0;
    return J;
}

// (re)builds the index of up to two point sets in one set of launches
static int grid_build_grids_packed(mlh_ctx *ctx, MapGrid **grids, int n_grids, bool recompute_bounds, int *pub_oob, HostPublish *pub, unsigned long long pub_seq, const PackJobs *pack);
int grid_build_grids(mlh_ctx *ctx, MapGrid **grids, int n_grids, bool recompute_bounds, int *pub_oob, HostPublish *pub, unsigned long long pub_seq)
{
    return grid_build_grids_packed(ctx, grids, n_grids, recompute_bounds, pub_oob, pub, pub_seq, nullptr);
}

// pack: the staged clouds' records, when the pack + fit check is to ride in the build's first launch (map_stage_and_build; grids[k] <-> pack->j[k])
static int grid_build_grids_packed(mlh_ctx *ctx, MapGrid **grids, int n_grids, bool recompute_bounds, int *pub_oob, HostPublish *pub, unsigned long long pub_seq, const PackJobs *pack)
{
    hipStream_t st = ctx->stream;
    GridJobs G;
    std::memset(&G, 0, sizeof(G));
    G.pub_oob = pub_oob; G.pub = pub; G.pub_seq = pub_seq;
    int nj = 0;
    for (int k = 0; k < n_grids && k < 2; ++k)
        if (grids[k]->n <= 0 || !grids[k]->raw.p) return fail(ctx, MLH_ERR_STATE, "no points staged for this index");
    if (recompute_bounds) {
        // the bounding boxes of all point sets in ONE host round trip (which also completes whatever the caller enqueued before)
        int hp[2][6 * BOUNDS_BLOCKS];
        for (int k = 0; k < n_grids && k < 2; ++k) { int rc = bounds_launch(ctx, *grids[k], hp[k]); if (rc) return rc; }
        MLH_HIP(ctx, hipStreamSynchronize(st));
        for (int k = 0; k < n_grids && k < 2; ++k) { int rc = bounds_finish(ctx, *grids[k], hp[k], grids[k]->min_match_sq_dis); if (rc) return rc; }
    }
    for (int k = 0; k < n_grids && k < 2; ++k) {
        MapGrid &g = *grids[k];
        if (g.want_occ) {               // occupancy partials: one per scanning workgroup
            g.occ_parts = int((g.ncell + SCAN_CHUNK) / SCAN_CHUNK);
            MLH_HIP(ctx, g.occ.ensure(sizeof(long long) * size_t(2 + 2 * g.occ_parts)));
        }
        G.j[nj++] = make_job(g);
    }
    if (nj == 0) return MLH_OK;
    const int nb_scan = G.j[0].nb_scan + G.j[1].nb_scan, nb_pts = G.j[0].nb_pts + G.j[1].nb_pts;
    prof_begin(ctx, MLH_K_GRID_BUILD);
    if (G.j[0].need_zero || G.j[1].need_zero) MLH_LAUNCH(zero_cells_kernel, dim3(nb_scan), dim3(256), 0, st, G);
    if (pack) { G.pub_stage = 1; MLH_LAUNCH(pack_count_kernel, dim3(nb_pts), dim3(256), 0, st, G, *pack); }
    else MLH_LAUNCH(cell_count_kernel, dim3(nb_pts), dim3(256), 0, st, G);
    MLH_LAUNCH(scan_local_kernel, dim3(nb_scan), dim3(256), 0, st, G);
    if (G.j[0].sums_scanned || G.j[1].sums_scanned) MLH_LAUNCH(scan_sums_kernel, dim3(nj), dim3(256), 0, st, G);
    MLH_LAUNCH(scan_add_kernel, dim3(nb_scan), dim3(256), 0, st, G);
    MLH_LAUNCH(scatter_kernel, dim3(nb_pts), dim3(256), 0, st, G);
    prof_end(ctx, MLH_K_GRID_BUILD);
    MLH_HIP(ctx, hipGetLastError());
    for (int k = 0; k < nj; ++k) grids[k]->built = true;
    return MLH_OK;
}

// ---- a frame's local map arrives as a cloud of caller records: pack to float4 {x,y,z,index} and, in the same pass, check that every
// point lies inside the grid box already set up for this kind (flag word, OR-ed). One launch for both maps.
__global__ __launch_bounds__(256) void pack_check_kernel(PackJobs G)
{
    const int job = blockIdx.x >= G.j[0].nb ? 1 : 0;
    const PackJob &J = G.j[job];
    const int i = (job ? blockIdx.x - G.j[0].nb : blockIdx.x) * 256 + threadIdx.x;
    bool bad = false;
    if (i < J.n) {
        const float *rec = reinterpret_cast<const float *>(J.src + size_t(i) * J.stride);
        const float x = rec[0], y = rec[1], z = rec[2];
        J.out[i] = make_float4(x, y, z, __int_as_float(i));
        bad = !(x >= J.lo[0] && x < J.hi[0] && y >= J.lo[1] && y < J.hi[1] && z >= J.lo[2] && z < J.hi[2]);   // NaN -> bad
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(G.oob, 1 << job);
}

// the fit flags go to the host as soon as the clouds are packed (the index build that follows does not change them)
__global__ void publish_flag_kernel(int *oob, HostPublish *h, unsigned long long seq)
{
    if (threadIdx.x == 0) {
        h->done = *oob;          // the record's `done` slot carries the flag word here
        *oob = 0;
        __hip_atomic_store(&h->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// Stages 1 or 2 clouds (device pointers to strided records) and builds their indices. When a kind's grid geometry from an earlier call
// still applies (same acceptance radius), nothing waits for the host until the end: pack + fit check, index build, then ONE pinned-memory
// hand-shake that returns the fit flags; only a cloud that has outgrown its box takes the bounds round trip (and gets a new box).
int map_stage_and_build(mlh_ctx *ctx, int n_maps, const int *kinds, const unsigned char *const *src, const int *n, int stride, const float *sq_dis,
                        HostPublish *pub, unsigned long long seq)
{
    hipStream_t st = ctx->stream;
    PackJobs G;
    std::memset(&G, 0, sizeof(G));
    MLH_HIP(ctx, ctx->oob_flag.ensure(sizeof(int)));
    if (!ctx->oob_init) { MLH_HIP(ctx, hipMemsetAsync(ctx->oob_flag.p, 0, sizeof(int), st)); ctx->oob_init = true; }
    G.oob = ctx->oob_flag.as<int>();
    MapGrid *grids[2];
    int need_bounds = 0;
    for (int k = 0; k < n_maps; ++k) {
        MapGrid &g = ctx->map[kinds[k]];
        grids[k] = &g;
        g.built = false;
        MLH_HIP(ctx, g.raw.ensure(sizeof(float4) * size_t(n[k])));
        g.want_occ = true;
        const bool reuse = g.geom_valid && g.geom_sq_dis == sq_dis[k];
        if (!reuse) need_bounds |= 1 << k;
        g.n = n[k];
        g.min_match_sq_dis = sq_dis[k];
        PackJob &J = G.j[k];
        J.src = src[k]; J.out = g.raw.as<float4>(); J.n = n[k]; J.stride = stride; J.nb = (n[k] + 255) / 256;
        // a box that accepts everything when there is no geometry to check against (the bounds pass follows anyway)
        J.lo[0] = reuse ? g.ox : -INFINITY; J.lo[1] = reuse ? g.oy : -INFINITY; J.lo[2] = reuse ? g.oz : -INFINITY;
        J.hi[0] = reuse ? g.ox + float(g.nx) * g.h : INFINITY; J.hi[1] = reuse ? g.oy + float(g.ny) * g.h : INFINITY;
        J.hi[2] = reuse ? g.oz + float(g.nz) * g.h : INFINITY;
    }
    // every staged kind keeps its geometry (a mapper between keyframes): pack + fit check ride in the index build's first launch (pack_count_kernel);
    // MLH_GRID_PACK_LAUNCH=1 keeps the launch of their own (A/B runs)
    static const bool pack_apart = std::getenv("MLH_GRID_PACK_LAUNCH") && std::atoi(std::getenv("MLH_GRID_PACK_LAUNCH")) != 0;
    const bool pack_in_build = need_bounds == 0 && !pack_apart;
    if (!pack_in_build) MLH_LAUNCH(pack_check_kernel, dim3(G.j[0].nb + G.j[1].nb), dim3(256), 0, st, G);
    // the fit flags are final once the clouds are packed: they leave for the host NOW, and the optimistic index build of the kinds whose
    // geometry is reused is enqueued behind them -- the host learns the outcome (and can go on enqueueing the frame's solver launches)
    // while the GPU is still building the index, instead of the GPU idling through the host's reaction time after the build. When such a
    // build follows, its first launch carries the publication (one thread of its first workgroup); otherwise a launch of its own does.
    const bool build_follows = need_bounds != ((1 << n_maps) - 1);
    if (!build_follows) MLH_LAUNCH(publish_flag_kernel, dim3(1), dim3(64), 0, st, G.oob, pub, seq);
    MLH_HIP(ctx, hipGetLastError());
    if (!ctx->h_occ) {
        MLH_HIP(ctx, hipHostMalloc(&ctx->h_occ, sizeof(long long) * 4, hipHostMallocDefault));
        std::memset(ctx->h_occ, 0, sizeof(long long) * 4);
    }
    long long *h_occ = static_cast<long long *>(ctx->h_occ);
    for (int k = 0; k < n_maps; ++k) grids[k]->occ_host = h_occ + 2 * kinds[k];
    // occupancy statistics (they only steer the lanes-per-query choice): what the previous staging call's builds left in the pinned mirror
    for (int k = 0; k < n_maps; ++k) if (!(need_bounds & (1 << k)) && h_occ[2 * kinds[k]] > 0) {
        grids[k]->occupied = int(h_occ[2 * kinds[k]]); grids[k]->pop_sq = h_occ[2 * kinds[k] + 1];
    }
    if (build_follows) {
        MapGrid *fast[2];
        int nf = 0;
        for (int k = 0; k < n_maps; ++k) if (!(need_bounds & (1 << k))) {
            MapGrid &g = *grids[k];
            MLH_HIP(ctx, g.sorted.ensure(sizeof(float4) * size_t(g.n)));
            MLH_HIP(ctx, g.cell_id.ensure(sizeof(int) * size_t(g.n)));
            fast[nf++] = &g;
        }
        int rc = grid_build_grids_packed(ctx, fast, nf, false, G.oob, pub, seq, pack_in_build ? &G : nullptr);
        if (rc) return rc;
    }
    // spin on the pinned record (pack + fit check have completed when the sequence number arrives; the builds may still be running)
    {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(&pub->seq, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                MLH_HIP(ctx, hipStreamSynchronize(st));
                if (__atomic_load_n(&pub->seq, __ATOMIC_ACQUIRE) != seq) return fail(ctx, MLH_ERR_HIP, "map staging did not complete");
                break;
            }
            host_wait_relax(spins);
        }
    }
    const int oob = int(pub->done);
    for (int k = 0; k < n_maps; ++k) if (oob & (1 << k)) need_bounds |= 1 << k;
    if (need_bounds) {
        MapGrid *slow[2];
        int ns = 0;
        for (int k = 0; k < n_maps; ++k) if (need_bounds & (1 << k)) { grids[k]->built = false; grids[k]->geom_valid = false; slow[ns++] = grids[k]; }
        int rc = grid_build_grids(ctx, slow, ns, true);
        if (rc) return rc;
        long long hocc[2][2] = {{0, 0}, {0, 0}};
        for (int k = 0; k < ns; ++k) MLH_HIP(ctx, hipMemcpyAsync(hocc[k], slow[k]->occ.p, 2 * sizeof(long long), hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        for (int k = 0; k < ns; ++k) { slow[k]->occupied = int(hocc[k][0]); slow[k]->pop_sq = hocc[k][1]; }
    }
    return MLH_OK;
}

// (re)builds the index of the maps selected by kind_mask
int grid_build(mlh_ctx *ctx, int kind_mask, bool recompute_bounds)
{
    MapGrid *g[2];
    int n = 0;
    for (int k = 0; k < 2; ++k) {
        if (!(kind_mask & (1 << k))) continue;
        if (ctx->map[k].n <= 0 || !ctx->map[k].raw.p) return fail(ctx, MLH_ERR_STATE, "map_set has not been called for this kind");
        g[n++] = &ctx->map[k];
    }
    return grid_build_grids(ctx, g, n, recompute_bounds);
}

}  // namespace mlh
