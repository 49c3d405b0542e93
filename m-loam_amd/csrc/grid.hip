// Local-map index build for gfx950: a dense cell grid over the map cloud, points counting-sorted by cell.
// Replaces the FLANN kd-tree build the reference performs every frame
// (pcl::KdTreeFLANN::setInputCloud, estimator/src/lidarMapper/lidar_mapper_keyframe.cpp:433-434).
//
// Layout in HBM: raw[n] and sorted[n] as float4 {x, y, z, original index}; cell_start[ncell+1] (int32, exclusive prefix),
// cells ordered x fastest so the 3 x-adjacent cells of a query form ONE contiguous run of `sorted` (9 runs per query).
// Kernels (all streaming, HBM-bound): bounds (16 B/pt read), count (16 B read + 4 B write + 1 atomic), 3-phase exclusive
// scan over the cells (8 B/cell), scatter (16 B + 4 B read, 16 B write, 1 atomic).
#include "ctx.hpp"
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

__global__ void bounds_init_kernel(int *b)
{
    if (threadIdx.x < 3) b[threadIdx.x] = INT_MAX;
    else if (threadIdx.x < 6) b[threadIdx.x] = INT_MIN;
}

__global__ __launch_bounds__(256) void bounds_kernel(const float4 *__restrict__ pts, int n, int *__restrict__ b)
{
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
        for (int d = 0; d < 3; ++d) { atomicMin(&b[d], mn[d]); atomicMax(&b[3 + d], mx[d]); }
    }
}

__device__ __forceinline__ int cell_coord(float v, float o, float inv_h, int n)
{
    float f = floorf((v - o) * inv_h);
    f = fminf(fmaxf(f, 0.f), float(n - 1));
    return int(f);
}

__global__ __launch_bounds__(256) void cell_count_kernel(const float4 *__restrict__ pts, int n, float ox, float oy, float oz,
                                                         float inv_h, int nx, int ny, int nz,
                                                         int *__restrict__ cell_id, int *__restrict__ cell_cnt)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 p = pts[i];
        int cx = cell_coord(p.x, ox, inv_h, nx), cy = cell_coord(p.y, oy, inv_h, ny), cz = cell_coord(p.z, oz, inv_h, nz);
        int c = (cz * ny + cy) * nx + cx;
        cell_id[i] = c;
        atomicAdd(&cell_cnt[c], 1);
    }
}

// ---- exclusive scan, 2048 items per block (256 threads x 8)
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_CHUNK = 256 * SCAN_ITEMS;

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

__global__ __launch_bounds__(256) void scan_local_kernel(const int *__restrict__ in, long long n, int *__restrict__ out,
                                                         int *__restrict__ block_sums)
{
    __shared__ int lds[4];
    long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
    int v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { v[k] = (base + k < n) ? in[base + k] : 0; s += v[k]; }
    int total;
    int ex = block_exclusive_scan_256(s, lds, total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void scan_sums_kernel(int *__restrict__ block_sums, int nb)
{
    __shared__ int lds[4];
    int carry = 0;
    for (int start = 0; start < nb; start += 256) {
        int i = start + threadIdx.x;
        int v = (i < nb) ? block_sums[i] : 0;
        int total;
        int ex = block_exclusive_scan_256(v, lds, total);
        if (i < nb) block_sums[i] = carry + ex;
        carry += total;
    }
}

__global__ __launch_bounds__(256) void scan_add_kernel(int *__restrict__ out, long long n, const int *__restrict__ block_sums,
                                                       int n_points)
{
    long long base = (long long)blockIdx.x * SCAN_CHUNK + (long long)threadIdx.x * SCAN_ITEMS;
    int add = block_sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) out[base + k] += add;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[n] = n_points;
}

__global__ __launch_bounds__(256) void scatter_kernel(const float4 *__restrict__ raw, int n, const int *__restrict__ cell_id,
                                                      const int *__restrict__ cell_start, int *__restrict__ cursor,
                                                      float4 *__restrict__ sorted)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int c = cell_id[i];
        int pos = cell_start[c] + atomicAdd(&cursor[c], 1);
        sorted[pos] = raw[i];
    }
}

int grid_build(mlh_ctx *ctx, MapGrid &g, float min_match_sq_dis, bool recompute_bounds)
{
    hipStream_t st = ctx->stream;
    const int n = g.n;
    if (n <= 0) return fail(ctx, MLH_ERR_INVALID, "map cloud is empty");
    const int grid_pts = std::min((n + 255) / 256, 2048);
    if (recompute_bounds) {
        MLH_HIP(ctx, g.bounds.ensure(6 * sizeof(int)));
        hipLaunchKernelGGL(bounds_init_kernel, dim3(1), dim3(64), 0, st, g.bounds.as<int>());
        hipLaunchKernelGGL(bounds_kernel, dim3(grid_pts), dim3(256), 0, st, g.raw.as<float4>(), n, g.bounds.as<int>());
        int hb[6];
        MLH_HIP(ctx, hipMemcpyAsync(hb, g.bounds.p, sizeof(hb), hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        float mn[3], mx[3];
        for (int d = 0; d < 3; ++d) { mn[d] = key_to_float(hb[d]); mx[d] = key_to_float(hb[3 + d]); }
        for (int d = 0; d < 3; ++d)
            if (!std::isfinite(mn[d]) || !std::isfinite(mx[d])) return fail(ctx, MLH_ERR_INVALID, "map cloud has non-finite coordinates");
        g.h = std::sqrt(min_match_sq_dis) * 1.001f;
        g.inv_h = 1.0f / g.h;
        g.ox = mn[0]; g.oy = mn[1]; g.oz = mn[2];
        g.nx = int(std::floor((mx[0] - g.ox) * g.inv_h)) + 1;
        g.ny = int(std::floor((mx[1] - g.oy) * g.inv_h)) + 1;
        g.nz = int(std::floor((mx[2] - g.oz) * g.inv_h)) + 1;
        g.ncell = (long long)g.nx * g.ny * g.nz;
        if (g.ncell >= (1ll << 31) - SCAN_CHUNK) return fail(ctx, MLH_ERR_UNSUPPORTED, "map extent needs more than 2^31 cells");
    }
    const long long ncell = g.ncell;
    const int nb = int((ncell + SCAN_CHUNK - 1) / SCAN_CHUNK);
    MLH_HIP(ctx, g.sorted.ensure(sizeof(float4) * size_t(n)));
    MLH_HIP(ctx, g.cell_id.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, g.cell_start.ensure(sizeof(int) * size_t(ncell + 1)));
    MLH_HIP(ctx, g.cell_fill.ensure(sizeof(int) * size_t(ncell)));
    MLH_HIP(ctx, g.block_sums.ensure(sizeof(int) * size_t(nb + 1)));

    prof_begin(ctx, MLH_K_GRID_BUILD);
    MLH_HIP(ctx, hipMemsetAsync(g.cell_fill.p, 0, sizeof(int) * size_t(ncell), st));
    hipLaunchKernelGGL(cell_count_kernel, dim3(grid_pts), dim3(256), 0, st, g.raw.as<float4>(), n, g.ox, g.oy, g.oz, g.inv_h,
                       g.nx, g.ny, g.nz, g.cell_id.as<int>(), g.cell_fill.as<int>());
    hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(256), 0, st, g.cell_fill.as<int>(), ncell, g.cell_start.as<int>(),
                       g.block_sums.as<int>());
    hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(256), 0, st, g.block_sums.as<int>(), nb);
    hipLaunchKernelGGL(scan_add_kernel, dim3(nb), dim3(256), 0, st, g.cell_start.as<int>(), ncell, g.block_sums.as<int>(), n);
    MLH_HIP(ctx, hipMemsetAsync(g.cell_fill.p, 0, sizeof(int) * size_t(ncell), st));
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_pts), dim3(256), 0, st, g.raw.as<float4>(), n, g.cell_id.as<int>(),
                       g.cell_start.as<int>(), g.cell_fill.as<int>(), g.sorted.as<float4>());
    prof_end(ctx, MLH_K_GRID_BUILD);
    MLH_HIP(ctx, hipGetLastError());
    g.built = true;
    return MLH_OK;
}

}  // namespace mlh
