// VoxelGridCovarianceMLOAM<PointT>::applyFilter on gfx950 -- the covariance-aware voxel thinning that produces the local map
// and the thinned scan features (mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp:68-457; call sites
// lidar_mapper_keyframe.cpp:343-347, 359-368).
//
// The reference sorts (voxel index, point index) pairs with std::sort and walks the runs. Here the voxel index (same f32
// arithmetic: floor(x * inv_leaf) - min_b, x fastest) drives a counting sort over the dense voxel grid -- count, in-place
// exclusive scan, scatter of POINT INDICES -- so no comparison sort is needed; the first sorted position of every occupied
// voxel is its leader, a prefix sum over the leader flags gives the output slot (= ascending voxel index, the reference's
// output order), and the leader sorts its few member indices and accumulates them in index order (deterministic; the
// reference's order inside a voxel is whatever its unstable sort left, so sums agree up to f32 rounding).
//   cov branch (:296-333)   w = thr - tr; |tr| >= thr dropped; mu = sum w p / sum w; cov = sum w^2 cov_i / (sum w)^2; intensity of the
//                           heaviest member (first wins); trace recomputed from the diagonal
//   plain branch (:392-420) xyz mean over the members, intensity of the last member
// Streaming kernels: ~ (stride + 8) B/point per pass + 8 B/voxel for the scan.
#include "ctx.hpp"
#include <cfloat>
#include <cmath>

namespace mlh {

constexpr int VS_ITEMS = 8, VS_CHUNK = 256 * VS_ITEMS;

__device__ __forceinline__ int vblock_scan(int v, int *lds, int &total)
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

// generic in-place exclusive scan of int data[0..n): local / sums / add
__global__ __launch_bounds__(256) void vscan_local_kernel(int *__restrict__ data, long long n, int *__restrict__ sums)
{
    __shared__ int lds[4];
    const long long base = (long long)blockIdx.x * VS_CHUNK + threadIdx.x * VS_ITEMS;
    int v[VS_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) { v[k] = (base + k < n) ? data[base + k] : 0; s += v[k]; }
    int total;
    int ex = vblock_scan(s, lds, total);
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) { if (base + k < n) data[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}
__global__ __launch_bounds__(256) void vscan_sums_kernel(int *__restrict__ sums, int nb, int *__restrict__ grand_total)
{
    __shared__ int lds[4];
    int carry = 0;
    for (int start = 0; start < nb; start += 256) {
        const int i = start + threadIdx.x;
        const int v = (i < nb) ? sums[i] : 0;
        int total;
        const int ex = vblock_scan(v, lds, total);
        if (i < nb) sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0 && grand_total) *grand_total = carry;
}
__global__ __launch_bounds__(256) void vscan_add_kernel(int *__restrict__ data, long long n, const int *__restrict__ sums)
{
    const long long base = (long long)blockIdx.x * VS_CHUNK + threadIdx.x * VS_ITEMS;
    const int add = sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) if (base + k < n) data[base + k] += add;
}

int device_exclusive_scan(mlh_ctx *ctx, int *data, long long n, DevBuf &sums, int *grand_total)
{
    const int nb = int((n + VS_CHUNK - 1) / VS_CHUNK);
    MLH_HIP(ctx, sums.ensure(sizeof(int) * size_t(nb + 1)));
    hipLaunchKernelGGL(vscan_local_kernel, dim3(nb), dim3(256), 0, ctx->stream, data, n, sums.as<int>());
    hipLaunchKernelGGL(vscan_sums_kernel, dim3(1), dim3(256), 0, ctx->stream, sums.as<int>(), nb, grand_total);
    hipLaunchKernelGGL(vscan_add_kernel, dim3(nb), dim3(256), 0, ctx->stream, data, n, sums.as<int>());
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

struct VoxArgs {
    const unsigned char *src;
    int stride, n, intensity_off, cov_off, trace_off;
    float inv_leaf;
    int min_b[3], mul1, mul2;
    float trace_thr;
    int *vox_of;        // n: voxel index per point
    int *cell;          // ncell + 1: counts -> starts (shifted by one as in grid.hip)
    int *sorted_idx;    // n
    int *leader;        // n: leader flags -> output slots
    unsigned char *out; // n_out records (same layout as the input)
};

__device__ __forceinline__ const float *vrec(const VoxArgs &A, int i) { return reinterpret_cast<const float *>(A.src + size_t(i) * A.stride); }

__global__ __launch_bounds__(256) void vox_count_kernel(VoxArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float *p = vrec(A, i);
    const int ijk0 = int(floorf(p[0] * A.inv_leaf) - float(A.min_b[0]));
    const int ijk1 = int(floorf(p[1] * A.inv_leaf) - float(A.min_b[1]));
    const int ijk2 = int(floorf(p[2] * A.inv_leaf) - float(A.min_b[2]));
    const int v = ijk0 + ijk1 * A.mul1 + ijk2 * A.mul2;
    A.vox_of[i] = v;
    atomicAdd(&A.cell[v + 1], 1);
}

__global__ __launch_bounds__(256) void vox_scatter_kernel(VoxArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const int pos = atomicAdd(&A.cell[A.vox_of[i] + 1], 1);
    A.sorted_idx[pos] = i;
}

// after the scatter cell[v] = start[v], cell[v+1] = end[v]
__global__ __launch_bounds__(256) void vox_leader_kernel(VoxArgs A)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= A.n) return;
    const int v = A.vox_of[A.sorted_idx[p]];
    A.leader[p] = (p == A.cell[v]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void vox_aggregate_kernel(VoxArgs A, const int *__restrict__ slot /* exclusive scan of the leader flags */,
                                                            const int *__restrict__ flag_total)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= A.n) return;
    const int v = A.vox_of[A.sorted_idx[p]];
    const int b = A.cell[v];
    if (p != b) return;
    const int e = A.cell[v + 1];
    (void)flag_total;
    // member indices in ascending order: selection by repeated minimum (runs are short)
    float mu[3] = {0.f, 0.f, 0.f}, ity = 0.f, cov[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, weight_total = 0.f, w_max = 0.f;
    int cnt = 0;
    int last = -1;
    for (int k = b; k < e; ++k) {
        int cur = 0x7fffffff;
        for (int u = b; u < e; ++u) { const int id = A.sorted_idx[u]; if (id > last && id < cur) cur = id; }
        last = cur;
        const float *q = vrec(A, cur);
        const float inten = A.intensity_off >= 0 ? *reinterpret_cast<const float *>(A.src + size_t(cur) * A.stride + A.intensity_off) : 0.f;
        if (A.cov_off >= 0) {
            const float *c = reinterpret_cast<const float *>(A.src + size_t(cur) * A.stride + A.cov_off);
            const float tr = c[0] + c[3] + c[5];
            if (fabsf(tr) >= A.trace_thr) continue;
            const float w = A.trace_thr - tr;
            mu[0] += w * q[0]; mu[1] += w * q[1]; mu[2] += w * q[2];
            ity = w > w_max ? inten : ity;
            w_max = w > w_max ? w : w_max;
            const float w2 = w * w;
#pragma unroll
            for (int j = 0; j < 6; ++j) cov[j] += w2 * c[j];
            weight_total += w;
        } else {
            mu[0] += q[0]; mu[1] += q[1]; mu[2] += q[2];
            ity = inten;                       // the last member's intensity
            ++cnt;
        }
    }
    unsigned char *o = A.out + size_t(slot[p]) * A.stride;
    for (int j = 0; j < A.stride / 4; ++j) reinterpret_cast<float *>(o)[j] = 0.f;
    float *ox = reinterpret_cast<float *>(o);
    if (A.cov_off >= 0) {
        if (weight_total == 0.f) weight_total = 1.0f;
        ox[0] = mu[0] / weight_total; ox[1] = mu[1] / weight_total; ox[2] = mu[2] / weight_total;
        const float wt2 = weight_total * weight_total;
        float *oc = reinterpret_cast<float *>(o + A.cov_off);
#pragma unroll
        for (int j = 0; j < 6; ++j) oc[j] = cov[j] / wt2;
        if (A.trace_off >= 0) *reinterpret_cast<float *>(o + A.trace_off) = oc[0] + oc[3] + oc[5];
    } else {
        const float fc = float(cnt > 0 ? cnt : 1);
        ox[0] = mu[0] / fc; ox[1] = mu[1] / fc; ox[2] = mu[2] / fc;
    }
    if (A.stride >= 16 && A.intensity_off != 12 && A.cov_off != 12 && A.trace_off != 12) ox[3] = 1.0f;   // PCL_ADD_POINT4D padding
    if (A.intensity_off >= 0) *reinterpret_cast<float *>(o + A.intensity_off) = ity;
}

__global__ void vbounds_init_kernel(float *b)
{
    if (threadIdx.x < 3) b[threadIdx.x] = FLT_MAX;
    else if (threadIdx.x < 6) b[threadIdx.x] = -FLT_MAX;
}
__device__ __forceinline__ void atomic_minf(float *a, float v) { if (v >= 0.f) atomicMin(reinterpret_cast<int *>(a), __float_as_int(v)); else atomicMax(reinterpret_cast<unsigned *>(a), __float_as_uint(v)); }
__device__ __forceinline__ void atomic_maxf(float *a, float v) { if (v >= 0.f) atomicMax(reinterpret_cast<int *>(a), __float_as_int(v)); else atomicMin(reinterpret_cast<unsigned *>(a), __float_as_uint(v)); }
__global__ __launch_bounds__(256) void vbounds_kernel(const unsigned char *src, int stride, int n, float *b)
{
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float *p = reinterpret_cast<const float *>(src + size_t(i) * stride);
#pragma unroll
        for (int d = 0; d < 3; ++d) { mn[d] = fminf(mn[d], p[d]); mx[d] = fmaxf(mx[d], p[d]); }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { mn[d] = fminf(mn[d], __shfl_xor(mn[d], off)); mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], off)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 3; ++d) { atomic_minf(&b[d], mn[d]); atomic_maxf(&b[3 + d], mx[d]); }
    }
}

int voxel_filter_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int cov_off, int trace_off, float leaf,
                     float trace_thr, void *out_host, int *n_out, int mem)
{
    if (!points || n <= 0 || stride < 12 || (stride & 3) || !(leaf > 0.f) || !n_out) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    // out_host == nullptr: the thinned records stay in ctx->vox.out (device) for the caller's next kernel
    hipStream_t st = ctx->stream;
    VoxBuf &V = ctx->vox;
    const unsigned char *src = static_cast<const unsigned char *>(points);
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, V.in.ensure(size_t(n) * stride));
        MLH_HIP(ctx, hipMemcpyAsync(V.in.p, points, size_t(n) * stride, hipMemcpyHostToDevice, st));
        src = V.in.as<unsigned char>();
    }
    // bounds -> min_b / div_b (getMinMax3D + the floor arithmetic of applyFilter :84-116)
    MLH_HIP(ctx, V.bounds.ensure(sizeof(float) * 8));
    hipLaunchKernelGGL(vbounds_init_kernel, dim3(1), dim3(64), 0, st, V.bounds.as<float>());
    hipLaunchKernelGGL(vbounds_kernel, dim3(std::min((n + 255) / 256, 2048)), dim3(256), 0, st, src, stride, n, V.bounds.as<float>());
    float hb[6];
    MLH_HIP(ctx, hipMemcpyAsync(hb, V.bounds.p, sizeof(hb), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    const float inv = 1.0f / leaf;
    long long ext[3];
    int min_b[3], div_b[3];
    for (int d = 0; d < 3; ++d) {
        if (!std::isfinite(hb[d]) || !std::isfinite(hb[3 + d])) return fail(ctx, MLH_ERR_INVALID, "non-finite coordinates");
        ext[d] = (long long)((hb[3 + d] - hb[d]) * inv) + 1;
        min_b[d] = int(std::floor(hb[d] * inv));
        div_b[d] = int(std::floor(hb[3 + d] * inv)) - min_b[d] + 1;
    }
    // stepwise, so that the test itself cannot overflow 64 bits on absurd extents
    const bool too_many = ext[0] > 2147483647ll || ext[1] > 2147483647ll || ext[2] > 2147483647ll || ext[0] * ext[1] > 2147483647ll ||
                          ext[0] * ext[1] * ext[2] > 2147483647ll;
    if (too_many) {
        // "Leaf size is too small for the input dataset": the reference returns the input cloud unchanged
        if (!out_host) { MLH_HIP(ctx, V.out.ensure(size_t(n) * stride)); MLH_HIP(ctx, hipMemcpyAsync(V.out.p, src, size_t(n) * stride, hipMemcpyDeviceToDevice, st)); }
        else MLH_HIP(ctx, hipMemcpyAsync(out_host, src, size_t(n) * stride, mem == MLH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        *n_out = n;
        return MLH_OK;
    }
    const long long ncell = (long long)div_b[0] * div_b[1] * div_b[2];
    MLH_HIP(ctx, V.cell.ensure(sizeof(int) * size_t(ncell + 2)));
    MLH_HIP(ctx, V.vox_of.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.sorted_idx.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.leader.ensure(sizeof(int) * size_t(n + 1)));
    MLH_HIP(ctx, V.out.ensure(size_t(n) * stride));
    MLH_HIP(ctx, V.total.ensure(sizeof(int) * 2));
    VoxArgs A;
    A.src = src; A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.cov_off = cov_off; A.trace_off = trace_off;
    A.inv_leaf = inv; A.min_b[0] = min_b[0]; A.min_b[1] = min_b[1]; A.min_b[2] = min_b[2];
    A.mul1 = div_b[0]; A.mul2 = div_b[0] * div_b[1];
    A.trace_thr = trace_thr; A.vox_of = V.vox_of.as<int>(); A.cell = V.cell.as<int>(); A.sorted_idx = V.sorted_idx.as<int>();
    A.leader = V.leader.as<int>(); A.out = V.out.as<unsigned char>();
    const int nbp = (n + 255) / 256;
    MLH_HIP(ctx, hipMemsetAsync(V.cell.p, 0, sizeof(int) * size_t(ncell + 2), st));
    hipLaunchKernelGGL(vox_count_kernel, dim3(nbp), dim3(256), 0, st, A);
    int rc = device_exclusive_scan(ctx, A.cell + 1, ncell, V.sums, nullptr);   // cell[v+1] <- start[v]
    if (rc) return rc;
    hipLaunchKernelGGL(vox_scatter_kernel, dim3(nbp), dim3(256), 0, st, A);    // cell[v+1] <- start[v+1]
    hipLaunchKernelGGL(vox_leader_kernel, dim3(nbp), dim3(256), 0, st, A);
    rc = device_exclusive_scan(ctx, A.leader, n, V.sums, V.total.as<int>());   // leader flags -> output slots, total = occupied voxels
    if (rc) return rc;
    hipLaunchKernelGGL(vox_aggregate_kernel, dim3(nbp), dim3(256), 0, st, A, (const int *)A.leader, (const int *)V.total.as<int>());
    MLH_HIP(ctx, hipGetLastError());
    int total = 0;
    MLH_HIP(ctx, hipMemcpyAsync(&total, V.total.p, sizeof(int), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    *n_out = total;
    if (total > 0 && out_host) {
        MLH_HIP(ctx, hipMemcpyAsync(out_host, V.out.p, size_t(total) * stride, mem == MLH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
    }
    return MLH_OK;
}

}  // namespace mlh
