// Voxel thinning and uncertainty propagation kernels (gfx950).
//
// ring_voxel_kernel      the per-ring pcl::VoxelGrid<PointXYZI>(0.2 m) that closes FeatureExtract::extractCloud
//                        (estimator/src/featureExtract/feature_extract.cpp:266-271; PCL 1.8.0 filters/impl/voxel_grid.hpp):
//                        one workgroup per ring, the ring's less-flat points (label <= 0, cpp:258-264) are keyed with PCL's
//                        voxel index (floor(x * inv_leaf) - min_b, x fastest), sorted in LDS on 64-bit (voxel, position) keys
//                        with a bitonic network, and every voxel's members are averaged (CentroidPoint: f32 sums / count)
//                        in position order. Output order = ring asc, voxel index asc, as the reference concatenates them.
// point_uncertainty_kernel   evalPointUncertainty (estimator/src/lidarMapper/associate_uct.hpp:196-215) as used by
//                        downsampleCurrentScan (lidar_mapper_keyframe.cpp:375-418): per point, Sigma_p = [G diag(Sigma_ext,
//                        Sigma_meas) G^T]_3x3 with G = [(T p)^odot | T D], f64, stored to the f32 cov_vec of PointXYZIWithCov;
//                        points whose trace exceeds TRACE_THRESHOLD_MAPPING are dropped (order-preserving compaction).
#include "ctx.hpp"
#include "dev_math.hpp"
#include <cfloat>

namespace mlh {

// ---------------------------------------------------------------- per-ring voxel grid
struct RingVoxelArgs {
    const float4 *pts;          // scan points
    const int *list3;           // less-flat positions (all rings, ring-major)
    const int *ring_counts;     // [ring][4]
    const int *ring_offsets;    // [ring][4]
    float4 *stage;              // staged centroids, at the ring's offset in the less-flat list
    int *ring_vox;              // voxels per ring
    float leaf;
    int sort_p;                 // power of two >= the longest less-flat run
};

__device__ __forceinline__ float block_reduce_minmax(float v, bool is_min, float *lds)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        float o = __shfl_xor(v, off);
        v = is_min ? fminf(v, o) : fmaxf(v, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = lds[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) r = is_min ? fminf(r, lds[w]) : fmaxf(r, lds[w]);
    return r;
}

__global__ __launch_bounds__(256) void ring_voxel_kernel(RingVoxelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float s_red[4];
    __shared__ int s_scan[4];
    __shared__ int s_total;
    const int ring = blockIdx.x;
    const int n = A.ring_counts[ring * 4 + 3];
    const int off = A.ring_offsets[ring * 4 + 3];
    if (n <= 0) { if (threadIdx.x == 0) A.ring_vox[ring] = 0; return; }
    const int P = A.sort_p;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
    const float inv = 1.0f / A.leaf;

    // bounds (getMinMax3D) -> min_b, div_b (voxel_grid.hpp)
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int t = threadIdx.x; t < n; t += 256) {
        const float4 p = A.pts[A.list3[off + t]];
        mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
        mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
        mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    }
    int min_b[3], div_b[3];
    long long cells = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float lo = block_reduce_minmax(mn[d], true, s_red);
        const float hi = block_reduce_minmax(mx[d], false, s_red);
        min_b[d] = int(floorf(lo * inv));
        div_b[d] = int(floorf(hi * inv)) - min_b[d] + 1;
        cells *= (long long)((hi - lo) * inv) + 1;
    }
    if (cells > 2147483647ll) {   // "Leaf size is too small for the input dataset": PCL returns the input unchanged
        for (int t = threadIdx.x; t < n; t += 256) A.stage[off + t] = A.pts[A.list3[off + t]];
        if (threadIdx.x == 0) A.ring_vox[ring] = n;
        return;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    // keys: (voxel index << 32) | position in the ring's less-flat run; padding = all ones
    for (int t = threadIdx.x; t < P; t += 256) {
        unsigned long long key = ~0ull;
        if (t < n) {
            const float4 p = A.pts[A.list3[off + t]];
            const int ijk0 = int(floorf(p.x * inv) - float(min_b[0]));
            const int ijk1 = int(floorf(p.y * inv) - float(min_b[1]));
            const int ijk2 = int(floorf(p.z * inv) - float(min_b[2]));
            const unsigned idx = unsigned(ijk0 + ijk1 * mul1 + ijk2 * mul2);
            key = ((unsigned long long)idx << 32) | unsigned(t);
        }
        keys[t] = key;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int ixj = i | j;
                const bool up = ((i & k) == 0) || (k == P);
                unsigned long long a = keys[i], b = keys[ixj];
                if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
            }
            __syncthreads();
        }
    }
    // every voxel start computes its centroid; output slot = number of voxel starts before it
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 256) {
        const int t = c0 + threadIdx.x;
        bool start = false;
        unsigned vox = 0;
        if (t < n) {
            vox = unsigned(keys[t] >> 32);
            start = (t == 0) || (unsigned(keys[t - 1] >> 32) != vox);
        }
        const unsigned long long m = __ballot(start);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_scan[wave] = __popcll(m);
        __syncthreads();
        int wbase = s_total;
        for (int w = 0; w < wave; ++w) wbase += s_scan[w];
        if (start) {
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            int cnt = 0;
            for (int u = t; u < n && unsigned(keys[u] >> 32) == vox; ++u) {
                const float4 p = A.pts[A.list3[off + int(unsigned(keys[u]))]];
                sx += p.x; sy += p.y; sz += p.z; si += p.w;
                ++cnt;
            }
            const float fc = float(cnt);
            A.stage[off + wbase + before] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_total += s_scan[0] + s_scan[1] + s_scan[2] + s_scan[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.ring_vox[ring] = s_total;
}

// exclusive scan of the per-ring voxel counts (single workgroup) + total
__global__ void ring_vox_offsets_kernel(const int *__restrict__ ring_vox, int n_rings, int *__restrict__ vox_off, int *__restrict__ total)
{
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int r = 0; r < n_rings; ++r) { vox_off[r] = acc; acc += ring_vox[r]; }
        *total = acc;
    }
}

__global__ __launch_bounds__(256) void ring_vox_compact_kernel(const float4 *__restrict__ stage, const int *__restrict__ ring_offsets,
                                                               const int *__restrict__ ring_vox, const int *__restrict__ vox_off,
                                                               float4 *__restrict__ out)
{
    const int ring = blockIdx.x;
    const int n = ring_vox[ring], src = ring_offsets[ring * 4 + 3], dst = vox_off[ring];
    for (int t = threadIdx.x; t < n; t += 256) out[dst + t] = stage[src + t];
}

int ring_voxel_run(mlh_ctx *ctx, float leaf)
{
    ScanBuf &sb = ctx->scan;
    if (!sb.extracted) return fail(ctx, MLH_ERR_STATE, "extract_run has not been called");
    hipStream_t st = ctx->stream;
    const int R = sb.n_rings;
    MLH_HIP(ctx, sb.vox_stage.ensure(sizeof(float4) * size_t(sb.n)));
    MLH_HIP(ctx, sb.vox_out.ensure(sizeof(float4) * size_t(sb.n)));
    MLH_HIP(ctx, sb.ring_vox.ensure(sizeof(int) * size_t(2 * R + 1)));
    int P = 64;
    while (P < sb.max_ring_len + 1) P <<= 1;
    const size_t lds = sizeof(unsigned long long) * size_t(P);
    if (lds > 150 * 1024) return fail(ctx, MLH_ERR_UNSUPPORTED, "ring too long for the LDS-resident voxel sort");
    MLH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(ring_voxel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    RingVoxelArgs A;
    A.pts = sb.pts.as<float4>(); A.list3 = sb.lists[3].as<int>(); A.ring_counts = sb.ring_counts.as<int>();
    A.ring_offsets = sb.ring_offsets.as<int>(); A.stage = sb.vox_stage.as<float4>(); A.ring_vox = sb.ring_vox.as<int>();
    A.leaf = leaf; A.sort_p = P;
    prof_begin(ctx, MLH_K_EXTRACT);
    hipLaunchKernelGGL(ring_voxel_kernel, dim3(R), dim3(256), lds, st, A);
    hipLaunchKernelGGL(ring_vox_offsets_kernel, dim3(1), dim3(64), 0, st, sb.ring_vox.as<int>(), R, sb.ring_vox.as<int>() + R, sb.ring_vox.as<int>() + 2 * R);
    hipLaunchKernelGGL(ring_vox_compact_kernel, dim3(R), dim3(256), 0, st, sb.vox_stage.as<float4>(), sb.ring_offsets.as<int>(),
                       sb.ring_vox.as<int>(), sb.ring_vox.as<int>() + R, sb.vox_out.as<float4>());
    prof_end(ctx, MLH_K_EXTRACT);
    MLH_HIP(ctx, hipGetLastError());
    sb.voxelised = true;
    return MLH_OK;
}

// ---------------------------------------------------------------- point uncertainty
struct UctArgs {
    const unsigned char *src;   // records
    int stride, n, intensity_off;
    const double *ext;          // n_lidar x 7  [t, q(xyzw)]
    const double *ext_cov;      // n_lidar x 36
    int n_lidar;
    double meas[9];
    double trace_thr;           // <= 0: keep everything
    float *cov6;                // n x 6 (f32 cov_vec)
    int *keep;                  // n
};

__global__ __launch_bounds__(256) void point_uncertainty_kernel(UctArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    const float *rec = reinterpret_cast<const float *>(A.src + size_t(i) * A.stride);
    const float inten = A.intensity_off >= 0 ? *reinterpret_cast<const float *>(A.src + size_t(i) * A.stride + A.intensity_off) : 0.f;
    int idx = int(inten);
    idx = idx < 0 ? 0 : (idx >= A.n_lidar ? A.n_lidar - 1 : idx);
    const double *e = A.ext + idx * 7;
    const q4 q{e[3], e[4], e[5], e[6]};
    const d3 t{e[0], e[1], e[2]};
    // point_sel = pose_ext^-1 * point_ori, through pointAssociateToMap (f64 math, f32 store) -- cpp:382
    const q4 qi{-q.x, -q.y, -q.z, q.w};
    const d3 mt = qrot(qi, t);
    const d3 ps = qrot(qi, d3{double(rec[0]), double(rec[1]), double(rec[2])});
    const float sel[3] = {float(ps.x - mt.x), float(ps.y - mt.y), float(ps.z - mt.z)};
    // T * [p; 1]
    double R[9];
    qtorot(q, R);
    const double p[3] = {double(sel[0]), double(sel[1]), double(sel[2])};
    double tp[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) tp[r] = R[r * 3 + 0] * p[0] + R[r * 3 + 1] * p[1] + R[r * 3 + 2] * p[2] + (r == 0 ? t.x : (r == 1 ? t.y : t.z));
    // G = [ I | -[tp]x | R ]  (3 x 9)
    double G[3][9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) { G[r][c] = (r == c) ? 1.0 : 0.0; G[r][6 + c] = R[r * 3 + c]; }
    G[0][3] = 0.0;    G[0][4] = tp[2];  G[0][5] = -tp[1];
    G[1][3] = -tp[2]; G[1][4] = 0.0;    G[1][5] = tp[0];
    G[2][3] = tp[1];  G[2][4] = -tp[0]; G[2][5] = 0.0;
    const double *Cp = A.ext_cov + idx * 36;
    double GC[3][9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) s += G[r][k] * Cp[k * 6 + c];
            GC[r][c] = s;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += G[r][6 + k] * A.meas[k * 3 + c];
            GC[r][6 + c] = s;
        }
    }
    double cov[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) s += GC[r][k] * G[c][k];
            cov[r][c] = s;
        }
    const double tr = cov[0][0] + cov[1][1] + cov[2][2];
    A.keep[i] = (A.trace_thr > 0.0 && tr > A.trace_thr) ? 0 : 1;
    float *o = A.cov6 + size_t(i) * 6;
    o[0] = float(cov[0][0]); o[1] = float(cov[0][1]); o[2] = float(cov[0][2]);
    o[3] = float(cov[1][1]); o[4] = float(cov[1][2]); o[5] = float(cov[2][2]);
}

int point_uncertainty_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int mem, const double *ext_poses,
                          const double *ext_covs, int n_lidar, const double cov_meas[9], double trace_thr, float *cov6_host, int *keep_host)
{
    if (!points || n <= 0 || stride < 12 || (stride & 3) || n_lidar <= 0 || n_lidar > 16) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    hipStream_t st = ctx->stream;
    const unsigned char *src = static_cast<const unsigned char *>(points);
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, ctx->tmp.ensure(size_t(n) * stride));
        MLH_HIP(ctx, hipMemcpyAsync(ctx->tmp.p, points, size_t(n) * stride, hipMemcpyHostToDevice, st));
        src = ctx->tmp.as<unsigned char>();
    }
    MLH_HIP(ctx, ctx->uct_buf.ensure(sizeof(double) * size_t(n_lidar) * 43 + sizeof(float) * 6 * size_t(n) + sizeof(int) * size_t(n) + 64));
    double *d_ext = ctx->uct_buf.as<double>();
    double *d_cov = d_ext + size_t(n_lidar) * 7;
    float *d_c6 = reinterpret_cast<float *>(d_cov + size_t(n_lidar) * 36);
    int *d_keep = reinterpret_cast<int *>(d_c6 + size_t(n) * 6);
    MLH_HIP(ctx, hipMemcpyAsync(d_ext, ext_poses, sizeof(double) * 7 * n_lidar, hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipMemcpyAsync(d_cov, ext_covs, sizeof(double) * 36 * n_lidar, hipMemcpyHostToDevice, st));
    UctArgs A;
    A.src = src; A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.ext = d_ext; A.ext_cov = d_cov; A.n_lidar = n_lidar;
    for (int i = 0; i < 9; ++i) A.meas[i] = cov_meas[i];
    A.trace_thr = trace_thr; A.cov6 = d_c6; A.keep = d_keep;
    hipLaunchKernelGGL(point_uncertainty_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(cov6_host, d_c6, sizeof(float) * 6 * size_t(n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipMemcpyAsync(keep_host, d_keep, sizeof(int) * size_t(n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    return MLH_OK;
}

}  // namespace mlh
