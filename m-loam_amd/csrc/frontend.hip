// Front-end kernels around the extractor (gfx950): per-point rigid transforms and gathers that keep the clouds on the device between
// extractCloud, the tracker and the mapper.
//   fuse_append_kernel        transformCloudFeature (estimator/src/utility/visualization.cpp:39-51) + concatenation
//   transform_cloud_kernel    pcl::transformPointCloud with a float 4x4 (estimator.cpp:1185-1192)
//   transform_to_end_kernel   TransformToEnd / TransformToStart (estimator/src/utility/utility.h:55-100; estimator.cpp:376-410)
//   gather_points_kernel      a feature list of the extractor as a dense cloud (estimator.cpp:426-427, 532-543)
#include "ctx.hpp"
#include "dev_math.hpp"
#include <cfloat>

namespace mlh {

struct FuseXf { float r[9], t[3], id; };     // a rigid transform in single precision (+ the LiDAR index of transformCloudFeature)

// rotation of the unit quaternion in double, rounded once to float: what Eigen::Matrix4f holds after `.cast<float>()`
static FuseXf xf_from_pose(const double pose[7], float id)
{
    const double tx = pose[0], ty = pose[1], tz = pose[2], qx = pose[3], qy = pose[4], qz = pose[5], qw = pose[6];
    const double R[9] = {1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw),
                         2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw),
                         2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)};
    FuseXf xf;
    for (int i = 0; i < 9; ++i) xf.r[i] = float(R[i]);
    xf.t[0] = float(tx); xf.t[1] = float(ty); xf.t[2] = float(tz);
    xf.id = id;
    return xf;
}

__global__ __launch_bounds__(256) void gather_points_kernel(const float4 *__restrict__ pts, const int *__restrict__ list, int n, float4 *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = pts[list[i]];
}

void gather_points_launch(mlh_ctx *ctx, const float4 *pts, const int *list, int n, float4 *out)
{
    MLH_LAUNCH(gather_points_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pts, list, n, out);
}

// pcl::transformPointCloud(cloud, cloud, pose.T_.cast<float>()) in place: p' = R p + t in single precision (R rounded once from the
// double rotation matrix of the unit quaternion), every other field kept -- the window clouds on their way into the pivot frame
// (estimator.cpp:1185-1192)
__global__ __launch_bounds__(256) void transform_cloud_kernel(unsigned char *p, int stride, int n, FuseXf xf)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float *r = reinterpret_cast<float *>(p + size_t(i) * stride);
    const float x = r[0], y = r[1], z = r[2];
    r[0] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(xf.r[0], x), __fmul_rn(xf.r[1], y)), __fmul_rn(xf.r[2], z)), xf.t[0]);
    r[1] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(xf.r[3], x), __fmul_rn(xf.r[4], y)), __fmul_rn(xf.r[5], z)), xf.t[1]);
    r[2] = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(xf.r[6], x), __fmul_rn(xf.r[7], y)), __fmul_rn(xf.r[8], z)), xf.t[2]);
}

int transform_cloud_launch(mlh_ctx *ctx, void *dev, int stride, int n, const double pose[7])
{
    if (n <= 0) return MLH_OK;
    MLH_LAUNCH(transform_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, static_cast<unsigned char *>(dev), stride, n, xf_from_pose(pose, 0.f));
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// TransformToEnd (utility.h:79-100) in place over strided records: p^b = T^-1 T(s) p^c with s = frac(intensity) / scan_period when
// b_distortion, else 1; T(s) = (slerp(s, q), s t) (Eigen 3.3 slerp from the identity), f64 math, the intermediate and the result
// rounded to f32 exactly where the reference stores them into float points
struct UndistArgs { unsigned char *p; int stride, n, intensity_off, b_distortion; float scan_period; double pose[7]; };
__global__ __launch_bounds__(256) void transform_to_end_kernel(UndistArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    float *rec = reinterpret_cast<float *>(A.p + size_t(i) * A.stride);
    const float inten = *reinterpret_cast<const float *>(A.p + size_t(i) * A.stride + A.intensity_off);
    double sI = 1.0;
    if (A.b_distortion) sI = double((inten - float(int(inten))) / A.scan_period);
    const q4 q{A.pose[3], A.pose[4], A.pose[5], A.pose[6]};
    const d3 t{A.pose[0], A.pose[1], A.pose[2]};
    // Identity.slerp(s, q)
    const double one = 1.0 - 2.220446049250313e-16;
    const double d = q.w, absD = fabs(d);
    double scale0, scale1;
    if (absD >= one) { scale0 = 1.0 - sI; scale1 = sI; }
    else {
        const double theta = acos(absD), sinTheta = sin(theta);
        scale0 = sin((1.0 - sI) * theta) / sinTheta;
        scale1 = sin(sI * theta) / sinTheta;
    }
    if (d < 0.0) scale1 = -scale1;
    const q4 qs{scale0 * 0.0 + scale1 * q.x, scale0 * 0.0 + scale1 * q.y, scale0 * 0.0 + scale1 * q.z, scale0 * 1.0 + scale1 * q.w};
    const d3 r = qrot(qs, d3{double(rec[0]), double(rec[1]), double(rec[2])});
    const float ux = float(r.x + sI * t.x), uy = float(r.y + sI * t.y), uz = float(r.z + sI * t.z);     // un_point_tmp (a float point)
    const double n2 = (q.x * q.x + q.z * q.z) + (q.y * q.y + q.w * q.w);
    const q4 qi{-q.x / n2, -q.y / n2, -q.z / n2, q.w / n2};
    const d3 e = qrot(qi, d3{double(ux) - t.x, double(uy) - t.y, double(uz) - t.z});
    rec[0] = float(e.x); rec[1] = float(e.y); rec[2] = float(e.z);
}

int transform_to_end_launch(mlh_ctx *ctx, void *dev, int stride, int n, int intensity_off, const double pose[7], int b_distortion, float scan_period)
{
    if (n <= 0) return MLH_OK;
    UndistArgs A;
    A.p = static_cast<unsigned char *>(dev); A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.b_distortion = b_distortion;
    A.scan_period = scan_period;
    for (int i = 0; i < 7; ++i) A.pose[i] = pose[i];
    MLH_LAUNCH(transform_to_end_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, A);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// transformCloudFeature (visualization.cpp:39-51): p' = R p + t in single precision, intensity <- LiDAR index
struct FuseArgs {
    const float4 *pts, *vox_out;
    const int *list1, *ring_offsets, *vox_off;
    int rb, re;                // rings [rb, re) of the scan
    FuseXf xf;
    float4 *out[2];            // fused surf / corner clouds
    const int *cnt;            // record counts before this append (not read by the first append after a reset: `first`) ...
    int first;
    int *cnt_next;             // ... and after it (the next append's base): a prefix table, one pair per append
    float *part;               // this append's partial bounds: [kind][FUSE_BLOCKS][6]
};
__global__ __launch_bounds__(256) void fuse_append_kernel(FuseArgs A)
{
    __shared__ float lds[4][6];
    const int kind = blockIdx.y;                       // 0: surf <- voxel-thinned less-flat, 1: corner <- less-sharp
    const int b = kind == 0 ? A.vox_off[A.rb] : A.ring_offsets[A.rb * 4 + 1];
    const int e = kind == 0 ? A.vox_off[A.re] : A.ring_offsets[A.re * 4 + 1];
    const int base = A.first ? 0 : A.cnt[kind];
    if (blockIdx.x == 0 && threadIdx.x == 0) A.cnt_next[kind] = base + (e - b);      // depends on nothing the other workgroups do: no bump launch
    const FuseXf &xf = A.xf;
    float m[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < e - b; i += FUSE_BLOCKS * 256) {
        const float4 p = kind == 0 ? A.vox_out[b + i] : A.pts[A.list1[b + i]];
        float4 o;
        // products and sums kept separate (no contraction) so that the result is one well-defined float32 expression
        o.x = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(xf.r[0], p.x), __fmul_rn(xf.r[1], p.y)), __fmul_rn(xf.r[2], p.z)), xf.t[0]);
        o.y = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(xf.r[3], p.x), __fmul_rn(xf.r[4], p.y)), __fmul_rn(xf.r[5], p.z)), xf.t[1]);
        o.z = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(xf.r[6], p.x), __fmul_rn(xf.r[7], p.y)), __fmul_rn(xf.r[8], p.z)), xf.t[2]);
        o.w = xf.id;
        A.out[kind][base + i] = o;
        m[0] = fminf(m[0], o.x); m[1] = fminf(m[1], o.y); m[2] = fminf(m[2], o.z);
        m[3] = fmaxf(m[3], o.x); m[4] = fmaxf(m[4], o.y); m[5] = fmaxf(m[5], o.z);
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { m[d] = fminf(m[d], __shfl_xor(m[d], off)); m[3 + d] = fmaxf(m[3 + d], __shfl_xor(m[3 + d], off)); }
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) lds[threadIdx.x >> 6][d] = m[d];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int d = threadIdx.x;
        float r = lds[0][d];
        for (int w = 1; w < 4; ++w) r = d < 3 ? fminf(r, lds[w][d]) : fmaxf(r, lds[w][d]);
        A.part[(kind * FUSE_BLOCKS + blockIdx.x) * 6 + d] = r;
    }
}
// the scan's thinned less-flat cloud -> fused SURF, its less-sharp corners -> fused CORNER (rings [ring_begin, ring_end)); counts on the device
int fuse_append_launch(mlh_ctx *ctx, ScanBuf &sb, int ring_begin, int ring_end, int lidar_idx, const double ext_pose[7])
{
    hipStream_t st = ctx->stream;
    FuseArgs A;
    A.xf = xf_from_pose(ext_pose, float(lidar_idx));
    // the counts live on the device (no host round trip per scan); capacity follows a host-side upper bound: the scan's point count
    for (int k = 0; k < 2; ++k) {
        MLH_HIP(ctx, ctx->fused[k].grow(sizeof(float4) * (ctx->fused_bound[k] + size_t(sb.n)), sizeof(float4) * ctx->fused_bound[k], st));
        ctx->fused_bound[k] += size_t(sb.n);
        A.out[k] = ctx->fused[k].as<float4>();
    }
    A.pts = sb.pts.as<float4>(); A.vox_out = sb.vox_out.as<float4>(); A.list1 = sb.lists[1].as<int>();
    A.ring_offsets = sb.ring_offsets.as<int>(); A.vox_off = sb.ring_vox.as<int>() + sb.n_rings;
    // counts: prefix table [append][kind]; this append reads row `fused_parts` and writes row `fused_parts + 1`
    MLH_HIP(ctx, ctx->fused_cnt.grow(sizeof(int) * 2 * size_t(ctx->fused_parts + 2), sizeof(int) * 2 * size_t(ctx->fused_parts + 1), st));
    A.first = ctx->fused_parts == 0 ? 1 : 0;
    A.rb = ring_begin; A.re = ring_end; A.cnt = ctx->fused_cnt.as<int>() + 2 * ctx->fused_parts; A.cnt_next = ctx->fused_cnt.as<int>() + 2 * (ctx->fused_parts + 1);
    const size_t part_floats = size_t(2) * FUSE_BLOCKS * 6;
    MLH_HIP(ctx, ctx->fused_part.grow(sizeof(float) * part_floats * size_t(ctx->fused_parts + 1), sizeof(float) * part_floats * size_t(ctx->fused_parts), st));
    A.part = ctx->fused_part.as<float>() + part_floats * size_t(ctx->fused_parts);
    ++ctx->fused_parts;
    MLH_LAUNCH(fuse_append_kernel, dim3(FUSE_BLOCKS, 2), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

}  // namespace mlh
