// Voxel thinning and uncertainty propagation kernels (gfx950).
//
// ring_voxel_kernel      the per-ring pcl::VoxelGrid<PointXYZI>(0.2 m) that closes FeatureExtract::extractCloud
//                        (estimator/src/featureExtract/feature_extract.cpp:266-271; PCL 1.8.0 filters/impl/voxel_grid.hpp):
//                        one workgroup per ring, the ring's less-flat points (label <= 0, cpp:258-264) are keyed with PCL's
//                        voxel index (floor(x * inv_leaf) - min_b, x fastest), sorted on 64-bit (voxel, position) keys by a
//                        register-resident bitonic network (sort_dev.hpp: 1024 threads, only the cross-wavefront stages touch
//                        LDS), and every voxel's members are averaged (CentroidPoint: f32 sums / count) -- by default in the
//                        order libstdc++'s std::sort leaves them (PCL's own order: MODE 1 writes the voxel indices, stdsort.hip
//                        sorts every ring's list, MODE 2 sums along that permutation), in position order with member order 0.
//                        Output order = ring asc, voxel index asc, as the reference concatenates them.
// point_uncertainty_kernel   evalPointUncertainty (estimator/src/lidarMapper/associate_uct.hpp:196-215) as used by
//                        downsampleCurrentScan (lidar_mapper_keyframe.cpp:375-418): per point, Sigma_p = [G diag(Sigma_ext,
//                        Sigma_meas) G^T]_3x3 with G = [(T p)^odot | T D], f64, stored to the f32 cov_vec of PointXYZIWithCov;
//                        points whose trace exceeds TRACE_THRESHOLD_MAPPING are dropped (order-preserving compaction).
#include "ctx.hpp"
#include <chrono>
#include <cstdlib>
#include "dev_math.hpp"
#include "sort_dev.hpp"
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
    int *vkeys;                 // MODE 1: PCL voxel index of every less-flat point, in list order (the keys std::sort sees)
    const int *perm;            // MODE 2: the order std::sort leaves them in (global positions of the less-flat list), voxel after voxel
    const int *skeys;           // MODE 2: the voxel indices in that order (the sort's own key array, sorted ring by ring)
    int *zero;                  // MODE 1: n_zero words the first workgroup clears on its way (the range counters of the sort that follows)
    int n_zero;
};

constexpr int RV_TPB = 1024, RV_WAVES = RV_TPB / 64;

// KPL keys per thread: a ring of up to 1024 * KPL less-flat points. PTS_IN_LDS: the ring's points are parked in LDS on the first
// pass so that the centroid walks (a dependent chain per voxel) never go back to memory.
// MODE 0: a voxel's members are summed in position order (what a stable sort would give). MODE 1 + 2: in the order libstdc++'s unstable
// std::sort leaves them, which is the order PCL sums them in (voxel_grid.hpp: std::sort(index_vector) with a comparator on the voxel index
// only) -- pass 1 writes the voxel indices, stdsort.hip produces std::sort's permutation per ring, pass 2 sums along it: the f32 sums
// then associate as the reference's do and the centroids are its centroids bit for bit.
template <int KPL, bool PTS_IN_LDS, int MODE>
__global__ __launch_bounds__(RV_TPB) void ring_voxel_kernel(RingVoxelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int N = RV_TPB * KPL;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem);
    float4 *spts = reinterpret_cast<float4 *>(keys + N);
    __shared__ float s_red[RV_WAVES][6];
    __shared__ int s_scan[RV_WAVES];
    const int ring = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = A.ring_counts[ring * 4 + 3];
    const int off = A.ring_offsets[ring * 4 + 3];
    if (MODE == 1 && ring == 0 && tid < A.n_zero) A.zero[tid] = 0;
    if (n <= 0) { if (tid == 0) A.ring_vox[ring] = 0; return; }
    const float inv = 1.0f / A.leaf;

    // the thread's KPL points (element t = tid * KPL + r of the ring's less-flat run), all loads in flight together
    float4 p[KPL];
    int gi[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) gi[r] = A.list3[off + min(tid * KPL + r, n - 1)];
#pragma unroll
    for (int r = 0; r < KPL; ++r) p[r] = A.pts[gi[r]];
    // bounds (getMinMax3D) -> min_b, div_b (voxel_grid.hpp)
    float m[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        if (tid * KPL + r < n) {
            m[0] = fminf(m[0], p[r].x); m[3] = fmaxf(m[3], p[r].x);
            m[1] = fminf(m[1], p[r].y); m[4] = fmaxf(m[4], p[r].y);
            m[2] = fminf(m[2], p[r].z); m[5] = fmaxf(m[5], p[r].z);
            if (PTS_IN_LDS) spts[tid * KPL + r] = p[r];
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m[d] = fminf(m[d], __shfl_xor(m[d], o)); m[3 + d] = fmaxf(m[3 + d], __shfl_xor(m[3 + d], o)); }
    }
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 6; ++d) s_red[wave][d] = m[d];
    }
    __syncthreads();
    int min_b[3], div_b[3];
    long long cells = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float lo = s_red[0][d], hi = s_red[0][3 + d];
#pragma unroll
        for (int w = 1; w < RV_WAVES; ++w) { lo = fminf(lo, s_red[w][d]); hi = fmaxf(hi, s_red[w][3 + d]); }
        min_b[d] = int(floorf(lo * inv));
        div_b[d] = int(floorf(hi * inv)) - min_b[d] + 1;
        cells *= (long long)((hi - lo) * inv) + 1;
    }
    if (cells > 2147483647ll) {   // "Leaf size is too small for the input dataset": PCL returns the input unchanged
#pragma unroll
        for (int r = 0; r < KPL; ++r) if (tid * KPL + r < n) { if (MODE == 1) A.vkeys[off + tid * KPL + r] = 0; else A.stage[off + tid * KPL + r] = p[r]; }
        if (MODE != 1 && tid == 0) A.ring_vox[ring] = n;
        return;
    }
    const int mul1 = div_b[0], mul2 = div_b[0] * div_b[1];
    // keys: (voxel index << 32) | position in the ring's less-flat run; padding = all ones
    unsigned long long v[KPL];
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int t = tid * KPL + r;
        const int ijk0 = int(floorf(p[r].x * inv) - float(min_b[0]));
        const int ijk1 = int(floorf(p[r].y * inv) - float(min_b[1]));
        const int ijk2 = int(floorf(p[r].z * inv) - float(min_b[2]));
        const unsigned idx = unsigned(ijk0 + ijk1 * mul1 + ijk2 * mul2);
        v[r] = t < n ? (((unsigned long long)idx << 32) | unsigned(t)) : ~0ull;
        if (MODE == 1 && t < n) A.vkeys[off + t] = int(idx);
    }
    if (MODE == 1) return;
    if constexpr (MODE == 2) {
        // std::sort has already grouped the ring's points by voxel (ascending voxel index, its own order inside a voxel): element e of the sorted run is
        // (voxel, position of the member in the ring's list) -- no second sort here, and the walk below never leaves LDS (round 4 sorted (voxel, position)
        // keys again with a bitonic network and fetched every member's position from the permutation in HBM: 33 us of the front end)
#pragma unroll
        for (int r = 0; r < KPL; ++r) {
            const int e = tid * KPL + r;
            v[r] = e < n ? ((static_cast<unsigned long long>(unsigned(A.skeys[off + e])) << 32) | unsigned(A.perm[off + e] - off)) : ~0ull;
        }
    } else {
        block_bitonic_sort<KPL, RV_WAVES>(v, keys, tid);
    }
#pragma unroll
    for (int r = 0; r < KPL; ++r) keys[tid * KPL + r] = v[r];
    __syncthreads();
    // every voxel start computes its centroid (members in position order); output slot = number of voxel starts before it
    bool start[KPL];
    int mine = 0;
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        const int e = tid * KPL + r;
        const unsigned vox = unsigned(v[r] >> 32);
        const unsigned prev = r > 0 ? unsigned(v[r > 0 ? r - 1 : 0] >> 32) : (e > 0 ? unsigned(keys[e - 1] >> 32) : ~vox);
        start[r] = e < n && (e == 0 || prev != vox);
        mine += start[r] ? 1 : 0;
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int slot = incl - mine, total = 0;
#pragma unroll
    for (int w = 0; w < RV_WAVES; ++w) { const int c = s_scan[w]; slot += w < wave ? c : 0; total += c; }
#pragma unroll
    for (int r = 0; r < KPL; ++r) {
        if (start[r]) {
            const unsigned vox = unsigned(v[r] >> 32);
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            int cnt = 0;
            for (int u = tid * KPL + r; u < n; ++u) {
                const unsigned long long ku = keys[u];
                if (unsigned(ku >> 32) != vox) break;
                const int pos = int(unsigned(ku));
                const float4 q = PTS_IN_LDS ? spts[pos] : A.pts[A.list3[off + pos]];
                sx += q.x; sy += q.y; sz += q.z; si += q.w;
                ++cnt;
            }
            const float fc = float(cnt);
            A.stage[off + slot] = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
            ++slot;
        }
    }
    if (tid == 0) A.ring_vox[ring] = total;
}

// concatenation of the rings' centroids (ring asc): every workgroup sums the counts of the rings before its own (a few dozen
// words) and notes it in vox_off (ring_vox + n_rings; entry n_rings = the total)
__global__ __launch_bounds__(256) void ring_vox_compact_kernel(const float4 *__restrict__ stage, const int *__restrict__ ring_offsets,
                                                               const int *__restrict__ ring_vox, int n_rings, int *__restrict__ vox_off /* n_rings + 1 */,
                                                               float4 *__restrict__ out)
{
    __shared__ int s_dst;
    const int ring = blockIdx.x;
    if (threadIdx.x < 64) {
        int acc = 0;
        for (int r = threadIdx.x; r < ring; r += 64) acc += ring_vox[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (threadIdx.x == 0) s_dst = acc;
    }
    __syncthreads();
    const int n = ring_vox[ring], src = ring_offsets[ring * 4 + 3], dst = s_dst;
    if (threadIdx.x == 0) { vox_off[ring] = dst; if (ring == n_rings - 1) vox_off[n_rings] = dst + n; }
    for (int t = threadIdx.x; t < n; t += 256) out[dst + t] = stage[src + t];
}

template <int KPL, bool PTS_IN_LDS, int MODE>
static hipError_t ring_voxel_launch(const RingVoxelArgs &A, int R, hipStream_t st)
{
    const size_t lds = size_t(RV_TPB) * KPL * (sizeof(unsigned long long) + (PTS_IN_LDS ? sizeof(float4) : 0));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ring_voxel_kernel<KPL, PTS_IN_LDS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
    if (e != hipSuccess) return e;
    MLH_LAUNCH((ring_voxel_kernel<KPL, PTS_IN_LDS, MODE>), dim3(R), dim3(RV_TPB), lds, st, A);
    return hipSuccess;
}
template <int MODE>
static hipError_t ring_voxel_launch_any(const RingVoxelArgs &A, int R, int longest, hipStream_t st)
{
    if (longest <= RV_TPB) return ring_voxel_launch<1, true, MODE>(A, R, st);
    if (longest <= RV_TPB * 2) return ring_voxel_launch<2, true, MODE>(A, R, st);
    if (longest <= RV_TPB * 4) return ring_voxel_launch<4, true, MODE>(A, R, st);
    return ring_voxel_launch<8, false, MODE>(A, R, st);
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
    const int longest = sb.max_ring_len + 1;      // upper bound of a ring's less-flat count
    RingVoxelArgs A;
    A.pts = sb.pts.as<float4>(); A.list3 = sb.lists[3].as<int>(); A.ring_counts = sb.ring_counts.as<int>();
    A.ring_offsets = sb.ring_offsets.as<int>(); A.stage = sb.vox_stage.as<float4>(); A.ring_vox = sb.ring_vox.as<int>();
    A.leaf = leaf; A.vkeys = nullptr; A.perm = nullptr; A.skeys = nullptr; A.zero = nullptr; A.n_zero = 0;
    if (longest > RV_TPB * 8) return fail(ctx, MLH_ERR_UNSUPPORTED, "ring longer than 8192 points: too long for the LDS-resident voxel sort");
    prof_begin(ctx, MLH_K_EXTRACT);
    if (ctx->vox_member_order != 0) {
        // the reference's member order: voxel indices out, std::sort's permutation of every ring's list (stdsort.hip), sums along it
        MLH_HIP(ctx, sb.vox_keys.ensure(sizeof(int) * size_t(sb.n)));
        MLH_HIP(ctx, sb.vox_perm.ensure(sizeof(int) * size_t(sb.n)));
        A.vkeys = sb.vox_keys.as<int>();
        A.zero = device_std_sort_counters(ctx, sb.n, &A.n_zero);         // cleared by pass 1's first workgroup: no fill launches in front of the sort
        if (!A.zero) return fail(ctx, MLH_ERR_HIP, "std::sort scratch");
        MLH_HIP(ctx, ring_voxel_launch_any<1>(A, R, longest, st));
        int rc = device_std_sort_segments(ctx, A.vkeys, A.ring_counts, A.ring_offsets, 4, 3, R, sb.n, longest, sb.vox_perm.as<int>(), true);
        if (rc) return rc;
        A.perm = sb.vox_perm.as<int>();
        A.skeys = device_std_sort_keys(ctx, sb.n);
        if (!A.skeys) return fail(ctx, MLH_ERR_HIP, "std::sort scratch");
        MLH_HIP(ctx, ring_voxel_launch_any<2>(A, R, longest, st));
    } else {
        MLH_HIP(ctx, ring_voxel_launch_any<0>(A, R, longest, st));
    }
    MLH_LAUNCH(ring_vox_compact_kernel, dim3(R), dim3(256), 0, st, (const float4 *)sb.vox_stage.as<float4>(), (const int *)sb.ring_offsets.as<int>(),
                       (const int *)sb.ring_vox.as<int>(), R, sb.ring_vox.as<int>() + R, sb.vox_out.as<float4>());
    prof_end(ctx, MLH_K_EXTRACT);
    MLH_HIP(ctx, hipGetLastError());
    sb.voxelised = true;
    sb.h_vox_valid = false;
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
    float *cov6;                // n x 6 (f32 cov_vec), or null
    int *keep;                  // n
    // cloudUCTAssociateToMap mode (rec_out != null): the uncertainty is evaluated with the compound pose of the point's LiDAR,
    // the record is copied, moved to the map frame with gpose and receives cov_vec / cov_trace
    const double *upose;        // n_lidar x 7: pose handed to evalPointUncertainty (= ext for downsampleCurrentScan)
    const double *upose_cov;    // n_lidar x 36
    unsigned char *rec_out;     // n staged records (input layout), or null
    double gpose[7];
    int cov_off, trace_off, with_ua;
    int *keep2 = nullptr;
    const int *n_dev = nullptr;  // optional device-side record count (<= n): the launch covers n, records past *n_dev are dropped
};

// evalPointUncertainty of one record (associate_uct.hpp:196-215) as downsampleCurrentScan / cloudUCTAssociateToMap call it: the point is taken back into its
// LiDAR's frame through that LiDAR's extrinsic (pointAssociateToMap: f64 math, f32 store), then cov = G diag(pose covariance, measurement covariance) G^T with
// G = [ I | -[T p]x | R ]. One body for point_uncertainty_kernel and the thinning pipeline's fused aggregate (vsp_aggregate_kernel): the same operations in the same order.
__device__ __forceinline__ void eval_point_cov(const double *ext, const double *upose, const double *upose_cov, int n_lidar, const double *meas, int with_ua,
                                               float x, float y, float z, float inten, double (&cov)[3][3])
{
    int idx = int(inten);
    idx = idx < 0 ? 0 : (idx >= n_lidar ? n_lidar - 1 : idx);
    for (int r_ = 0; r_ < 3; ++r_) for (int c_ = 0; c_ < 3; ++c_) cov[r_][c_] = 0.0;
    if (with_ua) {
        const double *e = ext + idx * 7;
        const q4 qe{e[3], e[4], e[5], e[6]};
        const d3 te{e[0], e[1], e[2]};
        // point_sel = pose_ext^-1 * point_ori, through pointAssociateToMap (f64 math, f32 store) -- cpp:382 / cpp:1148
        const q4 qi{-qe.x, -qe.y, -qe.z, qe.w};
        const d3 mt = qrot(qi, te);
        const d3 ps = qrot(qi, d3{double(x), double(y), double(z)});
        const float sel[3] = {float(ps.x - mt.x), float(ps.y - mt.y), float(ps.z - mt.z)};
        const double *u = upose + idx * 7;
        const q4 q{u[3], u[4], u[5], u[6]};
        const d3 t{u[0], u[1], u[2]};
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
        const double *Cp = upose_cov + idx * 36;
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
                for (int k = 0; k < 3; ++k) s += G[r][6 + k] * meas[k * 3 + c];
                GC[r][6 + c] = s;
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 9; ++k) s += GC[r][k] * G[c][k];
                cov[r][c] = s;
            }
    }
}

__global__ __launch_bounds__(256) void point_uncertainty_kernel(UctArgs A)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A.n) return;
    if (A.n_dev && i >= *A.n_dev) { A.keep[i] = 0; if (A.keep2) A.keep2[i] = 0; return; }
    const float *rec = reinterpret_cast<const float *>(A.src + size_t(i) * A.stride);
    const float inten = A.intensity_off >= 0 ? *reinterpret_cast<const float *>(A.src + size_t(i) * A.stride + A.intensity_off) : 0.f;
    double cov[3][3];
    eval_point_cov(A.ext, A.upose, A.upose_cov, A.n_lidar, A.meas, A.with_ua, rec[0], rec[1], rec[2], inten, cov);
    const double tr = cov[0][0] + cov[1][1] + cov[2][2];
    const int keep = (A.with_ua && A.trace_thr > 0.0 && tr > A.trace_thr) ? 0 : 1;
    A.keep[i] = keep;
    if (A.keep2) A.keep2[i] = keep;       // a second copy for the in-place scan that turns the flags into output slots
    const float c6[6] = {float(cov[0][0]), float(cov[0][1]), float(cov[0][2]), float(cov[1][1]), float(cov[1][2]), float(cov[2][2])};
    if (A.cov6) {
        float *o = A.cov6 + size_t(i) * 6;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[k] = c6[k];
    }
    if (A.rec_out && keep) {
        // point_cov = pose_global * point_ori with the new covariance (cpp:1152-1154); the other fields travel unchanged (po = pi)
        unsigned char *o = A.rec_out + size_t(i) * A.stride;
        for (int k = 0; k < A.stride / 4; ++k) reinterpret_cast<float *>(o)[k] = rec[k];
        const d3 g = qrot(q4{A.gpose[3], A.gpose[4], A.gpose[5], A.gpose[6]}, d3{double(rec[0]), double(rec[1]), double(rec[2])});
        float *ox = reinterpret_cast<float *>(o);
        ox[0] = float(g.x + A.gpose[0]); ox[1] = float(g.y + A.gpose[1]); ox[2] = float(g.z + A.gpose[2]);
        if (A.cov_off >= 0) {
            float *oc = reinterpret_cast<float *>(o + A.cov_off);
#pragma unroll
            for (int k = 0; k < 6; ++k) oc[k] = c6[k];
        }
        if (A.trace_off >= 0) *reinterpret_cast<float *>(o + A.trace_off) = float(tr);
    }
}

__global__ __launch_bounds__(256) void compact_records_kernel(const unsigned char *__restrict__ staged, const int *__restrict__ keep,
                                                              const int *__restrict__ slot, int n, int stride, unsigned char *__restrict__ dst)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const float *s = reinterpret_cast<const float *>(staged + size_t(i) * stride);
    float *d = reinterpret_cast<float *>(dst + size_t(slot[i]) * stride);
    for (int k = 0; k < stride / 4; ++k) d[k] = s[k];
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
    A.upose = d_ext; A.upose_cov = d_cov; A.rec_out = nullptr; A.cov_off = A.trace_off = -1; A.with_ua = 1;
    for (int i = 0; i < 7; ++i) A.gpose[i] = 0.0;
    MLH_LAUNCH(point_uncertainty_kernel, dim3((n + 255) / 256), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    MLH_HIP(ctx, hipMemcpyAsync(cov6_host, d_c6, sizeof(float) * 6 * size_t(n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipMemcpyAsync(keep_host, d_keep, sizeof(int) * size_t(n), hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    return MLH_OK;
}

// ---------------------------------------------------------------- cloudUCTAssociateToMap
namespace {
struct M3 { double a[9]; };
struct M6 { double a[36]; };
M3 mul(const M3 &A, const M3 &B) { M3 C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += A.a[r * 3 + k] * B.a[k * 3 + c]; C.a[r * 3 + c] = s; } return C; }
M3 tr(const M3 &A) { M3 B; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B.a[r * 3 + c] = A.a[c * 3 + r]; return B; }
M3 add(const M3 &A, const M3 &B) { M3 C; for (int i = 0; i < 9; ++i) C.a[i] = A.a[i] + B.a[i]; return C; }
M6 mul(const M6 &A, const M6 &B) { M6 C; for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += A.a[r * 6 + k] * B.a[k * 6 + c]; C.a[r * 6 + c] = s; } return C; }
M6 tr(const M6 &A) { M6 B; for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) B.a[r * 6 + c] = A.a[c * 6 + r]; return B; }
M3 blk(const M6 &A, int r0, int c0) { M3 B; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B.a[r * 3 + c] = A.a[(r0 + r) * 6 + c0 + c]; return B; }
void put(M6 &A, int r0, int c0, const M3 &B) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) A.a[(r0 + r) * 6 + c0 + c] = B.a[r * 3 + c]; }
// associate_uct.hpp:18-28: op1(B) = -tr(B) I + B;  op2(B, C) = op1(B) op1(C) + op1(C B)
M3 op1(const M3 &B) { M3 A = B; const double t = B.a[0] + B.a[4] + B.a[8]; A.a[0] -= t; A.a[4] -= t; A.a[8] -= t; return A; }
M3 op2(const M3 &B, const M3 &C) { return add(mul(op1(B), op1(C)), op1(mul(C, B))); }
}  // namespace

// compoundPoseWithCov, method 2 (associate_uct.hpp:90-147): pose_cp = pose_1 * pose_2 with the fourth-order covariance of the compound
void compound_pose_with_cov(const double p1[7], const double c1[36], const double p2[7], const double c2[36], double pc[7], double cc[36])
{
    const q4 q1{p1[3], p1[4], p1[5], p1[6]}, q2{p2[3], p2[4], p2[5], p2[6]};
    const q4 q = qmul(q1, q2);
    const d3 rt = qrot(q1, d3{p2[0], p2[1], p2[2]});
    pc[0] = rt.x + p1[0]; pc[1] = rt.y + p1[1]; pc[2] = rt.z + p1[2];
    pc[3] = q.x; pc[4] = q.y; pc[5] = q.z; pc[6] = q.w;
    M3 R, S;
    qtorot(q1, R.a);
    const double sk[9] = {0.0, -p1[2], p1[1], p1[2], 0.0, -p1[0], -p1[1], p1[0], 0.0};
    for (int i = 0; i < 9; ++i) S.a[i] = sk[i];
    M6 Ad{}, C1, C2;
    put(Ad, 0, 0, R); put(Ad, 0, 3, mul(S, R)); put(Ad, 3, 3, R);
    for (int i = 0; i < 36; ++i) { C1.a[i] = c1[i]; C2.a[i] = c2[i]; }
    const M6 C2p = mul(mul(Ad, C2), tr(Ad));
    const M3 rr1 = blk(C1, 0, 0), rp1 = blk(C1, 0, 3), pp1 = blk(C1, 3, 3);
    const M3 rr2 = blk(C2p, 0, 0), rp2 = blk(C2p, 0, 3), pp2 = blk(C2p, 3, 3);
    M6 A1{}, A2{}, B{};
    put(A1, 0, 0, op1(pp1)); put(A1, 0, 3, op1(add(rp1, tr(rp1)))); put(A1, 3, 3, op1(pp1));
    put(A2, 0, 0, op1(pp2)); put(A2, 0, 3, op1(add(rp2, tr(rp2)))); put(A2, 3, 3, op1(pp2));
    const M3 Brr = add(add(add(op2(pp1, rr2), op2(tr(rp1), rp2)), op2(rp1, tr(rp2))), op2(rr1, pp2));
    const M3 Brp = add(op2(pp1, tr(rp2)), op2(tr(rp1), pp2));
    put(B, 0, 0, Brr); put(B, 0, 3, Brp); put(B, 3, 0, tr(Brp)); put(B, 3, 3, op2(pp1, pp2));
    const M6 m1 = mul(A1, C2p), m2 = mul(C2p, tr(A1)), m3 = mul(A2, C1), m4 = mul(C1, tr(A2));
    for (int i = 0; i < 36; ++i) cc[i] = C1.a[i] + C2p.a[i] + (((m1.a[i] + m2.a[i]) + m3.a[i]) + m4.a[i]) / 12 + B.a[i] / 4;
}

int cloud_uct_associate_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int cov_off, int trace_off,
                            const double pose_global[7], const double cov_global[36], const double *ext_poses, const double *ext_covs,
                            int n_lidar, const double cov_meas[9], int with_ua, double trace_thr, void *out, int *n_out, int mem)
{
    if (!points || n <= 0 || stride < 12 || (stride & 3) || n_lidar <= 0 || n_lidar > 16 || !out || !n_out || !pose_global || !ext_poses)
        return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    if (with_ua && (!cov_global || !ext_covs || !cov_meas)) return fail(ctx, MLH_ERR_INVALID, "with_ua needs the pose / extrinsic / measurement covariances");
    hipStream_t st = ctx->stream;
    const unsigned char *src = static_cast<const unsigned char *>(points);
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, ctx->tmp.ensure(size_t(n) * stride));
        MLH_HIP(ctx, hipMemcpyAsync(ctx->tmp.p, points, size_t(n) * stride, hipMemcpyHostToDevice, st));
        src = ctx->tmp.as<unsigned char>();
    }
    // per-LiDAR constants: extrinsics (for pose_ext^-1), compound poses and their covariances
    std::vector<double> h(size_t(n_lidar) * (7 + 7 + 36), 0.0);
    double *h_ext = h.data(), *h_cp = h_ext + size_t(n_lidar) * 7, *h_cc = h_cp + size_t(n_lidar) * 7;
    for (int l = 0; l < n_lidar; ++l) {
        for (int i = 0; i < 7; ++i) h_ext[l * 7 + i] = ext_poses[l * 7 + i];
        if (with_ua) compound_pose_with_cov(pose_global, cov_global, ext_poses + l * 7, ext_covs + l * 36, h_cp + l * 7, h_cc + l * 36);
    }
    VoxBuf &V = ctx->vox;
    MLH_HIP(ctx, ctx->uct_buf.ensure(sizeof(double) * h.size() + 64));
    MLH_HIP(ctx, V.out.ensure(size_t(n) * stride));          // staged records
    MLH_HIP(ctx, V.leader.ensure(sizeof(int) * size_t(n + 1)));   // keep flags
    MLH_HIP(ctx, V.vox_of.ensure(sizeof(int) * size_t(n + 1)));   // output slots
    MLH_HIP(ctx, V.total.ensure(sizeof(int) * 2));
    double *d_c = ctx->uct_buf.as<double>();
    MLH_HIP(ctx, hipMemcpyAsync(d_c, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));                  // h is a local
    UctArgs A;
    A.src = src; A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.ext = d_c; A.ext_cov = nullptr; A.n_lidar = n_lidar;
    for (int i = 0; i < 9; ++i) A.meas[i] = cov_meas ? cov_meas[i] : 0.0;
    A.trace_thr = trace_thr; A.cov6 = nullptr; A.keep = V.leader.as<int>();
    A.upose = d_c + size_t(n_lidar) * 7; A.upose_cov = d_c + size_t(n_lidar) * 14; A.rec_out = V.out.as<unsigned char>(); A.n_dev = nullptr;
    for (int i = 0; i < 7; ++i) A.gpose[i] = pose_global[i];
    A.cov_off = cov_off; A.trace_off = trace_off; A.with_ua = with_ua ? 1 : 0;
    const int nb = (n + 255) / 256;
    MLH_LAUNCH(point_uncertainty_kernel, dim3(nb), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipMemcpyAsync(V.vox_of.p, V.leader.p, sizeof(int) * size_t(n), hipMemcpyDeviceToDevice, st));
    int rc = device_exclusive_scan(ctx, V.vox_of.as<int>(), n, V.sums, V.total.as<int>());
    if (rc) return rc;
    unsigned char *dst = static_cast<unsigned char *>(out);
    if (mem == MLH_MEM_HOST) {
        MLH_HIP(ctx, V.in.ensure(size_t(n) * stride));
        dst = V.in.as<unsigned char>();
    }
    MLH_LAUNCH(compact_records_kernel, dim3(nb), dim3(256), 0, st, (const unsigned char *)V.out.as<unsigned char>(), (const int *)V.leader.as<int>(),
                       (const int *)V.vox_of.as<int>(), n, stride, dst);
    MLH_HIP(ctx, hipGetLastError());
    int total = 0;
    MLH_HIP(ctx, read_back_int(ctx, V.total.p, &total));
    *n_out = total;
    if (mem == MLH_MEM_HOST && total > 0) {
        MLH_HIP(ctx, hipMemcpyAsync(out, dst, size_t(total) * stride, hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
    }
    return device_error_check(ctx);
}

// ---------------------------------------------------------------- downsampleCurrentScan (lidar_mapper_keyframe.cpp:356-421)
__global__ __launch_bounds__(256) void features_from_kept_kernel(const unsigned char *__restrict__ recs, int stride, int intensity_off, int n,
                                                                  const int *__restrict__ keep, const int *__restrict__ slot, const float *__restrict__ cov6,
                                                                  float4 *__restrict__ pts, float4 *__restrict__ covd, float *__restrict__ out11)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n || !keep[i]) return;
    const float *r = reinterpret_cast<const float *>(recs + size_t(i) * stride);
    const float inten = intensity_off >= 0 ? *reinterpret_cast<const float *>(recs + size_t(i) * stride + intensity_off) : 0.f;
    const float *c = cov6 + size_t(i) * 6;
    const int s = slot[i];
    pts[s] = make_float4(r[0], r[1], r[2], inten);
    covd[s] = make_float4(c[0], c[3], c[5], 0.f);
    if (out11) {
        float *o = out11 + size_t(s) * 11;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = inten;
#pragma unroll
        for (int k = 0; k < 6; ++k) o[4 + k] = c[k];
        o[10] = c[0] + c[3] + c[5];
    }
}

// voxel thinning (plain branch) -> per-point uncertainty -> trace gate -> the kind's feature set, all on the device.
// out11_dev: optional device buffer (n x 11 floats) that receives the kept PointXYZIWithCov records [x y z i cov6 trace].
int downsample_current_scan_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int mem, float leaf, const double *ext_poses,
                                const double *ext_covs, int n_lidar, const double cov_meas[9], int with_ua, double trace_thr, DevBuf &pts_out,
                                DevBuf &covd_out, float *out11_dev, int *n_out, const float *known_bounds)
{
    if (n_lidar <= 0 || n_lidar > 16 || !ext_poses || (with_ua && (!ext_covs || !cov_meas))) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    hipStream_t st = ctx->stream;
    VoxBuf &V = ctx->vox;
    if (n <= 0) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    // the extrinsics go first: the voxel filter's own synchronisations (bounds, count) then cover this upload as well, and the
    // sources (the caller's temporaries / `zero`) are free again when it returns
    MLH_HIP(ctx, ctx->uct_buf.ensure(sizeof(double) * size_t(n_lidar) * 43 + sizeof(float) * 6 * size_t(n) + 64));
    double *d_ext = ctx->uct_buf.as<double>();
    double *d_cov = d_ext + size_t(n_lidar) * 7;
    float *d_c6 = reinterpret_cast<float *>(d_cov + size_t(n_lidar) * 36);
    std::vector<double> zero(size_t(n_lidar) * 36, 0.0);
    MLH_HIP(ctx, hipMemcpyAsync(d_ext, ext_poses, sizeof(double) * 7 * n_lidar, hipMemcpyHostToDevice, st));
    MLH_HIP(ctx, hipMemcpyAsync(d_cov, ext_covs ? ext_covs : zero.data(), sizeof(double) * 36 * n_lidar, hipMemcpyHostToDevice, st));
    // the thinned records stay in V.out and their count on the device (V.total): everything downstream is launched over the
    // upper bound n and guarded by that count, so the only host round trip left after the bounds is the final feature count
    int n_ds = 0;
    int rc = voxel_filter_run(ctx, points, stride, n, intensity_off, -1, -1, leaf, 0.f, nullptr, &n_ds, mem, known_bounds, false);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    *n_out = 0;
    MLH_HIP(ctx, V.leader.ensure(sizeof(int) * size_t(n + 1)));
    MLH_HIP(ctx, V.vox_of.ensure(sizeof(int) * size_t(n + 1)));
    UctArgs A;
    A.src = V.out.as<unsigned char>(); A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.ext = d_ext; A.ext_cov = d_cov; A.n_lidar = n_lidar;
    for (int i = 0; i < 9; ++i) A.meas[i] = cov_meas ? cov_meas[i] : 0.0;
    A.trace_thr = trace_thr; A.cov6 = d_c6; A.keep = V.leader.as<int>();
    A.upose = d_ext; A.upose_cov = d_cov; A.rec_out = nullptr; A.cov_off = A.trace_off = -1; A.with_ua = with_ua ? 1 : 0;
    A.n_dev = V.total.as<int>();
    A.keep2 = V.vox_of.as<int>();
    for (int i = 0; i < 7; ++i) A.gpose[i] = 0.0;
    const int nb = (n + 255) / 256;
    MLH_LAUNCH(point_uncertainty_kernel, dim3(nb), dim3(256), 0, st, A);
    if ((rc = device_exclusive_scan(ctx, V.vox_of.as<int>(), n, V.sums, V.total.as<int>() + 1))) { (void)hipStreamSynchronize(st); return rc; }
    MLH_HIP(ctx, pts_out.ensure(sizeof(float4) * size_t(n)));
    MLH_HIP(ctx, covd_out.ensure(sizeof(float4) * size_t(n)));
    MLH_LAUNCH(features_from_kept_kernel, dim3(nb), dim3(256), 0, st, (const unsigned char *)V.out.as<unsigned char>(), stride, intensity_off, n,
                       (const int *)V.leader.as<int>(), (const int *)V.vox_of.as<int>(), (const float *)d_c6, pts_out.as<float4>(), covd_out.as<float4>(), out11_dev);
    MLH_HIP(ctx, hipGetLastError());
    int total = 0;
    MLH_HIP(ctx, read_back_int(ctx, V.total.as<int>() + 1, &total));
    *n_out = total;
    return MLH_OK;
}

// features_from_kept for the two-cloud pipeline: kept record i belongs to the first cloud when its (voxel) position i lies before
// first_records = wpre[first_word]; the kinds' feature sets are filled side by side and their counts left in counts[0..1]
__global__ __launch_bounds__(256) void features_from_kept_pair_kernel(const unsigned char *__restrict__ recs, int stride, int intensity_off, int n,
                                                                       const int *__restrict__ n_records, const int *__restrict__ wpre, int first_word,
                                                                       const int *__restrict__ keep, const int *__restrict__ slot, const int *__restrict__ kept_total,
                                                                       const float *__restrict__ cov6, float4 *__restrict__ pts0, float4 *__restrict__ covd0,
                                                                       float4 *__restrict__ pts1, float4 *__restrict__ covd1, int *__restrict__ counts,
                                                                       int *host_counts, unsigned long long *host_seq, unsigned long long seq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int first_records = wpre[first_word];                  // records (occupied voxels) of the first cloud
    const int total_records = *n_records;
    // kept records of the first cloud = exclusive scan of the keep flags at position first_records
    const int first_kept = first_records < total_records ? slot[first_records] : *kept_total;
    if (i == 0) {
        counts[0] = first_kept; counts[1] = *kept_total - first_kept;
        // the two counts are all the host waits for, and they are known before this launch has moved a single record: they go to pinned host memory from
        // here (no copy and no marker launch behind the kernel), and the host is back enqueueing the solver's launches while the records are still being moved
        if (host_seq) {
            host_counts[0] = first_kept; host_counts[1] = *kept_total - first_kept;
            __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (i >= n || i >= total_records || !keep[i]) return;
    const float *r = reinterpret_cast<const float *>(recs + size_t(i) * stride);
    const float inten = intensity_off >= 0 ? *reinterpret_cast<const float *>(recs + size_t(i) * stride + intensity_off) : 0.f;
    const float *c = cov6 + size_t(i) * 6;
    const bool second = i >= first_records;
    const int s = second ? slot[i] - first_kept : slot[i];
    (second ? pts1 : pts0)[s] = make_float4(r[0], r[1], r[2], inten);
    (second ? covd1 : covd0)[s] = make_float4(c[0], c[3], c[5], 0.f);
}

// ---------------------------------------------------------------- the frame's thinning, sort first (round 5)
// downsampleCurrentScan for both fused clouds used to be 25 launches: voxel -> occupancy bits -> popcount prefix (the output slot) -> counts -> prefix -> scatter,
// THEN the device std::sort on the slots for the member order, aggregate, uncertainty, prefix over the keep flags, features. The sort only looks at the ORDER of
// its keys, and a voxel's output slot is monotone in its voxel index: sorting the voxel indices themselves leaves the same permutation -- and leaves the points
// grouped by voxel, ascending, which is everything the slot machinery in front of it was computing. So: keys = voxel indices, made inside the sort's init launch;
// behind the sort a voxel is a run of equal keys. Three launches finish the job, every workgroup on 2048 consecutive sorted positions of ONE cloud:
//   vsp_heads_kernel      run heads per chunk;
//   vsp_aggregate_kernel  chunk offset = heads of the chunks before (summed here), local scan -> the head's output slot; the head's thread walks its run (the voxel's
//                         members in std::sort's order): centroid as vox_aggregate_kernel's plain branch, evalPointUncertainty of the centroid (eval_point_cov),
//                         trace gate -> record, cov, keep flag per slot; kept records per chunk;
//   vsp_features_kernel   kept-before-this-chunk (summed here), local scan of the keep flags -> the kinds' feature sets; the counts go to pinned host memory.
// 17 launches instead of 25, no atomics, no occupancy grid; bit for bit the former results (same sums in the same order).
#ifndef MLH_VSP_CH
#define MLH_VSP_CH 512
#endif
// positions per workgroup. 2048 (first version): a chunk of the corner cloud holds ~1 460 voxels (1.4 members each), six per thread one after the other, each with its
// evalPointUncertainty -- vsp_aggregate_kernel 22.5 us, vsp_features_kernel 11 us by rocprofv3 (profiles/r05_frame_kernel_stats.txt); 512: one or two voxels per thread
constexpr int VSP_CH = MLH_VSP_CH, VSP_TPB = 256, VSP_PER = VSP_CH / VSP_TPB;
struct VspArgs {
    const int *keys, *members;            // sorted voxel indices / point indices in std::sort's order
    const unsigned char *src0, *src1;
    int stride, intensity_off, n0, n, nch0, nch;
    int *heads, *kept;                    // per chunk
    float4 *rec;                          // per slot: centroid + the last member's intensity
    float *cov6;                          // per slot
    int *keep;                            // per slot
    double ext[4 * 7], ext_cov[4 * 36], meas[9];      // extrinsics as kernel arguments (n_lidar <= 4)
    int n_lidar, with_ua;
    double trace_thr;
    float4 *pts0, *covd0, *pts1, *covd1;
    int *counts;                          // device: [0..1] features per kind
    int *host_counts;
    unsigned long long *host_seq, seq;
};

__device__ __forceinline__ void vsp_chunk(const VspArgs &A, int c, int &lo, int &hi)
{
    if (c < A.nch0) { lo = c * VSP_CH; hi = min(lo + VSP_CH, A.n0); }
    else { lo = A.n0 + (c - A.nch0) * VSP_CH; hi = min(lo + VSP_CH, A.n); }
}
__device__ __forceinline__ bool vsp_head(const VspArgs &A, int i) { return i == 0 || i == A.n0 || A.keys[i] != A.keys[i - 1]; }

// block-wide exclusive scan of one int per thread (256 threads); total = the sum
__device__ __forceinline__ int vsp_block_scan(int v, int *lds4, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    if (lane == 63) lds4[wave] = incl;
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) if (w < wave) base += lds4[w];
    total = lds4[0] + lds4[1] + lds4[2] + lds4[3];
    __syncthreads();
    return base + incl - v;
}
// sum of a[0 .. upto) over the workgroup (every thread gets it) and, on the way, of a[0 .. upto2)
__device__ __forceinline__ void vsp_prefix2(const int *a, int upto, int upto2, int *lds8, int &s1, int &s2)
{
    int p1 = 0, p2 = 0;
    for (int j = threadIdx.x; j < max(upto, upto2); j += VSP_TPB) { const int v = a[j]; p1 += j < upto ? v : 0; p2 += j < upto2 ? v : 0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { p1 += __shfl_xor(p1, off); p2 += __shfl_xor(p2, off); }
    if ((threadIdx.x & 63) == 0) { lds8[threadIdx.x >> 6] = p1; lds8[4 + (threadIdx.x >> 6)] = p2; }
    __syncthreads();
    s1 = lds8[0] + lds8[1] + lds8[2] + lds8[3];
    s2 = lds8[4] + lds8[5] + lds8[6] + lds8[7];
    __syncthreads();
}

__global__ __launch_bounds__(VSP_TPB) void vsp_heads_kernel(VspArgs A)
{
    __shared__ int lds4[4];
    int lo, hi;
    vsp_chunk(A, blockIdx.x, lo, hi);
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < VSP_PER; ++u) { const int i = lo + threadIdx.x * VSP_PER + u; cnt += (i < hi && vsp_head(A, i)) ? 1 : 0; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) A.heads[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ __launch_bounds__(VSP_TPB) void vsp_aggregate_kernel(VspArgs A)
{
    __shared__ int lds4[4], lds8[8];
    __shared__ int s_head[VSP_CH + 1];                 // the chunk's run heads (positions), in order
    int lo, hi;
    vsp_chunk(A, blockIdx.x, lo, hi);
    const int cloud_end = blockIdx.x < A.nch0 ? A.n0 : A.n;
    int slot0, unused;
    vsp_prefix2(A.heads, blockIdx.x, 0, lds8, slot0, unused);
    unsigned flags = 0;
    int cnt = 0;
#pragma unroll
    for (int u = 0; u < VSP_PER; ++u) { const int i = lo + threadIdx.x * VSP_PER + u; if (i < hi && vsp_head(A, i)) { flags |= 1u << u; ++cnt; } }
    int h;
    int li = vsp_block_scan(cnt, lds4, h);
#pragma unroll
    for (int u = 0; u < VSP_PER; ++u) if (flags & (1u << u)) s_head[li++] = lo + threadIdx.x * VSP_PER + u;
    __syncthreads();
    // one voxel per thread and trip (the heads dealt round-robin: a thread that found eight heads in its eight positions does not walk eight voxels by itself);
    // a voxel = the positions from its head to the next head -- the last one of a chunk may run on into the next chunk
    int kept = 0;
    for (int ls = threadIdx.x; ls < h; ls += VSP_TPB) {
        const int b = s_head[ls];
        int e;
        if (ls + 1 < h) e = s_head[ls + 1];
        else { const int key = A.keys[b]; e = hi; while (e < cloud_end && A.keys[e] == key) ++e; }
        // the voxel's members, in the order std::sort left them: xyz mean, the LAST member's intensity (vox_aggregate_kernel's plain branch, term for term);
        // four member records in flight
        float mu0 = 0.f, mu1 = 0.f, mu2 = 0.f, ity = 0.f;
        for (int k = b; k < e; k += 4) {
            int id[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) id[u] = A.members[min(k + u, e - 1)];
            float q[4][3], it[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned char *rb = id[u] < A.n0 ? A.src0 + size_t(id[u]) * A.stride : A.src1 + size_t(id[u] - A.n0) * A.stride;
                const float *f = reinterpret_cast<const float *>(rb);
                q[u][0] = f[0]; q[u][1] = f[1]; q[u][2] = f[2];
                it[u] = A.intensity_off >= 0 ? *reinterpret_cast<const float *>(rb + A.intensity_off) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k + u >= e) break;
                mu0 += q[u][0]; mu1 += q[u][1]; mu2 += q[u][2];
                ity = it[u];
            }
        }
        const int m = e - b;
        const float fc = float(m > 0 ? m : 1);
        const float x = mu0 / fc, y = mu1 / fc, z = mu2 / fc;
        double cov[3][3];
        eval_point_cov(A.ext, A.ext, A.ext_cov, A.n_lidar, A.meas, A.with_ua, x, y, z, ity, cov);
        const double tr = cov[0][0] + cov[1][1] + cov[2][2];
        const int keep = (A.with_ua && A.trace_thr > 0.0 && tr > A.trace_thr) ? 0 : 1;
        const int slot = slot0 + ls;
        A.rec[slot] = make_float4(x, y, z, ity);
        float *o = A.cov6 + size_t(slot) * 6;
        o[0] = float(cov[0][0]); o[1] = float(cov[0][1]); o[2] = float(cov[0][2]); o[3] = float(cov[1][1]); o[4] = float(cov[1][2]); o[5] = float(cov[2][2]);
        A.keep[slot] = keep;
        kept += keep;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) kept += __shfl_xor(kept, off);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) A.kept[blockIdx.x] = lds4[0] + lds4[1] + lds4[2] + lds4[3];
}

__global__ __launch_bounds__(VSP_TPB) void vsp_features_kernel(VspArgs A)
{
    __shared__ int lds4[4], lds8[8];
    const int c = blockIdx.x;
    int slot0, first_records, kept0, first_kept, all_records, all_kept;
    vsp_prefix2(A.heads, c, A.nch0, lds8, slot0, first_records);
    vsp_prefix2(A.kept, c, A.nch0, lds8, kept0, first_kept);
    if (c == 0) {
        // the two counts are all the host waits for: published before a single record is moved
        vsp_prefix2(A.kept, A.nch, 0, lds8, all_kept, all_records);
        if (threadIdx.x == 0) {
            A.counts[0] = first_kept; A.counts[1] = all_kept - first_kept;
            if (A.host_seq) {
                A.host_counts[0] = first_kept; A.host_counts[1] = all_kept - first_kept;
                __hip_atomic_store(A.host_seq, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    const int h = A.heads[c];
    int flags = 0, cnt = 0;
#pragma unroll
    for (int u = 0; u < VSP_PER; ++u) { const int k = threadIdx.x * VSP_PER + u; if (k < h && A.keep[slot0 + k]) { flags |= 1 << u; ++cnt; } }
    int total;
    int dst = kept0 + vsp_block_scan(cnt, lds4, total);
    const bool second = c >= A.nch0;
    for (int u = 0; u < VSP_PER; ++u) {
        if (!(flags & (1 << u))) continue;
        const int s = slot0 + threadIdx.x * VSP_PER + u;
        const float4 r = A.rec[s];
        const float *cv = A.cov6 + size_t(s) * 6;
        const int d = second ? dst - first_kept : dst;
        (second ? A.pts1 : A.pts0)[d] = r;
        (second ? A.covd1 : A.covd0)[d] = make_float4(cv[0], cv[3], cv[5], 0.f);
        ++dst;
    }
}

// downsampleCurrentScan for the surf AND the corner cloud in one set of launches (device-resident clouds with known bounding boxes:
// the fused clouds). Same results as two downsample_current_scan_run calls.
int downsample_current_scan_pair_run(mlh_ctx *ctx, const void *surf, int n_surf, const float bounds_surf[6], float leaf_surf, const void *corner, int n_corner,
                                     const float bounds_corner[6], float leaf_corner, int stride, int intensity_off, const double *ext_poses, const double *ext_covs,
                                     int n_lidar, const double cov_meas[9], int with_ua, double trace_thr, int *n_surf_out, int *n_corner_out, bool defer)
{
    if (n_lidar <= 0 || n_lidar > 16 || !ext_poses || (with_ua && (!ext_covs || !cov_meas))) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    hipStream_t st = ctx->stream;
    VoxBuf &V = ctx->vox;
    const int n = n_surf + n_corner;
    static const bool old_pipeline = std::getenv("MLH_THIN_SLOTS_FIRST") != nullptr;       // (A/B runs: the round-4 pipeline)
    if (ctx->vox_member_order == 1 && n_lidar <= 4 && n_surf > 0 && n_corner > 0 && stride >= 12 && !(stride & 3) && leaf_surf > 0.f && leaf_corner > 0.f && !old_pipeline) {
        // ---- sort first (see above). The two grids' geometry as voxel_filter_run2 lays it out.
        VoxKeyGen G;
        const float *hb[2] = {bounds_surf, bounds_corner};
        const float inv[2] = {1.0f / leaf_surf, 1.0f / leaf_corner};
        int min_b[2][3], div_b[2][3];
        long long ncell[2];
        bool fits = true;
        for (int k = 0; k < 2; ++k) {
            long long ext[3];
            for (int d = 0; d < 3; ++d) {
                if (!std::isfinite(hb[k][d]) || !std::isfinite(hb[k][3 + d])) return fail(ctx, MLH_ERR_INVALID, "non-finite coordinates");
                ext[d] = (long long)((hb[k][3 + d] - hb[k][d]) * inv[k]) + 1;
                min_b[k][d] = int(std::floor(hb[k][d] * inv[k]));
                div_b[k][d] = int(std::floor(hb[k][3 + d] * inv[k])) - min_b[k][d] + 1;
            }
            if (ext[0] > 2147483647ll || ext[1] > 2147483647ll || ext[2] > 2147483647ll || ext[0] * ext[1] > 2147483647ll || ext[0] * ext[1] * ext[2] > 2147483647ll) fits = false;
            ncell[k] = (long long)div_b[k][0] * div_b[k][1] * div_b[k][2];
        }
        const long long off1 = ((ncell[0] + 31) / 32) * 32;
        if (fits && off1 + ncell[1] <= 2147483647ll - 64) {
            G.src0 = static_cast<const unsigned char *>(surf); G.src1 = static_cast<const unsigned char *>(corner); G.stride = stride; G.n0 = n_surf;
            G.inv_leaf0 = inv[0]; G.inv_leaf1 = inv[1];
            for (int d = 0; d < 3; ++d) { G.min_b0[d] = min_b[0][d]; G.min_b1[d] = min_b[1][d]; }
            G.mul1_0 = div_b[0][0]; G.mul2_0 = div_b[0][0] * div_b[0][1]; G.mul1_1 = div_b[1][0]; G.mul2_1 = div_b[1][0] * div_b[1][1]; G.cell_off1 = int(off1);
            VspArgs A;
            A.nch0 = (n_surf + VSP_CH - 1) / VSP_CH; A.nch = A.nch0 + (n_corner + VSP_CH - 1) / VSP_CH;
            MLH_HIP(ctx, V.members.ensure(sizeof(int) * size_t(n)));
            MLH_HIP(ctx, V.sums.ensure(sizeof(int) * size_t(2 * A.nch + 2)));
            MLH_HIP(ctx, V.out.ensure(sizeof(float4) * size_t(n)));
            MLH_HIP(ctx, V.leader.ensure(sizeof(int) * size_t(n + 1)));
            MLH_HIP(ctx, V.total.ensure(sizeof(int) * 4));
            MLH_HIP(ctx, ctx->uct_buf.ensure(sizeof(float) * 6 * size_t(n) + 64));
            FeatSet &f0 = ctx->feat[MLH_SURF], &f1 = ctx->feat[MLH_CORNER];
            MLH_HIP(ctx, f0.pts.ensure(sizeof(float4) * size_t(n_surf))); MLH_HIP(ctx, f0.covd.ensure(sizeof(float4) * size_t(n_surf)));
            MLH_HIP(ctx, f1.pts.ensure(sizeof(float4) * size_t(n_corner))); MLH_HIP(ctx, f1.covd.ensure(sizeof(float4) * size_t(n_corner)));
            int rc = device_std_sort_by_key(ctx, nullptr, n_surf, n, V.members.as<int>(), &G);
            if (rc) { (void)hipStreamSynchronize(st); return rc; }
            A.keys = device_std_sort_keys(ctx, n); A.members = V.members.as<int>();
            A.src0 = G.src0; A.src1 = G.src1; A.stride = stride; A.intensity_off = intensity_off; A.n0 = n_surf; A.n = n;
            A.heads = V.sums.as<int>(); A.kept = A.heads + A.nch + 1;
            A.rec = V.out.as<float4>(); A.cov6 = ctx->uct_buf.as<float>(); A.keep = V.leader.as<int>();
            for (int i = 0; i < 4 * 7; ++i) A.ext[i] = i < 7 * n_lidar ? ext_poses[i] : 0.0;
            for (int i = 0; i < 4 * 36; ++i) A.ext_cov[i] = (ext_covs && i < 36 * n_lidar) ? ext_covs[i] : 0.0;
            for (int i = 0; i < 9; ++i) A.meas[i] = cov_meas ? cov_meas[i] : 0.0;
            A.n_lidar = n_lidar; A.with_ua = with_ua ? 1 : 0; A.trace_thr = trace_thr;
            A.pts0 = f0.pts.as<float4>(); A.covd0 = f0.covd.as<float4>(); A.pts1 = f1.pts.as<float4>(); A.covd1 = f1.covd.as<float4>();
            A.counts = V.total.as<int>() + 2;
            A.host_counts = nullptr; A.host_seq = nullptr; A.seq = 0;
            if (int *hp = pinned_ints(ctx)) {
                A.host_counts = hp + 16;
                A.host_seq = reinterpret_cast<unsigned long long *>(hp + 32);
                if (ctx->counts_seq == 0) *A.host_seq = 0;
                A.seq = ++ctx->counts_seq;
            }
            MLH_LAUNCH(vsp_heads_kernel, dim3(A.nch), dim3(VSP_TPB), 0, st, A);
            MLH_LAUNCH(vsp_aggregate_kernel, dim3(A.nch), dim3(VSP_TPB), 0, st, A);
            MLH_LAUNCH(vsp_features_kernel, dim3(A.nch), dim3(VSP_TPB), 0, st, A);
            MLH_HIP(ctx, hipGetLastError());
            if (defer && A.host_seq) {
                // the caller enqueues the solve behind this without knowing the counts (mlh_downsample_scan2map): they are read when the pose is
                ctx->thin_counts_dev = A.counts; ctx->thin_counts_host = A.host_counts; ctx->thin_seq_host = A.host_seq; ctx->thin_seq = A.seq;
                *n_surf_out = -1; *n_corner_out = -1;
                return MLH_OK;
            }
            if (A.host_seq) {
                const auto t0 = std::chrono::steady_clock::now();
                unsigned spins = 0;
                while (__atomic_load_n(A.host_seq, __ATOMIC_ACQUIRE) != A.seq) {
                    if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                        MLH_HIP(ctx, hipStreamSynchronize(st));
                        if (__atomic_load_n(A.host_seq, __ATOMIC_ACQUIRE) != A.seq) return fail(ctx, MLH_ERR_HIP, "the thinned feature counts did not arrive");
                        break;
                    }
                    host_wait_relax(spins);
                }
                *n_surf_out = A.host_counts[0];
                *n_corner_out = A.host_counts[1];
            } else {
                int stack_counts[2] = {0, 0};
                MLH_HIP(ctx, hipMemcpyAsync(stack_counts, A.counts, sizeof(stack_counts), hipMemcpyDeviceToHost, st));
                MLH_HIP(ctx, hipStreamSynchronize(st));
                *n_surf_out = stack_counts[0];
                *n_corner_out = stack_counts[1];
            }
            return device_error_check(ctx);
        }
    }
    MLH_HIP(ctx, ctx->uct_buf.ensure(sizeof(double) * size_t(n_lidar) * 43 + sizeof(float) * 6 * size_t(n) + 64));
    double *d_ext = ctx->uct_buf.as<double>();
    double *d_cov = d_ext + size_t(n_lidar) * 7;
    float *d_c6 = reinterpret_cast<float *>(d_cov + size_t(n_lidar) * 36);
    std::vector<double> h_ec(size_t(n_lidar) * 43, 0.0);                 // extrinsics + covariances in one upload (the final wait covers it)
    for (int i = 0; i < 7 * n_lidar; ++i) h_ec[i] = ext_poses[i];
    if (ext_covs) for (int i = 0; i < 36 * n_lidar; ++i) h_ec[size_t(n_lidar) * 7 + i] = ext_covs[i];
    MLH_HIP(ctx, hipMemcpyAsync(d_ext, h_ec.data(), sizeof(double) * h_ec.size(), hipMemcpyHostToDevice, st));
    int first_word = 0;
    int rc = voxel_filter_run2(ctx, surf, n_surf, bounds_surf, leaf_surf, corner, n_corner, bounds_corner, leaf_corner, stride, intensity_off, &first_word);
    if (rc) { (void)hipStreamSynchronize(st); return rc; }
    MLH_HIP(ctx, V.leader.ensure(sizeof(int) * size_t(n + 1)));
    MLH_HIP(ctx, V.vox_of.ensure(sizeof(int) * size_t(n + 2)));
    MLH_HIP(ctx, V.total.ensure(sizeof(int) * 4));
    UctArgs A;
    A.src = V.out.as<unsigned char>(); A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.ext = d_ext; A.ext_cov = d_cov; A.n_lidar = n_lidar;
    for (int i = 0; i < 9; ++i) A.meas[i] = cov_meas ? cov_meas[i] : 0.0;
    A.trace_thr = trace_thr; A.cov6 = d_c6; A.keep = V.leader.as<int>();
    A.upose = d_ext; A.upose_cov = d_cov; A.rec_out = nullptr; A.cov_off = A.trace_off = -1; A.with_ua = with_ua ? 1 : 0;
    A.n_dev = V.total.as<int>();
    A.keep2 = V.vox_of.as<int>();
    for (int i = 0; i < 7; ++i) A.gpose[i] = 0.0;
    const int nb = (n + 255) / 256;
    MLH_LAUNCH(point_uncertainty_kernel, dim3(nb), dim3(256), 0, st, A);
    if ((rc = device_exclusive_scan(ctx, V.vox_of.as<int>(), n, V.sums, V.total.as<int>() + 1))) { (void)hipStreamSynchronize(st); return rc; }
    FeatSet &f0 = ctx->feat[MLH_SURF], &f1 = ctx->feat[MLH_CORNER];
    MLH_HIP(ctx, f0.pts.ensure(sizeof(float4) * size_t(n_surf))); MLH_HIP(ctx, f0.covd.ensure(sizeof(float4) * size_t(n_surf)));
    MLH_HIP(ctx, f1.pts.ensure(sizeof(float4) * size_t(n_corner))); MLH_HIP(ctx, f1.covd.ensure(sizeof(float4) * size_t(n_corner)));
    // pinned record for the counts: ints 16..17 and the 64-bit word at byte 128 of the context's pinned scratch block
    int *h_counts = nullptr;
    unsigned long long *h_seq = nullptr, seq = 0;
    if (int *hp = pinned_ints(ctx)) {
        h_counts = hp + 16;
        h_seq = reinterpret_cast<unsigned long long *>(hp + 32);
        if (ctx->counts_seq == 0) *h_seq = 0;
        seq = ++ctx->counts_seq;
    }
    MLH_LAUNCH(features_from_kept_pair_kernel, dim3(nb), dim3(256), 0, st, (const unsigned char *)V.out.as<unsigned char>(), stride, intensity_off, n,
                       (const int *)V.total.as<int>(), (const int *)V.wpre.as<int>(), first_word, (const int *)V.leader.as<int>(), (const int *)V.vox_of.as<int>(),
                       (const int *)(V.total.as<int>() + 1), (const float *)d_c6, f0.pts.as<float4>(), f0.covd.as<float4>(), f1.pts.as<float4>(), f1.covd.as<float4>(),
                       V.total.as<int>() + 2, h_counts, h_seq, seq);
    MLH_HIP(ctx, hipGetLastError());
    if (h_seq) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0x3ff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) {
                MLH_HIP(ctx, hipStreamSynchronize(st));
                if (__atomic_load_n(h_seq, __ATOMIC_ACQUIRE) != seq) return fail(ctx, MLH_ERR_HIP, "the thinned feature counts did not arrive");
                break;
            }
            host_wait_relax(spins);
        }
        *n_surf_out = h_counts[0];
        *n_corner_out = h_counts[1];
    } else {
        int stack_counts[2] = {0, 0};
        MLH_HIP(ctx, hipMemcpyAsync(stack_counts, V.total.as<int>() + 2, sizeof(stack_counts), hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        *n_surf_out = stack_counts[0];
        *n_corner_out = stack_counts[1];
    }
    return device_error_check(ctx);
}

}  // namespace mlh
