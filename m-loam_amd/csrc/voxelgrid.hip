// VoxelGridCovarianceMLOAM<PointT>::applyFilter on gfx950 -- the covariance-aware voxel thinning that produces the local map
// and the thinned scan features (mloam_pcl/include/mloam_pcl/voxel_grid_covariance_mloam_impl.hpp:68-457; call sites
// lidar_mapper_keyframe.cpp:343-347, 359-368).
//
// The reference sorts (voxel index, point index) pairs with std::sort and walks the runs. Here the voxel index (same f32
// arithmetic: floor(x * inv_leaf) - min_b, x fastest) sets one occupancy bit of the dense voxel grid; the number of set bits
// below a voxel's bit (a popcount prefix sum over the words, 1/32 of the cells) IS its output slot (= ascending voxel index, the
// reference's output order), a counting sort over the occupied slots groups the POINT INDICES, every member ranks itself
// inside its voxel, and one thread per voxel accumulates the members in index order (deterministic; the reference's order
// inside a voxel is whatever its unstable sort left, so sums agree up to f32 rounding). No comparison sort, no pass over an
// int-per-cell grid: a 0.2 m grid over a 120 x 120 x 20 m scan is 36 M cells = 4.5 MB of bits.
//   cov branch (:296-333)   w = thr - tr; |tr| >= thr dropped; mu = sum w p / sum w; cov = sum w^2 cov_i / (sum w)^2; intensity of the
//                           heaviest member (first wins); trace recomputed from the diagonal
//   plain branch (:392-420) xyz mean over the members, intensity of the last member
// Streaming kernels: ~ (stride + 8) B/point per pass + 1/8 B/cell (bits) + 8 B per 32 cells for the popcount scan.
#include "ctx.hpp"
#include "std_sort_mt.hpp"
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <system_error>
#include <thread>
#include <vector>

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

// popcount scan: out[w] = number of set bits in mask[0..w)
__global__ __launch_bounds__(256) void vscan_popc_local_kernel(const unsigned *__restrict__ mask, int *__restrict__ out, long long n, int *__restrict__ sums)
{
    __shared__ int lds[4];
    const long long base = (long long)blockIdx.x * VS_CHUNK + threadIdx.x * VS_ITEMS;
    int v[VS_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) { v[k] = (base + k < n) ? __popc(mask[base + k]) : 0; s += v[k]; }
    int total;
    int ex = vblock_scan(s, lds, total);
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// Two-launch variant for up to VS_DIRECT_BLOCKS chunks: chunk totals first, then every workgroup adds up the totals of the chunks
// before its own (a few hundred words) and scans its chunk -- the data is read twice and written once, and the single-workgroup
// middle launch disappears. POPC: the input is a bit mask and the scanned quantity its words' popcounts.
constexpr int VS_DIRECT_BLOCKS = 1024;
template <bool POPC>
__device__ __forceinline__ int vs_item(const int *in, long long i) { return POPC ? __popc(reinterpret_cast<const unsigned *>(in)[i]) : in[i]; }
template <bool POPC>
__global__ __launch_bounds__(256) void vscan_reduce_kernel(const int *__restrict__ in, long long n, int *__restrict__ sums)
{
    __shared__ int lds[4];
    const long long base = (long long)blockIdx.x * VS_CHUNK + threadIdx.x * VS_ITEMS;
    int s = 0;
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) s += (base + k < n) ? vs_item<POPC>(in, base + k) : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
}
template <bool POPC>
__global__ __launch_bounds__(256) void vscan_final_kernel(const int *in, int *out, long long n, const int *__restrict__ sums, int *__restrict__ grand_total)
{
    __shared__ int lds[4], lds_off[4];
    int acc = 0;
    for (int j = threadIdx.x; j < int(blockIdx.x); j += 256) acc += sums[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) lds_off[threadIdx.x >> 6] = acc;
    const long long base = (long long)blockIdx.x * VS_CHUNK + threadIdx.x * VS_ITEMS;
    int v[VS_ITEMS], s = 0;
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) { v[k] = (base + k < n) ? vs_item<POPC>(in, base + k) : 0; s += v[k]; }
    int total;
    int ex = vblock_scan(s, lds, total);            // (its barriers also publish lds_off)
    const int offset = lds_off[0] + lds_off[1] + lds_off[2] + lds_off[3];
    ex += offset;
#pragma unroll
    for (int k = 0; k < VS_ITEMS; ++k) { if (base + k < n) out[base + k] = ex; ex += v[k]; }
    if (grand_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *grand_total = offset + total;
}

template <bool POPC>
static int scan_launch(mlh_ctx *ctx, const int *in, int *out, long long n, DevBuf &sums, int *grand_total)
{
    const int nb = int((n + VS_CHUNK - 1) / VS_CHUNK);
    MLH_HIP(ctx, sums.ensure(sizeof(int) * size_t(nb + 1)));
    if (nb <= VS_DIRECT_BLOCKS) {
        MLH_LAUNCH(vscan_reduce_kernel<POPC>, dim3(nb), dim3(256), 0, ctx->stream, in, n, sums.as<int>());
        MLH_LAUNCH(vscan_final_kernel<POPC>, dim3(nb), dim3(256), 0, ctx->stream, in, out, n, (const int *)sums.as<int>(), grand_total);
    } else {
        if (POPC) MLH_LAUNCH(vscan_popc_local_kernel, dim3(nb), dim3(256), 0, ctx->stream, reinterpret_cast<const unsigned *>(in), out, n, sums.as<int>());
        else MLH_LAUNCH(vscan_local_kernel, dim3(nb), dim3(256), 0, ctx->stream, out, n, sums.as<int>());
        MLH_LAUNCH(vscan_sums_kernel, dim3(1), dim3(256), 0, ctx->stream, sums.as<int>(), nb, grand_total);
        MLH_LAUNCH(vscan_add_kernel, dim3(nb), dim3(256), 0, ctx->stream, out, n, (const int *)sums.as<int>());
    }
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int device_exclusive_scan(mlh_ctx *ctx, int *data, long long n, DevBuf &sums, int *grand_total)
{
    return scan_launch<false>(ctx, data, data, n, sums, grand_total);
}

struct VoxArgs {
    const unsigned char *src;
    int stride, n, intensity_off, cov_off, trace_off;
    float inv_leaf;
    int min_b[3], mul1, mul2;
    // optional second cloud thinned in the same launches (points n0 .. n-1 come from src1 and live in their own grid, whose voxels
    // are numbered behind the first grid's: output = the first cloud's voxels, then the second's). n0 = n: no second cloud.
    const unsigned char *src1;
    int n0, cell_off1;
    float inv_leaf1;
    int min_b1[3], mul1_1, mul2_1;
    float trace_thr;
    int centroid_all;   // plain branch as pcl::VoxelGrid<PointXYZI>: the intensity is averaged with the coordinates (CentroidPoint)
    int *vox_of;        // n: voxel index per point, then its output slot
    int *word_of;       // n: the point's occupancy word (kept so that the words can be cleared again)
    unsigned *mask;     // one occupancy bit per voxel of the dense grid
    const int *wpre;    // set bits before each mask word
    int *cnt;           // n + 2: members per output slot (shifted by one as in grid.hip) -> starts
    int *sorted_idx;    // n: point indices grouped by slot, arrival order
    int *members;       // n: the same, ascending inside every slot
    const int *total;   // occupied voxels = output records
    unsigned char *out; // n_out records (same layout as the input)
};

__device__ __forceinline__ const unsigned char *vrec_bytes(const VoxArgs &A, int i)
{
    return i < A.n0 ? A.src + size_t(i) * A.stride : A.src1 + size_t(i - A.n0) * A.stride;
}
__device__ __forceinline__ const float *vrec(const VoxArgs &A, int i) { return reinterpret_cast<const float *>(vrec_bytes(A, i)); }

__global__ __launch_bounds__(256) void vox_mark_kernel(VoxArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const float *p = vrec(A, i);
    int v;
    if (i < A.n0) {
        const int ijk0 = int(floorf(p[0] * A.inv_leaf) - float(A.min_b[0]));
        const int ijk1 = int(floorf(p[1] * A.inv_leaf) - float(A.min_b[1]));
        const int ijk2 = int(floorf(p[2] * A.inv_leaf) - float(A.min_b[2]));
        v = ijk0 + ijk1 * A.mul1 + ijk2 * A.mul2;
    } else {
        const int ijk0 = int(floorf(p[0] * A.inv_leaf1) - float(A.min_b1[0]));
        const int ijk1 = int(floorf(p[1] * A.inv_leaf1) - float(A.min_b1[1]));
        const int ijk2 = int(floorf(p[2] * A.inv_leaf1) - float(A.min_b1[2]));
        v = A.cell_off1 + ijk0 + ijk1 * A.mul1_1 + ijk2 * A.mul2_1;
    }
    A.vox_of[i] = v;
    // consecutive points of a scan line mostly share their voxel: one atomic per run of equal voxels inside the wavefront
    const int prev = __shfl_up(v, 1);
    if ((threadIdx.x & 63) == 0 || prev != v) atomicOr(&A.mask[v >> 5], 1u << (v & 31));
    A.cnt[i] = 0;
    if (i == 0) { A.cnt[A.n] = 0; A.cnt[A.n + 1] = 0; }
}

// output slot of a voxel = number of occupied voxels with a smaller index (ascending voxel index = the reference's output order)
__global__ __launch_bounds__(256) void vox_slot_kernel(VoxArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const int v = A.vox_of[i], w = v >> 5;
    const int s = A.wpre[w] + __popc(A.mask[w] & ((1u << (v & 31)) - 1u));
    A.vox_of[i] = s;
    A.word_of[i] = w;
    atomicAdd(&A.cnt[s + 1], 1);
}

__global__ __launch_bounds__(256) void vox_scatter_kernel(VoxArgs A)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= A.n) return;
    const int pos = atomicAdd(&A.cnt[A.vox_of[i] + 1], 1);
    A.sorted_idx[pos] = i;
    A.mask[A.word_of[i]] = 0u;          // every slot has been derived: leave the occupancy words clean for the next call
}

// after the scatter cnt[s] = start[s], cnt[s+1] = end[s]. Every member finds its rank among its voxel's members (independent
// loads over a short contiguous run) so that the aggregation can walk them in ascending point index without searching.
__global__ __launch_bounds__(256) void vox_rank_kernel(VoxArgs A)
{
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= A.n) return;
    const int id = A.sorted_idx[p];
    const int s = A.vox_of[id];
    const int b = A.cnt[s], e = A.cnt[s + 1];
    int rank = 0;
    for (int u = b; u < e; ++u) rank += (A.sorted_idx[u] < id) ? 1 : 0;
    A.members[b + rank] = id;
}

constexpr int VAG = 4;   // member records in flight per voxel
__global__ __launch_bounds__(256) void vox_aggregate_kernel(VoxArgs A)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= *A.total) return;
    const int b = A.cnt[s], e = A.cnt[s + 1];
    float mu[3] = {0.f, 0.f, 0.f}, ity = 0.f, cov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, weight_total = 0.f, w_max = 0.f;
    int cnt = 0;
    const bool has_cov = A.cov_off >= 0, has_i = A.intensity_off >= 0;
    for (int k = b; k < e; k += VAG) {
        int id[VAG];
#pragma unroll
        for (int u = 0; u < VAG; ++u) id[u] = A.members[min(k + u, e - 1)];
        float q[VAG][3], inten[VAG], c[VAG][6];
#pragma unroll
        for (int u = 0; u < VAG; ++u) {
            const unsigned char *rec = vrec_bytes(A, id[u]);
            const float *f = reinterpret_cast<const float *>(rec);
            q[u][0] = f[0]; q[u][1] = f[1]; q[u][2] = f[2];
            inten[u] = has_i ? *reinterpret_cast<const float *>(rec + A.intensity_off) : 0.f;
            if (has_cov) {
                const float *cc = reinterpret_cast<const float *>(rec + A.cov_off);
#pragma unroll
                for (int j = 0; j < 6; ++j) c[u][j] = cc[j];
            }
        }
#pragma unroll
        for (int u = 0; u < VAG; ++u) {
            if (k + u >= e) break;
            if (has_cov) {
                const float tr = c[u][0] + c[u][3] + c[u][5];
                if (fabsf(tr) >= A.trace_thr) continue;
                const float w = A.trace_thr - tr;
                mu[0] += w * q[u][0]; mu[1] += w * q[u][1]; mu[2] += w * q[u][2];
                ity = w > w_max ? inten[u] : ity;
                w_max = w > w_max ? w : w_max;
                const float w2 = w * w;
#pragma unroll
                for (int j = 0; j < 6; ++j) cov[j] += w2 * c[u][j];
                weight_total += w;
            } else {
                mu[0] += q[u][0]; mu[1] += q[u][1]; mu[2] += q[u][2];
                ity = A.centroid_all ? ity + inten[u] : inten[u];     // averaged (pcl::VoxelGrid) / the last member's (VoxelGridCovarianceMLOAM)
                ++cnt;
            }
        }
    }
    unsigned char *o = A.out + size_t(s) * A.stride;
    for (int j = 0; j < A.stride / 4; ++j) reinterpret_cast<float *>(o)[j] = 0.f;
    float *ox = reinterpret_cast<float *>(o);
    if (has_cov) {
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
        if (A.centroid_all) ity = ity / fc;
    }
    if (A.stride >= 16 && A.intensity_off != 12 && A.cov_off != 12 && A.trace_off != 12) ox[3] = 1.0f;   // PCL_ADD_POINT4D padding
    if (has_i) *reinterpret_cast<float *>(o + A.intensity_off) = ity;
}

// Reference member order: the reference groups a voxel's members with an UNSTABLE std::sort whose comparator sees the voxel index only
// (voxel_grid_covariance_mloam_impl.hpp:215-236), so the order of the members inside a voxel -- and with it "the last member's
// intensity" of the plain branch, the first-maximum-weight intensity of the covariance branch and the association of every f32 sum -- is
// whatever libstdc++'s introsort leaves. That order cannot be derived without running the same algorithm on the same sequence (the
// points' output slots: ascending voxel index, order-isomorphic to PCL's idx, hence the same comparisons and the same introsort path).
//   mode 1 (default): stdsort.hip runs libstdc++'s algorithm, restated data-parallel, on the device -- nothing leaves HBM;
//   mode 2: the slots come back into pinned memory, the platform's own std::sort runs there per cloud on (slot, point index) pairs in
//           point order (std_sort_mt.hpp: libstdc++'s own partition steps, the recursion's independent halves on other threads; the
//           second cloud of a pair call on threads of its own), the member lists go back: a host round trip and ~1.2 ms for a frame's
//           78 k points, for a standard library whose std::sort is not the algorithm mode 1 restates;
//   mode 0: members in ascending point index (vox_rank_kernel): same voxels, same centroids to f32 rounding, a different surviving
//           intensity where a voxel mixes them.
// members[lo..hi) <- the point indices lo..hi-1 in the order std::sort leaves them when it compares slot[] only
void host_std_sort_permutation(const int *slot, int lo, int hi, int *members)
{
    if (hi <= lo) return;
    struct IdxPt {
        unsigned int idx, cloud_point_index;
        bool operator<(const IdxPt &o) const { return idx < o.idx; }      // cloud_point_index_idx::operator< (voxel_grid.h): idx only
    };
    std::vector<IdxPt> iv;
    iv.reserve(size_t(hi - lo));
    for (int i = lo; i < hi; ++i) iv.push_back(IdxPt{(unsigned)slot[size_t(i)], (unsigned)i});
    std_sort_mt(iv.data(), iv.data() + iv.size(), hi - lo > 8192 ? 2 : 0);   // std::sort's result, its independent halves on up to 4 threads
    int *dst = members + lo;
    for (const IdxPt &e : iv) *dst++ = int(e.cloud_point_index);
}

static int members_in_std_sort_order(mlh_ctx *ctx, const VoxArgs &A)
{
    hipStream_t st = ctx->stream;
    const size_t n = size_t(A.n), need = 2 * sizeof(int) * n;
    if (need > ctx->vox_order_host_cap) {
        MLH_HIP(ctx, hipStreamSynchronize(st));                           // an earlier call's upload may still be reading the old block
        if (ctx->vox_order_host) (void)hipHostFree(ctx->vox_order_host);
        ctx->vox_order_host = nullptr; ctx->vox_order_host_cap = 0;
        MLH_HIP(ctx, hipHostMalloc(&ctx->vox_order_host, need + need / 4, hipHostMallocDefault));
        ctx->vox_order_host_cap = need + need / 4;
    }
    int *slot = static_cast<int *>(ctx->vox_order_host), *members = slot + n;
    MLH_HIP(ctx, hipMemcpyAsync(slot, A.vox_of, sizeof(int) * n, hipMemcpyDeviceToHost, st));
    MLH_HIP(ctx, hipStreamSynchronize(st));
    // one filter call per cloud in the reference: one sort per cloud
    auto sort_range = [slot, members](int lo, int hi) { host_std_sort_permutation(slot, lo, hi, members); };
    bool split = false;
    if (A.n0 > 4096 && A.n - A.n0 > 4096) {
        try {
            std::thread second(sort_range, A.n0, A.n);
            split = true;
            sort_range(0, A.n0);
            second.join();
        } catch (const std::system_error &) {}                           // no thread to be had: one after the other
    }
    if (!split) {
        sort_range(0, A.n0);
        sort_range(A.n0, A.n);
    }
    MLH_HIP(ctx, hipMemcpyAsync(A.members, members, sizeof(int) * n, hipMemcpyHostToDevice, st));
    return MLH_OK;                                                        // the staging block is the context's: the next call on it syncs the stream first
}

// getMinMax3D: per-workgroup partial bounds (6 floats each); the host, which needs them for the grid extents anyway, folds them
constexpr int VB_BLOCKS = 128;
__global__ __launch_bounds__(256) void vbounds_kernel(const unsigned char *src, int stride, int n, float *partial)
{
    __shared__ float lds[4][6];
    float m[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float *p = reinterpret_cast<const float *>(src + size_t(i) * stride);
#pragma unroll
        for (int d = 0; d < 3; ++d) { m[d] = fminf(m[d], p[d]); m[3 + d] = fmaxf(m[3 + d], p[d]); }
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
        partial[blockIdx.x * 6 + d] = r;
    }
}

// known_bounds: the exact min/max of the points if the caller already has them (skips the bounds pass and its round trip).
// sync_total = false (only with out_host == nullptr): the record count stays on the device (V.total[0]); *n_out is left at -1.
int voxel_filter_run(mlh_ctx *ctx, const void *points, int stride, int n, int intensity_off, int cov_off, int trace_off, float leaf,
                     float trace_thr, void *out_host, int *n_out, int mem, const float *known_bounds, bool sync_total, bool centroid_all)
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
    float hb[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
    if (known_bounds) {
        for (int d = 0; d < 6; ++d) hb[d] = known_bounds[d];
    } else {
        const int vb = std::min((n + 255) / 256, VB_BLOCKS);
        MLH_HIP(ctx, V.bounds.ensure(sizeof(float) * 6 * VB_BLOCKS));
        MLH_LAUNCH(vbounds_kernel, dim3(vb), dim3(256), 0, st, src, stride, n, V.bounds.as<float>());
        float hp[6 * VB_BLOCKS];
        MLH_HIP(ctx, hipMemcpyAsync(hp, V.bounds.p, sizeof(float) * 6 * size_t(vb), hipMemcpyDeviceToHost, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        for (int b = 0; b < vb; ++b)
            for (int d = 0; d < 3; ++d) { hb[d] = std::fmin(hb[d], hp[b * 6 + d]); hb[3 + d] = std::fmax(hb[3 + d], hp[b * 6 + 3 + d]); }
    }
    MLH_HIP(ctx, V.total.ensure(sizeof(int) * 2));
    if (!sync_total && out_host) return fail(ctx, MLH_ERR_INVALID, "sync_total = false needs the device-resident result");
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
        if (!out_host) {
            MLH_HIP(ctx, V.out.ensure(size_t(n) * stride));
            MLH_HIP(ctx, hipMemcpyAsync(V.out.p, src, size_t(n) * stride, hipMemcpyDeviceToDevice, st));
            MLH_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(V.total.p), n, 1, st));
        }
        else MLH_HIP(ctx, hipMemcpyAsync(out_host, src, size_t(n) * stride, mem == MLH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
        *n_out = n;
        return MLH_OK;
    }
    // The dense voxel grid exists only as one occupancy BIT per voxel (ncell / 8 bytes): a point's output slot is the number of set
    // bits below its voxel's (popcount prefix per 32-voxel word + popcount inside the word), and the counting sort that groups the
    // members runs over the at most n occupied slots instead of the ncell cells.
    const long long ncell = (long long)div_b[0] * div_b[1] * div_b[2];
    const long long nwords = (ncell + 31) / 32 + 1;
    if (sizeof(unsigned) * size_t(nwords) > V.cell.cap) {      // a fresh allocation is cleared once; afterwards every call leaves the words it set at zero
        MLH_HIP(ctx, V.cell.ensure(sizeof(unsigned) * size_t(nwords)));
        MLH_HIP(ctx, hipMemsetAsync(V.cell.p, 0, V.cell.cap, st));
    }
    MLH_HIP(ctx, V.word_of.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.wpre.ensure(sizeof(int) * size_t(nwords)));
    MLH_HIP(ctx, V.cnt.ensure(sizeof(int) * size_t(n + 2)));
    MLH_HIP(ctx, V.vox_of.ensure(sizeof(int) * size_t(n + 1)));
    MLH_HIP(ctx, V.sorted_idx.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.members.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.out.ensure(size_t(n) * stride));
    MLH_HIP(ctx, V.total.ensure(sizeof(int) * 2));
    const int nbw = int((nwords + VS_CHUNK - 1) / VS_CHUNK);
    MLH_HIP(ctx, V.sums.ensure(sizeof(int) * size_t(std::max(nbw, (n + VS_CHUNK - 1) / VS_CHUNK) + 1)));
    VoxArgs A;
    A.src = src; A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.cov_off = cov_off; A.trace_off = trace_off;
    A.src1 = nullptr; A.n0 = n; A.cell_off1 = 0; A.inv_leaf1 = 0.f; A.min_b1[0] = A.min_b1[1] = A.min_b1[2] = 0; A.mul1_1 = A.mul2_1 = 0;
    A.inv_leaf = inv; A.min_b[0] = min_b[0]; A.min_b[1] = min_b[1]; A.min_b[2] = min_b[2];
    A.mul1 = div_b[0]; A.mul2 = div_b[0] * div_b[1];
    A.trace_thr = trace_thr; A.centroid_all = centroid_all ? 1 : 0; A.vox_of = V.vox_of.as<int>(); A.word_of = V.word_of.as<int>(); A.mask = V.cell.as<unsigned>(); A.wpre = V.wpre.as<int>(); A.cnt = V.cnt.as<int>();
    A.sorted_idx = V.sorted_idx.as<int>(); A.members = V.members.as<int>(); A.total = V.total.as<int>(); A.out = V.out.as<unsigned char>();
    const int nbp = (n + 255) / 256;
    MLH_LAUNCH(vox_mark_kernel, dim3(nbp), dim3(256), 0, st, A);
    int rc = scan_launch<true>(ctx, reinterpret_cast<const int *>(A.mask), V.wpre.as<int>(), nwords, V.sums, V.total.as<int>());   // total = occupied voxels
    if (rc) return rc;
    MLH_LAUNCH(vox_slot_kernel, dim3(nbp), dim3(256), 0, st, A);
    rc = device_exclusive_scan(ctx, A.cnt + 1, n, V.sums, nullptr);         // cnt[s+1] <- start[s]
    if (rc) return rc;
    MLH_LAUNCH(vox_scatter_kernel, dim3(nbp), dim3(256), 0, st, A);    // cnt[s+1] <- start[s+1]
    if (ctx->vox_member_order == 1) { if ((rc = device_std_sort_by_key(ctx, A.vox_of, A.n0, A.n, A.members))) return rc; }
    else if (ctx->vox_member_order == 2) { if ((rc = members_in_std_sort_order(ctx, A))) return rc; }
    else MLH_LAUNCH(vox_rank_kernel, dim3(nbp), dim3(256), 0, st, A);
    MLH_LAUNCH(vox_aggregate_kernel, dim3(nbp), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    if (!sync_total) { *n_out = -1; return MLH_OK; }
    int total = 0;
    MLH_HIP(ctx, read_back_int(ctx, V.total.p, &total));
    *n_out = total;
    if (total > 0 && out_host) {
        MLH_HIP(ctx, hipMemcpyAsync(out_host, V.out.p, size_t(total) * stride, mem == MLH_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, st));
        MLH_HIP(ctx, hipStreamSynchronize(st));
    }
    return device_error_check(ctx);
}

// Two device-resident clouds of the same record layout thinned (plain branch) in ONE set of launches: the second cloud's voxels are
// numbered behind the first grid's, so the output holds the first cloud's centroids, then the second's, each in ascending voxel
// index -- exactly what two separate calls produce, at half the launches (these pipelines are dispatch-bound). bounds*: the exact
// min / max of each cloud. On return (nothing waited for): V.out = the records, V.total[0] = their number, *first_voxels_word =
// the word of V.wpre that holds the number of records of the first cloud. Returns MLH_ERR_UNSUPPORTED when a grid is too large
// for the shared index space (the caller then thins the clouds one by one).
int voxel_filter_run2(mlh_ctx *ctx, const void *src0, int n0, const float bounds0[6], float leaf0, const void *src1, int n1, const float bounds1[6],
                      float leaf1, int stride, int intensity_off, int *first_voxels_word)
{
    if (!src0 || !src1 || n0 <= 0 || n1 <= 0 || stride < 12 || (stride & 3) || !(leaf0 > 0.f) || !(leaf1 > 0.f)) return fail(ctx, MLH_ERR_INVALID, "bad arguments");
    hipStream_t st = ctx->stream;
    VoxBuf &V = ctx->vox;
    const float *hb[2] = {bounds0, bounds1};
    const float inv[2] = {1.0f / leaf0, 1.0f / leaf1};
    int min_b[2][3], div_b[2][3];
    long long ncell[2];
    for (int k = 0; k < 2; ++k) {
        long long ext[3];
        for (int d = 0; d < 3; ++d) {
            if (!std::isfinite(hb[k][d]) || !std::isfinite(hb[k][3 + d])) return fail(ctx, MLH_ERR_INVALID, "non-finite coordinates");
            ext[d] = (long long)((hb[k][3 + d] - hb[k][d]) * inv[k]) + 1;
            min_b[k][d] = int(std::floor(hb[k][d] * inv[k]));
            div_b[k][d] = int(std::floor(hb[k][3 + d] * inv[k])) - min_b[k][d] + 1;
        }
        if (ext[0] > 2147483647ll || ext[1] > 2147483647ll || ext[2] > 2147483647ll || ext[0] * ext[1] > 2147483647ll || ext[0] * ext[1] * ext[2] > 2147483647ll)
            return MLH_ERR_UNSUPPORTED;
        ncell[k] = (long long)div_b[k][0] * div_b[k][1] * div_b[k][2];
    }
    const long long off1 = ((ncell[0] + 31) / 32) * 32;                   // the second grid starts on a word boundary
    if (off1 + ncell[1] > 2147483647ll - 64) return MLH_ERR_UNSUPPORTED;
    const int n = n0 + n1;
    const long long nwords = (off1 + ncell[1] + 31) / 32 + 1;
    if (sizeof(unsigned) * size_t(nwords) > V.cell.cap) {
        MLH_HIP(ctx, V.cell.ensure(sizeof(unsigned) * size_t(nwords)));
        MLH_HIP(ctx, hipMemsetAsync(V.cell.p, 0, V.cell.cap, st));
    }
    MLH_HIP(ctx, V.word_of.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.wpre.ensure(sizeof(int) * size_t(nwords)));
    MLH_HIP(ctx, V.cnt.ensure(sizeof(int) * size_t(n + 2)));
    MLH_HIP(ctx, V.vox_of.ensure(sizeof(int) * size_t(n + 1)));
    MLH_HIP(ctx, V.sorted_idx.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.members.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, V.out.ensure(size_t(n) * stride));
    MLH_HIP(ctx, V.total.ensure(sizeof(int) * 4));
    const int nbw = int((nwords + VS_CHUNK - 1) / VS_CHUNK);
    MLH_HIP(ctx, V.sums.ensure(sizeof(int) * size_t(std::max(nbw, (n + VS_CHUNK - 1) / VS_CHUNK) + 1)));
    VoxArgs A;
    A.src = static_cast<const unsigned char *>(src0); A.stride = stride; A.n = n; A.intensity_off = intensity_off; A.cov_off = -1; A.trace_off = -1;
    A.inv_leaf = inv[0]; A.min_b[0] = min_b[0][0]; A.min_b[1] = min_b[0][1]; A.min_b[2] = min_b[0][2];
    A.mul1 = div_b[0][0]; A.mul2 = div_b[0][0] * div_b[0][1];
    A.src1 = static_cast<const unsigned char *>(src1); A.n0 = n0; A.cell_off1 = int(off1); A.inv_leaf1 = inv[1];
    A.min_b1[0] = min_b[1][0]; A.min_b1[1] = min_b[1][1]; A.min_b1[2] = min_b[1][2];
    A.mul1_1 = div_b[1][0]; A.mul2_1 = div_b[1][0] * div_b[1][1];
    A.trace_thr = 0.f; A.centroid_all = 0; A.vox_of = V.vox_of.as<int>(); A.word_of = V.word_of.as<int>(); A.mask = V.cell.as<unsigned>(); A.wpre = V.wpre.as<int>(); A.cnt = V.cnt.as<int>();
    A.sorted_idx = V.sorted_idx.as<int>(); A.members = V.members.as<int>(); A.total = V.total.as<int>(); A.out = V.out.as<unsigned char>();
    const int nbp = (n + 255) / 256;
    MLH_LAUNCH(vox_mark_kernel, dim3(nbp), dim3(256), 0, st, A);
    int rc = scan_launch<true>(ctx, reinterpret_cast<const int *>(A.mask), V.wpre.as<int>(), nwords, V.sums, V.total.as<int>());
    if (rc) return rc;
    MLH_LAUNCH(vox_slot_kernel, dim3(nbp), dim3(256), 0, st, A);
    rc = device_exclusive_scan(ctx, A.cnt + 1, n, V.sums, nullptr);
    if (rc) return rc;
    MLH_LAUNCH(vox_scatter_kernel, dim3(nbp), dim3(256), 0, st, A);
    if (ctx->vox_member_order == 1) { if ((rc = device_std_sort_by_key(ctx, A.vox_of, A.n0, A.n, A.members))) return rc; }
    else if (ctx->vox_member_order == 2) { if ((rc = members_in_std_sort_order(ctx, A))) return rc; }
    else MLH_LAUNCH(vox_rank_kernel, dim3(nbp), dim3(256), 0, st, A);
    MLH_LAUNCH(vox_aggregate_kernel, dim3(nbp), dim3(256), 0, st, A);
    MLH_HIP(ctx, hipGetLastError());
    *first_voxels_word = int(off1 >> 5);
    return MLH_OK;
}

}  // namespace mlh
