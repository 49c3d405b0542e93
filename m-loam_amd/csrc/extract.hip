// LOAM feature extraction on gfx950: FeatureExtract::extractCloud
// (estimator/src/featureExtract/feature_extract.cpp:118-297).
//
//   curvature_kernel  cpp:133-142  one thread per point, neighbours staged in LDS (21 B/point algorithmic HBM traffic);
//                     f32 sum in the reference's exact left-to-right order, no FMA (-ffp-contract=off)
//   label_kernel      cpp:152-265  one workgroup per ring: ring xyz + curvature + picked flags resident in LDS, all six
//                     sectors sorted at once with a bitonic network on 64-bit (curvature, index) keys, then wave 0 runs the
//                     greedy edge / flat walks: 64 sorted candidates are tested per step with a ballot, the first eligible
//                     one is taken, its +-5 neighbour suppression is evaluated by 10 lanes with a ballot prefix. The walk is
//                     sequential over sectors because suppression marks cross sector boundaries (cpp:201, 212).
//   offsets_kernel    exclusive scan of the per-ring list sizes (emission order = ring asc, sector asc, pick order)
//   emit_kernel       writes the four index lists; less-flat = positions with label <= 0 (cpp:258-264), stream-compacted.
#include "ctx.hpp"
#include <algorithm>

namespace mlh {

__global__ __launch_bounds__(256) void curvature_kernel(const float4 *__restrict__ pts, int n, float *__restrict__ curv,
                                                        int *__restrict__ label, int *__restrict__ picked)
{
    __shared__ float sx[256 + 10], sy[256 + 10], sz[256 + 10];
    const int base = blockIdx.x * 256;
    for (int t = threadIdx.x; t < 266; t += 256) {
        int gi = base - 5 + t;
        float4 p = (gi >= 0 && gi < n) ? pts[gi] : make_float4(0.f, 0.f, 0.f, 0.f);
        sx[t] = p.x; sy[t] = p.y; sz[t] = p.z;
    }
    __syncthreads();
    const int i = base + threadIdx.x;
    if (i >= n) return;
    float c = 0.f;
    if (i >= 5 && i < n - 5) {
        const int t = threadIdx.x + 5;
        float dx = sx[t - 5] + sx[t - 4] + sx[t - 3] + sx[t - 2] + sx[t - 1] - 10 * sx[t] + sx[t + 1] + sx[t + 2] + sx[t + 3] + sx[t + 4] + sx[t + 5];
        float dy = sy[t - 5] + sy[t - 4] + sy[t - 3] + sy[t - 2] + sy[t - 1] - 10 * sy[t] + sy[t + 1] + sy[t + 2] + sy[t + 3] + sy[t + 4] + sy[t + 5];
        float dz = sz[t - 5] + sz[t - 4] + sz[t - 3] + sz[t - 2] + sz[t - 1] - 10 * sz[t] + sz[t + 1] + sz[t + 2] + sz[t + 3] + sz[t + 4] + sz[t + 5];
        c = dx * dx + dy * dy + dz * dz;
    }
    curv[i] = c;
    label[i] = 0;
    picked[i] = 0;
}

constexpr int STAGE_SHARP = 12, STAGE_LESS = 120, STAGE_FLAT = 24;
constexpr int STAGE_STRIDE = STAGE_SHARP + STAGE_LESS + STAGE_FLAT;

struct LabelArgs {
    const float4 *pts;
    const float *curv;
    const int *start, *end;
    int *label, *picked;
    int *stage;         // [ring][STAGE_STRIDE]
    int *ring_counts;   // [ring][4]
    int n, max_span, sort_p;
};

// gap test of the suppression loops (cpp:192-213): squared distance between consecutive points > 0.05 (double literal)
__device__ __forceinline__ bool gap_exceeds(const float *sx, const float *sy, const float *sz, int a, int b)
{
    float dx = sx[a] - sx[b], dy = sy[a] - sy[b], dz = sz[a] - sz[b];
    return double(dx * dx + dy * dy + dz * dz) > 0.05;
}

// executed by all 64 lanes of wave 0; li = local index of the picked point
__device__ __forceinline__ void suppress_neighbours(const float *sx, const float *sy, const float *sz, int *spicked, int li, int lane)
{
    bool gap = false;
    if (lane < 5) gap = gap_exceeds(sx, sy, sz, li + lane + 1, li + lane);                // l = lane+1 : p[ind+l] - p[ind+l-1]
    else if (lane >= 8 && lane < 13) gap = gap_exceeds(sx, sy, sz, li - (lane - 8) - 1, li - (lane - 8));   // l = -(lane-8)-1
    const unsigned long long m = __ballot(gap);
    const unsigned fwd = unsigned(m & 0x1full), bwd = unsigned((m >> 8) & 0x1full);
    const int nf = fwd ? (__ffs(fwd) - 1) : 5;   // number of forward neighbours marked before the first gap
    const int nb = bwd ? (__ffs(bwd) - 1) : 5;
    if (lane < nf) spicked[li + lane + 1] = 1;
    if (lane >= 8 && (lane - 8) < nb) spicked[li - (lane - 8) - 1] = 1;
}

__global__ __launch_bounds__(256) void label_kernel(LabelArgs A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int ring = blockIdx.x;
    const int s = A.start[ring], e = A.end[ring];
    int *rc = A.ring_counts + ring * 4;
    if (e - s < 6 || s < 5 || e + 5 > A.n) {   // cpp:155 (and a guard against ring tables that are not inset)
        if (threadIdx.x < 4) rc[threadIdx.x] = 0;
        return;
    }
    const int span = e - s + 11;           // [s-5, e+5]
    const int g0 = s - 5;                  // global index of local 0
    const int P = A.sort_p;
    float *sx = reinterpret_cast<float *>(smem);
    float *sy = sx + A.max_span;
    float *sz = sy + A.max_span;
    float *sc = sz + A.max_span;
    int *spicked = reinterpret_cast<int *>(sc + A.max_span);
    int *slabel = spicked + A.max_span;
    unsigned long long *keys = reinterpret_cast<unsigned long long *>(slabel + A.max_span + ((A.max_span & 1) ? 1 : 0));
    __shared__ int s_stage[STAGE_STRIDE];
    __shared__ int s_cnt[4];

    for (int t = threadIdx.x; t < span; t += 256) {
        float4 p = A.pts[g0 + t];
        sx[t] = p.x; sy[t] = p.y; sz[t] = p.z;
        sc[t] = A.curv[g0 + t];
        spicked[t] = 0;
        slabel[t] = 0;
    }
    if (threadIdx.x < 4) s_cnt[threadIdx.x] = 0;
    // sector bounds (cpp:160-161), local coordinates (local = global - g0)
    int sp[6], ep[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        sp[j] = s + (e - s) * j / 6 - g0;
        ep[j] = s + (e - s) * (j + 1) / 6 - 1 - g0;
    }
    __syncthreads();
    // keys: (curvature bits << 32) | local index, padded with all-ones
    for (int t = threadIdx.x; t < 6 * P; t += 256) {
        const int j = t / P, k = t - j * P;
        const int len = ep[j] - sp[j] + 1;
        unsigned long long key = ~0ull;
        if (k < len) {
            const int li = sp[j] + k;
            key = ((unsigned long long)__float_as_uint(sc[li]) << 32) | (unsigned)li;
        }
        keys[t] = key;
    }
    __syncthreads();
    // bitonic sort, ascending, all sectors at once (segments of P are independent because every stride divides P)
    const int half = 3 * P;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += 256) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // index with bit j cleared
                const int ixj = i | j;
                const bool up = (((i & (P - 1)) & k) == 0);             // ascending block (direction relative to the sector's segment)
                unsigned long long a = keys[i], b = keys[ixj];
                if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
            }
            __syncthreads();
        }
    }
    // greedy walks: wave 0 only
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        int n_sharp = 0, n_less = 0, n_flat = 0;
        for (int j = 0; j < 6; ++j) {
            const int len = ep[j] - sp[j] + 1;
            const unsigned long long *kj = keys + j * P;
            // ---- edge walk, descending curvature (cpp:166-215)
            int largest_picked_num = 0;
            bool stop = false;
            for (int base = len - 1; base >= 0 && !stop; base -= 64) {
                const int k = base - lane;
                const int li = (k >= 0) ? int(unsigned(kj[k])) : 0;
                const bool c_ok = (k >= 0) && (double(sc[li]) > 0.1);
                while (true) {
                    const bool elig = c_ok && (spicked[li] == 0);
                    const unsigned long long m = __ballot(elig);
                    if (!m) break;
                    const int l = __ffsll((long long)m) - 1;
                    const int sel = __shfl(li, l);
                    largest_picked_num++;
                    if (largest_picked_num <= 2) {
                        if (lane == 0) { slabel[sel] = 2; s_stage[n_sharp] = sel + g0; s_stage[STAGE_SHARP + n_less] = sel + g0; }
                        n_sharp++; n_less++;
                    } else if (largest_picked_num <= 20) {
                        if (lane == 0) { slabel[sel] = 1; s_stage[STAGE_SHARP + n_less] = sel + g0; }
                        n_less++;
                    } else { stop = true; break; }
                    if (lane == 0) spicked[sel] = 1;
                    suppress_neighbours(sx, sy, sz, spicked, sel, lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
                // sorted descending: once a candidate fails c > 0.1 every later one fails too
                const unsigned long long inb = __ballot(k >= 0);
                if (__ballot(c_ok) != inb) stop = true;
            }
            // ---- flat walk, ascending curvature (cpp:219-256)
            int smallest_picked_num = 0;
            stop = false;
            for (int base = 0; base < len && !stop; base += 64) {
                const int k = base + lane;
                const int li = (k < len) ? int(unsigned(kj[k])) : 0;
                const bool c_ok = (k < len) && (double(sc[li]) < 0.1);
                while (true) {
                    const bool elig = c_ok && (spicked[li] == 0);
                    const unsigned long long m = __ballot(elig);
                    if (!m) break;
                    const int l = __ffsll((long long)m) - 1;
                    const int sel = __shfl(li, l);
                    if (lane == 0) { slabel[sel] = -1; s_stage[STAGE_SHARP + STAGE_LESS + n_flat] = sel + g0; }
                    n_flat++;
                    smallest_picked_num++;
                    if (smallest_picked_num >= 4) { stop = true; break; }
                    if (lane == 0) spicked[sel] = 1;
                    suppress_neighbours(sx, sy, sz, spicked, sel, lane);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                }
                const unsigned long long inb = __ballot(k < len);
                if (__ballot(c_ok) != inb) stop = true;
            }
        }
        if (lane == 0) { s_cnt[0] = n_sharp; s_cnt[1] = n_less; s_cnt[2] = n_flat; }
    }
    __syncthreads();
    // write back labels / picked, count less-flat (label <= 0 over [s, e-1])
    int my_lf = 0;
    for (int t = threadIdx.x; t < span; t += 256) {
        const int gi = g0 + t;
        if (spicked[t]) A.picked[gi] = 1;
        if (gi >= s && gi <= e - 1) {
            const int lb = slabel[t];
            if (lb != 0) A.label[gi] = lb;
            if (lb <= 0) my_lf++;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) my_lf += __shfl_xor(my_lf, off);
    if ((threadIdx.x & 63) == 0 && my_lf) atomicAdd(&s_cnt[3], my_lf);
    for (int t = threadIdx.x; t < STAGE_STRIDE; t += 256) A.stage[ring * STAGE_STRIDE + t] = s_stage[t];
    __syncthreads();
    if (threadIdx.x < 4) rc[threadIdx.x] = s_cnt[threadIdx.x];
}

// exclusive scan over rings of the 4 list sizes; single workgroup
__global__ __launch_bounds__(256) void offsets_kernel(const int *__restrict__ ring_counts, int n_rings, int *__restrict__ ring_offsets,
                                                      int *__restrict__ totals)
{
    if (threadIdx.x < 4) {
        int acc = 0;
        for (int r = 0; r < n_rings; ++r) {
            ring_offsets[r * 4 + threadIdx.x] = acc;
            acc += ring_counts[r * 4 + threadIdx.x];
        }
        totals[threadIdx.x] = acc;
    }
}

struct EmitArgs {
    const int *start, *end, *label, *stage, *ring_counts, *ring_offsets;
    int *list0, *list1, *list2, *list3;
};

__global__ __launch_bounds__(256) void emit_kernel(EmitArgs A)
{
    __shared__ int s_w[4];
    __shared__ int s_base;
    const int ring = blockIdx.x;
    const int *rc = A.ring_counts + ring * 4, *ro = A.ring_offsets + ring * 4;
    const int *st = A.stage + ring * STAGE_STRIDE;
    for (int t = threadIdx.x; t < rc[0]; t += 256) A.list0[ro[0] + t] = st[t];
    for (int t = threadIdx.x; t < rc[1]; t += 256) A.list1[ro[1] + t] = st[STAGE_SHARP + t];
    for (int t = threadIdx.x; t < rc[2]; t += 256) A.list2[ro[2] + t] = st[STAGE_SHARP + STAGE_LESS + t];
    if (rc[3] == 0) return;
    const int s = A.start[ring], e = A.end[ring];
    if (threadIdx.x == 0) s_base = ro[3];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c0 = s; c0 <= e - 1; c0 += 256) {
        const int gi = c0 + threadIdx.x;
        const bool take = (gi <= e - 1) && (A.label[gi] <= 0);
        const unsigned long long m = __ballot(take);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_w[wave] = __popcll(m);
        __syncthreads();
        int wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_w[w];
        const int total = s_w[0] + s_w[1] + s_w[2] + s_w[3];
        if (take) A.list3[s_base + wbase + before] = gi;
        __syncthreads();
        if (threadIdx.x == 0) s_base += total;
        __syncthreads();
    }
}

int extract_run(mlh_ctx *ctx)
{
    ScanBuf &sb = ctx->scan;
    if (sb.n <= 0 || sb.n_rings <= 0) return fail(ctx, MLH_ERR_STATE, "scan_upload has not been called");
    hipStream_t st = ctx->stream;
    const int n = sb.n, R = sb.n_rings;
    MLH_HIP(ctx, sb.curvature.ensure(sizeof(float) * size_t(n)));
    MLH_HIP(ctx, sb.label.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, sb.picked.ensure(sizeof(int) * size_t(n)));
    MLH_HIP(ctx, sb.stage.ensure(sizeof(int) * STAGE_STRIDE * size_t(R)));
    MLH_HIP(ctx, sb.ring_counts.ensure(sizeof(int) * 4 * size_t(R)));
    MLH_HIP(ctx, sb.ring_offsets.ensure(sizeof(int) * 4 * size_t(R)));
    MLH_HIP(ctx, sb.totals.ensure(sizeof(int) * 4));
    MLH_HIP(ctx, sb.lists[0].ensure(sizeof(int) * STAGE_SHARP * size_t(R)));
    MLH_HIP(ctx, sb.lists[1].ensure(sizeof(int) * STAGE_LESS * size_t(R)));
    MLH_HIP(ctx, sb.lists[2].ensure(sizeof(int) * STAGE_FLAT * size_t(R)));
    MLH_HIP(ctx, sb.lists[3].ensure(sizeof(int) * size_t(n)));

    // LDS budget of the label kernel from the widest ring
    const int max_span = ((sb.max_ring_len + 11 + 3) / 4) * 4;
    int max_sector = (sb.max_ring_len + 5) / 6 + 1;
    int P = 64;
    while (P < max_sector) P <<= 1;
    const size_t lds = sizeof(float) * 4 * size_t(max_span) + sizeof(int) * 2 * size_t(max_span) + 8 + sizeof(unsigned long long) * 6 * size_t(P);
    if (lds > 160 * 1024 - 1024) return fail(ctx, MLH_ERR_UNSUPPORTED, "ring too long for the LDS-resident label kernel");
    MLH_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(label_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));

    prof_begin(ctx, MLH_K_EXTRACT);
    hipLaunchKernelGGL(curvature_kernel, dim3((n + 255) / 256), dim3(256), 0, st, sb.pts.as<float4>(), n, sb.curvature.as<float>(),
                       sb.label.as<int>(), sb.picked.as<int>());
    LabelArgs la;
    la.pts = sb.pts.as<float4>(); la.curv = sb.curvature.as<float>(); la.start = sb.start.as<int>(); la.end = sb.end.as<int>();
    la.label = sb.label.as<int>(); la.picked = sb.picked.as<int>(); la.stage = sb.stage.as<int>(); la.ring_counts = sb.ring_counts.as<int>();
    la.n = n; la.max_span = max_span; la.sort_p = P;
    hipLaunchKernelGGL(label_kernel, dim3(R), dim3(256), lds, st, la);
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(256), 0, st, sb.ring_counts.as<int>(), R, sb.ring_offsets.as<int>(), sb.totals.as<int>());
    EmitArgs ea;
    ea.start = sb.start.as<int>(); ea.end = sb.end.as<int>(); ea.label = sb.label.as<int>(); ea.stage = sb.stage.as<int>();
    ea.ring_counts = sb.ring_counts.as<int>(); ea.ring_offsets = sb.ring_offsets.as<int>();
    ea.list0 = sb.lists[0].as<int>(); ea.list1 = sb.lists[1].as<int>(); ea.list2 = sb.lists[2].as<int>(); ea.list3 = sb.lists[3].as<int>();
    hipLaunchKernelGGL(emit_kernel, dim3(R), dim3(256), 0, st, ea);
    prof_end(ctx, MLH_K_EXTRACT);
    MLH_HIP(ctx, hipGetLastError());
    sb.extracted = true;
    return MLH_OK;
}

}  // namespace mlh
