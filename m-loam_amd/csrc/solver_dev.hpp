// Device-side pieces of the optimiser shared by solver.hip (stand-alone update kernels) and match.hip (the fit kernel's
// last-arriving workgroup finishes the Gauss-Newton iteration itself): fixed-order summation of the per-tile partial normal
// equations, evalDegenracy (lidar_mapper_keyframe.cpp:1172-1204), register-resident 6x6 Cholesky, the GN step.
#pragma once
#include "ctx.hpp"
#include "dev_math.hpp"

namespace mlh {

struct SumArgs {
    const double *p;   // per-tile partial records (NE_STRIDE doubles each), surf tiles first, then corner tiles
    int lo[2], hi[2];  // two tile ranges [lo, hi) summed in this order (a pose block's surf tiles, then its corner tiles)
};

// NT threads (256 or 512): column c = tid & 31, slice s = tid >> 5 sums every (NT/32)-th tile of the ranges; the slices are combined in
// fixed order.
// Record layout: [0..20] J^T J upper, [21..26] J^T r, [27] cost, [28] count, [29] surf count, [30] corner count.
template <int NT = 256, int U = 12>
// `a` travels by reference: the caller's copy is the 32 bytes of scratch rocprof shows for the kernels that end in this call (only the last
// workgroup's serial tail touches them). Passing the six words by value removes the scratch and was measured 0.5 us SLOWER per launch
// (A/B inside one gpurun call, three alternations: 15.9 vs 15.4 us for fit_linearize_kernel<5,false>), so it stays a reference.
__device__ __noinline__ void sum_partials(const SumArgs &a, double *ne /*LDS, NE_STRIDE*/, double *cnt2 /*LDS, 2*/, double *scratch /*LDS (NT/32)*32*/)
{
    constexpr int NS = NT / 32;
    const int c = threadIdx.x & 31, s = threadIdx.x >> 5;
    double v = 0.0;
    if (a.p) {
        // the two tile ranges (surf, corner) are walked as one list; the U loads of a trip are independent (in flight together) and
        // feed four chains in a fixed association -> deterministic, and ~ntiles/(U NS) round trips instead of one per tile
        const int n0 = max(a.hi[0] - a.lo[0], 0), ntot = n0 + max(a.hi[1] - a.lo[1], 0);
        double ch[4] = {0.0, 0.0, 0.0, 0.0};
        for (int j = s; j < ntot; j += U * NS) {
            double t[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + NS * u;
                const int tile = jj < n0 ? a.lo[0] + jj : a.lo[1] + (jj - n0);
                t[u] = jj < ntot ? a.p[size_t(tile) * NE_STRIDE + c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) ch[u & 3] += t[u];
        }
        v = (ch[0] + ch[1]) + (ch[2] + ch[3]);
    }
    scratch[s * 32 + c] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < NS; ++q) t += scratch[q * 32 + c];
        ne[c] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) { cnt2[0] = ne[NE_CNT + 1]; cnt2[1] = ne[NE_CNT + 2]; }
    __syncthreads();
}

// either sum this rank's partials, or (multi-GPU) take the already all-reduced record from the solver state
__device__ inline void gather_ne(const SumArgs &a, const SolverState *S, int pre_reduced, double *ne, double *cnt2, double *scratch)
{
    if (pre_reduced) {
        if (threadIdx.x < NE_STRIDE) ne[threadIdx.x] = S->ne[threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) { cnt2[0] = ne[NE_CNT + 1]; cnt2[1] = ne[NE_CNT + 2]; }
        __syncthreads();
    } else {
        sum_partials(a, ne, cnt2, scratch);
    }
}

__device__ __forceinline__ void unpack_H(const double *ne, double (&H)[36])
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = i; j < 6; ++j) {
            const int q = i * 6 - (i * (i - 1)) / 2 + (j - i);   // packed upper-triangular index
            H[i * 6 + j] = ne[q]; H[j * 6 + i] = ne[q];
        }
}

// cyclic Jacobi, eigenvalues ascending, eigenvectors in the columns of V (row-major 6x6).
// `a` (36) and `V` (36) must be in LDS (or global): they are indexed dynamically, and keeping them out of private memory
// keeps the calling kernels free of scratch. One lane runs this; it is the rare path (degenerate geometry or stats requested).
__device__ __forceinline__ void jacobi6_mem(double *a, double *V, double *ev)
{
    // rare path (degenerate geometry / statistics requested): loops kept rolled so that this function stays at a few dozen VGPRs --
    // a kernel's allocation is the maximum over everything it can call
#pragma unroll 1
    for (int i = 0; i < 36; ++i) V[i] = ((i % 7) == 0) ? 1.0 : 0.0;
#pragma unroll 1
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll 1
        for (int i = 0; i < 6; ++i) {
            dg += a[i * 6 + i] * a[i * 6 + i];
#pragma unroll 1
            for (int j = i + 1; j < 6; ++j) off += a[i * 6 + j] * a[i * 6 + j];
        }
        if (off <= 1e-32 * dg || off == 0.0) break;
#pragma unroll 1
        for (int p = 0; p < 5; ++p)
#pragma unroll 1
            for (int q = p + 1; q < 6; ++q) {
                double apq = a[p * 6 + q];
                if (apq == 0.0) continue;
                double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll 1
                for (int k = 0; k < 6; ++k) {
                    double akp = a[k * 6 + p], akq = a[k * 6 + q];
                    a[k * 6 + p] = c * akp - s * akq;
                    a[k * 6 + q] = s * akp + c * akq;
                }
#pragma unroll 1
                for (int k = 0; k < 6; ++k) {
                    double apk = a[p * 6 + k], aqk = a[q * 6 + k];
                    a[p * 6 + k] = c * apk - s * aqk;
                    a[q * 6 + k] = s * apk + c * aqk;
                }
#pragma unroll 1
                for (int k = 0; k < 6; ++k) {
                    double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
                    V[k * 6 + p] = c * vkp - s * vkq;
                    V[k * 6 + q] = s * vkp + c * vkq;
                }
            }
    }
#pragma unroll 1
    for (int i = 0; i < 6; ++i) ev[i] = a[i * 6 + i];
#pragma unroll 1
    for (int i = 0; i < 5; ++i) {
        int k = i;
#pragma unroll 1
        for (int j = i + 1; j < 6; ++j) if (ev[j] < ev[k]) k = j;
        if (k != i) {
            double t = ev[i]; ev[i] = ev[k]; ev[k] = t;
#pragma unroll 1
            for (int r = 0; r < 6; ++r) { double u = V[r * 6 + i]; V[r * 6 + i] = V[r * 6 + k]; V[r * 6 + k] = u; }
        }
    }
}

// evalDegenracy: zero the eigenvectors below the threshold (ascending, stop at the first one above),
// V_update = (V_f^T)^-1 V_p^T = V_f V_p^T for orthonormal V_f; identity when nothing is degenerate.
// work: 36 + 36 + 6 + 36 doubles of LDS/global scratch space; results: work[72..77] = eigenvalues, work[78..113] = V_update.
constexpr int DEG_WORK = 36 + 36 + 6 + 36;
__device__ __noinline__ bool eval_degeneracy_mem(const double *ne, double thre, double *work)
{
    double *a = work, *Vf = work + 36, *ev = work + 72, *Vupd = work + 78;
    int q = 0;
#pragma unroll 1
    for (int i = 0; i < 6; ++i)
#pragma unroll 1
        for (int j = i; j < 6; ++j) { a[i * 6 + j] = ne[q]; a[j * 6 + i] = ne[q]; ++q; }
    jacobi6_mem(a, Vf, ev);
    bool deg = false, stop = false;
    int first_kept = 6;
#pragma unroll 1
    for (int j = 0; j < 6; ++j) {
        if (!stop && ev[j] < thre) deg = true;
        else { if (!stop) first_kept = j; stop = true; }
    }
#pragma unroll 1
    for (int r = 0; r < 6; ++r)
#pragma unroll 1
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            if (deg) {
#pragma unroll 1
                for (int j = first_kept; j < 6; ++j) s += Vf[r * 6 + j] * Vf[c * 6 + j];
            } else s = (r == c) ? 1.0 : 0.0;
            Vupd[r * 6 + c] = s;
        }
    return deg;
}

// The same cyclic Jacobi with the matrices in REGISTERS (all loops over matrix indices unrolled -> static indexing): identical
// arithmetic and rotation order, but no LDS round trip per element access -- 5x faster than jacobi6_mem, at ~150 VGPRs. Used by
// the single-workgroup LM kernels (where the register footprint costs nothing), NOT by the fit kernel's fused finish.
__device__ __forceinline__ void jacobi6_reg(double (&a)[36], double *V_out, double *ev)
{
    double V[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) V[i] = ((i % 7) == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            dg += a[i * 6 + i] * a[i * 6 + i];
#pragma unroll
            for (int j = i + 1; j < 6; ++j) off += a[i * 6 + j] * a[i * 6 + j];
        }
        if (off <= 1e-32 * dg || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
            for (int q = p + 1; q < 6; ++q) {
                const double apq = a[p * 6 + q];
                if (apq != 0.0) {
                    const double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double akp = a[k * 6 + p], akq = a[k * 6 + q];
                        a[k * 6 + p] = c * akp - sn * akq;
                        a[k * 6 + q] = sn * akp + c * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double apk = a[p * 6 + k], aqk = a[q * 6 + k];
                        a[p * 6 + k] = c * apk - sn * aqk;
                        a[q * 6 + k] = sn * apk + c * aqk;
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
                        V[k * 6 + p] = c * vkp - sn * vkq;
                        V[k * 6 + q] = sn * vkp + c * vkq;
                    }
                }
            }
    }
#pragma unroll
    for (int i = 0; i < 36; ++i) V_out[i] = V[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) ev[i] = a[i * 6 + i];
    // ascending selection sort on the memory copies (dynamic indices)
    for (int i = 0; i < 5; ++i) {
        int k = i;
        for (int j = i + 1; j < 6; ++j) if (ev[j] < ev[k]) k = j;
        if (k != i) {
            double t = ev[i]; ev[i] = ev[k]; ev[k] = t;
            for (int r = 0; r < 6; ++r) { double u = V_out[r * 6 + i]; V_out[r * 6 + i] = V_out[r * 6 + k]; V_out[r * 6 + k] = u; }
        }
    }
}

__device__ __forceinline__ bool eval_degeneracy_reg(const double *ne, double thre, double *work)
{
    double *Vf = work + 36, *ev = work + 72, *Vupd = work + 78;
    double a[36];
    unpack_H(ne, a);
    jacobi6_reg(a, Vf, ev);
    bool deg = false, stop = false;
    int first_kept = 6;
    for (int j = 0; j < 6; ++j) {
        if (!stop && ev[j] < thre) deg = true;
        else { if (!stop) first_kept = j; stop = true; }
    }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            if (deg) { for (int j = first_kept; j < 6; ++j) s += Vf[r * 6 + j] * Vf[c * 6 + j]; }
            else s = (r == c) ? 1.0 : 0.0;
            Vupd[r * 6 + c] = s;
        }
    return deg;
}

// Cholesky factor / solve of a 6x6 SPD system on the PACKED lower triangle (21 doubles, entry (i, j), i >= j, at i(i+1)/2 + j),
// factorised in place and fully unrolled so everything lives in registers (no scratch): 42 VGPRs instead of the 144 a pair of
// full matrices takes -- this code is inlined into the fit kernel's fused finish and would otherwise set that kernel's
// register allocation. The dependent chain is what costs time here (one lane, ~1 wavefront on the chip), so each column takes
// ONE long operation -- r = rsqrt(s) -- and the column and the later substitutions multiply by it instead of dividing.
#define MLH_LT(i, j) ((i) * ((i) + 1) / 2 + (j))
__device__ __forceinline__ bool chol6p_factor(double (&a)[21], double (&inv_d)[6])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = a[MLH_LT(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= a[MLH_LT(j, k)] * a[MLH_LT(j, k)];
        ok = ok && (s > 0.0);
        const double r = rsqrt(s);
        inv_d[j] = r;
        a[MLH_LT(j, j)] = s * r;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = a[MLH_LT(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= a[MLH_LT(i, k)] * a[MLH_LT(j, k)];
            a[MLH_LT(i, j)] = t * r;
        }
    }
    return ok;
}

// L L^T x = b
__device__ __forceinline__ void chol6p_substitute(const double (&L)[21], const double (&inv_d)[6], const double (&b)[6], double (&x)[6])
{
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[MLH_LT(i, k)] * y[k];
        y[i] = s * inv_d[i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[MLH_LT(k, i)] * x[k];
        x[i] = s * inv_d[i];
    }
}

// full row-major symmetric A (lower triangle read)
__device__ __forceinline__ bool chol6_solve(const double (&A)[36], const double (&b)[6], double (&x)[6])
{
    double a[21], inv_d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) a[MLH_LT(i, j)] = A[i * 6 + j];
    if (!chol6p_factor(a, inv_d)) return false;
    chol6p_substitute(a, inv_d, b, x);
    return true;
}

// packed lower triangle of H - shift*I from the reduced record (upper-packed J^T J at ne[0..20])
__device__ __forceinline__ void pack_lower_from_ne(const double *ne, double shift, double (&a)[21])
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            const int q = j * 6 - (j * (j - 1)) / 2 + (i - j);   // upper-packed index of (j, i)
            a[MLH_LT(i, j)] = ne[q] - ((i == j) ? shift : 0.0);
        }
}

__device__ __noinline__ void write_stat_common(IterStatDev *st, const double *ne, const double *cnt2, const double *ev, bool deg)
{
    st->n_surf = int(cnt2[0] + 0.5);
    st->n_corner = int(cnt2[1] + 0.5);
    st->is_degenerate = deg ? 1 : 0;
    st->cost = ne[NE_COST];
    for (int i = 0; i < 6; ++i) { st->eigval[i] = ev[i]; st->g[i] = ne[NE_G + i]; }
    int q = 0;
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { st->H[i * 6 + j] = ne[q]; st->H[j * 6 + i] = ne[q]; ++q; }
}


// The tail of a Gauss-Newton iteration for one pose block: degeneracy test -> solve H d = -g -> x <- Plus(x, V_update d).
// Called by lanes 0 and 1 of one wavefront (converged): both run the same register-resident Cholesky factorisation in
// lock-step -- lane 0 on H (for the solve), lane 1 on H - thre*I (positive definite <=> lambda_min > thre <=> nothing is
// degenerate) -- so the degeneracy test costs no extra time. Lane 0 then finishes.
//   freeze = 0: evalDegenracy (lidar_mapper_keyframe.cpp:1172-1204): project the weak directions out of the update
//   freeze = 1: an extrinsic block whose lambda_min is below the threshold is not updated at all (estimator.cpp:1662-1676)
// `ne` / `cnt2`: the reduced record in LDS; x: the block's pose; S (nullable): the solver state that mirrors block 0.
template <bool REG_JACOBI = false>
__device__ __forceinline__ void gn_finish2(const double *ne, const double *cnt2, double *x, SolverState *S, double eig_thre, int freeze,
                                  IterStatDev *stat, double *work /*LDS, DEG_WORK*/)
{
    const int lane = threadIdx.x & 63;
    double xc[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) xc[i] = x[i];          // issued before the factorisation: the pose arrives while it runs
    double L[21], inv_d[6];
    pack_lower_from_ne(ne, (lane == 1) ? eig_thre * (1.0 + 1e-9) : 0.0, L);
    const bool pd = chol6p_factor(L, inv_d);
    const bool not_degenerate_fast = __shfl(pd ? 1 : 0, 1) != 0;
    if (lane != 0) return;
    bool deg = false;
    const bool slow = !(stat == nullptr && not_degenerate_fast);
    if (slow) deg = REG_JACOBI ? eval_degeneracy_reg(ne, eig_thre, work) : eval_degeneracy_mem(ne, eig_thre, work);
    const bool frozen = freeze && (slow ? deg : false);
    double d[6], rhs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) rhs[i] = -ne[NE_G + i];
    bool ok = pd;
    if (ok) {
        chol6p_substitute(L, inv_d, rhs, d);
    } else {
        pack_lower_from_ne(ne, -1e-6, L);                  // H + 1e-6 I
        ok = chol6p_factor(L, inv_d);
        if (ok) chol6p_substitute(L, inv_d, rhs, d);
    }
    if (ok && !frozen) {
        double xn[7];
        pose_plus(xc, d, slow ? work + 78 : nullptr, xn);   // V_update = I on the fast path
#pragma unroll
        for (int i = 0; i < 7; ++i) x[i] = xn[i];
    }
    if (S) {
        for (int i = 0; i < NE_STRIDE; ++i) S->ne[i] = ne[i];
        for (int i = 0; i < 36; ++i) S->V[i] = slow ? work[78 + i] : (((i % 7) == 0) ? 1.0 : 0.0);
    }
    if (stat) {
        write_stat_common(stat, ne, cnt2, work + 72, deg);
        stat->final_cost = ne[NE_COST];
        stat->lm_iterations = 0; stat->successful_steps = 0; stat->termination = frozen ? 1 : 0;
        for (int i = 0; i < 7; ++i) stat->pose_after[i] = x[i];
    }
}


// ---------------------------------------------------------------- the same tail, spread over the lanes of ONE wavefront
// gn_finish2 keeps two whole packed factorisations in the registers of two lanes (~120 VGPRs): inlined into a kernel whose other
// 500 threads run the correspondence search, that allocation would set the whole launch's occupancy. Here every lane owns ONE ROW
// of the lower triangle: lanes 0..5 factorise H, lanes 8..13 factorise H - thre*I (positive definite <=> nothing degenerate) in
// lock-step; a column step broadcasts row j's finished entries with v_readlane (uniform source lane, no LDS trip), so a lane
// carries 6 + 6 doubles. Same operations in the same order as chol6p_factor / chol6p_substitute.
__device__ __forceinline__ double bcast_pair(double v, int j, bool grp_b)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int alo = __builtin_amdgcn_readlane(lo, j), ahi = __builtin_amdgcn_readlane(hi, j);
    const int blo = __builtin_amdgcn_readlane(lo, 8 + j), bhi = __builtin_amdgcn_readlane(hi, 8 + j);
    return __hiloint2double(grp_b ? bhi : ahi, grp_b ? blo : alo);
}

// (used by gn_finish_wave and, further down, by the wavefront forms of the LM bodies)
__device__ __forceinline__ double wave_bcast(double v, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}

// pose_plus (dev_math.hpp): dx = V delta on lanes 0..5 (row r, columns in order), the four quotients p / |p| on lanes 0..3. V: 36 doubles or null (identity).
__device__ __forceinline__ void pose_plus_wave(const double (&x)[7], const double (&delta)[6], const double *V, double (&out)[7], int lane)
{
    const int r = lane < 6 ? lane : 0;
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < 6; ++c) s += (V ? V[r * 6 + c] : (r == c ? 1.0 : 0.0)) * delta[c];
    double dx[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) dx[k] = wave_bcast(s, k);
    const q4 q{x[3], x[4], x[5], x[6]};
    const q4 dq{dx[3] / 2.0, dx[4] / 2.0, dx[5] / 2.0, 1.0};
    q4 p = qmul(q, dq);
    const double n2 = p.x * p.x + p.y * p.y + p.z * p.z + p.w * p.w;
    if (n2 > 0.0) {
        const double n = sqrt(n2);
        const double mine = (lane & 3) == 0 ? p.x : ((lane & 3) == 1 ? p.y : ((lane & 3) == 2 ? p.z : p.w));
        const double quo = mine / n;
        p.x = wave_bcast(quo, 0); p.y = wave_bcast(quo, 1); p.z = wave_bcast(quo, 2); p.w = wave_bcast(quo, 3);
    }
    out[0] = x[0] + dx[0]; out[1] = x[1] + dx[1]; out[2] = x[2] + dx[2];
    out[3] = p.x; out[4] = p.y; out[5] = p.z; out[6] = p.w;
}

// Called by ALL 64 lanes of one wavefront, converged. Arguments as gn_finish2.
// x_in: where the block's pose is read from when that is not where the updated pose goes (null: x).
__device__ __forceinline__ void gn_finish_wave(const double *ne, const double *cnt2, double *x, SolverState *S, double eig_thre, int freeze,
                                               IterStatDev *stat, double *work /*LDS, DEG_WORK*/, double (&x_out)[7], const double *x_in = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int i = lane & 7;
    const bool grp_b = (lane & 8) != 0;
    double xc[7];
#pragma unroll
    for (int q = 0; q < 7; ++q) xc[q] = (x_in ? x_in : x)[q];          // issued before the factorisation: the pose arrives while it runs
    double a[6], colv[6], rinv[6], d[6];
    bool pd = false, not_degenerate_fast = false;
#pragma unroll 1
    for (int attempt = 0; attempt < 2 && !pd; ++attempt) {
        // attempt 0: H (lanes 0..7) and H - thre*I (lanes 8..15); attempt 1 (H not positive definite): H + 1e-6 I
        const double shift = attempt ? -1e-6 : (grp_b ? eig_thre * (1.0 + 1e-9) : 0.0);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const int q = k * 6 - (k * (k - 1)) / 2 + (i - k);          // upper-packed index of (k, i), k <= i
            a[k] = (i < 6 && k <= i) ? ne[q] - ((k == i) ? shift : 0.0) : 0.0;
            colv[k] = 0.0;
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double rj[6];
#pragma unroll
            for (int k = 0; k < j; ++k) rj[k] = bcast_pair(a[k], j, grp_b);     // L(j, k)
            double t = a[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= a[k] * rj[k];
            const double sj = bcast_pair(t, j, grp_b);
            ok = ok && (sj > 0.0);
            const double r = rsqrt(sj);
            rinv[j] = r;
            a[j] = (i == j) ? sj * r : t * r;
#pragma unroll
            for (int k = 0; k < j; ++k) if (i == k) colv[j] = rj[k];            // the transpose: colv[j] = L(j, i), j > i
        }
        pd = __builtin_amdgcn_readlane(ok ? 1 : 0, 0) != 0;
        if (attempt == 0) not_degenerate_fast = __builtin_amdgcn_readlane(ok ? 1 : 0, 8) != 0;
    }
    if (pd) {
        // L y = -g, then L^T d = y (group A; the other lanes compute along and are ignored)
        double b = (i < 6) ? -ne[NE_G + i] : 0.0, y = 0.0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double yk = b * rinv[k];
            const double ykb = bcast_pair(yk, k, false);
            if (i == k) y = yk;
            if (i > k) b -= a[k] * ykb;
        }
#pragma unroll
        for (int k = 5; k >= 0; --k) {
            double sacc = y;
#pragma unroll
            for (int m = k + 1; m < 6; ++m) sacc -= colv[m] * d[m];
            d[k] = bcast_pair(sacc * rinv[k], k, false);
        }
    }
    // x_out: the block's pose as this call leaves it, uniform over the wavefront (a publication that follows takes it from here instead of reading x back)
#pragma unroll
    for (int q = 0; q < 7; ++q) x_out[q] = xc[q];
    const bool slow = !(stat == nullptr && not_degenerate_fast);
    if (!slow) {
        // the common case -- nothing degenerate, nobody asked for eigenvalues: V_update = I, and the update runs on the wavefront (the four quotients of the
        // quaternion's normalisation on four lanes instead of one after the other)
        if (pd) {
            double dd[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) dd[q] = d[q];
            pose_plus_wave(xc, dd, nullptr, x_out, lane);
            if (lane < 7) x[lane] = lane == 0 ? x_out[0] : (lane == 1 ? x_out[1] : (lane == 2 ? x_out[2] : (lane == 3 ? x_out[3] : (lane == 4 ? x_out[4] : (lane == 5 ? x_out[5] : x_out[6])))));
        }
        if (lane != 0) return;
        if (S) {
            for (int q = 0; q < NE_STRIDE; ++q) S->ne[q] = ne[q];
            for (int q = 0; q < 36; ++q) S->V[q] = ((q % 7) == 0) ? 1.0 : 0.0;
        }
        return;
    }
    if (lane == 0) {
        bool deg = eval_degeneracy_mem(ne, eig_thre, work);
        const bool frozen = freeze && deg;
        if (pd && !frozen) {
            double xn[7];
            pose_plus(xc, d, work + 78, xn);
#pragma unroll
            for (int q = 0; q < 7; ++q) { x[q] = xn[q]; x_out[q] = xn[q]; }
        }
        if (S) {
            for (int q = 0; q < NE_STRIDE; ++q) S->ne[q] = ne[q];
            for (int q = 0; q < 36; ++q) S->V[q] = work[78 + q];
        }
        if (stat) {
            write_stat_common(stat, ne, cnt2, work + 72, deg);
            stat->final_cost = ne[NE_COST];
            stat->lm_iterations = 0; stat->successful_steps = 0; stat->termination = frozen ? 1 : 0;
            for (int q = 0; q < 7; ++q) stat->pose_after[q] = x[q];
        }
    }
#pragma unroll
    for (int q = 0; q < 7; ++q) x_out[q] = wave_bcast(x_out[q], 0);
}

// ---------------------------------------------------------------- Levenberg-Marquardt (Ceres trust-region semantics)
// Bodies of the LM begin / step, run by ONE thread after the record has been summed into LDS. They are device functions so
// that both the stand-alone single-workgroup kernels (solver.hip: multi-GPU, tracker, good-feature paths) and the last-arriving
// workgroup of the linearisation kernels (match.hip: scan2map on one GPU -- no extra launch per LM iteration) can run them.
__device__ __forceinline__ double gradient_max_norm(const SolverState *S)
{
    double ng[6], xp[7];
    for (int i = 0; i < 6; ++i) ng[i] = -S->ne[NE_G + i];
    pose_plus(S->x, ng, S->V, xp);
    double m = 0.0;
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(S->x[i] - xp[i]));
    return m;
}

__device__ __forceinline__ void lm_propose(SolverState *S, int max_it)
{
    while (true) {
        if (S->iteration >= max_it) { S->done = 1; S->termination = 0; return; }
        if (S->gmax <= 1e-10) { S->done = 1; S->termination = 1; return; }
        if (S->radius <= 1e-32) { S->done = 1; S->termination = 4; return; }
        S->iteration++;
        double H[36], A[36], gs[6];
        unpack_H(S->ne, H);
        for (int r = 0; r < 6; ++r) {
            gs[r] = S->S[r] * S->ne[NE_G + r];
            for (int c = 0; c < 6; ++c) A[r * 6 + c] = S->S[r] * H[r * 6 + c] * S->S[c];
        }
        if (!S->reuse_diagonal)
            for (int i = 0; i < 6; ++i) S->diag[i] = fmin(fmax(A[i * 6 + i], 1e-6), 1e32);
        double lhs[36];
        for (int i = 0; i < 36; ++i) lhs[i] = A[i];
        for (int i = 0; i < 6; ++i) lhs[i * 6 + i] += S->diag[i] / S->radius;
        double y[6], step[6];
        bool ok = chol6_solve(lhs, gs, y);
        S->reuse_diagonal = 1;
        bool valid = false;
        double mcc = 0.0;
        if (ok) {
            double sg = 0.0, sAs = 0.0;
            for (int r = 0; r < 6; ++r) step[r] = -y[r];
            for (int r = 0; r < 6; ++r) {
                sg += step[r] * gs[r];
                double t = 0.0;
                for (int c = 0; c < 6; ++c) t += A[r * 6 + c] * step[c];
                sAs += step[r] * t;
            }
            mcc = -(sg + 0.5 * sAs);
            valid = mcc > 0.0;
        }
        if (!valid) {
            if (++S->num_invalid >= 5) { S->done = 1; S->termination = 4; return; }
            S->radius /= S->decrease_factor; S->decrease_factor *= 2.0; S->reuse_diagonal = 1;
            continue;
        }
        S->num_invalid = 0;
        double delta[6];
        for (int i = 0; i < 6; ++i) delta[i] = step[i] * S->S[i];
        pose_plus(S->x, delta, S->V, S->cand);
        S->model_cost_change = mcc;
        return;
    }
}

// ne / cnt2 / scratch: LDS. eig_thre < 0: no degeneracy handling (V_update = I); stat may be null
__device__ __forceinline__ void lm_begin_body(const double *ne, const double *cnt2, double *scratch, SolverState *S, double eig_thre, int max_it,
                                     IterStatDev *stat, int min_blocks)
{
    // evalDegenracy. Nobody asked for the eigenvalues (stat == null): H - thre*I positive definite <=> lambda_min > thre <=> nothing
    // is degenerate, V_update = I -- one Cholesky factorisation instead of the eigen-decomposition; otherwise the full procedure.
    // eig_thre < 0: the caller has no degeneracy handling at all (LidarTracker: V_update stays the identity)
    bool deg = false, fast = eig_thre < 0.0;
    if (!fast && !stat) {
        double L[21], inv_d[6];
        pack_lower_from_ne(ne, eig_thre * (1.0 + 1e-9), L);
        fast = chol6p_factor(L, inv_d);
    }
    if (!fast) deg = eval_degeneracy_reg(ne, eig_thre, scratch);
    for (int i = 0; i < NE_STRIDE; ++i) S->ne[i] = ne[i];
    for (int i = 0; i < 36; ++i) S->V[i] = fast ? (((i % 7) == 0) ? 1.0 : 0.0) : scratch[78 + i];
    {
        int q = 0;
        for (int i = 0; i < 6; ++i) { S->S[i] = 1.0 / (1.0 + sqrt(ne[q])); q += 6 - i; }   // Jacobi scaling from diag(J^T J)
    }
    S->radius = 1e4; S->decrease_factor = 2.0; S->reuse_diagonal = 0;
    S->iteration = 0; S->done = 0; S->termination = 0; S->num_successful = 0; S->num_invalid = 0; S->evaluations = 1;
    S->gmax = gradient_max_norm(S);
    if (stat) {
        if (fast) for (int i = 0; i < 6; ++i) scratch[72 + i] = 0.0;     // no eigenvalues were computed
        write_stat_common(stat, ne, cnt2, scratch + 72, deg);
        stat->final_cost = ne[NE_COST];
    }
    // too few residual blocks (lidar_tracker.cpp:66-70 "less correspondence": the round is skipped)
    if (ne[NE_CNT] < double(min_blocks)) { S->done = 1; S->termination = 4; return; }
    lm_propose(S, max_it);
}

// ce: the summed record at the candidate pose (LDS)
__device__ __forceinline__ void lm_step_body(const double *ce, SolverState *S, int max_it)
{
    S->evaluations++;
    double step_norm = 0.0, x_norm = 0.0;
    for (int i = 0; i < 7; ++i) { double d = S->x[i] - S->cand[i]; step_norm += d * d; x_norm += S->x[i] * S->x[i]; }
    step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { S->done = 1; S->termination = 2; return; }
    const double x_cost = S->ne[NE_COST];
    const double cost_change = x_cost - ce[NE_COST];
    if (fabs(cost_change) <= 1e-6 * x_cost) { S->done = 1; S->termination = 3; return; }
    const double rd = cost_change / S->model_cost_change;
    if (rd > 1e-3) {
        for (int i = 0; i < 7; ++i) S->x[i] = S->cand[i];
        for (int i = 0; i < NE_STRIDE; ++i) S->ne[i] = ce[i];
        S->num_successful++;
        double t = 2.0 * rd - 1.0;
        S->radius = S->radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
        S->radius = fmin(1e16, S->radius);
        S->decrease_factor = 2.0;
        S->reuse_diagonal = 0;
        S->gmax = gradient_max_norm(S);
    } else {
        S->radius /= S->decrease_factor; S->decrease_factor *= 2.0; S->reuse_diagonal = 1;
    }
    lm_propose(S, max_it);
}

// ---------------------------------------------------------------- the same bodies on ONE WAVEFRONT
// Run by one thread, an LM step costs ~4.9 us of a 12.4 us launch (knock-out timing, profiles/r03_knockout_experiments.txt): ~500 dependent f64 operations and two
// dozen divisions / square roots on a single lane. Here all 64 lanes of a wavefront enter converged and hold the small vectors (pose, step, scalars) uniformly;
// lane r < 6 owns ROW r of every 6 x 6 object (V_update, the Jacobi-scaled J^T J, its Cholesky factor), independent divisions go to different lanes, results
// travel by v_readlane. Every output element is computed by the same operations in the same order as in lm_step_body / lm_propose / pose_plus above, so the two
// forms are interchangeable bit for bit (the stand-alone LM kernels of solver.hip -- multi-GPU and good-feature paths -- keep the one-thread form, the fused
// single-GPU scan2map and the tracker run this one; tests/test_gpu_parity.py::test_rccl_single_rank_path compares their poses for equality).
#ifdef MLH_STAGE_CLOCK
// debug build only: stamps inside the LM step of workgroup 0 (scripts/stageclock_loop.py)
static __device__ unsigned long long g_step_clk[16];
#define MLH_STEP_STAMP(i) do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) g_step_clk[i] = wall_clock64(); } while (0)
#else
#define MLH_STEP_STAMP(i) do { } while (0)
#endif
// the LM part of the solver state, uniform over the wavefront
struct LmRegs {
    double x[7], g[6], Sv[6], diag[6];
    double radius, decrease_factor, model_cost_change, gmax;
    int reuse_diagonal, iteration, done, termination, num_successful, num_invalid, evaluations;
};

__device__ __forceinline__ double gradient_max_norm_wave(const LmRegs &R, const double *V, int lane)
{
    double ng[6], xp[7];
#pragma unroll
    for (int i = 0; i < 6; ++i) ng[i] = -R.g[i];
    pose_plus_wave(R.x, ng, V, xp, lane);
    double m = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(R.x[i] - xp[i]));
    return m;
}

// lm_propose: ne = the record at x (any memory, read by lanes 0..5), V = V_update (36 doubles). On return `cand` holds the candidate when !R.done.
__device__ __forceinline__ void lm_propose_wave(LmRegs &R, const double *ne, const double *V, double (&cand)[7], int max_it, int lane)
{
    const int r = lane < 6 ? lane : 0;
    double Hrow[6];                                   // row r of J^T J (upper-packed record)
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const int i = r < c ? r : c, j = r < c ? c : r;
        Hrow[c] = ne[i * 6 - (i * (i - 1)) / 2 + (j - i)];
    }
    const double Sr = lane == 0 ? R.Sv[0] : (lane == 1 ? R.Sv[1] : (lane == 2 ? R.Sv[2] : (lane == 3 ? R.Sv[3] : (lane == 4 ? R.Sv[4] : R.Sv[5]))));
    while (true) {
        if (R.iteration >= max_it) { R.done = 1; R.termination = 0; return; }
        if (R.gmax <= 1e-10) { R.done = 1; R.termination = 1; return; }
        if (R.radius <= 1e-32) { R.done = 1; R.termination = 4; return; }
        R.iteration++;
        double A[6];                                  // row r of the Jacobi-scaled matrix
#pragma unroll
        for (int c = 0; c < 6; ++c) A[c] = Sr * Hrow[c] * R.Sv[c];
        const double gs_r = Sr * (lane == 0 ? R.g[0] : (lane == 1 ? R.g[1] : (lane == 2 ? R.g[2] : (lane == 3 ? R.g[3] : (lane == 4 ? R.g[4] : R.g[5])))));
        const double Arr = lane == 0 ? A[0] : (lane == 1 ? A[1] : (lane == 2 ? A[2] : (lane == 3 ? A[3] : (lane == 4 ? A[4] : A[5]))));
        if (!R.reuse_diagonal) {
            const double dr = fmin(fmax(Arr, 1e-6), 1e32);
#pragma unroll
            for (int i = 0; i < 6; ++i) R.diag[i] = wave_bcast(dr, i);
        }
        const double diag_r = lane == 0 ? R.diag[0] : (lane == 1 ? R.diag[1] : (lane == 2 ? R.diag[2] : (lane == 3 ? R.diag[3] : (lane == 4 ? R.diag[4] : R.diag[5]))));
        const double lhs_rr = Arr + diag_r / R.radius;
        MLH_STEP_STAMP(3);
        // Cholesky of lhs, row r on lane r: chol6p_factor's operations in its order (packed lower triangle: a[k] = L(r, k))
        double a[6], colv[6], rinv[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) { a[k] = (k < r) ? A[k] : ((k == r) ? lhs_rr : 0.0); colv[k] = 0.0; }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            double rj[6];
#pragma unroll
            for (int k = 0; k < j; ++k) rj[k] = wave_bcast(a[k], j);          // L(j, k)
            double t = a[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= a[k] * rj[k];
            const double sj = wave_bcast(t, j);
            ok = ok && (sj > 0.0);
            const double rr = rsqrt(sj);
            rinv[j] = rr;
            a[j] = (r == j) ? sj * rr : t * rr;
#pragma unroll
            for (int k = 0; k < j; ++k) if (r == k) colv[j] = rj[k];          // colv[j] = L(j, r), j > r
        }
        R.reuse_diagonal = 1;
        MLH_STEP_STAMP(4);
        bool valid = false;
        double mcc = 0.0, step[6];
        if (ok) {
            // L y' = gs, L^T y = y' (chol6p_substitute), step = -y
            double b = gs_r, yv = 0.0, y[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const double yk = b * rinv[k];
                const double ykb = wave_bcast(yk, k);
                if (r == k) yv = yk;
                if (r > k) b -= a[k] * ykb;
            }
#pragma unroll
            for (int k = 5; k >= 0; --k) {
                double sacc = yv;
#pragma unroll
                for (int m = k + 1; m < 6; ++m) sacc -= colv[m] * y[m];
                y[k] = wave_bcast(sacc * rinv[k], k);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) step[i] = -y[i];
            double t = 0.0;
#pragma unroll
            for (int c = 0; c < 6; ++c) t += A[c] * step[c];
            double sg = 0.0, sAs = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                sg += step[i] * wave_bcast(gs_r, i);
                sAs += step[i] * wave_bcast(t, i);
            }
            mcc = -(sg + 0.5 * sAs);
            valid = mcc > 0.0;
        }
        if (!valid) {
            if (++R.num_invalid >= 5) { R.done = 1; R.termination = 4; return; }
            R.radius /= R.decrease_factor; R.decrease_factor *= 2.0; R.reuse_diagonal = 1;
            continue;
        }
        R.num_invalid = 0;
        MLH_STEP_STAMP(5);
        double delta[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) delta[i] = step[i] * R.Sv[i];
        pose_plus_wave(R.x, delta, V, cand, lane);
        MLH_STEP_STAMP(6);
        R.model_cost_change = mcc;
        return;
    }
}

template <class ST>
__device__ __forceinline__ void lm_regs_load(LmRegs &R, const ST *S)
{
#pragma unroll
    for (int i = 0; i < 7; ++i) R.x[i] = S->x[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) { R.g[i] = S->ne[NE_G + i]; R.Sv[i] = S->S[i]; R.diag[i] = S->diag[i]; }
    R.radius = S->radius; R.decrease_factor = S->decrease_factor; R.model_cost_change = S->model_cost_change; R.gmax = S->gmax;
    R.reuse_diagonal = S->reuse_diagonal; R.iteration = S->iteration; R.done = S->done; R.termination = S->termination;
    R.num_successful = S->num_successful; R.num_invalid = S->num_invalid; R.evaluations = S->evaluations;
}

// everything but x / ne (stored by the caller when a step is accepted) and cand
template <class ST>
__device__ __forceinline__ void lm_regs_store(const LmRegs &R, const double (&cand)[7], ST *S, int lane)
{
    if (lane < 7 && !R.done) S->cand[lane] = lane == 0 ? cand[0] : (lane == 1 ? cand[1] : (lane == 2 ? cand[2] : (lane == 3 ? cand[3] : (lane == 4 ? cand[4] : (lane == 5 ? cand[5] : cand[6])))));
    if (lane >= 8 && lane < 14) { const int i = lane - 8; S->diag[i] = i == 0 ? R.diag[0] : (i == 1 ? R.diag[1] : (i == 2 ? R.diag[2] : (i == 3 ? R.diag[3] : (i == 4 ? R.diag[4] : R.diag[5])))); }
    if (lane == 16) {
        S->radius = R.radius; S->decrease_factor = R.decrease_factor; S->model_cost_change = R.model_cost_change; S->gmax = R.gmax;
        S->reuse_diagonal = R.reuse_diagonal; S->iteration = R.iteration; S->done = R.done; S->termination = R.termination;
        S->num_successful = R.num_successful; S->num_invalid = R.num_invalid; S->evaluations = R.evaluations;
    }
}

// lm_step_body on a wavefront. ce: the summed record at the candidate pose (LDS). All 64 lanes, converged. x_out / done_out: the state's pose and `done`
// flag as this call leaves them (uniform; what a publication that follows needs, without reading back what other lanes have just stored).
__device__ __forceinline__ void lm_step_body_wave(const double *ce, SolverState *S, int max_it, double (&x_out)[7], int &done_out)
{
    const int lane = threadIdx.x & 63;
    LmRegs R;
    lm_regs_load(R, S);
    double cand[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) cand[i] = S->cand[i];
    const double x_cost = S->ne[NE_COST];
    R.evaluations++;
    double step_norm = 0.0, x_norm = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { const double d = R.x[i] - cand[i]; step_norm += d * d; x_norm += R.x[i] * R.x[i]; }
    step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
    bool stop = false;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { R.done = 1; R.termination = 2; stop = true; }
    const double cost_change = x_cost - ce[NE_COST];
    if (!stop && fabs(cost_change) <= 1e-6 * x_cost) { R.done = 1; R.termination = 3; stop = true; }
    const double *ne_now = S->ne;                      // the record the next proposal is built from
    if (!stop) {
        const double rd = cost_change / R.model_cost_change;
        if (rd > 1e-3) {
#pragma unroll
            for (int i = 0; i < 7; ++i) R.x[i] = cand[i];
            if (lane < 7) S->x[lane] = lane == 0 ? cand[0] : (lane == 1 ? cand[1] : (lane == 2 ? cand[2] : (lane == 3 ? cand[3] : (lane == 4 ? cand[4] : (lane == 5 ? cand[5] : cand[6])))));
            if (lane < NE_STRIDE) S->ne[lane] = ce[lane];
#pragma unroll
            for (int i = 0; i < 6; ++i) R.g[i] = ce[NE_G + i];
            ne_now = ce;                               // same values; LDS, and no wait for the stores above
            R.num_successful++;
            const double t = 2.0 * rd - 1.0;
            R.radius = R.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            R.radius = fmin(1e16, R.radius);
            R.decrease_factor = 2.0;
            R.reuse_diagonal = 0;
            R.gmax = gradient_max_norm_wave(R, S->V, lane);
        } else {
            R.radius /= R.decrease_factor; R.decrease_factor *= 2.0; R.reuse_diagonal = 1;
        }
        lm_propose_wave(R, ne_now, S->V, cand, max_it, lane);
    }
    lm_regs_store(R, cand, S, lane);
#pragma unroll
    for (int i = 0; i < 7; ++i) x_out[i] = R.x[i];
    done_out = R.done;
}

// lm_begin_body on a wavefront (ne / cnt2 / scratch in LDS; all 64 lanes, converged): the degeneracy decision stays on lane 0 (a Cholesky test, or the
// eigen-decomposition when it fails or statistics are requested), the copies, the Jacobi scaling, the gradient norm and the first proposal are spread out.
__device__ __forceinline__ void lm_begin_body_wave(const double *ne, const double *cnt2, double *scratch, SolverState *S, double eig_thre, int max_it,
                                                   IterStatDev *stat, int min_blocks, double (&x_out)[7], int &done_out)
{
    const int lane = threadIdx.x & 63;
    int flags = 0;                                     // bit 0: fast (V_update = I), bit 1: degenerate
    if (lane == 0) {
        bool deg = false, fast = eig_thre < 0.0;
        if (!fast && !stat) {
            double L[21], inv_d[6];
            pack_lower_from_ne(ne, eig_thre * (1.0 + 1e-9), L);
            fast = chol6p_factor(L, inv_d);
        }
        if (!fast) deg = eval_degeneracy_reg(ne, eig_thre, scratch);
        flags = (fast ? 1 : 0) | (deg ? 2 : 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");       // scratch (V_update, eigenvalues) was written by lane 0
    __builtin_amdgcn_wave_barrier();
    flags = __builtin_amdgcn_readlane(flags, 0);
    const bool fast = (flags & 1) != 0, deg = (flags & 2) != 0;
    if (lane < NE_STRIDE) S->ne[lane] = ne[lane];
    if (lane < 36) S->V[lane] = fast ? (((lane % 7) == 0) ? 1.0 : 0.0) : scratch[78 + lane];
    // V_update for this call's own use: LDS (scratch[78..113]) -- on the fast path the identity is written there too, so that nobody reads S->V back
    if (fast && lane < 36) scratch[78 + lane] = ((lane % 7) == 0) ? 1.0 : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    LmRegs R;
#pragma unroll
    for (int i = 0; i < 7; ++i) R.x[i] = S->x[i];
    {
        const int r = lane < 6 ? lane : 0;
        const double sc = 1.0 / (1.0 + sqrt(ne[r * 6 - (r * (r - 1)) / 2]));      // Jacobi scaling from diag(J^T J): upper-packed index of (r, r)
#pragma unroll
        for (int i = 0; i < 6; ++i) { R.Sv[i] = wave_bcast(sc, i); R.g[i] = ne[NE_G + i]; R.diag[i] = S->diag[i]; }
        if (lane < 6) S->S[lane] = sc;
    }
    R.radius = 1e4; R.decrease_factor = 2.0; R.reuse_diagonal = 0; R.model_cost_change = S->model_cost_change;
    R.iteration = 0; R.done = 0; R.termination = 0; R.num_successful = 0; R.num_invalid = 0; R.evaluations = 1;
    R.gmax = gradient_max_norm_wave(R, scratch + 78, lane);
    if (stat && lane == 0) {
        if (fast) for (int i = 0; i < 6; ++i) scratch[72 + i] = 0.0;     // no eigenvalues were computed
        write_stat_common(stat, ne, cnt2, scratch + 72, deg);
        stat->final_cost = ne[NE_COST];
    }
    double cand[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) cand[i] = 0.0;
    // too few residual blocks (lidar_tracker.cpp:66-70 "less correspondence": the round is skipped)
    if (ne[NE_CNT] < double(min_blocks)) { R.done = 1; R.termination = 4; }
    else lm_propose_wave(R, ne, scratch + 78, cand, max_it, lane);
    lm_regs_store(R, cand, S, lane);
#pragma unroll
    for (int i = 0; i < 7; ++i) x_out[i] = R.x[i];
    done_out = R.done;
}

// ---------------------------------------------------------------- the same two bodies for the consumer-side schedule (match.hip: lm_consume_kernel)
// EVERY workgroup of a launch runs them on the same inputs -- the state the previous launch's writer left (Si), the summed records in LDS -- and keeps the outcome
// in registers (R, cand); only the launch's writer (`write`) stores the state the next launch reads (So != Si: the other workgroups may still be reading Si).
// Operation for operation lm_step_body_wave / lm_begin_body_wave above: the same bits.
__device__ __forceinline__ double pick7(const double (&v)[7], int i)
{
    return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : (i == 3 ? v[3] : (i == 4 ? v[4] : (i == 5 ? v[5] : v[6])))));
}
__device__ __forceinline__ double pick6(const double (&v)[6], int i)
{
    return i == 0 ? v[0] : (i == 1 ? v[1] : (i == 2 ? v[2] : (i == 3 ? v[3] : (i == 4 ? v[4] : v[5]))));
}

// what a state record holds beyond lm_regs_store's share: the pose, the record at the pose, V_update, the Jacobi scaling (ne_now / V: any memory)
__device__ __forceinline__ void lm_state_store_pp(const LmRegs &R, const double (&cand)[7], const double *ne_now, const double *V, LmState *So, int lane)
{
    lm_regs_store(R, cand, So, lane);
    if (lane < 7) So->x[lane] = pick7(R.x, lane);
    if (lane < NE_STRIDE) So->ne[lane] = ne_now[lane];
    if (lane < 36) So->V[lane] = V[lane];
    if (lane >= 40 && lane < 46) So->S[lane - 40] = pick6(R.Sv, lane - 40);
}

__device__ __forceinline__ void lm_step_wave_pp(const double *ce, const LmState *Si, LmState *So, bool write, int max_it, LmRegs &R, double (&cand)[7])
{
    const int lane = threadIdx.x & 63;
    MLH_STEP_STAMP(0);
    lm_regs_load(R, Si);
#pragma unroll
    for (int i = 0; i < 7; ++i) cand[i] = Si->cand[i];
    const double x_cost = Si->ne[NE_COST];
    R.evaluations++;
    double step_norm = 0.0, x_norm = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { const double d = R.x[i] - cand[i]; step_norm += d * d; x_norm += R.x[i] * R.x[i]; }
    step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
    bool stop = false;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { R.done = 1; R.termination = 2; stop = true; }
    const double cost_change = x_cost - ce[NE_COST];
    if (!stop && fabs(cost_change) <= 1e-6 * x_cost) { R.done = 1; R.termination = 3; stop = true; }
    const double *ne_now = Si->ne;
    MLH_STEP_STAMP(1);
    if (!stop) {
        const double rd = cost_change / R.model_cost_change;
        if (rd > 1e-3) {
#pragma unroll
            for (int i = 0; i < 7; ++i) R.x[i] = cand[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) R.g[i] = ce[NE_G + i];
            ne_now = ce;
            R.num_successful++;
            const double t = 2.0 * rd - 1.0;
            R.radius = R.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            R.radius = fmin(1e16, R.radius);
            R.decrease_factor = 2.0;
            R.reuse_diagonal = 0;
            R.gmax = gradient_max_norm_wave(R, Si->V, lane);
        } else {
            R.radius /= R.decrease_factor; R.decrease_factor *= 2.0; R.reuse_diagonal = 1;
        }
        MLH_STEP_STAMP(2);
        lm_propose_wave(R, ne_now, Si->V, cand, max_it, lane);
    }
    if (write) lm_state_store_pp(R, cand, ne_now, Si->V, So, lane);
    MLH_STEP_STAMP(7);
}

// lm_step_wave_pp for a caller that KEEPS the state's registers from one step to the next (match.hip: lm_loop_kernel, MLH_LOOP_KEEP_REGS): R / cand / x_cost come in as
// the previous call (or lm_begin_wave_pp) left them and go out updated; only what other lanes read from memory -- the accepted record (S->ne: the rows of a later,
// rejected step's proposal) -- is stored. No state load in front, no state store behind: the caller stores the state once, behind its loop (lm_state_store_pp).
// Operation for operation lm_step_wave_pp: the same bits.
__device__ __forceinline__ void lm_step_wave_keep(const double *ce, LmState *S, int max_it, LmRegs &R, double (&cand)[7], double &x_cost)
{
    const int lane = threadIdx.x & 63;
    // (the gradient is NOT kept: it is the accepted record's, one LDS trip that nothing waits for before the proposal -- and twelve registers fewer across the
    // evaluation, which is what keeps the kernel at two workgroups per compute unit: 256 + 8 registers made it one)
#pragma unroll
    for (int i = 0; i < 6; ++i) R.g[i] = S->ne[NE_G + i];
    R.evaluations++;
    double step_norm = 0.0, x_norm = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { const double d = R.x[i] - cand[i]; step_norm += d * d; x_norm += R.x[i] * R.x[i]; }
    step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
    bool stop = false;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { R.done = 1; R.termination = 2; stop = true; }
    const double cost_change = x_cost - ce[NE_COST];
    if (!stop && fabs(cost_change) <= 1e-6 * x_cost) { R.done = 1; R.termination = 3; stop = true; }
    const double *ne_now = S->ne;
    if (!stop) {
        const double rd = cost_change / R.model_cost_change;
        if (rd > 1e-3) {
#pragma unroll
            for (int i = 0; i < 7; ++i) R.x[i] = cand[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) R.g[i] = ce[NE_G + i];
            ne_now = ce;
            x_cost = ce[NE_COST];
            if (lane < NE_STRIDE) S->ne[lane] = ce[lane];          // (read back by a LATER step only: behind the caller's barrier)
            R.num_successful++;
            const double t = 2.0 * rd - 1.0;
            R.radius = R.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            R.radius = fmin(1e16, R.radius);
            R.decrease_factor = 2.0;
            R.reuse_diagonal = 0;
            R.gmax = gradient_max_norm_wave(R, S->V, lane);
        } else {
            R.radius /= R.decrease_factor; R.decrease_factor *= 2.0; R.reuse_diagonal = 1;
        }
        lm_propose_wave(R, ne_now, S->V, cand, max_it, lane);
    }
}

// The step in two parts, for a caller that has ANOTHER wavefront compute the gradient max-norm an accepted step needs (gradient_max_norm_wave at the candidate, with
// the candidate's gradient: inputs that exist before the step begins) while this one goes on: part 1 is lm_step_wave_pp with the proposal made as if that norm were
// above the tolerance; part 2 -- behind a barrier, with the norm in hand -- takes the proposal back if it is not (Ceres tests the norm before it proposes: the loop
// ends with the accepted pose, termination 1, the iteration count as it was), and stores the state. Same operations on every value that survives: the same bits.
struct LmStepSpec {
    const double *ne_now;
    int accepted, iteration_before, hit_max_it;
};
__device__ __forceinline__ void lm_step_wave_spec1(const double *ce, const LmState *Si, int max_it, LmRegs &R, double (&cand)[7], LmStepSpec &sp)
{
    const int lane = threadIdx.x & 63;
    lm_regs_load(R, Si);
#pragma unroll
    for (int i = 0; i < 7; ++i) cand[i] = Si->cand[i];
    const double x_cost = Si->ne[NE_COST];
    R.evaluations++;
    double step_norm = 0.0, x_norm = 0.0;
#pragma unroll
    for (int i = 0; i < 7; ++i) { const double d = R.x[i] - cand[i]; step_norm += d * d; x_norm += R.x[i] * R.x[i]; }
    step_norm = sqrt(step_norm); x_norm = sqrt(x_norm);
    bool stop = false;
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { R.done = 1; R.termination = 2; stop = true; }
    const double cost_change = x_cost - ce[NE_COST];
    if (!stop && fabs(cost_change) <= 1e-6 * x_cost) { R.done = 1; R.termination = 3; stop = true; }
    sp.ne_now = Si->ne; sp.accepted = 0; sp.iteration_before = R.iteration; sp.hit_max_it = 0;
    if (!stop) {
        const double rd = cost_change / R.model_cost_change;
        const double gmax_state = R.gmax;
        if (rd > 1e-3) {
#pragma unroll
            for (int i = 0; i < 7; ++i) R.x[i] = cand[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) R.g[i] = ce[NE_G + i];
            sp.ne_now = ce;
            sp.accepted = 1;
            R.num_successful++;
            const double t = 2.0 * rd - 1.0;
            R.radius = R.radius / fmax(1.0 / 3.0, 1.0 - t * t * t);
            R.radius = fmin(1e16, R.radius);
            R.decrease_factor = 2.0;
            R.reuse_diagonal = 0;
            R.gmax = 1.0;                              // (above the tolerance: the real one arrives in part 2)
        } else {
            R.radius /= R.decrease_factor; R.decrease_factor *= 2.0; R.reuse_diagonal = 1;
        }
        sp.iteration_before = R.iteration;
        sp.hit_max_it = R.iteration >= max_it ? 1 : 0;
        lm_propose_wave(R, sp.ne_now, Si->V, cand, max_it, lane);
        if (!sp.accepted) R.gmax = gmax_state;
    }
}
__device__ __forceinline__ void lm_step_wave_spec2(const LmState *Si, LmState *So, bool write, LmRegs &R, const double (&cand)[7], const LmStepSpec &sp, double gmax_accepted)
{
    const int lane = threadIdx.x & 63;
    if (sp.accepted) {
        R.gmax = gmax_accepted;
        if (!sp.hit_max_it && gmax_accepted <= 1e-10) { R.done = 1; R.termination = 1; R.iteration = sp.iteration_before; }
    }
    if (write) lm_state_store_pp(R, cand, sp.ne_now, Si->V, So, lane);
}

// x: the pose the records in `ne` were taken at. ne / scratch: LDS. No statistics in this schedule (the classic launches serve a caller who asks for them).
__device__ __forceinline__ void lm_begin_wave_pp(const double *ne, double *scratch, const double (&x)[7], LmState *So, bool write, double eig_thre, int max_it, int min_blocks,
                                                 LmRegs &R, double (&cand)[7])
{
    const int lane = threadIdx.x & 63;
    int flags = 0;
    if (lane == 0) {
        bool fast = eig_thre < 0.0;
        if (!fast) {
            double L[21], inv_d[6];
            pack_lower_from_ne(ne, eig_thre * (1.0 + 1e-9), L);
            fast = chol6p_factor(L, inv_d);
        }
        if (!fast) (void)eval_degeneracy_reg(ne, eig_thre, scratch);
        flags = fast ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
    flags = __builtin_amdgcn_readlane(flags, 0);
    const bool fast = (flags & 1) != 0;
    if (fast && lane < 36) scratch[78 + lane] = ((lane % 7) == 0) ? 1.0 : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 7; ++i) R.x[i] = x[i];
    {
        const int r = lane < 6 ? lane : 0;
        const double sc = 1.0 / (1.0 + sqrt(ne[r * 6 - (r * (r - 1)) / 2]));
#pragma unroll
        for (int i = 0; i < 6; ++i) { R.Sv[i] = wave_bcast(sc, i); R.g[i] = ne[NE_G + i]; R.diag[i] = 0.0; }
    }
    R.radius = 1e4; R.decrease_factor = 2.0; R.reuse_diagonal = 0; R.model_cost_change = 0.0;
    R.iteration = 0; R.done = 0; R.termination = 0; R.num_successful = 0; R.num_invalid = 0; R.evaluations = 1;
    R.gmax = gradient_max_norm_wave(R, scratch + 78, lane);
#pragma unroll
    for (int i = 0; i < 7; ++i) cand[i] = 0.0;
    if (ne[NE_CNT] < double(min_blocks)) { R.done = 1; R.termination = 4; }
    else lm_propose_wave(R, ne, scratch + 78, cand, max_it, lane);
    if (write) lm_state_store_pp(R, cand, ne, scratch + 78, So, lane);
}

}  // namespace mlh
