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

// 256 threads: column c = tid & 31, slice s = tid >> 5 sums every 8th tile of the ranges; the 8 slices are combined in fixed order.
// Record layout: [0..20] J^T J upper, [21..26] J^T r, [27] cost, [28] count, [29] surf count, [30] corner count.
__device__ inline void sum_partials(const SumArgs &a, double *ne /*LDS, NE_STRIDE*/, double *cnt2 /*LDS, 2*/, double *scratch /*LDS 8*32*/)
{
    const int c = threadIdx.x & 31, s = threadIdx.x >> 5;
    double v = 0.0;
    if (a.p) {
        // the two tile ranges (surf, corner) are walked as one list; 12 loads per trip are independent (in flight together) and
        // feed four chains in a fixed association -> deterministic, and ~ntiles/96 round trips instead of one per tile (a frame's
        // ~80 tiles: one)
        const int n0 = max(a.hi[0] - a.lo[0], 0), ntot = n0 + max(a.hi[1] - a.lo[1], 0);
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        for (int j = s; j < ntot; j += 96) {
            double t[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int jj = j + 8 * u;
                const int tile = jj < n0 ? a.lo[0] + jj : a.lo[1] + (jj - n0);
                t[u] = jj < ntot ? a.p[size_t(tile) * NE_STRIDE + c] : 0.0;
            }
            v0 += t[0]; v1 += t[1]; v2 += t[2]; v3 += t[3];
            v0 += t[4]; v1 += t[5]; v2 += t[6]; v3 += t[7];
            v0 += t[8]; v1 += t[9]; v2 += t[10]; v3 += t[11];
        }
        v = (v0 + v1) + (v2 + v3);
    }
    scratch[s * 32 + c] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += scratch[q * 32 + c];
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
__device__ __noinline__ void jacobi6_mem(double *a, double *V, double *ev)
{
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) V[i * 6 + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dg = 0.0;
        for (int i = 0; i < 6; ++i) { dg += a[i * 6 + i] * a[i * 6 + i]; for (int j = i + 1; j < 6; ++j) off += a[i * 6 + j] * a[i * 6 + j]; }
        if (off <= 1e-32 * dg || off == 0.0) break;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double apq = a[p * 6 + q];
                if (apq == 0.0) continue;
                double theta = (a[q * 6 + q] - a[p * 6 + p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    double akp = a[k * 6 + p], akq = a[k * 6 + q];
                    a[k * 6 + p] = c * akp - s * akq;
                    a[k * 6 + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    double apk = a[p * 6 + k], aqk = a[q * 6 + k];
                    a[p * 6 + k] = c * apk - s * aqk;
                    a[q * 6 + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    double vkp = V[k * 6 + p], vkq = V[k * 6 + q];
                    V[k * 6 + p] = c * vkp - s * vkq;
                    V[k * 6 + q] = s * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 6; ++i) ev[i] = a[i * 6 + i];
    for (int i = 0; i < 5; ++i) {
        int k = i;
        for (int j = i + 1; j < 6; ++j) if (ev[j] < ev[k]) k = j;
        if (k != i) {
            double t = ev[i]; ev[i] = ev[k]; ev[k] = t;
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
    for (int i = 0; i < 6; ++i) for (int j = i; j < 6; ++j) { a[i * 6 + j] = ne[q]; a[j * 6 + i] = ne[q]; ++q; }
    jacobi6_mem(a, Vf, ev);
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
__device__ inline void gn_finish2(const double *ne, const double *cnt2, double *x, SolverState *S, double eig_thre, int freeze,
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


// ---------------------------------------------------------------- Levenberg-Marquardt (Ceres trust-region semantics)
// Bodies of the LM begin / step, run by ONE thread after the record has been summed into LDS. They are device functions so
// that both the stand-alone single-workgroup kernels (solver.hip: multi-GPU, tracker, good-feature paths) and the last-arriving
// workgroup of the linearisation kernels (match.hip: scan2map on one GPU -- no extra launch per LM iteration) can run them.
__device__ inline double gradient_max_norm(const SolverState *S)
{
    double ng[6], xp[7];
    for (int i = 0; i < 6; ++i) ng[i] = -S->ne[NE_G + i];
    pose_plus(S->x, ng, S->V, xp);
    double m = 0.0;
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(S->x[i] - xp[i]));
    return m;
}

__device__ inline void lm_propose(SolverState *S, int max_it)
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
__device__ inline void lm_begin_body(const double *ne, const double *cnt2, double *scratch, SolverState *S, double eig_thre, int max_it,
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
__device__ inline void lm_step_body(const double *ce, SolverState *S, int max_it)
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

}  // namespace mlh
