// Device-resident optimiser steps for the scan-to-map path (single-workgroup kernels, f64).
// They consume the per-workgroup partial normal equations written by match.hip and keep the pose / trust-region state in
// HBM so that a Gauss-Newton or Levenberg-Marquardt iteration never needs a host round trip.
//
//   gn_update_kernel  : sum partials -> J^T J, J^T r -> evalDegenracy (lidar_mapper_keyframe.cpp:1172-1204) ->
//                       solve H d = -g -> PoseLocalParameterization::Plus (pose_local_parameterization.cpp:26-45)
//   lm_begin/lm_step  : Ceres 1.12 trust-region Levenberg-Marquardt semantics on one 6-dof block (Jacobi scaling,
//                       diag clamp [1e-6,1e32], radius 1e4, step acceptance by relative decrease > 1e-3,
//                       function/parameter/gradient tolerances 1e-6/1e-8/1e-10) -- what ceres::Solve does at
//                       lidar_mapper_keyframe.cpp:586-596 (DENSE_SCHUR on a single block == dense 6x6 solve).
#include "ctx.hpp"
#include "dev_math.hpp"

namespace mlh {

struct SumArgs {
    const double *p[2];
    int nb[2];
};

// 256 threads: column c = tid & 31, slice s = tid >> 5 sums tiles s, s+8, ...; slices combined in fixed order.
__device__ void sum_partials(const SumArgs &a, double *ne /*LDS, NE_STRIDE*/, double *cnt2 /*LDS, 2*/, double *scratch /*LDS 2*8*32*/)
{
    const int c = threadIdx.x & 31, s = threadIdx.x >> 5;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        double v = 0.0;
        if (a.p[k]) for (int b = s; b < a.nb[k]; b += 8) v += a.p[k][size_t(b) * NE_STRIDE + c];
        scratch[(k * 8 + s) * 32 + c] = v;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { t0 += scratch[(0 * 8 + q) * 32 + c]; t1 += scratch[(1 * 8 + q) * 32 + c]; }
        ne[c] = t0 + t1;
        if (c == NE_CNT) { cnt2[0] = t0; cnt2[1] = t1; }
    }
    __syncthreads();
    if (threadIdx.x == 0) { ne[NE_CNT + 1] = cnt2[0]; ne[NE_CNT + 2] = cnt2[1]; }
    __syncthreads();
}

// either sum this rank's partials, or (multi-GPU) take the already all-reduced record from the solver state
__device__ void gather_ne(const SumArgs &a, const SolverState *S, int pre_reduced, double *ne, double *cnt2, double *scratch)
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

// cyclic Jacobi, eigenvalues ascending, eigenvectors in the columns of V (row-major 6x6)
__device__ void jacobi6(const double *Hin, double *ev, double *V)
{
    double a[36];
    for (int i = 0; i < 36; ++i) a[i] = Hin[i];
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
__device__ __noinline__ bool eval_degeneracy_dev(const double *H, double thre, double *ev, double *Vupd)
{
    double Vf[36];
    jacobi6(H, ev, Vf);
    bool keep[6];
    bool deg = false, stop = false;
    for (int j = 0; j < 6; ++j) {
        if (!stop && ev[j] < thre) { keep[j] = false; deg = true; }
        else { keep[j] = true; stop = true; }
    }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double s = 0.0;
            if (deg) { for (int j = 0; j < 6; ++j) if (keep[j]) s += Vf[r * 6 + j] * Vf[c * 6 + j]; }
            else s = (r == c) ? 1.0 : 0.0;
            Vupd[r * 6 + c] = s;
        }
    return deg;
}

// Cholesky factor / solve of a 6x6 SPD system, fully unrolled so that A, L, y live in registers (no scratch).
__device__ __forceinline__ bool chol6_factor(const double (&A)[36], double (&L)[36])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double s = A[j * 6 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[j * 6 + k] * L[j * 6 + k];
        ok = ok && (s > 0.0);
        const double ljj = sqrt(s);
        L[j * 6 + j] = ljj;
#pragma unroll
        for (int i = j + 1; i < 6; ++i) {
            double t = A[i * 6 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[i * 6 + k] * L[j * 6 + k];
            L[i * 6 + j] = t / ljj;
        }
    }
    return ok;
}

__device__ __forceinline__ bool chol6_solve(const double (&A)[36], const double (&b)[6], double (&x)[6])
{
    double L[36];
    if (!chol6_factor(A, L)) return false;
    double y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double s = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) s -= L[i * 6 + k] * y[k];
        y[i] = s / L[i * 6 + i];
    }
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; ++k) s -= L[k * 6 + i] * x[k];
        x[i] = s / L[i * 6 + i];
    }
    return true;
}

// evalDegenracy with a fast path: H - thre*I positive definite  <=>  lambda_min > thre  =>  nothing is degenerate and
// V_update = I; the eigen-decomposition is only run when that test fails or when the eigenvalues are wanted for the
// per-iteration record (the reference logs them, lidar_mapper_keyframe.cpp:1190-1193).
__device__ bool degeneracy(const double (&H)[36], double thre, bool need_eig, double (&ev)[6], double (&Vupd)[36])
{
    if (!need_eig) {
        double A[36], L[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) A[i] = H[i];
        const double sh = thre * (1.0 + 1e-9);
#pragma unroll
        for (int i = 0; i < 6; ++i) A[i * 6 + i] -= sh;
        if (chol6_factor(A, L)) {
#pragma unroll
            for (int i = 0; i < 36; ++i) Vupd[i] = ((i % 7) == 0) ? 1.0 : 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) ev[i] = 0.0;
            return false;
        }
    }
    return eval_degeneracy_dev(H, thre, ev, Vupd);
}

__device__ void write_stat_common(IterStatDev *st, const double *ne, const double *cnt2, const double *H, const double *ev, bool deg)
{
    st->n_surf = int(cnt2[0] + 0.5);
    st->n_corner = int(cnt2[1] + 0.5);
    st->is_degenerate = deg ? 1 : 0;
    st->cost = ne[NE_COST];
    for (int i = 0; i < 6; ++i) { st->eigval[i] = ev[i]; st->g[i] = ne[NE_G + i]; }
    for (int i = 0; i < 36; ++i) st->H[i] = H[i];
}

__global__ __launch_bounds__(256) void gn_update_kernel(SumArgs sa, SolverState *S, double eig_thre, IterStatDev *stat, int pre_reduced)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[2 * 8 * 32];
    gather_ne(sa, S, pre_reduced, ne, cnt2, scratch);
    if (threadIdx.x != 0) return;
    double H[36], ev[6], V[36];
    unpack_H(ne, H);
    const bool deg = degeneracy(H, eig_thre, stat != nullptr, ev, V);
    double rhs[6], d[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) rhs[i] = -ne[NE_G + i];
    bool ok = chol6_solve(H, rhs, d);
    if (!ok) {
        double Hd[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) Hd[i] = H[i] + (((i % 7) == 0) ? 1e-6 : 0.0);
        ok = chol6_solve(Hd, rhs, d);
    }
    if (ok) {
        double xc[7], xn[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) xc[i] = S->x[i];
        pose_plus(xc, d, V, xn);
#pragma unroll
        for (int i = 0; i < 7; ++i) S->x[i] = xn[i];
    }
    for (int i = 0; i < NE_STRIDE; ++i) S->ne[i] = ne[i];
    for (int i = 0; i < 36; ++i) S->V[i] = V[i];
    if (stat) {
        write_stat_common(stat, ne, cnt2, H, ev, deg);
        stat->final_cost = ne[NE_COST];
        stat->lm_iterations = 0; stat->successful_steps = 0; stat->termination = 0;
        for (int i = 0; i < 7; ++i) stat->pose_after[i] = S->x[i];
    }
}

// reduce only: S->ne <- sum of partials (used by the host-driven mlh_match_linearize / mlh_linearize)
__global__ __launch_bounds__(256) void reduce_only_kernel(SumArgs sa, SolverState *S, int to_ce)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[2 * 8 * 32];
    sum_partials(sa, ne, cnt2, scratch);
    if (threadIdx.x < NE_STRIDE) (to_ce ? S->ce : S->ne)[threadIdx.x] = ne[threadIdx.x];
}

// ---------------------------------------------------------------- Levenberg-Marquardt (Ceres trust-region semantics)
__device__ double gradient_max_norm(const SolverState *S)
{
    double ng[6], xp[7];
    for (int i = 0; i < 6; ++i) ng[i] = -S->ne[NE_G + i];
    pose_plus(S->x, ng, S->V, xp);
    double m = 0.0;
    for (int i = 0; i < 7; ++i) m = fmax(m, fabs(S->x[i] - xp[i]));
    return m;
}

__device__ void lm_propose(SolverState *S, int max_it)
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

__global__ __launch_bounds__(256) void lm_begin_kernel(SumArgs sa, SolverState *S, double eig_thre, int max_it, IterStatDev *stat, int pre_reduced)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[2 * 8 * 32];
    gather_ne(sa, S, pre_reduced, ne, cnt2, scratch);
    if (threadIdx.x != 0) return;
    double H[36], ev[6], V[36];
    unpack_H(ne, H);
    bool deg = eval_degeneracy_dev(H, eig_thre, ev, V);
    for (int i = 0; i < NE_STRIDE; ++i) S->ne[i] = ne[i];
    for (int i = 0; i < 36; ++i) S->V[i] = V[i];
    for (int i = 0; i < 6; ++i) S->S[i] = 1.0 / (1.0 + sqrt(H[i * 6 + i]));
    S->radius = 1e4; S->decrease_factor = 2.0; S->reuse_diagonal = 0;
    S->iteration = 0; S->done = 0; S->termination = 0; S->num_successful = 0; S->num_invalid = 0; S->evaluations = 1;
    S->gmax = gradient_max_norm(S);
    if (stat) {
        write_stat_common(stat, ne, cnt2, H, ev, deg);
        stat->final_cost = ne[NE_COST];
    }
    lm_propose(S, max_it);
}

__global__ __launch_bounds__(256) void lm_step_kernel(SumArgs sa, SolverState *S, int max_it, int pre_reduced)
{
    __shared__ double ce[NE_STRIDE], cnt2[2], scratch[2 * 8 * 32];
    if (S->done) return;
    if (pre_reduced) {
        if (threadIdx.x < NE_STRIDE) ce[threadIdx.x] = S->ce[threadIdx.x];
        __syncthreads();
    } else {
        sum_partials(sa, ce, cnt2, scratch);
    }
    if (threadIdx.x != 0) return;
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

__global__ void lm_finish_kernel(const SolverState *S, IterStatDev *stat)
{
    if (threadIdx.x != 0 || !stat) return;
    stat->lm_iterations = S->iteration;
    stat->successful_steps = S->num_successful;
    stat->termination = S->termination;
    stat->final_cost = S->ne[NE_COST];
    for (int i = 0; i < 7; ++i) stat->pose_after[i] = S->x[i];
}

// ---------------------------------------------------------------- host launchers
static SumArgs make_sum_args(mlh_ctx *ctx, int kind_mask)
{
    SumArgs sa;
    for (int k = 0; k < 2; ++k) {
        const FeatSet &fs = ctx->feat[k];
        bool use = (kind_mask & (1 << k)) && fs.m > 0 && fs.n_blocks > 0 && fs.partials.p;
        sa.p[k] = use ? fs.partials.as<double>() : nullptr;
        sa.nb[k] = use ? fs.n_blocks : 0;
    }
    return sa;
}

static IterStatDev *stat_ptr(mlh_ctx *ctx, int slot)
{
    return slot >= 0 ? ctx->stats.as<IterStatDev>() + slot : nullptr;
}

int reduce_only_launch(mlh_ctx *ctx, int kind_mask, int to_ce)
{
    hipLaunchKernelGGL(reduce_only_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx, kind_mask), ctx->state.as<SolverState>(), to_ce);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// multi-GPU: local reduce -> one all-reduce of the 32-double record -> the update kernel consumes the reduced record
static int pre_reduce(mlh_ctx *ctx, int to_ce, int &pre_reduced)
{
    pre_reduced = 0;
    if (!ctx->comm) return MLH_OK;
    int rc = reduce_only_launch(ctx, 3, to_ce);
    if (rc) return rc;
    if ((rc = comm_allreduce_state(ctx, to_ce))) return rc;
    pre_reduced = 1;
    return MLH_OK;
}

int gn_update_launch(mlh_ctx *ctx, double map_eig_thre, int stat_slot)
{
    int pre = 0, rc = pre_reduce(ctx, 0, pre);
    if (rc) return rc;
    prof_begin(ctx, MLH_K_SOLVE);
    hipLaunchKernelGGL(gn_update_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx, 3), ctx->state.as<SolverState>(),
                       map_eig_thre, stat_ptr(ctx, stat_slot), pre);
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_begin_launch(mlh_ctx *ctx, double map_eig_thre, int max_iterations, int stat_slot)
{
    int pre = 0, rc = pre_reduce(ctx, 0, pre);
    if (rc) return rc;
    prof_begin(ctx, MLH_K_SOLVE);
    hipLaunchKernelGGL(lm_begin_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx, 3), ctx->state.as<SolverState>(),
                       map_eig_thre, max_iterations, stat_ptr(ctx, stat_slot), pre);
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_step_launch(mlh_ctx *ctx, int max_iterations, int stat_slot)
{
    (void)stat_slot;
    int pre = 0, rc = pre_reduce(ctx, 1, pre);
    if (rc) return rc;
    prof_begin(ctx, MLH_K_SOLVE);
    hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx, 3), ctx->state.as<SolverState>(), max_iterations, pre);
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_finish_launch(mlh_ctx *ctx, int stat_slot)
{
    hipLaunchKernelGGL(lm_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->state.as<SolverState>(), stat_ptr(ctx, stat_slot));
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

}  // namespace mlh
