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
#include "solver_dev.hpp"

namespace mlh {

__global__ __launch_bounds__(256) void gn_update_kernel(SumArgs sa, SolverState *S, double eig_thre, IterStatDev *stat, int pre_reduced)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[8 * 32];
    gather_ne(sa, S, pre_reduced, ne, cnt2, scratch);
    if (threadIdx.x >= 2) return;
    gn_finish2<true>(ne, cnt2, S->x, S, eig_thre, 0, stat, scratch);
}

// multi-GPU pose-block mode: the per-block records arrive all-reduced in S->neb; every rank runs the identical per-block updates
struct BlockUpd { int n; double thre[8]; int freeze[8]; };
__global__ __launch_bounds__(256) void gn_update_blocks_kernel(SolverState *S, BlockUpd U, IterStatDev *stat)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[8 * 32];
    for (int b = 0; b < U.n; ++b) {
        if (threadIdx.x < NE_STRIDE) ne[threadIdx.x] = S->neb[b][threadIdx.x];
        __syncthreads();
        if (threadIdx.x == 0) { cnt2[0] = ne[NE_CNT + 1]; cnt2[1] = ne[NE_CNT + 2]; }
        __syncthreads();
        if (threadIdx.x < 2) gn_finish2<true>(ne, cnt2, b == 0 ? S->x : S->xb[b], b == 0 ? S : nullptr, U.thre[b], U.freeze[b], stat ? stat + b : nullptr, scratch);
        __syncthreads();
    }
}

// reduce only: S->ne <- sum of partials (used by the host-driven mlh_match_linearize / mlh_linearize)
__global__ __launch_bounds__(256) void reduce_only_kernel(SumArgs sa, SolverState *S, int to_ce)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[8 * 32];
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

__global__ __launch_bounds__(256) void lm_begin_kernel(SumArgs sa, SolverState *S, double eig_thre, int max_it, IterStatDev *stat, int pre_reduced,
                                                       int min_blocks)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[8 * 32];
    gather_ne(sa, S, pre_reduced, ne, cnt2, scratch);
    if (threadIdx.x != 0) return;
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

__global__ __launch_bounds__(256) void lm_step_kernel(SumArgs sa, SolverState *S, int max_it, int pre_reduced)
{
    __shared__ double ce[NE_STRIDE], cnt2[2], scratch[8 * 32];
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
static SumArgs make_sum_args(mlh_ctx *ctx)
{
    SumArgs sa;
    sa.p = ctx->partials.as<double>();
    sa.lo[0] = 0; sa.hi[0] = ctx->n_partial_tiles;
    sa.lo[1] = 0; sa.hi[1] = 0;
    return sa;
}

static IterStatDev *stat_ptr(mlh_ctx *ctx, int slot)
{
    return slot >= 0 ? ctx->stats.as<IterStatDev>() + slot : nullptr;
}

int reduce_only_launch(mlh_ctx *ctx, int to_ce)
{
    hipLaunchKernelGGL(reduce_only_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(), to_ce);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// multi-GPU: local reduce -> one all-reduce of the 32-double record -> the update kernel consumes the reduced record
static int pre_reduce(mlh_ctx *ctx, int to_ce, int &pre_reduced)
{
    pre_reduced = 0;
    if (!ctx->comm) return MLH_OK;
    int rc = reduce_only_launch(ctx, to_ce);
    if (rc) return rc;
    if ((rc = comm_allreduce_state(ctx, to_ce))) return rc;
    pre_reduced = 1;
    return MLH_OK;
}

int gn_update_prereduced_launch(mlh_ctx *ctx, double map_eig_thre, int stat_slot)
{
    prof_begin(ctx, MLH_K_SOLVE);
    hipLaunchKernelGGL(gn_update_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(),
                       map_eig_thre, stat_ptr(ctx, stat_slot), 1);
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int gn_update_blocks_prereduced_launch(mlh_ctx *ctx, int n_blocks, const double *eig_thre, const int *freeze, int stat_slot)
{
    BlockUpd U;
    U.n = n_blocks;
    for (int b = 0; b < 8; ++b) { U.thre[b] = b < n_blocks ? eig_thre[b] : 0.0; U.freeze[b] = b < n_blocks ? freeze[b] : 0; }
    prof_begin(ctx, MLH_K_SOLVE);
    hipLaunchKernelGGL(gn_update_blocks_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->state.as<SolverState>(), U, stat_ptr(ctx, stat_slot));
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_begin_launch(mlh_ctx *ctx, double map_eig_thre, int max_iterations, int stat_slot, int min_blocks)
{
    int pre = 0, rc = pre_reduce(ctx, 0, pre);
    if (rc) return rc;
    prof_begin(ctx, MLH_K_SOLVE);
    hipLaunchKernelGGL(lm_begin_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(),
                       map_eig_thre, max_iterations, stat_ptr(ctx, stat_slot), pre, min_blocks);
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
    hipLaunchKernelGGL(lm_step_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(), max_iterations, pre);
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
