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

// ---------------------------------------------------------------- Levenberg-Marquardt kernels (bodies in solver_dev.hpp)
struct InitPose { int use; double x[7]; };   // the pose this solve starts from, when it has not been stored in the state yet
__global__ __launch_bounds__(256) void lm_begin_kernel(SumArgs sa, SolverState *S, double eig_thre, int max_it, IterStatDev *stat, int pre_reduced,
                                                       int min_blocks, InitPose ip)
{
    __shared__ double ne[NE_STRIDE], cnt2[2], scratch[8 * 32];
    if (ip.use && threadIdx.x < 7) S->x[threadIdx.x] = ip.x[threadIdx.x];     // ordered before the body by gather_ne's barriers
    gather_ne(sa, S, pre_reduced, ne, cnt2, scratch);
    if (threadIdx.x != 0) return;
    lm_begin_body(ne, cnt2, scratch, S, eig_thre, max_it, stat, min_blocks);
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
    lm_step_body(ce, S, max_it);
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
    MLH_LAUNCH(reduce_only_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(), to_ce);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

// multi-GPU: local reduce -> one all-reduce of the 32-double record -> the update kernel consumes the reduced record
static int pre_reduce(mlh_ctx *ctx, int to_ce, int &pre_reduced)
{
    pre_reduced = 0;
    if (!distributed(ctx)) return MLH_OK;
    int rc = reduce_only_launch(ctx, to_ce);
    if (rc) return rc;
    if ((rc = comm_allreduce_state(ctx, to_ce))) return rc;
    pre_reduced = 1;
    return MLH_OK;
}

int gn_update_prereduced_launch(mlh_ctx *ctx, double map_eig_thre, int stat_slot)
{
    prof_begin(ctx, MLH_K_SOLVE);
    MLH_LAUNCH(gn_update_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(),
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
    MLH_LAUNCH(gn_update_blocks_kernel, dim3(1), dim3(256), 0, ctx->stream, ctx->state.as<SolverState>(), U, stat_ptr(ctx, stat_slot));
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_begin_launch(mlh_ctx *ctx, double map_eig_thre, int max_iterations, int stat_slot, int min_blocks, const double *init_pose)
{
    int pre = 0, rc = pre_reduce(ctx, 0, pre);
    if (rc) return rc;
    InitPose ip;
    ip.use = init_pose ? 1 : 0;
    for (int i = 0; i < 7; ++i) ip.x[i] = init_pose ? init_pose[i] : 0.0;
    prof_begin(ctx, MLH_K_SOLVE);
    MLH_LAUNCH(lm_begin_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(),
                       map_eig_thre, max_iterations, stat_ptr(ctx, stat_slot), pre, min_blocks, ip);
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
    MLH_LAUNCH(lm_step_kernel, dim3(1), dim3(256), 0, ctx->stream, make_sum_args(ctx), ctx->state.as<SolverState>(), max_iterations, pre);
    prof_end(ctx, MLH_K_SOLVE);
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

int lm_finish_launch(mlh_ctx *ctx, int stat_slot)
{
    MLH_LAUNCH(lm_finish_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->state.as<SolverState>(), stat_ptr(ctx, stat_slot));
    MLH_HIP(ctx, hipGetLastError());
    return MLH_OK;
}

}  // namespace mlh
