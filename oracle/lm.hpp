// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (Ceres absent: restated from its published algorithm).
//
// ceres::Solve on ONE 6-dof pose block (and, with the dynamic-size entry, on the odometry window's blocks) -- trust-region Levenberg-Marquardt, DENSE_SCHUR, Jacobi scaling, Ceres 1.12.0's defaults
// (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc) -- as a template over "evaluate the loss-corrected normal
// equations at x" and "x (+) delta", so that the SAME iteration drives both the oracle's restated residual blocks (mapper.cpp) and, in oracle/_ref, the
// Ceres-shaped shim under the reference's own scan2MapOptimization / trackCloud lines (oracle/ref/ref_shim.cpp).
#pragma once
#include "linalg.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <utility>
#include <vector>

namespace orc {

struct NormalEq {
    double H[36];   // J^T J, row-major 6x6 (loss-corrected, weighted rows)
    double g[6];    // J^T r
    double cost;    // sum 0.5 * rho(r^2)
    int n;
};

struct SolveSummary {
    int num_iterations = 0;           // index of the last iteration (Ceres counts iteration 0)
    int num_successful_steps = 0;
    int num_evaluations = 0;          // residual(+jacobian) evaluations
    double initial_cost = 0, final_cost = 0;
    int termination = 0;              // 0 no-convergence (max iters), 1 gradient tol, 2 parameter tol, 3 function tol, 4 failure
};

// The iteration for a problem of n local dimensions over nx stored values (one 6-dof pose block: n = 6, nx = 7; the odometry window: the free pose and
// extrinsic blocks side by side). evaluate(const double *x, NormalEqDyn &ne) fills H (n x n row-major), g (n), cost; plus(x, delta, out) is x (+) delta.
struct NormalEqDyn {
    std::vector<double> H, g;
    double cost = 0.0;
    int count = 0;
};

template <typename Evaluate, typename Plus>
void ceres_like_solve_dyn(Evaluate &&evaluate, Plus &&plus, double *x, int n, int nx, int max_num_iterations, SolveSummary &sum)
{
    sum = SolveSummary();
    const double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    const double min_relative_decrease = 1e-3;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int num_consecutive_invalid = 0;
    const size_t N = size_t(n), NX = size_t(nx);

    NormalEqDyn ne, ce;
    ne.H.assign(N * N, 0.0); ne.g.assign(N, 0.0);
    evaluate(x, ne);
    sum.num_evaluations++;
    sum.initial_cost = sum.final_cost = ne.cost;
    std::vector<double> S(N), neg_g(N), xp(NX), diag(N, 0.0), A(N * N), gs(N), lhs(N * N), y(N), step(N), delta(N), cand(NX);
    for (size_t i = 0; i < N; ++i) S[i] = 1.0 / (1.0 + std::sqrt(ne.H[i * N + i]));

    auto gradient_max_norm = [&](const NormalEqDyn &e) {
        for (size_t i = 0; i < N; ++i) neg_g[i] = -e.g[i];
        plus(x, neg_g.data(), xp.data());
        double m = 0.0;
        for (size_t i = 0; i < NX; ++i) m = std::max(m, std::fabs(x[i] - xp[i]));
        return m;
    };
    double gmax = gradient_max_norm(ne);
    int iteration = 0;
    while (true) {
        if (iteration >= max_num_iterations) { sum.termination = 0; break; }
        if (gmax <= gradient_tolerance) { sum.termination = 1; break; }
        if (radius <= min_radius) { sum.termination = 4; break; }
        iteration++;
        sum.num_iterations = iteration;

        for (size_t r = 0; r < N; ++r) {
            gs[r] = S[r] * ne.g[r];
            for (size_t c = 0; c < N; ++c) A[r * N + c] = S[r] * ne.H[r * N + c] * S[c];
        }
        if (!reuse_diagonal)
            for (size_t i = 0; i < N; ++i) diag[i] = std::min(std::max(A[i * N + i], min_lm_diagonal), max_lm_diagonal);
        lhs = A;
        for (size_t i = 0; i < N; ++i) lhs[i * N + i] += diag[i] / radius;
        bool ok = chol_solve_d(lhs.data(), gs.data(), n, y.data());
        reuse_diagonal = true;
        bool step_valid = false;
        double model_cost_change = 0.0;
        if (ok) {
            for (size_t i = 0; i < N; ++i) step[i] = -y[i];
            double sg = 0.0, sAs = 0.0;
            for (size_t r = 0; r < N; ++r) {
                sg += step[r] * gs[r];
                double t = 0.0;
                for (size_t c = 0; c < N; ++c) t += A[r * N + c] * step[c];
                sAs += step[r] * t;
            }
            model_cost_change = -(sg + 0.5 * sAs);
            step_valid = model_cost_change > 0.0;
        }
        if (!step_valid) {
            if (++num_consecutive_invalid >= 5) { sum.termination = 4; break; }
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            continue;
        }
        num_consecutive_invalid = 0;
        for (size_t i = 0; i < N; ++i) delta[i] = step[i] * S[i];
        plus(x, delta.data(), cand.data());
        ce.H.assign(N * N, 0.0); ce.g.assign(N, 0.0); ce.cost = 0.0; ce.count = 0;
        evaluate(cand.data(), ce);   // Ceres evaluates cost only here and J after acceptance
        sum.num_evaluations++;
        double step_norm = 0.0, x_norm = 0.0;
        for (size_t i = 0; i < NX; ++i) { step_norm += (x[i] - cand[i]) * (x[i] - cand[i]); x_norm += x[i] * x[i]; }
        step_norm = std::sqrt(step_norm); x_norm = std::sqrt(x_norm);
        if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { sum.termination = 2; break; }
        double cost_change = ne.cost - ce.cost;
        if (std::fabs(cost_change) <= function_tolerance * ne.cost) { sum.termination = 3; break; }
        double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > min_relative_decrease) {
            std::memcpy(x, cand.data(), sizeof(double) * NX);
            std::swap(ne, ce);
            sum.num_successful_steps++;
            sum.final_cost = ne.cost;
            radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
            radius = std::min(max_radius, radius);
            decrease_factor = 2.0;
            reuse_diagonal = false;
            gmax = gradient_max_norm(ne);
        } else {
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
        }
    }
}

// ONE 6-dof pose block: evaluate(const double x[7], NormalEq &ne); plus(const double x[7], const double delta[6], double out[7])
template <typename Evaluate, typename Plus>
void ceres_like_solve_generic(Evaluate &&evaluate, Plus &&plus, double x[7], int max_num_iterations, SolveSummary &sum)
{
    ceres_like_solve_dyn([&](const double *at, NormalEqDyn &d) {
                             NormalEq ne;
                             evaluate(at, ne);
                             std::memcpy(d.H.data(), ne.H, sizeof(ne.H));
                             std::memcpy(d.g.data(), ne.g, sizeof(ne.g));
                             d.cost = ne.cost; d.count = ne.n;
                         },
                         plus, x, 6, 7, max_num_iterations, sum);
}

}  // namespace orc
