// TEST INFRASTRUCTURE ONLY (see oracle/linalg.hpp header). PARITY UNPINNED (Ceres absent: restated from its published algorithm).
//
// ceres::Solve on ONE 6-dof pose block -- trust-region Levenberg-Marquardt, DENSE_SCHUR, Jacobi scaling, Ceres 1.12.0's defaults
// (trust_region_minimizer.cc, levenberg_marquardt_strategy.cc) -- as a template over "evaluate the loss-corrected normal
// equations at x" and "x (+) delta", so that the SAME iteration drives both the oracle's restated residual blocks (mapper.cpp) and, in oracle/_ref, the
// Ceres-shaped shim under the reference's own scan2MapOptimization / trackCloud lines (oracle/ref/ref_shim.cpp).
#pragma once
#include "linalg.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>

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

// evaluate(const double x[7], NormalEq &ne); plus(const double x[7], const double delta[6], double out[7])
template <typename Evaluate, typename Plus>
void ceres_like_solve_generic(Evaluate &&evaluate, Plus &&plus, double x[7], int max_num_iterations, SolveSummary &sum)
{
    sum = SolveSummary();
    const double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
    const double min_relative_decrease = 1e-3;
    const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
    const double max_radius = 1e16, min_radius = 1e-32;
    double radius = 1e4, decrease_factor = 2.0;
    bool reuse_diagonal = false;
    int num_consecutive_invalid = 0;

    NormalEq ne;
    evaluate(x, ne);
    sum.num_evaluations++;
    sum.initial_cost = sum.final_cost = ne.cost;
    double S[6];
    for (int i = 0; i < 6; ++i) S[i] = 1.0 / (1.0 + std::sqrt(ne.H[i * 6 + i]));

    auto gradient_max_norm = [&](const NormalEq &e) {
        double neg_g[6], xp[7];
        for (int i = 0; i < 6; ++i) neg_g[i] = -e.g[i];
        plus(x, neg_g, xp);
        double m = 0.0;
        for (int i = 0; i < 7; ++i) m = std::max(m, std::fabs(x[i] - xp[i]));
        return m;
    };
    double gmax = gradient_max_norm(ne);
    double diag[6] = {0, 0, 0, 0, 0, 0};
    int iteration = 0;
    while (true) {
        if (iteration >= max_num_iterations) { sum.termination = 0; break; }
        if (gmax <= gradient_tolerance) { sum.termination = 1; break; }
        if (radius <= min_radius) { sum.termination = 4; break; }
        iteration++;
        sum.num_iterations = iteration;

        double A[36], gs[6];
        for (int r = 0; r < 6; ++r) {
            gs[r] = S[r] * ne.g[r];
            for (int c = 0; c < 6; ++c) A[r * 6 + c] = S[r] * ne.H[r * 6 + c] * S[c];
        }
        if (!reuse_diagonal)
            for (int i = 0; i < 6; ++i) diag[i] = std::min(std::max(A[i * 6 + i], min_lm_diagonal), max_lm_diagonal);
        double lhs[36];
        std::memcpy(lhs, A, sizeof(lhs));
        for (int i = 0; i < 6; ++i) lhs[i * 6 + i] += diag[i] / radius;
        double y[6], step[6];
        bool ok = chol_solve_d(lhs, gs, 6, y);
        reuse_diagonal = true;
        bool step_valid = false;
        double model_cost_change = 0.0;
        if (ok) {
            for (int i = 0; i < 6; ++i) step[i] = -y[i];
            double sg = 0.0, sAs = 0.0;
            for (int r = 0; r < 6; ++r) {
                sg += step[r] * gs[r];
                double t = 0.0;
                for (int c = 0; c < 6; ++c) t += A[r * 6 + c] * step[c];
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
        double delta[6], cand[7];
        for (int i = 0; i < 6; ++i) delta[i] = step[i] * S[i];
        plus(x, delta, cand);
        NormalEq ce;
        evaluate(cand, ce);   // Ceres evaluates cost only here and J after acceptance
        sum.num_evaluations++;
        double step_norm = 0.0, x_norm = 0.0;
        for (int i = 0; i < 7; ++i) { step_norm += (x[i] - cand[i]) * (x[i] - cand[i]); x_norm += x[i] * x[i]; }
        step_norm = std::sqrt(step_norm); x_norm = std::sqrt(x_norm);
        if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { sum.termination = 2; break; }
        double cost_change = ne.cost - ce.cost;
        if (std::fabs(cost_change) <= function_tolerance * ne.cost) { sum.termination = 3; break; }
        double relative_decrease = cost_change / model_cost_change;
        if (relative_decrease > min_relative_decrease) {
            std::memcpy(x, cand, sizeof(double) * 7);
            ne = ce;
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

}  // namespace orc
