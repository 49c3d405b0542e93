// TEST INFRASTRUCTURE: the part of Ceres 1.12's public interface the scan-to-map call sites touch (lidar_mapper_keyframe.cpp:440-452, 537-581), with Ceres'
// ownership rules: ceres::Problem OWNS the cost functions, loss functions and local parameterisations handed to it (Problem::Options defaults: TAKE_OWNERSHIP; a
// loss function shared by many residual blocks is deleted once) and deletes them in its destructor. Evaluate here is the plain per-block walk -- cost function,
// then the loss's corrector (rho'' <= 0 for Huber: residual and Jacobian rows scaled by sqrt(rho')), then the local parameterisation's 7 x 6 Jacobian -- enough to
// check what a façade factor returns THROUGH the interface Ceres calls it by; it is not a minimiser.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <memory>
#include <set>
#include <vector>
namespace ceres {
class CostFunction {
public:
    CostFunction() : num_residuals_(0) {}
    virtual ~CostFunction() {}
    virtual bool Evaluate(double const *const *parameters, double *residuals, double **jacobians) const = 0;
    const std::vector<int32_t> &parameter_block_sizes() const { return parameter_block_sizes_; }
    int num_residuals() const { return num_residuals_; }
protected:
    std::vector<int32_t> *mutable_parameter_block_sizes() { return &parameter_block_sizes_; }
    void set_num_residuals(int n) { num_residuals_ = n; }
private:
    std::vector<int32_t> parameter_block_sizes_;
    int num_residuals_;
};
template <int kNumResiduals, int... Ns>
class SizedCostFunction : public CostFunction {
public:
    SizedCostFunction()
    {
        set_num_residuals(kNumResiduals);
        const int sizes[] = {Ns...};
        for (int s : sizes) mutable_parameter_block_sizes()->push_back(s);
    }
    virtual ~SizedCostFunction() {}
};
class LossFunction {
public:
    virtual ~LossFunction() {}
    virtual void Evaluate(double sq_norm, double out[3]) const = 0;
};
class HuberLoss : public LossFunction {          // ceres/loss_function.cc
public:
    explicit HuberLoss(double a) : a_(a), b_(a * a) {}
    virtual void Evaluate(double s, double rho[3]) const
    {
        if (s > b_) { const double r = std::sqrt(s); rho[0] = 2.0 * a_ * r - b_; rho[1] = std::max(std::numeric_limits<double>::min(), a_ / r); rho[2] = -rho[1] / (2.0 * s); }
        else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
    }
private:
    const double a_, b_;
};
class LocalParameterization {
public:
    virtual ~LocalParameterization() {}
    virtual bool Plus(const double *x, const double *delta, double *x_plus_delta) const = 0;
    virtual bool ComputeJacobian(const double *x, double *jacobian) const = 0;
    virtual int GlobalSize() const = 0;
    virtual int LocalSize() const = 0;
};
struct CRSMatrix {
    int num_rows = 0, num_cols = 0;
    std::vector<int> cols, rows;
    std::vector<double> values;
};
namespace internal {
struct ResidualBlock {
    const CostFunction *cost_function;
    const LossFunction *loss_function;
    std::vector<double *> parameter_blocks;
};
}  // namespace internal
typedef internal::ResidualBlock *ResidualBlockId;
class Problem {
public:
    struct EvaluateOptions {
        std::vector<double *> parameter_blocks;
        std::vector<ResidualBlockId> residual_blocks;
        bool apply_loss_function = true;
    };
    Problem() {}
    Problem(const Problem &) = delete;
    Problem &operator=(const Problem &) = delete;
    ~Problem()
    {
        for (const CostFunction *c : costs_) delete c;
        for (const LossFunction *l : losses_) delete l;
        for (const LocalParameterization *p : params_owned_) delete p;
    }
    void AddParameterBlock(double *values, int size, LocalParameterization *lp = nullptr)
    {
        for (ParamBlock &b : blocks_) if (b.values == values) { if (lp) { b.lp = lp; params_owned_.insert(lp); } return; }
        blocks_.push_back(ParamBlock{values, size, lp});
        if (lp) params_owned_.insert(lp);
    }
    template <typename... Ps>
    ResidualBlockId AddResidualBlock(CostFunction *cost_function, LossFunction *loss_function, double *x0, Ps *...xs)
    {
        std::unique_ptr<internal::ResidualBlock> rb(new internal::ResidualBlock{cost_function, loss_function, {x0, xs...}});
        costs_.insert(cost_function);
        if (loss_function) losses_.insert(loss_function);
        const std::vector<int32_t> &sz = cost_function->parameter_block_sizes();
        for (size_t k = 0; k < rb->parameter_blocks.size(); ++k) AddParameterBlock(rb->parameter_blocks[k], k < sz.size() ? sz[k] : 0);
        residual_blocks_.push_back(std::move(rb));
        return residual_blocks_.back().get();
    }
    int NumResidualBlocks() const { return int(residual_blocks_.size()); }
    int NumParameterBlocks() const { return int(blocks_.size()); }
    // One residual block through the interface Ceres evaluates it by: r (loss-corrected when asked) and, per parameter block, the num_residuals x local_size
    // Jacobian (J_global * ComputeJacobian when the block has a parameterisation). 1-residual blocks only (every factor of this path).
    bool EvaluateBlock(ResidualBlockId id, bool apply_loss, double *residual, std::vector<std::vector<double>> *local_jacobians) const
    {
        const size_t nb = id->parameter_blocks.size();
        std::vector<std::vector<double>> Jg(nb);
        std::vector<double *> Jp(nb);
        const std::vector<int32_t> &sz = id->cost_function->parameter_block_sizes();
        for (size_t k = 0; k < nb; ++k) { Jg[k].assign(size_t(sz[k]), 0.0); Jp[k] = Jg[k].data(); }
        double r = 0.0;
        if (!id->cost_function->Evaluate(id->parameter_blocks.data(), &r, Jp.data())) return false;
        double scale = 1.0;
        if (apply_loss && id->loss_function) {
            double rho[3];
            id->loss_function->Evaluate(r * r, rho);
            scale = std::sqrt(rho[1]);
        }
        *residual = scale * r;
        if (local_jacobians) {
            local_jacobians->assign(nb, std::vector<double>());
            for (size_t k = 0; k < nb; ++k) {
                const LocalParameterization *lp = nullptr;
                for (const ParamBlock &b : blocks_) if (b.values == id->parameter_blocks[k]) lp = b.lp;
                std::vector<double> &out = (*local_jacobians)[k];
                if (!lp) { out = Jg[k]; for (double &v : out) v *= scale; continue; }
                const int g = lp->GlobalSize(), l = lp->LocalSize();
                std::vector<double> P(size_t(g) * l);
                lp->ComputeJacobian(id->parameter_blocks[k], P.data());
                out.assign(size_t(l), 0.0);
                for (int c = 0; c < l; ++c) { double s = 0.0; for (int a = 0; a < g; ++a) s += Jg[k][size_t(a)] * P[size_t(a) * l + c]; out[size_t(c)] = scale * s; }
            }
        }
        return true;
    }
    // ownership bookkeeping the tests look at
    size_t NumOwnedCostFunctions() const { return costs_.size(); }
    size_t NumOwnedLossFunctions() const { return losses_.size(); }
private:
    struct ParamBlock { double *values; int size; LocalParameterization *lp; };
    std::vector<ParamBlock> blocks_;
    std::vector<std::unique_ptr<internal::ResidualBlock>> residual_blocks_;
    std::set<const CostFunction *> costs_;
    std::set<const LossFunction *> losses_;
    std::set<const LocalParameterization *> params_owned_;
};
}  // namespace ceres
