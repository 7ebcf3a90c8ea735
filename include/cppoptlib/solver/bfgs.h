// cppoptlib/solver/bfgs.h — dense BFGS on the MI355X engine.
//
// Drop-in for the reference's solver/bfgs.h: `Bfgs<FunctionType, LineSearch>` (:39-44), quasi-Newton
// with an explicit inverse-Hessian approximation (InitializeSolver :65-71, OptimizationStep :73-137)
// under Solver::Minimize (solver/solver.h:181-224).  Every start state is one problem of a batch solved
// by the device kernel (csrc/lbfgs_kernel.hpp, ALG = dense BFGS; the n x n matrix lives in LDS, n <= 64)
// through mi355_bfgs_minimize_batch_host.  No CPU fallback: the function type needs a device twin.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_BFGS_H_
#define INCLUDE_CPPOPTLIB_SOLVER_BFGS_H_

#include <memory>
#include <tuple>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "../linesearch/more_thuente.h"
#include "../mi355/context.h"
#include "solver.h"

namespace cppoptlib::solver {

template <typename FunctionType, template <class, int> class LineSearch = linesearch::MoreThuente>
class Bfgs : public Solver<FunctionType, cppoptlib::function::FunctionState<typename FunctionType::ScalarType,
                                                                            FunctionType::Dimension>> {
  static_assert(FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::First ||
                    FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second,
                "Bfgs only supports first- or second-order differentiable functions");
  static_assert(std::is_same<typename FunctionType::ScalarType, double>::value,
                "the MI355X engine computes in fp64 (ScalarType must be double)");
  static_assert(cppoptlib::mi355::HasDeviceObjective<FunctionType>::value,
                "FunctionType has no device twin (kDeviceObjective / DeviceParams, see "
                "cppoptlib/mi355/objectives.h); the MI355X engine has no CPU fallback");
  static_assert(!cppoptlib::mi355::HasPerProblemData<FunctionType>::value,
                "the device Bfgs kernel is built for objectives without per-problem data");

 public:
  using StateType =
      cppoptlib::function::FunctionState<typename FunctionType::ScalarType, FunctionType::Dimension>;
  using Superclass = Solver<FunctionType, StateType>;
  using ProgressType = typename Superclass::ProgressType;
  using ScalarType = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  using MatrixType = typename FunctionType::MatrixType;

  using Superclass::Superclass;

  void SetContext(std::shared_ptr<cppoptlib::mi355::Context> ctx) { ctx_ = std::move(ctx); }

  std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                               const StateType& function_state) override {
    this->step_callback_(function, StateType(function, function_state.x), ProgressType());
    const std::vector<StateType> one{function_state};
    auto out = MinimizeBatch(function, one);
    this->step_callback_(function, std::get<0>(out[0]), std::get<1>(out[0]));
    return out[0];
  }

  // Solves every start state independently in one kernel launch.
  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const FunctionType& function,
                                                                 const std::vector<StateType>& states) {
    std::vector<std::tuple<StateType, ProgressType>> result;
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return result;
    const int n = static_cast<int>(states[0].x.size());
    std::vector<double> x0(static_cast<size_t>(B) * n), x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    for (int64_t b = 0; b < B; ++b) {
      if (static_cast<int>(states[b].x.size()) != n) cppoptlib::mi355::Fail("MinimizeBatch: mixed dimensions");
      for (int i = 0; i < n; ++i) x0[static_cast<size_t>(b) * n + i] = states[b].x[i];
    }
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    const std::vector<double> params = function.DeviceParams();
    mi355_lbfgs_desc d{};
    d.objective = FunctionType::kDeviceObjective;
    d.linesearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;
    d.n = n;
    d.m = 1;  // not used by Bfgs
    d.objective_params = params.empty() ? nullptr : params.data();
    d.n_params = static_cast<int32_t>(params.size());
    d.stop = this->stopping_progress.ToDeviceStop();
    cppoptlib::mi355::Check(
        mi355_bfgs_minimize_batch_host(ctx_->get(), &d, B, x0.data(), x.data(), f.data(), g.data(), prog.data()),
        "mi355_bfgs_minimize_batch_host");
    result.reserve(static_cast<size_t>(B));
    for (int64_t b = 0; b < B; ++b) {
      VectorType xv(n), gv(n);
      for (int i = 0; i < n; ++i) {
        xv[i] = x[static_cast<size_t>(b) * n + i];
        gv[i] = g[static_cast<size_t>(b) * n + i];
      }
      result.emplace_back(StateType(std::move(xv), f[static_cast<size_t>(b)], std::move(gv)),
                          ProgressType::FromDevice(prog[static_cast<size_t>(b)]));
    }
    return result;
  }

 private:
  std::shared_ptr<cppoptlib::mi355::Context> ctx_;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_BFGS_H_
