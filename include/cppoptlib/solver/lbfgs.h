// cppoptlib/solver/lbfgs.h — L-BFGS on the MI355X, cppoptlib-shaped.
//
// Drop-in for the reference's solver/lbfgs.h: same class template
//   Lbfgs<FunctionType, m = 10, LineSearch = linesearch::MoreThuente>   (lbfgs.h:40-45)
// same StateType / ProgressType aliases, same Minimize signature and return
// type.  The solve itself — two-loop recursion with the (s, y) ring
// (lbfgs.h:145-196, :248-298), Moré–Thuente line search, stopping tests — runs
// in one HIP kernel behind the C-ABI of include/mi355_lbfgs.h.
//
// New here: MinimizeBatch — the reference is one-problem-at-a-time; a GPU wants
// thousands of independent problems per launch.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_LBFGS_H_
#define INCLUDE_CPPOPTLIB_SOLVER_LBFGS_H_

#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "../linesearch/more_thuente.h"
#include "../mi355/context.h"
#include "solver.h"

namespace cppoptlib::solver {

template <typename FunctionType, int m = 10, template <class, int> class LineSearch = linesearch::MoreThuente>
class Lbfgs : public Solver<FunctionType, cppoptlib::function::FunctionState<typename FunctionType::ScalarType,
                                                                             FunctionType::Dimension>> {
  static_assert(FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::First ||
                    FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second,
                "L-BFGS only supports first- or second-order differentiable functions");
  static_assert(std::is_same<typename FunctionType::ScalarType, double>::value,
                "the MI355X engine computes in fp64 (ScalarType must be double)");
  static_assert(cppoptlib::mi355::HasDeviceObjective<FunctionType>::value,
                "FunctionType has no device twin (kDeviceObjective / DeviceParams, see "
                "cppoptlib/mi355/objectives.h); the MI355X engine has no CPU fallback");
  static_assert(m >= 1 && m <= MI355_LBFGS_MAX_M, "history size m out of range");

 public:
  using StateType =
      cppoptlib::function::FunctionState<typename FunctionType::ScalarType, FunctionType::Dimension>;
  using Superclass = Solver<FunctionType, StateType>;
  using ProgressType = typename Superclass::ProgressType;
  using ScalarType = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  using MatrixType = typename FunctionType::MatrixType;

  static constexpr int kHistorySize = m;
  static constexpr int kLineSearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;

  using Superclass::Superclass;

  // Engine context (device 0 by default); set before the first Minimize to pick a GPU.
  void SetContext(std::shared_ptr<cppoptlib::mi355::Context> ctx) { ctx_ = std::move(ctx); }

  // Lbfgs::Minimize of the reference (solver/solver.h:181-224): only `.x` of the
  // incoming state is used, value and gradient are evaluated at x0 (:189-192).
  std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                               const StateType& function_state) override {
    this->step_callback_(function, StateType(function, function_state.x), ProgressType());
    std::vector<StateType> one{function_state};
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
    MinimizeBatchRaw(function, n, B, x0.data(), x.data(), f.data(), g.data(), prog.data());
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

  // Batch-major host arrays in and out: x0[B][n] -> x[B][n], f[B], g[B][n], progress[B].
  void MinimizeBatchRaw(const FunctionType& function, int n, int64_t B, const double* x0, double* x,
                        double* f, double* g, mi355_lbfgs_progress* progress) {
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    // (functions whose parameter blob depends on the dimension, e.g. the augmented-Lagrangian composite of
    //  function_penalty.h, take n)
    std::vector<double> params;
    if constexpr (cppoptlib::mi355::HasDeviceParamsOfDimension<FunctionType>::value) {
      params = function.DeviceParams(n);
    } else {
      params = function.DeviceParams();
    }
    mi355_lbfgs_desc d{};
    d.objective = FunctionType::kDeviceObjective;
    d.linesearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;
    d.n = n;
    d.m = m;
    d.objective_params = params.empty() ? nullptr : params.data();
    d.n_params = static_cast<int32_t>(params.size());
    // objectives with per-problem data: this function object describes ONE problem, so its
    // data row is replicated for every start state of the batch
    std::vector<double> per_problem;
    d.per_problem_data = nullptr;
    d.per_problem_stride = 0;
    if constexpr (cppoptlib::mi355::HasPerProblemData<FunctionType>::value) {
      const std::vector<double> row = function.DevicePerProblem();
      per_problem.reserve(row.size() * static_cast<size_t>(B));
      for (int64_t b = 0; b < B; ++b) per_problem.insert(per_problem.end(), row.begin(), row.end());
      d.per_problem_data = per_problem.data();
      d.per_problem_stride = static_cast<int32_t>(row.size());
    }
    d.lanes_per_problem = 0;
    d.elems_per_lane = 0;
    d.history_placement = MI355_HISTORY_AUTO;
    // lbfgs.h:116-139 of the reference: Second-mode functions get the diagonal preconditioner
    std::vector<double> hessian_diagonal;
    d.hessian_diagonal = nullptr;
    if constexpr (FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second) {
      hessian_diagonal = function.DeviceHessianDiagonal();
      if (static_cast<int>(hessian_diagonal.size()) != n) cppoptlib::mi355::Fail("DeviceHessianDiagonal: size != n");
      d.hessian_diagonal = hessian_diagonal.data();
    }
    d.stop = this->stopping_progress.ToDeviceStop();
    cppoptlib::mi355::Check(mi355_lbfgs_minimize_batch_host(ctx_->get(), &d, B, x0, x, f, g, progress),
                            "mi355_lbfgs_minimize_batch_host");
  }

 private:
  std::shared_ptr<cppoptlib::mi355::Context> ctx_;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_LBFGS_H_
