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
#include "../mi355/batch_driver.h"
#include "../mi355/context.h"
#include "solver.h"

namespace cppoptlib::solver {

template <typename FunctionType, template <class, int> class LineSearch = linesearch::MoreThuente>
class Bfgs : public Solver<FunctionType, cppoptlib::function::FunctionState<typename FunctionType::ScalarType,
                                                                            FunctionType::Dimension>> {
  static_assert(FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::First ||
                    FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second,
                "Bfgs only supports first- or second-order differentiable functions");
  static_assert(std::is_floating_point<typename FunctionType::ScalarType>::value,
                "ScalarType must be float or double (the MI355X engine computes in fp64 either way: a float function type is "
                "widened at the boundary and its results are rounded back, see INTEGRATION.md)");
  static_assert(cppoptlib::mi355::kHasDeviceTwin<FunctionType>,
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

  // With a callback set the solve is traced on the device and the callback replayed afterwards
  // (cppoptlib/mi355/batch_driver.h); without one nothing is evaluated on the host.
  std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                               const StateType& function_state) override {
    // (a Second-mode function: the reference's Progress::Update computes condition_hessian under every solver —
    //  reported here from the host functor's Hessian, batch_driver.h)
    return cppoptlib::mi355::MinimizeOneReportingCondition<StateType, ProgressType, VectorType>(
        "Bfgs", function, function_state, this->HasCallback(), this->step_callback_,
        static_cast<uint64_t>(this->stopping_progress.num_iterations),
        static_cast<double>(this->stopping_progress.condition_hessian),
        [&](int n, int64_t B, const double* x0, double* x, double* f, double* g, mi355_lbfgs_progress* prog,
            const mi355_lbfgs_trace* trace) { MinimizeBatchRaw(function, n, B, x0, x, f, g, prog, trace); });
  }

  // Solves every start state independently in one kernel launch.
  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const FunctionType& function,
                                                                 const std::vector<StateType>& states) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    MinimizeBatchRaw(function, n, B, x0.data(), x.data(), f.data(), g.data(), prog.data());
    return cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
  }

  void MinimizeBatchRaw(const FunctionType& function, int n, int64_t B, const double* x0, double* x, double* f,
                        double* g, mi355_lbfgs_progress* progress, const mi355_lbfgs_trace* trace = nullptr) {
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    cppoptlib::mi355::RequireObjective(function, "Bfgs");
    if (cppoptlib::mi355::CarriesPerProblemData(function))
      cppoptlib::mi355::Fail("Bfgs: the device Bfgs kernel is built for objectives without per-problem data");
    const std::vector<double> params = cppoptlib::mi355::ObjectiveParams(function, n);
    mi355_lbfgs_desc d{};
    d.objective = cppoptlib::mi355::PlainObjectiveId(function);
    d.linesearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;
    d.n = n;
    d.m = 1;  // not used by Bfgs
    d.objective_params = params.empty() ? nullptr : params.data();
    d.n_params = static_cast<int32_t>(params.size());
    d.trace = trace;
    d.stop = this->stopping_progress.ToDeviceStop();
    cppoptlib::mi355::Check(mi355_bfgs_minimize_batch_host(ctx_->get(), &d, B, x0, x, f, g, progress),
                            "mi355_bfgs_minimize_batch_host");
  }

 private:
  std::shared_ptr<cppoptlib::mi355::Context> ctx_;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_BFGS_H_
