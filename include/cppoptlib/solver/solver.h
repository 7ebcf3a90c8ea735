// cppoptlib/solver/solver.h — solver base class of the MI355X engine's host API.
//
// Public surface of the reference's Solver (solver/solver.h:156-231): a public
// `stopping_progress`, a constructor taking the stopping Progress, SetCallback,
// and a virtual Minimize(function, state) -> tuple<State, Progress>.  The
// reference drives OptimizationStep in a host loop (:196-220); here the loop
// lives on the GPU, so derived solvers implement Minimize / MinimizeBatch
// directly and there is no per-iteration virtual step.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_SOLVER_H_
#define INCLUDE_CPPOPTLIB_SOLVER_SOLVER_H_

#include <functional>
#include <iomanip>
#include <iostream>
#include <tuple>
#include <utility>

#include "../function.h"
#include "progress.h"

namespace cppoptlib::solver {

template <class FunctionType, class StateType>
auto NoOpCallback() {
  return [](const FunctionType&, const StateType&, const Progress<FunctionType, StateType>&) {};
}

// Prints one line per invocation: iteration, value, deltas, gradient norm, status.
template <class FunctionType, class StateType>
auto PrintProgressCallback(std::ostream& out) {
  return [&out](const FunctionType&, const StateType& state, const Progress<FunctionType, StateType>& p) {
    out << "iter " << std::setw(6) << p.num_iterations << "  f = " << std::setprecision(10) << state.value
        << "  |dx| = " << p.x_delta << "  |df| = " << p.f_delta << "  |g| = " << p.gradient_norm << "  "
        << p.status << "\n";
  };
}

template <typename FunctionTypeT, typename StateTypeT>
class Solver {
 public:
  using StateType = StateTypeT;
  using FunctionType = FunctionTypeT;
  using ProgressType = Progress<FunctionType, StateType>;
  using CallbackType = std::function<void(const FunctionType&, const StateType&, const ProgressType&)>;

  ProgressType stopping_progress;

  explicit Solver(const ProgressType& progress = DefaultStoppingSolverProgress<FunctionType, StateType>())
      : stopping_progress(progress), step_callback_(NoOpCallback<FunctionType, StateType>()) {}
  virtual ~Solver() = default;

  // The callback is invoked with the evaluated start state before the solve and
  // with the final state after it (the reference also calls it before every
  // step, solver.h:197; a fused GPU solve has no host-visible intermediate steps).
  void SetCallback(CallbackType callback) { step_callback_ = std::move(callback); }

  virtual std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                                       const StateType& function_state) = 0;

 protected:
  CallbackType step_callback_;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_SOLVER_H_
