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

  // Callback contract.  The reference calls step_callback_ before every step and once after the loop
  // (solver.h:197, :222).  Here the loop runs on the device, so Minimize records the solve's per-iteration
  // trace (mi355_lbfgs_trace: value, deltas, gradient norm, status of every iteration, and x / g where the
  // solver keeps them) and REPLAYS it into the callback after the kernel returns: the callback sees the same
  // sequence of (state, progress) pairs, in the same order, as the reference's — only later.  Without a
  // callback nothing is recorded and nothing is evaluated on the host.
  void SetCallback(CallbackType callback) {
    step_callback_ = std::move(callback);
    has_callback_ = true;
  }
  bool HasCallback() const { return has_callback_; }

  virtual std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                                       const StateType& function_state) = 0;

  CallbackType step_callback_;  // public, as in the reference (solver.h:230)

 protected:
  bool has_callback_ = false;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_SOLVER_H_
