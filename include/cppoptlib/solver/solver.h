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
#include <type_traits>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <tuple>
#include <utility>

#include "../function.h"
#include "progress.h"

namespace cppoptlib::solver {

// A state that caches (value, gradient) next to x — FunctionState, as opposed to AugmentedLagrangeState (reference
// solver.h:48-54): what the progress printer asks before it reuses the cached numbers.
template <class, class = void>
struct IsFunctionState : std::false_type {};
template <class S>
struct IsFunctionState<S, std::void_t<decltype(std::declval<S>().value), decltype(std::declval<S>().gradient)>> : std::true_type {};

template <class FunctionType, class StateType>
auto NoOpCallback() {
  return [](const FunctionType&, const StateType&, const Progress<FunctionType, StateType>&) {};
}

namespace detail {
// a vector on one line: `v.transpose()` where the vector type has it (Eigen), the vector itself otherwise
template <class V, class = void>
struct RowText {
  static void Print(std::ostream& os, const V& v) { os << v; }
};
template <class V>
struct RowText<V, std::void_t<decltype(std::declval<const V&>().transpose())>> {
  static void Print(std::ostream& os, const V& v) { os << v.transpose(); }
};
}  // namespace detail

// The reference's progress printer (solver/solver.h:57-139), same block per invocation:
//   --- Iteration:     N ---
//     Value: / X: / Gradient: / Gradient Norm: / X Delta: / F Delta: / Hessian Cond.: (Second mode)
//   -------------------------
// labels left-aligned in 18 columns, numbers right-aligned in 15, fixed with 6 decimals; x and the gradient through a
// string stream of their own (so they keep the default float format, as there).  Value and gradient come from the
// state (every state the solvers replay carries them, function_base.h:297-332).
template <class FunctionType, class StateType>
auto PrintProgressCallback(std::ostream& out) {
  return [&out](const FunctionType&, const StateType& state, const Progress<FunctionType, StateType>& p) {
    constexpr int kLabel = 18, kNumber = 15;
    auto line = [&out](const char* label, auto value) {
      out << std::left << std::setw(kLabel) << label << std::right << std::setw(kNumber) << value << "\n";
    };
    auto vector_line = [&out](const char* label, const auto& v) {
      std::stringstream text;
      detail::RowText<std::decay_t<decltype(v)>>::Print(text, v);
      out << std::left << std::setw(kLabel) << label << " " << text.str() << "\n";
    };
    out << std::fixed << std::setprecision(6);
    out << "--- Iteration: " << std::setw(5) << std::right << p.num_iterations << " ---\n";
    line("  Value:", state.value);
    vector_line("  X:", state.x);
    if constexpr (static_cast<int>(FunctionType::Differentiability) >=
                  static_cast<int>(cppoptlib::function::DifferentiabilityMode::First)) {
      vector_line("  Gradient:", state.gradient);
      line("  Gradient Norm:", p.gradient_norm);
    }
    line("  X Delta:", p.x_delta);
    line("  F Delta:", p.f_delta);
    if constexpr (FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second) {
      if (p.condition_hessian == p.condition_hessian) {
        line("  Hessian Cond.:", p.condition_hessian);
      } else {
        line("  Hessian Cond.:", "N/A");
      }
    }
    out << "-------------------------" << std::endl;
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

  // The callback member: a CallbackType that remembers whether anything was ever assigned to it, so that code written
  // against the reference — which assigns `solver.step_callback_ = ...` directly (the member is public there,
  // solver.h:230) — gets the traced, replayed solve exactly as SetCallback users do.
  class CallbackSlot {
   public:
    explicit CallbackSlot(CallbackType noop) : fn_(std::move(noop)) {}
    template <class F, class = std::enable_if_t<!std::is_same<std::decay_t<F>, CallbackSlot>::value>>
    CallbackSlot& operator=(F&& f) {
      fn_ = std::forward<F>(f);
      set_ = true;
      return *this;
    }
    void operator()(const FunctionType& function, const StateType& state, const ProgressType& progress) const {
      fn_(function, state, progress);
    }
    operator const CallbackType&() const { return fn_; }
    bool is_set() const { return set_; }

   private:
    CallbackType fn_;
    bool set_ = false;
  };

  explicit Solver(const ProgressType& progress = DefaultStoppingSolverProgress<FunctionType, StateType>())
      : stopping_progress(progress), step_callback_(NoOpCallback<FunctionType, StateType>()) {}
  virtual ~Solver() = default;

  // Callback contract.  The reference calls step_callback_ before every step and once after the loop
  // (solver.h:197, :222).  Here the loop runs on the device, so Minimize records the solve's per-iteration
  // trace (mi355_lbfgs_trace: value, deltas, gradient norm, status of every iteration, and x / g where the
  // solver keeps them) and REPLAYS it into the callback after the kernel returns: the callback sees the same
  // sequence of (state, progress) pairs, in the same order, as the reference's — only later.  Without a
  // callback nothing is recorded and nothing is evaluated on the host.
  void SetCallback(CallbackType callback) { step_callback_ = std::move(callback); }
  bool HasCallback() const { return step_callback_.is_set(); }

  virtual std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                                       const StateType& function_state) = 0;

  CallbackSlot step_callback_;  // public and assignable, as in the reference (solver.h:230)
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_SOLVER_H_
