// cppoptlib/solver/progress.h — Status / Progress / stopping presets.
//
// Field-for-field the public surface of the reference's solver/progress.h
// (Status :37-47, Progress fields :87-136, DefaultStoppingSolverProgress
// :353-431, ConservativeStoppingSolverProgress :456-464).  The stopping TESTS
// themselves (reference Progress::Update, :153-327) run on the GPU inside the
// solve kernel; here a Progress is a plain value: the criteria a solver is given
// and the record it hands back.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_PROGRESS_H_
#define INCLUDE_CPPOPTLIB_SOLVER_PROGRESS_H_

#include <cstdlib>
#include <cstddef>
#include <cstdint>
#include <ostream>

#include "../../mi355_lbfgs.h"

namespace cppoptlib::solver {

enum class Status {
  NotStarted = -1,
  Continue = 0,
  IterationLimit,
  XDeltaViolation,
  FDeltaViolation,
  GradientNormViolation,
  HessianConditionViolation,
  Finished
};

inline std::ostream& operator<<(std::ostream& os, const Status& s) {
  static const char* const kText[] = {
      "Solver not started.",
      "Convergence criteria not reached.",
      "Iteration limit reached.",
      "Change in parameter vector too small.",
      "Change in cost function value too small.",
      "Gradient vector norm too small.",
      "Condition of Hessian/Covariance matrix too large.",
      "Finished"};
  return os << kText[static_cast<int>(s) + 1];
}

template <class FunctionType, class StateType>
struct Progress {
  using ScalarType = typename FunctionType::ScalarType;

  size_t num_iterations = 0;
  ScalarType x_delta = ScalarType(0);
  int x_delta_violations = 0;
  ScalarType f_delta = ScalarType(0);
  int f_delta_violations = 0;
  bool f_delta_relative = false;
  ScalarType gradient_norm = ScalarType(0);
  bool gradient_norm_relative = true;
  ScalarType condition_hessian = ScalarType(0);
  ScalarType constraint_threshold = ScalarType(0);
  ScalarType kkt_stationarity_threshold = ScalarType(1e-4);
  Status status = Status::NotStarted;
  int past = 0;
  ScalarType past_delta = ScalarType(1e-6);
  // accounting the reference does not have (objective evaluations, sum of used pairs)
  size_t num_function_evaluations = 0;
  size_t history_pairs_used = 0;

  // The stopping fields as the C-ABI POD.
  mi355_lbfgs_stop ToDeviceStop() const {
    mi355_lbfgs_stop s;
    s.num_iterations = static_cast<uint64_t>(num_iterations);
    s.x_delta = static_cast<double>(x_delta);
    s.x_delta_violations = x_delta_violations;
    s.f_delta = static_cast<double>(f_delta);
    s.f_delta_violations = f_delta_violations;
    s.f_delta_relative = f_delta_relative ? 1 : 0;
    s.gradient_norm = static_cast<double>(gradient_norm);
    s.gradient_norm_relative = gradient_norm_relative ? 1 : 0;
    s.past = past;
    s.past_delta = static_cast<double>(past_delta);
    return s;
  }
  // Result record of one problem.
  static Progress FromDevice(const mi355_lbfgs_progress& p) {
    Progress r;
    r.num_iterations = p.num_iterations;
    r.x_delta = static_cast<ScalarType>(p.x_delta);
    r.f_delta = static_cast<ScalarType>(p.f_delta);
    r.gradient_norm = static_cast<ScalarType>(p.gradient_norm);
    r.status = static_cast<Status>(p.status);
    r.num_function_evaluations = p.nfev;
    r.history_pairs_used = p.sum_k;
    return r;
  }
};

template <class FunctionType, class StateType>
Progress<FunctionType, StateType> DefaultStoppingSolverProgress() {
  using S = typename FunctionType::ScalarType;
  Progress<FunctionType, StateType> p;
  p.num_iterations = 10000;
  p.x_delta = S(1e-9);
  p.x_delta_violations = 1;
  p.f_delta = S(0);
  p.f_delta_violations = 1;
  p.gradient_norm = S(1e-5);
  p.condition_hessian = S(0);
  p.constraint_threshold = S(1e-5);
  p.past = 3;
  p.past_delta = S(1e-6);
#ifdef CPPOPT_SWEEP
  // The reference's parameter-sweep build (progress.h:359-381, -DCPPOPT_SWEEP): five of the default preset's fields are
  // read from the environment at every call, so that one binary can be re-run with different stopping settings.
  // Unset variables keep the preset; the values are parsed with atof / atoi as the reference does.
  const struct { const char* name; S* real; int* whole; } sweep[] = {
      {"CPPOPT_X_DELTA", &p.x_delta, nullptr},       {"CPPOPT_X_DELTA_VIOL", nullptr, &p.x_delta_violations},
      {"CPPOPT_GRAD_NORM", &p.gradient_norm, nullptr}, {"CPPOPT_PAST", nullptr, &p.past},
      {"CPPOPT_PAST_DELTA", &p.past_delta, nullptr}};
  for (const auto& knob : sweep) {
    const char* text = std::getenv(knob.name);
    if (!text) continue;
    if (knob.real) *knob.real = static_cast<S>(std::atof(text));
    if (knob.whole) *knob.whole = std::atoi(text);
  }
#endif  // CPPOPT_SWEEP
  p.status = Status::NotStarted;
  return p;
}

template <class FunctionType, class StateType>
Progress<FunctionType, StateType> ConservativeStoppingSolverProgress() {
  using S = typename FunctionType::ScalarType;
  auto p = DefaultStoppingSolverProgress<FunctionType, StateType>();
  p.gradient_norm = S(5e-6);
  p.past = 5;
  p.past_delta = S(1e-10);
  return p;
}

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_PROGRESS_H_
