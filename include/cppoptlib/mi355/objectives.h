// cppoptlib/mi355/objectives.h — objectives that have a device twin.
//
// The reference takes an arbitrary host functor (FunctionCRTP::operator(),
// function_base.h:103-120).  A GPU engine needs the objective as device code, so
// a function type advertises its twin through two members:
//     static constexpr int kDeviceObjective;        // an mi355_objective id
//     std::vector<double> DeviceParams() const;     // shared parameter blob
// The classes below are ordinary FunctionCRTP functors (operator() works on the
// host, same operation order as the device code) that carry those members.
// Lbfgs<F>::Minimize refuses at compile time a function type without a twin —
// there is no CPU fallback.
#ifndef CPPOPTLIB_MI355_OBJECTIVES_H_
#define CPPOPTLIB_MI355_OBJECTIVES_H_

#include <type_traits>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "../function_base.h"

namespace cppoptlib::mi355 {

template <class F, class = void>
struct HasDeviceObjective : std::false_type {};
template <class F>
struct HasDeviceObjective<F, std::void_t<decltype(F::kDeviceObjective),
                                         decltype(std::declval<const F&>().DeviceParams())>>
    : std::true_type {};

}  // namespace cppoptlib::mi355

namespace cppoptlib::function {

// Chained Rosenbrock-N: f = sum_{i<N-1} (1-x_i)^2 + 100 (x_{i+1}-x_i^2)^2.
// At N = 2 this is the reference's test functor (src/test/verify.cc:58-69).
template <int TDimension = kDynamicDimension>
class Rosenbrock : public FunctionCRTP<Rosenbrock<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<Rosenbrock<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_ROSENBROCK;
  std::vector<double> DeviceParams() const { return {}; }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    if (gradient) gradient->resize(n);
    ScalarType f = 0;
    for (int i = 0; i < n; ++i) {
      const bool a = i + 1 < n, b = i > 0;
      ScalarType ga = 0, gb = 0;
      if (a) {
        const ScalarType t1 = 1 - x[i], t2 = x[i + 1] - x[i] * x[i];
        const ScalarType term = t1 * t1 + (100 * t2) * t2;
        f = (i == 0) ? term : f + term;
        ga = -2 * (1 - x[i]) + (200 * t2) * (-2 * x[i]);
      }
      if (b) gb = 200 * (x[i] - x[i - 1] * x[i - 1]);
      if (gradient) (*gradient)[i] = (a && b) ? ga + gb : (a ? ga : gb);
    }
    return f;
  }
};

// f(x) = sum_i a_i x_i^2 + c  (README.md:21-28 quick start: a = (5, 100), c = 5).
template <int TDimension = kDynamicDimension>
class DiagQuadratic
    : public FunctionCRTP<DiagQuadratic<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<DiagQuadratic<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_DIAG_QUADRATIC;

  DiagQuadratic(std::vector<double> a, double c) : a_(std::move(a)), c_(c) {}
  std::vector<double> DeviceParams() const {
    std::vector<double> p = a_;
    p.push_back(c_);
    return p;
  }
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    if (gradient) gradient->resize(n);
    ScalarType f = 0;
    for (int i = 0; i < n; ++i) {
      const ScalarType term = (a_[i] * x[i]) * x[i];
      f = (i == 0) ? term : f + term;
      if (gradient) (*gradient)[i] = (2 * a_[i]) * x[i];
    }
    return f + c_;
  }

 private:
  std::vector<double> a_;
  double c_;
};

}  // namespace cppoptlib::function
#endif  // CPPOPTLIB_MI355_OBJECTIVES_H_
