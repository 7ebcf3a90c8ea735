// cppoptlib/function_base.h — function model of the MI355X engine's host API.
//
// Same names, template parameters and call signatures as the reference's
// include/cppoptlib/function_base.h (DifferentiabilityMode :42-46,
// FunctionInterface :52-65, FunctionCRTP :94-126, FunctionState :297-332), so
// objective classes written for PatWie/CppNumericalSolvers keep compiling.
// What differs is where the arithmetic runs: a solver of this library never
// calls operator() in its hot loop.  It asks the function type for its DEVICE
// twin (`kDeviceObjective` + `DeviceParams()`, see cppoptlib/mi355/objectives.h)
// and the whole solve runs in a HIP kernel.  operator() stays available for
// host-side use (callbacks, spot checks).
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_BASE_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_BASE_H_

#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <utility>

#include "mi355/dense.h"

namespace cppoptlib::function {

enum class DifferentiabilityMode { None = 0, First = 1, Second = 2 };

#ifdef CPPOPTLIB_MI355_HAVE_EIGEN
constexpr int kDynamicDimension = Eigen::Dynamic;
#else
constexpr int kDynamicDimension = cppoptlib::mi355::kDynamic;
#endif

template <class TScalar, DifferentiabilityMode Mode, int TDimension = kDynamicDimension>
struct FunctionInterface {
  static constexpr int Dimension = TDimension;
  static constexpr DifferentiabilityMode Differentiability = Mode;
  using ScalarType = TScalar;
  using VectorType = cppoptlib::mi355::Vector<TScalar, TDimension>;
  using MatrixType = cppoptlib::mi355::SquareMatrix<TScalar, TDimension>;

  virtual ~FunctionInterface() = default;
  virtual ScalarType operator()(const VectorType& x, VectorType* grad = nullptr,
                                MatrixType* hess = nullptr) const = 0;
  virtual std::unique_ptr<FunctionInterface> clone() const = 0;
};

#ifdef __GNUC__
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Woverloaded-virtual"
#endif
// Derived supplies operator()(x) / (x, grad) / (x, grad, hess) according to TMode.
template <class Derived, class TScalar, DifferentiabilityMode TMode, int TDimension = kDynamicDimension>
struct FunctionCRTP : public FunctionInterface<TScalar, TMode, TDimension> {
  using Base = FunctionInterface<TScalar, TMode, TDimension>;
  static constexpr int Dimension = TDimension;
  static constexpr DifferentiabilityMode Differentiability = TMode;
  using ScalarType = TScalar;
  using VectorType = typename Base::VectorType;
  using MatrixType = typename Base::MatrixType;

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr,
                        MatrixType* hess = nullptr) const override {
    const Derived& self = static_cast<const Derived&>(*this);
    if constexpr (TMode == DifferentiabilityMode::None) {
      return self(x);
    } else if constexpr (TMode == DifferentiabilityMode::First) {
      if (hess != nullptr) {  // same contract as the reference: throw, or abort without exceptions
#if defined(__cpp_exceptions) || defined(__EXCEPTIONS) || defined(_CPPUNWIND)
        throw std::runtime_error("Hessian not available for first order function.");
#else
        std::abort();
#endif
      }
      return self(x, grad);
    } else {
      return self(x, grad, hess);
    }
  }
  std::unique_ptr<Base> clone() const override {
    return std::make_unique<Derived>(static_cast<const Derived&>(*this));
  }
};
#ifdef __GNUC__
#pragma GCC diagnostic pop
#endif

// A point of the trajectory with the objective value and gradient AT that point.
template <class TScalar, int TDimension = kDynamicDimension>
struct FunctionState {
  static constexpr bool IsConstrained = false;
  using ScalarType = TScalar;
  using VectorType = cppoptlib::mi355::Vector<TScalar, TDimension>;

  VectorType x;
  ScalarType value = ScalarType(0);
  VectorType gradient;

  explicit FunctionState(VectorType x_in) : x(std::move(x_in)) {}
  template <class FunctionT>
  FunctionState(const FunctionT& function, VectorType x_in) : x(std::move(x_in)) {
    if constexpr (FunctionT::Differentiability == DifferentiabilityMode::None) {
      value = function(x);
    } else {
      value = function(x, &gradient);
    }
  }
  FunctionState(VectorType x_in, ScalarType value_in, VectorType gradient_in)
      : x(std::move(x_in)), value(value_in), gradient(std::move(gradient_in)) {}
};

template <class TScalar, int TDimension>
FunctionState(cppoptlib::mi355::Vector<TScalar, TDimension>) -> FunctionState<TScalar, TDimension>;

}  // namespace cppoptlib::function
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_BASE_H_
