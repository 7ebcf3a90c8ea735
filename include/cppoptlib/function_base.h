// cppoptlib/function_base.h — function model of the MI355X engine's host API.
//
// Same names, template parameters and call signatures as the reference's
// include/cppoptlib/function_base.h (DifferentiabilityMode :42-46,
// FunctionInterface :52-65, FunctionCRTP :94-126, FunctionState :297-332,
// ModeDowngradeAdapter :151-189, FunctionExpr :191-268 with its deduction guide), so
// objective classes written for PatWie/CppNumericalSolvers keep compiling.
// What differs is where the arithmetic runs: a solver of this library never
// calls operator() in its hot loop.  It asks the function for its DEVICE
// twin (cppoptlib/mi355/device_twin.h: the static `kDeviceObjective` +
// `DeviceParams()` members, or the run-time record a type-erased FunctionExpr
// carries next to its host clone) and the whole solve runs in a HIP kernel.
// operator() stays available for host-side use (callbacks, spot checks).
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_BASE_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_BASE_H_

#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <utility>

#include "mi355/dense.h"
#include "mi355/device_twin.h"

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

// ---------------------------------------------------------------------------------------------------------------------
// FunctionExpr: the reference's type-erasing wrapper (function_base.h:191-268).
//
// Same template parameters, same members (`ptr`, converting constructor with mode downgrade, deep copy, `clone()`,
// call operator, deduction guide).  What this side adds is `device_twin`: the run-time record of what the engine runs
// for the wrapped function (cppoptlib/mi355/device_twin.h), taken from the source's static twin members or its
// DeviceTwin() hook when the wrapper is built, so that `Lbfgs<FunctionExprXd>` / `ConstrainedOptimizationProblem`
// work on the erased type exactly as on the static one.
// ---------------------------------------------------------------------------------------------------------------------
namespace detail {
// The record of a function object; expression nodes and FunctionExpr itself specialise it (function_expressions.h).
template <class F, class = void>
struct TwinOf {
  static cppoptlib::mi355::TwinRecord Make(const F& f) { return cppoptlib::mi355::RecordOfFunction(f); }
};
}  // namespace detail

// Presents a higher-mode FunctionInterface as a lower-mode one (reference :151-189): the extra derivative pointers are
// simply never passed through.  Upgrades are refused at compile time.
template <typename TScalar, DifferentiabilityMode SourceMode, DifferentiabilityMode TargetMode, int TDimension>
struct ModeDowngradeAdapter : public FunctionInterface<TScalar, TargetMode, TDimension> {
  static_assert(static_cast<int>(SourceMode) >= static_cast<int>(TargetMode),
                "ModeDowngradeAdapter only lowers the differentiability mode -- attempting to upgrade.");
  using Base = FunctionInterface<TScalar, TargetMode, TDimension>;
  using VectorType = typename Base::VectorType;
  using MatrixType = typename Base::MatrixType;

  std::unique_ptr<FunctionInterface<TScalar, SourceMode, TDimension>> source;

  explicit ModeDowngradeAdapter(std::unique_ptr<FunctionInterface<TScalar, SourceMode, TDimension>> s)
      : source(std::move(s)) {}

  TScalar operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const override {
    if constexpr (TargetMode == DifferentiabilityMode::None) {
      (void)grad;
      (void)hess;
      return (*source)(x, nullptr, nullptr);
    } else if constexpr (TargetMode == DifferentiabilityMode::First) {
      (void)hess;
      return (*source)(x, grad, nullptr);
    } else {
      return (*source)(x, grad, hess);
    }
  }
  std::unique_ptr<Base> clone() const override {
    return std::make_unique<ModeDowngradeAdapter<TScalar, SourceMode, TargetMode, TDimension>>(source->clone());
  }
};

template <typename TScalar, DifferentiabilityMode TMode = DifferentiabilityMode::First,
          int TDimension = kDynamicDimension>
struct FunctionExpr {
  static constexpr int Dimension = TDimension;
  using ScalarType = TScalar;
  using VectorType = cppoptlib::mi355::Vector<TScalar, TDimension>;
  using MatrixType = cppoptlib::mi355::SquareMatrix<TScalar, TDimension>;
  static constexpr DifferentiabilityMode Differentiability = TMode;

  std::unique_ptr<FunctionInterface<TScalar, TMode, TDimension>> ptr;  // the host clone (reference :203)
  cppoptlib::mi355::TwinRecord device_twin;                            // what the engine runs for it

  // Converting constructor (reference :210-232): any non-FunctionExpr function whose differentiability is at least
  // TMode; a stronger source is wrapped in the downgrade adapter, a weaker one is refused at compile time.
  template <typename F, typename = std::enable_if_t<!std::is_same_v<std::decay_t<F>, FunctionExpr>>>
  FunctionExpr(const F& f) {  // NOLINT: implicit, as in the reference
    static_assert(static_cast<int>(F::Differentiability) >= static_cast<int>(TMode),
                  "Differentiability mode mismatch: source must supply at least as much derivative information as the "
                  "target mode requires (downgrades are accepted, upgrades are not).");
    static_assert(F::Dimension == TDimension, "Dimension mismatch");
    static_assert(std::is_same<typename F::ScalarType, ScalarType>::value, "Compile-time scalar-type mismatch");
    if constexpr (F::Differentiability == TMode) {
      ptr = f.clone();
    } else {
      ptr = std::make_unique<ModeDowngradeAdapter<TScalar, F::Differentiability, TMode, TDimension>>(f.clone());
    }
    device_twin = detail::TwinOf<F>::Make(f);
  }

  FunctionExpr(const FunctionExpr& other)
      : ptr(other.ptr ? other.ptr->clone() : nullptr), device_twin(other.device_twin) {}
  FunctionExpr& operator=(const FunctionExpr& other) {
    if (this != &other) {
      ptr = other.ptr ? other.ptr->clone() : nullptr;
      device_twin = other.device_twin;
    }
    return *this;
  }
  FunctionExpr(FunctionExpr&&) noexcept = default;
  FunctionExpr& operator=(FunctionExpr&&) noexcept = default;

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    return (*ptr)(x, grad, hess);
  }

  // A deep copy of the stored interface (reference :254-256); FunctionExpr itself is not a FunctionInterface.
  std::unique_ptr<FunctionInterface<TScalar, TMode, TDimension>> clone() const { return ptr ? ptr->clone() : nullptr; }
};

template <typename Expr>
FunctionExpr(const Expr&) -> FunctionExpr<typename Expr::ScalarType, Expr::Differentiability, Expr::Dimension>;

template <class T>
struct IsFunctionExpr : std::false_type {};
template <typename TScalar, DifferentiabilityMode TMode, int TDimension>
struct IsFunctionExpr<FunctionExpr<TScalar, TMode, TDimension>> : std::true_type {};

namespace detail {
template <typename TScalar, DifferentiabilityMode TMode, int TDimension>
struct TwinOf<FunctionExpr<TScalar, TMode, TDimension>> {
  static cppoptlib::mi355::TwinRecord Make(const FunctionExpr<TScalar, TMode, TDimension>& f) { return f.device_twin; }
};
}  // namespace detail

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
