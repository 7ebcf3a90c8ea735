// cppoptlib/function_expressions.h — sums and scalar multiples of functions that keep a device twin.
//
// Mirrors the part of the reference's expression layer the README ridge example uses
// (include/cppoptlib/function_expressions.h: AddExpression :91-143 — value fx_f + fx_g, gradient
// grad_f + grad_g, Hessian hess_f + hess_g; MulExpression :200-254 — c * fx, c * grad, c * hess;
// FunctionExpr :316-372 — the type-erasing wrapper whose decltype is handed to the solver):
//
//     FunctionExpr objective = SquaredError<>(rows, n, A, y) + lambda * L2Reg<>(n);
//     Lbfgs<decltype(objective)> solver;                     // README.md:159-164
//
// On the host every expression evaluates through its operands (same operation order as the
// reference's templates).  On the device an expression needs a twin kernel, so only the shapes in
// the DeviceTwin table at the bottom can be handed to Lbfgs / Lbfgsb; any other composition is a
// compile-time error there (no CPU fallback).  The differentiability of an expression is the
// weaker of its operands' modes (reference :60-66).
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_EXPRESSIONS_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_EXPRESSIONS_H_

#include <type_traits>
#include <utility>
#include <vector>

#include "function_base.h"
#include "mi355/objectives.h"

namespace cppoptlib::function {

constexpr DifferentiabilityMode WeakerMode(DifferentiabilityMode a, DifferentiabilityMode b) {
  return static_cast<int>(a) < static_cast<int>(b) ? a : b;
}

namespace detail {
// Evaluates `fn` with as many derivative outputs as its own mode provides.
template <class F, class V, class M>
auto EvaluateUpTo(const F& fn, const V& x, V* grad, M* hess) {
  if constexpr (F::Differentiability == DifferentiabilityMode::Second) {
    return fn(x, grad, hess);
  } else if constexpr (F::Differentiability == DifferentiabilityMode::First) {
    return fn(x, grad);
  } else {
    return fn(x);
  }
}
}  // namespace detail

// c * f
template <class F>
class ScaledFunction : public FunctionCRTP<ScaledFunction<F>, typename F::ScalarType, F::Differentiability,
                                           F::Dimension> {
 public:
  using Super = FunctionCRTP<ScaledFunction<F>, typename F::ScalarType, F::Differentiability, F::Dimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  ScaledFunction(ScalarType c, F f) : c_(c), f_(std::move(f)) {}
  ScalarType factor() const { return c_; }
  const F& function() const { return f_; }

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    const ScalarType v = detail::EvaluateUpTo(f_, x, grad, hess);
    if (grad)
      for (std::ptrdiff_t i = 0; i < grad->size(); ++i) (*grad)[i] = c_ * (*grad)[i];
    if (hess)
      for (std::ptrdiff_t i = 0; i < hess->rows(); ++i)
        for (std::ptrdiff_t j = 0; j < hess->rows(); ++j) (*hess)(i, j) = c_ * (*hess)(i, j);
    return c_ * v;
  }

 private:
  ScalarType c_;
  F f_;
};

// f + g
template <class F, class G>
class SumFunction
    : public FunctionCRTP<SumFunction<F, G>, typename F::ScalarType,
                          WeakerMode(F::Differentiability, G::Differentiability), F::Dimension> {
 public:
  static_assert(std::is_same<typename F::ScalarType, typename G::ScalarType>::value, "scalar types differ");
  using Super = FunctionCRTP<SumFunction<F, G>, typename F::ScalarType,
                             WeakerMode(F::Differentiability, G::Differentiability), F::Dimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  SumFunction(F f, G g) : f_(std::move(f)), g_(std::move(g)) {}
  const F& left() const { return f_; }
  const G& right() const { return g_; }

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    VectorType grad_g;
    MatrixType hess_g;
    const ScalarType vf = detail::EvaluateUpTo(f_, x, grad, hess);
    const ScalarType vg = detail::EvaluateUpTo(g_, x, grad ? &grad_g : nullptr, hess ? &hess_g : nullptr);
    if (grad)
      for (std::ptrdiff_t i = 0; i < grad->size(); ++i) (*grad)[i] = (*grad)[i] + grad_g[i];
    if (hess)
      for (std::ptrdiff_t i = 0; i < hess->rows(); ++i)
        for (std::ptrdiff_t j = 0; j < hess->rows(); ++j) (*hess)(i, j) = (*hess)(i, j) + hess_g(i, j);
    return vf + vg;
  }

 private:
  F f_;
  G g_;
};

// f * g  (reference ProdExpression, function_expressions.h:260-315: value fx * gx, gradient gx * grad_f + fx * grad_g).
// First-order on this side: it exists to be a TERM of a constrained problem (the device evaluates the same two
// primitives and the same product rule: MI355_AL_PARTS_PRODUCT).
template <class F, class G>
class ProductFunction
    : public FunctionCRTP<ProductFunction<F, G>, typename F::ScalarType, DifferentiabilityMode::First, F::Dimension> {
 public:
  static_assert(std::is_same<typename F::ScalarType, typename G::ScalarType>::value, "scalar types differ");
  static_assert(F::Dimension == G::Dimension, "dimensions differ");
  using Super = FunctionCRTP<ProductFunction<F, G>, typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  ProductFunction(F f, G g) : f_(std::move(f)), g_(std::move(g)) {}
  const F& left() const { return f_; }
  const G& right() const { return g_; }

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    VectorType grad_f, grad_g;
    const ScalarType fx = f_(x, &grad_f);
    const ScalarType gx = g_(x, &grad_g);
    if (grad) {
      *grad = VectorType(x.size());
      for (std::ptrdiff_t i = 0; i < grad->size(); ++i) (*grad)[i] = gx * grad_f[i] + fx * grad_g[i];
    }
    return fx * gx;
  }

 private:
  F f_;
  G g_;
};

template <class T, class = void>
struct IsFunction : std::false_type {};
template <class T>
struct IsFunction<T, std::void_t<typename T::ScalarType, decltype(T::Differentiability), decltype(T::Dimension)>>
    : std::is_base_of<FunctionInterface<typename T::ScalarType, T::Differentiability, T::Dimension>, T> {};

template <class F, class G, class = std::enable_if_t<IsFunction<F>::value && IsFunction<G>::value>>
SumFunction<F, G> operator+(F f, G g) {
  return SumFunction<F, G>(std::move(f), std::move(g));
}
template <class F, class G, class = std::enable_if_t<IsFunction<F>::value && IsFunction<G>::value>>
ProductFunction<F, G> operator*(F f, G g) {
  return ProductFunction<F, G>(std::move(f), std::move(g));
}
template <class F, class = std::enable_if_t<IsFunction<F>::value>>
ScaledFunction<F> operator*(double c, F f) {
  return ScaledFunction<F>(static_cast<typename F::ScalarType>(c), std::move(f));
}
template <class F, class = std::enable_if_t<IsFunction<F>::value>>
ScaledFunction<F> operator*(F f, double c) {
  return ScaledFunction<F>(static_cast<typename F::ScalarType>(c), std::move(f));
}

// f - k and k - f (reference :497-518: SubExpression with a ConstExpression operand — value fx - k / k - fx,
// gradient grad - 0 / 0 - grad).  These are how constraints are written: `circle - 2.0`, `2.0 - circle`
// (src/examples/constrained_simple2.cc:56-62).
template <class F, bool kConstantFirst>
class OffsetFunction : public FunctionCRTP<OffsetFunction<F, kConstantFirst>, typename F::ScalarType,
                                           F::Differentiability, F::Dimension> {
 public:
  using Super =
      FunctionCRTP<OffsetFunction<F, kConstantFirst>, typename F::ScalarType, F::Differentiability, F::Dimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr bool kIsConstantFirst = kConstantFirst;
  OffsetFunction(F f, ScalarType k) : f_(std::move(f)), k_(k) {}
  const F& function() const { return f_; }
  ScalarType constant() const { return k_; }

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    const ScalarType v = detail::EvaluateUpTo(f_, x, grad, hess);
    if (grad)
      for (std::ptrdiff_t i = 0; i < grad->size(); ++i)
        (*grad)[i] = kConstantFirst ? ScalarType(0) - (*grad)[i] : (*grad)[i] - ScalarType(0);
    if (hess)
      for (std::ptrdiff_t i = 0; i < hess->rows(); ++i)
        for (std::ptrdiff_t j = 0; j < hess->rows(); ++j)
          (*hess)(i, j) = kConstantFirst ? ScalarType(0) - (*hess)(i, j) : (*hess)(i, j) - ScalarType(0);
    return kConstantFirst ? k_ - v : v - k_;
  }

 private:
  F f_;
  ScalarType k_;
};
template <class F, class = std::enable_if_t<IsFunction<F>::value>>
OffsetFunction<F, false> operator-(F f, double k) {
  return OffsetFunction<F, false>(std::move(f), static_cast<typename F::ScalarType>(k));
}
template <class F, class = std::enable_if_t<IsFunction<F>::value>>
OffsetFunction<F, true> operator-(double k, F f) {
  return OffsetFunction<F, true>(std::move(f), static_cast<typename F::ScalarType>(k));
}

// ---------------------------------------------------------------------------------------------
// Device twins of expressions.  DeviceTwin<Expr>::Make(expr) returns the single library objective
// (mi355/objectives.h) whose kernel computes the expression with the expression's own operation
// order; the primary template has no Make, which is what rejects unsupported shapes.
// ---------------------------------------------------------------------------------------------
template <class Expr>
struct DeviceTwin {};

//   SquaredError + lambda * L2Reg   ->   SquaredErrorRidge   (README.md:159: value r.r + lambda*(x.x),
//   gradient 2 A^T r + lambda*(2 x), Hessian diagonal (2 A^T A)_jj + lambda*2 — term by term what
//   AddExpression / MulExpression produce from the two operands)
template <int D, DifferentiabilityMode M1, DifferentiabilityMode M2>
struct DeviceTwin<SumFunction<SquaredError<D, M1>, ScaledFunction<L2Reg<D, M2>>>> {
  using type = SquaredErrorRidge<D, WeakerMode(M1, M2)>;
  static type Make(const SumFunction<SquaredError<D, M1>, ScaledFunction<L2Reg<D, M2>>>& e) {
    const auto& se = e.left();
    return type(se.rows(), se.cols(), se.matrix(), se.rhs(), e.right().factor());
  }
};

// The README's wrapper: holds any expression; its decltype is the solver's function type.  It is a
// function itself (host evaluation forwards to the expression) and carries the expression's twin.
template <class Expr>
class FunctionExpr
    : public FunctionCRTP<FunctionExpr<Expr>, typename Expr::ScalarType, Expr::Differentiability, Expr::Dimension> {
 public:
  using Super = FunctionCRTP<FunctionExpr<Expr>, typename Expr::ScalarType, Expr::Differentiability, Expr::Dimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  using Twin = typename DeviceTwin<Expr>::type;  // a composition without a device kernel fails here
  static constexpr int kDeviceObjective = Twin::kDeviceObjective;
  static constexpr int kDeviceObjectiveFused = cppoptlib::mi355::FusedDeviceObjective<Twin>::Of(MI355_ARITH_FMA);

  FunctionExpr(Expr e) : expr_(std::move(e)), twin_(DeviceTwin<Expr>::Make(expr_)) {}  // NOLINT: implicit, as in the README

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    return detail::EvaluateUpTo(expr_, x, grad, hess);
  }
  std::vector<double> DeviceParams() const { return twin_.DeviceParams(); }
  std::vector<double> DevicePerProblem() const { return twin_.DevicePerProblem(); }
  std::vector<double> DeviceHessianDiagonal() const { return twin_.DeviceHessianDiagonal(); }
  // the twin's own-matrix form (a batch of expressions over DIFFERENT matrices: cppoptlib/mi355/batch_driver.h)
  static constexpr int kDeviceObjectiveOwnMatrix = Twin::kDeviceObjectiveOwnMatrix;
  std::vector<double> DeviceOwnMatrixParams() const { return twin_.DeviceOwnMatrixParams(); }
  std::vector<double> DeviceOwnMatrixRow() const { return twin_.DeviceOwnMatrixRow(); }
  auto DeviceFingerprint() const { return twin_.DeviceFingerprint(); }
  auto DeviceOwnMatrixKey() const { return twin_.DeviceOwnMatrixKey(); }
  uint64_t DeviceParamsHash() const { return twin_.DeviceParamsHash(); }
  double NormalEquationConditionBound() const { return twin_.NormalEquationConditionBound(); }

 private:
  Expr expr_;
  Twin twin_;
};

}  // namespace cppoptlib::function
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_EXPRESSIONS_H_
