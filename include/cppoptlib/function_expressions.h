// cppoptlib/function_expressions.h — sums, products and scalar multiples of functions that keep a device twin.
//
// Mirrors the reference's expression layer (include/cppoptlib/function_expressions.h: AddExpression :91-143 — value
// fx_f + fx_g, gradient grad_f + grad_g, Hessian hess_f + hess_g; MulExpression :200-254 — c * fx, c * grad, c * hess;
// ProdExpression :260-315; the operators :420-518, `f - k` / `k - f` being SubExpression with a ConstExpression operand):
//
//     FunctionExpr objective = SquaredError<>(rows, n, A, y) + lambda * L2Reg<>(n);
//     Lbfgs<decltype(objective)> solver;                     // README.md:159-164
//
// The operands are function objects of any static type or type-erased FunctionExpr wrappers (function_base.h), as in the
// reference.  On the host every expression evaluates through its operands (same operation order as the reference's
// templates).  On the device an expression needs a twin kernel: the record of an expression is composed from the records
// of its operands by the rules of cppoptlib/mi355/device_twin.h / objectives.h (TwinOf specialisations at the bottom), and
// a composition the device has no kernel for simply has no twin — handing it to a solver fails there, loudly; there is no
// CPU fallback.  The differentiability of an expression is the weaker of its operands' modes (reference :60-66).
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
    : std::integral_constant<bool, std::is_base_of<FunctionInterface<typename T::ScalarType, T::Differentiability,
                                                                      T::Dimension>, T>::value ||
                                       IsFunctionExpr<T>::value> {};

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

// f - g for two functions (reference SubExpression :148-196).  Formed as f + (-1) * g: (-1) * v is exact and a + (-b) is
// a - b bit for bit, so value, gradient and Hessian equal the reference node's, and the node keeps a device twin wherever
// the sum and the scaled operand have one.
template <class F, class G, class = std::enable_if_t<IsFunction<F>::value && IsFunction<G>::value>>
SumFunction<F, ScaledFunction<G>> operator-(F f, G g) {
  return SumFunction<F, ScaledFunction<G>>(std::move(f), ScaledFunction<G>(typename G::ScalarType(-1), std::move(g)));
}

// The constant function (reference ConstExpression :46-87): value c, zero gradient and Hessian of the argument's size.
template <typename TScalar, DifferentiabilityMode TMode = DifferentiabilityMode::First, int TDimension = kDynamicDimension>
class ConstExpression : public FunctionCRTP<ConstExpression<TScalar, TMode, TDimension>, TScalar, TMode, TDimension> {
 public:
  using Super = FunctionCRTP<ConstExpression<TScalar, TMode, TDimension>, TScalar, TMode, TDimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  explicit ConstExpression(ScalarType c_) : c(c_) {}
  ScalarType c;

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    const std::ptrdiff_t n = static_cast<std::ptrdiff_t>(x.size());
    if (grad) {
      *grad = VectorType(n);
      for (std::ptrdiff_t i = 0; i < n; ++i) (*grad)[i] = ScalarType(0);
    }
    if (hess) {
      *hess = MatrixType(n, n);
      for (std::ptrdiff_t i = 0; i < n; ++i)
        for (std::ptrdiff_t j = 0; j < n; ++j) (*hess)(i, j) = ScalarType(0);
    }
    return c;
  }
};

// min{0, f} and max{0, f} (reference MinZeroExpression :319-358, MaxZeroExpression :362-399): f where it is strictly on
// the kept side of zero, the constant 0 — value, gradient and Hessian — elsewhere (f == 0 counts as clipped in both).
// Host-side nodes: the building blocks of the penalty helpers in function_penalty.h.  They have no device term of their
// own (a solver handed one refuses with that reason): on the device the clipped part of an inequality lives inside the
// augmented-Lagrangian composite (MI355_OBJ_AL_COMPOSITE), which is what ToAugmentedLagrangian / AugmentedLagrangian use.
namespace detail {
template <class F, bool kKeepNegative>
class ClippedAtZero : public FunctionCRTP<ClippedAtZero<F, kKeepNegative>, typename F::ScalarType, F::Differentiability,
                                          F::Dimension> {
 public:
  using Super = FunctionCRTP<ClippedAtZero<F, kKeepNegative>, typename F::ScalarType, F::Differentiability, F::Dimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  explicit ClippedAtZero(const F& f_) : f(f_) {}
  F f;

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr, MatrixType* hess = nullptr) const {
    const ScalarType value = EvaluateUpTo(f, x, grad, hess);
    const bool clipped = kKeepNegative ? (value >= 0) : (value <= 0);
    if (!clipped) return value;
    return ConstExpression<ScalarType, F::Differentiability, F::Dimension>(ScalarType(0))(x, grad, hess);
  }
};
}  // namespace detail
template <class F>
struct MinZeroExpression : public detail::ClippedAtZero<F, true> {
  explicit MinZeroExpression(const F& f_) : detail::ClippedAtZero<F, true>(f_) {}
};
template <class F>
struct MaxZeroExpression : public detail::ClippedAtZero<F, false> {
  explicit MaxZeroExpression(const F& f_) : detail::ClippedAtZero<F, false>(f_) {}
};

// The reference's names for the mode of a binary node and for the nodes themselves (function_expressions.h:74-88, :91,
// :146, :200, :260), for code that spells them out instead of using the operators.
template <typename F, typename G>
struct MinDifferentiability {
  static constexpr DifferentiabilityMode value = WeakerMode(F::Differentiability, G::Differentiability);
};
template <DifferentiabilityMode A, DifferentiabilityMode B>
struct MinDifferentiabilityMode {
  static constexpr DifferentiabilityMode value = WeakerMode(A, B);
};
template <class F, class G>
using AddExpression = SumFunction<F, G>;
template <class F, class G>
using SubExpression = SumFunction<F, ScaledFunction<G>>;   // f + (-1) g: the same bits as f - g (see operator- above)
template <class F>
using MulExpression = ScaledFunction<F>;
template <class F, class G>
using ProdExpression = ProductFunction<F, G>;

// ---------------------------------------------------------------------------------------------
// Device twins of expressions: the record of a node is composed from the records of its operands
// (detail::TwinOf, function_base.h), so `circle - 2.0` has a twin whether `circle` is a SquaredNorm<>,
// a user class with a DeviceTwin() hook or a FunctionExpr wrapped around either.
// ---------------------------------------------------------------------------------------------
namespace detail {
template <class F>
struct TwinOf<ScaledFunction<F>> {
  static cppoptlib::mi355::TwinRecord Make(const ScaledFunction<F>& e) {
    return cppoptlib::mi355::ScaledRecord(static_cast<double>(e.factor()), TwinOf<F>::Make(e.function()));
  }
};
template <class F, class G>
struct TwinOf<SumFunction<F, G>> {
  static cppoptlib::mi355::TwinRecord Make(const SumFunction<F, G>& e) {
    return cppoptlib::mi355::SumRecord(TwinOf<F>::Make(e.left()), TwinOf<G>::Make(e.right()));
  }
};
template <class F, class G>
struct TwinOf<ProductFunction<F, G>> {
  static cppoptlib::mi355::TwinRecord Make(const ProductFunction<F, G>& e) {
    return cppoptlib::mi355::ProductRecord(TwinOf<F>::Make(e.left()), TwinOf<G>::Make(e.right()));
  }
};
template <class F, bool kConstantFirst>
struct TwinOf<OffsetFunction<F, kConstantFirst>> {
  static cppoptlib::mi355::TwinRecord Make(const OffsetFunction<F, kConstantFirst>& e) {
    return cppoptlib::mi355::OffsetRecord(TwinOf<F>::Make(e.function()), static_cast<double>(e.constant()), kConstantFirst);
  }
};
}  // namespace detail

}  // namespace cppoptlib::function
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_EXPRESSIONS_H_
