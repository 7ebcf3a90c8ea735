// cppoptlib/function_problem.h — constrained problems whose terms have device twins.
//
// Mirrors include/cppoptlib/function_problem.h of the reference (ConstrainedOptimizationProblem :44-74 and its
// deduction guides :81-101): an objective, equality constraints c(x) = 0 and inequality constraints g(x) >= 0.
// The reference stores type-erased FunctionExpr objects around arbitrary host functors; the device evaluates a
// closed menu of terms (mi355_al_term_kind in include/mi355_lbfgs.h), so the type-erased holder here, TermExpr,
// accepts exactly the shapes that have a twin:
//     P          a primitive with kAlTermKind   (Rosenbrock, DiagQuadratic, LinearForm, SquaredNorm)
//     P - k      OffsetFunction<P, false>       (`circle - 2.0`)
//     k - P      OffsetFunction<P, true>        (`2.0 - circle`)
// Anything else does not convert, which is the compile-time error that replaces a CPU fallback.
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_PROBLEM_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_PROBLEM_H_

#include <functional>
#include <initializer_list>
#include <type_traits>
#include <utility>
#include <vector>

#include "function_base.h"
#include "function_expressions.h"

namespace cppoptlib::mi355 {
template <class F, class = void>
struct IsAlPrimitive : std::false_type {};
template <class F>
struct IsAlPrimitive<F, std::void_t<decltype(F::kAlTermKind), decltype(std::declval<const F&>().AlCoefficients(1))>>
    : std::true_type {};
}  // namespace cppoptlib::mi355

namespace cppoptlib::function {

template <int TDimension = kDynamicDimension>
class TermExpr : public FunctionCRTP<TermExpr<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<TermExpr<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;

  template <class P, class = std::enable_if_t<cppoptlib::mi355::IsAlPrimitive<P>::value>>
  TermExpr(const P& p)  // NOLINT: implicit, like the reference's FunctionExpr
      : kind_(P::kAlTermKind), form_(MI355_AL_FORM_PLAIN), k_(0),
        eval_([p](const VectorType& x, VectorType* g) { return p(x, g); }),
        coef_([p](int n) { return p.AlCoefficients(n); }) {}
  template <class P, bool kConstantFirst, class = std::enable_if_t<cppoptlib::mi355::IsAlPrimitive<P>::value>>
  TermExpr(const OffsetFunction<P, kConstantFirst>& e)  // NOLINT
      : kind_(P::kAlTermKind), form_(kConstantFirst ? MI355_AL_FORM_K_MINUS_VALUE : MI355_AL_FORM_VALUE_MINUS_K),
        k_(e.constant()), eval_([e](const VectorType& x, VectorType* g) { return e(x, g); }),
        coef_([p = e.function()](int n) { return p.AlCoefficients(n); }) {}

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const { return eval_(x, gradient); }

  int kind() const { return kind_; }
  int form() const { return form_; }
  double constant() const { return k_; }
  // coefficient row [n + 1] of the C-ABI; empty when the term was built for another dimension
  std::vector<double> Coefficients(int n) const { return coef_(n); }

 private:
  int kind_, form_;
  double k_;
  std::function<ScalarType(const VectorType&, VectorType*)> eval_;
  std::function<std::vector<double>(int)> coef_;
};

template <typename TScalar = double, DifferentiabilityMode Mode = DifferentiabilityMode::First,
          int TDimension = kDynamicDimension>
struct ConstrainedOptimizationProblem {
  static_assert(std::is_same<TScalar, double>::value, "the MI355X engine computes in fp64");
  static_assert(Mode == DifferentiabilityMode::First, "terms are first-order functions");
  static constexpr int Dimension = TDimension;
  using ScalarType = TScalar;
  using VectorType = cppoptlib::mi355::Vector<TScalar, TDimension>;
  using MatrixType = cppoptlib::mi355::SquareMatrix<TScalar, TDimension>;
  using ObjectiveFunctionType = TermExpr<TDimension>;
  using ConstraintFunctionType = TermExpr<TDimension>;
  static constexpr DifferentiabilityMode Differentiability = Mode;

  const TermExpr<TDimension> objective;                            // f(x)
  const std::vector<TermExpr<TDimension>> equality_constraints;    // c(x) == 0
  const std::vector<TermExpr<TDimension>> inequality_constraints;  // c(x) >= 0

  ConstrainedOptimizationProblem(TermExpr<TDimension> obj, std::vector<TermExpr<TDimension>> eq_constraints = {},
                                 std::vector<TermExpr<TDimension>> ineq_constraints = {})
      : objective(std::move(obj)),
        equality_constraints(std::move(eq_constraints)),
        inequality_constraints(std::move(ineq_constraints)) {}
};

template <class F, class = std::enable_if_t<IsFunction<F>::value>>
ConstrainedOptimizationProblem(const F&)
    -> ConstrainedOptimizationProblem<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;
template <class F, class = std::enable_if_t<IsFunction<F>::value>>
ConstrainedOptimizationProblem(const F&, std::initializer_list<TermExpr<F::Dimension>>)
    -> ConstrainedOptimizationProblem<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;
template <class F, class = std::enable_if_t<IsFunction<F>::value>>
ConstrainedOptimizationProblem(const F&, std::initializer_list<TermExpr<F::Dimension>>,
                               std::initializer_list<TermExpr<F::Dimension>>)
    -> ConstrainedOptimizationProblem<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;

}  // namespace cppoptlib::function
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_PROBLEM_H_
