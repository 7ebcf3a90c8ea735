// cppoptlib/function_problem.h — constrained problems whose terms have device twins.
//
// The reference's ConstrainedOptimizationProblem<TScalar, Mode, TDimension> (include/cppoptlib/function_problem.h:44-84)
// with its three deduction guides (:86-105): an objective, equality constraints c(x) = 0 and inequality constraints
// g(x) >= 0, each stored as a type-erased FunctionExpr<TScalar, Mode, TDimension>.  The reference's wrappers hold arbitrary
// host functors; here every wrapper also carries the function's device twin (function_base.h, mi355/device_twin.h), and a
// solver reads its TERM facet — the closed menu the device evaluates (mi355_al_term_kind in include/mi355_lbfgs.h):
//     S          a primitive (Rosenbrock, DiagQuadratic, LinearForm, SquaredNorm, a least-squares residual, a user
//                functor compiled into the library), or a left-nested sum of primitives P1 + P2 + ...  (AddExpression)
//     P1 * P2    the product of two primitives                                                          (ProdExpression)
//     S - k      OffsetFunction<S, false>   (`circle - 2.0`)          -1 * (S - k) = k - S, -1 * S = 0 - S
//     k - S      OffsetFunction<S, true>    (`2.0 - circle`)
// Anything else converts (it is a function, it evaluates on the host) but has no term facet, and AugmentedLagrangian
// refuses the problem with the reason — the run-time form of "no CPU fallback".
// The constraint vectors may be of any length (function_problem.h:57-84 of the reference): up to
// MI355_AL_MAX_CONSTRAINTS terms of each kind go into the device's term table, the affine constraints `LinearForm(a) - k`
// that FOLLOW them travel as a constraint family (a matrix; up to mi355_auglag_family_capacity(n) rows).
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_PROBLEM_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_PROBLEM_H_

#include <initializer_list>
#include <type_traits>
#include <utility>
#include <vector>

#include "function_base.h"
#include "function_expressions.h"

namespace cppoptlib::function {

// A term of a double-precision first-order problem: the name the round-3..5 headers used for their own holder; now the
// reference's wrapper itself.
template <int TDimension = kDynamicDimension>
using TermExpr = FunctionExpr<double, DifferentiabilityMode::First, TDimension>;

template <typename TScalar = double, DifferentiabilityMode Mode = DifferentiabilityMode::First,
          int TDimension = kDynamicDimension>
struct ConstrainedOptimizationProblem {
  static_assert(std::is_floating_point<TScalar>::value,
                "ScalarType must be float or double (the MI355X engine computes in fp64 either way)");
  static constexpr int Dimension = TDimension;
  using ScalarType = TScalar;
  using VectorType = cppoptlib::mi355::Vector<TScalar, TDimension>;
  using MatrixType = cppoptlib::mi355::SquareMatrix<TScalar, TDimension>;
  using ObjectiveFunctionType = FunctionExpr<TScalar, Mode, TDimension>;
  using ConstraintFunctionType = FunctionExpr<TScalar, Mode, TDimension>;
  static constexpr DifferentiabilityMode Differentiability = Mode;

  const FunctionExpr<TScalar, Mode, TDimension> objective;                            // f(x)
  const std::vector<FunctionExpr<TScalar, Mode, TDimension>> equality_constraints;    // c(x) == 0
  const std::vector<FunctionExpr<TScalar, Mode, TDimension>> inequality_constraints;  // c(x) >= 0

  ConstrainedOptimizationProblem(const FunctionExpr<TScalar, Mode, TDimension> obj,
                                 const std::vector<FunctionExpr<TScalar, Mode, TDimension>> eq_constraints = {},
                                 const std::vector<FunctionExpr<TScalar, Mode, TDimension>> ineq_constraints = {})
      : objective(std::move(obj)),
        equality_constraints(std::move(eq_constraints)),
        inequality_constraints(std::move(ineq_constraints)) {}
};

// the reference's guides (function_problem.h:86-105): from type-erased operands
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
ConstrainedOptimizationProblem(const FunctionExpr<TScalar, Mode, TDim>&) -> ConstrainedOptimizationProblem<TScalar, Mode, TDim>;
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
ConstrainedOptimizationProblem(const FunctionExpr<TScalar, Mode, TDim>&, std::initializer_list<FunctionExpr<TScalar, Mode, TDim>>)
    -> ConstrainedOptimizationProblem<TScalar, Mode, TDim>;
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
ConstrainedOptimizationProblem(const FunctionExpr<TScalar, Mode, TDim>&, std::initializer_list<FunctionExpr<TScalar, Mode, TDim>>,
                               std::initializer_list<FunctionExpr<TScalar, Mode, TDim>>)
    -> ConstrainedOptimizationProblem<TScalar, Mode, TDim>;

// ... and, beyond the reference, straight from function objects of a static type (`ConstrainedOptimizationProblem
// prob(LinearForm<>(a), {circle - 2.0})`): a first-order problem of the objective's scalar type and dimension
template <class F, class = std::enable_if_t<IsFunction<F>::value && !IsFunctionExpr<F>::value>>
ConstrainedOptimizationProblem(const F&)
    -> ConstrainedOptimizationProblem<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;
template <class F, class = std::enable_if_t<IsFunction<F>::value && !IsFunctionExpr<F>::value>>
ConstrainedOptimizationProblem(
    const F&, std::initializer_list<FunctionExpr<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>>)
    -> ConstrainedOptimizationProblem<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;
template <class F, class = std::enable_if_t<IsFunction<F>::value && !IsFunctionExpr<F>::value>>
ConstrainedOptimizationProblem(
    const F&, std::initializer_list<FunctionExpr<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>>,
    std::initializer_list<FunctionExpr<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>>)
    -> ConstrainedOptimizationProblem<typename F::ScalarType, DifferentiabilityMode::First, F::Dimension>;

}  // namespace cppoptlib::function

namespace cppoptlib::mi355 {
// The TERM facet of a wrapped function, or the refusal a solver shows.
template <class Expr>
const TwinTerm& RequireTerm(const Expr& f, const char* what) {
  if (!f.device_twin.term.valid)
    Fail(std::string(what) + " has no device twin as a term of a constrained problem — " + f.device_twin.why_no_term +
         "; the MI355X engine has no CPU fallback");
  return f.device_twin.term;
}
}  // namespace cppoptlib::mi355
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_PROBLEM_H_
