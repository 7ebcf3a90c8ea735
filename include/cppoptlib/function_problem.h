// cppoptlib/function_problem.h — constrained problems whose terms have device twins.
//
// Mirrors include/cppoptlib/function_problem.h of the reference (ConstrainedOptimizationProblem :44-74 and its
// deduction guides :81-101): an objective, equality constraints c(x) = 0 and inequality constraints g(x) >= 0.
// The reference stores type-erased FunctionExpr objects around arbitrary host functors; the device evaluates a
// closed menu of terms (mi355_al_term_kind in include/mi355_lbfgs.h), so the type-erased holder here, TermExpr,
// accepts exactly the shapes that have a twin:
//     S          a primitive with kAlTermKind (Rosenbrock, DiagQuadratic, LinearForm, SquaredNorm), or a sum of
//                primitives P1 + P2 + ...       (SumFunction, the reference's AddExpression)
//     S - k      OffsetFunction<S, false>       (`circle - 2.0`)
//     k - S      OffsetFunction<S, true>        (`2.0 - circle`)
// Anything else does not convert, which is the compile-time error that replaces a CPU fallback.
// The constraint vectors may be of any length (function_problem.h:57-84 of the reference): up to
// MI355_AL_MAX_CONSTRAINTS terms of each kind go into the device's term table, the affine constraints `LinearForm(a) - k`
// that FOLLOW them travel as a constraint family (a matrix; up to mi355_auglag_family_capacity(n) rows).
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

// A primitive, or a left-nested sum of primitives `(P1 + P2) + P3` — the order AddExpression evaluates
// `P1 + P2 + P3` in, which is the order the device sums a term's primitives in.
template <class F>
struct IsAlSum : IsAlPrimitive<F> {};
template <class L, class P>
struct IsAlSum<cppoptlib::function::SumFunction<L, P>>
    : std::integral_constant<bool, IsAlSum<L>::value && IsAlPrimitive<P>::value> {};

// the product of two primitives `P1 * P2` (ProdExpression): a term of its own kind (MI355_AL_PARTS_PRODUCT)
template <class F>
struct IsAlProduct : std::false_type {};
template <class L, class R>
struct IsAlProduct<cppoptlib::function::ProductFunction<L, R>>
    : std::integral_constant<bool, IsAlPrimitive<L>::value && IsAlPrimitive<R>::value> {};

// A USER primitive whose device functor takes a parameter blob of its own (kTermParamsFromProblem: the same blob it takes
// as an objective — the kernel matrix of src/examples/svm_dual_al.cc:45-50) hands it over through AlUserParams().
template <class F, class = void>
struct HasAlUserParams : std::false_type {};
template <class F>
struct HasAlUserParams<F, std::void_t<decltype(std::declval<const F&>().AlUserParams())>> : std::true_type {};

// kinds and coefficient-row builders of the primitives of such a sum, left to right
struct AlPrimitiveList {
  std::vector<int> kinds;
  std::vector<std::function<std::vector<double>(int)>> rows;
  std::vector<std::function<std::vector<double>()>> user_params;  // of the primitives that have one
};
template <class P, class = std::enable_if_t<IsAlPrimitive<P>::value>>
void AppendAlPrimitives(const P& p, AlPrimitiveList* out) {
  out->kinds.push_back(P::kAlTermKind);
  out->rows.push_back([p](int n) { return p.AlCoefficients(n); });
  if constexpr (HasAlUserParams<P>::value) out->user_params.push_back([p]() { return p.AlUserParams(); });
}
template <class L, class P>
void AppendAlPrimitives(const cppoptlib::function::SumFunction<L, P>& s, AlPrimitiveList* out) {
  AppendAlPrimitives(s.left(), out);
  AppendAlPrimitives(s.right(), out);
}
}  // namespace cppoptlib::mi355

namespace cppoptlib::function {

template <int TDimension = kDynamicDimension>
class TermExpr : public FunctionCRTP<TermExpr<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<TermExpr<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;

  template <class S, class = std::enable_if_t<cppoptlib::mi355::IsAlSum<S>::value>>
  TermExpr(const S& s)  // NOLINT: implicit, like the reference's FunctionExpr
      : form_(MI355_AL_FORM_PLAIN), k_(0), eval_([s](const VectorType& x, VectorType* g) { return s(x, g); }) {
    cppoptlib::mi355::AppendAlPrimitives(s, &prims_);
  }
  template <class S, bool kConstantFirst, class = std::enable_if_t<cppoptlib::mi355::IsAlSum<S>::value>>
  TermExpr(const OffsetFunction<S, kConstantFirst>& e)  // NOLINT
      : form_(kConstantFirst ? MI355_AL_FORM_K_MINUS_VALUE : MI355_AL_FORM_VALUE_MINUS_K), k_(e.constant()),
        eval_([e](const VectorType& x, VectorType* g) { return e(x, g); }) {
    cppoptlib::mi355::AppendAlPrimitives(e.function(), &prims_);
  }

  // `P1 * P2`, `P1 * P2 - k`, `k - P1 * P2`: the reference's ProdExpression of two functions as a term
  template <class L, class R, class = std::enable_if_t<cppoptlib::mi355::IsAlProduct<ProductFunction<L, R>>::value>>
  TermExpr(const ProductFunction<L, R>& p)  // NOLINT
      : form_(MI355_AL_FORM_PLAIN), k_(0), product_(true),
        eval_([p](const VectorType& x, VectorType* g) { return p(x, g); }) {
    cppoptlib::mi355::AppendAlPrimitives(p.left(), &prims_);
    cppoptlib::mi355::AppendAlPrimitives(p.right(), &prims_);
  }
  template <class L, class R, bool kConstantFirst,
            class = std::enable_if_t<cppoptlib::mi355::IsAlProduct<ProductFunction<L, R>>::value>>
  TermExpr(const OffsetFunction<ProductFunction<L, R>, kConstantFirst>& e)  // NOLINT
      : form_(kConstantFirst ? MI355_AL_FORM_K_MINUS_VALUE : MI355_AL_FORM_VALUE_MINUS_K), k_(e.constant()), product_(true),
        eval_([e](const VectorType& x, VectorType* g) { return e(x, g); }) {
    cppoptlib::mi355::AppendAlPrimitives(e.function().left(), &prims_);
    cppoptlib::mi355::AppendAlPrimitives(e.function().right(), &prims_);
  }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const { return eval_(x, gradient); }

  // `LinearForm(a)` or `LinearForm(a) - k`: an affine constraint a . x - k.  Constraint vectors longer than the term table
  // holds (MI355_AL_MAX_CONSTRAINTS per kind) travel as a FAMILY — a matrix of such rows (mi355_al_problem.family_*,
  // solver/augmented_lagrangian.h) — which is how src/examples/svm_primal_al.cc:139-147 with its 200 constraints runs.
  bool IsAffineRow() const {
    return !product_ && prims_.kinds.size() == 1 && prims_.kinds[0] == MI355_AL_TERM_LINEAR &&
           (form_ == MI355_AL_FORM_PLAIN || form_ == MI355_AL_FORM_VALUE_MINUS_K);
  }
  // mi355_al_problem.parts of this term: the number of primitives summed, or MI355_AL_PARTS_PRODUCT
  int parts() const { return product_ ? MI355_AL_PARTS_PRODUCT : static_cast<int>(prims_.kinds.size()); }
  int rows() const { return static_cast<int>(prims_.kinds.size()); }
  const std::vector<int>& kinds() const { return prims_.kinds; }
  int form() const { return form_; }
  double constant() const { return k_; }
  // mi355_al_problem.user_params of this term's primitives (empty: none of them takes a blob)
  std::vector<std::vector<double>> UserParams() const {
    std::vector<std::vector<double>> all;
    for (const auto& blob : prims_.user_params) all.push_back(blob());
    return all;
  }
  // the coefficient rows [parts][n + 1] of the C-ABI, concatenated; empty when a primitive was built for another
  // dimension
  std::vector<double> Coefficients(int n) const {
    std::vector<double> all;
    for (const auto& row : prims_.rows) {
      const std::vector<double> r = row(n);
      if (static_cast<int>(r.size()) != n + 1) return {};
      all.insert(all.end(), r.begin(), r.end());
    }
    return all;
  }

 private:
  int form_;
  double k_;
  bool product_ = false;
  std::function<ScalarType(const VectorType&, VectorType*)> eval_;
  cppoptlib::mi355::AlPrimitiveList prims_;
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
