// cppoptlib/function_penalty.h — multiplier / penalty state and the augmented-Lagrangian composite.
//
// Mirrors include/cppoptlib/function_penalty.h of the reference: LagrangeMultiplierState :65-78, PenaltyState
// :82-87 and ToAugmentedLagrangian :239-246,
//     L(x) = f + sum_i lambda_i c_i + sum_i rho (0.5 (c_i c_i))
//              + sum_j [ (1/(2 rho)) max(0, mu_j - rho g_j)^2 - mu_j^2 / (2 rho) ]      (PHR form, :154-194).
// The returned function evaluates on the host node by node in the order of the reference's expression templates
// and carries a device twin (MI355_OBJ_AL_COMPOSITE), so it can be handed to Lbfgs like any other objective —
// which is what an augmented-Lagrangian step does with it, and what a pure penalty-method experiment needs.
#ifndef INCLUDE_CPPOPTLIB_FUNCTION_PENALTY_H_
#define INCLUDE_CPPOPTLIB_FUNCTION_PENALTY_H_

#include <initializer_list>
#include <vector>

#include "function_problem.h"
#include "mi355/context.h"

namespace cppoptlib::function {

// The three quadratic penalties of the reference (function_penalty.h:40-61), as expressions over the constraint c:
//   equality        c = 0  ->  0.5 (c c)
//   inequality >=   c >= 0 ->  0.5 (min{0, c})^2
//   inequality <    c < 0  ->  0.5 (max{0, c})^2
// Host-evaluating expression nodes (function_expressions.h); the equality penalty keeps a device term wherever the
// product of the operand with itself has one, the clipped ones are host-only (see MinZeroExpression).
template <typename F>
auto QuadraticEqualityPenalty(const F& c) {
  return 0.5 * (c * c);
}
template <typename F>
auto QuadraticInequalityPenaltyGe(const F& c) {
  const MinZeroExpression<F> negative_part(c);
  return 0.5 * (negative_part * negative_part);
}
template <typename F>
auto QuadraticInequalityPenaltyLt(const F& c) {
  const MaxZeroExpression<F> positive_part(c);
  return 0.5 * (positive_part * positive_part);
}

template <typename TScalar>
struct LagrangeMultiplierState {
  std::vector<TScalar> equality_multipliers;
  std::vector<TScalar> inequality_multipliers;
  LagrangeMultiplierState(size_t num_eq = 0, size_t num_ineq = 0, TScalar value = TScalar{0})
      : equality_multipliers(num_eq, value), inequality_multipliers(num_ineq, value) {}
  LagrangeMultiplierState(std::initializer_list<TScalar> eq, std::initializer_list<TScalar> ineq)
      : equality_multipliers(eq), inequality_multipliers(ineq) {}
};

template <typename TScalar>
struct PenaltyState {
  TScalar penalty;
  explicit PenaltyState(TScalar pen = TScalar(0)) : penalty(pen) {}
};

template <int TDimension = kDynamicDimension>
class AugmentedLagrangianFunction
    : public FunctionCRTP<AugmentedLagrangianFunction<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super =
      FunctionCRTP<AugmentedLagrangianFunction<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  using Term = TermExpr<TDimension>;  // = FunctionExpr<double, First, TDimension>, the problem's own wrappers
  static constexpr int kDeviceObjective = MI355_OBJ_AL_COMPOSITE;

  AugmentedLagrangianFunction(Term objective, std::vector<Term> eq, std::vector<Term> ineq,
                              LagrangeMultiplierState<double> multipliers, PenaltyState<double> penalty)
      : objective_(std::move(objective)), eq_(std::move(eq)), ineq_(std::move(ineq)),
        multipliers_(std::move(multipliers)), penalty_(penalty) {
    if (multipliers_.equality_multipliers.size() != eq_.size() ||
        multipliers_.inequality_multipliers.size() != ineq_.size())
      cppoptlib::mi355::Fail("ToAugmentedLagrangian: one multiplier per constraint");
  }

  // C-ABI layout of MI355_OBJ_AL_COMPOSITE: n_eq, n_ineq, rows, then per term (parts, form, k), then per row
  // (kind, coefficient row [n + 1]).
  std::vector<double> DeviceParams(int n) const {
    using cppoptlib::mi355::RequireTerm;
    using cppoptlib::mi355::TwinTerm;
    std::vector<const TwinTerm*> terms{&RequireTerm(objective_, "ToAugmentedLagrangian: the objective")};
    for (const Term& t : eq_) terms.push_back(&RequireTerm(t, "ToAugmentedLagrangian: an equality constraint"));
    for (const Term& t : ineq_) terms.push_back(&RequireTerm(t, "ToAugmentedLagrangian: an inequality constraint"));
    int rows = 0;
    for (const TwinTerm* t : terms) rows += t->rows();
    std::vector<double> p{static_cast<double>(eq_.size()), static_cast<double>(ineq_.size()), static_cast<double>(rows)};
    for (const TwinTerm* t : terms) {
      p.push_back(t->parts());
      p.push_back(t->form);
      p.push_back(t->constant());
    }
    for (const TwinTerm* t : terms) {
      const std::vector<double> coef = t->Coefficients(n);
      if (static_cast<int>(coef.size()) != t->rows() * (n + 1))
        cppoptlib::mi355::Fail("constrained problem: a term was built for another dimension");
      for (int r = 0; r < t->rows(); ++r) {
        p.push_back(t->kinds()[static_cast<size_t>(r)]);
        p.insert(p.end(), coef.begin() + static_cast<std::ptrdiff_t>(r) * (n + 1),
                 coef.begin() + static_cast<std::ptrdiff_t>(r + 1) * (n + 1));
      }
    }
    return p;
  }
  std::vector<double> DevicePerProblem() const {
    std::vector<double> row = multipliers_.equality_multipliers;
    row.insert(row.end(), multipliers_.inequality_multipliers.begin(), multipliers_.inequality_multipliers.end());
    row.push_back(penalty_.penalty);
    return row;
  }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const std::ptrdiff_t n = x.size();
    const double rho = penalty_.penalty;
    VectorType g(n), tg(n), part(n);
    // MulExpression (reference function_expressions.h:203-236): c == 0 gives exact zeros
    auto scale = [&](double c, double v, VectorType& grad) {
      for (std::ptrdiff_t i = 0; i < n; ++i) grad[i] = (c == 0.0) ? 0.0 : c * grad[i];
      return (c == 0.0) ? 0.0 : c * v;
    };
    // ProdExpression of a function with itself (:262-271)
    auto square = [&](double v, VectorType& grad) {
      for (std::ptrdiff_t i = 0; i < n; ++i) grad[i] = v * grad[i] + v * grad[i];
      return v * v;
    };
    double value = objective_(x, &g);
    double lv = 0.0;  // FormLagrangianPart
    for (std::ptrdiff_t i = 0; i < n; ++i) part[i] = 0.0;
    for (size_t c = 0; c < eq_.size(); ++c) {
      double tv = eq_[c](x, &tg);
      tv = scale(multipliers_.equality_multipliers[c], tv, tg);
      lv = lv + tv;
      for (std::ptrdiff_t i = 0; i < n; ++i) part[i] = part[i] + tg[i];
    }
    value = value + lv;
    for (std::ptrdiff_t i = 0; i < n; ++i) g[i] = g[i] + part[i];
    double pv = 0.0;  // FormPenaltyPart
    for (std::ptrdiff_t i = 0; i < n; ++i) part[i] = 0.0;
    for (size_t c = 0; c < eq_.size(); ++c) {
      double tv = eq_[c](x, &tg);
      tv = square(tv, tg);
      tv = scale(0.5, tv, tg);
      tv = scale(rho, tv, tg);
      pv = pv + tv;
      for (std::ptrdiff_t i = 0; i < n; ++i) part[i] = part[i] + tg[i];
    }
    value = value + pv;
    for (std::ptrdiff_t i = 0; i < n; ++i) g[i] = g[i] + part[i];
    double iv = 0.0;  // FormInequalityPart
    for (std::ptrdiff_t i = 0; i < n; ++i) part[i] = 0.0;
    if (!(rho <= 0.0)) {
      const double half_inv_rho = 1.0 / (2.0 * rho);
      for (size_t c = 0; c < ineq_.size(); ++c) {
        const double mu = multipliers_.inequality_multipliers[c];
        double tv = ineq_[c](x, &tg);
        tv = scale(rho, tv, tg);
        tv = mu - tv;
        const bool clamp = tv <= 0.0;  // MaxZeroExpression
        for (std::ptrdiff_t i = 0; i < n; ++i) tg[i] = clamp ? 0.0 : 0.0 - tg[i];
        if (clamp) tv = 0.0;
        tv = square(tv, tg);
        tv = scale(half_inv_rho, tv, tg);
        iv = iv + tv;
        iv = iv - mu * mu * half_inv_rho;
        for (std::ptrdiff_t i = 0; i < n; ++i) part[i] = (part[i] + tg[i]) - 0.0;
      }
    }
    value = value + iv;
    for (std::ptrdiff_t i = 0; i < n; ++i) g[i] = g[i] + part[i];
    if (gradient) *gradient = g;
    return value;
  }

 private:
  Term objective_;
  std::vector<Term> eq_, ineq_;
  LagrangeMultiplierState<double> multipliers_;
  PenaltyState<double> penalty_;
};

// The parts of the composite as expressions of their own (reference FormLagrangianPart :97-110, FormPenaltyPart :115-128,
// FormInequalityPart :154-194, ToPenalty :203-222): type-erased sums built term by term in the reference's order,
//   sum_i lambda_i c_i,   sum_i rho (0.5 (c_i c_i)),   sum_j [ (1 / (2 rho)) max{0, mu_j - rho g_j}^2 - mu_j^2 / (2 rho) ],
//   f + sum_i rho (0.5 c_i^2) + sum_j rho (0.5 min{0, g_j}^2).
// They evaluate on the host (spot checks, closed-form tests, a penalty-method experiment's bookkeeping); what the engine
// SOLVES is the whole composite — ToAugmentedLagrangian below / the AugmentedLagrangian solver — so these carry no twin.
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
FunctionExpr<TScalar, Mode, TDim> FormLagrangianPart(const ConstrainedOptimizationProblem<TScalar, Mode, TDim>& prob,
                                                     const LagrangeMultiplierState<TScalar>& mult_state) {
  FunctionExpr<TScalar, Mode, TDim> part = ConstExpression<TScalar, Mode, TDim>(TScalar(0));
  for (size_t i = 0; i < prob.equality_constraints.size(); ++i)
    part = part + mult_state.equality_multipliers[i] * prob.equality_constraints[i];
  return part;
}
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
FunctionExpr<TScalar, Mode, TDim> FormPenaltyPart(const ConstrainedOptimizationProblem<TScalar, Mode, TDim>& prob,
                                                  const PenaltyState<TScalar>& pen_state) {
  FunctionExpr<TScalar, Mode, TDim> part = ConstExpression<TScalar, Mode, TDim>(TScalar(0));
  for (size_t i = 0; i < prob.equality_constraints.size(); ++i)
    part = part + pen_state.penalty * QuadraticEqualityPenalty(prob.equality_constraints[i]);
  return part;
}
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
FunctionExpr<TScalar, Mode, TDim> FormInequalityPart(const ConstrainedOptimizationProblem<TScalar, Mode, TDim>& prob,
                                                     const LagrangeMultiplierState<TScalar>& mult_state,
                                                     const PenaltyState<TScalar>& pen_state) {
  FunctionExpr<TScalar, Mode, TDim> part = ConstExpression<TScalar, Mode, TDim>(TScalar(0));
  const TScalar rho = pen_state.penalty;
  if (rho <= TScalar(0)) return part;   // (1 / (2 rho) undefined: the inequalities contribute nothing, as in the reference)
  const TScalar half_inv_rho = TScalar(1) / (TScalar(2) * rho);
  for (size_t j = 0; j < prob.inequality_constraints.size(); ++j) {
    const TScalar mu = mult_state.inequality_multipliers[j];
    auto shifted = mu - rho * prob.inequality_constraints[j];
    const MaxZeroExpression<decltype(shifted)> positive_part(shifted);
    part = part + half_inv_rho * (positive_part * positive_part);
    part = part - ConstExpression<TScalar, Mode, TDim>(mu * mu * half_inv_rho);
  }
  return part;
}
template <typename TScalar, DifferentiabilityMode Mode, int TDim>
FunctionExpr<TScalar, Mode, TDim> ToPenalty(const ConstrainedOptimizationProblem<TScalar, Mode, TDim>& prob,
                                            const PenaltyState<TScalar>& pen_state) {
  FunctionExpr<TScalar, Mode, TDim> part = ConstExpression<TScalar, Mode, TDim>(TScalar(0));
  for (size_t i = 0; i < prob.equality_constraints.size(); ++i)
    part = part + pen_state.penalty * QuadraticEqualityPenalty(prob.equality_constraints[i]);
  for (size_t i = 0; i < prob.inequality_constraints.size(); ++i)
    part = part + pen_state.penalty * QuadraticInequalityPenaltyGe(prob.inequality_constraints[i]);
  return prob.objective + part;
}

template <typename TScalar, DifferentiabilityMode Mode, int TDim>
AugmentedLagrangianFunction<TDim> ToAugmentedLagrangian(const ConstrainedOptimizationProblem<TScalar, Mode, TDim>& prob,
                                                        const LagrangeMultiplierState<TScalar>& mult_state,
                                                        const PenaltyState<TScalar>& pen_state) {
  static_assert(std::is_same<TScalar, double>::value,
                "the composite as an objective of its own is built for double problems (a float problem still solves "
                "through AugmentedLagrangian, which widens at the boundary)");
  return AugmentedLagrangianFunction<TDim>(prob.objective, prob.equality_constraints, prob.inequality_constraints,
                                           mult_state, pen_state);
}

}  // namespace cppoptlib::function
#endif  // INCLUDE_CPPOPTLIB_FUNCTION_PENALTY_H_
