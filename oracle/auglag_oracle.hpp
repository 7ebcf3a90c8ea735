// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement of the reference's augmented-Lagrangian path:
//   AugmentedLagrangian::Minimize -> OptimizationStep          solver/augmented_lagrangian.h
//     -> ToAugmentedLagrangian (composite assembly)            function_penalty.h:97-246
//     -> inner Lbfgs::Minimize on the composite                solver/lbfgs.h (oracle::Lbfgs)
//     -> multiplier / penalty update, KKT norm, best-iterate filter
//   Progress::Update, IsConstrained branch                     solver/progress.h:162-252
//
// The reference composes arbitrary host functors through expression templates.  The device
// engine evaluates a closed menu of TERMS (objective and constraints alike), so this restatement
// models exactly that menu; every expression node the reference would build for such a problem
// (ConstExpression, AddExpression, SubExpression, MulExpression incl. its c == 0 short circuit,
// ProdExpression, MaxZeroExpression; function_expressions.h:38-388) is applied in the reference's
// own order, so values and gradients are bit-identical to the reference built over
// oracle/eigen_shim (tests/test_oracle.py pins this against oracle/_ref/libref.so).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <type_traits>
#include <vector>

#include "lbfgs_oracle.hpp"
#include "lbfgsb_oracle.hpp"

namespace oracle {

enum TermKind : int {
  kTermRosenbrock = 0,
  kTermDiagQuadratic = 1,
  kTermLinear = 2,
  kTermSquaredNorm = 3,
  // A row of a constraint FAMILY (mi355_al_problem::family_eq / family_ineq): `a.dot(x)` like kTermLinear, but evaluated
  // as the ascending chain under EVERY reduction policy — the device gives each family constraint to one lane, which
  // forms the reference's own chain (csrc/auglag_device.hpp family_values); under the sequential policy the two kinds
  // coincide.  The ordered sums over the family constraints need no twin: the composite below already adds them in
  // constraint order.
  kTermLinearChain = 50,
  // One residual of a least-squares function (MI355_AL_TERM_SQUARED_AFFINE): r = a.dot(x) - c, value r * r, gradient
  // (2 r) a — a reference user's `r * r` / `2 * r * a` for the rows of src/examples/linear_regression.cc:14-39.
  kTermSquaredAffine = 4,
  // twins of the USER term functors of examples/user_al_terms/hs_terms.hpp (MI355_AL_TERM_USER; the functions of the
  // reference's src/test/augmented_lagrangian_test.cc:945-962, :1090-1113, in the reference classes' operation order)
  kTermHs024Objective = 100,
  kTermProductObjective = 101,
  kTermHs029Ellipse = 102,
  // twin of examples/user_objective_svm_dual/svm_dual.hpp as a term (the objective of the reference's
  // src/examples/svm_dual_al.cc:36-60): parameters [n, Q] from the problem's user blob (mi355_al_problem::user_params)
  kTermSvmDual = 103
};
// How the primitive's value v enters the problem: v, `v - k` (SubExpression<F, Const>) or
// `k - v` (SubExpression<Const, F>); function_expressions.h:139-186, :497-518.
enum TermForm : int { kFormPlain = 0, kFormValueMinusK = 1, kFormKMinusValue = 2 };

struct Primitive {
  int kind = kTermLinear;
  std::vector<double> coef;  // a[0..n) and, for the diagonal quadratic, c at [n]
  const double* user = nullptr;  // the problem's user blob (kinds that take their parameters from it)

  double eval(const double* x, double* g, int n, const Reducer& red) const {
    switch (kind) {
      case kTermRosenbrock: {
        Rosenbrock fn;
        return fn.eval(x, g, n, red);
      }
      case kTermDiagQuadratic: {
        double term[1024];
        for (int i = 0; i < n; ++i) {
          term[i] = (coef[i] * x[i]) * x[i];
          g[i] = (2.0 * coef[i]) * x[i];
        }
        return red.sum(term, n) + coef[n];
      }
      case kTermLinear: {  // a.dot(x), gradient a
        for (int i = 0; i < n; ++i) g[i] = coef[i];
        return red.dot(coef.data(), x, n);
      }
      case kTermSquaredAffine: {
        const double r = red.dot(coef.data(), x, n) - coef[n];
        for (int i = 0; i < n; ++i) g[i] = (2.0 * r) * coef[i];
        return r * r;
      }
      case kTermLinearChain: {
        for (int i = 0; i < n; ++i) g[i] = coef[i];
        double s = coef[0] * x[0];
        for (int i = 1; i < n; ++i) s = s + coef[i] * x[i];
        return s;
      }
      case kTermSvmDual: {
        SvmDual fn;
        fn.ns = static_cast<int>(user[0]);
        fn.Q = user + 1;
        return fn.eval(x, g, n, red);
      }
      case kTermHs024Objective:
      case kTermProductObjective:
      case kTermHs029Ellipse: {
        // the device hands x0 / x1 to every lane through a segment sum with zeros: a -0.0 arrives as +0.0
        const bool device_order = red.kind == Reduction::Butterfly;
        const double x0 = device_order ? x[0] + 0.0 : x[0], x1 = device_order ? x[1] + 0.0 : x[1];
        for (int i = 0; i < n; ++i) g[i] = 0.0;
        if (kind == kTermHs024Objective) {
          const double bracket = (x0 - 3.0) * (x0 - 3.0) - 9.0;
          const double scale = 1.0 / (27.0 * std::sqrt(3.0));
          g[0] = 2.0 * (x0 - 3.0) * x1 * x1 * x1 * scale;
          g[1] = 3.0 * bracket * x1 * x1 * scale;
          return bracket * x1 * x1 * x1 * scale;
        }
        if (kind == kTermProductObjective) {
          g[0] = -x1;
          g[1] = -x0;
          return -x0 * x1;
        }
        g[0] = -2.0 * x0;
        g[1] = -4.0 * x1;
        return 48.0 - x0 * x0 - 2.0 * x1 * x1;
      }
      default: {  // x.squaredNorm(), gradient 2 x  (src/examples/constrained_simple2.cc:29-39)
        for (int i = 0; i < n; ++i) g[i] = 2.0 * x[i];
        return red.dot(x, x, n);
      }
    }
  }
};

// A term: the sum of its primitives, left to right (AddExpression, function_expressions.h:91-143: value
// fx_f + fx_g, gradient grad_f + grad_g) — or, `product`, the product of its two primitives (ProdExpression,
// function_expressions.h:282-293: value fx * gx, gradient gx * grad_f + fx * grad_g) — then its form.
struct Term {
  std::vector<Primitive> parts;
  bool product = false;
  int form = kFormPlain;
  double k = 0.0;

  double eval(const double* x, double* g, int n, const Reducer& red) const {
    double v = parts[0].eval(x, g, n, red);
    if (product) {
      double g2[1024];
      const double v2 = parts[1].eval(x, g2, n, red);
      for (int i = 0; i < n; ++i) g[i] = v2 * g[i] + v * g2[i];
      v = v * v2;
    }
    for (size_t r = 1; !product && r < parts.size(); ++r) {
      double g2[1024];
      const double v2 = parts[r].eval(x, g2, n, red);
      v = v + v2;
      for (int i = 0; i < n; ++i) g[i] = g[i] + g2[i];
    }
    if (form == kFormValueMinusK) {
      for (int i = 0; i < n; ++i) g[i] = g[i] - 0.0;
      return v - k;
    }
    if (form == kFormKMinusValue) {
      for (int i = 0; i < n; ++i) g[i] = 0.0 - g[i];
      return k - v;
    }
    return v;
  }
  double value(const double* x, int n, const Reducer& red) const {
    double g[1024];
    return eval(x, g, n, red);
  }
};

struct ConstrainedProblem {  // function_problem.h:44-74
  Term objective;
  std::vector<Term> equality;    // c(x) == 0
  std::vector<Term> inequality;  // c(x) >= 0
};

// ToAugmentedLagrangian(prob, multipliers, penalty), function_penalty.h:239-246.
struct AugLagComposite final : Objective {
  const ConstrainedProblem* prob = nullptr;
  std::vector<double> lambda, mu;
  double rho = 0.0;

  // MulExpression::operator() (function_expressions.h:203-236): c == 0 short-circuits to exact zeros.
  static double scale(double c, double v, double* g, int n) {
    if (c == 0.0) {
      for (int i = 0; i < n; ++i) g[i] = 0.0;
      return 0.0;
    }
    for (int i = 0; i < n; ++i) g[i] = c * g[i];
    return c * v;
  }
  // ProdExpression of a function with itself (:262-271): value fx*gx, gradient gx*grad_f + fx*grad_g.
  static double square(double v, double* g, int n) {
    for (int i = 0; i < n; ++i) g[i] = v * g[i] + v * g[i];
    return v * v;
  }

  double eval(const double* x, double* g, int n, const Reducer& red) const override {
    double tg[1024], part[1024];
    // objective
    const double fv = prob->objective.eval(x, g, n, red);
    // FormLagrangianPart (:97-108)
    double lv = 0.0;
    for (int i = 0; i < n; ++i) part[i] = 0.0;
    for (size_t c = 0; c < prob->equality.size(); ++c) {
      double tv = prob->equality[c].eval(x, tg, n, red);
      tv = scale(lambda[c], tv, tg, n);
      lv = lv + tv;
      for (int i = 0; i < n; ++i) part[i] = part[i] + tg[i];
    }
    double value = fv + lv;
    for (int i = 0; i < n; ++i) g[i] = g[i] + part[i];
    // FormPenaltyPart (:115-127): 0 + rho * (0.5 * (c * c)) per equality
    double pv = 0.0;
    for (int i = 0; i < n; ++i) part[i] = 0.0;
    for (size_t c = 0; c < prob->equality.size(); ++c) {
      double tv = prob->equality[c].eval(x, tg, n, red);
      tv = square(tv, tg, n);
      tv = scale(0.5, tv, tg, n);
      tv = scale(rho, tv, tg, n);
      pv = pv + tv;
      for (int i = 0; i < n; ++i) part[i] = part[i] + tg[i];
    }
    value = value + pv;
    for (int i = 0; i < n; ++i) g[i] = g[i] + part[i];
    // FormInequalityPart (:154-194), Powell-Hestenes-Rockafellar
    double iv = 0.0;
    for (int i = 0; i < n; ++i) part[i] = 0.0;
    if (!(rho <= 0.0)) {
      for (size_t c = 0; c < prob->inequality.size(); ++c) {
        const double m = mu[c];
        double tv = prob->inequality[c].eval(x, tg, n, red);
        tv = scale(rho, tv, tg, n);              // penalty * g
        tv = m - tv;                             // mu - penalty * g   (SubExpression<Const, Mul>)
        for (int i = 0; i < n; ++i) tg[i] = 0.0 - tg[i];
        if (tv <= 0.0) {                         // MaxZeroExpression (:347-363)
          tv = 0.0;
          for (int i = 0; i < n; ++i) tg[i] = 0.0;
        }
        tv = square(tv, tg, n);
        const double half_inv_rho = 1.0 / (2.0 * rho);
        tv = scale(half_inv_rho, tv, tg, n);
        iv = iv + tv;
        for (int i = 0; i < n; ++i) part[i] = part[i] + tg[i];
        const double constant_offset = m * m * half_inv_rho;
        iv = iv - constant_offset;
        for (int i = 0; i < n; ++i) part[i] = part[i] - 0.0;
      }
    }
    value = value + iv;
    for (int i = 0; i < n; ++i) g[i] = g[i] + part[i];
    return value;
  }
};

struct AugLagConfig {  // solver/augmented_lagrangian.h, AugmentedLagrangianConfig
  double penalty_growth_factor = 10.0;
  double violation_shrink_ratio = 0.25;
  bool auto_scale_initial_penalty = true;
  double penalty_auto_objective_scale = 10.0;
  double penalty_auto_min = 1e-8;
  double penalty_auto_max = 1e8;
  int warmup_max_inner_iterations = 10;
  double warmup_inner_gradient_tolerance = 1e-2;
  double multiplier_max = 1e20;
};

struct AugLagStopping {  // the Progress fields the constrained branch reads (progress.h:212-252)
  uint64_t num_iterations = 10000;
  double constraint_threshold = 1e-5;
  double kkt_stationarity_threshold = 1e-4;
};

struct AugLagState {  // AugmentedLagrangeState
  std::vector<double> x;
  std::vector<double> lambda, mu;
  double penalty = 0.0;
  double max_violation = 0.0;
  double max_lagrangian_gradient = std::numeric_limits<double>::infinity();
  bool penalty_was_auto_scaled = false;
};

struct AugLagProgress {
  uint64_t num_iterations = 0;
  double x_delta = 0.0, f_delta = 0.0, gradient_norm = 0.0;
  Status status = NotStarted;
  // accounting (not in the reference)
  uint64_t inner_iterations = 0, nfev = 0, sum_k = 0;
};

// Inner: oracle::Lbfgs or oracle::Lbfgsb (solver_t of the reference)
template <class Inner>
struct AugmentedLagrangianT {
  const ConstrainedProblem* prob;
  Inner inner_template;  // unconstrained_solver_template_
  AugLagConfig config;
  AugLagStopping stopping_progress;
  Reducer red;

  uint64_t outer_iteration_count_ = 0;
  uint64_t inner_iterations_ = 0, nfev_ = 0, sum_k_ = 0;
  // best-iterate filter
  bool best_recorded_ = false;
  AugLagState best_;
  double best_objective_ = 0.0;

  AugmentedLagrangianT(const ConstrainedProblem* p, const Inner& inner, Reducer r)
      : prob(p), inner_template(inner), red(r) {}

  double ComputeAutoScaledPenalty(const std::vector<double>& x) const {
    const int n = static_cast<int>(x.size());
    double objective_magnitude = std::fabs(prob->objective.value(x.data(), n, red));
    objective_magnitude = std::max(objective_magnitude, 1.0);
    double squared_residual_sum = 0.0;
    for (const Term& c : prob->equality) {
      const double value = c.value(x.data(), n, red);
      squared_residual_sum += 0.5 * value * value;
    }
    for (const Term& c : prob->inequality) {
      const double value = c.value(x.data(), n, red);
      if (value < 0.0) squared_residual_sum += 0.5 * value * value;
    }
    const double denom = std::max(squared_residual_sum, 1.0);
    const double rho = config.penalty_auto_objective_scale * objective_magnitude / denom;
    return std::min(std::max(rho, config.penalty_auto_min), config.penalty_auto_max);  // std::clamp
  }
  double ClampEqualityMultiplier(double candidate) const {
    if (!std::isfinite(candidate)) return 0.0;
    return std::min(std::max(candidate, -config.multiplier_max), config.multiplier_max);
  }
  double ClampInequalityMultiplier(double candidate) const {
    if (!std::isfinite(candidate)) return 0.0;
    return std::min(std::max(candidate, 0.0), config.multiplier_max);
  }
  double KktNorm(const AugLagState& s) const {
    const int n = static_cast<int>(s.x.size());
    std::vector<double> sum_grad(n), buf(n);
    prob->objective.eval(s.x.data(), sum_grad.data(), n, red);
    for (size_t i = 0; i < prob->equality.size(); ++i) {
      prob->equality[i].eval(s.x.data(), buf.data(), n, red);
      for (int k = 0; k < n; ++k) sum_grad[k] = sum_grad[k] + s.lambda[i] * buf[k];
    }
    for (size_t j = 0; j < prob->inequality.size(); ++j) {
      prob->inequality[j].eval(s.x.data(), buf.data(), n, red);
      for (int k = 0; k < n; ++k) sum_grad[k] = sum_grad[k] - s.mu[j] * buf[k];
    }
    // HasProjectedGradientInfNorm<solver_t> (augmented_lagrangian.h:47-58): the TEMPLATE solver's bounds
    if constexpr (std::is_same<Inner, Lbfgsb>::value) {
      if (!inner_template.lower.empty()) return inner_template.ProjectedGradientInfNorm(s.x, sum_grad);
      return Reducer::amax(sum_grad.data(), n);  // bounds never set: gradient.lpNorm<Infinity>() (lbfgsb.h:107-109)
    }
    double sup = 0.0;
    for (int k = 0; k < n; ++k) sup = std::max(sup, std::fabs(sum_grad[k]));
    return sup;
  }
  void RecordBest(const AugLagState& c, double objective) {
    best_recorded_ = true;
    best_ = c;
    best_objective_ = objective;
  }
  void UpdateBestIterate(const AugLagState& c) {
    constexpr double filter_feasibility_tolerance = 1e-5;
    const int n = static_cast<int>(c.x.size());
    const double objective = prob->objective.value(c.x.data(), n, red);
    bool finite = std::isfinite(objective) && std::isfinite(c.max_violation);
    for (int k = 0; finite && k < n; ++k) finite = finite && std::isfinite(c.x[k]);
    if (!finite) return;
    if (!best_recorded_) return RecordBest(c, objective);
    const bool cf = c.max_violation <= filter_feasibility_tolerance;
    const bool bf = best_.max_violation <= filter_feasibility_tolerance;
    if (cf && !bf) return RecordBest(c, objective);
    if (!cf && bf) return;
    if (cf && bf) {
      if (objective < best_objective_) RecordBest(c, objective);
      return;
    }
    if (c.max_violation < best_.max_violation ||
        (c.max_violation == best_.max_violation && objective < best_objective_))
      RecordBest(c, objective);
  }

  AugLagState OptimizationStep(const AugLagState& state) {
    ++outer_iteration_count_;
    AugLagState next = state;
    const int n = static_cast<int>(state.x.size());
    if (outer_iteration_count_ == 1 && config.auto_scale_initial_penalty && !next.penalty_was_auto_scaled &&
        next.penalty == 0.0) {
      next.penalty = ComputeAutoScaledPenalty(next.x);
      next.penalty_was_auto_scaled = true;
    }
    AugLagComposite composite;
    composite.prob = prob;
    composite.lambda = next.lambda;
    composite.mu = next.mu;
    composite.rho = next.penalty;
    Inner working_inner = inner_template;  // ConfigureInnerSubproblem
    working_inner.stopping_progress.f_delta = 0.0;
    const bool has_general = !prob->equality.empty() || !prob->inequality.empty();
    if (outer_iteration_count_ == 1 && has_general && config.warmup_max_inner_iterations > 0) {
      working_inner.stopping_progress.num_iterations = static_cast<uint64_t>(config.warmup_max_inner_iterations);
      working_inner.stopping_progress.gradient_norm = config.warmup_inner_gradient_tolerance;
    }
    Progress inner_progress;
    const State solved = working_inner.Minimize(composite, next.x, &inner_progress);
    inner_iterations_ += inner_progress.num_iterations;
    nfev_ += working_inner.nfev;
    sum_k_ += working_inner.sum_k;
    next.x = solved.x;
    const double penalty = next.penalty;
    double max_violation = 0.0;
    for (size_t i = 0; i < prob->equality.size(); ++i) {
      const double cv = prob->equality[i].value(next.x.data(), n, red);
      max_violation = std::max(max_violation, std::fabs(cv));
      next.lambda[i] = ClampEqualityMultiplier(next.lambda[i] + penalty * cv);
    }
    for (size_t i = 0; i < prob->inequality.size(); ++i) {
      const double cv = prob->inequality[i].value(next.x.data(), n, red);
      const double violation = std::max(0.0, -cv);
      max_violation = std::max(max_violation, violation);
      next.mu[i] = ClampInequalityMultiplier(std::max(0.0, next.mu[i] - penalty * cv));
    }
    next.max_lagrangian_gradient = KktNorm(next);
    next.max_violation = max_violation;
    UpdateBestIterate(next);
    const bool shrank = max_violation <= config.violation_shrink_ratio * state.max_violation;
    next.penalty = shrank ? penalty : penalty * config.penalty_growth_factor;
    return next;
  }

  // Progress::Update, IsConstrained branch (progress.h:162-252).
  void UpdateProgress(AugLagProgress* p, const AugLagState& prev, const AugLagState& cur) const {
    const int n = static_cast<int>(cur.x.size());
    AugLagComposite pf, cf;
    pf.prob = cf.prob = prob;
    pf.lambda = prev.lambda; pf.mu = prev.mu; pf.rho = prev.penalty;
    cf.lambda = cur.lambda;  cf.mu = cur.mu;  cf.rho = cur.penalty;
    std::vector<double> pg(n), cg(n);
    const double previous_value = pf.eval(prev.x.data(), pg.data(), n, red);
    const double current_value = cf.eval(cur.x.data(), cg.data(), n, red);
    p->num_iterations++;
    p->f_delta = std::fabs(current_value - previous_value);
    double m = 0.0;
    for (int i = 0; i < n; ++i) m = std::max(m, std::fabs(cur.x[i] - prev.x[i]));
    p->x_delta = m;
    p->gradient_norm = Reducer::amax(cg.data(), n);
    if (stopping_progress.num_iterations > 0 && p->num_iterations > stopping_progress.num_iterations) {
      p->status = IterationLimit;
      return;
    }
    if (!std::isfinite(cur.max_violation) || !std::isfinite(cur.max_lagrangian_gradient)) {
      p->status = IterationLimit;
      return;
    }
    const bool primal_feasible = std::fabs(cur.max_violation) <= stopping_progress.constraint_threshold;
    const bool kkt_stationary = (stopping_progress.kkt_stationarity_threshold <= 0.0) ||
                                (cur.max_lagrangian_gradient <= stopping_progress.kkt_stationarity_threshold);
    p->status = (primal_feasible && kkt_stationary) ? Finished : Continue;
  }

  // AugmentedLagrangian::Minimize (best-iterate wrapper) over Solver::Minimize (solver.h:181-224).
  AugLagState Minimize(const AugLagState& initial, AugLagProgress* progress_out) {
    best_recorded_ = false;
    outer_iteration_count_ = 0;
    inner_iterations_ = nfev_ = sum_k_ = 0;
    AugLagProgress progress;
    AugLagState cur = initial;
    do {
      const AugLagState prev = cur;
      cur = OptimizationStep(prev);
      UpdateProgress(&progress, prev, cur);
    } while (progress.status == Continue);
    if (best_recorded_) {
      cur.x = best_.x;
      cur.lambda = best_.lambda;
      cur.mu = best_.mu;
      cur.penalty = best_.penalty;
      cur.max_violation = best_.max_violation;
      cur.max_lagrangian_gradient = best_.max_lagrangian_gradient;
    }
    progress.inner_iterations = inner_iterations_;
    progress.nfev = nfev_;
    progress.sum_k = sum_k_;
    if (progress_out) *progress_out = progress;
    return cur;
  }
};

using AugmentedLagrangian = AugmentedLagrangianT<Lbfgs>;

}  // namespace oracle
