// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C entry point around the UNMODIFIED reference augmented-Lagrangian solver
// (solver/augmented_lagrangian.h, function_penalty.h, function_problem.h), compiled where the
// headers lie under /root/reference/include over oracle/eigen_shim, into oracle/_ref/libref.so.
// The outer loop, the composite assembly, the inner L-BFGS and the constrained Progress::Update
// executed here ARE the reference's; only the primitive functors below are ours — they are the
// terms of the device engine's menu written the way a reference user would write them
// (compare src/examples/constrained_simple2.cc:13-39).
#include <cmath>
#include <cstdint>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/linesearch/hager_zhang.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"

namespace {

using cppoptlib::function::FunctionXd;
using FExpr = cppoptlib::function::FunctionExprXd;

class RosenbrockTerm : public FunctionXd<RosenbrockTerm> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    double f = 0.0;
    for (int i = 0; i + 1 < n; ++i) {
      const double t1 = 1.0 - x[i];
      const double t2 = x[i + 1] - x[i] * x[i];
      const double term = t1 * t1 + (100.0 * t2) * t2;
      f = (i == 0) ? term : f + term;
    }
    if (gradient) {
      *gradient = VectorType::Zero(n);
      for (int i = 0; i < n; ++i) {
        const bool has_a = (i + 1 < n), has_b = (i > 0);
        double a = 0.0, b = 0.0;
        if (has_a) a = -2.0 * (1.0 - x[i]) + (200.0 * (x[i + 1] - x[i] * x[i])) * (-2.0 * x[i]);
        if (has_b) b = 200.0 * (x[i] - x[i - 1] * x[i - 1]);
        (*gradient)[i] = (has_a && has_b) ? (a + b) : (has_a ? a : b);
      }
    }
    return f;
  }
};

class DiagQuadraticTerm : public FunctionXd<DiagQuadraticTerm> {
 public:
  std::vector<double> a;
  double c = 0.0;
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    double f = 0.0;
    if (gradient) *gradient = VectorType::Zero(n);
    for (int i = 0; i < n; ++i) {
      const double term = (a[i] * x[i]) * x[i];
      f = (i == 0) ? term : f + term;
      if (gradient) (*gradient)[i] = (2.0 * a[i]) * x[i];
    }
    return f + c;
  }
};

class LinearTerm : public FunctionXd<LinearTerm> {
 public:
  VectorType a;
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) *gradient = a;
    return a.dot(x);
  }
};

// one residual of a least-squares function, as a reference user writes it: r = a.x - c, f = r r, grad = (2 r) a
// (the rows of src/examples/linear_regression.cc:14-39)
class SquaredAffineTerm : public FunctionXd<SquaredAffineTerm> {
 public:
  VectorType a;
  double c = 0.0;
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const double r = a.dot(x) - c;
    if (gradient) *gradient = (2.0 * r) * a;
    return r * r;
  }
};

// the `Circle` of src/examples/constrained_simple2.cc:29-39
class SquaredNormTerm : public FunctionXd<SquaredNormTerm> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) *gradient = 2 * x;
    return x.squaredNorm();
  }
};

// The functions of the reference's non-convex tests (src/test/augmented_lagrangian_test.cc:945-962 HS024 objective,
// :1090-1113 the product objective and the ellipse of the 2-D HS029), written as a reference user writes a functor:
// kinds 100-102, the device's user term functors (examples/user_al_terms/hs_terms.hpp).
class Hs024ObjectiveTerm : public FunctionXd<Hs024ObjectiveTerm> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const double shifted = x[0] - 3.0;
    const double bracket = shifted * shifted - 9.0;
    const double scale = 1.0 / (27.0 * std::sqrt(3.0));
    if (gradient) {
      *gradient = VectorType::Zero(x.size());
      (*gradient)[0] = 2.0 * shifted * x[1] * x[1] * x[1] * scale;
      (*gradient)[1] = 3.0 * bracket * x[1] * x[1] * scale;
    }
    return bracket * x[1] * x[1] * x[1] * scale;
  }
};
class ProductObjectiveTerm : public FunctionXd<ProductObjectiveTerm> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) {
      *gradient = VectorType::Zero(x.size());
      (*gradient)[0] = -x[1];
      (*gradient)[1] = -x[0];
    }
    return -x[0] * x[1];
  }
};
class Hs029EllipseTerm : public FunctionXd<Hs029EllipseTerm> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) {
      *gradient = VectorType::Zero(x.size());
      (*gradient)[0] = -2.0 * x[0];
      (*gradient)[1] = -4.0 * x[1];
    }
    return 48.0 - x[0] * x[0] - 2.0 * x[1] * x[1];
  }
};

// The objective of the reference's src/examples/svm_dual_al.cc:36-60 (own main() and anonymous namespace, so the class
// is restated; kernel_matrix is handed in, the Eigen expressions as the loops a loop-based Eigen evaluates them with):
// kind 103, the device's SvmDual functor as a term; parameters [n, Q] from ref_auglag_set_user_params.
std::vector<double> g_user_params;
class SvmDualObjectiveTerm : public FunctionXd<SvmDualObjectiveTerm> {
 public:
  const double* Q = nullptr;
  int ns = 0;
  ScalarType operator()(const VectorType& alpha, VectorType* grad = nullptr) const {
    std::vector<double> q(static_cast<size_t>(ns));
    for (int i = 0; i < ns; ++i) {                        // kernel_matrix * alpha
      double acc = Q[static_cast<size_t>(i) * ns] * alpha[0];
      for (int j = 1; j < ns; ++j) acc = acc + Q[static_cast<size_t>(i) * ns + j] * alpha[j];
      q[static_cast<size_t>(i)] = acc;
    }
    double aq = alpha[0] * q[0], sa = alpha[0];           // alpha.dot(q_alpha), alpha.sum()
    for (int i = 1; i < ns; ++i) {
      aq = aq + alpha[i] * q[static_cast<size_t>(i)];
      sa = sa + alpha[i];
    }
    if (grad) {
      *grad = VectorType::Zero(ns);
      for (int i = 0; i < ns; ++i) (*grad)[i] = q[static_cast<size_t>(i)] - 1.0;
    }
    return 0.5 * aq - sa;
  }
};

FExpr make_primitive(int kind, const double* coef, int n) {
  if (kind == 103) {
    SvmDualObjectiveTerm t;
    t.ns = static_cast<int>(g_user_params.at(0));
    t.Q = g_user_params.data() + 1;
    return t;
  }
  if (kind == 100) return Hs024ObjectiveTerm();
  if (kind == 101) return ProductObjectiveTerm();
  if (kind == 102) return Hs029EllipseTerm();
  if (kind == 0) return RosenbrockTerm();
  if (kind == 1) {
    DiagQuadraticTerm t;
    t.a.assign(coef, coef + n);
    t.c = coef[n];
    return t;
  }
  if (kind == 4) {
    SquaredAffineTerm t;
    t.a = Eigen::VectorXd(n);
    for (int i = 0; i < n; ++i) t.a[i] = coef[i];
    t.c = coef[n];
    return t;
  }
  if (kind == 2 || kind == 50) {   // (50: a row of a constraint family — to the reference just another LinearTerm)
    LinearTerm t;
    t.a = Eigen::VectorXd(n);
    for (int i = 0; i < n; ++i) t.a[i] = coef[i];
    return t;
  }
  return SquaredNormTerm();
}

// `count` primitives starting at table row `row` summed left to right with the reference's operator+
// (AddExpression), then `F`, `F - k` or `k - F`.
// count = -2: the product of the two primitives with the reference's operator* (ProdExpression).
FExpr make_term(const int32_t* kinds, const double* coef, int n, int row, int count, int form, double k) {
  FExpr base = make_primitive(kinds[row], coef + static_cast<size_t>(row) * (n + 1), n);
  if (count == -2) {
    FExpr other = make_primitive(kinds[row + 1], coef + static_cast<size_t>(row + 1) * (n + 1), n);
    FExpr prod = base * other;
    base = prod;
    count = 1;
  }
  for (int r = 1; r < count; ++r) {
    FExpr next = make_primitive(kinds[row + r], coef + static_cast<size_t>(row + r) * (n + 1), n);
    base = base + next;
  }
  if (form == 1) return base - k;
  if (form == 2) return k - base;
  return base;
}

}  // namespace

extern "C" {

struct ref_al_config {
  double penalty_growth_factor, violation_shrink_ratio;
  int32_t auto_scale_initial_penalty;
  double penalty_auto_objective_scale, penalty_auto_min, penalty_auto_max;
  int32_t warmup_max_inner_iterations;
  double warmup_inner_gradient_tolerance, multiplier_max;
  uint64_t outer_num_iterations;
  double constraint_threshold, kkt_stationarity_threshold;
  int32_t loop;  // (device execution mode of the product's struct; not used here)
};
struct ref_al_inner_stop {
  uint64_t num_iterations;
  double x_delta;
  int32_t x_delta_violations;
  double f_delta;
  int32_t f_delta_violations;
  int32_t f_delta_relative;
  double gradient_norm;
  int32_t gradient_norm_relative;
  int32_t past;
  double past_delta;
};
struct ref_al_progress {
  int32_t status;
  uint32_t num_iterations;
  double x_delta, f_delta, gradient_norm;
  uint64_t inner_iterations, nfev, sum_k;
};

}  // extern "C"

namespace {
template <class Inner>
int run_auglag(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms, const double* ks,
               const double* coef, const ref_al_config* cfg, const ref_al_inner_stop* st, double* x, double* lambda,
               double* mu, double* penalty, double* violation, double* kkt, ref_al_progress* prog,
               const double* ks_batch, const int32_t* parts, const double* lower = nullptr,
               const double* upper = nullptr) {
  using cppoptlib::solver::AugmentedLagrangeState;
  using Problem = cppoptlib::function::ConstrainedOptimizationProblem<
      double, cppoptlib::function::DifferentiabilityMode::First, Eigen::Dynamic>;
  // ks_batch (null, or [B][1 + n_eq + n_ineq]): problem b is built with its own constants
  auto build = [&](int64_t b) {
    const double* k = ks_batch ? ks_batch + b * (1 + n_eq + n_ineq) : ks;
    int row = 0;
    auto term = [&](int t) {
      const int count = parts ? parts[t] : 1;
      FExpr e = make_term(kinds, coef, n, row, count, forms[t], k[t]);
      row += count < 0 ? -count : count;
      return e;
    };
    FExpr objective = term(0);
    std::vector<FExpr> eq, ineq;
    for (int t = 0; t < n_eq; ++t) eq.push_back(term(1 + t));
    for (int t = 0; t < n_ineq; ++t) ineq.push_back(term(1 + n_eq + t));
    return Problem(objective, eq, ineq);
  };
  Inner inner;
  if constexpr (cppoptlib::solver::HasProjectedGradientInfNorm<Inner>::value) {  // Lbfgsb::SetBounds
    if (lower) {
      Eigen::VectorXd lo(n), up(n);
      for (int i = 0; i < n; ++i) {
        lo[i] = lower[i];
        up[i] = upper[i];
      }
      inner.SetBounds(lo, up);
    }
  }
  inner.stopping_progress.num_iterations = st->num_iterations;
  inner.stopping_progress.x_delta = st->x_delta;
  inner.stopping_progress.x_delta_violations = st->x_delta_violations;
  inner.stopping_progress.f_delta = st->f_delta;
  inner.stopping_progress.f_delta_violations = st->f_delta_violations;
  inner.stopping_progress.f_delta_relative = st->f_delta_relative != 0;
  inner.stopping_progress.gradient_norm = st->gradient_norm;
  inner.stopping_progress.gradient_norm_relative = st->gradient_norm_relative != 0;
  inner.stopping_progress.past = st->past;
  inner.stopping_progress.past_delta = st->past_delta;
  cppoptlib::solver::AugmentedLagrangianConfig<double> config;
  config.penalty_growth_factor = cfg->penalty_growth_factor;
  config.violation_shrink_ratio = cfg->violation_shrink_ratio;
  config.auto_scale_initial_penalty = cfg->auto_scale_initial_penalty != 0;
  config.penalty_auto_objective_scale = cfg->penalty_auto_objective_scale;
  config.penalty_auto_min = cfg->penalty_auto_min;
  config.penalty_auto_max = cfg->penalty_auto_max;
  config.warmup_max_inner_iterations = cfg->warmup_max_inner_iterations;
  config.warmup_inner_gradient_tolerance = cfg->warmup_inner_gradient_tolerance;
  config.multiplier_max = cfg->multiplier_max;
  for (int64_t b = 0; b < B; ++b) {
    const Problem prob = build(b);
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(prob, inner, config);
    solver.stopping_progress.num_iterations = cfg->outer_num_iterations;
    solver.stopping_progress.constraint_threshold = cfg->constraint_threshold;
    solver.stopping_progress.kkt_stationarity_threshold = cfg->kkt_stationarity_threshold;
    Eigen::VectorXd x0(n);
    for (int i = 0; i < n; ++i) x0[i] = x[b * n + i];
    AugmentedLagrangeState<double> state(x0, n_eq, n_ineq, penalty[b]);
    state.max_violation = violation[b];  // in/out (augmented_lagrangian.h:435 reads the incoming value)
    for (int i = 0; i < n_eq; ++i) state.multiplier_state.equality_multipliers[i] = lambda[b * n_eq + i];
    for (int i = 0; i < n_ineq; ++i) state.multiplier_state.inequality_multipliers[i] = mu[b * n_ineq + i];
    auto [sol, pr] = solver.Minimize(state);
    for (int i = 0; i < n; ++i) x[b * n + i] = sol.x[i];
    for (int i = 0; i < n_eq; ++i) lambda[b * n_eq + i] = sol.multiplier_state.equality_multipliers[i];
    for (int i = 0; i < n_ineq; ++i) mu[b * n_ineq + i] = sol.multiplier_state.inequality_multipliers[i];
    penalty[b] = sol.penalty_state.penalty;
    violation[b] = sol.max_violation;
    kkt[b] = sol.max_lagrangian_gradient;
    if (prog) {
      prog[b].status = static_cast<int32_t>(pr.status);
      prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog[b].x_delta = pr.x_delta;
      prog[b].f_delta = pr.f_delta;
      prog[b].gradient_norm = pr.gradient_norm;
      prog[b].inner_iterations = 0;
      prog[b].nfev = 0;
      prog[b].sum_k = 0;
    }
  }
  return 0;
}

}  // namespace

extern "C" {

// Term t of the problem: kinds[t], forms[t], ks[t], coef + t*(n+1); t = 0 is the objective, then
// n_eq equalities, then n_ineq inequalities (g >= 0).  x, lambda, mu, penalty are in/out.
// The parameter blob of the terms that take one (kind 103: [n, Q]); kept until replaced.
int ref_auglag_set_user_params(const double* params, int64_t count) {
  if (count < 0 || (count > 0 && !params)) return -1;
  g_user_params.assign(params, params + count);
  return 0;
}

// linesearch: the LineSearch template argument of the inner Lbfgs (0 MoreThuente, 1 HagerZhang).
int ref_auglag_minimize_batch(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms,
                              const double* ks, const double* coef, const ref_al_config* cfg,
                              const ref_al_inner_stop* st, double* x, double* lambda, double* mu, double* penalty,
                              double* violation, double* kkt, ref_al_progress* prog, int linesearch,
                              const double* ks_batch, const int32_t* parts) {
  if (linesearch == 1)
    return run_auglag<cppoptlib::solver::Lbfgs<FExpr, 10, cppoptlib::solver::linesearch::HagerZhang>>(
        n, B, n_eq, n_ineq, kinds, forms, ks, coef, cfg, st, x, lambda, mu, penalty, violation, kkt, prog, ks_batch,
        parts);
  return run_auglag<cppoptlib::solver::Lbfgs<FExpr>>(n, B, n_eq, n_ineq, kinds, forms, ks, coef, cfg, st, x, lambda,
                                                     mu, penalty, violation, kkt, prog, ks_batch, parts);
}

// The same with Lbfgsb<FunctionExpr> (m = 5) as the inner solver; lower / upper: n doubles, or both null.
int ref_auglag_box_minimize_batch(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms,
                                  const double* ks, const double* coef, const ref_al_config* cfg,
                                  const ref_al_inner_stop* st, double* x, double* lambda, double* mu, double* penalty,
                                  double* violation, double* kkt, ref_al_progress* prog, int linesearch,
                                  const double* ks_batch, const double* lower, const double* upper,
                                  const int32_t* parts) {
  if (linesearch == 1)
    return run_auglag<cppoptlib::solver::Lbfgsb<FExpr, 5, cppoptlib::solver::linesearch::HagerZhang>>(
        n, B, n_eq, n_ineq, kinds, forms, ks, coef, cfg, st, x, lambda, mu, penalty, violation, kkt, prog, ks_batch,
        parts, lower, upper);
  return run_auglag<cppoptlib::solver::Lbfgsb<FExpr>>(n, B, n_eq, n_ineq, kinds, forms, ks, coef, cfg, st, x, lambda,
                                                      mu, penalty, violation, kkt, prog, ks_batch, parts, lower,
                                                      upper);
}

}  // extern "C"
