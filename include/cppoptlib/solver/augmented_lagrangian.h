// cppoptlib/solver/augmented_lagrangian.h — augmented Lagrangian on the MI355X, cppoptlib-shaped.
//
// Drop-in for the reference's solver/augmented_lagrangian.h: AugmentedLagrangianConfig (:64-196),
// AugmentedLagrangeState (:201-240) and AugmentedLagrangian<ProblemType, solver_t> (:245-700) with the same
// constructor, Minimize overloads and return type.  The whole outer loop — auto-scaled initial penalty, the
// inner L-BFGS solves, multiplier and penalty updates, KKT norm, best-iterate filter, the constrained stopping
// test — runs on the GPU behind mi355_auglag_minimize_batch (include/mi355_lbfgs.h).  `solver_t` is an
// Lbfgs<...> or an Lbfgsb<...> (box on x by the inner solver, mi355_auglag_box_minimize_batch) whose history
// size, line search, bounds and stopping_progress configure the inner solves, as in the reference.
//
// New here: MinimizeBatch — many start states of the same problem in one call.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_AUGMENTED_LAGRANGIAN_H_
#define INCLUDE_CPPOPTLIB_SOLVER_AUGMENTED_LAGRANGIAN_H_

#include <initializer_list>
#include <limits>
#include <memory>
#include <tuple>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "../function_penalty.h"
#include "../function_problem.h"
#include "../mi355/context.h"
#include "solver.h"

namespace cppoptlib::solver {

// Inner solvers with a box (Lbfgsb): the outer loop passes their bounds on and measures stationarity with the
// projected gradient norm (reference :47-58, HasProjectedGradientInfNorm).
template <class InnerSolver, class = void>
struct HasBox : std::false_type {};
template <class InnerSolver>
struct HasBox<InnerSolver, std::void_t<decltype(std::declval<const InnerSolver&>().LowerBound())>> : std::true_type {};

template <typename TScalar>
struct AugmentedLagrangianConfig {
  TScalar penalty_growth_factor = TScalar{10};
  TScalar violation_shrink_ratio = TScalar{0.25};
  bool auto_scale_initial_penalty = true;
  TScalar penalty_auto_objective_scale = TScalar{10};
  TScalar penalty_auto_min = TScalar{1e-8};
  TScalar penalty_auto_max = TScalar{1e8};
  int warmup_max_inner_iterations = 10;
  TScalar warmup_inner_gradient_tolerance = TScalar{1e-2};
  TScalar multiplier_max = TScalar{1e20};
  TScalar kkt_gradient_tolerance = TScalar{1e-4};  // carried for source compatibility; the reference never reads it
};

template <typename TScalar, int TDimension = cppoptlib::function::kDynamicDimension>
struct AugmentedLagrangeState {
  static constexpr bool IsConstrained = true;
  using VectorType = cppoptlib::mi355::Vector<TScalar, TDimension>;
  VectorType x;
  cppoptlib::function::LagrangeMultiplierState<TScalar> multiplier_state;
  cppoptlib::function::PenaltyState<TScalar> penalty_state;
  TScalar max_violation;
  TScalar max_lagrangian_gradient;
  bool penalty_was_auto_scaled;

  AugmentedLagrangeState(const VectorType& init_x, std::initializer_list<TScalar> eq_multipliers,
                         std::initializer_list<TScalar> ineq_multipliers, TScalar penalty)
      : x(init_x), multiplier_state(eq_multipliers, ineq_multipliers), penalty_state(penalty), max_violation(0),
        max_lagrangian_gradient(std::numeric_limits<TScalar>::infinity()), penalty_was_auto_scaled(false) {}
  AugmentedLagrangeState(const VectorType& init_x, size_t num_eq, size_t num_ineq, TScalar penalty = TScalar(0))
      : x(init_x), multiplier_state(num_eq, num_ineq, TScalar(0)), penalty_state(penalty), max_violation(0),
        max_lagrangian_gradient(std::numeric_limits<TScalar>::infinity()), penalty_was_auto_scaled(false) {}
};

template <typename ProblemType, typename solver_t>
class AugmentedLagrangian
    : public Solver<ProblemType, AugmentedLagrangeState<typename ProblemType::ScalarType, ProblemType::Dimension>> {
 public:
  using StateType = AugmentedLagrangeState<typename ProblemType::ScalarType, ProblemType::Dimension>;
  using Superclass = Solver<ProblemType, StateType>;
  using ProgressType = typename Superclass::ProgressType;
  using ScalarType = typename ProblemType::ScalarType;
  using VectorType = typename ProblemType::VectorType;
  using MatrixType = typename ProblemType::MatrixType;

  AugmentedLagrangian(const ProblemType& problem, const solver_t& unconstrained_solver,
                      AugmentedLagrangianConfig<ScalarType> config = {})
      : problem_(problem), unconstrained_solver_template_(unconstrained_solver), config_(config) {}

  void SetContext(std::shared_ptr<cppoptlib::mi355::Context> ctx) { ctx_ = std::move(ctx); }

  std::tuple<StateType, ProgressType> Minimize(const StateType& state) { return Minimize(problem_, state); }

  std::tuple<StateType, ProgressType> Minimize(const ProblemType& function, const StateType& state) override {
    this->step_callback_(function, state, ProgressType());
    auto out = MinimizeBatch(function, std::vector<StateType>{state});
    this->step_callback_(function, std::get<0>(out[0]), std::get<1>(out[0]));
    return out[0];
  }

  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const std::vector<StateType>& states) {
    return MinimizeBatch(problem_, states);
  }

  // Every start state (x, multipliers, penalty) is solved independently, all of them in lock step on the GPU.
  // term_constants: empty, or one row per state with 1 + n_eq + n_ineq constants that replace the k of the
  // problem's terms (`F - k`, `k - F`) for that state — B different problems of one shape in one call.
  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(
      const ProblemType& function, const std::vector<StateType>& states,
      const std::vector<std::vector<double>>& term_constants = {}) {
    std::vector<std::tuple<StateType, ProgressType>> result;
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return result;
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    const int n = static_cast<int>(states[0].x.size());
    const int n_eq = static_cast<int>(function.equality_constraints.size());
    const int n_ineq = static_cast<int>(function.inequality_constraints.size());
    // every wrapped function as a TERM of the device menu (or the refusal that replaces a CPU fallback)
    using cppoptlib::mi355::TwinTerm;
    const TwinTerm& objective_term = cppoptlib::mi355::RequireTerm(function.objective, "AugmentedLagrangian: the objective");
    std::vector<const TwinTerm*> eq_terms, ineq_terms;
    for (const auto& c : function.equality_constraints)
      eq_terms.push_back(&cppoptlib::mi355::RequireTerm(c, "AugmentedLagrangian: an equality constraint"));
    for (const auto& c : function.inequality_constraints)
      ineq_terms.push_back(&cppoptlib::mi355::RequireTerm(c, "AugmentedLagrangian: an inequality constraint"));
    // A constraint vector longer than the device's term table (MI355_AL_MAX_CONSTRAINTS per kind) is split: its leading
    // constraints stay table terms, the trailing run of affine constraints `LinearForm(a) - k` becomes a constraint FAMILY
    // (mi355_al_problem.family_*: one matrix row each).  The order of the constraints — and of their multipliers in the
    // state — is unchanged: the C-ABI places the family rows after the table's terms of their kind.
    auto table_count = [](const std::vector<const TwinTerm*>& v) {
      size_t first_family = v.size();
      if (v.size() > static_cast<size_t>(MI355_AL_MAX_CONSTRAINTS))
        while (first_family > 0 && v[first_family - 1]->IsAffineRow()) --first_family;
      if (first_family > static_cast<size_t>(MI355_AL_MAX_CONSTRAINTS))
        cppoptlib::mi355::Fail("AugmentedLagrangian: more than MI355_AL_MAX_CONSTRAINTS constraints of one kind that are "
                               "not affine (`LinearForm(a) - k`); only affine constraints travel as a family");
      return static_cast<int>(first_family);
    };
    const int t_eq = table_count(eq_terms), t_ineq = table_count(ineq_terms);
    const int f_eq = n_eq - t_eq, f_ineq = n_ineq - t_ineq;
    if (f_eq + f_ineq > 0 && !term_constants.empty())
      cppoptlib::mi355::Fail("AugmentedLagrangian: per-state term constants are not available with constraint families");
    std::vector<double> family_eq, family_ineq;   // rows (a_i[0..n), k_i)
    auto add_family_row = [&](const TwinTerm& t, std::vector<double>* rows) {
      std::vector<double> r = t.Coefficients(n);
      if (static_cast<int>(r.size()) != n + 1)
        cppoptlib::mi355::Fail("AugmentedLagrangian: a constraint was built for another dimension");
      r[static_cast<size_t>(n)] = t.constant();
      rows->insert(rows->end(), r.begin(), r.end());
    };
    for (int i = t_eq; i < n_eq; ++i) add_family_row(*eq_terms[static_cast<size_t>(i)], &family_eq);
    for (int i = t_ineq; i < n_ineq; ++i) add_family_row(*ineq_terms[static_cast<size_t>(i)], &family_ineq);
    // problem description (host arrays of the C-ABI)
    std::vector<int32_t> kinds, forms, parts;
    std::vector<double> ks, coef, user_params;
    auto add = [&](const TwinTerm& t) {
      for (const std::vector<double>& blob : t.UserParams()) {  // one blob per problem (mi355_al_problem.user_params)
        if (!user_params.empty() && blob != user_params)
          cppoptlib::mi355::Fail("AugmentedLagrangian: the user terms of one problem share one parameter blob");
        user_params = blob;
      }
      const std::vector<double> rows = t.Coefficients(n);
      if (static_cast<int>(rows.size()) != t.rows() * (n + 1))
        cppoptlib::mi355::Fail("AugmentedLagrangian: a term was built for another dimension");
      parts.push_back(t.parts());
      for (int kind : t.kinds()) kinds.push_back(kind);
      forms.push_back(t.form);
      ks.push_back(t.constant());
      coef.insert(coef.end(), rows.begin(), rows.end());
    };
    add(objective_term);
    for (int i = 0; i < t_eq; ++i) add(*eq_terms[static_cast<size_t>(i)]);
    for (int i = 0; i < t_ineq; ++i) add(*ineq_terms[static_cast<size_t>(i)]);
    if (kinds.size() > static_cast<size_t>(MI355_AL_MAX_ROWS))
      cppoptlib::mi355::Fail("AugmentedLagrangian: the problem's terms hold more than MI355_AL_MAX_ROWS primitives");
    mi355_al_problem p{};
    p.n = n;
    p.n_eq = t_eq;
    p.n_ineq = t_ineq;
    p.n_family_eq = f_eq;
    p.n_family_ineq = f_ineq;
    p.family_eq = family_eq.empty() ? nullptr : family_eq.data();
    p.family_ineq = family_ineq.empty() ? nullptr : family_ineq.data();
    p.kinds = kinds.data();
    p.forms = forms.data();
    p.ks = ks.data();
    p.coef = coef.data();
    p.parts = parts.data();
    p.user_params = user_params.empty() ? nullptr : user_params.data();
    p.user_params_count = static_cast<int64_t>(user_params.size());
    mi355_al_config c;
    c.penalty_growth_factor = config_.penalty_growth_factor;
    c.violation_shrink_ratio = config_.violation_shrink_ratio;
    c.auto_scale_initial_penalty = config_.auto_scale_initial_penalty ? 1 : 0;
    c.penalty_auto_objective_scale = config_.penalty_auto_objective_scale;
    c.penalty_auto_min = config_.penalty_auto_min;
    c.penalty_auto_max = config_.penalty_auto_max;
    c.warmup_max_inner_iterations = config_.warmup_max_inner_iterations;
    c.warmup_inner_gradient_tolerance = config_.warmup_inner_gradient_tolerance;
    c.multiplier_max = config_.multiplier_max;
    c.outer_num_iterations = static_cast<uint64_t>(this->stopping_progress.num_iterations);
    c.constraint_threshold = this->stopping_progress.constraint_threshold;
    c.kkt_stationarity_threshold = this->stopping_progress.kkt_stationarity_threshold;
    c.loop = MI355_AL_LOOP_AUTO;
    const mi355_lbfgs_stop inner_stop = unconstrained_solver_template_.stopping_progress.ToDeviceStop();

    const size_t b = static_cast<size_t>(B);
    std::vector<double> x(b * n), lambda(b * n_eq), mu(b * n_ineq), penalty(b), violation(b), kkt(b);
    std::vector<mi355_al_progress> prog(b);
    for (size_t i = 0; i < b; ++i) {
      const StateType& s = states[i];
      if (static_cast<int>(s.x.size()) != n ||
          static_cast<int>(s.multiplier_state.equality_multipliers.size()) != n_eq ||
          static_cast<int>(s.multiplier_state.inequality_multipliers.size()) != n_ineq)
        cppoptlib::mi355::Fail("AugmentedLagrangian: state does not match the problem");
      // (a state carrying penalty_was_auto_scaled = true with penalty 0 would skip auto-scaling in the
      //  reference; such a state cannot come out of a solve, so it is rejected rather than modelled)
      if (s.penalty_was_auto_scaled && s.penalty_state.penalty == 0)
        cppoptlib::mi355::Fail("AugmentedLagrangian: auto-scaled state with a zero penalty");
      for (int j = 0; j < n; ++j) x[i * n + j] = s.x[j];
      for (int j = 0; j < n_eq; ++j) lambda[i * n_eq + j] = s.multiplier_state.equality_multipliers[j];
      for (int j = 0; j < n_ineq; ++j) mu[i * n_ineq + j] = s.multiplier_state.inequality_multipliers[j];
      penalty[i] = s.penalty_state.penalty;
      violation[i] = s.max_violation;  // in/out: read by the first outer step's penalty-growth test (:435 of the reference)
    }
    std::vector<double> constants;
    if (!term_constants.empty()) {
      if (term_constants.size() != b) cppoptlib::mi355::Fail("AugmentedLagrangian: one row of term constants per state");
      for (const auto& row : term_constants) {
        if (static_cast<int>(row.size()) != 1 + t_eq + t_ineq)
          cppoptlib::mi355::Fail("AugmentedLagrangian: a row of term constants holds 1 + n_eq + n_ineq values");
        constants.insert(constants.end(), row.begin(), row.end());
      }
    }
    if constexpr (HasBox<solver_t>::value) {
      const std::vector<double>& lo = unconstrained_solver_template_.LowerBound();
      const std::vector<double>& up = unconstrained_solver_template_.UpperBound();
      if (!lo.empty() && static_cast<int>(lo.size()) != n) cppoptlib::mi355::Fail("SetBounds: dimension mismatch");
      cppoptlib::mi355::Check(
          mi355_auglag_box_minimize_batch_host(ctx_->get(), &p, &c, &inner_stop, solver_t::kHistorySize,
                                               solver_t::kLineSearch, lo.empty() ? nullptr : lo.data(),
                                               lo.empty() ? nullptr : up.data(), B,
                                               constants.empty() ? nullptr : constants.data(), x.data(),
                                               n_eq ? lambda.data() : nullptr, n_ineq ? mu.data() : nullptr,
                                               penalty.data(), violation.data(), kkt.data(), prog.data()),
          "mi355_auglag_box_minimize_batch_host");
    } else {
      cppoptlib::mi355::Check(
          mi355_auglag_minimize_batch_host(ctx_->get(), &p, &c, &inner_stop, solver_t::kHistorySize,
                                           solver_t::kLineSearch, B, constants.empty() ? nullptr : constants.data(),
                                           x.data(), n_eq ? lambda.data() : nullptr, n_ineq ? mu.data() : nullptr,
                                           penalty.data(), violation.data(), kkt.data(), prog.data()),
          "mi355_auglag_minimize_batch_host");
    }
    result.reserve(b);
    for (size_t i = 0; i < b; ++i) {
      StateType s = states[i];
      // (a float problem is widened at this boundary and its results rounded back, like the unconstrained solvers)
      for (int j = 0; j < n; ++j) s.x[j] = static_cast<ScalarType>(x[i * n + j]);
      for (int j = 0; j < n_eq; ++j)
        s.multiplier_state.equality_multipliers[j] = static_cast<ScalarType>(lambda[i * n_eq + j]);
      for (int j = 0; j < n_ineq; ++j)
        s.multiplier_state.inequality_multipliers[j] = static_cast<ScalarType>(mu[i * n_ineq + j]);
      s.penalty_state.penalty = static_cast<ScalarType>(penalty[i]);
      s.max_violation = static_cast<ScalarType>(violation[i]);
      s.max_lagrangian_gradient = static_cast<ScalarType>(kkt[i]);
      s.penalty_was_auto_scaled = states[i].penalty_was_auto_scaled ||
                                  (config_.auto_scale_initial_penalty && states[i].penalty_state.penalty == 0);
      ProgressType pr;
      pr.num_iterations = prog[i].num_iterations;
      pr.x_delta = static_cast<ScalarType>(prog[i].x_delta);
      pr.f_delta = static_cast<ScalarType>(prog[i].f_delta);
      pr.gradient_norm = static_cast<ScalarType>(prog[i].gradient_norm);
      pr.status = static_cast<Status>(prog[i].status);
      pr.num_function_evaluations = static_cast<size_t>(prog[i].nfev);
      pr.history_pairs_used = static_cast<size_t>(prog[i].sum_k);
      result.emplace_back(std::move(s), pr);
    }
    return result;
  }

 private:
  ProblemType problem_;
  solver_t unconstrained_solver_template_;
  AugmentedLagrangianConfig<ScalarType> config_;
  std::shared_ptr<cppoptlib::mi355::Context> ctx_;
};

template <typename ProblemType, typename solver_t>
AugmentedLagrangian(const ProblemType&, const solver_t&) -> AugmentedLagrangian<ProblemType, solver_t>;
template <typename ProblemType, typename solver_t, typename TScalar>
AugmentedLagrangian(const ProblemType&, const solver_t&, AugmentedLagrangianConfig<TScalar>)
    -> AugmentedLagrangian<ProblemType, solver_t>;

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_AUGMENTED_LAGRANGIAN_H_
