// Mirrors the constrained test of the reference's src/test/verify.cc:290-312 (same problem as
// src/examples/constrained_simple2.cc) and three KKT / composite cases of src/test/augmented_lagrangian_test.cc
// (:397-414, :492-516, :541-575) on the device AugmentedLagrangian; plus ToAugmentedLagrangian handed straight to
// Lbfgs, and the batched entry point.
#include "cppoptlib/function.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;
using cppoptlib::solver::AugmentedLagrangeState;
using Problem = ConstrainedOptimizationProblem<>;
using Inner = cppoptlib::solver::Lbfgs<AugmentedLagrangianFunction<>>;
using Vec = Problem::VectorType;

static Vec MakeVec(std::initializer_list<double> v) {
  Vec x(static_cast<int>(v.size()));
  int i = 0;
  for (double e : v) x[i++] = e;
  return x;
}

int main() {
  {
    // verify.cc:290-312: min x0 + x1  s.t.  |x|^2 - 2 = 0,  2 - |x|^2 >= 0;  expects (-1, -1) within 1e-3
    LinearForm<> objective(std::vector<double>{1.0, 1.0});  // x.sum()
    SquaredNorm<> circle;
    ConstrainedOptimizationProblem prob(objective,
                                        /* equality constraints */ {circle - 2.0},
                                        /* inequality constraints */ {2.0 - circle});
    Inner inner_solver;
    cppoptlib::solver::AugmentedLagrangian solver(prob, inner_solver);
    AugmentedLagrangeState<double> l_state(MakeVec({2.0, 10.0}), 1, 1, 1.0);
    auto [solution, solver_state] = solver.Minimize(l_state);
    EXPECT_NEAR(solution.x[0], -1, 1e-3);
    EXPECT_NEAR(solution.x[1], -1, 1e-3);
    EXPECT_TRUE(solver_state.status == cppoptlib::solver::Status::Finished);
    EXPECT_TRUE(solution.max_violation <= 1e-5);
  }
  const DiagQuadratic<> half_squared_norm(std::vector<double>{0.5, 0.5}, 0.0);  // HalfSquaredNorm2D
  const LinearForm<> x0(std::vector<double>{1.0, 0.0});
  {
    // ToAugmentedLagrangian.EqualityOnlyMatchesClosedForm: 22.5 at (3, 4) with lambda = 2, rho = 3
    Problem problem(half_squared_norm, {x0 - 1.0});
    auto augmented = ToAugmentedLagrangian(problem, LagrangeMultiplierState<double>({2.0}, {}), PenaltyState<double>(3.0));
    EXPECT_NEAR(22.5, augmented(MakeVec({3.0, 4.0})), 1e-12);
    // the composite is an objective of its own: Lbfgs minimises it on the device
    // (closed form: x0 = (rho - lambda) / (1 + rho) = 0.25, x1 = 0)
    Inner lbfgs;
    auto [sol, st] = lbfgs.Minimize(augmented, FunctionState(MakeVec({3.0, 4.0})));
    EXPECT_NEAR(0.25, sol.x[0], 1e-5);
    EXPECT_NEAR(0.0, sol.x[1], 1e-5);
    Vec g(2);
    EXPECT_EQ(augmented(sol.x, &g), sol.value);
    EXPECT_EQ(g[0], sol.gradient[0]);
  }
  {
    // AugmentedLagrangianKKT.EqualityOnlyQuadratic: x* = (1, 0), lambda* = -1
    Problem problem(half_squared_norm, {x0 - 1.0});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, Inner());
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(MakeVec({5.0, 5.0}), 1, 0, 1.0));
    EXPECT_NEAR(1.0, solution.x[0], 1e-3);
    EXPECT_NEAR(0.0, solution.x[1], 1e-3);
    EXPECT_TRUE(std::fabs(solution.x[0] - 1.0) <= 1e-5);
    EXPECT_NEAR(-1.0, solution.multiplier_state.equality_multipliers[0], 1e-2);
  }
  {
    // after AugmentedLagrangianKKT.InequalityActiveRecoversMultiplier: x0 >= 1 active, mu* = 1
    Problem problem(half_squared_norm, {}, {x0 - 1.0});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, Inner());
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(MakeVec({5.0, 5.0}), 0, 1, 1.0));
    EXPECT_NEAR(1.0, solution.x[0], 1e-3);
    EXPECT_TRUE(solution.x[0] - 1.0 >= -1e-5);
    EXPECT_NEAR(1.0, solution.multiplier_state.inequality_multipliers[0], 1e-2);
  }
  {
    // batched: 64 starts of a 12-dimensional problem (sum x = 1, x0 <= 0.2), auto-scaled penalty
    const int n = 12;
    std::vector<double> a(n), ones(n, 1.0), e0(n, 0.0);
    for (int i = 0; i < n; ++i) a[i] = 0.5 + 0.25 * i;
    e0[0] = 1.0;
    Problem problem(DiagQuadratic<>(a, 0.5), {LinearForm<>(ones) - 1.0}, {0.2 - LinearForm<>(e0)});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, Inner());
    // (with the default inner stopping test some starts never reach the 1e-4 stationarity threshold — in the
    //  reference as well — and would run to the 10000-iteration limit: bound the outer loop, take the best iterate)
    solver.stopping_progress.num_iterations = 60;
    std::vector<AugmentedLagrangeState<double>> starts;
    for (int b = 0; b < 64; ++b) {
      Vec x(n);
      for (int i = 0; i < n; ++i) x[i] = std::sin(0.37 * (b + 1) * (i + 1));
      starts.emplace_back(x, 1, 1);
    }
    auto out = solver.MinimizeBatch(starts);
    EXPECT_EQ(out.size(), size_t(64));
    for (auto& [sol, st] : out) {
      EXPECT_TRUE(st.status == cppoptlib::solver::Status::Finished ||
                  st.status == cppoptlib::solver::Status::IterationLimit);
      EXPECT_TRUE(st.num_iterations <= 61);
      double sum = 0;
      for (int i = 0; i < n; ++i) sum += sol.x[i];
      EXPECT_NEAR(1.0, sum, 1e-4);
      EXPECT_TRUE(sol.x[0] <= 0.2 + 1e-4);
      EXPECT_TRUE(sol.max_violation <= 1e-4);
      EXPECT_TRUE(sol.penalty_state.penalty > 0 && sol.penalty_was_auto_scaled);
    }
  }
  {
    // per-state term constants: state b solves  min 0.5 |x|^2  s.t.  x0 = t_b   ->  x* = (t_b, 0), lambda* = -t_b
    Problem problem(half_squared_norm, {x0 - 1.0});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, Inner());
    std::vector<AugmentedLagrangeState<double>> starts;
    std::vector<std::vector<double>> constants;
    for (int b = 0; b < 8; ++b) {
      starts.emplace_back(MakeVec({5.0, 5.0}), 1, 0, 1.0);
      constants.push_back({0.0, 0.5 + 0.25 * b});
    }
    auto out = solver.MinimizeBatch(problem, starts, constants);
    for (int b = 0; b < 8; ++b) {
      auto& [sol, st] = out[b];
      EXPECT_NEAR(0.5 + 0.25 * b, sol.x[0], 1e-3);
      EXPECT_NEAR(0.0, sol.x[1], 1e-3);
      EXPECT_NEAR(-(0.5 + 0.25 * b), sol.multiplier_state.equality_multipliers[0], 1e-2);
    }
  }
  {
    // Lbfgsb as the inner solver (after AugmentedLagrangianBoxInterface.BoxPinnedOptimumStopsOnKkt, :1198-1275):
    // min 0.5 |x|^2  s.t.  x0 + x1 = 2,  box x0 <= 0.5   ->   x* = (0.5, 1.5); the box handles x0, the outer loop
    // the equality, and stationarity is measured with the projected gradient
    using BoxInner = cppoptlib::solver::Lbfgsb<AugmentedLagrangianFunction<>>;
    Problem problem(half_squared_norm, {LinearForm<>(std::vector<double>{1.0, 1.0}) - 2.0});
    BoxInner inner;
    inner.SetBounds(MakeVec({-10.0, -10.0}), MakeVec({0.5, 10.0}));
    cppoptlib::solver::AugmentedLagrangian<Problem, BoxInner> solver(problem, inner);
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(MakeVec({-2.0, 1.0}), 1, 0, 0.0));
    EXPECT_TRUE(progress.status == cppoptlib::solver::Status::Finished);
    EXPECT_TRUE(progress.num_iterations < 20);
    EXPECT_NEAR(0.5, solution.x[0], 1e-6);
    EXPECT_NEAR(1.5, solution.x[1], 1e-3);
    EXPECT_NEAR(-1.5, solution.multiplier_state.equality_multipliers[0], 1e-2);
  }
  {
    // AugmentedLagrangianBoxInterface.BoxPinnedOptimumStopsOnKkt (:1198-1275, HS016): 2-D Rosenbrock,
    // x0^2 + x1 >= 0, x0 + x1^2 >= 0 (each a sum of two menu primitives), box [-0.5, 0.5] x [-1e20, 1]
    using BoxInner = cppoptlib::solver::Lbfgsb<AugmentedLagrangianFunction<>>;
    const LinearForm<> e0(std::vector<double>{1.0, 0.0}), e1(std::vector<double>{0.0, 1.0});
    const DiagQuadratic<> sq0(std::vector<double>{1.0, 0.0}, 0.0), sq1(std::vector<double>{0.0, 1.0}, 0.0);
    Problem problem(Rosenbrock<>(), {}, {sq0 + e1, e0 + sq1});
    BoxInner inner;
    inner.SetBounds(MakeVec({-0.5, -1e20}), MakeVec({0.5, 1.0}));
    cppoptlib::solver::AugmentedLagrangian<Problem, BoxInner> solver(problem, inner);
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(MakeVec({-2.0, 1.0}), 0, 2, 0.0));
    EXPECT_TRUE(progress.status == cppoptlib::solver::Status::Finished);
    EXPECT_TRUE(progress.num_iterations < 20);
    EXPECT_NEAR(solution.x[0], 0.5, 1e-4);
    EXPECT_NEAR(solution.x[1], 0.25, 1e-4);
  }
  {
    // AugmentedLagrangianKKT.BothEqualityAndInequalityActive (:583-621): QuadraticAt12 as a sum of two primitives
    Problem problem(DiagQuadratic<>(std::vector<double>{1.0, 1.0}, 5.0) + LinearForm<>(std::vector<double>{-2.0, -4.0}),
                    {x0 - 0.5}, {2.0 - LinearForm<>(std::vector<double>{1.0, 1.0})});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, Inner());
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(MakeVec({1.0, 1.0}), 1, 1, 1.0));
    EXPECT_NEAR(0.5, solution.x[0], 1e-3);
    EXPECT_NEAR(1.5, solution.x[1], 1e-3);
    EXPECT_TRUE(std::fabs(solution.x[0] - 0.5) <= 1e-5);
    EXPECT_TRUE(2.0 - (solution.x[0] + solution.x[1]) >= -1e-5);
    EXPECT_TRUE(solution.multiplier_state.inequality_multipliers[0] >= -1e-2);
  }
  {
    // Hs029EllipseEscapesOrigin (:1064-1150) written over the menu with the reference's operator* of two functions
    // (ProdExpression, function_expressions.h:260-315, :453-461): objective (-x0) * x1, constraint
    // 48 - (x0^2 + 2 x1^2) >= 0; start (1, 1); optimum (2 sqrt 6, 2 sqrt 3), f* = -12 sqrt 2
    const LinearForm<> minus_x0(std::vector<double>{-1.0, 0.0}), x1(std::vector<double>{0.0, 1.0});
    const auto objective = minus_x0 * x1;
    Vec g(2);
    EXPECT_EQ(objective(MakeVec({3.0, 4.0}), &g), -12.0);       // host evaluation: the product rule
    EXPECT_EQ(g[0], -4.0);
    EXPECT_EQ(g[1], -3.0);
    Problem problem(objective, {}, {48.0 - DiagQuadratic<>(std::vector<double>{1.0, 2.0}, 0.0)});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, Inner());
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(MakeVec({1.0, 1.0}), 0, 1, 0.0));
    EXPECT_TRUE(progress.status == cppoptlib::solver::Status::Finished);
    EXPECT_NEAR(solution.x[0], 2.0 * std::sqrt(6.0), 1e-3);
    EXPECT_NEAR(solution.x[1], 2.0 * std::sqrt(3.0), 1e-3);
    EXPECT_NEAR(objective(solution.x), -12.0 * std::sqrt(2.0), 1e-3);
    // a product as a constraint term, with a constant: (x0 + x1) * x1 - 2 = 0 on min |x|^2
    Problem product_constraint(SquaredNorm<>(), {LinearForm<>(std::vector<double>{1.0, 1.0}) * x1 - 2.0});
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver2(product_constraint, Inner());
    auto [s2, p2] = solver2.Minimize(AugmentedLagrangeState<double>(MakeVec({1.0, 1.0}), 1, 0, 1.0));
    EXPECT_TRUE(p2.status == cppoptlib::solver::Status::Finished);
    EXPECT_NEAR((s2.x[0] + s2.x[1]) * s2.x[1], 2.0, 1e-4);
  }
  {
    // Constraint vectors of any length (function_problem.h:57-84 of the reference): more constraints of a kind than the
    // device's term table holds travel as a constraint FAMILY — here a box written as 2 n affine inequalities plus the
    // simplex equality, on min |x - t|^2:   x_i >= 0.1,  -x_i >= -0.6,  sum x = 2   (n = 6: twelve inequalities).
    // The projection of t onto that set is known in closed form (clip a shifted t; the shift found by bisection).
    const int n = 6;
    const std::vector<double> t{1.4, -0.3, 0.55, 0.2, 0.9, 0.05};
    std::vector<double> ones(n, 1.0), minus_two_t(n);
    for (int i = 0; i < n; ++i) minus_two_t[i] = -2.0 * t[i];
    double tt = 0;
    for (double v : t) tt += v * v;
    const auto objective = DiagQuadratic<>(ones, tt) + LinearForm<>(minus_two_t);   // |x|^2 - 2 t.x + |t|^2
    std::vector<Problem::ConstraintFunctionType> inequalities;
    for (int i = 0; i < n; ++i) {
      std::vector<double> e(n, 0.0);
      e[i] = 1.0;
      inequalities.emplace_back(LinearForm<>(e) - 0.1);             // x_i - 0.1 >= 0
    }
    for (int i = 0; i < n; ++i) {
      std::vector<double> e(n, 0.0);
      e[i] = -1.0;
      inequalities.emplace_back(LinearForm<>(e) - (-0.6));          // -x_i + 0.6 >= 0
    }
    Problem problem(objective, {LinearForm<>(ones) - 2.0}, inequalities);
    Inner inner;
    inner.stopping_progress.gradient_norm = 1e-9;
    inner.stopping_progress.x_delta = 1e-12;
    inner.stopping_progress.past = 0;
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, inner);
    solver.stopping_progress.constraint_threshold = 1e-8;
    solver.stopping_progress.kkt_stationarity_threshold = 1e-6;
    Vec start(n);
    for (int i = 0; i < n; ++i) start[i] = 0.0;
    auto [solution, progress] = solver.Minimize(AugmentedLagrangeState<double>(start, 1, 2 * n, 0.0));
    double lo = -5, hi = 5;   // sum_i clip(t_i - s, 0.1, 0.6) = 2
    for (int it = 0; it < 200; ++it) {
      const double s = 0.5 * (lo + hi);
      double sum = 0;
      for (double v : t) sum += std::fmin(0.6, std::fmax(0.1, v - s));
      (sum > 2.0 ? lo : hi) = s;
    }
    EXPECT_TRUE(progress.status == cppoptlib::solver::Status::Finished);
    EXPECT_EQ(solution.multiplier_state.inequality_multipliers.size(), size_t(2 * n));
    for (int i = 0; i < n; ++i) EXPECT_NEAR(solution.x[i], std::fmin(0.6, std::fmax(0.1, t[i] - 0.5 * (lo + hi))), 1e-5);
    // multipliers sit on the active bounds only (coordinate 0 is at its upper bound, coordinate 1 at its lower)
    EXPECT_TRUE(solution.multiplier_state.inequality_multipliers[n + 0] > 1e-3);
    EXPECT_TRUE(solution.multiplier_state.inequality_multipliers[1] > 1e-3);
    EXPECT_TRUE(solution.multiplier_state.inequality_multipliers[0] <= 1e-6);
    // ... but only AFFINE constraints travel as a family: five quadratic inequalities are refused
    std::vector<Problem::ConstraintFunctionType> quadratic(5, Problem::ConstraintFunctionType(4.0 - SquaredNorm<>()));
    Problem too_many(objective, {}, quadratic);
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> refusing(too_many, inner);
    bool threw = false;
    try {
      refusing.Minimize(AugmentedLagrangeState<double>(start, 0, 5, 0.0));
    } catch (const std::exception&) {
      threw = true;
    }
    EXPECT_TRUE(threw);
  }
  TEST_MAIN_END();
}
