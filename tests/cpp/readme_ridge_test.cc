// The reference's README ridge example (README.md:122-167) against the MI355X engine, written the
// way the README writes it: two functions composed with + and a scalar *, wrapped in a
// FunctionExpr whose decltype parameterises Lbfgs.  The functors are Second-mode, so Lbfgs takes
// the diagonal-preconditioner branch (lbfgs.h:116-139).  Expected numbers are the reference's own
// output for this program (its unmodified headers built over oracle/eigen_shim): 9 iterations to
//   x* = (-4.11960228757013, 5.01359151630839),  f* = 5.73059498641439.
#include <cmath>

#include "cppoptlib/function.h"
#include "cppoptlib/function_expressions.h"
#include "cppoptlib/solver/lbfgs.h"
#include "mini_test.h"

using namespace cppoptlib::function;

int main() {
  const std::vector<double> A = {1, 2, 3, 4, 5, 6};  // 3 x 2, row major
  const std::vector<double> y = {7, 8, 9};
  const double lambda = 0.1;
  FunctionExpr objective = SquaredError<>(3, 2, A, y) + lambda * L2Reg<>(2);
  using Objective = decltype(objective);
  static_assert(Objective::Differentiability == DifferentiabilityMode::Second, "README functors are Second-mode");

  Objective::VectorType x(2);
  x[0] = 0;
  x[1] = 0;
  cppoptlib::solver::Lbfgs<Objective> solver;
  auto [solution, progress] = solver.Minimize(objective, FunctionState(x));
  std::printf("x* = (%.14f, %.14f) f* = %.14f iterations = %zu status = %d\n", solution.x[0], solution.x[1],
              solution.value, progress.num_iterations, static_cast<int>(progress.status));
  EXPECT_NEAR(solution.x[0], -4.11960228757013, 1e-6);
  EXPECT_NEAR(solution.x[1], 5.01359151630839, 1e-6);
  EXPECT_NEAR(solution.value, 5.73059498641439, 1e-6);
  EXPECT_EQ(progress.num_iterations, size_t(9));

  // host evaluation of the expression == the returned state, and == term-by-term composition
  Objective::VectorType g, gs, gl;
  Objective::MatrixType h;
  EXPECT_NEAR(objective(solution.x, &g, &h), solution.value, 1e-12);
  EXPECT_NEAR(g[0], solution.gradient[0], 1e-9);
  EXPECT_NEAR(g[1], solution.gradient[1], 1e-9);
  SquaredError<> se(3, 2, A, y);
  L2Reg<> l2(2);
  const double v = se(solution.x, &gs) + lambda * l2(solution.x, &gl);
  EXPECT_EQ(v, objective(solution.x));
  EXPECT_EQ(g[0], gs[0] + lambda * gl[0]);
  EXPECT_EQ(h(0, 0), 2 * 35.0 + lambda * 2);
  EXPECT_EQ(h(1, 0), 2 * 44.0);

  // condition_hessian (progress.h:203-210, :318-325): Second-mode functions report ||H|| ||H^-1|| in the Progress
  // and may stop on it; H = [[70.2, 88], [88, 112.2]]
  {
    const double a = 2 * 35.0 + lambda * 2, b = 2 * 44.0, c = 2 * 56.0 + lambda * 2, det = a * c - b * b;
    const double frob = std::sqrt(a * a + 2 * b * b + c * c);
    EXPECT_NEAR(progress.condition_hessian, frob * frob / std::fabs(det), 1e-9 * frob * frob / std::fabs(det));
    cppoptlib::solver::Lbfgs<Objective> strict;
    strict.stopping_progress.condition_hessian = 10.0;   // the matrix above is worse conditioned than that
    auto [sc, pc] = strict.Minimize(objective, FunctionState(x));
    EXPECT_TRUE(pc.status == cppoptlib::solver::Status::HessianConditionViolation);
    EXPECT_EQ(pc.num_iterations, size_t(1));
    cppoptlib::solver::Lbfgs<Objective> lax;
    lax.stopping_progress.condition_hessian = 1e12;
    auto [sl2, pl2] = lax.Minimize(objective, FunctionState(x));
    EXPECT_EQ(pl2.num_iterations, size_t(9));
    EXPECT_TRUE(pl2.status != cppoptlib::solver::Status::HessianConditionViolation);
  }

  // First-mode declaration of the same functors: the plain two-loop path, same minimiser
  FunctionExpr first = SquaredError<kDynamicDimension, DifferentiabilityMode::First>(3, 2, A, y) +
                       lambda * L2Reg<kDynamicDimension, DifferentiabilityMode::First>(2);
  using First = decltype(first);
  static_assert(First::Differentiability == DifferentiabilityMode::First, "weaker mode wins");
  cppoptlib::solver::Lbfgs<First> solver1;
  auto [s1, p1] = solver1.Minimize(first, FunctionState(x));
  EXPECT_NEAR(s1.x[0], solution.x[0], 1e-4);
  EXPECT_NEAR(s1.x[1], solution.x[1], 1e-4);
  EXPECT_TRUE(s1.x[0] != solution.x[0]);  // a different iteration, not a relabelling

  // each operand has a twin of its own
  cppoptlib::solver::Lbfgs<L2Reg<>> solver2;
  Objective::VectorType x2(2);
  x2[0] = 3;
  x2[1] = -4;
  auto [s2, p2] = solver2.Minimize(l2, FunctionState(x2));
  EXPECT_TRUE(std::fabs(s2.x[0]) < 1e-6 && std::fabs(s2.x[1]) < 1e-6);
  TEST_MAIN_END();
}
