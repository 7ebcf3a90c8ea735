// The L-BFGS-B part of the reference's src/test/verify.cc (:190 LbfgsbTest RosenbrockGradientFar /
// Near: default-constructed solver, unbounded box, EXPECT_NEAR(0, f(x*), 1e-4)), a box-constrained
// case in the style of src/test/augmented_lagrangian_test.cc:1198-1280 (SetBounds, active bound at
// the solution), and the README ridge composition (README.md:122-167) through Lbfgs.
#include <cmath>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

constexpr double PRECISION = 1e-4;
using Function = cppoptlib::function::Rosenbrock<>;

static void SolveProblem(double a, double b) {
  using Solver = cppoptlib::solver::Lbfgsb<Function>;
  Function f;
  Function::VectorType x(2);
  x[0] = a;
  x[1] = b;
  Solver solver;
  auto [solution, solver_state] = solver.Minimize(f, cppoptlib::function::FunctionState(x));
  EXPECT_TRUE(solver_state.status != cppoptlib::solver::Status::IterationLimit);
  EXPECT_NEAR(0.0, f(solution.x), PRECISION);
}

int main() {
  SolveProblem(15.0, 8.0);  // LbfgsbTest.RosenbrockGradientFar
  SolveProblem(-1.0, 2.0);  // LbfgsbTest.RosenbrockGradientNear
  {
    // Rosenbrock-8 in the box [-1.5, 0.8]^8: the unconstrained minimiser (1,...,1) is infeasible,
    // so at least one upper bound is active at the solution and the solution stays in the box.
    using Solver = cppoptlib::solver::Lbfgsb<Function>;
    Function f;
    const int n = 8;
    Function::VectorType x(n), lo(n), hi(n);
    for (int i = 0; i < n; ++i) {
      x[i] = (i % 2) ? 1.9 : -1.9;  // infeasible start: clipped by the solver (lbfgsb.h:148)
      lo[i] = -1.5;
      hi[i] = 0.8;
    }
    Solver solver;
    solver.SetBounds(lo, hi);
    auto [sol, st] = solver.Minimize(f, cppoptlib::function::FunctionState(x));
    bool inside = true, active = false;
    for (int i = 0; i < n; ++i) {
      inside = inside && sol.x[i] >= -1.5 && sol.x[i] <= 0.8;
      active = active || sol.x[i] == 0.8;
    }
    EXPECT_TRUE(inside);
    EXPECT_TRUE(active);
    EXPECT_TRUE(st.status != cppoptlib::solver::Status::IterationLimit);
    // (the default build of Lbfgsb is the relaxed-algebra kernel, whose objective fuses its multiply-adds: the host
    //  functor agrees to rounding; under MI355_ARITH_EXACT the two are the same bits)
    EXPECT_NEAR(f(sol.x), sol.value, 1e-12 * (1.0 + std::fabs(sol.value)));
    {
      Solver exact;
      exact.SetBounds(lo, hi);
      exact.SetArithmetic(MI355_ARITH_EXACT);
      auto [se, pe] = exact.Minimize(f, cppoptlib::function::FunctionState(x));
      EXPECT_EQ(f(se.x), se.value);
      for (int i = 0; i < n; ++i) EXPECT_NEAR(se.x[i], sol.x[i], 1e-6);
    }
    // batched: same problem 16 times == the single solve
    std::vector<Solver::StateType> starts(16, Solver::StateType(x));
    auto batch = solver.MinimizeBatch(f, starts);
    for (int b : {0, 7, 15}) EXPECT_EQ(std::get<0>(batch[b]).value, sol.value);
  }
  {
    // The history size is a template argument of the reference (lbfgsb.h:44-49): Lbfgsb<F, 8> (sixteen rows of the
    // compact representation, one DPP row) and Lbfgsb<F, 10> (32 lanes per problem) reach the same boxed minimiser.
    Function f;
    const int n = 12;
    Function::VectorType x(n), lo(n), hi(n);
    for (int i = 0; i < n; ++i) {
      x[i] = (i % 2) ? 0.5 : -1.2;
      lo[i] = -1.5;
      hi[i] = 0.8;
    }
    cppoptlib::solver::Lbfgsb<Function> s5;
    cppoptlib::solver::Lbfgsb<Function, 8> s8;
    cppoptlib::solver::Lbfgsb<Function, 10> s10;
    s5.SetBounds(lo, hi);
    s8.SetBounds(lo, hi);
    s10.SetBounds(lo, hi);
    auto [a5, p5] = s5.Minimize(f, cppoptlib::function::FunctionState(x));
    auto [a8, p8] = s8.Minimize(f, cppoptlib::function::FunctionState(x));
    auto [a10, p10] = s10.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_TRUE(p8.status != cppoptlib::solver::Status::IterationLimit);
    EXPECT_TRUE(p10.status != cppoptlib::solver::Status::IterationLimit);
    EXPECT_NEAR(a5.value, a8.value, 1e-6);
    EXPECT_NEAR(a5.value, a10.value, 1e-6);
    for (int i = 0; i < n; ++i) {
      EXPECT_NEAR(a5.x[i], a8.x[i], 1e-5);
      EXPECT_NEAR(a5.x[i], a10.x[i], 1e-5);
      EXPECT_TRUE(a10.x[i] >= -1.5 && a10.x[i] <= 0.8);
    }
  }
  {
    // README ridge example data (README.md:154-157): A = [1 2; 3 4; 5 6], y = (7, 8, 9), lambda = 0.1
    using Ridge = cppoptlib::function::SquaredErrorRidge<>;
    Ridge objective(3, 2, {1, 2, 3, 4, 5, 6}, {7, 8, 9}, 0.1);
    Ridge::VectorType x0(2);
    x0[0] = 0;
    x0[1] = 0;
    auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<Ridge, cppoptlib::solver::Lbfgs<Ridge>::StateType>();
    stop.past = 0;
    stop.gradient_norm = 1e-9;
    cppoptlib::solver::Lbfgs<Ridge> solver(stop);
    auto [sol, st] = solver.Minimize(objective, cppoptlib::function::FunctionState(x0));
    // closed form (A^T A + 0.1 I)^-1 A^T y
    const double a11 = 35.1, a12 = 44, a22 = 56.1, b1 = 76, b2 = 100;
    const double det = a11 * a22 - a12 * a12;
    EXPECT_NEAR(sol.x[0], (a22 * b1 - a12 * b2) / det, 1e-6);
    EXPECT_NEAR(sol.x[1], (a11 * b2 - a12 * b1) / det, 1e-6);
    EXPECT_NEAR(objective(sol.x), sol.value, 1e-9);
  }
  {
    // The same example declared Second-mode, as the README prints it: Lbfgs takes the diagonal
    // preconditioner branch (lbfgs.h:116-139).  Expected values: the reference's own result for
    // this program (unmodified headers over oracle/eigen_shim): x* = (-4.11960228757013,
    // 5.01359151630839), f* = 5.73059498641439, 9 iterations, status FDeltaViolation.
    using Ridge2 = cppoptlib::function::SquaredErrorRidge<cppoptlib::function::kDynamicDimension,
                                                          cppoptlib::function::DifferentiabilityMode::Second>;
    Ridge2 objective(3, 2, {1, 2, 3, 4, 5, 6}, {7, 8, 9}, 0.1);
    Ridge2::VectorType x0(2);
    x0[0] = 0;
    x0[1] = 0;
    cppoptlib::solver::Lbfgs<Ridge2> solver;
    auto [sol, st] = solver.Minimize(objective, cppoptlib::function::FunctionState(x0));
    EXPECT_NEAR(sol.x[0], -4.11960228757013, 1e-6);
    EXPECT_NEAR(sol.x[1], 5.01359151630839, 1e-6);
    EXPECT_NEAR(sol.value, 5.73059498641439, 1e-6);
    EXPECT_EQ(static_cast<int>(st.num_iterations), 9);
    Ridge2::MatrixType h;
    Ridge2::VectorType gr;
    objective(sol.x, &gr, &h);
    EXPECT_NEAR(h(0, 0), 2 * 35 + 0.2, 1e-12);
    EXPECT_NEAR(h(0, 1), 2 * 44, 1e-12);
    EXPECT_NEAR(h(1, 1), 2 * 56 + 0.2, 1e-12);
  }
  TEST_MAIN_END();
}
