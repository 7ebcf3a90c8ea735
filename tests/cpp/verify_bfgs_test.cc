// Mirrors the dense-BFGS part of the reference's src/test/verify.cc (the BfgsTest instantiation of the
// same typed cases the L-BFGS test uses: RosenbrockGradientFar (15, 8), RosenbrockGradientNear (-1, 2),
// default stopping, EXPECT_NEAR(0, f(x*), 1e-4)), on the device Bfgs<F, LineSearch>; plus the Hager-Zhang
// line search as its template argument and the batched entry point.
#include "cppoptlib/function.h"
#include "cppoptlib/linesearch/hager_zhang.h"
#include "cppoptlib/solver/bfgs.h"
#include "mini_test.h"

constexpr double PRECISION = 1e-4;
using Function = cppoptlib::function::Rosenbrock<>;

template <class Solver>
static void SolveProblem(double a, double b) {
  Function f;
  Function::VectorType x(2);
  x[0] = a;
  x[1] = b;
  Solver solver;
  auto [solution, solver_state] = solver.Minimize(f, cppoptlib::function::FunctionState(x));
  EXPECT_TRUE(solver_state.status != cppoptlib::solver::Status::IterationLimit);
  EXPECT_NEAR(0.0, f(solution.x), PRECISION);
  Function::VectorType g(2);
  EXPECT_EQ(f(solution.x, &g), solution.value);   // the returned state is self-consistent
  EXPECT_EQ(g[0], solution.gradient[0]);
}

int main() {
  using Bfgs = cppoptlib::solver::Bfgs<Function>;
  using BfgsHz = cppoptlib::solver::Bfgs<Function, cppoptlib::solver::linesearch::HagerZhang>;
  SolveProblem<Bfgs>(15.0, 8.0);     // BfgsTest.RosenbrockGradientFar
  SolveProblem<Bfgs>(-1.0, 2.0);     // BfgsTest.RosenbrockGradientNear
  SolveProblem<BfgsHz>(15.0, 8.0);
  SolveProblem<BfgsHz>(-1.0, 2.0);
  {
    // batched: 24 starts in 10 dimensions, every one stops at a (possibly local) stationary point below its start
    Function f;
    std::vector<Bfgs::StateType> starts;
    for (int b = 0; b < 24; ++b) {
      Function::VectorType x(10);
      for (int i = 0; i < 10; ++i) x[i] = ((i % 2) ? 1.0 : -1.2) + 0.01 * b;
      starts.emplace_back(x);
    }
    Bfgs solver;
    auto out = solver.MinimizeBatch(f, starts);
    EXPECT_EQ(out.size(), size_t(24));
    for (size_t b = 0; b < out.size(); ++b) {
      auto& [sol, st] = out[b];
      EXPECT_TRUE(st.status != cppoptlib::solver::Status::IterationLimit);
      EXPECT_TRUE(sol.value < f(starts[b].x));
      double gmax = 0;
      for (int i = 0; i < 10; ++i) gmax = std::max(gmax, std::fabs(sol.gradient[i]));
      EXPECT_TRUE(gmax < 1e-2);
    }
  }
  TEST_MAIN_END();
}
