// The reference's Hager-Zhang unit tests (src/test/hager_zhang_test.cc:102-134) against the device
// line search, through the same `HagerZhang<F, 1>::Search` call they use: 1-D functions, unit
// direction, so alpha indexes phi(alpha) directly.  The reference's test functions are host-only
// functors; the ones here are the same parabolas written with the engine's diagonal quadratic
// (phi(a) = a^2 - 2a is x^2 - 1 seen from x = -1; 1e6 (a - 0.5)^2 is 1e6 x^2 seen from x = -0.5).
// Then `Lbfgs<F, m, HagerZhang>` — the drop-in use the header advertises (hager_zhang.h:39-42).
#include "cppoptlib/function.h"
#include "cppoptlib/linesearch/hager_zhang.h"
#include "cppoptlib/solver/lbfgs.h"
#include "mini_test.h"

using cppoptlib::function::DiagQuadratic;
using cppoptlib::function::Rosenbrock;
using cppoptlib::solver::linesearch::HagerZhang;

template <class F>
static std::pair<double, double> RunSearch(const F& function, double x0, double alpha_init) {
  typename F::VectorType x(1), s(1), x_out, g_out, g0(1);
  x[0] = x0;
  s[0] = 1.0;
  double f_out = 0.0;
  const double f0 = function(x, &g0);
  const double alpha = HagerZhang<F, 1>::Search(x, f0, g0, s, function, alpha_init, &x_out, &f_out, &g_out);
  return {alpha, f_out};
}

int main() {
  {  // case 1: convex quadratic, minimiser at alpha = 1 (hager_zhang_test.cc:102-107)
    DiagQuadratic<> phi({1.0}, -1.0);
    auto [alpha, f_at] = RunSearch(phi, -1.0, 1.0);
    EXPECT_NEAR(1.0, alpha, 1e-6);
    EXPECT_NEAR(-1.0, f_at, 1e-6);
  }
  {  // case 3: ill-scaled quadratic 1e6 (a - 0.5)^2 (:125-134)
    DiagQuadratic<> phi({1e6}, 0.0);
    auto [alpha, f_at] = RunSearch(phi, -0.5, 1.0);
    EXPECT_NEAR(0.5, alpha, 1e-6);
    EXPECT_NEAR(0.0, f_at, 1e-3);
    EXPECT_TRUE(alpha > 0.0 && alpha < 1.0);
  }
  {  // State overload + step-only overload on Rosenbrock-2 along steepest descent
    using F = Rosenbrock<>;
    F f;
    F::VectorType x(2), g(2), s(2);
    x[0] = -1.2;
    x[1] = 1.0;
    const double f0 = f(x, &g);
    s[0] = -g[0];
    s[1] = -g[1];
    const cppoptlib::function::FunctionState start(f, x);
    double alpha = 0;
    const auto next = HagerZhang<F, 1>::Search(start, s, f, 1e-3, &alpha);
    EXPECT_TRUE(alpha > 0.0);
    EXPECT_TRUE(next.value < f0);
    EXPECT_EQ(f(next.x), next.value);                       // the returned state is self-consistent
    const double alpha_only = HagerZhang<F, 1>::Search(x, s, f, 1e-3);
    EXPECT_EQ(alpha_only, alpha);
    EXPECT_NEAR(next.x[0], x[0] + alpha * s[0], 1e-15);
  }
  {  // Lbfgs<F, m, HagerZhang>
    using F = Rosenbrock<>;
    F f;
    F::VectorType x(2);
    x[0] = -1.2;
    x[1] = 1.0;
    cppoptlib::solver::Lbfgs<F, 10, HagerZhang> solver;
    auto [sol, st] = solver.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_NEAR(sol.x[0], 1.0, 1e-3);
    EXPECT_NEAR(sol.x[1], 1.0, 1e-3);
    EXPECT_TRUE(st.status != cppoptlib::solver::Status::IterationLimit);
    cppoptlib::solver::Lbfgs<F, 10> mt;   // the default line search takes a different path to the same point
    auto [sol2, st2] = mt.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_NEAR(sol.x[0], sol2.x[0], 1e-3);
    EXPECT_TRUE(st.num_iterations != st2.num_iterations || sol.value != sol2.value);
  }
  TEST_MAIN_END();
}
