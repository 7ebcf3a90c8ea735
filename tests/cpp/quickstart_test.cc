// The reference's README quick start (README.md:14-36 / Dockerfile.test:13-49) against the
// MI355X engine: the user class keeps its host operator() and gains two lines that name
// its device twin.  Asserts what Dockerfile.test:39-42 asserts.
#include "cppoptlib/function.h"
#include "cppoptlib/solver/lbfgs.h"
#include "mini_test.h"

// f(x) = 5*x0^2 + 100*x1^2 + 5
class Quadratic : public cppoptlib::function::FunctionCRTP<
                      Quadratic, double, cppoptlib::function::DifferentiabilityMode::First, 2> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* grad) const {
    if (grad) *grad = VectorType(10 * x[0], 200 * x[1]);
    return 5 * x[0] * x[0] + 100 * x[1] * x[1] + 5;
  }
  // --- the two lines a user adds: the device twin of this functor ---
  static constexpr int kDeviceObjective = MI355_OBJ_DIAG_QUADRATIC;
  std::vector<double> DeviceParams() const { return {5.0, 100.0, 5.0}; }
};

int main() {
  Quadratic f;
  Quadratic::VectorType x0(-10, 2);
  cppoptlib::solver::Lbfgs<Quadratic> solver;
  auto [solution, state] = solver.Minimize(f, cppoptlib::function::FunctionState(x0));
  std::printf("x* = (%.3e, %.3e) f* = %.12f iterations = %zu status = %d\n", solution.x[0], solution.x[1],
              solution.value, state.num_iterations, static_cast<int>(state.status));
  EXPECT_TRUE(std::fabs(solution.x[0]) < 1e-4);
  EXPECT_TRUE(std::fabs(solution.x[1]) < 1e-4);
  EXPECT_TRUE(std::fabs(solution.value - 5.0) < 1e-4);
  // the returned state carries value and gradient AT x* (function_base.h:297-332 invariant)
  Quadratic::VectorType g(2);
  EXPECT_EQ(f(solution.x, &g), solution.value);
  EXPECT_EQ(g[0], solution.gradient[0]);
  EXPECT_EQ(g[1], solution.gradient[1]);
  EXPECT_EQ(state.num_iterations, size_t(10));
  EXPECT_EQ(state.num_function_evaluations, size_t(11));
  EXPECT_TRUE(state.status == cppoptlib::solver::Status::GradientNormViolation);
  TEST_MAIN_END();
}
