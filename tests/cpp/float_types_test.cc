// float function types (the reference ships FunctionXf, include/cppoptlib/function.h:38, and solves a float problem in
// src/examples/linear_regression.cc).  The MI355X engine computes in fp64; a float function type is widened at the
// boundary and its results are rounded back -- so a float user gets the fp64 answer rounded to float.
#include <cmath>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/bfgs.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;

// src/examples/linear_regression.cc:14-39 -- residuals (b1 + 2 b2 - 4, 3 b1 + b2 - 5) -- with its device twin named: the
// ridge objective with A = [[1, 2], [3, 1]], y = (4, 5), lambda = 0
class LinearRegression : public FunctionXf<LinearRegression> {
 public:
  static constexpr int kDeviceObjective = MI355_OBJ_SQUARED_ERROR_RIDGE;
  std::vector<double> DeviceParams() const { return {2.0, 0.0, 1.0, 2.0, 3.0, 1.0}; }   // rows, lambda, A (row major)
  std::vector<double> DevicePerProblem() const { return {4.0, 5.0}; }
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const ScalarType r1 = x[0] + 2 * x[1] - 4, r2 = 3 * x[0] + x[1] - 5;
    if (gradient) {
      gradient->resize(x.size());
      (*gradient)[0] = 2 * (r1 + 3 * r2);
      (*gradient)[1] = 2 * (2 * r1 + r2);
    }
    return r1 * r1 + r2 * r2;
  }
};

// the 2-D Rosenbrock function of src/test/verify.cc:58-69 as a float type
class RosenbrockF : public FunctionXf<RosenbrockF> {
 public:
  static constexpr int kDeviceObjective = MI355_OBJ_ROSENBROCK;
  std::vector<double> DeviceParams() const { return {}; }
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const ScalarType t1 = 1 - x[0], t2 = x[1] - x[0] * x[0];
    if (gradient) {
      gradient->resize(2);
      (*gradient)[0] = -2 * t1 + 200 * t2 * (-2 * x[0]);
      (*gradient)[1] = 200 * t2;
    }
    return t1 * t1 + 100 * t2 * t2;
  }
};

int main() {
  static_assert(std::is_same<LinearRegression::ScalarType, float>::value, "a float function type");
  {
    // linear_regression.cc:58-74: "optimal solution is suppose to be [1, 1.6]" under the box [0, 1] x [1, 2]
    cppoptlib::solver::Lbfgsb<LinearRegression> solver;
    LinearRegression f;
    LinearRegression::VectorType x(2), lb(2), ub(2);
    x[0] = -1; x[1] = 2;
    lb[0] = 0; lb[1] = 1;
    ub[0] = 1; ub[1] = 2;
    solver.SetBounds(lb, ub);
    auto [solution, solver_state] = solver.Minimize(f, FunctionState(x));
    static_assert(std::is_same<decltype(solution.value), float>::value, "results come back in the function's scalar type");
    EXPECT_NEAR(1.0, solution.x[0], 1e-5);
    EXPECT_NEAR(1.6, solution.x[1], 1e-5);
    EXPECT_NEAR(f(solution.x), solution.value, 1e-5);
    EXPECT_TRUE(solver_state.status != cppoptlib::solver::Status::IterationLimit);
  }
  {
    // unconstrained: the normal equations give (1.2, 1.4), residual 0
    cppoptlib::solver::Lbfgs<LinearRegression> solver;
    LinearRegression f;
    LinearRegression::VectorType x(2);
    x[0] = -1; x[1] = 2;
    auto [solution, st] = solver.Minimize(f, FunctionState(x));
    EXPECT_NEAR(1.2, solution.x[0], 1e-4);
    EXPECT_NEAR(1.4, solution.x[1], 1e-4);
    EXPECT_NEAR(0.0, solution.value, 1e-6);
  }
  {
    // Lbfgs / Bfgs on a float Rosenbrock from the reference test's starts (verify.cc:117-129), and a batch
    RosenbrockF f;
    RosenbrockF::VectorType x(2);
    x[0] = -1; x[1] = 2;
    cppoptlib::solver::Lbfgs<RosenbrockF> lbfgs;
    auto [s1, p1] = lbfgs.Minimize(f, FunctionState(x));
    EXPECT_NEAR(0.0, f(s1.x), 1e-4);
    cppoptlib::solver::Bfgs<RosenbrockF> bfgs;
    auto [s2, p2] = bfgs.Minimize(f, FunctionState(x));
    EXPECT_NEAR(0.0, f(s2.x), 1e-4);
    std::vector<FunctionState<float, RosenbrockF::Dimension>> starts;
    for (int b = 0; b < 40; ++b) {
      RosenbrockF::VectorType s(2);
      s[0] = -1.5f + 0.07f * b;
      s[1] = 2.0f - 0.05f * b;
      starts.emplace_back(s);
    }
    const auto out = lbfgs.MinimizeBatch(f, starts);
    EXPECT_EQ(out.size(), size_t(40));
    for (const auto& [sol, st] : out) {
      EXPECT_NEAR(1.0, sol.x[0], 1e-3);
      EXPECT_NEAR(1.0, sol.x[1], 1e-3);
    }
  }
  TEST_MAIN_END();
}
