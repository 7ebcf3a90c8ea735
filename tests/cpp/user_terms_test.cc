// The reference's non-convex augmented-Lagrangian tests (src/test/augmented_lagrangian_test.cc:945-1060 HS024,
// :1064-1150 HS029) on the device, written as the reference writes them: USER classes as the objective / constraint of a
// ConstrainedOptimizationProblem.  A user class becomes a device term by naming the id of its device functor
// (kAlTermKind >= MI355_AL_TERM_USER; examples/user_al_terms/hs_terms.hpp, compiled into libmi355_lbfgs_hs.so by
// __graft_entry__.build()) next to its host operator().  Linked against that build of the library.
#include <cmath>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;
using cppoptlib::solver::AugmentedLagrangeState;
using Problem = ConstrainedOptimizationProblem<>;
using Vec = Problem::VectorType;

static Vec MakeVec(std::initializer_list<double> v) {
  Vec x(static_cast<int>(v.size()));
  int i = 0;
  for (double e : v) x[i++] = e;
  return x;
}

template <class Derived>
using UserFunction = FunctionCRTP<Derived, double, DifferentiabilityMode::First, kDynamicDimension>;
static std::vector<double> NoCoefficients(int n) { return std::vector<double>(static_cast<size_t>(n) + 1, 0.0); }

class Hs024Objective : public UserFunction<Hs024Objective> {
 public:
  static constexpr int kAlTermKind = MI355_AL_TERM_USER + 0;  // user_examples::Hs024Objective
  std::vector<double> AlCoefficients(int n) const { return NoCoefficients(n); }
  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    const double bracket = (x[0] - 3.0) * (x[0] - 3.0) - 9.0, scale = 1.0 / (27.0 * std::sqrt(3.0));
    if (grad) {
      grad->resize(2);
      (*grad)[0] = 2.0 * (x[0] - 3.0) * x[1] * x[1] * x[1] * scale;
      (*grad)[1] = 3.0 * bracket * x[1] * x[1] * scale;
    }
    return bracket * x[1] * x[1] * x[1] * scale;
  }
};
class ProductObjective : public UserFunction<ProductObjective> {
 public:
  static constexpr int kAlTermKind = MI355_AL_TERM_USER + 1;  // user_examples::ProductObjective
  std::vector<double> AlCoefficients(int n) const { return NoCoefficients(n); }
  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    if (grad) {
      grad->resize(2);
      (*grad)[0] = -x[1];
      (*grad)[1] = -x[0];
    }
    return -x[0] * x[1];
  }
};
class Hs029Ellipse : public UserFunction<Hs029Ellipse> {
 public:
  static constexpr int kAlTermKind = MI355_AL_TERM_USER + 2;  // user_examples::Hs029Ellipse
  std::vector<double> AlCoefficients(int n) const { return NoCoefficients(n); }
  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    if (grad) {
      grad->resize(2);
      (*grad)[0] = -2.0 * x[0];
      (*grad)[1] = -4.0 * x[1];
    }
    return 48.0 - x[0] * x[0] - 2.0 * x[1] * x[1];
  }
};

int main() {
  {
    // AugmentedLagrangianNonConvex.Hs024TriangleEscapesSpuriousOrigin (:1018-1060)
    const double sqrt3 = std::sqrt(3.0);
    Hs024Objective objective;
    const LinearForm<> g0(std::vector<double>{1.0 / sqrt3, -1.0}), edge(std::vector<double>{1.0, sqrt3});
    Problem problem(objective, /*eq=*/{}, {g0, edge, 6.0 - edge});
    using BoxInner = cppoptlib::solver::Lbfgsb<AugmentedLagrangianFunction<>>;
    BoxInner inner_solver;
    inner_solver.SetBounds(MakeVec({0.0, 0.0}), MakeVec({1e20, 1e20}));
    cppoptlib::solver::AugmentedLagrangian<Problem, BoxInner> solver(problem, inner_solver);
    AugmentedLagrangeState<double> state(MakeVec({1.0, 0.5}), /*num_eq=*/0, /*num_ineq=*/3, /*penalty=*/0.0);
    auto [solution, progress] = solver.Minimize(state);
    const double f_final = objective(solution.x);
    EXPECT_NEAR(3.0, solution.x[0], 1e-1);
    EXPECT_NEAR(sqrt3, solution.x[1], 1e-1);
    EXPECT_NEAR(-1.0, f_final, 0.5);
    // ... and much closer than the reference asks: the golden output of the reference itself is (3.00023, 1.73167)
    EXPECT_NEAR(3.00023246, solution.x[0], 1e-6);
    EXPECT_NEAR(1.73167016, solution.x[1], 1e-6);
    EXPECT_TRUE(progress.status == cppoptlib::solver::Status::Finished);
  }
  {
    // AugmentedLagrangianNonConvex.Hs029EllipseEscapesOrigin (:1115-1150)
    ProductObjective objective;
    Hs029Ellipse ellipse;
    Problem problem(objective, /*eq=*/{}, {ellipse});
    using Inner = cppoptlib::solver::Lbfgs<AugmentedLagrangianFunction<>>;
    Inner inner_solver;
    cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, inner_solver);
    AugmentedLagrangeState<double> state(MakeVec({1.0, 1.0}), /*num_eq=*/0, /*num_ineq=*/1, /*penalty=*/0.0);
    auto [solution, progress] = solver.Minimize(state);
    EXPECT_NEAR(2.0 * std::sqrt(6.0), solution.x[0], 2e-1);
    EXPECT_NEAR(2.0 * std::sqrt(3.0), solution.x[1], 2e-1);
    EXPECT_NEAR(-12.0 * std::sqrt(2.0), objective(solution.x), 5e-1);
    EXPECT_NEAR(4.89896762, solution.x[0], 1e-6);   // the reference's own output from this start
    EXPECT_NEAR(3.46410949, solution.x[1], 1e-6);
    EXPECT_TRUE(solution.max_violation <= 1e-5);
  }
  TEST_MAIN_END();
}
