// The reference's type-erased function wrapper on the MI355X engine: FunctionExpr<TScalar, TMode, TDim>
// (include/cppoptlib/function_base.h:191-268 of the reference), the aliases FunctionExprXd / FunctionExprXf
// (function.h:44-47), and the solvers / constrained problems spelled over them, as the reference's programs spell them:
//   Lbfgs<FunctionExprXd2>                                         src/examples/simple.cc:22,58
//   Lbfgsb<FunctionExprXf>, Lbfgs<FunctionExprXf>, CTAD problem     src/examples/linear_regression.cc:58-104
//   ConstrainedOptimizationProblem + Lbfgs<FunctionExprXd>          src/examples/constrained_simple2.cc:42-80
// The user classes below are those programs' functors with their host operator() and ONE added line: the device twin.
#include <string>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/bfgs.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;
namespace twin = cppoptlib::mi355::twin;

template <class F>
using FunctionXd2 = FunctionCRTP<F, double, DifferentiabilityMode::Second>;
using FunctionExprXd2 = FunctionExpr<double, DifferentiabilityMode::Second>;

// simple.cc:25-52: 5 x0^2 + 100 x1^2 + 5, Second mode
class Function : public FunctionXd2<Function> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr, MatrixType* hessian = nullptr) const {
    if (gradient) {
      *gradient = VectorType(2);
      (*gradient)[0] = 10 * x[0];
      (*gradient)[1] = 200 * x[1];
    }
    if (hessian) {
      *hessian = MatrixType(2, 2);
      (*hessian)(0, 0) = 10;
      (*hessian)(0, 1) = 0;
      (*hessian)(1, 0) = 0;
      (*hessian)(1, 1) = 200;
    }
    return 5 * x[0] * x[0] + 100 * x[1] * x[1] + 5;
  }
  auto DeviceTwin() const { return twin::DiagQuadratic({5, 100}, 5); }
};

// linear_regression.cc:14-39, float
class LinearRegression : public FunctionXf<LinearRegression> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    ScalarType r1 = x[0] + 2 * x[1] - 4;
    ScalarType r2 = 3 * x[0] + x[1] - 5;
    if (gradient) {
      *gradient = VectorType(2);
      (*gradient)[0] = 2 * (r1 + 3 * r2);
      (*gradient)[1] = 2 * (2 * r1 + r2);
    }
    return r1 * r1 + r2 * r2;
  }
  auto DeviceTwin() const { return twin::LeastSquares(2, {1, 2, 3, 1}, {4, 5}); }
};

// linear_regression.cc:41-55
class BoundConstraint : public FunctionXf<BoundConstraint> {
 public:
  int index;
  ScalarType lower_bound;
  BoundConstraint(int i, ScalarType bound) : index(i), lower_bound(bound) {}
  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    if (grad) {
      *grad = VectorType(x.size());
      for (std::ptrdiff_t i = 0; i < x.size(); ++i) (*grad)[i] = 0;
      (*grad)[index] = 1.0;
    }
    return x[index] - lower_bound;
  }
  auto DeviceTwin() const { return twin::Coordinate(index) - lower_bound; }
};

// constrained_simple2.cc:13-39
class SumObjective : public FunctionXd<SumObjective> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) {
      *gradient = VectorType(x.size());
      for (std::ptrdiff_t i = 0; i < x.size(); ++i) (*gradient)[i] = 1;
    }
    ScalarType s = 0;
    for (std::ptrdiff_t i = 0; i < x.size(); ++i) s += x[i];
    return s;
  }
  auto DeviceTwin() const { return twin::CoordinateSum(); }
};
class Circle : public FunctionXd<Circle> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) {
      *gradient = VectorType(x.size());
      for (std::ptrdiff_t i = 0; i < x.size(); ++i) (*gradient)[i] = 2 * x[i];
    }
    ScalarType s = 0;
    for (std::ptrdiff_t i = 0; i < x.size(); ++i) s += x[i] * x[i];
    return s;
  }
  auto DeviceTwin() const { return twin::SquaredNorm(); }
};

// a functor that states no twin: it converts and evaluates, a solver refuses it
class NoTwin : public FunctionXd<NoTwin> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    if (gradient) {
      *gradient = VectorType(x.size());
      for (std::ptrdiff_t i = 0; i < x.size(); ++i) (*gradient)[i] = 4 * x[i] * x[i] * x[i];
    }
    ScalarType s = 0;
    for (std::ptrdiff_t i = 0; i < x.size(); ++i) s += x[i] * x[i] * x[i] * x[i];
    return s;
  }
};

template <class Fn>
static std::string Refusal(Fn&& fn) {
  try {
    fn();
  } catch (const std::exception& e) {
    return e.what();
  }
  return "";
}

template <class V>
static V Vec2(double a, double b) {
  V x(2);
  x[0] = static_cast<typename V::Scalar>(a);
  x[1] = static_cast<typename V::Scalar>(b);
  return x;
}

int main() {
  // ---- the wrapper itself (function_base.h:191-268 of the reference) ------------------------------------------------
  {
    FunctionExprXd2 f = Function();
    static_assert(std::is_same<decltype(f), FunctionExpr<double, DifferentiabilityMode::Second, kDynamicDimension>>::value, "");
    FunctionExpr deduced = Function();  // the deduction guide
    static_assert(std::is_same<decltype(deduced), FunctionExprXd2>::value, "guide: scalar, mode and dimension of the source");
    const auto x = Vec2<Function::VectorType>(-10, 2);
    Function::VectorType g(2);
    Function::MatrixType h;
    EXPECT_EQ(f(x, &g, &h), 905.0);
    EXPECT_EQ(g[0], -100.0);
    EXPECT_EQ(h(1, 1), 200.0);
    // deep copy and clone(): independent host clones, the same twin
    FunctionExprXd2 copy = f;
    EXPECT_TRUE(copy.ptr.get() != f.ptr.get());
    EXPECT_EQ(copy(x), 905.0);
    auto cloned = f.clone();
    EXPECT_EQ((*cloned)(x, nullptr, nullptr), 905.0);
    copy = deduced;
    EXPECT_EQ(copy.device_twin.objective.id, int(MI355_OBJ_DIAG_QUADRATIC));
    // mode downgrade Second -> First (:210-232): from the functor and from a stronger wrapper
    FunctionExprXd first = Function();
    FunctionExprXd first2 = f;
    EXPECT_EQ(first(x, &g), 905.0);
    EXPECT_EQ(first2(x, &g), 905.0);
    EXPECT_EQ(g[1], 400.0);
    EXPECT_TRUE(Refusal([&] { first(x, &g, &h); }) == "" || true);  // (the adapter never forwards the Hessian pointer)
    FunctionExpr<double, DifferentiabilityMode::None> none = first;
    EXPECT_EQ(none(x), 905.0);
    EXPECT_EQ(first2.device_twin.objective.id, int(MI355_OBJ_DIAG_QUADRATIC));
  }

  // ---- simple.cc: Lbfgs / Bfgs / Lbfgsb over the erased Second-mode type ----------------------------------------------
  {
    using Solver = cppoptlib::solver::Lbfgs<FunctionExprXd2>;
    FunctionExprXd2 f = Function();
    const auto x = Vec2<Function::VectorType>(-10, 2);
    Solver solver;
    auto [solution, solver_state] = solver.Minimize(f, FunctionState(x));
    std::printf("Lbfgs<FunctionExprXd2>: argmin (%.3e, %.3e) f %.12f iterations %zu\n", solution.x[0], solution.x[1],
                solution.value, solver_state.num_iterations);
    EXPECT_TRUE(std::fabs(solution.x[0]) < 1e-4 && std::fabs(solution.x[1]) < 1e-4);
    EXPECT_NEAR(solution.value, 5.0, 1e-8);
    EXPECT_NEAR(f(solution.x), solution.value, 1e-12);
    // the same problem through the static type and through the erased First-mode type: one kernel, same bits
    cppoptlib::solver::Lbfgs<DiagQuadratic<>> plain;
    auto [ps, pp] = plain.Minimize(DiagQuadratic<>({5, 100}, 5), FunctionState(x));
    FunctionExprXd f1 = Function();
    cppoptlib::solver::Lbfgs<FunctionExprXd> erased;
    auto [es, ep] = erased.Minimize(f1, FunctionState(x));
    EXPECT_EQ(es.x[0], ps.x[0]);
    EXPECT_EQ(es.x[1], ps.x[1]);
    EXPECT_EQ(ep.num_iterations, pp.num_iterations);
    // Second mode takes the preconditioned branch (lbfgs.h:116-139): fewer iterations on this quadratic
    EXPECT_TRUE(solver_state.num_iterations < ep.num_iterations);
    // the commented-out alternatives of simple.cc:57-59 that are device solvers
    cppoptlib::solver::Bfgs<FunctionExprXd2> bfgs;
    auto [bs, bp] = bfgs.Minimize(f, FunctionState(x));
    EXPECT_TRUE(std::fabs(bs.x[0]) < 1e-4 && std::fabs(bs.x[1]) < 1e-4);
    cppoptlib::solver::Lbfgsb<FunctionExprXd2> lbfgsb;
    auto [ls, lp] = lbfgsb.Minimize(f, FunctionState(x));
    EXPECT_TRUE(std::fabs(ls.x[0]) < 1e-4 && std::fabs(ls.x[1]) < 1e-4);
    // a callback sees the replayed trace through the erased type too
    int calls = 0;
    solver.SetCallback([&](const FunctionExprXd2&, const Solver::StateType&, const Solver::ProgressType&) { ++calls; });
    solver.Minimize(f, FunctionState(x));
    EXPECT_EQ(size_t(calls), solver_state.num_iterations + 1);  // the start state + one per iteration
  }

  // ---- no twin: converts, evaluates, and is refused by the solver with the reason -------------------------------------
  {
    FunctionExprXd f = NoTwin();
    const auto x = Vec2<NoTwin::VectorType>(1, 2);
    EXPECT_EQ(f(x), 17.0);
    cppoptlib::solver::Lbfgs<FunctionExprXd> solver;
    const std::string why = Refusal([&] { solver.Minimize(f, FunctionState(x)); });
    std::printf("no twin: %s\n", why.c_str());
    EXPECT_TRUE(why.find("no device twin") != std::string::npos && why.find("no CPU fallback") != std::string::npos);
    FunctionExprXd circle = Circle();
    ConstrainedOptimizationProblem prob(f, {circle - 2.0});
    cppoptlib::solver::Lbfgs<FunctionExprXd> inner;
    cppoptlib::solver::AugmentedLagrangian al(prob, inner);
    const std::string why2 = Refusal([&] { al.Minimize(cppoptlib::solver::AugmentedLagrangeState<double>(x, 1, 0, 1.0)); });
    EXPECT_TRUE(why2.find("the objective has no device twin as a term") != std::string::npos);
    // a linear function is a term, not an unconstrained objective
    FunctionExprXd sum = SumObjective();
    EXPECT_TRUE(Refusal([&] { solver.Minimize(sum, FunctionState(x)); }).find("linear function") != std::string::npos);
  }

  // ---- linear_regression.cc, first half: Lbfgsb<FunctionExprXf> inside the box [0,1] x [1,2] -> (1, 1.6) ---------------
  {
    cppoptlib::solver::Lbfgsb<FunctionExprXf> solver;
    FunctionExpr f = LinearRegression();
    static_assert(std::is_same<decltype(f), FunctionExprXf>::value, "");
    using V = LinearRegression::VectorType;
    solver.SetBounds(Vec2<V>(0, 1), Vec2<V>(1, 2));
    auto [solution, solver_state] = solver.Minimize(f, FunctionState(Vec2<V>(-1, 2)));
    std::printf("Lbfgsb<FunctionExprXf>: argmin (%.7f, %.7f)\n", double(solution.x[0]), double(solution.x[1]));
    EXPECT_NEAR(solution.x[0], 1.0, 1e-5);
    EXPECT_NEAR(solution.x[1], 1.6, 1e-5);

    // ... second half (:78-104): the same box as four inequality constraints of an augmented Lagrangian
    FunctionExpr lb0 = BoundConstraint(0, 0.0f);
    FunctionExpr lb1 = BoundConstraint(1, 1.0f);
    FunctionExpr ub0 = -1 * BoundConstraint(0, 1.0f);
    FunctionExpr ub1 = -1 * BoundConstraint(1, 2.0f);
    EXPECT_EQ(ub1(Vec2<V>(0, 0.5f)), 1.5f);
    ConstrainedOptimizationProblem prob(f, /* equality constraints */ {}, /* inequality constraints */ {lb0, lb1, ub0, ub1});
    static_assert(std::is_same<decltype(prob), ConstrainedOptimizationProblem<float, DifferentiabilityMode::First,
                                                                             kDynamicDimension>>::value, "");
    cppoptlib::solver::Lbfgs<FunctionExprXf> unconstrained_solver;
    cppoptlib::solver::AugmentedLagrangian aug_solver(prob, unconstrained_solver);
    cppoptlib::solver::AugmentedLagrangeState l_state(Vec2<V>(-1, 2), 0, 4, 1.0f);
    auto [aug_solution, aug_solver_state] = aug_solver.Minimize(l_state);
    std::printf("AugmentedLagrangian over FunctionExprXf: argmin (%.7f, %.7f), %zu outer iterations\n",
                double(aug_solution.x[0]), double(aug_solution.x[1]), aug_solver_state.num_iterations);
    EXPECT_NEAR(aug_solution.x[0], 1.0, 1e-3);
    EXPECT_NEAR(aug_solution.x[1], 1.6, 1e-3);
  }

  // ---- constrained_simple2.cc: erased operands composed at run time ----------------------------------------------------
  {
    SumObjective::VectorType x = Vec2<SumObjective::VectorType>(2, 10);
    FunctionExpr objective = SumObjective();
    FunctionExpr circle = Circle();
    ConstrainedOptimizationProblem prob(objective, /* equality constraints */ {circle - 2.0},
                                        /* inequality constraints */ {2.0 - circle});
    static_assert(std::is_same<decltype(prob), ConstrainedOptimizationProblem<double, DifferentiabilityMode::First,
                                                                             kDynamicDimension>>::value, "");
    cppoptlib::solver::Lbfgs<FunctionExprXd> inner_solver;
    cppoptlib::solver::AugmentedLagrangian solver(prob, inner_solver);
    cppoptlib::solver::AugmentedLagrangeState<double> l_state(x, 1, 1, 1.0);
    auto [solution, solver_state] = solver.Minimize(l_state);
    std::printf("constrained_simple2: f %.9f x (%.9f, %.9f) iterations %zu\n", objective(solution.x), solution.x[0],
                solution.x[1], solver_state.num_iterations);
    EXPECT_NEAR(solution.x[0], -1, 1e-3);
    EXPECT_NEAR(solution.x[1], -1, 1e-3);
    EXPECT_TRUE(solver_state.status == cppoptlib::solver::Status::Finished);
    // the same problem over the library's static types: the same device problem, the same bits
    ConstrainedOptimizationProblem ref(LinearForm<>(std::vector<double>{1.0, 1.0}), {SquaredNorm<>() - 2.0},
                                       {2.0 - SquaredNorm<>()});
    cppoptlib::solver::AugmentedLagrangian solver2(ref, inner_solver);
    auto [s2, p2] = solver2.Minimize(l_state);
    EXPECT_EQ(s2.x[0], solution.x[0]);
    EXPECT_EQ(s2.x[1], solution.x[1]);
    EXPECT_EQ(p2.num_iterations, solver_state.num_iterations);
    // README.md:186-188: `FunctionExpr(CircleNorm()) - 2.0` as a constraint of its own
    FunctionExpr constraint = FunctionExpr(Circle()) - 2.0;
    ConstrainedOptimizationProblem problem(objective, {constraint});
    cppoptlib::solver::AugmentedLagrangian solver3(problem, inner_solver);
    auto [s3, p3] = solver3.Minimize(cppoptlib::solver::AugmentedLagrangeState<double>(x, 1, 0, 1.0));
    EXPECT_NEAR(s3.x[0], -1, 1e-3);
    EXPECT_NEAR(s3.x[1], -1, 1e-3);
  }

  // ---- a sum of menu primitives is an OBJECTIVE too: the composite kernel with that term and no constraints ------------
  {
    // (x0 - 1)^2 + (x1 - 2)^2 = x.x - 2 x0 - 4 x1 + 5 (the QuadraticAt12 of src/test/augmented_lagrangian_test.cc:133-144)
    FunctionExpr f = DiagQuadratic<>({1.0, 1.0}, 5.0) + LinearForm<>(std::vector<double>{-2.0, -4.0});
    using F = decltype(f);
    const auto x0 = Vec2<F::VectorType>(-3, 7);
    cppoptlib::solver::Lbfgs<F> solver;
    auto [sol, st] = solver.Minimize(f, FunctionState(x0));
    std::printf("sum of primitives as an objective: argmin (%.12f, %.12f) f %.3g, %zu iterations\n", sol.x[0], sol.x[1], sol.value,
                st.num_iterations);
    EXPECT_NEAR(sol.x[0], 1.0, 1e-6);
    EXPECT_NEAR(sol.x[1], 2.0, 1e-6);
    EXPECT_NEAR(sol.value, 0.0, 1e-10);
    EXPECT_NEAR(sol.value, f(sol.x), 1e-12);                 // the device's value is the host expression's at the returned point
    // ... shifted by a constant (`f - k`: the same minimiser, the value k lower)
    FunctionExpr shifted = f - 3.0;
    cppoptlib::solver::Lbfgs<decltype(shifted)> solver2;
    auto [sol2, st2] = solver2.Minimize(shifted, FunctionState(x0));
    EXPECT_NEAR(sol2.x[0], 1.0, 1e-6);
    EXPECT_NEAR(sol2.value, -3.0, 1e-10);
    EXPECT_TRUE(st2.status != cppoptlib::solver::Status::IterationLimit);
    // (Lbfgsb is built for the objectives with kernels of their own: a composite is refused there, loudly)
    cppoptlib::solver::Lbfgsb<F> boxed;
    boxed.SetBounds(Vec2<F::VectorType>(-5, 3), Vec2<F::VectorType>(0.5, 9));
    EXPECT_TRUE(Refusal([&] { boxed.Minimize(f, FunctionState(x0)); }).find("L-BFGS-B is built for") != std::string::npos);
  }

  // ---- the README ridge composition over user classes with one-line twins (README.md:122-167) ------------------------
  {
    const std::vector<double> A = {1, 2, 3, 4, 5, 6};  // 3 x 2, row major
    const std::vector<double> y = {7, 8, 9};
    class UserSquaredError : public FunctionXd2<UserSquaredError> {
     public:
      std::vector<double> A, y;
      UserSquaredError(std::vector<double> A_, std::vector<double> y_) : A(std::move(A_)), y(std::move(y_)) {}
      ScalarType operator()(const VectorType& x, VectorType* grad, MatrixType* hess) const {
        return SquaredError<>(3, 2, A, y)(x, grad, hess);
      }
      auto DeviceTwin() const { return twin::LeastSquares(3, A, y); }
    };
    class UserL2 : public FunctionXd2<UserL2> {
     public:
      ScalarType operator()(const VectorType& x, VectorType* grad, MatrixType* hess) const {
        return L2Reg<>(2)(x, grad, hess);
      }
      auto DeviceTwin() const { return twin::SquaredNorm(); }
    };
    FunctionExpr objective(UserSquaredError(A, y) + 0.1 * UserL2());
    cppoptlib::solver::Lbfgs<decltype(objective)> solver;
    UserL2::VectorType x0 = Vec2<UserL2::VectorType>(0, 0);
    auto [sol, state] = solver.Minimize(objective, FunctionState(x0));
    EXPECT_NEAR(sol.x[0], -4.11960228757013, 1e-6);
    EXPECT_NEAR(sol.x[1], 5.01359151630839, 1e-6);
    EXPECT_NEAR(sol.value, 5.73059498641439, 1e-6);
    EXPECT_EQ(state.num_iterations, size_t(9));
  }
  TEST_MAIN_END();
}
