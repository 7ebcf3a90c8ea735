// Host-side expression nodes of the augmented-Lagrangian layer (reference function_penalty.h:40-61, 97-222;
// function_expressions.h ConstExpression :46-87, SubExpression :148-196, MinZeroExpression :319-358, MaxZeroExpression
// :362-399): closed forms, the clip at exactly zero, and the sum of the parts against the composite ToAugmentedLagrangian
// returns (the function the engine solves).  Nothing here touches the device: the binary runs on a GPU-less box too.
#include <cmath>

#include "cppoptlib/function.h"
#include "cppoptlib/function_penalty.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;
using Expr = FunctionExpr<double, DifferentiabilityMode::First>;
using Vec = LinearForm<>::VectorType;

static Vec MakeVec(std::initializer_list<double> v) {
  Vec x(static_cast<int>(v.size()));
  int i = 0;
  for (double e : v) x[i++] = e;
  return x;
}

int main() {
  const LinearForm<> c(std::vector<double>{2.0, -1.0});   // c(x) = 2 x0 - x1, gradient (2, -1)
  const Expr ce = c;
  Vec g(2);
  {  // P_eq = 0.5 c^2, gradient c grad c — on either side of zero
    const Vec x = MakeVec({1.0, 5.0});                    // c = -3
    EXPECT_EQ(QuadraticEqualityPenalty(ce)(x, &g), 4.5);
    EXPECT_EQ(g[0], -6.0);
    EXPECT_EQ(g[1], 3.0);
    EXPECT_EQ(QuadraticEqualityPenalty(ce)(MakeVec({3.0, 3.0}), &g), 4.5);   // c = +3: symmetric
    EXPECT_EQ(g[0], 6.0);
  }
  {  // P_ge = 0.5 min{0, c}^2 fires on c < 0 only; P_lt = 0.5 max{0, c}^2 on c > 0 only; both are 0 with zero gradient AT c = 0
    const Vec neg = MakeVec({1.0, 5.0}), pos = MakeVec({3.0, 3.0}), zero = MakeVec({1.0, 2.0});
    EXPECT_EQ(QuadraticInequalityPenaltyGe(ce)(neg, &g), 4.5);
    EXPECT_EQ(g[0], -6.0);
    EXPECT_EQ(QuadraticInequalityPenaltyGe(ce)(pos, &g), 0.0);
    EXPECT_TRUE(g[0] == 0.0 && g[1] == 0.0);
    EXPECT_EQ(QuadraticInequalityPenaltyLt(ce)(pos, &g), 4.5);
    EXPECT_EQ(g[1], -3.0);
    EXPECT_EQ(QuadraticInequalityPenaltyLt(ce)(neg, &g), 0.0);
    EXPECT_TRUE(g[0] == 0.0 && g[1] == 0.0);
    EXPECT_EQ(MinZeroExpression<Expr>(ce)(zero, &g), 0.0);    // f == 0 counts as clipped in both nodes
    EXPECT_TRUE(g[0] == 0.0 && g[1] == 0.0);
    EXPECT_EQ(MaxZeroExpression<Expr>(ce)(zero, &g), 0.0);
    EXPECT_TRUE(g[0] == 0.0 && g[1] == 0.0);
    EXPECT_EQ(MinZeroExpression<Expr>(ce)(neg, &g), -3.0);
    EXPECT_EQ(g[0], 2.0);
  }
  {  // the constant, and f - g
    const ConstExpression<double, DifferentiabilityMode::First> seven(7.0);
    EXPECT_EQ(seven(MakeVec({4.0, 4.0, 4.0}), &g), 7.0);
    EXPECT_TRUE(g.size() == 3 && g[0] == 0.0 && g[2] == 0.0);
    const auto d = SquaredNorm<>() - c;                  // |x|^2 - (2 x0 - x1)
    EXPECT_EQ(d(MakeVec({1.0, 5.0}), &g), 29.0);
    EXPECT_EQ(g[0], 0.0);                                // 2 x0 - 2
    EXPECT_EQ(g[1], 11.0);                               // 2 x1 + 1
  }
  {  // f + Lagrangian part + penalty part + inequality part == the composite of ToAugmentedLagrangian, value and gradient
    const Expr objective = SquaredNorm<>();
    const ConstrainedOptimizationProblem prob(objective, {ce}, {ce - 1.0});
    const LagrangeMultiplierState<double> mult({0.5}, {0.25});
    const PenaltyState<double> pen(2.0);
    const Vec x = MakeVec({1.0, 5.0});
    Vec gp(2), gc(2);
    const Expr parts = prob.objective + FormLagrangianPart(prob, mult) + FormPenaltyPart(prob, pen) + FormInequalityPart(prob, mult, pen);
    const double vp = parts(x, &gp);
    const double vc = ToAugmentedLagrangian(prob, mult, pen)(x, &gc);
    EXPECT_EQ(vp, 50.5);      // 26 - 1.5 + 9 + (8.25^2 / 4 - 0.0625 / 4)
    EXPECT_NEAR(vc, vp, 1e-13);
    EXPECT_NEAR(gp[0], -25.5, 1e-13);                    // 2 + 1 - 12 - 16.5
    EXPECT_NEAR(gp[1], 23.75, 1e-13);                    // 10 - 0.5 + 6 + 8.25
    EXPECT_NEAR(gc[0], gp[0], 1e-13);
    EXPECT_NEAR(gc[1], gp[1], 1e-13);
    // no penalty, no inequality part (1 / (2 rho) undefined): the reference returns the zero constant
    EXPECT_EQ(FormInequalityPart(prob, mult, PenaltyState<double>(0.0))(x, &g), 0.0);
    // ToPenalty: f + rho 0.5 c^2 + rho 0.5 min{0, g}^2 = 26 + 9 + 16
    EXPECT_EQ(ToPenalty(prob, pen)(x, &g), 51.0);
  }
  {  // names the reference's headers also export: the mode helpers, the node aliases, IsFunctionState, and the projected
     // gradient norm of a box (lbfgsb.h:105-118: unbounded without SetBounds)
    static_assert(MinDifferentiabilityMode<DifferentiabilityMode::First, DifferentiabilityMode::Second>::value ==
                  DifferentiabilityMode::First);
    static_assert(MinDifferentiability<SquaredNorm<>, LinearForm<>>::value == DifferentiabilityMode::First);
    static_assert(cppoptlib::solver::IsFunctionState<FunctionState<double>>::value);
    static_assert(!cppoptlib::solver::IsFunctionState<cppoptlib::solver::AugmentedLagrangeState<double>>::value);
    const AddExpression<SquaredNorm<>, LinearForm<>> sum(SquaredNorm<>(), c);
    EXPECT_EQ(sum(MakeVec({1.0, 5.0}), &g), 23.0);
    cppoptlib::solver::Lbfgsb<Rosenbrock<>> box_solver;
    const Vec x = MakeVec({0.0, 1.0, 0.5}), grad = MakeVec({5.0, -7.0, -2.0});
    EXPECT_EQ(box_solver.ProjectedGradientInfNorm(x, grad), 7.0);
    box_solver.SetBounds(MakeVec({0.0, -1.0, 0.0}), MakeVec({1.0, 1.0, 1.0}));
    EXPECT_EQ(box_solver.ProjectedGradientInfNorm(x, grad), 2.0);   // x0 on its lower bound with g > 0, x1 on its upper with g < 0
  }
  TEST_MAIN_END();
}
