// One function object PER problem through the batched drop-in classes — configs[3] as the reference's README writes
// it (README.md:126-167: `SquaredError(A, y) + lambda * L2Reg(n)`, one objective per right-hand side), B times:
//   * Lbfgs::MinimizeBatch(std::vector<FunctionType>, states) packs each function's per-problem row (its y) instead
//     of replicating one; every solution is checked against the closed form (A^T A + lambda I)^-1 A^T y_b;
//   * Lbfgsb on the same function type: one function (src/examples/linear_regression.cc runs Lbfgsb on a regression
//     objective), a vector of functions, and the sharded entry over a device group;
//   * a callback ASSIGNED to the public member step_callback_, as code written against the reference may do
//     (solver.h:230), is replayed like one handed to SetCallback.
#include <cmath>
#include <random>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/function_expressions.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;

// x = (A^T A + lambda I)^-1 A^T y by Gaussian elimination with partial pivoting (n small)
static std::vector<double> ClosedForm(int rows, int n, const std::vector<double>& A, const std::vector<double>& y, double lambda) {
  std::vector<double> M(size_t(n) * (n + 1), 0.0);
  for (int j = 0; j < n; ++j) {
    for (int k = 0; k < n; ++k) {
      double acc = (j == k) ? lambda : 0.0;
      for (int i = 0; i < rows; ++i) acc += A[size_t(i) * n + j] * A[size_t(i) * n + k];
      M[size_t(j) * (n + 1) + k] = acc;
    }
    double rhs = 0;
    for (int i = 0; i < rows; ++i) rhs += A[size_t(i) * n + j] * y[i];
    M[size_t(j) * (n + 1) + n] = rhs;
  }
  for (int c = 0; c < n; ++c) {
    int p = c;
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(M[size_t(r) * (n + 1) + c]) > std::fabs(M[size_t(p) * (n + 1) + c])) p = r;
    for (int k = 0; k <= n; ++k) std::swap(M[size_t(c) * (n + 1) + k], M[size_t(p) * (n + 1) + k]);
    for (int r = c + 1; r < n; ++r) {
      const double m = M[size_t(r) * (n + 1) + c] / M[size_t(c) * (n + 1) + c];
      for (int k = c; k <= n; ++k) M[size_t(r) * (n + 1) + k] -= m * M[size_t(c) * (n + 1) + k];
    }
  }
  std::vector<double> x(n);
  for (int c = n - 1; c >= 0; --c) {
    double acc = M[size_t(c) * (n + 1) + n];
    for (int k = c + 1; k < n; ++k) acc -= M[size_t(c) * (n + 1) + k] * x[k];
    x[c] = acc / M[size_t(c) * (n + 1) + c];
  }
  return x;
}

int main() {
  constexpr int rows = 24, n = 10, B = 300;
  const double lambda = 0.1;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> gauss(0.0, 1.0);
  std::vector<double> A(size_t(rows) * n);
  for (double& v : A) v = gauss(rng) / std::sqrt(double(rows));
  std::vector<std::vector<double>> Y(B, std::vector<double>(rows));
  for (auto& y : Y)
    for (double& v : y) v = gauss(rng);

  // the README's composition, First mode (the plain two-loop path), one objective per right-hand side
  using SE = SquaredError<kDynamicDimension, DifferentiabilityMode::First>;
  using L2 = L2Reg<kDynamicDimension, DifferentiabilityMode::First>;
  FunctionExpr proto = SE(rows, n, A, Y[0]) + lambda * L2(n);
  using Objective = decltype(proto);
  std::vector<Objective> objectives;
  using State = FunctionState<double, Objective::Dimension>;
  std::vector<State> starts;
  Objective::VectorType zero(n);
  for (int i = 0; i < n; ++i) zero[i] = 0;
  const State zero_state(zero);
  for (int b = 0; b < B; ++b) {
    objectives.push_back(SE(rows, n, A, Y[b]) + lambda * L2(n));
    starts.emplace_back(zero);
  }
  cppoptlib::solver::Lbfgs<Objective> solver;
  solver.stopping_progress.x_delta = 1e-11;
  solver.stopping_progress.gradient_norm = 1e-9;
  solver.stopping_progress.past = 0;
  const auto results = solver.MinimizeBatch(objectives, starts);
  EXPECT_EQ(results.size(), size_t(B));
  double worst = 0;
  for (int b = 0; b < B; ++b) {
    const std::vector<double> ref = ClosedForm(rows, n, A, Y[b], lambda);
    for (int i = 0; i < n; ++i) worst = std::fmax(worst, std::fabs(std::get<0>(results[b]).x[i] - ref[i]));
    // the returned value is the value of THAT problem's function at the returned point
    EXPECT_NEAR(std::get<0>(results[b]).value, objectives[b](std::get<0>(results[b]).x), 1e-10);
  }
  std::printf("Lbfgs, %d functions: max |x - closed form| = %.3g\n", B, worst);
  EXPECT_TRUE(worst <= 1e-6);
  // SetArithmetic(MI355_ARITH_FMA) takes the function type's fused twin — for the ridge functors the normal-equation
  // form (one Gram matrix for the batch, c_b = A^T y_b on the matrix cores): the same minimisers to 1e-6
  {
    cppoptlib::solver::Lbfgs<Objective> fused = solver;
    fused.SetArithmetic(MI355_ARITH_FMA);
    const auto fr = fused.MinimizeBatch(objectives, starts);
    double w2 = 0, diff = 0;
    for (int b = 0; b < B; ++b) {
      const std::vector<double> ref = ClosedForm(rows, n, A, Y[b], lambda);
      for (int i = 0; i < n; ++i) {
        w2 = std::fmax(w2, std::fabs(std::get<0>(fr[b]).x[i] - ref[i]));
        diff = std::fmax(diff, std::fabs(std::get<0>(fr[b]).x[i] - std::get<0>(results[b]).x[i]));
      }
    }
    std::printf("Lbfgs, fused twin (normal equations): max |x - closed form| = %.3g, vs the reference-order twin %.3g\n", w2, diff);
    EXPECT_TRUE(w2 <= 1e-6);
    EXPECT_TRUE(diff > 0.0);   // a different kernel, not a relabelling
  }
  // ... and it is not the replicated-row answer: problems 0 and 1 have different minimisers
  EXPECT_TRUE(std::fabs(std::get<0>(results[0]).x[0] - std::get<0>(results[1]).x[0]) > 1e-6);
  // MinimizeBatch(function, states) replicates the one row: every state of the batch gets problem 7's answer
  {
    const auto same = solver.MinimizeBatch(objectives[7], std::vector<State>(3, zero_state));
    for (int i = 0; i < n; ++i) EXPECT_EQ(std::get<0>(same[2]).x[i], std::get<0>(results[7]).x[i]);
  }
  // Functions that do NOT share their matrix — one `SquaredError(A_b, y_b) + lambda * L2Reg` per data set, as a reference
  // program builds them (README.md:126-160) — are solved with every problem's own matrix (objective id
  // MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM: normal equations per problem): each minimiser matches ITS closed form.
  {
    const int Bo = 40;
    std::vector<Objective> own;
    std::vector<std::vector<double>> As(Bo, A);
    std::vector<State> own_starts;
    for (int b = 0; b < Bo; ++b) {
      for (double& v : As[b]) v += 0.05 * gauss(rng);
      own.push_back(SE(rows, n, As[b], Y[b]) + lambda * L2(n));
      own_starts.emplace_back(zero);
    }
    const auto res = solver.MinimizeBatch(own, own_starts);
    EXPECT_EQ(res.size(), size_t(Bo));
    double w3 = 0;
    for (int b = 0; b < Bo; ++b) {
      const std::vector<double> ref = ClosedForm(rows, n, As[b], Y[b], lambda);
      for (int i = 0; i < n; ++i) w3 = std::fmax(w3, std::fabs(std::get<0>(res[b]).x[i] - ref[i]));
      EXPECT_NEAR(std::get<0>(res[b]).value, own[b](std::get<0>(res[b]).x), 1e-9);
      EXPECT_TRUE(std::get<1>(res[b]).status != cppoptlib::solver::Status::IterationLimit);
    }
    std::printf("Lbfgs, %d functions with their own matrices: max |x - closed form| = %.3g\n", Bo, w3);
    EXPECT_TRUE(w3 <= 1e-6);
    // a batch in which only ONE function differs is still recognised (fingerprint of every function)
    std::vector<Objective> mixed(objectives.begin(), objectives.begin() + 20);
    mixed[13] = own[13];
    const auto mr = solver.MinimizeBatch(mixed, std::vector<State>(20, zero_state));
    const std::vector<double> ref13 = ClosedForm(rows, n, As[13], Y[13], lambda), ref5 = ClosedForm(rows, n, A, Y[5], lambda);
    double w4 = 0;
    for (int i = 0; i < n; ++i) {
      w4 = std::fmax(w4, std::fabs(std::get<0>(mr[13]).x[i] - ref13[i]));
      w4 = std::fmax(w4, std::fabs(std::get<0>(mr[5]).x[i] - ref5[i]));
    }
    EXPECT_TRUE(w4 <= 1e-6);
    // Lbfgsb has no own-matrix form: the same mixed batch is refused there (not solved with the first function's matrix)
    bool threw = false;
    try {
      cppoptlib::solver::Lbfgsb<Objective> box_solver;
      box_solver.MinimizeBatch(mixed, std::vector<State>(20, zero_state));
    } catch (const std::exception&) {
      threw = true;
    }
    EXPECT_TRUE(threw);
  }

  // ---- Lbfgsb on the regression objective: inside a box that cuts some minimisers off ----------------------
  cppoptlib::solver::Lbfgsb<Objective> boxed;
  boxed.stopping_progress.x_delta = 1e-11;
  boxed.stopping_progress.f_delta = 0;
  boxed.stopping_progress.gradient_norm = 1e-9;
  boxed.stopping_progress.past = 0;
  Objective::VectorType lo(n), hi(n);
  for (int i = 0; i < n; ++i) {
    lo[i] = -0.05;
    hi[i] = 0.05;
  }
  boxed.SetBounds(lo, hi);
  const auto one = boxed.Minimize(objectives[3], starts[3]);           // one function: its row, replicated once
  const auto many = boxed.MinimizeBatch(objectives, starts);
  EXPECT_EQ(many.size(), size_t(B));
  int active = 0;
  for (int b = 0; b < B; ++b) {
    const auto& st = std::get<0>(many[b]);
    for (int i = 0; i < n; ++i) {
      EXPECT_TRUE(st.x[i] >= -0.05 && st.x[i] <= 0.05);
      active += (st.x[i] == -0.05 || st.x[i] == 0.05);
      // KKT: the gradient vanishes on free coordinates and points outward on active ones
      const double gi = st.gradient[i];
      if (st.x[i] > -0.05 && st.x[i] < 0.05) EXPECT_TRUE(std::fabs(gi) <= 1e-6);
      if (st.x[i] == -0.05) EXPECT_TRUE(gi >= -1e-6);
      if (st.x[i] == 0.05) EXPECT_TRUE(gi <= 1e-6);
    }
  }
  EXPECT_TRUE(active > 0);
  for (int i = 0; i < n; ++i) EXPECT_EQ(std::get<0>(one).x[i], std::get<0>(many[3]).x[i]);
  {
    cppoptlib::mi355::DeviceGroup group({0, 0});
    cppoptlib::mi355::GlobalFlag flag;
    const auto sharded = boxed.ShardedMinimizeBatch(objectives[3], std::vector<State>(70, zero_state),
                                                    group, &flag);
    EXPECT_EQ(flag.total, uint64_t(70));
    EXPECT_TRUE(flag.all_converged());
    for (int i = 0; i < n; ++i) EXPECT_EQ(std::get<0>(sharded[69]).x[i], std::get<0>(one).x[i]);
  }

  // ---- step_callback_ assigned directly ----------------------------------------------------------------------
  {
    cppoptlib::solver::Lbfgs<Objective> traced;
    size_t calls = 0, last_iteration = 0;
    traced.step_callback_ = [&](const Objective&, const FunctionState<double, Objective::Dimension>&,
                                const cppoptlib::solver::Progress<Objective, FunctionState<double, Objective::Dimension>>& p) {
      ++calls;
      last_iteration = p.num_iterations;
    };
    EXPECT_TRUE(traced.HasCallback());
    auto [s, p] = traced.Minimize(objectives[0], starts[0]);
    EXPECT_EQ(calls, p.num_iterations + 1);      // before every step and once after the loop (solver.h:197, :222)
    EXPECT_EQ(last_iteration, p.num_iterations);
    cppoptlib::solver::Lbfgs<Objective> quiet;
    EXPECT_TRUE(!quiet.HasCallback());
  }
  TEST_MAIN_END();
}
