// MinimizeBatch(functions, states): which batches are REFUSED, and why (round-4 advisor findings).  Every refusal below
// is decided on the host before anything touches the device, so this test also runs on a box without a GPU
// (tests/test_capi_and_host.py); on the GPU box it runs with the others.
//   * a regularisation sweep (same matrix, different lambda) is not solved with functions[0]'s lambda;
//   * SetArithmetic(MI355_ARITH_EXACT) is not silently dropped for functions with their own matrices;
//   * ONE function that differs anywhere in its parameter blob — not only at the three entries the old fingerprint
//     looked at, not only at a sampled position of the batch — is seen (a hash of the whole blob, every function);
//   * small-blob function types (DiagQuadratic) are compared in full, every function;
//   * an own-matrix batch whose sampled functions are outside the pinned conditioning envelope is refused under
//     MI355_ARITH_DEFAULT.
#include <cmath>
#include <random>
#include <string>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/function_expressions.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;

template <class Call>
static std::string Refusal(Call&& call) {
  try {
    call();
  } catch (const std::exception& e) {
    return e.what();
  }
  return "";
}

int main() {
  constexpr int rows = 12, n = 6, B = 100;
  std::mt19937_64 rng(11);
  std::normal_distribution<double> gauss(0.0, 1.0);
  std::vector<double> A(size_t(rows) * n), y(rows);
  for (double& v : A) v = gauss(rng) / std::sqrt(double(rows));
  for (double& v : y) v = gauss(rng);
  using SE = SquaredError<kDynamicDimension, DifferentiabilityMode::First>;
  using L2 = L2Reg<kDynamicDimension, DifferentiabilityMode::First>;
  FunctionExpr proto = SE(rows, n, A, y) + 0.1 * L2(n);
  using Objective = decltype(proto);
  using State = FunctionState<double, Objective::Dimension>;
  Objective::VectorType zero(n);
  for (int i = 0; i < n; ++i) zero[i] = 0;
  const std::vector<State> starts(B, State(zero));

  // ---- a regularisation sweep: same A, lambda_b = 0.1 + 0.01 b ---------------------------------------------------
  {
    std::vector<Objective> sweep;
    for (int b = 0; b < B; ++b) sweep.push_back(SE(rows, n, A, y) + (0.1 + 0.01 * b) * L2(n));
    cppoptlib::solver::Lbfgs<Objective> solver;
    const std::string why = Refusal([&] { solver.MinimizeBatch(sweep, starts); });
    std::printf("sweep over lambda: %s\n", why.c_str());
    EXPECT_TRUE(why.find("lambda") != std::string::npos);
    // ... also when only ONE function of the batch has another lambda, at a position no sample of 16 visits
    std::vector<Objective> one(B, SE(rows, n, A, y) + 0.1 * L2(n));
    one[37] = SE(rows, n, A, y) + 0.2 * L2(n);
    EXPECT_TRUE(Refusal([&] { solver.MinimizeBatch(one, starts); }).find("lambda") != std::string::npos);
    cppoptlib::solver::Lbfgsb<Objective> box_solver;
    EXPECT_TRUE(Refusal([&] { box_solver.MinimizeBatch(one, starts); }).find("share their device parameters") !=
                std::string::npos);
  }

  // ---- own matrices + MI355_ARITH_EXACT ---------------------------------------------------------------------------
  {
    std::vector<Objective> own;
    for (int b = 0; b < B; ++b) {
      std::vector<double> Ab(A);
      for (double& v : Ab) v += 0.05 * gauss(rng);
      own.push_back(SE(rows, n, Ab, y) + 0.1 * L2(n));
    }
    cppoptlib::solver::Lbfgs<Objective> solver;
    solver.SetArithmetic(MI355_ARITH_EXACT);
    const std::string why = Refusal([&] { solver.MinimizeBatch(own, starts); });
    std::printf("own matrices, exact arithmetic: %s\n", why.c_str());
    EXPECT_TRUE(why.find("MI355_ARITH_EXACT") != std::string::npos);
  }

  // ---- one function differs in ONE matrix entry the old fingerprint never read, at an unsampled position ----------
  {
    std::vector<Objective> batch(B, SE(rows, n, A, y) + 0.1 * L2(n));
    std::vector<double> A2(A);
    A2[5] += 0.25;                        // not entry 0, size / 2 or size - 1
    batch[41] = SE(rows, n, A2, y) + 0.1 * L2(n);
    EXPECT_TRUE(cppoptlib::mi355::ParamsHash(batch[41]) != cppoptlib::mi355::ParamsHash(batch[0]));
    EXPECT_TRUE(cppoptlib::mi355::ParamsHash(batch[40]) == cppoptlib::mi355::ParamsHash(batch[0]));
    EXPECT_TRUE(!cppoptlib::mi355::SharesDeviceParams(batch));
    cppoptlib::solver::Lbfgsb<Objective> box_solver;   // no own-matrix form there: refused, not solved with A
    EXPECT_TRUE(Refusal([&] { box_solver.MinimizeBatch(batch, starts); }).find("share their device parameters") !=
                std::string::npos);
  }

  // ---- small blobs are compared in full: DiagQuadratic with ONE different coefficient -----------------------------
  {
    using DQ = DiagQuadratic<>;
    std::vector<double> a(n, 2.0);
    std::vector<DQ> batch(B, DQ(a, 1.0));
    std::vector<double> a2(a);
    a2[3] = 2.5;
    batch[73] = DQ(a2, 1.0);
    DQ::VectorType z(n);
    for (int i = 0; i < n; ++i) z[i] = 1.0;
    const std::vector<FunctionState<double>> st(B, FunctionState<double>(z));
    cppoptlib::solver::Lbfgs<DQ> solver;
    const std::string why = Refusal([&] { solver.MinimizeBatch(batch, st); });
    std::printf("DiagQuadratic with one different coefficient: %s\n", why.c_str());
    EXPECT_TRUE(why.find("share their device parameters") != std::string::npos);
  }

  // ---- own matrices outside the pinned conditioning envelope (lambda tiny) under the DEFAULT policy ----------------
  {
    std::vector<Objective> own;
    for (int b = 0; b < B; ++b) {
      std::vector<double> Ab(A);
      for (double& v : Ab) v += 0.05 * gauss(rng);
      own.push_back(SE(rows, n, Ab, y) + 1e-6 * L2(n));
    }
    EXPECT_TRUE(cppoptlib::mi355::ConditionBound(own[0]) > MI355_RIDGE_GRAM_MAX_CONDITION_BOUND);
    cppoptlib::solver::Lbfgs<Objective> solver;
    const std::string why = Refusal([&] { solver.MinimizeBatch(own, starts); });
    std::printf("own matrices, lambda 1e-6: %s\n", why.c_str());
    EXPECT_TRUE(why.find("MI355_RIDGE_GRAM_MAX_CONDITION_BOUND") != std::string::npos);
    // the well-conditioned batch of batch_functions_test.cc is inside the envelope
    Objective fine = SE(rows, n, A, y) + 0.1 * L2(n);
    EXPECT_TRUE(cppoptlib::mi355::ConditionBound(fine) <= MI355_RIDGE_GRAM_MAX_CONDITION_BOUND);
    // the bound is the smaller of Gershgorin and the trace-of-powers bound (round-5 advisor: Gershgorin alone refused
    // ordinary batches with a small lambda): it stays above the true condition number — lambda_max by power iteration —
    // and within n^(1/32) (1.14 at n = 64) of it when lambda_min is the regulariser
    for (const double lambda : {0.1, 0.02}) {
      Objective f = SE(rows, n, A, y) + lambda * L2(n);
      std::vector<double> G(static_cast<size_t>(n) * n, 0.0), v(n, 1.0), w(n);
      double gershgorin = 0;
      for (int j = 0; j < n; ++j) {
        double row_sum = 0;
        for (int k = 0; k < n; ++k) {
          double acc = 0;
          for (int i = 0; i < rows; ++i) acc += A[static_cast<size_t>(i) * n + j] * A[static_cast<size_t>(i) * n + k];
          G[static_cast<size_t>(j) * n + k] = acc + (j == k ? lambda : 0.0);
          row_sum += std::fabs(G[static_cast<size_t>(j) * n + k]);
        }
        gershgorin = std::max(gershgorin, row_sum);
      }
      double lambda_max = 0;
      for (int it = 0; it < 2000; ++it) {
        double norm = 0;
        for (int j = 0; j < n; ++j) {
          w[j] = 0;
          for (int k = 0; k < n; ++k) w[j] += G[static_cast<size_t>(j) * n + k] * v[k];
          norm += w[j] * w[j];
        }
        lambda_max = std::sqrt(norm);
        for (int j = 0; j < n; ++j) v[j] = w[j] / lambda_max;
      }
      const double bound = cppoptlib::mi355::ConditionBound(f);
      std::printf("lambda %.2f: lambda_max %.4f, Gershgorin %.4f, bound * lambda %.4f\n", lambda, lambda_max, gershgorin, bound * lambda);
      EXPECT_TRUE(bound * lambda >= lambda_max * (1.0 - 1e-9));
      EXPECT_TRUE(bound * lambda <= lambda_max * 1.15);
      EXPECT_TRUE(bound * lambda <= gershgorin);
    }
  }
  TEST_MAIN_END();
}
