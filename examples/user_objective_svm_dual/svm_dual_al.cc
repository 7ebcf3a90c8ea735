// The reference's example src/examples/svm_dual_al.cc:84-160 over the drop-in headers: soft-margin SVM in the FULL Wolfe
// dual — box constraints 0 <= alpha_i <= C handled natively by the inner L-BFGS-B, the classifier equality
// sum_i alpha_i y_i = 0 by the outer augmented-Lagrangian loop ("AL outside, L-BFGS-B inside").  On the MI355X the whole
// outer loop runs inside one launch of the L-BFGS-B kernel (thirty-two lanes x four coordinates for the 100 variables);
// the objective is the USER functor of svm_dual.hpp as a term (kind 103) with the kernel matrix as its parameter blob,
// the equality the menu's linear form.
// Build:  g++ -std=c++17 -I include examples/user_objective_svm_dual/svm_dual_al.cc
//             -L cppnumericalsolvers_amd -l:libmi355_lbfgs_svm.so -Wl,-rpath,$PWD/cppnumericalsolvers_amd -o svm_dual_al
// (Data: as svm_dual_lbfgsb.cc — two z-scored Gaussian blobs of 100 samples stand in for the Iris table.)
#include <cmath>
#include <cstdint>
#include <iostream>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "svm_dual_function.h"

int main() {
  const int N = 100, d = 4;
  std::vector<double> features(static_cast<size_t>(N) * d), labels(N);
  uint64_t state = 88172645463325252ULL;   // xorshift64: deterministic synthetic data
  auto uniform = [&]() {
    state ^= state << 13; state ^= state >> 7; state ^= state << 17;
    return static_cast<double>(state >> 11) / 9007199254740992.0;
  };
  for (int i = 0; i < N; ++i) {
    labels[i] = (i % 2 == 0) ? 1.0 : -1.0;
    for (int j = 0; j < d; ++j) {
      const double gauss = std::sqrt(-2.0 * std::log(uniform() + 1e-300)) * std::cos(6.283185307179586 * uniform());
      features[static_cast<size_t>(i) * d + j] = gauss + labels[i] * (0.4 + j * 0.25);
    }
  }
  for (int j = 0; j < d; ++j) {   // z-score every feature (the reference's loader standardises the Iris columns)
    double mean = 0, var = 0;
    for (int i = 0; i < N; ++i) mean += features[static_cast<size_t>(i) * d + j];
    mean /= N;
    for (int i = 0; i < N; ++i) var += (features[static_cast<size_t>(i) * d + j] - mean) * (features[static_cast<size_t>(i) * d + j] - mean);
    const double sd = std::sqrt(var / N);
    for (int i = 0; i < N; ++i) features[static_cast<size_t>(i) * d + j] = (features[static_cast<size_t>(i) * d + j] - mean) / sd;
  }
  constexpr double regularisation_c = 1.0;

  using namespace cppoptlib::function;
  using Problem = ConstrainedOptimizationProblem<>;
  using Vector = Problem::VectorType;
  user_examples::SvmDualObjective objective(features, labels, d);
  const LinearForm<> equality(labels);                      // c(alpha) = sum_i alpha_i y_i, gradient y
  Problem problem(objective, /*eq=*/{equality});

  // inner solver: L-BFGS-B with the box on alpha; the augmented Lagrangian keeps a copy, bounds included
  using Inner = cppoptlib::solver::Lbfgsb<AugmentedLagrangianFunction<>>;
  Inner inner_solver;
  Vector lower_bound(N), upper_bound(N), initial_alpha(N);
  for (int i = 0; i < N; ++i) {
    lower_bound[i] = 0.0;
    upper_bound[i] = regularisation_c;
    initial_alpha[i] = 0.0;
  }
  inner_solver.SetBounds(lower_bound, upper_bound);
  cppoptlib::solver::AugmentedLagrangian<Problem, Inner> solver(problem, inner_solver);

  cppoptlib::solver::AugmentedLagrangeState<double> al_state(initial_alpha, /*num_eq=*/1, /*num_ineq=*/0, /*penalty=*/1.0);
  auto [solution, progress] = solver.Minimize(al_state);

  std::vector<double> w(d, 0.0);
  int support = 0;
  bool in_box = true;
  double weighted_sum = 0;
  for (int i = 0; i < N; ++i) {
    const double a = solution.x[i];
    in_box = in_box && a >= 0.0 && a <= regularisation_c;
    support += a > 1e-5;
    weighted_sum += a * labels[i];
    for (int j = 0; j < d; ++j) w[j] += a * labels[i] * features[static_cast<size_t>(i) * d + j];
  }
  int correct = 0;
  for (int i = 0; i < N; ++i) {
    double score = 0;
    for (int j = 0; j < d; ++j) score += features[static_cast<size_t>(i) * d + j] * w[j];
    correct += ((score >= 0) == (labels[i] > 0));
  }
  const double accuracy = static_cast<double>(correct) / N;
  std::cout << "SVM dual (augmented Lagrangian + L-BFGS-B on the MI355X)\n";
  std::cout << "  solver status:     " << progress.status << "\n";
  std::cout << "  outer iterations:  " << progress.num_iterations << "\n";
  std::cout << "  max violation:     " << solution.max_violation << "\n";
  std::cout << "  multiplier lambda: " << solution.multiplier_state.equality_multipliers[0] << "\n";
  std::cout << "  support vectors:   " << support << " / " << N << "\n";
  std::cout << "  sum(alpha * y):    " << weighted_sum << "\n";
  std::cout << "  w:                ";
  for (int j = 0; j < d; ++j) std::cout << " " << w[j];
  std::cout << "\n  accuracy:          " << accuracy << "\n";
  const bool ok = in_box && accuracy > 0.9 && support > 0 && support < N && std::fabs(weighted_sum) <= 1e-4 &&
                  solution.max_violation <= 1e-4 && progress.status != cppoptlib::solver::Status::IterationLimit;
  std::cout << (ok ? "PASS" : "FAIL") << "\n";
  return ok ? 0 : 1;
}
