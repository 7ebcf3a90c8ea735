// The reference's example src/examples/svm_dual_lbfgsb.cc:62-130 over the drop-in headers: soft-margin SVM in the dual
// formulation (intercept dropped), box constraints 0 <= alpha_i <= C handled natively by L-BFGS-B, start at alpha = 0.
// The objective is a USER objective: host functor svm_dual_function.h, device twin svm_dual.hpp, compiled into
// libmi355_lbfgs_svm.so by   python -c "import __graft_entry__ as g; g.build()"
// Build:  g++ -std=c++17 -I include examples/user_objective_svm_dual/svm_dual_lbfgsb.cc
//             -L cppnumericalsolvers_amd -l:libmi355_lbfgs_svm.so -Wl,-rpath,$PWD/cppnumericalsolvers_amd -o svm_dual
// (The reference trains on Iris versicolor / virginica; that table is not vendored here: two z-scored Gaussian blobs of
//  100 samples stand in, so the dual has the example's dimension.)
#include <cmath>
#include <cstdint>
#include <iostream>
#include <vector>

#include "cppoptlib/function.h"
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
  user_examples::SvmDualObjective dual_objective(features, labels, d);

  using Vector = user_examples::SvmDualObjective::VectorType;
  Vector lower_bound(N), upper_bound(N), initial_alpha(N);
  for (int i = 0; i < N; ++i) {
    lower_bound[i] = 0.0;
    upper_bound[i] = regularisation_c;
    initial_alpha[i] = 0.0;
  }
  cppoptlib::solver::Lbfgsb<user_examples::SvmDualObjective> solver;
  solver.SetBounds(lower_bound, upper_bound);
  auto [solution, progress] = solver.Minimize(dual_objective, cppoptlib::function::FunctionState(initial_alpha));

  // recover the primal weights (b = 0 by the intercept-dropped simplification)
  std::vector<double> w(d, 0.0);
  int support = 0, at_bound = 0;
  bool feasible = true;
  for (int i = 0; i < N; ++i) {
    const double a = solution.x[i];
    feasible = feasible && a >= 0.0 && a <= regularisation_c;
    support += a > 1e-5;
    at_bound += a == regularisation_c;
    for (int j = 0; j < d; ++j) w[j] += a * labels[i] * features[static_cast<size_t>(i) * d + j];
  }
  int correct = 0;
  for (int i = 0; i < N; ++i) {
    double score = 0;
    for (int j = 0; j < d; ++j) score += features[static_cast<size_t>(i) * d + j] * w[j];
    correct += ((score >= 0) == (labels[i] > 0));
  }
  const double accuracy = static_cast<double>(correct) / N;
  Vector g;
  const double host_value = dual_objective(solution.x, &g);
  double pg = 0;   // projected gradient sup-norm: the KKT residual of the box-constrained problem
  for (int i = 0; i < N; ++i) {
    double gi = g[i];
    if (solution.x[i] <= 0.0 && gi > 0) gi = 0;
    if (solution.x[i] >= regularisation_c && gi < 0) gi = 0;
    pg = std::fmax(pg, std::fabs(gi));
  }
  std::cout << "SVM dual (L-BFGS-B on the MI355X, b = 0 relaxation, user device objective)\n";
  std::cout << "  solver status:  " << progress.status << "\n";
  std::cout << "  iterations:     " << progress.num_iterations << "\n";
  std::cout << "  dual objective: " << solution.value << "\n";
  std::cout << "  support vecs:   " << support << " / " << N << " (" << at_bound << " at the bound C)\n";
  std::cout << "  w:             ";
  for (int j = 0; j < d; ++j) std::cout << " " << w[j];
  std::cout << "\n  accuracy:       " << accuracy << "\n";
  std::cout << "  projected gradient: " << pg << "\n";
  const bool ok = feasible && accuracy > 0.9 && support > 0 && support < N && at_bound > 0 && solution.value < 0 &&
                  std::fabs(host_value - solution.value) <= 1e-9 * std::fmax(1.0, std::fabs(host_value)) && pg < 1e-2 &&
                  progress.status != cppoptlib::solver::Status::IterationLimit;
  std::cout << (ok ? "PASS" : "FAIL") << "\n";
  return ok ? 0 : 1;
}
