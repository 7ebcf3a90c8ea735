// The reference's example src/examples/svm_primal_al.cc:121-195 over the drop-in headers: soft-margin SVM in its PRIMAL form,
//     min 0.5 ||w||^2 + C sum_i xi_i   s.t.   y_i (w . x_i + b) - 1 + xi_i >= 0,   xi_i >= 0,      i = 1..N,
// solved by AugmentedLagrangian over an Lbfgs inner solver: d + 1 + N = 105 variables and 2 N = 200 inequality constraints
// pushed one by one into the problem's constraint vector, exactly as the reference's main() does (:139-147).  Every
// constraint is affine, `LinearForm(a_i) - k_i`; a constraint vector longer than the device's term table travels to the
// GPU as a constraint FAMILY (a matrix: mi355_al_problem.family_ineq), and the whole outer loop — 200 multipliers per
// problem included — runs inside one launch of the L-BFGS kernel, one problem per wavefront.
// The objective is the sum of two menu functions: the diagonal quadratic 0.5 ||w||^2 and the linear form C sum(xi).
// Build:  g++ -std=c++17 -I include examples/svm_primal_al/svm_primal_al.cc
//             -L cppnumericalsolvers_amd -lmi355_lbfgs -Wl,-rpath,$PWD/cppnumericalsolvers_amd -o svm_primal_al
// (Data: two z-scored Gaussian blobs of 100 samples stand in for the reference's Iris table, as in the other SVM examples.)
#include <cmath>
#include <cstdint>
#include <iostream>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/augmented_lagrangian.h"
#include "cppoptlib/solver/lbfgs.h"

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
  const int variable_count = d + 1 + N;   // w, b, xi

  using namespace cppoptlib::function;
  using Problem = ConstrainedOptimizationProblem<>;
  using Vector = Problem::VectorType;
  using Constraint = Problem::ConstraintFunctionType;

  // f(w, b, xi) = 0.5 ||w||^2 + C sum(xi)
  std::vector<double> half(variable_count, 0.0), slack_weights(variable_count, 0.0);
  for (int j = 0; j < d; ++j) half[j] = 0.5;
  for (int i = 0; i < N; ++i) slack_weights[d + 1 + i] = regularisation_c;
  const auto objective = DiagQuadratic<>(half, 0.0) + LinearForm<>(slack_weights);

  // the constraint list is a flat vector built up by value, as in the reference's main()
  std::vector<Constraint> inequality_constraints;
  inequality_constraints.reserve(2 * N);
  for (int i = 0; i < N; ++i) {   // margin: y_i (w . x_i + b) - 1 + xi_i >= 0
    std::vector<double> a(variable_count, 0.0);
    for (int j = 0; j < d; ++j) a[j] = labels[i] * features[static_cast<size_t>(i) * d + j];
    a[d] = labels[i];
    a[d + 1 + i] = 1.0;
    inequality_constraints.emplace_back(LinearForm<>(a) - 1.0);
  }
  for (int i = 0; i < N; ++i) {   // slack: xi_i >= 0
    std::vector<double> a(variable_count, 0.0);
    a[d + 1 + i] = 1.0;
    inequality_constraints.emplace_back(LinearForm<>(a));
  }
  Problem problem(objective, /*eq=*/{}, inequality_constraints);

  cppoptlib::solver::Lbfgs<AugmentedLagrangianFunction<>> inner_solver;
  cppoptlib::solver::AugmentedLagrangian<Problem, decltype(inner_solver)> solver(problem, inner_solver);

  // start at the origin with slacks at zero (infeasible: most margins are negative), zero multipliers, penalty 1
  Vector initial_x(variable_count);
  for (int j = 0; j < variable_count; ++j) initial_x[j] = 0.0;
  cppoptlib::solver::AugmentedLagrangeState<double> al_state(initial_x, /*num_eq=*/0,
                                                             /*num_ineq=*/static_cast<size_t>(2 * N), /*penalty=*/1.0);
  auto [solution, progress] = solver.Minimize(al_state);

  double w_norm2 = 0, slack_sum = 0;
  for (int j = 0; j < d; ++j) w_norm2 += solution.x[j] * solution.x[j];
  for (int i = 0; i < N; ++i) slack_sum += solution.x[d + 1 + i];
  const double b = solution.x[d];
  int correct = 0, active = 0;
  double worst = 0;
  for (int i = 0; i < N; ++i) {
    double score = b;
    for (int j = 0; j < d; ++j) score += features[static_cast<size_t>(i) * d + j] * solution.x[j];
    correct += ((score >= 0) == (labels[i] > 0));
    const double margin = labels[i] * score - 1.0 + solution.x[d + 1 + i];
    worst = std::fmax(worst, std::fmax(-margin, -solution.x[d + 1 + i]));
    active += solution.multiplier_state.inequality_multipliers[i] > 1e-6;
  }
  const double accuracy = static_cast<double>(correct) / N;
  std::cout << "SVM primal (augmented Lagrangian + L-BFGS on the MI355X, 200 constraints as one family)\n";
  std::cout << "  solver status:     " << progress.status << "\n";
  std::cout << "  outer iterations:  " << progress.num_iterations << "\n";
  std::cout << "  max violation:     " << solution.max_violation << "\n";
  std::cout << "  objective:         " << 0.5 * w_norm2 + regularisation_c * slack_sum << "\n";
  std::cout << "  w:                ";
  for (int j = 0; j < d; ++j) std::cout << " " << solution.x[j];
  std::cout << "\n  b:                 " << b << "\n";
  std::cout << "  active margins:    " << active << " / " << N << "\n";
  std::cout << "  accuracy:          " << accuracy << "\n";
  const bool ok = accuracy > 0.9 && solution.max_violation <= 1e-4 && worst <= 1e-4 && active > 0 && active < N &&
                  solution.multiplier_state.inequality_multipliers.size() == static_cast<size_t>(2 * N);
  std::cout << (ok ? "PASS" : "FAIL") << "\n";
  return ok ? 0 : 1;
}
