// The reference's example src/examples/svm_primal_lbfgs.cc:105-140 over the drop-in headers: soft-margin SVM primal with
// a squared hinge loss, minimised by calling L-BFGS directly.  The objective is a USER objective: its host functor is
// svm_function.h, its device twin svm_squared_hinge.hpp, compiled into libmi355_lbfgs_svm.so by
//   python -c "import __graft_entry__ as g; g.build()"
// Build:  g++ -std=c++17 -I include examples/user_objective_svm/svm_primal_lbfgs.cc
//             -L cppnumericalsolvers_amd -l:libmi355_lbfgs_svm.so -Wl,-rpath,$PWD/cppnumericalsolvers_amd -o svm
// (The reference trains on Iris versicolor / virginica; that table is not vendored here, two Gaussian blobs stand in.)
#include <cmath>
#include <cstdint>
#include <iostream>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "svm_function.h"

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
      features[static_cast<size_t>(i) * d + j] = gauss + labels[i] * (0.5 + j * 0.33);
    }
  }
  constexpr double regularisation_c = 1.0;
  user_examples::SvmPrimalSquaredHinge objective(features, labels, d, regularisation_c);

  user_examples::SvmPrimalSquaredHinge::VectorType initial_x(d + 1);   // start at the origin: w = 0, b = 0
  for (int j = 0; j <= d; ++j) initial_x[j] = 0.0;

  cppoptlib::solver::Lbfgs<user_examples::SvmPrimalSquaredHinge> solver;
  int callbacks = 0;
  solver.SetCallback([&](const auto&, const auto&, const auto&) { ++callbacks; });
  auto [solution, progress] = solver.Minimize(objective, cppoptlib::function::FunctionState(initial_x));

  int correct = 0;
  for (int i = 0; i < N; ++i) {
    double score = solution.x[d];
    for (int j = 0; j < d; ++j) score += features[static_cast<size_t>(i) * d + j] * solution.x[j];
    correct += ((score >= 0) == (labels[i] > 0));
  }
  const double accuracy = static_cast<double>(correct) / N;
  user_examples::SvmPrimalSquaredHinge::VectorType g;
  const double host_value = objective(solution.x, &g);
  double gnorm = 0;
  for (int j = 0; j <= d; ++j) gnorm = std::fmax(gnorm, std::fabs(g[j]));

  std::cout << "SVM primal (L-BFGS on the MI355X, squared hinge, user device objective)\n";
  std::cout << "  solver status: " << progress.status << "\n";
  std::cout << "  iterations:    " << progress.num_iterations << " (callback invoked " << callbacks << " times)\n";
  std::cout << "  objective:     " << solution.value << "\n";
  std::cout << "  w:            ";
  for (int j = 0; j < d; ++j) std::cout << " " << solution.x[j];
  std::cout << "\n  b:             " << solution.x[d] << "\n";
  std::cout << "  accuracy:      " << accuracy << "\n";
  const bool ok = accuracy > 0.9 && std::fabs(host_value - solution.value) <= 1e-9 * std::fmax(1.0, std::fabs(host_value)) &&
                  gnorm < 1e-3 && callbacks == static_cast<int>(progress.num_iterations) + 1 &&
                  progress.status != cppoptlib::solver::Status::IterationLimit;
  // The same functor under the box-constrained solver (as src/examples/linear_regression.cc:58-74 uses Lbfgsb on its
  // own functor): weights held in [-0.25, 0.25], the offset free.
  cppoptlib::solver::Lbfgsb<user_examples::SvmPrimalSquaredHinge> boxed;
  user_examples::SvmPrimalSquaredHinge::VectorType lower(d + 1), upper(d + 1);
  for (int j = 0; j < d; ++j) {
    lower[j] = -0.25;
    upper[j] = 0.25;
  }
  lower[d] = -1e3;
  upper[d] = 1e3;
  boxed.SetBounds(lower, upper);
  auto [bsolution, bprogress] = boxed.Minimize(objective, cppoptlib::function::FunctionState(initial_x));
  bool inside = true, active = false;
  for (int j = 0; j < d; ++j) {
    inside = inside && bsolution.x[j] >= -0.25 && bsolution.x[j] <= 0.25;
    active = active || std::fabs(bsolution.x[j]) == 0.25;
  }
  const double bhost = objective(bsolution.x, nullptr);
  std::cout << "  boxed (Lbfgsb): objective " << bsolution.value << " after " << bprogress.num_iterations
            << " iterations, w =";
  for (int j = 0; j < d; ++j) std::cout << " " << bsolution.x[j];
  std::cout << "\n";
  const bool bok = inside && active && bsolution.value >= solution.value &&
                   std::fabs(bhost - bsolution.value) <= 1e-9 * std::fmax(1.0, std::fabs(bhost)) &&
                   bprogress.status != cppoptlib::solver::Status::IterationLimit;
  std::cout << ((ok && bok) ? "PASS" : "FAIL") << "\n";
  return (ok && bok) ? 0 : 1;
}
