// svm_dual_function.h — host side of the second user-objective example: the functor of the reference's
// src/examples/svm_dual_lbfgsb.cc:36-60 (dual soft-margin SVM on the alpha vector, kernel-with-labels matrix Q
// precomputed) as a FunctionCRTP class that names its device twin.  kDeviceObjective = 101 is the id the device functor
// of svm_dual.hpp was registered under when libmi355_lbfgs_svm.so was built; DeviceParams() is the blob its load() reads.
// kAlTermKind = 103 names the same functor as an augmented-Lagrangian term (svm_dual_al.cc).
#ifndef EXAMPLES_USER_OBJECTIVE_SVM_DUAL_SVM_DUAL_FUNCTION_H_
#define EXAMPLES_USER_OBJECTIVE_SVM_DUAL_SVM_DUAL_FUNCTION_H_

#include <vector>

#include "cppoptlib/function.h"

namespace user_examples {

class SvmDualObjective
    : public cppoptlib::function::FunctionCRTP<SvmDualObjective, double,
                                               cppoptlib::function::DifferentiabilityMode::First> {
 public:
  static constexpr int kDeviceObjective = MI355_OBJ_USER_FIRST + 1;  // 101
  // ... and as a TERM of a ConstrainedOptimizationProblem (src/examples/svm_dual_al.cc): the same device functor
  // registered as term kind 103, its parameters the same blob (mi355_al_problem.user_params)
  static constexpr int kAlTermKind = MI355_AL_TERM_USER + 3;
  std::vector<double> AlCoefficients(int n) const { return std::vector<double>(static_cast<size_t>(n) + 1, 0.0); }
  std::vector<double> AlUserParams() const { return DeviceParams(); }

  // features: N x d row major; labels: N values +/- 1.  Q = (X X^T) .* (y y^T), accumulated feature by feature so that it
  // is symmetric to the bit (the device functor walks columns where this class walks rows).
  SvmDualObjective(const std::vector<double>& features, const std::vector<double>& labels, int feature_count)
      : n_(static_cast<int>(labels.size())), q_(static_cast<size_t>(n_) * n_, 0.0) {
    for (int k = 0; k < feature_count; ++k)
      for (int i = 0; i < n_; ++i)
        for (int j = 0; j < n_; ++j) {
          const double t = features[static_cast<size_t>(i) * feature_count + k] * features[static_cast<size_t>(j) * feature_count + k];
          q_[static_cast<size_t>(i) * n_ + j] = (k == 0) ? t : q_[static_cast<size_t>(i) * n_ + j] + t;
        }
    for (int i = 0; i < n_; ++i)
      for (int j = 0; j < n_; ++j) q_[static_cast<size_t>(i) * n_ + j] = q_[static_cast<size_t>(i) * n_ + j] * (labels[i] * labels[j]);
  }

  int GetDimension() const { return n_; }
  std::vector<double> DeviceParams() const {
    std::vector<double> p{static_cast<double>(n_)};
    p.insert(p.end(), q_.begin(), q_.end());
    return p;
  }

  ScalarType operator()(const VectorType& alpha, VectorType* grad = nullptr) const {
    std::vector<double> q(static_cast<size_t>(n_));
    for (int i = 0; i < n_; ++i) {
      double acc = q_[static_cast<size_t>(i) * n_] * alpha[0];
      for (int j = 1; j < n_; ++j) acc = acc + q_[static_cast<size_t>(i) * n_ + j] * alpha[j];
      q[static_cast<size_t>(i)] = acc;
    }
    double aq = alpha[0] * q[0], sa = alpha[0];
    for (int i = 1; i < n_; ++i) {
      aq = aq + alpha[i] * q[static_cast<size_t>(i)];
      sa = sa + alpha[i];
    }
    if (grad) {
      grad->resize(n_);
      for (int i = 0; i < n_; ++i) (*grad)[i] = q[static_cast<size_t>(i)] - 1.0;
    }
    return 0.5 * aq - sa;
  }

 private:
  int n_;
  std::vector<double> q_;
};

}  // namespace user_examples
#endif  // EXAMPLES_USER_OBJECTIVE_SVM_DUAL_SVM_DUAL_FUNCTION_H_
