// svm_function.h — host side of the user-objective example: the functor of the reference's
// src/examples/svm_primal_lbfgs.cc:35-103 (soft-margin SVM primal, squared hinge) as a FunctionCRTP class that names
// its device twin.  `kDeviceObjective = 100` is the id the device functor of svm_squared_hinge.hpp was registered
// under when the library was built; DeviceParams() is the blob its load() reads.
#ifndef EXAMPLES_USER_OBJECTIVE_SVM_SVM_FUNCTION_H_
#define EXAMPLES_USER_OBJECTIVE_SVM_SVM_FUNCTION_H_

#include <vector>

#include "cppoptlib/function.h"

namespace user_examples {

class SvmPrimalSquaredHinge
    : public cppoptlib::function::FunctionCRTP<SvmPrimalSquaredHinge, double,
                                               cppoptlib::function::DifferentiabilityMode::First> {
 public:
  static constexpr int kDeviceObjective = MI355_OBJ_USER_FIRST;  // 100

  // features: N x d row major; labels: N values +/- 1
  SvmPrimalSquaredHinge(std::vector<double> features, std::vector<double> labels, int feature_count, double c)
      : features_(std::move(features)), labels_(std::move(labels)), d_(feature_count), c_(c) {}

  int GetDimension() const { return d_ + 1; }
  std::vector<double> DeviceParams() const {
    std::vector<double> p{static_cast<double>(labels_.size()), static_cast<double>(d_), c_};
    p.insert(p.end(), features_.begin(), features_.end());
    p.insert(p.end(), labels_.begin(), labels_.end());
    return p;
  }

  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    const int N = static_cast<int>(labels_.size());
    std::vector<double> ws(static_cast<size_t>(N));
    double hinge = 0, ww = 0;
    for (int i = 0; i < N; ++i) {
      double score = features_[static_cast<size_t>(i) * d_] * x[0];
      for (int j = 1; j < d_; ++j) score = score + features_[static_cast<size_t>(i) * d_ + j] * x[j];
      score = score + x[d_];
      const double t = 1.0 - labels_[i] * score;
      const double slack = (t < 0.0) ? 0.0 : t;
      ws[i] = (-2.0 * slack) * labels_[i];
      hinge = (i == 0) ? slack * slack : hinge + slack * slack;
    }
    for (int j = 0; j < d_; ++j) ww = (j == 0) ? x[0] * x[0] : ww + x[j] * x[j];
    if (grad) {
      grad->resize(d_ + 1);
      for (int j = 0; j < d_; ++j) {
        double acc = features_[j] * ws[0];
        for (int i = 1; i < N; ++i) acc = acc + features_[static_cast<size_t>(i) * d_ + j] * ws[i];
        (*grad)[j] = x[j] + c_ * acc;
      }
      double acc = ws[0];
      for (int i = 1; i < N; ++i) acc = acc + ws[i];
      (*grad)[d_] = c_ * acc;
    }
    return 0.5 * ww + c_ * hinge;
  }

 private:
  std::vector<double> features_, labels_;
  int d_;
  double c_;
};

}  // namespace user_examples
#endif  // EXAMPLES_USER_OBJECTIVE_SVM_SVM_FUNCTION_H_
