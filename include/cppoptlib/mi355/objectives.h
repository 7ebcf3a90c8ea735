// cppoptlib/mi355/objectives.h — objectives that have a device twin.
//
// The reference takes an arbitrary host functor (FunctionCRTP::operator(),
// function_base.h:103-120).  A GPU engine needs the objective as device code, so
// a function type advertises its twin through two members:
//     static constexpr int kDeviceObjective;        // an mi355_objective id
//     std::vector<double> DeviceParams() const;     // shared parameter blob
// (the traits that read them, and the run-time record a type-erased FunctionExpr
// carries instead, are in cppoptlib/mi355/device_twin.h).
// The classes below are ordinary FunctionCRTP functors (operator() works on the
// host, same operation order as the device code) that carry those members.
// Lbfgs<F>::Minimize refuses at compile time a function type without a twin —
// there is no CPU fallback.  At the bottom: the `twin::` builders, with which a
// user functor states its twin in one line (`auto DeviceTwin() const { return
// cppoptlib::mi355::twin::DiagQuadratic({5, 100}, 5); }`).
#ifndef CPPOPTLIB_MI355_OBJECTIVES_H_
#define CPPOPTLIB_MI355_OBJECTIVES_H_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <type_traits>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "../function_base.h"
#include "device_twin.h"

namespace cppoptlib::function {

// Chained Rosenbrock-N: f = sum_{i<N-1} (1-x_i)^2 + 100 (x_{i+1}-x_i^2)^2.
// At N = 2 this is the reference's test functor (src/test/verify.cc:58-69).
template <int TDimension = kDynamicDimension>
class Rosenbrock : public FunctionCRTP<Rosenbrock<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<Rosenbrock<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_ROSENBROCK;
  std::vector<double> DeviceParams() const { return {}; }
  // as a term of a constrained problem (function_problem.h): kind and coefficient row [n + 1]
  static constexpr int kAlTermKind = MI355_AL_TERM_ROSENBROCK;
  std::vector<double> AlCoefficients(int n) const { return std::vector<double>(static_cast<size_t>(n) + 1, 0.0); }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    if (gradient) gradient->resize(n);
    ScalarType f = 0;
    for (int i = 0; i < n; ++i) {
      const bool a = i + 1 < n, b = i > 0;
      ScalarType ga = 0, gb = 0;
      if (a) {
        const ScalarType t1 = 1 - x[i], t2 = x[i + 1] - x[i] * x[i];
        const ScalarType term = t1 * t1 + (100 * t2) * t2;
        f = (i == 0) ? term : f + term;
        ga = -2 * (1 - x[i]) + (200 * t2) * (-2 * x[i]);
      }
      if (b) gb = 200 * (x[i] - x[i - 1] * x[i - 1]);
      if (gradient) (*gradient)[i] = (a && b) ? ga + gb : (a ? ga : gb);
    }
    return f;
  }
};

// The same function declared Second mode, with its tridiagonal Hessian (at N = 2: the Hessian of
// src/examples/trust_region_newton_rosenbrock.cc).  The reference's Lbfgs rebuilds its diagonal preconditioner from
// the Hessian at every iterate (solver/lbfgs.h:116-139); the device takes diag H(x) from the functor's hess_diag.
template <int TDimension = kDynamicDimension>
class RosenbrockSecond
    : public FunctionCRTP<RosenbrockSecond<TDimension>, double, DifferentiabilityMode::Second, TDimension> {
 public:
  using Super = FunctionCRTP<RosenbrockSecond<TDimension>, double, DifferentiabilityMode::Second, TDimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_ROSENBROCK;
  static constexpr bool kDeviceHessianFromFunctor = true;
  std::vector<double> DeviceParams() const { return {}; }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr, MatrixType* hessian = nullptr) const {
    const ScalarType f = Rosenbrock<TDimension>()(x, gradient);
    if (hessian) {
      const int n = static_cast<int>(x.size());
      *hessian = MatrixType(n, n);  // zeros
      for (int i = 0; i < n; ++i) {
        const bool a = i + 1 < n, b = i > 0;
        const ScalarType da = a ? ((1200 * x[i]) * x[i] - 400 * x[i + 1]) + 2 : 0;
        (*hessian)(i, i) = (a && b) ? da + 200 : (a ? da : (b ? 200 : 0));
        if (a) (*hessian)(i, i + 1) = (*hessian)(i + 1, i) = -400 * x[i];
      }
    }
    return f;
  }
};

// f(x) = sum_i a_i x_i^2 + c  (README.md:21-28 quick start: a = (5, 100), c = 5).
template <int TDimension = kDynamicDimension>
class DiagQuadratic
    : public FunctionCRTP<DiagQuadratic<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<DiagQuadratic<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_DIAG_QUADRATIC;

  DiagQuadratic(std::vector<double> a, double c) : a_(std::move(a)), c_(c) {}
  std::vector<double> DeviceParams() const {
    std::vector<double> p = a_;
    p.push_back(c_);
    return p;
  }
  static constexpr int kAlTermKind = MI355_AL_TERM_DIAG_QUADRATIC;
  std::vector<double> AlCoefficients(int n) const {
    if (static_cast<int>(a_.size()) != n) return {};
    return DeviceParams();
  }
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    if (gradient) gradient->resize(n);
    ScalarType f = 0;
    for (int i = 0; i < n; ++i) {
      const ScalarType term = (a_[i] * x[i]) * x[i];
      f = (i == 0) ? term : f + term;
      if (gradient) (*gradient)[i] = (2 * a_[i]) * x[i];
    }
    return f + c_;
  }

 private:
  std::vector<double> a_;
  double c_;
};

// a.dot(x), gradient a — e.g. the `SumObjective` of src/examples/constrained_simple2.cc:13-25 with a = ones.
// A term of a constrained problem (function_problem.h); it has no unconstrained device objective of its own.
template <int TDimension = kDynamicDimension>
class LinearForm : public FunctionCRTP<LinearForm<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<LinearForm<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kAlTermKind = MI355_AL_TERM_LINEAR;

  explicit LinearForm(std::vector<double> a) : a_(std::move(a)) {}
  std::vector<double> AlCoefficients(int n) const {
    if (static_cast<int>(a_.size()) != n) return {};
    std::vector<double> row = a_;
    row.push_back(0.0);
    return row;
  }
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    if (gradient) gradient->resize(n);
    ScalarType f = 0;
    for (int i = 0; i < n; ++i) {
      const ScalarType term = a_[i] * x[i];
      f = (i == 0) ? term : f + term;
      if (gradient) (*gradient)[i] = a_[i];
    }
    return f;
  }

 private:
  std::vector<double> a_;
};

// x.squaredNorm(), gradient 2 x — the `Circle` of src/examples/constrained_simple2.cc:29-39.
template <int TDimension = kDynamicDimension>
class SquaredNorm : public FunctionCRTP<SquaredNorm<TDimension>, double, DifferentiabilityMode::First, TDimension> {
 public:
  using Super = FunctionCRTP<SquaredNorm<TDimension>, double, DifferentiabilityMode::First, TDimension>;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kAlTermKind = MI355_AL_TERM_SQUARED_NORM;
  std::vector<double> AlCoefficients(int n) const { return std::vector<double>(static_cast<size_t>(n) + 1, 0.0); }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    const int n = static_cast<int>(x.size());
    if (gradient) gradient->resize(n);
    ScalarType f = 0;
    for (int i = 0; i < n; ++i) {
      const ScalarType term = x[i] * x[i];
      f = (i == 0) ? term : f + term;
      if (gradient) (*gradient)[i] = 2 * x[i];
    }
    return f;
  }
};

// Ridge least squares f(x) = ||A x - y||^2 + lambda ||x||^2 — what the reference README builds as
// `SquaredError(A, y) + lambda * L2Reg(n)` (README.md:122-167), as one functor with a device twin.
// A is rows x n, row major; rows <= MI355_LBFGS_MAX_ROWS.  TMode = First is the plain L-BFGS path;
// TMode = Second (what the README declares) makes Lbfgs use the diagonal preconditioner
// 1/(|H_jj| + eps) of lbfgs.h:116-139 built from DeviceHessianDiagonal().
template <int TDimension = kDynamicDimension, DifferentiabilityMode TMode = DifferentiabilityMode::First>
class SquaredErrorRidge
    : public FunctionCRTP<SquaredErrorRidge<TDimension, TMode>, double, TMode, TDimension> {
 public:
  static_assert(TMode != DifferentiabilityMode::None, "the ridge functor is differentiable");
  using Super = FunctionCRTP<SquaredErrorRidge<TDimension, TMode>, double, TMode, TDimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_SQUARED_ERROR_RIDGE;
  static constexpr int kDeviceObjectiveFused = MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM;

  SquaredErrorRidge(int rows, int n, std::vector<double> a_row_major, std::vector<double> y, double lambda)
      : rows_(rows), n_(n), a_(std::move(a_row_major)), y_(std::move(y)), lambda_(lambda) {
    // one pass over the shared parameters at construction (the constructor already moved / copied them): a batch check
    // then compares B hashes instead of B blobs (cppoptlib/mi355/batch_driver.h CheckSharedParams)
    params_hash_ = cppoptlib::mi355::HashDoubles(a_.data(), a_.size(),
                                                 cppoptlib::mi355::HashDoubles(&lambda_, 1, 1469598103934665603ull ^
                                                                               (static_cast<uint64_t>(rows_) << 32) ^
                                                                               static_cast<uint64_t>(n_)));
  }
  // 64-bit hash of everything DeviceParams() returns (rows, n, lambda, every entry of A)
  uint64_t DeviceParamsHash() const { return params_hash_; }
  // Upper bound of cond(A^T A + lambda I) = lambda_max(G) / lambda_min(G), lambda_min(G) >= lambda.  lambda_max(G) is
  // bounded twice, both rigorously, and the smaller bound is taken:
  //   * Gershgorin: the largest row sum of |G| — loose by the factor the off-diagonal mass adds (~2 for the random
  //     128 x 64 matrices of the tests, more for correlated columns);
  //   * traces of powers: G is symmetric positive definite, so lambda_max^q <= trace(G^q) <= n lambda_max^q; with
  //     q = 2^p (p squarings of G scaled by its Gershgorin bound) the bound is within n^(1/q) of lambda_max — 14 % at
  //     n = 64, q = 32.  (Round-5 advisor: ordinary well-conditioned batches with a small lambda were refused on the
  //     Gershgorin bound alone.)
  // O(rows n^2 + p n^3): the solvers evaluate it on a SAMPLE of a batch before taking the normal-equation form by default.
  double NormalEquationConditionBound() const {
    if (!(lambda_ > 0)) return std::numeric_limits<double>::infinity();
    const size_t n = static_cast<size_t>(n_);
    std::vector<double> G(n * n);
    double gershgorin = 0;
    for (size_t j = 0; j < n; ++j) {
      double row_sum = 0;
      for (size_t k = 0; k < n; ++k) {
        double acc = 0;
        for (int i = 0; i < rows_; ++i) acc += a_[static_cast<size_t>(i) * n + j] * a_[static_cast<size_t>(i) * n + k];
        G[j * n + k] = acc + (j == k ? lambda_ : 0.0);
        row_sum += std::fabs(G[j * n + k]);
      }
      gershgorin = std::max(gershgorin, row_sum);
    }
    if (!(gershgorin > 0) || !std::isfinite(gershgorin)) return std::numeric_limits<double>::infinity();
    const int squarings = n_ <= 64 ? 5 : (n_ <= 128 ? 4 : 3);
    for (double& v : G) v /= gershgorin;          // eigenvalues in (0, 1]: the powers cannot overflow
    std::vector<double> T(n * n);
    double q = 1;
    for (int p = 0; p < squarings; ++p, q *= 2) {
      for (size_t j = 0; j < n; ++j)
        for (size_t k = j; k < n; ++k) {
          double acc = 0;
          for (size_t t = 0; t < n; ++t) acc += G[j * n + t] * G[t * n + k];
          T[j * n + k] = T[k * n + j] = acc;
        }
      G.swap(T);
    }
    double trace = 0;
    for (size_t j = 0; j < n; ++j) trace += G[j * n + j];
    // (1 + 1e-9: head-room for the rounding of the squarings — the bound stays a bound)
    const double by_trace = (trace > 0 && std::isfinite(trace)) ? gershgorin * std::pow(trace, 1.0 / q) * (1.0 + 1e-9)
                                                                : gershgorin;
    return std::min(gershgorin, by_trace) / lambda_;
  }
  std::vector<double> DeviceParams() const {
    std::vector<double> p{static_cast<double>(rows_), lambda_};
    p.insert(p.end(), a_.begin(), a_.end());
    return p;
  }
  std::vector<double> DevicePerProblem() const { return y_; }
  // A batch of functions with DIFFERENT matrices (MinimizeBatch(functions, states), cppoptlib/mi355/batch_driver.h):
  // objective id MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM, parameters (rows, lambda), per-problem row = A_b then y_b
  static constexpr int kDeviceObjectiveOwnMatrix = MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM;
  std::vector<double> DeviceOwnMatrixParams() const { return {static_cast<double>(rows_), lambda_}; }
  // what every function of an own-matrix batch must agree on: the kernel takes (rows, lambda) from the shared blob
  std::array<double, 3> DeviceOwnMatrixKey() const { return {static_cast<double>(rows_), static_cast<double>(n_), lambda_}; }
  std::vector<double> DeviceOwnMatrixRow() const {
    std::vector<double> r(a_);
    r.insert(r.end(), y_.begin(), y_.end());
    return r;
  }
  // a few entries of the parameters: functions with different matrices differ here with probability ~1
  std::array<double, 6> DeviceFingerprint() const {
    const size_t s = a_.size();
    return {static_cast<double>(rows_), static_cast<double>(n_), lambda_, s ? a_[0] : 0.0, s ? a_[s / 2] : 0.0,
            s ? a_[s - 1] : 0.0};
  }
  // H_jj = sum_i (2 A_ij) A_ij + lambda * 2 (README `hess` of SquaredError / L2Reg, ascending rows)
  std::vector<double> DeviceHessianDiagonal() const {
    std::vector<double> d(static_cast<size_t>(n_));
    for (int j = 0; j < n_; ++j) {
      double acc = (2.0 * a_[static_cast<size_t>(j)]) * a_[static_cast<size_t>(j)];
      for (int i = 1; i < rows_; ++i)
        acc = acc + (2.0 * a_[static_cast<size_t>(i) * n_ + j]) * a_[static_cast<size_t>(i) * n_ + j];
      d[static_cast<size_t>(j)] = acc + lambda_ * 2.0;
    }
    return d;
  }

  // Second-mode call signature (function_base.h:103-120); the Hessian is constant.
  ScalarType operator()(const VectorType& x, VectorType* gradient, MatrixType* hessian) const {
    if (hessian) {
      *hessian = MatrixType(n_, n_);
      for (int j = 0; j < n_; ++j)
        for (int k = 0; k < n_; ++k) {
          double acc = 0;
          for (int i = 0; i < rows_; ++i)
            acc = acc + (2.0 * a_[static_cast<size_t>(i) * n_ + j]) * a_[static_cast<size_t>(i) * n_ + k];
          (*hessian)(j, k) = acc + (j == k ? lambda_ * 2.0 : 0.0);
        }
    }
    return (*this)(x, gradient);
  }

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    std::vector<double> r(static_cast<size_t>(rows_));
    ScalarType f = 0, xx = 0;
    for (int i = 0; i < rows_; ++i) {
      ScalarType acc = 0;
      for (int j = 0; j < n_; ++j) acc += a_[static_cast<size_t>(i) * n_ + j] * x[j];
      r[static_cast<size_t>(i)] = acc - y_[static_cast<size_t>(i)];
      f += r[static_cast<size_t>(i)] * r[static_cast<size_t>(i)];
    }
    for (int j = 0; j < n_; ++j) xx += x[j] * x[j];
    if (gradient) {
      gradient->resize(n_);
      for (int j = 0; j < n_; ++j) {
        ScalarType acc = 0;
        for (int i = 0; i < rows_; ++i) acc += a_[static_cast<size_t>(i) * n_ + j] * r[static_cast<size_t>(i)];
        (*gradient)[j] = 2 * acc + lambda_ * (2 * x[j]);
      }
    }
    return f + lambda_ * xx;
  }

 private:
  int rows_, n_;
  std::vector<double> a_, y_;
  double lambda_;
  uint64_t params_hash_ = 0;
};

// The two README functors on their own (README.md:126-152), so that the example's composition
// `SquaredError(A, y) + lambda * L2Reg(n)` can be written as printed (function_expressions.h maps
// the sum onto SquaredErrorRidge).  Each also has a twin by itself: SquaredError is the ridge
// kernel at lambda = 0, L2Reg the diagonal quadratic with a = 1, c = 0.
template <int TDimension = kDynamicDimension, DifferentiabilityMode TMode = DifferentiabilityMode::Second>
class SquaredError : public FunctionCRTP<SquaredError<TDimension, TMode>, double, TMode, TDimension> {
 public:
  static_assert(TMode != DifferentiabilityMode::None, "SquaredError is differentiable");
  using Super = FunctionCRTP<SquaredError<TDimension, TMode>, double, TMode, TDimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_SQUARED_ERROR_RIDGE;
  static constexpr int kDeviceObjectiveFused = MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM;

  SquaredError(int rows, int n, std::vector<double> a_row_major, std::vector<double> y)
      : ridge_(rows, n, std::move(a_row_major), std::move(y), 0.0), rows_(rows), n_(n) {}
  int rows() const { return rows_; }
  int cols() const { return n_; }
  int GetDimension() const { return n_; }
  std::vector<double> matrix() const {
    const std::vector<double> p = ridge_.DeviceParams();
    return std::vector<double>(p.begin() + 2, p.end());
  }
  std::vector<double> rhs() const { return ridge_.DevicePerProblem(); }
  std::vector<double> DeviceParams() const { return ridge_.DeviceParams(); }
  uint64_t DeviceParamsHash() const { return ridge_.DeviceParamsHash(); }
  std::vector<double> DevicePerProblem() const { return ridge_.DevicePerProblem(); }
  std::vector<double> DeviceHessianDiagonal() const { return ridge_.DeviceHessianDiagonal(); }
  // the run-time record (least-squares shape kept: `+ lambda * L2Reg` resolves to the ridge kernel); defined below
  cppoptlib::mi355::TwinRecord DeviceTwin() const;

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr, MatrixType* hessian = nullptr) const {
    return ridge_(x, gradient, hessian);
  }

 private:
  SquaredErrorRidge<TDimension, DifferentiabilityMode::Second> ridge_;
  int rows_, n_;
};

template <int TDimension = kDynamicDimension, DifferentiabilityMode TMode = DifferentiabilityMode::Second>
class L2Reg : public FunctionCRTP<L2Reg<TDimension, TMode>, double, TMode, TDimension> {
 public:
  static_assert(TMode != DifferentiabilityMode::None, "L2Reg is differentiable");
  using Super = FunctionCRTP<L2Reg<TDimension, TMode>, double, TMode, TDimension>;
  using typename Super::MatrixType;
  using typename Super::ScalarType;
  using typename Super::VectorType;
  static constexpr int kDeviceObjective = MI355_OBJ_DIAG_QUADRATIC;

  explicit L2Reg(int n) : n_(n) {}
  int GetDimension() const { return n_; }
  std::vector<double> DeviceParams() const {
    std::vector<double> p(static_cast<size_t>(n_) + 1, 1.0);
    p.back() = 0.0;
    return p;
  }
  std::vector<double> DeviceHessianDiagonal() const { return std::vector<double>(static_cast<size_t>(n_), 2.0); }
  cppoptlib::mi355::TwinRecord DeviceTwin() const;  // the squared-norm record; defined below

  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr, MatrixType* hessian = nullptr) const {
    ScalarType xx = 0;
    for (int j = 0; j < n_; ++j) xx += x[j] * x[j];
    if (gradient) {
      *gradient = VectorType(n_);
      for (int j = 0; j < n_; ++j) (*gradient)[j] = 2 * x[j];
    }
    if (hessian) {
      *hessian = MatrixType(n_, n_);
      for (int j = 0; j < n_; ++j)
        for (int k = 0; k < n_; ++k) (*hessian)(j, k) = (j == k) ? 2.0 : 0.0;
    }
    return xx;
  }

 private:
  int n_;
};

}  // namespace cppoptlib::function

// ---------------------------------------------------------------------------------------------------------------------
// The `twin::` builders: run-time records (cppoptlib/mi355/device_twin.h) of the shapes the device has kernels for.  A
// user functor names its twin in ONE line,
//     auto DeviceTwin() const { return cppoptlib::mi355::twin::LeastSquares(A, y); }
// and keeps its own operator() for the host; records compose as the functions do (`twin::Coordinate(i) - lower`,
// `LeastSquares + lambda * SquaredNorm`).  Every builder is the record of one of the library functors above, so the
// device runs for a user functor exactly what it runs for them.
// ---------------------------------------------------------------------------------------------------------------------
namespace cppoptlib::mi355 {

// ||A x - y||^2 + lambda ||x||^2: the ridge kernels (MI355_OBJ_SQUARED_ERROR_RIDGE / _GRAM / _OWN_GRAM), Hessian
// diagonal for Second-mode wrappers included.  With lambda == 0 the record also keeps the least-squares shape, so a
// later `+ lambda * SquaredNorm` resolves to the ridge kernel, and it is a TERM: the sum of one squared affine
// primitive per row, r_i^2 with r_i = a_i . x - y_i (MI355_AL_TERM_SQUARED_AFFINE), left to right.
inline TwinRecord RidgeRecord(std::shared_ptr<const LeastSquaresShape> shape, double lambda) {
  using Ridge = cppoptlib::function::SquaredErrorRidge<cppoptlib::function::kDynamicDimension,
                                                       cppoptlib::function::DifferentiabilityMode::Second>;
  TwinRecord r;
  r.objective = ObjectiveOfStatic<Ridge>(
      std::make_shared<const Ridge>(shape->rows, shape->n, shape->a_row_major, shape->y, lambda));
  if (lambda == 0.0) {
    r.least_squares = shape;
    if (shape->rows <= MI355_AL_MAX_ROWS) {
      r.term.valid = true;
      for (int i = 0; i < shape->rows; ++i) {
        r.term.prims.kinds.push_back(MI355_AL_TERM_SQUARED_AFFINE);
        r.term.prims.rows.push_back([shape, i](int n) {
          if (n != shape->n) return std::vector<double>();
          std::vector<double> row(shape->a_row_major.begin() + static_cast<std::ptrdiff_t>(i) * n,
                                  shape->a_row_major.begin() + static_cast<std::ptrdiff_t>(i + 1) * n);
          row.push_back(shape->y[static_cast<size_t>(i)]);
          return row;
        });
      }
    } else {
      r.why_no_term = "a least-squares function with more than MI355_AL_MAX_ROWS residuals is not a term of the device menu";
    }
  } else {
    r.why_no_term = "the ridge function is an objective of Lbfgs / Lbfgsb / Bfgs, not a term of the device menu";
  }
  return r;
}

// `f + g` of two records: the term rule of device_twin.h, and the README's ridge composition (README.md:159:
// `SquaredError(A, y) + lambda * L2Reg(n)` — value r.r + lambda (x.x), gradient 2 A^T r + lambda (2 x), Hessian
// diagonal (2 A^T A)_jj + lambda 2: term by term what AddExpression / MulExpression produce from the two operands).
inline TwinRecord SumRecord(const TwinRecord& f, const TwinRecord& g) {
  TwinRecord r;
  if (f.least_squares && g.squared_norm) {
    r = RidgeRecord(f.least_squares, g.squared_norm_scale);
  } else {
    r.why_no_objective = "a sum has a device objective as `LeastSquares + lambda * SquaredNorm` (the ridge kernel) or as a left-nested sum of primitives of the device menu";
  }
  r.term = SumTerm(f, g, &r.why_no_term);
  ObjectiveFromTerm(&r);     // any other sum of menu primitives: the composite with that term as its objective
  return r;
}

// How a user functor's DeviceTwin() writes compositions: the operators of the reference's expression layer, on records.
inline TwinRecord operator-(const TwinRecord& f, double k) { return OffsetRecord(f, k, false); }
inline TwinRecord operator-(double k, const TwinRecord& f) { return OffsetRecord(f, k, true); }
inline TwinRecord operator*(double c, const TwinRecord& f) { return ScaledRecord(c, f); }
inline TwinRecord operator*(const TwinRecord& f, double c) { return ScaledRecord(c, f); }
inline TwinRecord operator-(const TwinRecord& f) { return ScaledRecord(-1.0, f); }
inline TwinRecord operator+(const TwinRecord& f, const TwinRecord& g) { return SumRecord(f, g); }
inline TwinRecord operator*(const TwinRecord& f, const TwinRecord& g) { return ProductRecord(f, g); }

namespace twin {

// chained Rosenbrock-N (MI355_OBJ_ROSENBROCK / MI355_AL_TERM_ROSENBROCK)
inline TwinRecord Rosenbrock() { return RecordOfFunction(cppoptlib::function::Rosenbrock<>()); }

// sum_i a_i x_i^2 + c (MI355_OBJ_DIAG_QUADRATIC / MI355_AL_TERM_DIAG_QUADRATIC); constant Hessian diag = 2 a
inline TwinRecord DiagQuadratic(std::vector<double> a, double c) {
  TwinRecord r = RecordOfFunction(cppoptlib::function::DiagQuadratic<>(a, c));
  r.objective.hessian_diagonal = [a](int n) {
    std::vector<double> d(a);
    for (double& v : d) v = 2.0 * v;
    if (static_cast<int>(d.size()) != n) d.clear();
    return d;
  };
  return r;
}

// a . x (MI355_AL_TERM_LINEAR): a term only — a linear function has no unconstrained minimiser to ask Lbfgs for
inline TwinRecord Linear(std::vector<double> a) { return RecordOfFunction(cppoptlib::function::LinearForm<>(std::move(a))); }
// x_i, for any dimension
inline TwinRecord Coordinate(int index) {
  TwinRecord r;
  r.term.valid = true;
  r.term.prims.kinds.push_back(MI355_AL_TERM_LINEAR);
  r.term.prims.rows.push_back([index](int n) {
    if (index < 0 || index >= n) return std::vector<double>();
    std::vector<double> row(static_cast<size_t>(n) + 1, 0.0);
    row[static_cast<size_t>(index)] = 1.0;
    return row;
  });
  r.why_no_objective = "a linear function is a term of a constrained problem, not an unconstrained objective";
  return r;
}
// x.sum(), for any dimension
inline TwinRecord CoordinateSum() {
  TwinRecord r;
  r.term.valid = true;
  r.term.prims.kinds.push_back(MI355_AL_TERM_LINEAR);
  r.term.prims.rows.push_back([](int n) {
    std::vector<double> row(static_cast<size_t>(n) + 1, 1.0);
    row.back() = 0.0;
    return row;
  });
  r.why_no_objective = "a linear function is a term of a constrained problem, not an unconstrained objective";
  return r;
}

// x.squaredNorm(), for any dimension: MI355_AL_TERM_SQUARED_NORM as a term, the unit diagonal quadratic as an
// objective (constant Hessian 2 I), and the regulariser of the ridge composition
inline TwinRecord SquaredNorm() {
  TwinRecord r = RecordOfFunction(cppoptlib::function::SquaredNorm<>());
  r.objective.valid = true;
  r.objective.id = MI355_OBJ_DIAG_QUADRATIC;
  r.objective.params = [](int n) {
    std::vector<double> p(static_cast<size_t>(n) + 1, 1.0);
    p.back() = 0.0;
    return p;
  };
  r.objective.hessian_diagonal = [](int n) { return std::vector<double>(static_cast<size_t>(n), 2.0); };
  r.squared_norm = true;
  return r;
}

// ||A x - y||^2 with A rows x n, row major
inline TwinRecord LeastSquares(int rows, std::vector<double> a_row_major, std::vector<double> y) {
  auto shape = std::make_shared<LeastSquaresShape>();
  shape->rows = rows;
  shape->n = rows > 0 ? static_cast<int>(a_row_major.size()) / rows : 0;
  shape->a_row_major = std::move(a_row_major);
  shape->y = std::move(y);
  if (rows <= 0 || static_cast<int>(shape->y.size()) != rows ||
      static_cast<size_t>(shape->n) * static_cast<size_t>(rows) != shape->a_row_major.size())
    Fail("twin::LeastSquares: A must hold rows x n entries and y one per row");
  return RidgeRecord(shape, 0.0);
}
// ... from any dense matrix / vector pair with rows(), cols(), (i, j) and [i] (Eigen::MatrixXd, Eigen::VectorXd)
template <class Matrix, class Vector, class = decltype(std::declval<const Matrix&>().cols())>
TwinRecord LeastSquares(const Matrix& A, const Vector& y) {
  const int rows = static_cast<int>(A.rows()), n = static_cast<int>(A.cols());
  std::vector<double> a(static_cast<size_t>(rows) * static_cast<size_t>(n)), rhs(static_cast<size_t>(rows));
  for (int i = 0; i < rows; ++i) {
    for (int j = 0; j < n; ++j) a[static_cast<size_t>(i) * n + j] = static_cast<double>(A(i, j));
    rhs[static_cast<size_t>(i)] = static_cast<double>(y[i]);
  }
  return LeastSquares(rows, std::move(a), std::move(rhs));
}

// Coefficient vectors for the builders above, written in one expression inside a DeviceTwin() line:
//   Fill(n, first, count, value)     value on [first, first + count), 0 elsewhere
//   Sparse(n).Set(i, v).Set(first, vector)...   single entries and runs copied from anything with size() and [i]
// (the functors of src/examples/svm_primal_al.cc: `0.5 ||w||^2 + C sum(xi)` on a (w, b, xi) layout, and the margin
//  constraint `y_i (w . x_i + b) - 1 + xi_i` whose coefficients are y_i x_i, y_i and a unit entry)
inline std::vector<double> Fill(int n, int first, int count, double value) {
  std::vector<double> v(static_cast<size_t>(n > 0 ? n : 0), 0.0);
  for (int i = first; i < first + count && i < n; ++i)
    if (i >= 0) v[static_cast<size_t>(i)] = value;
  return v;
}
class Sparse {
 public:
  explicit Sparse(int n) : v_(static_cast<size_t>(n > 0 ? n : 0), 0.0) {}
  Sparse& Set(int index, double value) {
    if (index < 0 || static_cast<size_t>(index) >= v_.size()) Fail("twin::Sparse::Set: index outside the vector");
    v_[static_cast<size_t>(index)] = value;
    return *this;
  }
  template <class Vector, class = decltype(std::declval<const Vector&>().size())>
  Sparse& Set(int first, const Vector& values) {
    const int count = static_cast<int>(values.size());
    if (first < 0 || static_cast<size_t>(first) + static_cast<size_t>(count) > v_.size())
      Fail("twin::Sparse::Set: run outside the vector");
    for (int i = 0; i < count; ++i) v_[static_cast<size_t>(first + i)] = static_cast<double>(values[i]);
    return *this;
  }
  operator std::vector<double>() const { return v_; }  // NOLINT: the builders take std::vector<double>

 private:
  std::vector<double> v_;
};

// ---- user twins: device functors compiled into a build of the library ----------------------
// Pack(...) flattens what a functor holds into the parameter blob its device functor's load() reads, left to right:
// arithmetic values as one double each, vectors (size(), [i]) coefficient by coefficient, matrices (rows(), cols(), (i, j))
// ROW major — `Pack(features.rows(), features.cols(), c, features, labels)` is the blob (N, d, C, X, y) of the squared-hinge
// SVM functor of src/examples/svm_primal_lbfgs.cc.
inline void PackInto(std::vector<double>*) {}
template <class First, class... Rest>
void PackInto(std::vector<double>* out, const First& first, const Rest&... rest) {
  if constexpr (std::is_arithmetic<First>::value) {
    out->push_back(static_cast<double>(first));
  } else if constexpr (std::is_same<First, std::vector<double>>::value) {
    out->insert(out->end(), first.begin(), first.end());
  } else {
    const auto rows = first.rows(), cols = first.cols();
    if (cols == 1) {
      for (decltype(first.rows()) i = 0; i < rows; ++i) out->push_back(static_cast<double>(first[i]));
    } else {
      for (decltype(first.rows()) i = 0; i < rows; ++i)
        for (decltype(first.cols()) j = 0; j < cols; ++j) out->push_back(static_cast<double>(first(i, j)));
    }
  }
  PackInto(out, rest...);
}
template <class... Parts>
std::vector<double> Pack(const Parts&... parts) {
  std::vector<double> blob;
  PackInto(&blob, parts...);
  return blob;
}
// the functor registered under objective id `id` (>= MI355_OBJ_USER_FIRST) when the library was built, with its blob:
// an unconstrained objective of Lbfgs / Lbfgsb / Bfgs
inline TwinRecord UserObjective(int id, std::vector<double> blob) {
  if (id < MI355_OBJ_USER_FIRST) Fail("twin::UserObjective: ids of user objectives start at MI355_OBJ_USER_FIRST");
  TwinRecord r;
  auto shared = std::make_shared<const std::vector<double>>(std::move(blob));
  r.objective.valid = true;
  r.objective.id = id;
  r.objective.params = [shared](int) { return *shared; };
  r.objective.params_hash = [shared]() { return HashDoubles(shared->data(), shared->size()); };
  r.why_no_term = "this user twin was registered as an objective; as a term of a constrained problem name its term kind "
                  "(twin::UserTerm)";
  return r;
}
// ... and the functor registered as TERM kind `kind` (>= MI355_AL_TERM_USER) that takes the problem's parameter blob
// (kTermParamsFromProblem: the dense dual SVM of src/examples/svm_dual_al.cc): a term of a constrained problem
inline TwinRecord UserTerm(int kind, std::vector<double> blob) {
  if (kind < MI355_AL_TERM_USER) Fail("twin::UserTerm: kinds of user terms start at MI355_AL_TERM_USER");
  TwinRecord r;
  auto shared = std::make_shared<const std::vector<double>>(std::move(blob));
  r.term.valid = true;
  r.term.prims.kinds.push_back(kind);
  r.term.prims.rows.push_back([](int n) { return std::vector<double>(static_cast<size_t>(n) + 1, 0.0); });
  r.term.prims.user_params.push_back([shared]() { return *shared; });
  r.why_no_objective = "this user twin was registered as a term kind; as an unconstrained objective name its objective id "
                       "(twin::UserObjective)";
  return r;
}

}  // namespace twin
}  // namespace cppoptlib::mi355

namespace cppoptlib::function {
template <int TDimension, DifferentiabilityMode TMode>
cppoptlib::mi355::TwinRecord SquaredError<TDimension, TMode>::DeviceTwin() const {
  return cppoptlib::mi355::twin::LeastSquares(rows_, matrix(), rhs());
}
template <int TDimension, DifferentiabilityMode TMode>
cppoptlib::mi355::TwinRecord L2Reg<TDimension, TMode>::DeviceTwin() const {
  return cppoptlib::mi355::twin::SquaredNorm();
}
}  // namespace cppoptlib::function
#endif  // CPPOPTLIB_MI355_OBJECTIVES_H_
