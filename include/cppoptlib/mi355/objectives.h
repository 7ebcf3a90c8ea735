// cppoptlib/mi355/objectives.h — objectives that have a device twin.
//
// The reference takes an arbitrary host functor (FunctionCRTP::operator(),
// function_base.h:103-120).  A GPU engine needs the objective as device code, so
// a function type advertises its twin through two members:
//     static constexpr int kDeviceObjective;        // an mi355_objective id
//     std::vector<double> DeviceParams() const;     // shared parameter blob
// The classes below are ordinary FunctionCRTP functors (operator() works on the
// host, same operation order as the device code) that carry those members.
// Lbfgs<F>::Minimize refuses at compile time a function type without a twin —
// there is no CPU fallback.
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

namespace cppoptlib::mi355 {

// FNV-1a over the bit patterns of `count` doubles, chained through `seed`: the parameter-blob hash a function type
// with a large blob computes once at construction (`uint64_t DeviceParamsHash() const`), so that a batch of B functions
// is checked for shared parameters with B integer comparisons.
inline uint64_t HashDoubles(const double* data, size_t count, uint64_t seed = 1469598103934665603ull) {
  uint64_t h = seed;
  for (size_t i = 0; i < count; ++i) {
    uint64_t bits;
    std::memcpy(&bits, data + i, sizeof bits);
    h = (h ^ bits) * 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}

template <class F, class = void>
struct HasDeviceParamsHash : std::false_type {};
template <class F>
struct HasDeviceParamsHash<F, std::void_t<decltype(std::declval<const F&>().DeviceParamsHash())>> : std::true_type {};

template <class F, class = void>
struct HasDeviceParamsOfDimension : std::false_type {};
template <class F>
struct HasDeviceParamsOfDimension<F, std::void_t<decltype(std::declval<const F&>().DeviceParams(1))>> : std::true_type {};

template <class F, class = void>
struct HasDeviceObjective : std::false_type {};
template <class F>
struct HasDeviceObjective<F, std::void_t<decltype(F::kDeviceObjective),
                                         decltype(std::declval<const F&>().DeviceParams())>>
    : std::true_type {};
template <class F>
struct HasDeviceObjective<F, std::enable_if_t<HasDeviceParamsOfDimension<F>::value, std::void_t<decltype(F::kDeviceObjective)>>>
    : std::true_type {};

// A function type may name a second device twin that evaluates the same function in a fused / re-associated form
// A Second-mode function whose Hessian is not constant says so with `static constexpr bool kDeviceHessianFromFunctor =
// true`: its device functor has a hess_diag and Lbfgs asks the kernel to rebuild the preconditioner at every iterate
// (mi355_lbfgs_desc::hessian_from_functor) instead of uploading DeviceHessianDiagonal() once.
template <class F, class = void>
struct HessianFromFunctor : std::false_type {};
template <class F>
struct HessianFromFunctor<F, std::void_t<decltype(F::kDeviceHessianFromFunctor)>>
    : std::integral_constant<bool, F::kDeviceHessianFromFunctor> {};

// (`static constexpr int kDeviceObjectiveFused`): the solvers take it when the caller asks for MI355_ARITH_FMA
// (SetArithmetic) and the reference-order twin otherwise.  For the ridge functors that is the normal-equation form
// (MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM: x*, f* within 1e-6 of the reference, ~4 x the throughput of id 2).
template <class F, class = void>
struct FusedDeviceObjective {
  static constexpr int Of(int /*arithmetic*/) { return F::kDeviceObjective; }
};
template <class F>
struct FusedDeviceObjective<F, std::void_t<decltype(F::kDeviceObjectiveFused)>> {
  static constexpr int Of(int arithmetic) {
    return arithmetic == MI355_ARITH_FMA ? F::kDeviceObjectiveFused : F::kDeviceObjective;
  }
};

// Objectives whose device twin needs data per problem (e.g. the right-hand side y) expose
//     std::vector<double> DevicePerProblem() const;
template <class F, class = void>
struct HasPerProblemData : std::false_type {};
template <class F>
struct HasPerProblemData<F, std::void_t<decltype(std::declval<const F&>().DevicePerProblem())>> : std::true_type {};

}  // namespace cppoptlib::mi355

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
  // Rigorous upper bound of cond(A^T A + lambda I): Gershgorin row sums of G over lambda (lambda_min(G) >= lambda).
  // O(rows n^2): the solvers evaluate it on a SAMPLE of a batch before taking the normal-equation form by default.
  double NormalEquationConditionBound() const {
    if (!(lambda_ > 0)) return std::numeric_limits<double>::infinity();
    double worst = 0;
    for (int j = 0; j < n_; ++j) {
      double row_sum = 0;
      for (int k = 0; k < n_; ++k) {
        double acc = 0;
        for (int i = 0; i < rows_; ++i) acc += a_[static_cast<size_t>(i) * n_ + j] * a_[static_cast<size_t>(i) * n_ + k];
        row_sum += std::fabs(acc + (j == k ? lambda_ : 0.0));
      }
      worst = std::max(worst, row_sum);
    }
    return worst / lambda_;
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
#endif  // CPPOPTLIB_MI355_OBJECTIVES_H_
