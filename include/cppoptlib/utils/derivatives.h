// cppoptlib/utils/derivatives.h — finite-difference checks of a functor's derivatives (host side).
//
// Same four entry points as the reference's include/cppoptlib/utils/derivatives.h (ComputeFiniteGradient :44-88,
// ComputeFiniteHessian :90-240, IsGradientCorrect :243-269, IsHessianCorrect :271-301; used by
// src/examples/simple.cc:68-69 on a FunctionExpr): central differences of order 2, 4, 6, 8 selected by `accuracy`
// 0..3, step h = sqrt(eps) max(|x_d|, 1) per coordinate, and the two checkers with the reference's tolerances
// (1e-2 and 1e-1 relative to max(1, |actual|, |expected|)).  A user calls them before handing a function to a solver,
// i.e. before anything reaches the GPU: they evaluate the HOST operator() of the function (or of whatever a type-erased
// FunctionExpr wraps) and are not part of the device path.
#ifndef INCLUDE_CPPOPTLIB_UTILS_DERIVATIVES_H_
#define INCLUDE_CPPOPTLIB_UTILS_DERIVATIVES_H_

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <limits>

#include "../function_base.h"

namespace cppoptlib::utils {

namespace detail {
// Central first-derivative stencils: offsets -k..-1, 1..k in units of h, integer weights over a common denominator.
struct Stencil {
  int points;
  int offset[8];
  int weight[8];
  int denominator;
};
inline const Stencil& FirstDerivativeStencil(int accuracy) {
  static const Stencil kStencils[4] = {
      {2, {1, -1}, {1, -1}, 2},
      {4, {-2, -1, 1, 2}, {1, -8, 8, -1}, 12},
      {6, {-3, -2, -1, 1, 2, 3}, {-1, 9, -45, 45, -9, 1}, 60},
      {8, {-4, -3, -2, -1, 1, 2, 3, 4}, {3, -32, 168, -672, 672, -168, 32, -3}, 840}};
  return kStencils[std::min(std::max(accuracy, 0), 3)];
}
template <class Scalar>
Scalar StepFor(Scalar coordinate) {
  return std::sqrt(std::numeric_limits<Scalar>::epsilon()) * std::max(std::abs(coordinate), Scalar(1));
}
}  // namespace detail

template <class FunctionType>
void ComputeFiniteGradient(
    const FunctionType& function,
    const cppoptlib::mi355::Vector<typename FunctionType::ScalarType, cppoptlib::function::kDynamicDimension>& x0,
    cppoptlib::mi355::Vector<typename FunctionType::ScalarType, cppoptlib::function::kDynamicDimension>* grad,
    const int accuracy = 0) {
  using Scalar = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  const detail::Stencil& stencil = detail::FirstDerivativeStencil(accuracy);
  const std::ptrdiff_t n = x0.size();
  grad->resize(n);
  VectorType x = x0;
  for (std::ptrdiff_t d = 0; d < n; ++d) {
    const Scalar h = detail::StepFor<Scalar>(x0[d]);
    const Scalar keep = x[d];
    Scalar sum = 0;
    for (int s = 0; s < stencil.points; ++s) {
      x[d] = keep;
      x[d] += Scalar(stencil.offset[s]) * h;
      sum += Scalar(stencil.weight[s]) * function(x);
    }
    x[d] = keep;
    (*grad)[d] = sum / (Scalar(stencil.denominator) * h);
  }
}

template <class FunctionType>
void ComputeFiniteHessian(
    const FunctionType& function,
    const cppoptlib::mi355::Vector<typename FunctionType::ScalarType, cppoptlib::function::kDynamicDimension>& x0,
    cppoptlib::mi355::SquareMatrix<typename FunctionType::ScalarType, cppoptlib::function::kDynamicDimension>* hessian,
    int accuracy = 0) {
  using Scalar = typename FunctionType::ScalarType;
  using VectorType = cppoptlib::mi355::Vector<Scalar, cppoptlib::function::kDynamicDimension>;
  using MatrixType = cppoptlib::mi355::SquareMatrix<Scalar, cppoptlib::function::kDynamicDimension>;
  const std::ptrdiff_t n = x0.size();
  *hessian = MatrixType(n, n);
  const Scalar f0 = function(x0);
  // f at x0 + a e_i + b e_j
  auto shifted = [&](std::ptrdiff_t i, Scalar a, std::ptrdiff_t j, Scalar b) {
    VectorType x = x0;
    x[i] = x0[i] + a;
    if (j >= 0) x[j] = x0[j] + b;
    return function(x);
  };
  for (std::ptrdiff_t i = 0; i < n; ++i) {
    const Scalar hi = detail::StepFor<Scalar>(x0[i]);
    (*hessian)(i, i) = (shifted(i, hi, -1, 0) - 2 * f0 + shifted(i, -hi, -1, 0)) / (hi * hi);
    for (std::ptrdiff_t j = i + 1; j < n; ++j) {
      const Scalar hj = detail::StepFor<Scalar>(x0[j]);
      Scalar mixed;
      if (accuracy == 0) {
        // the four corners of the (hi, hj) rectangle
        mixed = (shifted(i, hi, j, hj) - shifted(i, hi, j, -hj) - shifted(i, -hi, j, hj) + shifted(i, -hi, j, -hj)) /
                (4 * hi * hj);
      } else {
        // fourth-order mixed partial on the 5 x 5 grid of spacing h (16 points in four symmetry classes)
        const Scalar h = (hi + hj) / 2;
        const Scalar knight_minus = shifted(i, h, j, -2 * h) + shifted(i, 2 * h, j, -h) + shifted(i, -2 * h, j, h) +
                                    shifted(i, -h, j, 2 * h);
        const Scalar knight_plus = shifted(i, -h, j, -2 * h) + shifted(i, -2 * h, j, -h) + shifted(i, h, j, 2 * h) +
                                   shifted(i, 2 * h, j, h);
        const Scalar far_corners = shifted(i, 2 * h, j, -2 * h) + shifted(i, -2 * h, j, 2 * h) -
                                   shifted(i, -2 * h, j, -2 * h) - shifted(i, 2 * h, j, 2 * h);
        const Scalar near_corners = shifted(i, -h, j, -h) + shifted(i, h, j, h) - shifted(i, h, j, -h) - shifted(i, -h, j, h);
        mixed = (-63 * knight_minus + 63 * knight_plus + 44 * far_corners + 74 * near_corners) / (600 * h * h);
      }
      (*hessian)(i, j) = mixed;
      (*hessian)(j, i) = mixed;
    }
  }
}

// |actual - expected| <= tolerance * max(1, |actual|, |expected|) in every coordinate (tolerance 1e-2)
template <class FunctionType>
bool IsGradientCorrect(const FunctionType& function, const typename FunctionType::VectorType& x0, int accuracy = 3) {
  using Scalar = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  constexpr float tolerance = 1e-2f;
  const std::ptrdiff_t n = x0.size();
  VectorType actual;
  function(x0, &actual);
  VectorType expected(n);
  ComputeFiniteGradient(function, x0, &expected, accuracy);
  for (std::ptrdiff_t d = 0; d < n; ++d) {
    const Scalar scale = std::max(std::max(std::abs(actual[d]), std::abs(expected[d])), Scalar(1));
    if (std::abs(actual[d] - expected[d]) > tolerance * scale) return false;
  }
  return true;
}

// the same test over every entry of the Hessian (tolerance 1e-1)
template <class FunctionType>
bool IsHessianCorrect(const FunctionType& function, const typename FunctionType::VectorType& x0, int accuracy = 3) {
  using Scalar = typename FunctionType::ScalarType;
  using MatrixType = typename FunctionType::MatrixType;
  constexpr float tolerance = 1e-1f;
  const std::ptrdiff_t n = x0.size();
  MatrixType actual;
  function(x0, nullptr, &actual);
  MatrixType expected(n, n);
  ComputeFiniteHessian(function, x0, &expected, accuracy);
  for (std::ptrdiff_t d = 0; d < n; ++d)
    for (std::ptrdiff_t e = 0; e < n; ++e) {
      const Scalar scale = std::max(std::max(std::abs(actual(d, e)), std::abs(expected(d, e))), Scalar(1));
      if (std::abs(actual(d, e) - expected(d, e)) > tolerance * scale) return false;
    }
  return true;
}

}  // namespace cppoptlib::utils
#endif  // INCLUDE_CPPOPTLIB_UTILS_DERIVATIVES_H_
