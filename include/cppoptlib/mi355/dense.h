// cppoptlib/mi355/dense.h — vector type used by the cppoptlib-shaped host API.
//
// With Eigen on the include path the API uses the very types the reference
// uses (Eigen::Matrix<Scalar, Dim, 1>), so user code written against
// PatWie/CppNumericalSolvers compiles unchanged.  Without Eigen (e.g. the image
// this engine is developed in) a small self-contained vector with the same
// construction / indexing surface is used instead; it only has to carry data
// across the C-ABI, all arithmetic happens on the GPU.
#ifndef CPPOPTLIB_MI355_DENSE_H_
#define CPPOPTLIB_MI355_DENSE_H_

#include <cstddef>
#include <initializer_list>
#include <ostream>
#include <vector>

#if defined(__has_include)
#if __has_include(<Eigen/Core>) && !defined(CPPOPTLIB_MI355_NO_EIGEN)
#define CPPOPTLIB_MI355_HAVE_EIGEN 1
#include <Eigen/Core>
#endif
#endif

namespace cppoptlib::mi355 {

#ifdef CPPOPTLIB_MI355_HAVE_EIGEN
constexpr int kDynamic = Eigen::Dynamic;
template <class T, int Dim>
using Vector = Eigen::Matrix<T, Dim, 1>;
template <class T, int Dim>
using SquareMatrix = Eigen::Matrix<T, Dim, Dim>;
#else
constexpr int kDynamic = -1;

template <class T, int Dim>
class Vector {
 public:
  using Scalar = T;
  Vector() : v_(Dim > 0 ? Dim : 0, T(0)) {}
  explicit Vector(std::ptrdiff_t n) : v_(static_cast<size_t>(n), T(0)) {}
  Vector(std::initializer_list<T> il) : v_(il) {}
  // (x, y) / (x, y, z) coefficient construction for small fixed sizes
  Vector(T a, T b) : v_{a, b} { static_assert(Dim == 2 || Dim == kDynamic, "2 coefficients"); }
  Vector(T a, T b, T c) : v_{a, b, c} { static_assert(Dim == 3 || Dim == kDynamic, "3 coefficients"); }
  static Vector Zero(std::ptrdiff_t n = (Dim > 0 ? Dim : 0)) { return Vector(n); }
  static Vector Constant(std::ptrdiff_t n, T value) {
    Vector r(n);
    for (auto& e : r.v_) e = value;
    return r;
  }
  std::ptrdiff_t size() const { return static_cast<std::ptrdiff_t>(v_.size()); }
  std::ptrdiff_t rows() const { return size(); }
  void resize(std::ptrdiff_t n) { v_.assign(static_cast<size_t>(n), T(0)); }
  T& operator[](std::ptrdiff_t i) { return v_[static_cast<size_t>(i)]; }
  const T& operator[](std::ptrdiff_t i) const { return v_[static_cast<size_t>(i)]; }
  T& operator()(std::ptrdiff_t i) { return v_[static_cast<size_t>(i)]; }
  const T& operator()(std::ptrdiff_t i) const { return v_[static_cast<size_t>(i)]; }
  T* data() { return v_.data(); }
  const T* data() const { return v_.data(); }
  friend std::ostream& operator<<(std::ostream& os, const Vector& x) {
    for (std::ptrdiff_t i = 0; i < x.size(); ++i) os << (i ? " " : "") << x[i];
    return os;
  }

 private:
  std::vector<T> v_;
};

// Hessians never cross the C-ABI; the type only has to exist for signatures.
template <class T, int Dim>
class SquareMatrix {
 public:
  SquareMatrix() = default;
  SquareMatrix(std::ptrdiff_t r, std::ptrdiff_t c) : r_(r), v_(static_cast<size_t>(r * c), T(0)) {}
  T& operator()(std::ptrdiff_t i, std::ptrdiff_t j) { return v_[static_cast<size_t>(j * r_ + i)]; }
  const T& operator()(std::ptrdiff_t i, std::ptrdiff_t j) const { return v_[static_cast<size_t>(j * r_ + i)]; }
  std::ptrdiff_t rows() const { return r_; }

 private:
  std::ptrdiff_t r_ = 0;
  std::vector<T> v_;
};
#endif

}  // namespace cppoptlib::mi355
#endif  // CPPOPTLIB_MI355_DENSE_H_
