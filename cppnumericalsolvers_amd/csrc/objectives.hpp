// objectives.hpp — device objective functors.
//
// The reference objective is an arbitrary host functor reached through a
// virtual call (FunctionCRTP::operator(), function_base.h:103-120).  On the
// device the objective is a compile-time functor evaluated by the W lanes of a
// problem's segment: lane `sl` owns the E contiguous coordinates
// j = sl*E + e, coordinates j >= n are padding (x = g = 0).
//
// eval(x, g) returns the objective value (segment-uniform) and writes the
// gradient.  Operation order is fixed (no FMA contraction: the library is
// built with -ffp-contract=off) so values are bit-reproducible.
#pragma once
#include "wave_primitives.hpp"

namespace mi355 {

// Chained Rosenbrock-N:
//   f(x) = sum_{i=0}^{N-2} (1-x_i)^2 + 100 (x_{i+1} - x_i^2)^2
// the N-dimensional form that equals the reference's 2-D test functor
// (src/test/verify.cc:58-69) at N = 2, including its operation order:
//   t1 = 1-x0; t2 = x1-x0*x0; f = t1*t1 + 100*t2*t2
//   g0 = -2*(1-x0) + 200*(x1-x0*x0)*(-2*x0);  g1 = 200*(x1-x0*x0)
struct RosenbrockObjective {
  static constexpr int kParams = 0;
  __device__ __forceinline__ void load(const double*, int, int) {}

  template <int W, int E>
  __device__ __forceinline__ double eval(const double (&x)[E], double (&g)[E], int n, int sl) const {
    // x_{j+1}: next element in-lane, or element 0 of the next lane.
    const double x_next_lane = from_next_lane(x[0]);
    double t2[E], term[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      const double xn = (e + 1 < E) ? x[(e + 1 < E) ? e + 1 : e] : x_next_lane;
      const double t1 = 1.0 - x[e];
      t2[e] = xn - x[e] * x[e];
      const double v = t1 * t1 + (100.0 * t2[e]) * t2[e];
      term[e] = (j + 1 < n) ? v : 0.0;
    }
    // t2_{j-1}: previous element in-lane, or element E-1 of the previous lane.
    const double t2_prev_lane = from_prev_lane(t2[E - 1]);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      const bool has_a = (j + 1 < n);
      const bool has_b = (j > 0) && (j < n);
      const double a = -2.0 * (1.0 - x[e]) + (200.0 * t2[e]) * (-2.0 * x[e]);
      const double b = 200.0 * ((e > 0) ? t2[(e > 0) ? e - 1 : 0] : t2_prev_lane);
      g[e] = (has_a && has_b) ? (a + b) : (has_a ? a : (has_b ? b : 0.0));
    }
    return seg_sum<W>(lane_tree_sum<E>(term));
  }
};

// f(x) = sum_i a_i x_i^2 + c  with the README quick-start operation order
// (README.md:21-28): term_i = (a_i*x_i)*x_i, g_i = (2 a_i)*x_i, f = sum + c.
template <int E>
struct DiagQuadraticObjective {
  static constexpr int kParams = -1;  // n + 1
  double a[E];
  double c;
  // params: device pointer to a[0..n), c
  __device__ __forceinline__ void load(const double* params, int n, int sl) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      a[e] = (j < n) ? params[j] : 0.0;
    }
    c = params[n];
  }
  template <int W, int EE>
  __device__ __forceinline__ double eval(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(EE == E, "E");
    double term[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      term[e] = (j < n) ? (a[e] * x[e]) * x[e] : 0.0;
      g[e] = (j < n) ? (2.0 * a[e]) * x[e] : 0.0;
    }
    return seg_sum<W>(lane_tree_sum<E>(term)) + c;
  }
};

}  // namespace mi355
