// svm_dual.hpp — the second worked USER device objective: the dual soft-margin SVM of the reference's
// src/examples/svm_dual_lbfgsb.cc:36-77,
//     min_alpha  0.5 alpha^T Q alpha - 1^T alpha     subject to  0 <= alpha_i <= C,
// with the precomputed kernel-with-labels matrix Q_ij = y_i y_j (x_i . x_j); the box is handled natively by Lbfgsb
// (the example runs `Lbfgsb<SvmDualObjective>` — history size 5 — from alpha = 0 on the 100 Iris samples).
//
// Interface: that of cppnumericalsolvers_amd/csrc/objectives.hpp (see examples/user_objective_svm/svm_squared_hinge.hpp
// for the description).  params = [n, Q (n x n, row major)]; Q must be symmetric to the bit (the reference builds it as
// (X X^T) .* (y y^T), whose (i, j) and (j, i) entries are the same products in the same order): lane `sl` owns the
// coordinates i = sl * E + e and walks COLUMN j of Q for them — consecutive lanes, consecutive addresses — where the
// reference walks row i.
//
// Operation order (what the CPU twin oracle::SvmDual and the restated reference functor in oracle/ref_capi.cpp follow):
//   q_i   = ((Q_i0 a_0 + Q_i1 a_1) + ...) + Q_i,n-1 a_n-1     `kernel_matrix * alpha`, ascending columns
//   value = 0.5 (alpha . q) - sum(alpha)                      the two reductions follow the engine's policy
//   g_i   = q_i - 1
// eval_fma (MI355_ARITH_FMA / the relaxed-algebra Lbfgsb kernels): q_i as one chain of fused multiply-adds.
#pragma once

namespace user_examples {

constexpr int kSvmDualMaxSamples = 128;

template <int E>
struct SvmDual {
  static constexpr int kLdsDoubles = kSvmDualMaxSamples;  // alpha of the current evaluation, all coordinates
  // as an augmented-Lagrangian TERM (the objective of the reference's src/examples/svm_dual_al.cc:36-60) the functor
  // takes the same [n, Q] blob, from mi355_al_problem::user_params (csrc/auglag_device.hpp)
  static constexpr bool kTermParamsFromProblem = true;
  __host__ __device__ static constexpr int shared_lds_doubles() { return 0; }

  const double* Q;
  int ns;
  double* as;  // LDS

  __device__ __forceinline__ void load(const double* params, int, int, double* lds_problem, double*) {
    ns = static_cast<int>(params[0]);
    Q = params + 1;
    as = lds_problem;
  }
  __device__ __forceinline__ void begin_problem(const double*, long long, int, int) {}

  template <bool FMA, int W>
  __device__ __forceinline__ double eval_impl(const double (&x)[E], double (&g)[E], int n, int sl) const {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (sl * E + e < kSvmDualMaxSamples) as[sl * E + e] = (sl * E + e < n) ? x[e] : 0.0;
    mi355::segment_lds_fence();
    double q[E];
    const int i0 = sl * E;
    {
      const double a0 = as[0];
#pragma unroll
      for (int e = 0; e < E; ++e) q[e] = (i0 + e < n) ? Q[i0 + e] * a0 : 0.0;
    }
    for (int j = 1; j < n; ++j) {
      const double aj = as[j];
      const double* col = Q + static_cast<long long>(j) * n + i0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if (i0 + e < n) q[e] = FMA ? __builtin_fma(col[e], aj, q[e]) : q[e] + col[e] * aj;
      }
    }
    mi355::segment_lds_fence();  // the next evaluation overwrites `as`
    double ones[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      ones[e] = (i0 + e < n) ? 1.0 : 0.0;
      g[e] = (i0 + e < n) ? q[e] - 1.0 : 0.0;
    }
    using AR = std::conditional_t<FMA, mi355::ArithFma, mi355::ArithExact>;
    const double aq = mi355::seg_dot<W, E, AR>(x, q);
    const double sa = mi355::seg_dot<W, E, AR>(x, ones);
    return 0.5 * aq - sa;
  }
  template <int W, int EE>
  __device__ __forceinline__ double eval(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(EE == E, "E");
    return eval_impl<false, W>(x, g, n, sl);
  }
  template <int W, int EE>
  __device__ __forceinline__ double eval_fma(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(EE == E, "E");
    return eval_impl<true, W>(x, g, n, sl);
  }
};

}  // namespace user_examples
