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
// SEGMENT_FULL: the variant for problems that fill their segment exactly (n == W * E), picked by the dispatch for the
// register-history Lbfgs kernels (engine_internal.hpp, launch_solve_rosenbrock_full); same values, fewer selects.
template <bool SEGMENT_FULL>
struct RosenbrockObjectiveT {
  static constexpr int kLdsDoubles = 0;  // LDS scratch per problem
  __host__ __device__ static constexpr int shared_lds_doubles() { return 0; }  // per workgroup, read only
  __device__ __forceinline__ void load(const double*, int, int, double*, double*) {}
  __device__ __forceinline__ void begin_problem(const double*, long long, int, int) {}

  // Which coordinates carry the two halves of the gradient / a term of the sum.  When the problem fills its segment
  // exactly (n == W * E: every benchmark shape) only the first and the last coordinate of the segment differ from the
  // interior, and for the coordinates in the middle of a lane the predicates are compile-time constants — the same
  // values as the general `j + 1 < n`, `0 < j < n`, with 6 selects per evaluation instead of 32.  (A run-time branch
  // between the two forms inside one kernel costs registers: the E = 4, ten-column kernel went to 256 + scratch.)
  template <int W, int E, bool FULL>
  static __device__ __forceinline__ bool has_next(int e, int sl, int n) {   // j + 1 < n
    if constexpr (FULL) return (e + 1 < E) || (sl + 1 < W);
    return sl * E + e + 1 < n;
  }
  template <int W, int E, bool FULL>
  static __device__ __forceinline__ bool has_prev(int e, int sl, int n) {   // 0 < j < n
    if constexpr (FULL) return (e > 0) || (sl > 0);
    return (sl * E + e > 0) && (sl * E + e < n);
  }

  template <int W, int E>
  __device__ __forceinline__ double eval(const double (&x)[E], double (&g)[E], int n, int sl) const {
    return eval_impl<W, E, SEGMENT_FULL>(x, g, n, sl);
  }
  template <int W, int E, bool FULL>
  __device__ __forceinline__ double eval_impl(const double (&x)[E], double (&g)[E], int n, int sl) const {
    // x_{j+1}: next element in-lane, or element 0 of the next lane.
    const double x_next_lane = from_next_lane(x[0]);
    double t2[E], term[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double xn = (e + 1 < E) ? x[(e + 1 < E) ? e + 1 : e] : x_next_lane;
      const double t1 = 1.0 - x[e];
      t2[e] = xn - x[e] * x[e];
      const double v = t1 * t1 + (100.0 * t2[e]) * t2[e];
      term[e] = has_next<W, E, FULL>(e, sl, n) ? v : 0.0;
    }
    // t2_{j-1}: previous element in-lane, or element E-1 of the previous lane.
    const double t2_prev_lane = from_prev_lane(t2[E - 1]);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool has_a = has_next<W, E, FULL>(e, sl, n);
      const bool has_b = has_prev<W, E, FULL>(e, sl, n);
      const double a = -2.0 * (1.0 - x[e]) + (200.0 * t2[e]) * (-2.0 * x[e]);
      const double b = 200.0 * ((e > 0) ? t2[(e > 0) ? e - 1 : 0] : t2_prev_lane);
      g[e] = (has_a && has_b) ? (a + b) : (has_a ? a : (has_b ? b : 0.0));
    }
    return seg_sum<W>(lane_tree_sum<E>(term));
  }

  // The same function under the fused arithmetic policy (ArithFma, wave_primitives.hpp):
  //   t2 = fma(-x_i, x_i, x_{i+1});  term = fma(100 t2, t2, t1 t1);  a = fma(200 t2, -2 x_i, -2 t1)
  // and the E terms of a lane are summed as a chain before the butterfly (oracle twin: Rosenbrock::eval with
  // Reducer::fma_group = E).
  template <int W, int E>
  __device__ __forceinline__ double eval_fma(const double (&x)[E], double (&g)[E], int n, int sl) const {
    return eval_fma_impl<W, E, SEGMENT_FULL>(x, g, n, sl);
  }
  template <int W, int E, bool FULL>
  __device__ __forceinline__ double eval_fma_impl(const double (&x)[E], double (&g)[E], int n, int sl) const {
    const double x_next_lane = from_next_lane(x[0]);
    double t2[E];
    double term[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double xn = (e + 1 < E) ? x[(e + 1 < E) ? e + 1 : e] : x_next_lane;
      const double t1 = 1.0 - x[e];
      t2[e] = __builtin_fma(-x[e], x[e], xn);
      const double v = __builtin_fma(100.0 * t2[e], t2[e], t1 * t1);
      term[e] = has_next<W, E, FULL>(e, sl, n) ? v : 0.0;
    }
    // the lane's terms: ascending within groups of kFmaGroup coordinates, groups pairwise (E = 8: two groups)
    double gs[(E + kFmaGroup - 1) / kFmaGroup];
#pragma unroll
    for (int q = 0; q < (E + kFmaGroup - 1) / kFmaGroup; ++q) {
      gs[q] = term[q * kFmaGroup];
#pragma unroll
      for (int e = 1; e < kFmaGroup && q * kFmaGroup + e < E; ++e) gs[q] = gs[q] + term[q * kFmaGroup + e];
    }
    const double sum = lane_tree_sum<(E + kFmaGroup - 1) / kFmaGroup>(gs);
    const double t2_prev_lane = from_prev_lane(t2[E - 1]);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const bool has_a = has_next<W, E, FULL>(e, sl, n);
      const bool has_b = has_prev<W, E, FULL>(e, sl, n);
      const double a = __builtin_fma(200.0 * t2[e], -2.0 * x[e], -2.0 * (1.0 - x[e]));
      const double b = 200.0 * ((e > 0) ? t2[(e > 0) ? e - 1 : 0] : t2_prev_lane);
      g[e] = (has_a && has_b) ? (a + b) : (has_a ? a : (has_b ? b : 0.0));
    }
    return seg_sum<W>(sum);
  }

  // diag H(x), for Second-mode solves (SolveArgs::hess_from_functor; the reference rebuilds its diagonal
  // preconditioner from function(x, &g, &H) at every iterate, lbfgs.h:129-138):
  //   H_jj = [j + 1 < n] (((1200 x_j) x_j - 400 x_{j+1}) + 2)  +  [j > 0] 200
  template <int W, int E>
  __device__ __forceinline__ void hess_diag(const double (&x)[E], double (&h)[E], int n, int sl) const {
    const double x_next_lane = from_next_lane(x[0]);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double xn = (e + 1 < E) ? x[(e + 1 < E) ? e + 1 : e] : x_next_lane;
      const bool has_a = has_next<W, E, SEGMENT_FULL>(e, sl, n);
      const bool has_b = has_prev<W, E, SEGMENT_FULL>(e, sl, n);
      const double a = ((1200.0 * x[e]) * x[e] - 400.0 * xn) + 2.0;
      h[e] = (has_a && has_b) ? (a + 200.0) : (has_a ? a : (has_b ? 200.0 : 0.0));
    }
  }
};
using RosenbrockObjective = RosenbrockObjectiveT<false>;
using RosenbrockFullObjective = RosenbrockObjectiveT<true>;

// Rosenbrock for solves with the condition_hessian stopping test of a Second-mode function switched on: the same functor
// plus the full Hessian.  A type of its own so that only the kernels of THOSE solves carry the LU code
// (hessian_condition_device.hpp) — 15-25 more vector registers, which the ordinary LDS-ring kernels keep for occupancy.
struct RosenbrockConditionObjective : RosenbrockObjectiveT<false> {
  // H(x) in full, column major n x n in the segment's LDS — what function(current_x, nullptr, &H) hands Progress::Update
  // of a Second-mode function (progress.h:203-210; the condition_hessian test, hessian_condition_device.hpp): the
  // diagonal above, H(j, j + 1) = H(j + 1, j) = -400 x_j, zero elsewhere.
  template <int W, int E>
  __device__ __forceinline__ void hess_full(const double (&x)[E], double* Hm, int n, int sl) const {
    double hd[E];
    hess_diag<W, E>(x, hd, n, sl);
    for (int t = sl; t < n * n; t += W) Hm[t] = 0.0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      if (j < n) {
        Hm[j * n + j] = hd[e];
        if (j + 1 < n) {
          const double off = -400.0 * x[e];
          Hm[(j + 1) * n + j] = off;
          Hm[j * n + j + 1] = off;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// f(x) = sum_i a_i x_i^2 + c  with the README quick-start operation order
// (README.md:21-28): term_i = (a_i*x_i)*x_i, g_i = (2 a_i)*x_i, f = sum + c.
template <int E>
struct DiagQuadraticObjective {
  static constexpr int kLdsDoubles = 0;
  __host__ __device__ static constexpr int shared_lds_doubles() { return 0; }
  // The coefficients of the lane's coordinates: registers up to two per lane; from four on they are re-read from the
  // parameter blob at every evaluation (an L1 hit) — the four-coordinate kernels with ten y columns in registers have no
  // eight registers to spare (they spilled 52-60 bytes per lane with a[] resident).
  static constexpr bool kCoefficientsInMemory = (E >= 4);
  double a_reg[kCoefficientsInMemory ? 1 : E];
  const double* a_mem;
  int n_;
  double c_reg;
  __device__ __forceinline__ void begin_problem(const double*, long long, int, int) {}
  // params: device pointer to a[0..n), c
  __device__ __forceinline__ void load(const double* params, int n, int sl, double*, double*) {
    a_mem = params;
    n_ = n;
    if constexpr (!kCoefficientsInMemory) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        a_reg[e] = (j < n) ? params[j] : 0.0;
      }
    }
    if constexpr (!kCoefficientsInMemory) c_reg = params[n];
  }
  __device__ __forceinline__ double constant() const {
    if constexpr (kCoefficientsInMemory) return a_mem[n_];
    else return c_reg;
  }
  __device__ __forceinline__ double coefficient(int e, int sl) const {
    if constexpr (kCoefficientsInMemory) {
      const int j = sl * E + e;
      return (j < n_) ? a_mem[j] : 0.0;
    } else {
      return a_reg[e];
    }
  }
  template <int W, int EE>
  __device__ __forceinline__ double eval(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(EE == E, "E");
    double term[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      term[e] = (j < n) ? (coefficient(e, sl) * x[e]) * x[e] : 0.0;
      g[e] = (j < n) ? (2.0 * coefficient(e, sl)) * x[e] : 0.0;
    }
    return seg_sum<W>(lane_tree_sum<E>(term)) + constant();
  }
  // fused policy: term_i accumulates as the chain fma(a_i x_i, x_i, previous) over the lane's coordinates
  template <int W, int EE>
  __device__ __forceinline__ double eval_fma(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(EE == E, "E");
    double ax[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      ax[e] = (j < n) ? coefficient(e, sl) * x[e] : 0.0;
      g[e] = (j < n) ? (2.0 * coefficient(e, sl)) * x[e] : 0.0;
    }
    return seg_sum<W>(lane_fma_dot<E>(ax, x)) + constant();
  }
};

// Ridge least squares  f(x) = ||A x - y_b||^2 + lambda ||x||^2: the reference README's
// `SquaredError(A, y) + lambda * L2Reg(n)` (README.md:122-167) through the First-mode
// branches of AddExpression / MulExpression (function_expressions.h:115-124, :229-236):
//   value = r.r + lambda*(x.x),   gradient = 2 A^T r + lambda*(2 x),   r = A x - y_b.
//
// A (rows <= 128, shared by the whole batch) is copied once per workgroup into LDS as
// AT[j][i] with a row pitch of 129 doubles and serves both matrix-vector products without a
// cross-lane reduction:
//   r = A x    lane owns the residual rows i = sl + W*q (for a fixed q the lanes of a segment
//              read consecutive doubles: conflict free); x_j is a broadcast read of an LDS copy.
//   A^T r      lane owns its E columns j; r_i is a broadcast read of an LDS copy of r (the odd
//              pitch keeps the column reads at a 2-way bank conflict).
// Both products are ascending multiply-then-add sums — exactly what the README functors compute
// when the reference's headers are built over oracle/eigen_shim, so the oracle's sequential
// policy is bit-identical to the reference for this objective too (2 A^T r: the factor 2 is
// exact, so it commutes with every rounding).  ||r||^2 is the pairwise tree over the 128 padded
// rows in natural order, so values do not depend on the mapping.  y_b is held in registers.
constexpr int kRidgeMaxRows = 128;
constexpr int kRidgePitch = kRidgeMaxRows + 1;

template <int W, int E>
struct SquaredErrorRidgeObjective {
  static constexpr int RPL = kRidgeMaxRows / W;  // residual rows per lane
  static constexpr int P = W * E;
  static constexpr int kLdsDoubles = P + kRidgeMaxRows;  // per problem: x and r staging
  // even number of doubles: the per-wavefront regions behind it stay 16-byte aligned
  __host__ __device__ static constexpr int shared_lds_doubles() { return P * kRidgePitch + (P * kRidgePitch) % 2; }
  const double* at_global;
  const double* AT;  // LDS
  double lambda;
  int rows;
  double y[RPL];
  double* xs;
  double* rs;

  // params (device): rows, lambda, AT[P][129] (zero padded)
  __device__ __forceinline__ void load(const double* params, int, int, double* lds_scratch, double* lds_shared) {
    rows = static_cast<int>(params[0]);
    lambda = params[1];
    at_global = params + 2;
    AT = lds_shared;
    xs = lds_scratch;
    rs = lds_scratch + P;
  }
  __device__ __forceinline__ void fill_shared(double* lds_shared, int tid, int nthreads) const {
    for (int t = tid; t < P * kRidgePitch; t += nthreads) lds_shared[t] = at_global[t];
  }
  __device__ __forceinline__ void begin_problem(const double* per_problem, long long prob, int stride, int sl) {
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const int i = sl + W * q;
      y[q] = (i < rows) ? per_problem[prob * stride + i] : 0.0;
    }
  }

  template <int WW, int EE>
  __device__ __forceinline__ double eval(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(WW == W && EE == E, "mapping");
#pragma unroll
    for (int e = 0; e < E; ++e) xs[sl * E + e] = x[e];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double acc[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) acc[q] = 0.0;
    // Column j of A for this lane's rows is fetched one iteration ahead of its use, so the
    // LDS latency overlaps the previous column's multiply-adds.
    const double* col = AT + sl;
    double a_cur[RPL], x_cur = xs[0];
#pragma unroll
    for (int q = 0; q < RPL; ++q) a_cur[q] = col[W * q];
    for (int j = 0; j < n; ++j) {
      double a_nxt[RPL];
      const int jn = (j + 1 < n) ? j + 1 : j;  // the last prefetch re-reads a valid column
      const double x_nxt = xs[jn];
#pragma unroll
      for (int q = 0; q < RPL; ++q) a_nxt[q] = col[jn * kRidgePitch + W * q];
#pragma unroll
      for (int q = 0; q < RPL; ++q) acc[q] = acc[q] + a_cur[q] * x_cur;
#pragma unroll
      for (int q = 0; q < RPL; ++q) a_cur[q] = a_nxt[q];
      x_cur = x_nxt;
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) rs[sl + W * q] = acc[q] - y[q];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ||r||^2 over the rows in natural order: lane sl re-reads rows sl*RPL .. sl*RPL+RPL-1
    double rr[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
      const double r = rs[sl * RPL + q];
      rr[q] = r * r;
    }
    const double f1 = seg_sum<W>(lane_tree_sum<RPL>(rr));
    double ga[E];
#pragma unroll
    for (int e = 0; e < E; ++e) ga[e] = 0.0;
    const double* mine = AT + (sl * E) * kRidgePitch;
    double m_cur[E], r_cur = rs[0];
#pragma unroll
    for (int e = 0; e < E; ++e) m_cur[e] = mine[e * kRidgePitch];
    for (int i = 0; i < rows; ++i) {
      double m_nxt[E];
      const int in = (i + 1 < rows) ? i + 1 : i;
      const double r_nxt = rs[in];
#pragma unroll
      for (int e = 0; e < E; ++e) m_nxt[e] = mine[e * kRidgePitch + in];
#pragma unroll
      for (int e = 0; e < E; ++e) ga[e] = ga[e] + m_cur[e] * r_cur;
#pragma unroll
      for (int e = 0; e < E; ++e) m_cur[e] = m_nxt[e];
      r_cur = r_nxt;
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      g[e] = (j < n) ? 2.0 * ga[e] + lambda * (2.0 * x[e]) : 0.0;
    }
    const double xx = seg_dot<W, E>(x, x);
    return f1 + lambda * xx;
  }
};

}  // namespace mi355
