// hessian_condition_device.hpp — Progress::condition_hessian of a Second-mode function whose Hessian is NOT constant.
//
// The reference evaluates function(current_x, nullptr, &H) in every Progress::Update and sets
//     condition_hessian = H.norm() * H.inverse().norm()                      (solver/progress.h:203-210)
// which the stopping test `condition_hessian > stop.condition_hessian` reads last (:318-325).  For a constant Hessian the
// number is computed once on the host (mi355_lbfgs_hessian_condition); for a device functor with a `hess_full` the solve
// kernel computes it here, per problem and per iteration, when — and only when — the test is switched on:
//   * H (n x n, column major, n <= 64) lives in LDS of the problem's segment;
//   * LU with partial (first-maximum) pivoting, right-looking, in place: the rows of the trailing block are spread over
//     the W lanes (consecutive lanes touch consecutive doubles of a column), the pivot search is the sequential scan of
//     the reference's PartialPivLU — every element goes through exactly the operations of a loop-based Eigen
//     (oracle/eigen_shim/Eigen/LU, which the reference binary of oracle/_ref is built over);
//   * the inverse one unit vector per lane (W columns at a time), each lane doing the column-oriented substitutions of
//     PartialPivLU::solve on its own column buffer in LDS; the factor entries it reads are the same address in every lane
//     (LDS broadcasts);
//   * the two Frobenius norms as per-lane partial sums and one butterfly each.  The reference's norms are single ascending
//     chains over the column-major storage: the device's sums round differently in the last bits, so device and twin agree
//     on the DECISION `condition > threshold` (and on everything downstream of it) unless the condition number sits within
//     an ulp-scale distance of the threshold; the value itself is not part of the device's output (the host reports it at
//     the returned x, include/cppoptlib/solver/lbfgs.h ReportHessianCondition).
#pragma once
#include "wave_primitives.hpp"

namespace mi355 {

// LDS doubles the condition computation needs per problem: H, W column buffers of n + 1, the pivots (ints, two per double)
__host__ __device__ inline int hessian_condition_lds_doubles(int n, int W) { return n * n + W * (n + 1) + (n + 1) / 2 + 1; }

constexpr int kHessianConditionMaxN = 64;

template <int W>
__device__ __forceinline__ double seg_hessian_condition(double* Hm, double* colbuf, int* piv, int n, int sl) {
  double sh = 0.0;
  for (int t = sl; t < n * n; t += W) sh += Hm[t] * Hm[t];
  sh = seg_sum<W>(sh);
  constexpr int kRows = kHessianConditionMaxN / 8;   // rows of the trailing block a lane may own (W >= 8)
  for (int k = 0; k < n; ++k) {
    double* const colk = Hm + k * n;
    int p = k;
    double best = __builtin_fabs(colk[k]);
    for (int i = k + 1; i < n; ++i) {                 // PartialPivLU: the first maximum of |column k| from the diagonal down
      const double v = __builtin_fabs(colk[i]);
      if (v > best) {
        best = v;
        p = i;
      }
    }
    if (sl == 0) piv[k] = p;
    if (best != 0.0) {
      if (p != k) {
        for (int j = sl; j < n; j += W) {
          const double a = Hm[j * n + k], b = Hm[j * n + p];
          Hm[j * n + k] = b;
          Hm[j * n + p] = a;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const double pivot = colk[k];
      for (int i = k + 1 + sl; i < n; i += W) colk[i] = colk[i] / pivot;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    double lik[kRows];                                // this lane's multipliers l(i, k)
#pragma unroll
    for (int r = 0; r < kRows; ++r) {
      const int i = k + 1 + sl + r * W;
      lik[r] = (i < n) ? colk[i] : 0.0;
    }
    for (int j = k + 1; j < n; ++j) {
      double* const colj = Hm + j * n;
      const double ukj = colj[k];
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        const int i = k + 1 + sl + r * W;
        if (i < n) colj[i] = colj[i] - lik[r] * ukj;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // inverse: W unit vectors at a time, one per lane; si = sum of the squares of its entries
  double si = 0.0;
  double* const cb = colbuf + sl * (n + 1);
  for (int c0 = 0; c0 < n; c0 += W) {
    const int c = c0 + sl;
    const bool active = c < n;
    for (int i = 0; i < n; ++i) cb[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) {                     // the row interchanges, in order
      const int p = piv[k];
      const double a = cb[k], b = cb[p];
      cb[k] = b;
      cb[p] = a;
    }
    for (int j = 0; j < n; ++j) {                     // unit lower triangle, column oriented
      const double xj = cb[j];
      const double* const colj = Hm + j * n;
      for (int i = j + 1; i < n; ++i) cb[i] = cb[i] - xj * colj[i];
    }
    for (int j = n - 1; j >= 0; --j) {                // upper triangle, last column first
      const double* const colj = Hm + j * n;
      const double xj = cb[j] / colj[j];
      cb[j] = xj;
      for (int i = 0; i < j; ++i) cb[i] = cb[i] - xj * colj[i];
    }
    if (active)
      for (int i = 0; i < n; ++i) si += cb[i] * cb[i];
  }
  si = seg_sum<W>(si);
  return __builtin_sqrt(sh) * __builtin_sqrt(si);
}

}  // namespace mi355
