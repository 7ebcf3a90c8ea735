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

// cb[i] = cb[i] - xj * col[i] for i in [lo, hi): the element updates of one substitution step are independent of each other, so
// they go eight at a time — sixteen LDS loads in flight, then the products, then the stores — instead of one LDS round trip per
// element (the kernel runs one wavefront per CU at n = 64: nothing else hides that latency).  Same operations per element.
__device__ __forceinline__ void column_axpy(double* cb, const double* col, double xj, int lo, int hi) {
  constexpr int kChunk = 8;
  for (int i0 = lo; i0 < hi; i0 += kChunk) {
    double l[kChunk], c[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int i = (i0 + u < hi) ? i0 + u : hi - 1;
      l[u] = col[i];
      c[u] = cb[i];
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u)
      if (i0 + u < hi) cb[i0 + u] = c[u] - xj * l[u];
  }
}

template <int W>
__device__ __forceinline__ double seg_hessian_condition(double* Hm, double* colbuf, int* piv, int n, int sl) {
  constexpr int kChunk = 8;
  double sh = 0.0;
  for (int t0 = sl; t0 < n * n; t0 += kChunk * W) {    // the lane's ascending chain over t = sl, sl + W, ...; loads in batches
    double h[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; ++u) {
      const int t = t0 + u * W;
      h[u] = (t < n * n) ? Hm[t] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < kChunk; ++u)
      if (t0 + u * W < n * n) sh += h[u] * h[u];
  }
  sh = seg_sum<W>(sh);
  constexpr int kRows = kHessianConditionMaxN / 8;   // rows of the trailing block a lane may own (W >= 8)
  for (int k = 0; k < n; ++k) {
    double* const colk = Hm + k * n;
    int p = k;
    double best = __builtin_fabs(colk[k]);
    for (int i0 = k + 1; i0 < n; i0 += kChunk) {      // PartialPivLU: the first maximum of |column k| from the diagonal down
      double v[kChunk];
#pragma unroll
      for (int u = 0; u < kChunk; ++u) v[u] = __builtin_fabs(colk[(i0 + u < n) ? i0 + u : n - 1]);
#pragma unroll
      for (int u = 0; u < kChunk; ++u)
        if (i0 + u < n && v[u] > best) {
          best = v[u];
          p = i0 + u;
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
    const int rows = (n - (k + 1) - sl + W - 1) / W;  // rows of the trailing block this lane owns (<= 0: none)
    for (int j0 = k + 1; j0 < n; j0 += 4) {           // four columns of the rank-1 update at a time
      double ukj[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) ukj[c] = Hm[((j0 + c < n) ? j0 + c : n - 1) * n + k];
#pragma unroll
      for (int r = 0; r < kRows; ++r) {
        if (r < rows) {
          const int i = k + 1 + sl + r * W;
          double a[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = Hm[((j0 + c < n) ? j0 + c : n - 1) * n + i];
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (j0 + c < n) Hm[(j0 + c) * n + i] = a[c] - lik[r] * ukj[c];
        }
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
    for (int j = 0; j < n; ++j)                       // unit lower triangle, column oriented
      column_axpy(cb, Hm + j * n, cb[j], j + 1, n);
    for (int j = n - 1; j >= 0; --j) {                // upper triangle, last column first
      const double* const colj = Hm + j * n;
      const double xj = cb[j] / colj[j];
      cb[j] = xj;
      column_axpy(cb, colj, xj, 0, j);
    }
    if (active) {
      for (int i0 = 0; i0 < n; i0 += kChunk) {
        double v[kChunk];
#pragma unroll
        for (int u = 0; u < kChunk; ++u) v[u] = cb[(i0 + u < n) ? i0 + u : n - 1];
#pragma unroll
        for (int u = 0; u < kChunk; ++u)
          if (i0 + u < n) si += v[u] * v[u];
      }
    }
  }
  si = seg_sum<W>(si);
  return __builtin_sqrt(sh) * __builtin_sqrt(si);
}

}  // namespace mi355
