// svm_squared_hinge_wide.hpp — the same USER objective (svm_squared_hinge.hpp: the soft-margin SVM primal with a squared
// hinge loss of the reference's src/examples/svm_primal_lbfgs.cc:35-103) for MORE THAN 255 FEATURES.
//
// The reference example runs at any dimension.  Above n = 256 the engine gives a problem to a workgroup of 256 threads
// (cppnumericalsolvers_amd/csrc/lbfgs_wide_kernel.hpp); a user objective takes part in that regime with a second
// functor that has the workgroup interface:
//
//     void load(params, n);                                                   once per workgroup
//     template <int E> double eval(x, g, n, red, xmem, dir, gd_out);          value (uniform) + gradient [+ g . dir]
//
// x and g are mi355::WideVec<E>: thread t of T = mi355::wide_threads() owns the coordinates j = t, t + T, ... (mi355::wide_for<E> walks them); they
// live in registers (E > 0, small n) or in memory (E == 0).  `xmem` is n doubles of workspace through which a
// register-resident x reaches the other threads; `red` is the LDS scratch of mi355::wide_reduce.  When `dir` is given the
// evaluation also returns g . dir through `gd_out` (the line search wants it, and the gradient is in flight anyway).
// Compiled in by   _build.build(..., user_objectives=[dict(..., wide_type="user_examples::SvmSquaredHingeWide",
//                                                          wide_header=<this file>)])
//
// params as in svm_squared_hinge.hpp: [N, d, C, X (N x d, row major), y (N)], n = d + 1, N <= 256 samples.
// Operation order = the reference functor's (the CPU twin is the same one as for the small-n functor):
//   score_i  sequential over the features, by the thread that owns sample i          (X rows are read through the caches)
//   hinge    sequential over the samples, every thread adds the staged slack^2 in order
//   g_j      sequential over the samples, by the thread that owns coordinate j      (X columns: coalesced)
//   w . w    the engine's reduction over the coordinates (thread partial sums, then the pairwise tree)
#pragma once

namespace user_examples {

constexpr int kSvmWideMaxSamples = 256;

struct SvmSquaredHingeWide {
  const double* X;
  const double* y;
  int N, d;
  double C;

  __device__ __forceinline__ void load(const double* params, int) {
    N = static_cast<int>(params[0]);
    d = static_cast<int>(params[1]);
    C = params[2];
    X = params + 3;
    y = X + static_cast<long long>(N) * d;
  }

  template <int E>
  __device__ __forceinline__ double eval(mi355::WideVec<E>& x, mi355::WideVec<E>& g, int n, double* red, double* xmem,
                                         const mi355::WideVec<E>* dir = nullptr, double* gd_out = nullptr) const {
    __shared__ double ws[kSvmWideMaxSamples];  // (-2 slack_i) y_i
    __shared__ double sq[kSvmWideMaxSamples];  // slack_i^2
    const double* xs = x.mem;
    if constexpr (E > 0) {
      __syncthreads();  // the previous evaluation's readers of xmem are done
      mi355::wide_for<E>(n, [&](int j, int e) { xmem[j] = x.reg[e]; });
      xs = xmem;
    }
    __syncthreads();    // x is visible to the workgroup; ws / sq of the previous evaluation are no longer read
    const double b = xs[d];
    for (int i = threadIdx.x; i < N; i += mi355::wide_threads()) {
      const double* row = X + static_cast<long long>(i) * d;
      double score = row[0] * xs[0];
      for (int j = 1; j < d; ++j) score = score + row[j] * xs[j];
      score = score + b;
      const double t = 1.0 - y[i] * score;
      const double slack = (t < 0.0) ? 0.0 : t;
      ws[i] = (-2.0 * slack) * y[i];
      sq[i] = slack * slack;
    }
    __syncthreads();
    double hinge = sq[0];
    for (int i = 1; i < N; ++i) hinge = hinge + sq[i];
    double ww = 0.0, gd = 0.0;
    mi355::wide_for<E>(n, [&](int j, int e) {
      const double xj = x.get(j, e);
      double gj;
      if (j < d) {
        ww = ww + xj * xj;
        double acc = X[j] * ws[0];
        for (int i = 1; i < N; ++i) acc = acc + X[static_cast<long long>(i) * d + j] * ws[i];
        gj = xj + C * acc;
      } else {  // j == d: the bias
        double acc = ws[0];
        for (int i = 1; i < N; ++i) acc = acc + ws[i];
        gj = C * acc;
      }
      g.at(j, e) = gj;
      if (dir) gd = gd + gj * dir->get(j, e);
    });
    double sums[2] = {ww, gd}, none[1] = {0.0};
    mi355::wide_reduce<2, 0>(sums, none, red);
    if (gd_out) *gd_out = sums[1];
    return 0.5 * sums[0] + C * hinge;
  }
};

}  // namespace user_examples
