// svm_squared_hinge.hpp — a USER device objective for the MI355X L-BFGS engine (worked example).
//
// The reference accepts any host functor (function_base.h:103-120); its example src/examples/svm_primal_lbfgs.cc:35-103
// minimises the soft-margin SVM primal with a squared hinge loss,
//     f(w, b) = 0.5 ||w||^2 + C sum_i max(0, 1 - y_i (x_i . w + b))^2,
// by handing such a functor to Lbfgs.  On the GPU the objective has to exist as device code.  This header is all a
// user writes for that: a functor with the interface of cppnumericalsolvers_amd/csrc/objectives.hpp,
//
//     static constexpr int kLdsDoubles;              LDS scratch per problem (doubles)
//     static constexpr int shared_lds_doubles();      read-only LDS shared by a workgroup (0 = none)
//     void load(params, n, sl, lds_problem, lds_shared);   once per wavefront segment
//     void begin_problem(per_problem, prob, stride, sl);   once per fetched problem
//     template <int W, int E> double eval(x, g, n, sl);    value (segment-uniform) + gradient
//
// compiled into a build of the library by
//     _build.build(output=".../libmi355_lbfgs_svm.so",
//                  user_objectives=[dict(name="svm_squared_hinge", header=<this file>,
//                                        type="user_examples::SvmSquaredHinge", id=100)])
// and selected with objective id 100.  A problem is owned by W lanes, lane `sl` holds coordinates j = sl*E + e of x
// (padding coordinates are zero); `n` = d + 1: the first d coordinates are w, the last is the bias b.
//
// params = [N, d, C, X (N x d, row major), y (N)] — shared by the batch, read through the caches (they are a few KB).
// Operation order = the reference functor's, so that the CPU twin can be pinned against it:
//   score_i  = ((X_i0 w_0 + X_i1 w_1) + ...) + b          (features * w, ascending columns)
//   slack_i  = max(0, 1 - y_i score_i);   ws_i = (-2 slack_i) y_i
//   hinge    = sum_i slack_i^2  ascending i  (slacks are staged in LDS; every lane adds them in order)
//   g_j      = w_j + C ((X_0j ws_0 + X_1j ws_1) + ...),   g_d = C ((ws_0 + ws_1) + ...)
//   value    = 0.5 (w . w) + C hinge       (w . w: the engine's pairwise tree over the coordinates)
#pragma once

namespace user_examples {

constexpr int kSvmMaxCoordinates = 64;  // d + 1
constexpr int kSvmMaxSamples = 256;

struct SvmSquaredHinge {
  static constexpr int kLdsDoubles = kSvmMaxCoordinates + 2 * kSvmMaxSamples;  // staged x, weighted slacks, slack^2
  __host__ __device__ static constexpr int shared_lds_doubles() { return 0; }

  const double* X;
  const double* y;
  int N, d;
  double C;
  double* xs;  // LDS: x of the current evaluation, all coordinates
  double* ws;  // LDS: (-2 slack_i) y_i
  double* sq;  // LDS: slack_i^2

  __device__ __forceinline__ void load(const double* params, int, int, double* lds_problem, double*) {
    N = static_cast<int>(params[0]);
    d = static_cast<int>(params[1]);
    C = params[2];
    X = params + 3;
    y = X + static_cast<long long>(N) * d;
    xs = lds_problem;
    ws = lds_problem + kSvmMaxCoordinates;
    sq = ws + kSvmMaxSamples;
  }
  __device__ __forceinline__ void begin_problem(const double*, long long, int, int) {}

  template <int W, int E>
  __device__ __forceinline__ double eval(const double (&x)[E], double (&g)[E], int n, int sl) const {
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (sl * E + e < n) xs[sl * E + e] = x[e];
    mi355::segment_lds_fence();
    const double b = xs[d];
    for (int i = sl; i < N; i += W) {  // the lanes of the segment share the samples
      const double* row = X + static_cast<long long>(i) * d;
      double score = row[0] * xs[0];
      for (int j = 1; j < d; ++j) score = score + row[j] * xs[j];
      score = score + b;
      const double t = 1.0 - y[i] * score;
      const double slack = (t < 0.0) ? 0.0 : t;
      ws[i] = (-2.0 * slack) * y[i];
      sq[i] = slack * slack;
    }
    mi355::segment_lds_fence();
    double hinge = sq[0];
    for (int i = 1; i < N; ++i) hinge = hinge + sq[i];
    double wsq[E];
#pragma unroll
    for (int e = 0; e < E; ++e) wsq[e] = (sl * E + e < d) ? x[e] * x[e] : 0.0;
    const double ww = mi355::seg_sum<W>(mi355::lane_tree_sum<E>(wsq));
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      double acc = 0.0;
      if (j < d) {
        acc = X[j] * ws[0];
        for (int i = 1; i < N; ++i) acc = acc + X[static_cast<long long>(i) * d + j] * ws[i];
        g[e] = x[e] + C * acc;
      } else if (j == d) {
        acc = ws[0];
        for (int i = 1; i < N; ++i) acc = acc + ws[i];
        g[e] = C * acc;
      } else {
        g[e] = 0.0;
      }
    }
    mi355::segment_lds_fence();  // the next evaluation overwrites xs / ws / sq
    return 0.5 * ww + C * hinge;
  }
};

}  // namespace user_examples
