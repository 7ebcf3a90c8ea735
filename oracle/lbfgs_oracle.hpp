// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (plain C++17, no Eigen, no HIP) of the reference hot path
//   Solver::Minimize -> Lbfgs::OptimizationStep -> MoreThuente::cvsrch/cstep -> objective
// of PatWie/CppNumericalSolvers 2.0.0.  Every function cites the reference
// file:line it follows (paths relative to /root/reference/include/cppoptlib/).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// this file; the HIP engine under cppnumericalsolvers_amd/csrc never includes,
// links or calls it.
//
// Pinning status: cstep is pinned by the seven exact vectors of the reference's
// src/test/cstep_test.cc; the end-to-end loop is pinned (a) by the reference's
// own end-to-end expectations (src/test/verify.cc, README quick start,
// Dockerfile.test) and (b) bit-for-bit against the UNMODIFIED reference headers
// compiled over oracle/eigen_shim (oracle/_ref, see oracle/Makefile and
// tests/test_oracle_vs_reference.py).  Real Eigen is not installed here, so the
// summation order of Eigen's own dot/norm kernels is the one thing not pinned.
//
// Arithmetic policy: built with -ffp-contract=off, so every a*b+c is a rounded
// multiply followed by a rounded add -- the same as the HIP engine, which is
// also built with -ffp-contract=off.  The only degree of freedom is the
// summation tree of the n-element reductions:
//   Reduction::Sequential  s = ((v0+v1)+v2)+...        (what a scalar loop, and
//                          the eigen_shim build of the reference, computes)
//   Reduction::Butterfly   pairwise tree over a zero-padded power-of-two width
//                          W: level 1 (v0+v1),(v2+v3).., level 2 ((v0+v1)+(v2+v3)),..
//                          -- the tree an xor-butterfly over a W-lane wave
//                          segment produces in every lane.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

namespace oracle {

// Strided: the twin of the workgroup kernel for n > 256 (csrc/lbfgs_wide_kernel.hpp): lane t of `width` (= 256) lanes adds
// its own terms v[t], v[t + width], v[t + 2 width], ... in ascending order onto 0.0, then the partial sums go through
// the butterfly tree.  No fused multiply-adds.
enum class Reduction : int { Sequential = 0, Butterfly = 1, Strided = 2 };

// n doubles of scratch: on the stack up to 1024 (every wavefront-resident shape), on the heap above
struct Scratch {
  double stack_[1024];
  std::vector<double> heap_;
  double* p_;
  explicit Scratch(int n) {
    if (n <= 1024) {
      p_ = stack_;
    } else {
      heap_.resize(static_cast<size_t>(n));
      p_ = heap_.data();
    }
  }
  double& operator[](int i) { return p_[i]; }
  operator double*() { return p_; }
  double* data() { return p_; }
};

// Third policy, `butterfly_fma` (Butterfly with fma_group = E > 0): the twin of the engine's MI355_ARITH_FMA kernels
// (csrc/wave_primitives.hpp, ArithFma).  An inner product is a fused-multiply-add CHAIN over every group of E
// consecutive coordinates (the E coordinates one lane owns; positions past n are zeros, fused like the rest) followed
// by the butterfly tree over the groups; the axpys of the two-loop recursion, the trial point of the line search and
// the objectives' multiply-adds are single fused operations (madd / nmadd below).  Everything else is unchanged.
struct Reducer {
  Reduction kind = Reduction::Sequential;
  int width = 64;  // butterfly width W (power of two, >= n)
  int fma_group = 0;  // 0: no fused multiply-add anywhere; E > 0: the butterfly_fma policy with E coordinates per lane

  bool fma() const { return fma_group > 0; }
  double madd(double a, double b, double c) const { return fma_group > 0 ? std::fma(a, b, c) : a * b + c; }   // a*b + c
  double nmadd(double a, double b, double c) const { return fma_group > 0 ? std::fma(-a, b, c) : c - a * b; }  // c - a*b

  // Sum of v[0..n) under the chosen tree (w: butterfly width override, 0 = `width`).
  double sum(const double* v, int n, int w_override = 0) const {
    if (kind == Reduction::Sequential) {
      if (n == 0) return 0.0;
      double s = v[0];
      for (int i = 1; i < n; ++i) s = s + v[i];
      return s;
    }
    double buf[1024];
    const int w = w_override ? w_override : width;
    if (kind == Reduction::Strided) {
      for (int t = 0; t < w; ++t) {
        double acc = 0.0;
        for (int j = t; j < n; j += w) acc = acc + v[j];
        buf[t] = acc;
      }
    } else {
      for (int i = 0; i < w; ++i) buf[i] = (i < n) ? v[i] : 0.0;
    }
    for (int stride = 1; stride < w; stride <<= 1)
      for (int i = 0; i < w; i += 2 * stride) buf[i] = buf[i] + buf[i + stride];
    return buf[0];
  }
  double dot(const double* a, const double* b, int n) const {
    Scratch t(n > width ? n : width);
    if (fma_group > 0) {
      const int groups = width / fma_group;
      for (int l = 0; l < groups; ++l) {
        const int j0 = l * fma_group;
        double acc = (j0 < n) ? a[j0] * b[j0] : 0.0 * 0.0;
        for (int e = 1; e < fma_group; ++e) {
          const int j = j0 + e;
          acc = (j < n) ? std::fma(a[j], b[j], acc) : std::fma(0.0, 0.0, acc);
        }
        t[l] = acc;
      }
      return sum(t, groups, groups);
    }
    for (int i = 0; i < n; ++i) t[i] = a[i] * b[i];
    return sum(t, n);
  }
  // sum of v[0..n) the way the fused kernels add a lane's terms: a chain per group of fma_group consecutive
  // terms, then the butterfly tree over the groups (terms past n are zeros)
  double sum_grouped(const double* v, int n) const {
    if (fma_group <= 0) return sum(v, n);
    double t[1024];
    const int groups = width / fma_group;
    for (int l = 0; l < groups; ++l) {
      const int j0 = l * fma_group;
      double acc = (j0 < n) ? v[j0] : 0.0;
      for (int e = 1; e < fma_group; ++e) acc = acc + ((j0 + e < n) ? v[j0 + e] : 0.0);
      t[l] = acc;
    }
    return sum(t, groups, groups);
  }
  // Eigen `.norm()` == sqrt(squaredNorm()).
  double norm(const double* a, int n) const { return std::sqrt(dot(a, a, n)); }
  // Eigen `.lpNorm<Infinity>()` == max |a_i| (order independent, exact).
  static double amax(const double* a, int n) {
    double m = 0.0;
    for (int i = 0; i < n; ++i) {
      const double t = std::fabs(a[i]);
      if (m < t) m = t;
    }
    return m;
  }
};

// ---------------------------------------------------------------------------
// Objectives (device-functor twins live in cppnumericalsolvers_amd/csrc).
// value = eval(x, g): writes the gradient, returns f.  `red` decides the
// summation tree of the value only (gradients are element-wise).
// ---------------------------------------------------------------------------
struct Objective {
  virtual ~Objective() = default;
  virtual double eval(const double* x, double* g, int n, const Reducer& red) const = 0;
  // Objectives with per-problem data (e.g. the right-hand side y_b) switch to problem b.
  virtual void set_problem(int64_t /*b*/) {}
  // diag H(x) for Second-mode functions with a non-constant Hessian (Lbfgs::hessian_from_objective); false = none
  virtual bool hess_diag(const double* /*x*/, double* /*h*/, int /*n*/) const { return false; }
  // the full Hessian H(x), column major n x n (what function(x, nullptr, &H) hands Progress::Update, progress.h:203-210)
  virtual bool hess_full(const double* /*x*/, double* /*H*/, int /*n*/) const { return false; }
};

// progress.h:208: condition_hessian = H.norm() * H.inverse().norm() — Frobenius norms as the dot of the column-major storage
// with itself (one ascending chain), the inverse by LU with partial (first-maximum) pivoting and one column solve per unit
// vector: exactly what the loop-based Eigen stand-in of oracle/_ref computes (oracle/eigen_shim/Eigen/LU, Core:275-282), so
// that the sequential twin reproduces the reference binary's value bit for bit.  H: column major, n x n.
inline double ShimHessianCondition(const std::vector<double>& H, int n) {
  const size_t nn = static_cast<size_t>(n);
  std::vector<double> lu = H, inv(H.size(), 0.0), col(nn);
  std::vector<int> piv(nn);
  auto at = [&](std::vector<double>& m, int i, int j) -> double& { return m[static_cast<size_t>(j) * nn + i]; };
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = std::fabs(at(lu, k, k));
    for (int i = k + 1; i < n; ++i) {
      const double v = std::fabs(at(lu, i, k));
      if (v > best) {
        best = v;
        p = i;
      }
    }
    piv[static_cast<size_t>(k)] = p;
    if (best != 0.0) {
      if (p != k)
        for (int j = 0; j < n; ++j) std::swap(at(lu, k, j), at(lu, p, j));
      for (int i = k + 1; i < n; ++i) at(lu, i, k) = at(lu, i, k) / at(lu, k, k);
    }
    for (int j = k + 1; j < n; ++j)
      for (int i = k + 1; i < n; ++i) at(lu, i, j) = at(lu, i, j) - at(lu, i, k) * at(lu, k, j);
  }
  for (int c = 0; c < n; ++c) {
    std::fill(col.begin(), col.end(), 0.0);
    col[static_cast<size_t>(c)] = 1.0;
    for (int k = 0; k < n; ++k) std::swap(col[static_cast<size_t>(k)], col[static_cast<size_t>(piv[static_cast<size_t>(k)])]);
    for (int j = 0; j < n; ++j)
      for (int i = j + 1; i < n; ++i) col[static_cast<size_t>(i)] = col[static_cast<size_t>(i)] - col[static_cast<size_t>(j)] * at(lu, i, j);
    for (int j = n - 1; j >= 0; --j) {
      col[static_cast<size_t>(j)] = col[static_cast<size_t>(j)] / at(lu, j, j);
      for (int i = 0; i < j; ++i) col[static_cast<size_t>(i)] = col[static_cast<size_t>(i)] - col[static_cast<size_t>(j)] * at(lu, i, j);
    }
    for (int i = 0; i < n; ++i) at(inv, i, c) = col[static_cast<size_t>(i)];
  }
  double sh = 0.0, si = 0.0;
  for (size_t t = 0; t < nn * nn; ++t) {
    sh = (t == 0) ? H[0] * H[0] : sh + H[t] * H[t];
    si = (t == 0) ? inv[0] * inv[0] : si + inv[t] * inv[t];
  }
  return std::sqrt(sh) * std::sqrt(si);
}

// Chained Rosenbrock-N; reduces to src/test/verify.cc:58-69 at N = 2 with the
// same operation order: t1 = 1-x0, t2 = x1-x0*x0, f = t1*t1 + 100*t2*t2,
// g0 = -2*(1-x0) + 200*(x1-x0*x0)*(-2*x0), g1 = 200*(x1-x0*x0).
struct Rosenbrock final : Objective {
  double eval(const double* x, double* g, int n, const Reducer& red) const override {
    Scratch term(n), t2v(n);
    if (red.fma()) {  // RosenbrockObjective::eval_fma
      for (int i = 0; i + 1 < n; ++i) {
        const double t1 = 1.0 - x[i];
        const double t2 = std::fma(-x[i], x[i], x[i + 1]);
        t2v[i] = t2;
        term[i] = std::fma(100.0 * t2, t2, t1 * t1);
      }
      for (int i = 0; i < n; ++i) {
        const bool has_a = (i + 1 < n);
        const bool has_b = (i > 0);
        double a = 0.0, b = 0.0;
        if (has_a) a = std::fma(200.0 * t2v[i], -2.0 * x[i], -2.0 * (1.0 - x[i]));
        if (has_b) b = 200.0 * t2v[i - 1];
        g[i] = (has_a && has_b) ? (a + b) : (has_a ? a : b);
      }
      return red.sum_grouped(term, n > 0 ? n - 1 : 0);
    }
    for (int i = 0; i + 1 < n; ++i) {
      const double t1 = 1.0 - x[i];
      const double t2 = x[i + 1] - x[i] * x[i];
      t2v[i] = t2;
      term[i] = t1 * t1 + (100.0 * t2) * t2;
    }
    for (int i = 0; i < n; ++i) {
      const bool has_a = (i + 1 < n);
      const bool has_b = (i > 0);
      double a = 0.0, b = 0.0;
      if (has_a) a = -2.0 * (1.0 - x[i]) + (200.0 * t2v[i]) * (-2.0 * x[i]);
      if (has_b) b = 200.0 * t2v[i - 1];
      g[i] = (has_a && has_b) ? (a + b) : (has_a ? a : b);
    }
    return red.sum(term, n > 0 ? n - 1 : 0);
  }
  // H_jj = [j + 1 < n] (((1200 x_j) x_j - 400 x_{j+1}) + 2) + [j > 0] 200   (RosenbrockObjectiveT::hess_diag)
  bool hess_diag(const double* x, double* h, int n) const override {
    for (int i = 0; i < n; ++i) {
      const bool has_a = (i + 1 < n), has_b = (i > 0);
      const double a = has_a ? ((1200.0 * x[i]) * x[i] - 400.0 * x[i + 1]) + 2.0 : 0.0;
      h[i] = (has_a && has_b) ? (a + 200.0) : (has_a ? a : (has_b ? 200.0 : 0.0));
    }
    return true;
  }
  // the tridiagonal Hessian: the diagonal above, H(i, i + 1) = H(i + 1, i) = -400 x_i (oracle/ref_capi.cpp RosenbrockNSecond)
  bool hess_full(const double* x, double* H, int n) const override {
    std::vector<double> d(static_cast<size_t>(n));
    hess_diag(x, d.data(), n);
    for (size_t t = 0; t < static_cast<size_t>(n) * n; ++t) H[t] = 0.0;
    for (int i = 0; i < n; ++i) {
      H[static_cast<size_t>(i) * n + i] = d[static_cast<size_t>(i)];
      if (i + 1 < n) {
        H[static_cast<size_t>(i + 1) * n + i] = -400.0 * x[i];   // (i, i + 1), column major
        H[static_cast<size_t>(i) * n + i + 1] = -400.0 * x[i];   // (i + 1, i)
      }
    }
    return true;
  }
};

// f(x) = sum_i a_i x_i^2 + c with the README quick-start operation order
// (README.md:21-28: `5*x[0]*x[0] + 100*x[1]*x[1] + 5`, grad (10*x0, 200*x1)).
struct DiagQuadratic final : Objective {
  std::vector<double> a;
  double c = 0.0;
  double eval(const double* x, double* g, int n, const Reducer& red) const override {
    if (red.fma()) {  // DiagQuadraticObjective::eval_fma: the lane's terms accumulate as fma(a_i x_i, x_i, previous)
      std::vector<double> ax(n);
      for (int i = 0; i < n; ++i) {
        ax[i] = a[i] * x[i];
        g[i] = (2.0 * a[i]) * x[i];
      }
      return red.dot(ax.data(), x, n) + c;
    }
    Scratch term(n);
    term[0] = 0.0;
    for (int i = 0; i < n; ++i) {
      term[i] = (a[i] * x[i]) * x[i];
      g[i] = (2.0 * a[i]) * x[i];
    }
    return red.sum(term, n) + c;
  }
};

// Ridge least squares  f(x) = ||A x - y_b||^2 + lambda ||x||^2  with a shared A (rows x n,
// row major) and one right-hand side per problem: the reference README's composition
// `SquaredError(A, y) + lambda * L2Reg(n)` (README.md:122-167) evaluated through
// AddExpression (function_expressions.h:115-135: value fx_f + fx_g, gradient grad_f + grad_g,
// Hessian hess_f + hess_g) and MulExpression (:229-246: c * fx, c * grad_f, c * hess_f):
//   r = A x - y;  fx_f = r.r;  grad_f = 2 A^T r;  fx_g = lambda * (x.x);  grad_g = lambda * (2 x)
// The two matrix-vector products are ascending multiply-then-add sums (what the README functors
// compute over oracle/eigen_shim; `2 * A.transpose() * r` scales by an exact factor 2, so
// sum_i (2 A_ij) r_i == 2 sum_i A_ij r_i bit for bit).  r.r and x.x follow the Reducer policy.
// The Hessian is constant: diag_j = sum_i (2 A_ij) A_ij + lambda * 2 (README SquaredError / L2Reg
// `hess`), used by the Second-mode path of Lbfgs (lbfgs.h:116-139) through hessian_diagonal().
struct SquaredErrorRidge final : Objective {
  int rows = 0;
  double lambda = 0.0;
  const double* A = nullptr;        // rows x n, row major
  const double* y_all = nullptr;    // [B][rows]
  const double* y = nullptr;        // current problem
  // Twin of the matrix-core kernel (csrc/ridge_mfma_kernel.hpp, objective id 3): the two matrix-vector
  // products as ascending fused-multiply-add chains from 0 — what v_mfma_f64_16x16x4_f64 computes
  // when the tiles are walked in natural order (A^T r as two chains over rows 0..63 and 64..127, added).
  // Everything else is unchanged.
  bool fma_chains = false;
  // Twin of the normal-equation form (csrc/ridge_gram.hpp, objective id 5): f = x^T G x - 2 c^T x + y^T y with
  //   G = A^T A + lambda I   entry (i, j): ascending fused chain over the rows from 0, lambda added to the diagonal
  //   c = A^T y              ascending fused chain over the rows from 0 (what the matrix-core pre-pass accumulates)
  //   yy = y . y             four interleaved fused chains (rows r = k mod 4), added (s0 + s1) + (s2 + s3)
  // and per evaluation  t_i = chain_j fma(G[j][i], x_j),  h = t - c,  grad = 2 h,  f = dot(x, h - c) + yy  (the
  // Reducer's fused dot).  Chains run over the padded width; the padding is zero.
  bool gram = false;
  mutable std::vector<double> gram_G_, gram_c_;
  mutable double gram_yy_ = 0.0;
  mutable int gram_n_ = -1;
  mutable const double* gram_y_ = nullptr;
  // Own-matrix form (objective ids 6 / 7): every problem has ITS OWN A — what the reference's README builds when a
  // program makes one `SquaredError(A, y)` per data set (README.md:126-160).  per-problem row = A_b (rows x n, row major)
  // followed by y_b; own_stride = doubles per row (0 = shared A, the forms above).
  int64_t own_stride = 0;
  void set_problem(int64_t b) override {
    if (own_stride > 0) {
      A = y_all + b * own_stride;
      y = A + static_cast<int64_t>(rows) * own_n;
      gram_n_ = -1;   // a new matrix: the Gram cache is rebuilt
    } else {
      y = y_all + b * rows;
    }
  }
  int own_n = 0;
  double eval_gram(const double* x, double* g, int n, const Reducer& red) const {
    if (gram_n_ != n) {
      gram_G_.assign(static_cast<size_t>(n) * n, 0.0);
      for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
          double acc = 0.0;
          for (int r = 0; r < rows; ++r) acc = std::fma(A[static_cast<size_t>(r) * n + i], A[static_cast<size_t>(r) * n + j], acc);
          if (i == j) acc = acc + lambda;
          gram_G_[static_cast<size_t>(i) * n + j] = gram_G_[static_cast<size_t>(j) * n + i] = acc;
        }
      gram_n_ = n;
      gram_y_ = nullptr;
    }
    if (gram_y_ != y) {
      gram_c_.assign(n, 0.0);
      for (int j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int r = 0; r < rows; ++r) acc = std::fma(y[r], A[static_cast<size_t>(r) * n + j], acc);
        gram_c_[j] = acc;
      }
      double s[4] = {0.0, 0.0, 0.0, 0.0};
      for (int r = 0; r < rows; ++r) s[r & 3] = std::fma(y[r], y[r], s[r & 3]);
      gram_yy_ = (s[0] + s[1]) + (s[2] + s[3]);
      gram_y_ = y;
    }
    std::vector<double> u(n);
    for (int i = 0; i < n; ++i) {
      double t = 0.0;
      for (int j = 0; j < n; ++j) t = std::fma(gram_G_[static_cast<size_t>(j) * n + i], x[j], t);
      const double h = t - gram_c_[i];
      g[i] = 2.0 * h;
      u[i] = h - gram_c_[i];
    }
    return red.dot(x, u.data(), n) + gram_yy_;
  }
  double eval(const double* x, double* g, int n, const Reducer& red) const override {
    if (gram) return eval_gram(x, g, n, red);
    double r[1024], rr[1024];
    for (int i = 0; i < rows; ++i) {
      double acc = 0.0;
      if (fma_chains) {
        for (int j = 0; j < n; ++j) acc = std::fma(A[static_cast<size_t>(i) * n + j], x[j], acc);
      } else {
        for (int j = 0; j < n; ++j) acc = acc + A[static_cast<size_t>(i) * n + j] * x[j];
      }
      r[i] = acc - y[i];
      rr[i] = r[i] * r[i];
    }
    int wr = 1;
    while (wr < rows) wr <<= 1;
    const double f1 = red.sum(rr, rows, wr);
    const double xx = red.dot(x, x, n);
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      if (fma_chains) {  // two chains, rows 0..63 and 64..: the kernel's two halves of the row range
        double hi = 0.0;
        for (int i = 0; i < rows && i < 64; ++i) acc = std::fma(A[static_cast<size_t>(i) * n + j], r[i], acc);
        for (int i = 64; i < rows; ++i) hi = std::fma(A[static_cast<size_t>(i) * n + j], r[i], hi);
        acc = acc + hi;
      } else {
        for (int i = 0; i < rows; ++i) acc = acc + A[static_cast<size_t>(i) * n + j] * r[i];
      }
      g[j] = 2.0 * acc + lambda * (2.0 * x[j]);
    }
    return f1 + lambda * xx;
  }
  // progress.h:208 on the constant Hessian 2 A^T A + 2 lambda I (README `hess` functors, ascending rows): Frobenius
  // norms, inverse by LU with partial pivoting and column-by-column solves (what a loop-based Eigen computes)
  double hessian_condition(int n) const {
    std::vector<double> H(static_cast<size_t>(n) * n, 0.0);
    for (int j = 0; j < n; ++j)
      for (int k = 0; k < n; ++k) {
        double acc = 0.0;
        for (int i = 0; i < rows; ++i)
          acc = (i == 0) ? (2.0 * A[j]) * A[k]
                         : acc + (2.0 * A[static_cast<size_t>(i) * n + j]) * A[static_cast<size_t>(i) * n + k];
        H[static_cast<size_t>(j) * n + k] = acc + ((j == k) ? lambda * 2.0 : 0.0);
      }
    std::vector<double> lu = H, inv(H.size(), 0.0), col(n);
    std::vector<int> piv(n);
    auto at = [&](std::vector<double>& m, int i, int j) -> double& { return m[static_cast<size_t>(i) * n + j]; };
    for (int k = 0; k < n; ++k) {
      int p = k;
      double best = std::fabs(at(lu, k, k));
      for (int i = k + 1; i < n; ++i)
        if (std::fabs(at(lu, i, k)) > best) { best = std::fabs(at(lu, i, k)); p = i; }
      piv[k] = p;
      if (best != 0.0) {
        if (p != k) for (int j = 0; j < n; ++j) std::swap(at(lu, k, j), at(lu, p, j));
        for (int i = k + 1; i < n; ++i) at(lu, i, k) = at(lu, i, k) / at(lu, k, k);
      }
      for (int j = k + 1; j < n; ++j)
        for (int i = k + 1; i < n; ++i) at(lu, i, j) = at(lu, i, j) - at(lu, i, k) * at(lu, k, j);
    }
    for (int c = 0; c < n; ++c) {
      std::fill(col.begin(), col.end(), 0.0);
      col[c] = 1.0;
      for (int k = 0; k < n; ++k) std::swap(col[k], col[piv[k]]);
      for (int j = 0; j < n; ++j) for (int i = j + 1; i < n; ++i) col[i] = col[i] - col[j] * at(lu, i, j);
      for (int j = n - 1; j >= 0; --j) {
        col[j] = col[j] / at(lu, j, j);
        for (int i = 0; i < j; ++i) col[i] = col[i] - col[j] * at(lu, i, j);
      }
      for (int i = 0; i < n; ++i) at(inv, i, c) = col[i];
    }
    double sh = 0.0, si = 0.0;
    for (size_t t = 0; t < H.size(); ++t) { sh += H[t] * H[t]; si += inv[t] * inv[t]; }
    return std::sqrt(sh) * std::sqrt(si);
  }
  std::vector<double> hessian_diagonal(int n) const {
    std::vector<double> d(n);
    for (int j = 0; j < n; ++j) {
      double acc = 0.0;
      for (int i = 0; i < rows; ++i)
        acc = (i == 0) ? (2.0 * A[j]) * A[j]
                       : acc + (2.0 * A[static_cast<size_t>(i) * n + j]) * A[static_cast<size_t>(i) * n + j];
      d[j] = acc + lambda * 2.0;
    }
    return d;
  }
};

// Soft-margin SVM primal with a squared hinge loss — the functor of the reference's
// src/examples/svm_primal_lbfgs.cc:35-103, twin of the USER device objective
// examples/user_objective_svm/svm_squared_hinge.hpp (the worked example of the user-objective build path):
//   f(w, b) = 0.5 ||w||^2 + C sum_i max(0, 1 - y_i (x_i . w + b))^2,   x = (w, b), n = d + 1.
// Order of operations as the reference's Eigen expressions evaluate over a loop-based Eigen: `features * w` ascending
// columns, `+ b`, `labels * scores`, `(1 - margins).max(0)`, `slacks.squaredNorm()` ascending,
// `-2 * slacks * labels`, `features^T * weighted` ascending rows, `weighted.sum()` ascending.  Only w.squaredNorm()
// follows the Reducer policy (the device reduces the coordinates of a problem with its butterfly).
struct SvmSquaredHinge final : Objective {
  int N = 0, d = 0;
  double C = 1.0;
  const double* X = nullptr;  // N x d, row major
  const double* y = nullptr;  // N
  double eval(const double* x, double* g, int n, const Reducer& red) const override {
    (void)n;
    std::vector<double> ws(static_cast<size_t>(N));
    double hinge = 0.0;
    for (int i = 0; i < N; ++i) {
      const double* row = X + static_cast<size_t>(i) * d;
      double score = row[0] * x[0];
      for (int j = 1; j < d; ++j) score = score + row[j] * x[j];
      score = score + x[d];
      const double t = 1.0 - y[i] * score;
      const double slack = (t < 0.0) ? 0.0 : t;
      ws[i] = (-2.0 * slack) * y[i];
      hinge = (i == 0) ? slack * slack : hinge + slack * slack;
    }
    const double ww = red.dot(x, x, d);
    for (int j = 0; j < d; ++j) {
      double acc = X[j] * ws[0];
      for (int i = 1; i < N; ++i) acc = acc + X[static_cast<size_t>(i) * d + j] * ws[i];
      g[j] = x[j] + C * acc;
    }
    double acc = ws[0];
    for (int i = 1; i < N; ++i) acc = acc + ws[i];
    g[d] = C * acc;
    return 0.5 * ww + C * hinge;
  }
};

// Dual soft-margin SVM — the functor of the reference's src/examples/svm_dual_lbfgsb.cc:36-60, twin of the USER device
// objective examples/user_objective_svm_dual/svm_dual.hpp:  f(alpha) = 0.5 alpha^T Q alpha - 1^T alpha,  grad = Q alpha - 1.
// `kernel_matrix * alpha`: ascending columns per row (one fused chain under the fused policies); `alpha.dot(q_alpha)`
// and `alpha.sum()` (here alpha . 1) follow the Reducer policy.
struct SvmDual final : Objective {
  int ns = 0;
  const double* Q = nullptr;  // ns x ns, row major, symmetric
  double eval(const double* x, double* g, int n, const Reducer& red) const override {
    std::vector<double> q(static_cast<size_t>(n)), ones(static_cast<size_t>(n), 1.0);
    for (int i = 0; i < n; ++i) {
      const double* row = Q + static_cast<size_t>(i) * n;
      double acc = row[0] * x[0];
      for (int j = 1; j < n; ++j) acc = red.madd(row[j], x[j], acc);
      q[static_cast<size_t>(i)] = acc;
      g[i] = acc - 1.0;
    }
    const double aq = red.dot(x, q.data(), n);
    const double sa = red.dot(x, ones.data(), n);
    return 0.5 * aq - sa;
  }
};

// ---------------------------------------------------------------------------
// solver/progress.h:37-47
// ---------------------------------------------------------------------------
enum Status : int {
  NotStarted = -1,
  Continue = 0,
  IterationLimit = 1,
  XDeltaViolation = 2,
  FDeltaViolation = 3,
  GradientNormViolation = 4,
  HessianConditionViolation = 5,
  Finished = 6
};

// The stopping fields of solver/progress.h:87-136 that matter for a
// First-mode FunctionState.
struct Stopping {
  uint64_t num_iterations = 10000;
  double x_delta = 1e-9;
  int x_delta_violations = 1;
  double f_delta = 0.0;
  int f_delta_violations = 1;
  bool f_delta_relative = false;
  double gradient_norm = 1e-5;
  bool gradient_norm_relative = true;
  int past = 3;
  double past_delta = 1e-6;
  double condition_hessian = 0.0;  // Second-mode functions only (:110, :318-325); 0 = off
};
// solver/progress.h:353-431 (non-CPPOPT_SWEEP branch)
inline Stopping DefaultStopping() { return Stopping{}; }
// solver/progress.h:456-464
inline Stopping ConservativeStopping() {
  Stopping s;
  s.gradient_norm = 5e-6;
  s.past = 5;
  s.past_delta = 1e-10;
  return s;
}

struct State {  // function_base.h:297-332
  std::vector<double> x;
  double value = 0.0;
  std::vector<double> gradient;
};

struct Progress {  // solver/progress.h:82-140
  uint64_t num_iterations = 0;
  double x_delta = 0.0;
  int x_delta_violations = 0;
  double f_delta = 0.0;
  int f_delta_violations = 0;
  double gradient_norm = 0.0;
  Status status = NotStarted;
  std::vector<double> past_f_ring;
  int past_f_pos = 0;
  double condition_hessian = 0.0;  // :110; set by Update for Second-mode functions

  // solver/progress.h:153-327, FunctionState branch.  second_mode_condition: NaN for First-mode functions, else the
  // value :203-210 computes (`current_hessian.norm() * current_hessian.inverse().norm()`; constant Hessians only).
  void Update(const State& prev, const State& cur, const Stopping& stop,
              double second_mode_condition = std::numeric_limits<double>::quiet_NaN()) {
    const int n = static_cast<int>(cur.x.size());
    const double previous_value = prev.value;
    const double current_value = cur.value;
    num_iterations++;                                          // :188
    f_delta = std::fabs(current_value - previous_value);       // :189
    {                                                          // :190
      double m = 0.0;
      for (int i = 0; i < n; ++i) {
        const double t = std::fabs(cur.x[i] - prev.x[i]);
        if (m < t) m = t;
      }
      x_delta = m;
    }
    gradient_norm = Reducer::amax(cur.gradient.data(), n);     // :195
    const bool second_mode = (second_mode_condition == second_mode_condition);   // not NaN
    if (second_mode) condition_hessian = second_mode_condition;  // :203-210
    if ((stop.num_iterations > 0) && (num_iterations > stop.num_iterations)) {  // :212-216
      status = IterationLimit;
      return;
    }
    if ((stop.x_delta > 0) && (x_delta < stop.x_delta)) {      // :254-262
      x_delta_violations++;
      if (x_delta_violations >= stop.x_delta_violations) {
        status = XDeltaViolation;
        return;
      }
    } else {
      x_delta_violations = 0;
    }
    if ((stop.f_delta > 0) &&                                  // :263-277
        (f_delta < stop.f_delta * (stop.f_delta_relative
                                       ? std::max({std::fabs(current_value),
                                                   std::fabs(previous_value), 1.0})
                                       : 1.0))) {
      f_delta_violations++;
      if (f_delta_violations >= stop.f_delta_violations) {
        status = FDeltaViolation;
        return;
      }
    } else {
      f_delta_violations = 0;
    }
    if (stop.past > 0) {                                       // :280-298
      const int p = stop.past;
      if (static_cast<int>(past_f_ring.size()) != p) {
        past_f_ring.assign(p, current_value);
        past_f_pos = 0;
      }
      if (static_cast<int>(num_iterations) > p) {
        const double past_f = past_f_ring[past_f_pos];
        const double rate =
            std::fabs(past_f - current_value) / std::max(1.0, std::fabs(current_value));
        if (rate < stop.past_delta) {
          status = FDeltaViolation;
          return;
        }
      }
      past_f_ring[past_f_pos] = current_value;
      past_f_pos = (past_f_pos + 1) % p;
    }
    if (stop.gradient_norm > 0) {                              // :299-317
      const double scale = stop.gradient_norm_relative
                               ? std::max(1.0, Reducer::amax(cur.x.data(), n))
                               : 1.0;
      if (gradient_norm < stop.gradient_norm * scale) {
        status = GradientNormViolation;
        return;
      }
    }
    if (second_mode && (stop.condition_hessian > 0) && (condition_hessian > stop.condition_hessian)) {  // :318-325
      status = HessianConditionViolation;
      return;
    }
    status = Continue;                                         // :326
  }
};

// ---------------------------------------------------------------------------
// linesearch/more_thuente.h
// ---------------------------------------------------------------------------
struct MoreThuente {
  static double max_abs(double x, double y, double z) {        // :409-411
    return std::max(std::fabs(x), std::max(std::fabs(y), std::fabs(z)));
  }

  // :261-407.  stpmin/stpmax are the bracket bounds handed in by cvsrch.
  static int cstep(double& stx, double& fx, double& dx, double& sty, double& fy,
                   double& dy, double& stp, double& fp, double& dp, bool& brackt,
                   double& stpmin, double& stpmax, int& info) {
    info = 0;
    bool bound = false;
    if ((brackt && ((stp <= std::min(stx, sty)) || (stp >= std::max(stx, sty)))) ||  // :271-275
        (dx * (stp - stx) >= 0.0) || (stpmax < stpmin)) {
      return -1;
    }
    const double sgnd = dp * (dx / std::fabs(dx));             // :277
    double stpf = 0, stpc = 0, stpq = 0;
    if (fp > fx) {                                             // Case 1 :283-301
      info = 1;
      bound = true;
      const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
      const double s = max_abs(theta, dx, dp);
      double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
      if (stp < stx) gamma = -gamma;
      const double p = (gamma - dx) + theta;
      const double q = ((gamma - dx) + gamma) + dp;
      const double r = p / q;
      stpc = stx + r * (stp - stx);
      stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
      if (std::fabs(stpc - stx) < std::fabs(stpq - stx))
        stpf = stpc;
      else
        stpf = stpc + (stpq - stpc) / 2;
      brackt = true;
    } else if (sgnd < 0.0) {                                   // Case 2 :302-320
      info = 2;
      bound = false;
      const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
      const double s = max_abs(theta, dx, dp);
      double gamma = s * std::sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
      if (stp > stx) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + dx;
      const double r = p / q;
      stpc = stp + r * (stx - stp);
      stpq = stp + (dp / (dp - dx)) * (stx - stp);
      if (std::fabs(stpc - stp) > std::fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      brackt = true;
    } else if (std::fabs(dp) < std::fabs(dx)) {                // Case 3 :321-354
      info = 3;
      bound = true;
      const double theta = 3 * (fx - fp) / (stp - stx) + dx + dp;
      const double s = max_abs(theta, dx, dp);
      double gamma =
          s * std::sqrt(std::max(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
      if (stp > stx) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = (gamma + (dx - dp)) + gamma;
      const double r = p / q;
      if ((r < 0.0) & (gamma != 0.0)) {
        stpc = stp + r * (stx - stp);
      } else if (stp > stx) {
        stpc = stpmax;
      } else {
        stpc = stpmin;
      }
      stpq = stp + (dp / (dp - dx)) * (stx - stp);
      if (brackt) {
        stpf = (std::fabs(stp - stpc) < std::fabs(stp - stpq)) ? stpc : stpq;
      } else {
        stpf = (std::fabs(stp - stpc) > std::fabs(stp - stpq)) ? stpc : stpq;
      }
    } else {                                                   // Case 4 :355-375
      info = 4;
      bound = false;
      if (brackt) {
        const double theta = 3 * (fp - fy) / (sty - stp) + dy + dp;
        const double s = max_abs(theta, dy, dp);
        double gamma = s * std::sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
        if (stp > sty) gamma = -gamma;
        const double p = (gamma - dp) + theta;
        const double q = ((gamma - dp) + gamma) + dy;
        const double r = p / q;
        stpc = stp + r * (sty - stp);
        stpf = stpc;
      } else if (stp > stx) {
        stpf = stpmax;
      } else {
        stpf = stpmin;
      }
    }
    if (fp > fx) {                                             // :377-391
      sty = stp;
      fy = fp;
      dy = dp;
    } else {
      if (sgnd < 0.0) {
        sty = stx;
        fy = fx;
        dy = dx;
      }
      stx = stp;
      fx = fp;
      dx = dp;
    }
    stpf = std::clamp(stpf, stpmin, stpmax);                   // :393-394
    stp = stpf;
    if (brackt & bound) {                                      // :396-404
      if (sty > stx) {
        stp = std::min(stx + 0.66 * (sty - stx), stp);
      } else {
        stp = std::max(stx + 0.66 * (sty - stx), stp);
      }
    }
    return 0;
  }

  // :137-256.  x,f,g are in/out; returns like the reference (value ignored by
  // the caller).  nfev is incremented per objective evaluation.
  static int cvsrch(const Objective& function, const Reducer& red, std::vector<double>* x,
                    double* f, std::vector<double>* g, double* stp,
                    const std::vector<double>& s, uint64_t* nfev_total) {
    const int n = static_cast<int>(x->size());
    int info = 0;
    int infoc = 1;
    constexpr double xtol = 1e-15;
    constexpr double ftol = 1e-4;
    constexpr double gtol = 0.9;
    constexpr double stpmin = 1e-15;
    constexpr double stpmax = 1e15;
    constexpr double xtrapf = 4;
    constexpr int maxfev = 20;
    int nfev = 0;

    const double dginit = red.dot(g->data(), s.data(), n);     // :151
    if (dginit >= 0.0) return -1;                              // :152-156

    bool brackt = false;
    bool stage1 = true;
    const double finit = *f;
    const double dgtest = ftol * dginit;
    double width = stpmax - stpmin;
    double width1 = 2.0 * width;
    const std::vector<double> wa = *x;

    double stx = 0.0, fx = finit, dgx = dginit;
    double sty = 0.0, fy = finit, dgy = dginit;
    double stmin, stmax;

    while (true) {
      if (brackt) {                                            // :179-185
        stmin = std::min(stx, sty);
        stmax = std::max(stx, sty);
      } else {
        stmin = stx;
        stmax = *stp + xtrapf * (*stp - stx);
      }
      *stp = std::clamp(*stp, stpmin, stpmax);                 // :188
      if ((brackt && ((*stp <= stmin) || (*stp >= stmax))) || (nfev >= maxfev - 1) ||  // :191-195
          (infoc == 0) || (brackt && ((stmax - stmin) <= (xtol * stmax)))) {
        *stp = stx;
      }
      for (int i = 0; i < n; ++i) (*x)[i] = red.madd(*stp, s[i], wa[i]);   // :198  wa + stp * s
      *f = function.eval(x->data(), g->data(), n, red);        // :199
      nfev++;
      if (nfev_total) ++*nfev_total;
      const double dg = red.dot(g->data(), s.data(), n);       // :201
      const double ftest1 = finit + *stp * dgtest;

      if ((brackt & ((*stp <= stmin) | (*stp >= stmax))) | (infoc == 0)) info = 6;   // :205-216
      if ((*stp == stpmax) & (*f <= ftest1) & (dg <= dgtest)) info = 5;
      if ((*stp == stpmin) & ((*f > ftest1) | (dg >= dgtest))) info = 4;
      if (nfev >= maxfev) info = 3;
      if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
      if ((*f <= ftest1) & (std::fabs(dg) <= gtol * (-dginit))) info = 1;
      if (info != 0) return -1;                                // :219

      if (stage1 & (*f <= ftest1) & (dg >= std::min(ftol, gtol) * dginit)) stage1 = false;  // :221-223

      if (stage1 & (*f <= fx) & (*f > ftest1)) {               // :225-239
        double fm = *f - *stp * dgtest;
        double fxm = fx - stx * dgtest;
        double fym = fy - sty * dgtest;
        double dgm = dg - dgtest;
        double dgxm = dgx - dgtest;
        double dgym = dgy - dgtest;
        cstep(stx, fxm, dgxm, sty, fym, dgym, *stp, fm, dgm, brackt, stmin, stmax, infoc);
        fx = fxm + stx * dgtest;
        fy = fym + sty * dgtest;
        dgx = dgxm + dgtest;
        dgy = dgym + dgtest;
      } else {                                                 // :240-244
        double fcur = *f, dgcur = dg;
        cstep(stx, fx, dgx, sty, fy, dgy, *stp, fcur, dgcur, brackt, stmin, stmax, infoc);
      }
      if (brackt) {                                            // :246-252
        if (std::fabs(sty - stx) >= 0.66 * width1) *stp = stx + 0.5 * (sty - stx);
        width1 = width;
        width = std::fabs(sty - stx);
      }
    }
    return 0;
  }

  // :120-135 (State-returning overload).
  static State Search(const State& start, const std::vector<double>& search_direction,
                      const Objective& function, const Reducer& red, double alpha_init,
                      uint64_t* nfev_total) {
    double alpha = alpha_init;
    double f = start.value;
    std::vector<double> g = start.gradient;
    std::vector<double> xx = start.x;
    cvsrch(function, red, &xx, &f, &g, &alpha, search_direction, nfev_total);
    State out;
    out.x = std::move(xx);
    out.value = f;
    out.gradient = std::move(g);
    return out;
  }
};

// ---------------------------------------------------------------------------
// linesearch/hager_zhang.h — the alternative LineSearch template argument
// (Hager & Zhang 2006, via LineSearches.jl; constants and stage order :283-552).
// Every sampled point is recorded as (alpha, phi, dphi); brackets refer to samples by index, and
// two of the reference's decisions depend on index identity (Secant2's "which end moved", :257-259),
// so indices are kept here as well.
// ---------------------------------------------------------------------------
struct HagerZhang {
  struct Sample { double alpha, phi, dphi; };

  struct Run {  // one hzls call: the trajectory, the work vectors and the evaluation counter
    const Objective& function;
    const Reducer& red;
    const std::vector<double>& x0;
    const std::vector<double>& s;
    std::vector<double> xa, gx;       // the most recent evaluation point and its gradient
    std::vector<Sample> h;
    uint64_t* nfev;
    double phi_0 = 0, dphi_0 = 0, phi_lim = 0;
    static constexpr double delta = 1.0 / 10.0, sigma = 9.0 / 10.0;   // :286-287

    // phi(alpha) = f(x + alpha s), dphi = g(x + alpha s).s  (:150-157)
    Sample evaluate(double alpha) {
      const int n = static_cast<int>(x0.size());
      for (int j = 0; j < n; ++j) xa[j] = x0[j] + alpha * s[j];
      Sample r;
      r.alpha = alpha;
      r.phi = function.eval(xa.data(), gx.data(), n, red);
      if (nfev) ++*nfev;
      r.dphi = red.dot(gx.data(), s.data(), n);
      return r;
    }
    int push(const Sample& r) {
      h.push_back(r);
      return static_cast<int>(h.size()) - 1;
    }
    // T1 / T2 acceptance (:128-140)
    bool wolfe(const Sample& c) const {
      const bool w1 = (delta * dphi_0 >= (c.phi - phi_0) / c.alpha) && (c.dphi >= sigma * dphi_0);
      const bool w2 = ((2 * delta - 1) * dphi_0 >= c.dphi) && (c.dphi >= sigma * dphi_0) && (c.phi <= phi_lim);
      return w1 || w2;
    }
    struct Bracket { int ia, ib; bool hit; };
    // stage U3, theta = 1/2 (:186-214)
    Bracket bisect(int ia, int ib) {
      double a = h[ia].alpha, b = h[ib].alpha;
      while (b - a > std::numeric_limits<double>::epsilon() * b) {
        const double d = (a + b) / 2.0;
        const Sample r = evaluate(d);
        const int id = push(r);
        if (wolfe(r)) return {ia, id, true};
        if (r.dphi >= 0.0) return {ia, id, false};
        if (r.phi <= phi_lim) {
          a = d;
          ia = id;
        } else {
          b = d;
          ib = id;
        }
      }
      return {ia, ib, false};
    }
    // stages U0-U3 (:163-182)
    Bracket update(int ia, int ib, int ic) {
      const double a = h[ia].alpha, b = h[ib].alpha, c = h[ic].alpha;
      if (c < a || c > b) return {ia, ib, false};
      if (h[ic].dphi >= 0.0) return {ia, ic, false};
      if (h[ic].phi <= phi_lim) return {ic, ib, false};
      return bisect(ia, ic);
    }
    static double secant(double a, double b, double da, double db) {   // :143-146
      return (a * db - b * da) / (db - da);
    }
    // stages S1-S4 (:218-277); on a hit ia == ib == the accepted sample
    Bracket secant2(int ia, int ib) {
      double c = secant(h[ia].alpha, h[ib].alpha, h[ia].dphi, h[ib].dphi);
      if (!std::isfinite(c)) c = (h[ia].alpha + h[ib].alpha) / 2.0;
      const Sample r = evaluate(c);
      const int ic = push(r);
      if (wolfe(r)) return {ic, ic, true};
      const Bracket u = update(ia, ib, ic);
      if (u.hit) return {u.ib, u.ib, true};
      const double A = h[u.ia].alpha, B = h[u.ib].alpha;
      double c2 = c;
      const bool moved_b = (u.ib == ic), moved_a = (u.ia == ic);
      if (moved_b)
        c2 = secant(h[ib].alpha, h[u.ib].alpha, h[ib].dphi, h[u.ib].dphi);
      else if (moved_a)
        c2 = secant(h[ia].alpha, h[u.ia].alpha, h[ia].dphi, h[u.ia].dphi);
      if ((moved_a || moved_b) && A <= c2 && c2 <= B) {
        const Sample r2 = evaluate(c2);
        const int ic2 = push(r2);
        if (wolfe(r2)) return {ic2, ic2, true};
        const Bracket u2 = update(u.ia, u.ib, ic2);
        if (u2.hit) return {u2.ib, u2.ib, true};
        return {u2.ia, u2.ib, false};
      }
      return {u.ia, u.ib, false};
    }
  };

  // hzls (:282-548).  x, f, g, stp are in/out like cvsrch; returns 0 or -1 (ignored by callers).
  static int hzls(const Objective& function, const Reducer& red, std::vector<double>* x, double* f,
                  std::vector<double>* g, double* stp, const std::vector<double>& s, uint64_t* nfev) {
    constexpr double epsilon_k = 1e-6, gamma = 0.66, rho = 5.0, psi3 = 0.1;   // :288-291
    constexpr int maxlinesearch = 50, iterfinitemax = 60;
    const int n = static_cast<int>(x->size());
    const std::vector<double> x_start = *x;
    Run run{function, red, x_start, s, *x, *g, {}, nfev};
    run.phi_0 = *f;
    run.dphi_0 = red.dot(g->data(), s.data(), n);
    if (run.dphi_0 >= 0.0) return -1;                                        // :302
    run.phi_lim = run.phi_0 + epsilon_k * std::fabs(run.phi_0);
    run.push({0.0, run.phi_0, run.dphi_0});

    double best_alpha = 0.0, best_phi = run.phi_0;                           // :319-332
    std::vector<double> best_x = *x, best_g = *g;
    auto note_best = [&](const Sample& r) {
      if (r.alpha > 0.0 && r.phi < best_phi) {
        best_alpha = r.alpha;
        best_phi = r.phi;
        best_x = run.xa;
        best_g = run.gx;
      }
    };
    auto finish_last = [&](const Sample& w) {   // accept the most recently evaluated point
      *x = run.xa;
      *f = w.phi;
      *g = run.gx;
      *stp = w.alpha;
      return 0;
    };
    auto finish_best = [&]() {
      if (best_alpha > 0.0) {
        *x = best_x;
        *f = best_phi;
        *g = best_g;
        *stp = best_alpha;
        return 0;
      }
      *stp = 0.0;
      return -1;
    };
    auto finite = [](const Sample& r) { return std::isfinite(r.phi) && std::isfinite(r.dphi); };

    double c = *stp;                                                         // :336-337
    if (!(c > 0.0)) c = 1.0;
    Sample ec = run.evaluate(c);
    int iterfinite = 0;                                                      // stage I0 (:344-351)
    while (!finite(ec) && iterfinite < iterfinitemax) {
      c *= psi3;
      ec = run.evaluate(c);
      ++iterfinite;
    }
    if (!finite(ec)) {
      *stp = 0.0;
      return -1;
    }
    run.push(ec);
    note_best(ec);
    if (run.wolfe(ec)) return finish_last(ec);

    bool bracketed = false;                                                  // stages B0-B3 (:368-441)
    int ia = 0, ib = 1, iter = 1;
    while (!bracketed && iter < maxlinesearch) {
      const Sample last = run.h.back();
      if (last.dphi >= 0.0) {                                                // B1
        ib = static_cast<int>(run.h.size()) - 1;
        for (int i = ib - 1; i >= 0; --i) {
          if (run.h[i].phi <= run.phi_lim) {
            ia = i;
            break;
          }
        }
        bracketed = true;
      } else if (last.phi > run.phi_lim) {                                   // B2
        ib = static_cast<int>(run.h.size()) - 1;
        ia = 0;
        const Run::Bracket r = run.bisect(ia, ib);
        if (r.hit) return finish_last(run.h[r.ib]);
        ia = r.ia;
        ib = r.ib;
        bracketed = true;
      } else {                                                               // B3
        c *= rho;
        ec = run.evaluate(c);
        iterfinite = 0;
        while (!finite(ec) && iterfinite < iterfinitemax) {
          c = (run.h.back().alpha + c) / 2.0;
          ec = run.evaluate(c);
          ++iterfinite;
        }
        if (!finite(ec)) return finish_best();
        run.push(ec);
        note_best(ec);
        if (run.wolfe(ec)) return finish_last(ec);
      }
      ++iter;
    }
    if (!bracketed) return finish_best();                                    // :443-454

    while (iter < maxlinesearch) {                                           // :458-535
      const double a = run.h[ia].alpha, b = run.h[ib].alpha;
      if (b - a <= std::numeric_limits<double>::epsilon() * b) {
        if (a > 0.0) {
          ec = run.evaluate(a);
          return finish_last(ec);
        }
        return finish_best();
      }
      const Run::Bracket r = run.secant2(ia, ib);
      if (r.hit) return finish_last(run.h[r.ia]);
      const double A = run.h[r.ia].alpha, B = run.h[r.ib].alpha;
      if (B - A < gamma * (b - a)) {
        ia = r.ia;
        ib = r.ib;
      } else {                                                               // stage L2
        const double cm = (A + B) / 2.0;
        const Sample rm = run.evaluate(cm);
        const int ic = run.push(rm);
        note_best(rm);
        if (run.wolfe(rm)) return finish_last(rm);
        const Run::Bracket u = run.update(r.ia, r.ib, ic);
        if (u.hit) return finish_last(run.h[u.ib]);
        ia = u.ia;
        ib = u.ib;
      }
      ++iter;
    }
    return finish_best();                                                    // :537-547
  }

  // State overload (:100-116), the one Lbfgs calls
  static State Search(const State& start, const std::vector<double>& direction, const Objective& function,
                      const Reducer& red, double alpha_init, uint64_t* nfev) {
    State out = start;
    double alpha = alpha_init;
    hzls(function, red, &out.x, &out.value, &out.gradient, &alpha, direction, nfev);
    return out;
  }
};

// ---------------------------------------------------------------------------
// solver/lbfgs.h
// ---------------------------------------------------------------------------
struct Lbfgs {
  int m = 10;
  Stopping stopping_progress;
  Reducer red;

  // private state, solver/lbfgs.h:306-323
  int n_ = 0;
  std::vector<double> S_, Y_;  // m columns of n
  std::vector<double> alpha_;
  size_t mem_count_ = 0, mem_pos_ = 0;
  double scaling_factor_ = 1;

  // Second-mode functions (lbfgs.h:116-139, :177-179): H_0 = diag(|H_ii| + eps)^-1 replaces the
  // scalar scaling_factor_ at the centre of the two-loop recursion.  Empty = First-mode path.
  // (Only constant Hessian diagonals are modelled: the reference re-evaluates f, g, H at the
  // unchanged iterate every step, which changes nothing but the evaluation count.)
  std::vector<double> hessian_diagonal;
  // ||H||_F ||H^-1||_F of that constant Hessian (progress.h:203-210), NaN = First mode
  double hessian_condition = std::numeric_limits<double>::quiet_NaN();
  // Second-mode functions with a NON-constant Hessian: the diagonal is taken from the objective at every iterate
  // (lbfgs.h:129-134: function(current.x, &g, &H), diag(H).cwiseAbs() + eps, cwiseInverse)
  bool hessian_from_objective = false;
  double last_condition = std::numeric_limits<double>::quiet_NaN();  // Progress::condition_hessian after the last Update
  // (the reference pays an n x n inverse per iteration for every Second-mode function; the twin does when the value is
  //  asked for — the stopping test is on, or a test wants the number — since it changes nothing else)
  bool track_condition = false;

  int linesearch = 0;  // LineSearch template argument (lbfgs.h:41): 0 MoreThuente, 1 HagerZhang

  // accounting (not in the reference): evaluations and sum of history depth
  uint64_t nfev = 0;
  uint64_t sum_k = 0;

  explicit Lbfgs(int m_in = 10, Stopping stop = DefaultStopping(), Reducer r = Reducer{})
      : m(m_in), stopping_progress(stop), red(r) {}

  void InitializeSolver(int n) {                               // :72-87
    n_ = n;
    S_.assign(static_cast<size_t>(m) * n, 0.0);
    Y_.assign(static_cast<size_t>(m) * n, 0.0);
    alpha_.assign(m, 0.0);
    mem_count_ = 0;
    mem_pos_ = 0;
    scaling_factor_ = 1;
  }

  State OptimizationStep(const Objective& function, const State& current) {  // :89-303
    const int n = n_;
    constexpr double eps = std::numeric_limits<double>::epsilon();
    const double relative_eps = eps * std::max(1.0, red.norm(current.x.data(), n));  // :93-95
    const std::vector<double>& g = current.gradient;
    std::vector<double> d = g;                                 // :145
    const int k = static_cast<int>(mem_count_);
    sum_k += static_cast<uint64_t>(k);

    for (int i = k - 1; i >= 0; i--) {                         // :157-171
      const int idx = static_cast<int>(mem_count_ < static_cast<size_t>(m) ? i : ((mem_pos_ + i) % m));
      const double* s = &S_[static_cast<size_t>(idx) * n];
      const double* y = &Y_[static_cast<size_t>(idx) * n];
      const double denom = red.dot(s, y, n);
      if (std::fabs(denom) < eps) continue;
      const double rho = 1.0 / denom;
      alpha_[i] = rho * red.dot(s, d.data(), n);
      for (int j = 0; j < n; ++j) d[j] = red.nmadd(alpha_[i], y[j], d[j]);   // d - alpha y
    }
    if (hessian_from_objective) {
      std::vector<double> h(n);
      function.hess_diag(current.x.data(), h.data(), n);
      for (int j = 0; j < n; ++j) {
        const double pre = 1.0 / (std::fabs(h[j]) + eps);
        d[j] = pre * d[j];
      }
    } else if (!hessian_diagonal.empty()) {                     // :126-131, :177-179
      for (int j = 0; j < n; ++j) {
        const double pre = 1.0 / (std::fabs(hessian_diagonal[j]) + eps);
        d[j] = pre * d[j];
      }
    } else {
      for (int j = 0; j < n; ++j) d[j] = d[j] * scaling_factor_;   // :181
    }
    for (int i = 0; i < k; i++) {                              // :185-196
      const int idx = static_cast<int>(mem_count_ < static_cast<size_t>(m) ? i : ((mem_pos_ + i) % m));
      const double* s = &S_[static_cast<size_t>(idx) * n];
      const double* y = &Y_[static_cast<size_t>(idx) * n];
      const double denom = red.dot(s, y, n);
      if (std::fabs(denom) < eps) continue;
      const double rho = 1.0 / denom;
      const double beta = rho * red.dot(y, d.data(), n);
      const double c = alpha_[i] - beta;
      for (int j = 0; j < n; ++j) d[j] = red.madd(s[j], c, d[j]);            // d + s c
    }

    double descent_direction = -red.dot(g.data(), d.data(), n);   // :199
    double alpha_init = 1.0;                                   // :207-213
    if (mem_count_ == 0) {
      const double dn = red.norm(d.data(), n);
      alpha_init = (dn > eps) ? 1.0 / dn : 1.0;
    }
    if (!std::isfinite(descent_direction) || descent_direction > -eps * relative_eps) {  // :214-224
      for (int j = 0; j < n; ++j) d[j] = -g[j];
      mem_count_ = 0;
      mem_pos_ = 0;
      const double gn = red.norm(g.data(), n);
      alpha_init = (gn > eps) ? 1.0 / gn : 1.0;
    }

    std::vector<double> neg_d(n);
    for (int j = 0; j < n; ++j) neg_d[j] = -d[j];
    State next = (linesearch == 1) ? HagerZhang::Search(current, neg_d, function, red, alpha_init, &nfev)
                                   : MoreThuente::Search(current, neg_d, function, red, alpha_init, &nfev);  // :231-232

    if (!std::isfinite(next.value)) return current;            // :239-241

    std::vector<double> s(n), y(n);                            // :248-249
    for (int j = 0; j < n; ++j) s[j] = next.x[j] - current.x[j];
    for (int j = 0; j < n; ++j) y[j] = next.gradient[j] - g[j];

    const double sy = red.dot(s.data(), y.data(), n);          // :265
    const double sy_threshold = eps * red.norm(s.data(), n) * red.norm(y.data(), n);  // :266
    if (sy > sy_threshold) {                                   // :267-280
      if (mem_count_ < static_cast<size_t>(m)) {
        std::copy(s.begin(), s.end(), S_.begin() + mem_count_ * n);
        std::copy(y.begin(), y.end(), Y_.begin() + mem_count_ * n);
        mem_count_++;
      } else {
        std::copy(s.begin(), s.end(), S_.begin() + mem_pos_ * n);
        std::copy(y.begin(), y.end(), Y_.begin() + mem_pos_ * n);
        mem_pos_ = (mem_pos_ + 1) % m;
      }
    }
    constexpr double fallback_value = 1e7;                     // :289-298
    const double yy = red.dot(y.data(), y.data(), n);
    if (yy > eps) {
      const double temp_scaling = red.dot(y.data(), s.data(), n) / yy;
      if (std::isfinite(temp_scaling) && std::fabs(temp_scaling) <= fallback_value) {
        scaling_factor_ = std::max(temp_scaling, eps);
      }
    }
    return next;
  }

  // solver/solver.h:181-224
  State Minimize(const Objective& function, const std::vector<double>& x0, Progress* progress_out) {
    const int n = static_cast<int>(x0.size());
    Progress solver_state;
    State cur;                                                 // :189-192
    cur.x = x0;
    cur.gradient.assign(n, 0.0);
    cur.value = function.eval(cur.x.data(), cur.gradient.data(), n, red);
    nfev = 1;
    sum_k = 0;
    InitializeSolver(n);                                       // :194
    do {                                                       // :196-220
      State prev = cur;
      cur = OptimizationStep(function, prev);
      double condition = hessian_diagonal.empty() ? std::numeric_limits<double>::quiet_NaN() : hessian_condition;
      if (hessian_from_objective && track_condition) {   // progress.h:203-210: function(current_x, nullptr, &H) in EVERY Update
        std::vector<double> H(static_cast<size_t>(n) * n);
        if (function.hess_full(cur.x.data(), H.data(), n)) condition = ShimHessianCondition(H, n);
      }
      solver_state.Update(prev, cur, stopping_progress, condition);
      last_condition = condition;
    } while (solver_state.status == Continue);
    if (progress_out) *progress_out = solver_state;
    return cur;
  }
};

// ---------------------------------------------------------------------------
// solver/bfgs.h — dense BFGS: the same driver, line searches and stopping
// tests as Lbfgs, with an explicit n x n inverse-Hessian approximation instead of the (s, y) ring.
// Matrix-vector products are row sums in ascending column order (what `-H * g` and `H * y` evaluate to
// in the reference) under both reduction policies; dots and norms follow the Reducer.
// ---------------------------------------------------------------------------
struct Bfgs {
  Stopping stopping_progress;
  Reducer red;
  int linesearch = 0;  // LineSearch template argument (bfgs.h:40): 0 MoreThuente, 1 HagerZhang
  int n_ = 0;
  std::vector<double> H_;  // inverse_hessian_, row major
  bool fresh_ = true;      // fresh_inverse_hessian_ (:145-147)
  uint64_t nfev = 0;

  explicit Bfgs(Stopping stop = DefaultStopping(), Reducer r = Reducer{}) : stopping_progress(stop), red(r) {}

  void set_identity() {
    H_.assign(static_cast<size_t>(n_) * n_, 0.0);
    for (int i = 0; i < n_; ++i) H_[static_cast<size_t>(i) * n_ + i] = 1.0;
  }
  void InitializeSolver(int n) {                               // :65-71
    n_ = n;
    set_identity();
    fresh_ = true;
  }
  // (H v)_i = ((H_i0 v_0 + H_i1 v_1) + ...), the order of the reference's matrix * vector
  void matvec(const std::vector<double>& v, std::vector<double>* out) const {
    for (int i = 0; i < n_; ++i) {
      double acc = H_[static_cast<size_t>(i) * n_] * v[0];
      for (int j = 1; j < n_; ++j) acc = acc + H_[static_cast<size_t>(i) * n_ + j] * v[j];
      (*out)[i] = acc;
    }
  }

  State OptimizationStep(const Objective& function, const State& current) {  // :73-137
    const int n = n_;
    constexpr double eps = std::numeric_limits<double>::epsilon();
    const std::vector<double>& g = current.gradient;
    std::vector<double> Hg(n), direction(n);
    matvec(g, &Hg);
    for (int j = 0; j < n; ++j) direction[j] = -Hg[j];         // :81  (-H) * g == -(H g), term by term
    const double phi = red.dot(g.data(), direction.data(), n); // :87
    if ((phi > 0) || std::isnan(phi)) {                        // :88-92
      set_identity();
      for (int j = 0; j < n; ++j) direction[j] = -g[j];
      fresh_ = true;
    }
    double alpha_init = 1.0;                                   // :100-106
    if (fresh_) {
      const double dn = red.norm(direction.data(), n);
      alpha_init = (dn > eps) ? 1.0 / dn : 1.0;
    }
    const State next = (linesearch == 1) ? HagerZhang::Search(current, direction, function, red, alpha_init, &nfev)
                                         : MoreThuente::Search(current, direction, function, red, alpha_init, &nfev);
    std::vector<double> s(n), y(n);                            // :121-122
    for (int j = 0; j < n; ++j) s[j] = next.x[j] - current.x[j];
    for (int j = 0; j < n; ++j) y[j] = next.gradient[j] - g[j];
    const double ys = red.dot(y.data(), s.data(), n);
    if (ys > eps * red.norm(s.data(), n) * red.norm(y.data(), n)) {   // :124
      const double rho = 1.0 / ys;
      std::vector<double> Hy(n);
      matvec(y, &Hy);
      const double yHy = red.dot(y.data(), Hy.data(), n);
      const double c = rho * (rho * yHy + 1.0);
      for (int i = 0; i < n; ++i)                              // :128-130, coefficient by coefficient
        for (int j = 0; j < n; ++j) {
          double& h = H_[static_cast<size_t>(i) * n + j];
          h = (h - rho * (s[i] * Hy[j] + Hy[i] * s[j])) + c * (s[i] * s[j]);
        }
      fresh_ = false;
    }
    return next;
  }

  State Minimize(const Objective& function, const std::vector<double>& x0, Progress* progress_out) {  // solver.h:181-224
    const int n = static_cast<int>(x0.size());
    Progress solver_state;
    State cur;
    cur.x = x0;
    cur.gradient.assign(n, 0.0);
    cur.value = function.eval(cur.x.data(), cur.gradient.data(), n, red);
    nfev = 1;
    InitializeSolver(n);
    do {
      State prev = cur;
      cur = OptimizationStep(function, prev);
      solver_state.Update(prev, cur, stopping_progress);
    } while (solver_state.status == Continue);
    if (progress_out) *progress_out = solver_state;
    return cur;
  }
};

}  // namespace oracle
