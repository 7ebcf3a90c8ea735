// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// C entry points around the UNMODIFIED reference headers, compiled where they
// lie under /root/reference/include (never copied into this repository) over
// oracle/eigen_shim.  The output (oracle/_ref/libref.so) is a binary used to
// validate oracle/lbfgs_oracle.hpp: the solver, line search, stopping logic and
// driver loop executed here ARE the reference's; only the objective functors
// below (the reference ships no N-dimensional Rosenbrock) and the Eigen
// stand-in are ours.  Built by `make -C oracle ref` when the reference tree is
// present; on the GPU box only the prebuilt .so exists.
#include <cstdint>
#include <cstring>

#include "cppoptlib/function.h"
#include "cppoptlib/linesearch/hager_zhang.h"
#include "cppoptlib/linesearch/more_thuente.h"
#include "cppoptlib/solver/bfgs.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"

namespace {

using cppoptlib::function::DifferentiabilityMode;
using cppoptlib::function::FunctionCRTP;

// Chained Rosenbrock-N with the operation order of oracle::Rosenbrock (equals
// the reference's 2-D test functor, src/test/verify.cc:58-69, at N = 2).
class RosenbrockN : public FunctionCRTP<RosenbrockN, double, DifferentiabilityMode::First> {
 public:
  mutable uint64_t nfev = 0;
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    ++nfev;
    const int n = static_cast<int>(x.size());
    double f = 0.0;
    if (gradient) *gradient = VectorType::Zero(n);
    for (int i = 0; i + 1 < n; ++i) {
      const double t1 = 1.0 - x[i];
      const double t2 = x[i + 1] - x[i] * x[i];
      const double term = t1 * t1 + (100.0 * t2) * t2;
      f = (i == 0) ? term : f + term;
    }
    if (gradient) {
      for (int i = 0; i < n; ++i) {
        const bool has_a = (i + 1 < n), has_b = (i > 0);
        double a = 0.0, b = 0.0;
        if (has_a) a = -2.0 * (1.0 - x[i]) + (200.0 * (x[i + 1] - x[i] * x[i])) * (-2.0 * x[i]);
        if (has_b) b = 200.0 * (x[i] - x[i - 1] * x[i - 1]);
        (*gradient)[i] = (has_a && has_b) ? (a + b) : (has_a ? a : b);
      }
    }
    return f;
  }
};

// The same function declared Second mode, with its (tridiagonal) Hessian: Lbfgs then rebuilds its diagonal preconditioner
// from the Hessian at every iterate (solver/lbfgs.h:116-139).  H_jj = [j+1<n] (1200 x_j^2 - 400 x_{j+1} + 2) + [j>0] 200,
// H_{j,j+1} = -400 x_j  (at N = 2: src/examples/trust_region_newton_rosenbrock.cc's Hessian).
class RosenbrockNSecond : public FunctionCRTP<RosenbrockNSecond, double, DifferentiabilityMode::Second> {
 public:
  mutable uint64_t nfev = 0;
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr, MatrixType* hessian = nullptr) const {
    RosenbrockN first;
    const double f = first(x, gradient);
    ++nfev;
    if (hessian) {
      const int n = static_cast<int>(x.size());
      *hessian = MatrixType::Zero(n, n);
      for (int i = 0; i < n; ++i) {
        const bool has_a = (i + 1 < n), has_b = (i > 0);
        const double a = has_a ? ((1200.0 * x[i]) * x[i] - 400.0 * x[i + 1]) + 2.0 : 0.0;
        (*hessian)(i, i) = (has_a && has_b) ? (a + 200.0) : (has_a ? a : (has_b ? 200.0 : 0.0));
        if (has_a) {
          (*hessian)(i, i + 1) = -400.0 * x[i];
          (*hessian)(i + 1, i) = -400.0 * x[i];
        }
      }
    }
    return f;
  }
};

class DiagQuadraticN : public FunctionCRTP<DiagQuadraticN, double, DifferentiabilityMode::First> {
 public:
  const double* a = nullptr;
  double c = 0.0;
  mutable uint64_t nfev = 0;
  ScalarType operator()(const VectorType& x, VectorType* gradient = nullptr) const {
    ++nfev;
    const int n = static_cast<int>(x.size());
    double f = 0.0;
    if (gradient) *gradient = VectorType::Zero(n);
    for (int i = 0; i < n; ++i) {
      const double term = (a[i] * x[i]) * x[i];
      f = (i == 0) ? term : f + term;
      if (gradient) (*gradient)[i] = (2.0 * a[i]) * x[i];
    }
    return f + c;
  }
};

// README.md:126-152 functors, verbatim (Second mode) ...
class SquaredError : public FunctionCRTP<SquaredError, double, DifferentiabilityMode::Second> {
  const Eigen::MatrixXd& A;
  const Eigen::VectorXd& y;

 public:
  SquaredError(const Eigen::MatrixXd& A, const Eigen::VectorXd& y) : A(A), y(y) {}
  int GetDimension() const { return A.cols(); }
  ScalarType operator()(const VectorType& x, VectorType* grad, MatrixType* hess) const {
    Eigen::VectorXd r = A * x - y;
    if (grad) *grad = 2 * A.transpose() * r;
    if (hess) *hess = 2 * A.transpose() * A;
    return r.squaredNorm();
  }
};
class L2Reg : public FunctionCRTP<L2Reg, double, DifferentiabilityMode::Second> {
  int dim;

 public:
  explicit L2Reg(int d) : dim(d) {}
  int GetDimension() const { return dim; }
  ScalarType operator()(const VectorType& x, VectorType* grad, MatrixType* hess) const {
    if (grad) *grad = 2 * x;
    if (hess) {
      hess->setIdentity(dim, dim);
      *hess *= 2;
    }
    return x.squaredNorm();
  }
};
// ... and the same two functors declared First-mode (no Hessian), for the plain L-BFGS path.
class SquaredError1 : public FunctionCRTP<SquaredError1, double, DifferentiabilityMode::First> {
  const Eigen::MatrixXd& A;
  const Eigen::VectorXd& y;

 public:
  SquaredError1(const Eigen::MatrixXd& A, const Eigen::VectorXd& y) : A(A), y(y) {}
  ScalarType operator()(const VectorType& x, VectorType* grad) const {
    Eigen::VectorXd r = A * x - y;
    if (grad) *grad = 2 * A.transpose() * r;
    return r.squaredNorm();
  }
};
class L2Reg1 : public FunctionCRTP<L2Reg1, double, DifferentiabilityMode::First> {
 public:
  ScalarType operator()(const VectorType& x, VectorType* grad) const {
    if (grad) *grad = 2 * x;
    return x.squaredNorm();
  }
};

}  // namespace

extern "C" {

struct ref_stop {
  uint64_t num_iterations;
  double x_delta;
  int32_t x_delta_violations;
  double f_delta;
  int32_t f_delta_violations;
  int32_t f_delta_relative;
  double gradient_norm;
  int32_t gradient_norm_relative;
  int32_t past;
  double past_delta;
};
struct ref_progress {
  int32_t status;
  uint32_t num_iterations;
  uint32_t nfev;
  uint32_t sum_k;  // not observable in the reference (private members): always 0
  double x_delta;
  double f_delta;
  double gradient_norm;
};

}  // extern "C"

namespace {

template <class F, int M, template <class, int> class LineSearch = cppoptlib::solver::linesearch::MoreThuente>
void solve_batch(const F& fn, int n, int64_t B, const ref_stop* st, const double* x0, double* x_out,
                 double* f_out, double* g_out, ref_progress* prog) {
  using Solver = cppoptlib::solver::Lbfgs<F, M, LineSearch>;
  using State = typename Solver::StateType;
  auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
  stop.num_iterations = st->num_iterations;
  stop.x_delta = st->x_delta;
  stop.x_delta_violations = st->x_delta_violations;
  stop.f_delta = st->f_delta;
  stop.f_delta_violations = st->f_delta_violations;
  stop.f_delta_relative = st->f_delta_relative != 0;
  stop.gradient_norm = st->gradient_norm;
  stop.gradient_norm_relative = st->gradient_norm_relative != 0;
  stop.past = st->past;
  stop.past_delta = st->past_delta;
  for (int64_t b = 0; b < B; ++b) {
    typename F::VectorType x(n);
    for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
    Solver solver(stop);
    fn.nfev = 0;
    auto [sol, pr] = solver.Minimize(fn, cppoptlib::function::FunctionState(x));
    for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
    f_out[b] = sol.value;
    if (g_out)
      for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
    if (prog) {
      prog[b].status = static_cast<int32_t>(pr.status);
      prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog[b].nfev = static_cast<uint32_t>(fn.nfev);
      prog[b].sum_k = 0;
      prog[b].x_delta = pr.x_delta;
      prog[b].f_delta = pr.f_delta;
      prog[b].gradient_norm = pr.gradient_norm;
    }
  }
}

template <class F>
int solve_m(const F& fn, int m, int n, int64_t B, const ref_stop* st, const double* x0, double* x_out,
            double* f_out, double* g_out, ref_progress* prog) {
  switch (m) {  // `m` is a template parameter of the reference (solver/lbfgs.h:40)
#define CASE_M(M) case M: solve_batch<F, M>(fn, n, B, st, x0, x_out, f_out, g_out, prog); return 0;
    CASE_M(1) CASE_M(2) CASE_M(3) CASE_M(4) CASE_M(5) CASE_M(6) CASE_M(7) CASE_M(8) CASE_M(10) CASE_M(12)
    CASE_M(16) CASE_M(20)
#undef CASE_M
  }
  return -1;
}

// Lbfgs<F, m, HagerZhang>: the alternative LineSearch template argument (lbfgs.h:41, hager_zhang.h:39-42)
template <class F>
int solve_m_hz(const F& fn, int m, int n, int64_t B, const ref_stop* st, const double* x0, double* x_out,
               double* f_out, double* g_out, ref_progress* prog) {
  using cppoptlib::solver::linesearch::HagerZhang;
  switch (m) {
#define CASE_M(M) case M: solve_batch<F, M, HagerZhang>(fn, n, B, st, x0, x_out, f_out, g_out, prog); return 0;
    CASE_M(1) CASE_M(3) CASE_M(5) CASE_M(6) CASE_M(10)
#undef CASE_M
  }
  return -1;
}

}  // namespace

namespace {
// The functor of the reference's src/examples/svm_primal_lbfgs.cc:35-103 (the example file has its own main() and an
// anonymous namespace, so the class is restated here; its Eigen array expressions are written as the loops a
// loop-based Eigen evaluates them with): soft-margin SVM primal, squared hinge.  x = (w, b).
struct SvmPrimalSquaredHinge
    : public cppoptlib::function::FunctionCRTP<SvmPrimalSquaredHinge, double,
                                               cppoptlib::function::DifferentiabilityMode::First> {
  const double* X = nullptr;
  const double* y = nullptr;
  int N = 0, d = 0;
  double C = 1.0;
  mutable uint64_t nfev = 0;
  ScalarType operator()(const VectorType& x, VectorType* grad = nullptr) const {
    ++nfev;
    std::vector<double> ws(static_cast<size_t>(N));
    double hinge = 0.0;
    for (int i = 0; i < N; ++i) {
      double score = X[static_cast<size_t>(i) * d] * x[0];
      for (int j = 1; j < d; ++j) score = score + X[static_cast<size_t>(i) * d + j] * x[j];
      score = score + x[d];
      const double margin = y[i] * score;
      const double t = 1.0 - margin;
      const double slack = (t < 0.0) ? 0.0 : t;          // (1.0 - margins.array()).max(0.0)
      ws[i] = (-2.0 * slack) * y[i];                      // -2.0 * slacks.array() * labels.array()
      hinge = (i == 0) ? slack * slack : hinge + slack * slack;   // slacks.squaredNorm()
    }
    double ww = x[0] * x[0];                              // w.squaredNorm()
    for (int j = 1; j < d; ++j) ww = ww + x[j] * x[j];
    if (grad) {
      grad->resize(d + 1);
      for (int j = 0; j < d; ++j) {
        double acc = X[j] * ws[0];
        for (int i = 1; i < N; ++i) acc = acc + X[static_cast<size_t>(i) * d + j] * ws[i];
        (*grad)[j] = x[j] + C * acc;                       // w + C * (features.transpose() * weighted_slacks)
      }
      double acc = ws[0];
      for (int i = 1; i < N; ++i) acc = acc + ws[i];
      (*grad)[d] = C * acc;                                // C * weighted_slacks.sum()
    }
    return 0.5 * ww + C * hinge;
  }
};
}  // namespace

namespace {
// The functor of the reference's src/examples/svm_dual_lbfgsb.cc:36-60 (own main() and anonymous namespace, so the class
// is restated; the Eigen expressions written as the loops a loop-based Eigen evaluates them with): dual soft-margin SVM,
// f(alpha) = 0.5 alpha . (Q alpha) - sum(alpha), grad = Q alpha - 1.  Q (the example's kernel_matrix) is handed in.
struct SvmDualObjective
    : public cppoptlib::function::FunctionCRTP<SvmDualObjective, double,
                                               cppoptlib::function::DifferentiabilityMode::First> {
  const double* Q = nullptr;
  int ns = 0;
  mutable uint64_t nfev = 0;
  ScalarType operator()(const VectorType& alpha, VectorType* grad = nullptr) const {
    ++nfev;
    std::vector<double> q(static_cast<size_t>(ns));
    for (int i = 0; i < ns; ++i) {                        // kernel_matrix * alpha
      double acc = Q[static_cast<size_t>(i) * ns] * alpha[0];
      for (int j = 1; j < ns; ++j) acc = acc + Q[static_cast<size_t>(i) * ns + j] * alpha[j];
      q[static_cast<size_t>(i)] = acc;
    }
    double aq = alpha[0] * q[0], sa = alpha[0];           // alpha.dot(q_alpha), alpha.sum()
    for (int i = 1; i < ns; ++i) {
      aq = aq + alpha[i] * q[static_cast<size_t>(i)];
      sa = sa + alpha[i];
    }
    if (grad) {
      grad->resize(ns);
      for (int i = 0; i < ns; ++i) (*grad)[i] = q[static_cast<size_t>(i)] - 1.0;
    }
    return 0.5 * aq - sa;
  }
};
}  // namespace

extern "C" {

// Lbfgs<SvmPrimalSquaredHinge, m>::Minimize of the reference on every row of x0; params = N, d, C, X[N][d], y[N].
int ref_svm_minimize_batch(const double* params, int n, int m, int64_t B, const ref_stop* stop, const double* x0,
                           double* x_out, double* f_out, double* g_out, ref_progress* prog_out) {
  SvmPrimalSquaredHinge fn;
  fn.N = static_cast<int>(params[0]);
  fn.d = static_cast<int>(params[1]);
  fn.C = params[2];
  fn.X = params + 3;
  fn.y = fn.X + static_cast<size_t>(fn.N) * fn.d;
  if (fn.d + 1 != n) return -1;
  return solve_m(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
}

// Same contract as oracle_lbfgs_minimize_batch (oracle_capi.cpp).
int ref_lbfgs_minimize_batch(int objective, const double* params, int n, int m, int64_t B,
                             const ref_stop* stop, const double* x0, double* x_out, double* f_out,
                             double* g_out, ref_progress* prog_out) {
  if (objective == 0) {
    RosenbrockN fn;
    return solve_m(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
  }
  if (objective == 1) {
    DiagQuadraticN fn;
    fn.a = params;
    fn.c = params[n];
    return solve_m(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
  }
  if (objective == 10) {  // chained Rosenbrock as a Second-mode function (non-constant Hessian)
    RosenbrockNSecond fn;
    return solve_m(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
  }
  return -1;
}

// The Second-mode Rosenbrock function (objective 10) under Lbfgs<F, m> with stopping_progress.condition_hessian set
// (progress.h:110, :318-325: tested last in every Update, on the Hessian the reference re-evaluates at the current x,
// :203-210); condition_out[b] = the condition number its Progress reports after the last Update.  m in {5, 6, 10}.
int ref_rosenbrock_second_minimize_batch_cond(int n, int m, int64_t B, const ref_stop* st, double condition_hessian_stop,
                                              const double* x0, double* x_out, double* f_out, double* g_out,
                                              ref_progress* prog, double* condition_out) {
  auto run = [&](auto tag) {
    constexpr int M = decltype(tag)::value;
    using F = RosenbrockNSecond;
    using Solver = cppoptlib::solver::Lbfgs<F, M>;
    using State = typename Solver::StateType;
    auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
    stop.num_iterations = st->num_iterations;
    stop.x_delta = st->x_delta;
    stop.x_delta_violations = st->x_delta_violations;
    stop.f_delta = st->f_delta;
    stop.f_delta_violations = st->f_delta_violations;
    stop.f_delta_relative = st->f_delta_relative != 0;
    stop.gradient_norm = st->gradient_norm;
    stop.gradient_norm_relative = st->gradient_norm_relative != 0;
    stop.past = st->past;
    stop.past_delta = st->past_delta;
    stop.condition_hessian = condition_hessian_stop;
    F fn;
    for (int64_t b = 0; b < B; ++b) {
      typename F::VectorType x(n);
      for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
      Solver solver(stop);
      fn.nfev = 0;
      auto [sol, pr] = solver.Minimize(fn, cppoptlib::function::FunctionState(x));
      if (condition_out) condition_out[b] = pr.condition_hessian;
      for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
      f_out[b] = sol.value;
      if (g_out)
        for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
      if (prog) {
        prog[b].status = static_cast<int32_t>(pr.status);
        prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
        prog[b].nfev = static_cast<uint32_t>(fn.nfev);
        prog[b].sum_k = 0;
        prog[b].x_delta = pr.x_delta;
        prog[b].f_delta = pr.f_delta;
        prog[b].gradient_norm = pr.gradient_norm;
      }
    }
    return 0;
  };
  if (m == 5) return run(std::integral_constant<int, 5>{});
  if (m == 6) return run(std::integral_constant<int, 6>{});
  if (m == 10) return run(std::integral_constant<int, 10>{});
  return -1;
}

// README ridge example (README.md:122-167): `SquaredError(A, y) + lambda * L2Reg(n)` wrapped in a
// FunctionExpr and minimised by Lbfgs<decltype(objective)> (m = 10), one right-hand side per problem.
// second_mode = 1: the functors as printed (Second mode -> diagonal preconditioner path, quirk Q9);
// second_mode = 0: the First-mode twins (plain path).  params = rows, lambda, A (row major).
int ref_ridge_minimize_batch_cond(const double* params, int n, int64_t B, const ref_stop* st, int second_mode,
                                  double condition_hessian_stop, const double* y_all, const double* x0, double* x_out,
                                  double* f_out, double* g_out, ref_progress* prog, double* condition_out);
int ref_ridge_minimize_batch(const double* params, int n, int64_t B, const ref_stop* st, int second_mode,
                             const double* y_all, const double* x0, double* x_out, double* f_out,
                             double* g_out, ref_progress* prog) {
  return ref_ridge_minimize_batch_cond(params, n, B, st, second_mode, 0.0, y_all, x0, x_out, f_out, g_out, prog, nullptr);
}
// the same with stopping_progress.condition_hessian (progress.h:110, :318-325) and the condition number the
// reference's Progress reports (Second mode; First mode leaves it at 0)
int ref_ridge_minimize_batch_cond(const double* params, int n, int64_t B, const ref_stop* st, int second_mode,
                                  double condition_hessian_stop, const double* y_all, const double* x0, double* x_out,
                                  double* f_out, double* g_out, ref_progress* prog, double* condition_out) {
  const int rows = static_cast<int>(params[0]);
  const double lambda = params[1];
  Eigen::MatrixXd A(rows, n);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < n; ++j) A(i, j) = params[2 + static_cast<size_t>(i) * n + j];
  auto run = [&](auto make_objective) {
    for (int64_t b = 0; b < B; ++b) {
      Eigen::VectorXd y(rows);
      for (int i = 0; i < rows; ++i) y[i] = y_all[b * rows + i];
      auto objective = make_objective(y);
      using Obj = decltype(objective);
      using Solver = cppoptlib::solver::Lbfgs<Obj>;
      using State = typename Solver::StateType;
      auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<Obj, State>();
      stop.num_iterations = st->num_iterations;
      stop.x_delta = st->x_delta;
      stop.x_delta_violations = st->x_delta_violations;
      stop.f_delta = st->f_delta;
      stop.f_delta_violations = st->f_delta_violations;
      stop.f_delta_relative = st->f_delta_relative != 0;
      stop.gradient_norm = st->gradient_norm;
      stop.gradient_norm_relative = st->gradient_norm_relative != 0;
      stop.past = st->past;
      stop.past_delta = st->past_delta;
      stop.condition_hessian = condition_hessian_stop;
      Eigen::VectorXd x(n);
      for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
      Solver solver(stop);
      auto [sol, pr] = solver.Minimize(objective, cppoptlib::function::FunctionState(x));
      if (condition_out) condition_out[b] = pr.condition_hessian;
      for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
      f_out[b] = sol.value;
      if (g_out)
        for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
      if (prog) {
        prog[b].status = static_cast<int32_t>(pr.status);
        prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
        prog[b].nfev = 0;
        prog[b].sum_k = 0;
        prog[b].x_delta = pr.x_delta;
        prog[b].f_delta = pr.f_delta;
        prog[b].gradient_norm = pr.gradient_norm;
      }
    }
  };
  if (second_mode) {
    run([&](const Eigen::VectorXd& y) {
      return cppoptlib::function::FunctionExpr(SquaredError(A, y) + lambda * L2Reg(n));
    });
  } else {
    run([&](const Eigen::VectorXd& y) {
      return cppoptlib::function::FunctionExpr(SquaredError1(A, y) + lambda * L2Reg1());
    });
  }
  return 0;
}

// Lbfgsb<F, m> of the reference (solver/lbfgsb.h), bounds shared by the batch (NULL = default box).
int ref_lbfgsb_minimize_batch_ls(int objective, const double* params, int n, int m, int64_t B,
                                 const ref_stop* st, const double* lower, const double* upper, const double* x0,
                                 double* x_out, double* f_out, double* g_out, ref_progress* prog, int linesearch);
int ref_lbfgsb_minimize_batch(int objective, const double* params, int n, int m, int64_t B,
                              const ref_stop* st, const double* lower, const double* upper, const double* x0,
                              double* x_out, double* f_out, double* g_out, ref_progress* prog) {
  return ref_lbfgsb_minimize_batch_ls(objective, params, n, m, B, st, lower, upper, x0, x_out, f_out, g_out, prog, 0);
}
}  // extern "C"
namespace {
// Solver = Lbfgsb<F, m[, LineSearch]> of the reference on every row of x0
template <class Solver, class F>
void lbfgsb_rows(const F& fn, int n, int64_t B, const ref_stop* st, const double* lower, const double* upper,
                 const double* x0, double* x_out, double* f_out, double* g_out, ref_progress* prog) {
  using State = typename Solver::StateType;
  auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
  stop.num_iterations = st->num_iterations;
  stop.x_delta = st->x_delta;
  stop.x_delta_violations = st->x_delta_violations;
  stop.f_delta = st->f_delta;
  stop.f_delta_violations = st->f_delta_violations;
  stop.f_delta_relative = st->f_delta_relative != 0;
  stop.gradient_norm = st->gradient_norm;
  stop.gradient_norm_relative = st->gradient_norm_relative != 0;
  stop.past = st->past;
  stop.past_delta = st->past_delta;
  for (int64_t b = 0; b < B; ++b) {
    typename F::VectorType x(n), lo(n), hi(n);
    for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
    Solver solver(stop);
    if (lower && upper) {
      for (int i = 0; i < n; ++i) {
        lo[i] = lower[i];
        hi[i] = upper[i];
      }
      solver.SetBounds(lo, hi);
    }
    fn.nfev = 0;
    auto [sol, pr] = solver.Minimize(fn, cppoptlib::function::FunctionState(x));
    for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
    f_out[b] = sol.value;
    if (g_out)
      for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
    if (prog) {
      prog[b].status = static_cast<int32_t>(pr.status);
      prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog[b].nfev = static_cast<uint32_t>(fn.nfev);
      prog[b].sum_k = 0;
      prog[b].x_delta = pr.x_delta;
      prog[b].f_delta = pr.f_delta;
      prog[b].gradient_norm = pr.gradient_norm;
    }
  }
}
}  // namespace
extern "C" {
// linesearch: 0 = MoreThuente (the default template argument), 1 = HagerZhang (lbfgsb.h:45, hager_zhang.h:39-42)
// objective 0: Rosenbrock-N; 1: diagonal quadratic; 100: the SVM functor above (params = N, d, C, X, y; m = 5, More-Thuente)
int ref_lbfgsb_minimize_batch_ls(int objective, const double* params, int n, int m, int64_t B,
                                 const ref_stop* st, const double* lower, const double* upper, const double* x0,
                                 double* x_out, double* f_out, double* g_out, ref_progress* prog, int linesearch) {
  if (objective == 100) {
    SvmPrimalSquaredHinge fn;
    fn.N = static_cast<int>(params[0]);
    fn.d = static_cast<int>(params[1]);
    fn.C = params[2];
    fn.X = params + 3;
    fn.y = fn.X + static_cast<size_t>(fn.N) * fn.d;
    if (fn.d + 1 != n || m != 5 || linesearch != 0) return -1;
    lbfgsb_rows<cppoptlib::solver::Lbfgsb<SvmPrimalSquaredHinge, 5>>(fn, n, B, st, lower, upper, x0, x_out, f_out, g_out, prog);
    return 0;
  }
  if (objective == 101) {  // the dual SVM of svm_dual_lbfgsb.cc (params = n, Q): `Lbfgsb<SvmDualObjective>` as in the example
    SvmDualObjective fn;
    fn.ns = static_cast<int>(params[0]);
    fn.Q = params + 1;
    if (fn.ns != n || m != 5 || linesearch != 0) return -1;
    lbfgsb_rows<cppoptlib::solver::Lbfgsb<SvmDualObjective>>(fn, n, B, st, lower, upper, x0, x_out, f_out, g_out, prog);
    return 0;
  }
  if (objective == 1) {  // diagonal quadratic (params = a[n], c): Lbfgsb<F, m>, More-Thuente
    DiagQuadraticN dq;
    dq.a = params;
    dq.c = params[n];
    if (linesearch != 0) return -1;
    auto rundq = [&](auto solver_tag) {
      lbfgsb_rows<decltype(solver_tag)>(dq, n, B, st, lower, upper, x0, x_out, f_out, g_out, prog);
    };
    switch (m) {
      case 3: rundq(cppoptlib::solver::Lbfgsb<DiagQuadraticN, 3>()); return 0;
      case 5: rundq(cppoptlib::solver::Lbfgsb<DiagQuadraticN, 5>()); return 0;
      case 6: rundq(cppoptlib::solver::Lbfgsb<DiagQuadraticN, 6>()); return 0;
      case 8: rundq(cppoptlib::solver::Lbfgsb<DiagQuadraticN, 8>()); return 0;
      case 10: rundq(cppoptlib::solver::Lbfgsb<DiagQuadraticN, 10>()); return 0;
    }
    return -1;
  }
  if (objective != 0) return -1;
  RosenbrockN fn;
  auto run = [&](auto solver_tag) {
    lbfgsb_rows<decltype(solver_tag)>(fn, n, B, st, lower, upper, x0, x_out, f_out, g_out, prog);
  };
  if (linesearch == 1) {
    using cppoptlib::solver::linesearch::HagerZhang;
    switch (m) {
      case 3: run(cppoptlib::solver::Lbfgsb<RosenbrockN, 3, HagerZhang>()); return 0;
      case 5: run(cppoptlib::solver::Lbfgsb<RosenbrockN, 5, HagerZhang>()); return 0;
    }
    return -1;
  }
  switch (m) {
    case 3: run(cppoptlib::solver::Lbfgsb<RosenbrockN, 3>()); return 0;
    case 5: run(cppoptlib::solver::Lbfgsb<RosenbrockN, 5>()); return 0;
    case 6: run(cppoptlib::solver::Lbfgsb<RosenbrockN, 6>()); return 0;
    case 10: run(cppoptlib::solver::Lbfgsb<RosenbrockN, 10>()); return 0;
  }
  return -1;
}

// One `SquaredError(A_b, y_b) + lambda * L2Reg` PER PROBLEM (README.md:126-160: a program that builds the README objective
// once per data set), First-mode functors, minimised by the reference's Lbfgs<F> (m = 10).  data: [B][rows * n + rows] =
// A_b (row major), then y_b.
int ref_ridge_own_matrix_minimize_batch(int rows, double lambda, int n, int64_t B, const ref_stop* st, const double* data,
                                        const double* x0, double* x_out, double* f_out, double* g_out, ref_progress* prog) {
  const size_t stride = static_cast<size_t>(rows) * n + rows;
  for (int64_t b = 0; b < B; ++b) {
    const double* a = data + static_cast<size_t>(b) * stride;
    Eigen::MatrixXd A(rows, n);
    Eigen::VectorXd y(rows);
    for (int i = 0; i < rows; ++i) {
      for (int j = 0; j < n; ++j) A(i, j) = a[static_cast<size_t>(i) * n + j];
      y[i] = a[static_cast<size_t>(rows) * n + i];
    }
    auto objective = cppoptlib::function::FunctionExpr(SquaredError1(A, y) + lambda * L2Reg1());
    using Obj = decltype(objective);
    using Solver = cppoptlib::solver::Lbfgs<Obj>;
    using State = typename Solver::StateType;
    auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<Obj, State>();
    stop.num_iterations = st->num_iterations;
    stop.x_delta = st->x_delta;
    stop.x_delta_violations = st->x_delta_violations;
    stop.f_delta = st->f_delta;
    stop.f_delta_violations = st->f_delta_violations;
    stop.f_delta_relative = st->f_delta_relative != 0;
    stop.gradient_norm = st->gradient_norm;
    stop.gradient_norm_relative = st->gradient_norm_relative != 0;
    stop.past = st->past;
    stop.past_delta = st->past_delta;
    Eigen::VectorXd x(n);
    for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
    Solver solver(stop);
    auto [sol, pr] = solver.Minimize(objective, cppoptlib::function::FunctionState(x));
    for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
    f_out[b] = sol.value;
    if (g_out)
      for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
    if (prog) {
      prog[b].status = static_cast<int32_t>(pr.status);
      prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog[b].nfev = 0;
      prog[b].sum_k = 0;
      prog[b].x_delta = pr.x_delta;
      prog[b].f_delta = pr.f_delta;
      prog[b].gradient_norm = pr.gradient_norm;
    }
  }
  return 0;
}

// Lbfgsb<F, m> of the reference on the README regression objective (README.md:126-160: `SquaredError(A, y_b) +
// lambda * L2Reg`, First-mode functors, wrapped in a FunctionExpr exactly as src/examples/linear_regression.cc:58-74
// hands its regression objective to Lbfgsb), one right-hand side per problem, bounds shared by the batch (NULL = the
// default box).  params = rows, lambda, A (row major).
int ref_lbfgsb_ridge_minimize_batch(const double* params, int n, int m, int64_t B, const ref_stop* st,
                                    const double* lower, const double* upper, const double* y_all, const double* x0,
                                    double* x_out, double* f_out, double* g_out, ref_progress* prog) {
  const int rows = static_cast<int>(params[0]);
  const double lambda = params[1];
  Eigen::MatrixXd A(rows, n);
  for (int i = 0; i < rows; ++i)
    for (int j = 0; j < n; ++j) A(i, j) = params[2 + static_cast<size_t>(i) * n + j];
  auto run = [&](auto m_tag) {
    constexpr int M = decltype(m_tag)::value;
    for (int64_t b = 0; b < B; ++b) {
      Eigen::VectorXd y(rows);
      for (int i = 0; i < rows; ++i) y[i] = y_all[b * rows + i];
      auto objective = cppoptlib::function::FunctionExpr(SquaredError1(A, y) + lambda * L2Reg1());
      using Obj = decltype(objective);
      using Solver = cppoptlib::solver::Lbfgsb<Obj, M>;
      using State = typename Solver::StateType;
      auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<Obj, State>();
      stop.num_iterations = st->num_iterations;
      stop.x_delta = st->x_delta;
      stop.x_delta_violations = st->x_delta_violations;
      stop.f_delta = st->f_delta;
      stop.f_delta_violations = st->f_delta_violations;
      stop.f_delta_relative = st->f_delta_relative != 0;
      stop.gradient_norm = st->gradient_norm;
      stop.gradient_norm_relative = st->gradient_norm_relative != 0;
      stop.past = st->past;
      stop.past_delta = st->past_delta;
      Eigen::VectorXd x(n), lo(n), hi(n);
      for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
      Solver solver(stop);
      if (lower && upper) {
        for (int i = 0; i < n; ++i) {
          lo[i] = lower[i];
          hi[i] = upper[i];
        }
        solver.SetBounds(lo, hi);
      }
      auto [sol, pr] = solver.Minimize(objective, cppoptlib::function::FunctionState(x));
      for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
      f_out[b] = sol.value;
      if (g_out)
        for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
      if (prog) {
        prog[b].status = static_cast<int32_t>(pr.status);
        prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
        prog[b].nfev = 0;
        prog[b].sum_k = 0;
        prog[b].x_delta = pr.x_delta;
        prog[b].f_delta = pr.f_delta;
        prog[b].gradient_norm = pr.gradient_norm;
      }
    }
  };
  switch (m) {
    case 5: run(std::integral_constant<int, 5>()); return 0;
    case 8: run(std::integral_constant<int, 8>()); return 0;
    case 10: run(std::integral_constant<int, 10>()); return 0;
  }
  return -1;
}

// The reference's MoreThuente::cstep (linesearch/more_thuente.h:261-407).
// Lbfgs<F, m, HagerZhang>::Minimize, same contract as ref_lbfgs_minimize_batch
int ref_lbfgs_hz_minimize_batch(int objective, const double* params, int n, int m, int64_t B,
                                const ref_stop* stop, const double* x0, double* x_out, double* f_out,
                                double* g_out, ref_progress* prog_out) {
  if (objective == 0) {
    RosenbrockN fn;
    return solve_m_hz(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
  }
  if (objective == 1) {
    DiagQuadraticN fn;
    fn.a = params;
    fn.c = params[n];
    return solve_m_hz(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
  }
  if (objective == 10) {
    RosenbrockNSecond fn;
    return solve_m_hz(fn, m, n, B, stop, x0, x_out, f_out, g_out, prog_out);
  }
  return -1;
}

// Bfgs<F, LineSearch>::Minimize of the reference (solver/bfgs.h), same contract as ref_lbfgs_minimize_batch.
int ref_bfgs_minimize_batch(int objective, const double* params, int n, int64_t B, const ref_stop* st,
                            const double* x0, double* x_out, double* f_out, double* g_out, ref_progress* prog,
                            int linesearch) {
  auto run = [&](auto& fn, auto solver_tag) {
    using F = std::decay_t<decltype(fn)>;
    using Solver = decltype(solver_tag);
    using State = typename Solver::StateType;
    auto stop = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
    stop.num_iterations = st->num_iterations;
    stop.x_delta = st->x_delta;
    stop.x_delta_violations = st->x_delta_violations;
    stop.f_delta = st->f_delta;
    stop.f_delta_violations = st->f_delta_violations;
    stop.f_delta_relative = st->f_delta_relative != 0;
    stop.gradient_norm = st->gradient_norm;
    stop.gradient_norm_relative = st->gradient_norm_relative != 0;
    stop.past = st->past;
    stop.past_delta = st->past_delta;
    for (int64_t b = 0; b < B; ++b) {
      typename F::VectorType x(n);
      for (int i = 0; i < n; ++i) x[i] = x0[b * n + i];
      Solver solver(stop);
      fn.nfev = 0;
      auto [sol, pr] = solver.Minimize(fn, cppoptlib::function::FunctionState(x));
      for (int i = 0; i < n; ++i) x_out[b * n + i] = sol.x[i];
      f_out[b] = sol.value;
      if (g_out)
        for (int i = 0; i < n; ++i) g_out[b * n + i] = sol.gradient[i];
      if (prog) {
        prog[b].status = static_cast<int32_t>(pr.status);
        prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
        prog[b].nfev = static_cast<uint32_t>(fn.nfev);
        prog[b].sum_k = 0;
        prog[b].x_delta = pr.x_delta;
        prog[b].f_delta = pr.f_delta;
        prog[b].gradient_norm = pr.gradient_norm;
      }
    }
  };
  using cppoptlib::solver::linesearch::HagerZhang;
  if (objective == 0) {
    RosenbrockN fn;
    if (linesearch == 1) run(fn, cppoptlib::solver::Bfgs<RosenbrockN, HagerZhang>());
    else run(fn, cppoptlib::solver::Bfgs<RosenbrockN>());
    return 0;
  }
  if (objective == 1) {
    DiagQuadraticN fn;
    fn.a = params;
    fn.c = params[n];
    if (linesearch == 1) run(fn, cppoptlib::solver::Bfgs<DiagQuadraticN, HagerZhang>());
    else run(fn, cppoptlib::solver::Bfgs<DiagQuadraticN>());
    return 0;
  }
  return -1;
}

// One HagerZhang::Search (State overload, hager_zhang.h:100-116) per row: from x[b] along s[b] with
// the initial step alpha_init[b].  Outputs the accepted point, value, gradient and step width.
int ref_hz_search(int objective, const double* params, int n, int64_t B, const double* x, const double* s,
                  const double* alpha_init, double* x_out, double* f_out, double* g_out, double* alpha_out) {
  auto run = [&](auto& fn) {
    using F = std::decay_t<decltype(fn)>;
    using LS = cppoptlib::solver::linesearch::HagerZhang<F, 1>;
    for (int64_t b = 0; b < B; ++b) {
      typename F::VectorType xv(n), sv(n);
      for (int i = 0; i < n; ++i) {
        xv[i] = x[b * n + i];
        sv[i] = s[b * n + i];
      }
      const cppoptlib::function::FunctionState start(fn, xv);
      double alpha = 0;
      const auto next = LS::Search(start, sv, fn, alpha_init[b], &alpha);
      for (int i = 0; i < n; ++i) {
        x_out[b * n + i] = next.x[i];
        g_out[b * n + i] = next.gradient[i];
      }
      f_out[b] = next.value;
      alpha_out[b] = alpha;
    }
  };
  if (objective == 0) {
    RosenbrockN fn;
    run(fn);
    return 0;
  }
  if (objective == 1) {
    DiagQuadraticN fn;
    fn.a = params;
    fn.c = params[n];
    run(fn);
    return 0;
  }
  return -1;
}

int ref_cstep(double* v, double fp, double dp, int* brackt, double stpmin, double stpmax, int* info) {
  using LS = cppoptlib::solver::linesearch::MoreThuente<RosenbrockN, 1>;
  bool b = *brackt != 0;
  const int rc = LS::cstep(v[0], v[1], v[2], v[3], v[4], v[5], v[6], fp, dp, b, stpmin, stpmax, *info);
  *brackt = b ? 1 : 0;
  return rc;
}

// Reference default presets (solver/progress.h:353-431, :456-464).
void ref_default_stop(ref_stop* s, int preset) {
  using State = cppoptlib::function::FunctionState<double, Eigen::Dynamic>;
  auto d = preset == 1 ? cppoptlib::solver::ConservativeStoppingSolverProgress<RosenbrockN, State>()
                       : cppoptlib::solver::DefaultStoppingSolverProgress<RosenbrockN, State>();
  s->num_iterations = d.num_iterations;
  s->x_delta = d.x_delta;
  s->x_delta_violations = d.x_delta_violations;
  s->f_delta = d.f_delta;
  s->f_delta_violations = d.f_delta_violations;
  s->f_delta_relative = d.f_delta_relative;
  s->gradient_norm = d.gradient_norm;
  s->gradient_norm_relative = d.gradient_norm_relative;
  s->past = d.past;
  s->past_delta = d.past_delta;
}

}  // extern "C"
