// ORACLE — TEST INFRASTRUCTURE ONLY (see lbfgs_oracle.hpp).  C entry points so
// tests/ and bench.py's cpu_baseline leg can drive the CPU restatement through
// ctypes.  Never linked into the product library.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "lbfgs_oracle.hpp"
#include "auglag_oracle.hpp"
#include "lbfgsb_oracle.hpp"
#include "lbfgsb_fast_oracle.hpp"

extern "C" {

struct oracle_stop {
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

struct oracle_progress {
  int32_t status;
  uint32_t num_iterations;
  uint32_t nfev;
  uint32_t sum_k;
  double x_delta;
  double f_delta;
  double gradient_norm;
};

static oracle::Stopping to_stop(const oracle_stop* s) {
  oracle::Stopping o;
  o.num_iterations = s->num_iterations;
  o.x_delta = s->x_delta;
  o.x_delta_violations = s->x_delta_violations;
  o.f_delta = s->f_delta;
  o.f_delta_violations = s->f_delta_violations;
  o.f_delta_relative = s->f_delta_relative != 0;
  o.gradient_norm = s->gradient_norm;
  o.gradient_norm_relative = s->gradient_norm_relative != 0;
  o.past = s->past;
  o.past_delta = s->past_delta;
  return o;
}

static std::unique_ptr<oracle::Objective> make_objective(int id, const double* params, int n,
                                                         const double* per_problem = nullptr) {
  if (id == 0) return std::make_unique<oracle::Rosenbrock>();
  if (id == 2 || id == 3 || id == 5) {  // params = rows, lambda, A[rows][n]; per_problem = y[B][rows]; 3 = matrix-core
    auto q = std::make_unique<oracle::SquaredErrorRidge>();          // twin, 5 = normal-equation (Gram) twin
    q->fma_chains = (id == 3);
    q->gram = (id == 5);
    q->rows = static_cast<int>(params[0]);
    q->lambda = params[1];
    q->A = params + 2;
    q->y_all = per_problem;
    q->y = per_problem;
    return q;
  }
  if (id == 6 || id == 7) {  // own-matrix ridge: params = rows, lambda; per_problem = [B][rows * n + rows] (A_b, then y_b)
    auto q = std::make_unique<oracle::SquaredErrorRidge>();          // 6 = normal-equation (Gram) twin, 7 = reference order
    q->gram = (id == 6);
    q->rows = static_cast<int>(params[0]);
    q->lambda = params[1];
    q->own_n = n;
    q->own_stride = static_cast<int64_t>(q->rows) * n + q->rows;
    q->y_all = per_problem;
    q->A = per_problem;
    q->y = per_problem ? per_problem + static_cast<int64_t>(q->rows) * n : nullptr;
    return q;
  }
  if (id == 100) {  // params = N, d, C, X[N][d], y[N]  (the user-objective example, MI355_OBJ_USER_FIRST)
    auto q = std::make_unique<oracle::SvmSquaredHinge>();
    q->N = static_cast<int>(params[0]);
    q->d = static_cast<int>(params[1]);
    q->C = params[2];
    q->X = params + 3;
    q->y = q->X + static_cast<size_t>(q->N) * q->d;
    if (q->d + 1 != n || q->N < 1) return nullptr;
    return q;
  }
  if (id == 101) {  // params = n, Q[n][n]  (the second user-objective example: the dual SVM of svm_dual_lbfgsb.cc)
    auto q = std::make_unique<oracle::SvmDual>();
    q->ns = static_cast<int>(params[0]);
    q->Q = params + 1;
    if (q->ns != n) return nullptr;
    return q;
  }
  if (id == 1) {
    auto q = std::make_unique<oracle::DiagQuadratic>();
    q->a.assign(params, params + n);
    q->c = params[n];
    return q;
  }
  return nullptr;
}

void oracle_default_stop(oracle_stop* s, int preset) {
  oracle::Stopping d = preset == 1 ? oracle::ConservativeStopping() : oracle::DefaultStopping();
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

// stopping_progress.condition_hessian for the NEXT Second-mode solves (0 = off, the default); test helper, not
// thread safe.  The condition number itself is computed by the oracle (SquaredErrorRidge::hessian_condition).
static double g_condition_hessian_stop = 0.0;
void oracle_set_condition_hessian_stop(double v) { g_condition_hessian_stop = v; }
static double g_last_hessian_condition = 0.0;
double oracle_last_hessian_condition() { return g_last_hessian_condition; }
// Second-mode functions whose Hessian is NOT constant (second_mode = 2): Progress::condition_hessian after the last Update of
// every problem of the most recent batch (filled when the stopping test is on or oracle_track_hessian_condition(1) was called)
static int g_track_condition = 0;
static std::vector<double> g_hessian_conditions;
void oracle_track_hessian_condition(int on) { g_track_condition = on; }
int64_t oracle_hessian_conditions(double* out, int64_t count) {
  const int64_t k = std::min<int64_t>(count, static_cast<int64_t>(g_hessian_conditions.size()));
  for (int64_t i = 0; i < k; ++i) out[i] = g_hessian_conditions[static_cast<size_t>(i)];
  return k;
}

// objective: 0 = Rosenbrock-N, 1 = DiagQuadratic (params = a[0..n), c), 2 = SquaredErrorRidge
// (params = rows, lambda, A[rows][n]; per_problem = y[B][rows]).
// reduction: 0 sequential, 1 butterfly over `width` lanes.
// Returns 0, or -1 for a bad argument.
int oracle_lbfgs_minimize_batch(int objective, const double* params, int n, int m, int64_t B,
                                const oracle_stop* stop, int reduction, int width,
                                const double* x0, double* x_out, double* f_out, double* g_out,
                                oracle_progress* prog_out, int nthreads, const double* per_problem,
                                int second_mode, int linesearch) {
  // (n > 1024: the sequential and strided policies on the Rosenbrock / DiagQuadratic objectives -- the large-n twin)
  if (n <= 0 || m <= 0 || B < 0) return -1;
  if (n > 1024 && ((reduction & 0xff) == 1 || (objective != 0 && objective != 1) || second_mode == 1)) return -1;
  // second_mode: 1 = constant Hessian (the ridge objective), 2 = diag H(x) from the objective at every iterate (Rosenbrock)
  if (second_mode == 1 && objective != 2 && objective != 3 && objective != 5) return -1;
  if (second_mode == 2 && objective != 0) return -1;
  if (second_mode < 0 || second_mode > 2) return -1;
  // reduction: 0 sequential, 1 butterfly; butterfly_fma = 1 | (E << 8) with E = coordinates per lane of the twin kernel
  const int fma_group = reduction >> 8;
  reduction &= 0xff;
  if (fma_group && (reduction != 1 || (fma_group & (fma_group - 1)) || fma_group > width)) return -1;
  if (reduction == 1 && (width < n || width > 1024 || (width & (width - 1)))) return -1;
  if (reduction == 2 && (width < 1 || width > 1024 || (width & (width - 1)) || fma_group)) return -1;   // strided
  auto probe = make_objective(objective, params, n, per_problem);
  if (!probe) return -1;
  if ((objective == 2 || objective == 3 || objective == 5 || objective == 6 || objective == 7) && !per_problem) return -1;
  const oracle::Stopping st = to_stop(stop);
  oracle::Reducer red;
  red.kind = reduction == 2 ? oracle::Reduction::Strided
                            : (reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential);
  red.width = width;
  red.fma_group = fma_group;
  const bool track = second_mode == 2 && (g_condition_hessian_stop > 0 || g_track_condition);
  g_hessian_conditions.assign(track ? static_cast<size_t>(B) : 0, 0.0);
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    auto fn = make_objective(objective, params, n, per_problem);
    oracle::Lbfgs solver(m, st, red);
    solver.linesearch = linesearch;
    if (second_mode == 2) {
      solver.hessian_from_objective = true;
      solver.track_condition = track;
      solver.stopping_progress.condition_hessian = g_condition_hessian_stop;
    }
    if (second_mode == 1) {
      auto* ridge = static_cast<oracle::SquaredErrorRidge*>(fn.get());
      solver.hessian_diagonal = ridge->hessian_diagonal(n);
      solver.hessian_condition = ridge->hessian_condition(n);
      solver.stopping_progress.condition_hessian = g_condition_hessian_stop;
      g_last_hessian_condition = solver.hessian_condition;
    }
    std::vector<double> x(n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t b = 0; b < B; ++b) {
      std::memcpy(x.data(), x0 + b * n, sizeof(double) * n);
      fn->set_problem(b);
      oracle::Progress pr;
      oracle::State sol = solver.Minimize(*fn, x, &pr);
      if (track) g_hessian_conditions[static_cast<size_t>(b)] = solver.last_condition;
      std::memcpy(x_out + b * n, sol.x.data(), sizeof(double) * n);
      f_out[b] = sol.value;
      if (g_out) std::memcpy(g_out + b * n, sol.gradient.data(), sizeof(double) * n);
      if (prog_out) {
        oracle_progress& p = prog_out[b];
        p.status = pr.status;
        p.num_iterations = static_cast<uint32_t>(pr.num_iterations);
        p.nfev = static_cast<uint32_t>(solver.nfev);
        p.sum_k = static_cast<uint32_t>(solver.sum_k);
        p.x_delta = pr.x_delta;
        p.f_delta = pr.f_delta;
        p.gradient_norm = pr.gradient_norm;
      }
    }
  }
  return 0;
}

// Operation counts of the most recent oracle_lbfgsb_minimize_batch call, summed over its problems (bench.py's useful-
// flop model of configs[4]): out[0] = flops of the reference's algebra (Lbfgsb::ReferenceStepFlops, evaluations of the
// line search excluded), out[1] = breakpoints examined, out[2] = free variables, out[3] = OptimizationSteps.
static double g_lbfgsb_counts[4] = {0, 0, 0, 0};
void oracle_lbfgsb_last_model_counts(double* out) {
  for (int i = 0; i < 4; ++i) out[i] = g_lbfgsb_counts[i];
}

// Box-constrained solver (lbfgsb_oracle.hpp).  lower/upper: n doubles each shared by the batch,
// or NULL for the reference's default unbounded box.  Otherwise like oracle_lbfgs_minimize_batch.
int oracle_lbfgsb_minimize_batch(int objective, const double* params, int n, int m, int64_t B,
                                 const oracle_stop* stop, int reduction, int width, const double* lower,
                                 const double* upper, const double* x0, double* x_out, double* f_out,
                                 double* g_out, oracle_progress* prog_out, int nthreads,
                                 const double* per_problem, int std_sort_order, int linesearch) {
  if (n <= 0 || n > 1024 || m <= 0 || B < 0) return -1;
  if (reduction == 1 && (width < n || width > 1024 || (width & (width - 1)))) return -1;
  auto probe = make_objective(objective, params, n, per_problem);
  if (!probe) return -1;
  const oracle::Stopping st = to_stop(stop);
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  double cnt_flops = 0, cnt_bp = 0, cnt_free = 0, cnt_steps = 0;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads) reduction(+ : cnt_flops, cnt_bp, cnt_free, cnt_steps)
#endif
  {
    auto fn = make_objective(objective, params, n, per_problem);
    std::vector<double> x(n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t b = 0; b < B; ++b) {
      oracle::Lbfgsb solver(m, st, red);
      solver.std_sort_order = std_sort_order != 0;
      solver.linesearch = linesearch;
      if (lower && upper) {
        solver.lower.assign(lower, lower + n);
        solver.upper.assign(upper, upper + n);
      }
      std::memcpy(x.data(), x0 + b * n, sizeof(double) * n);
      fn->set_problem(b);
      oracle::Progress pr;
      oracle::State sol = solver.Minimize(*fn, x, &pr);
      cnt_flops += solver.model_flops;
      cnt_bp += static_cast<double>(solver.sum_breakpoints);
      cnt_free += static_cast<double>(solver.sum_free);
      cnt_steps += static_cast<double>(pr.num_iterations);
      std::memcpy(x_out + b * n, sol.x.data(), sizeof(double) * n);
      f_out[b] = sol.value;
      if (g_out) std::memcpy(g_out + b * n, sol.gradient.data(), sizeof(double) * n);
      if (prog_out) {
        oracle_progress& p = prog_out[b];
        p.status = pr.status;
        p.num_iterations = static_cast<uint32_t>(pr.num_iterations);
        p.nfev = static_cast<uint32_t>(solver.nfev);
        p.sum_k = static_cast<uint32_t>(solver.sum_k);
        p.x_delta = pr.x_delta;
        p.f_delta = pr.f_delta;
        p.gradient_norm = pr.gradient_norm;
      }
    }
  }
  g_lbfgsb_counts[0] = cnt_flops;
  g_lbfgsb_counts[1] = cnt_bp;
  g_lbfgsb_counts[2] = cnt_free;
  g_lbfgsb_counts[3] = cnt_steps;
  return 0;
}

// Twin of the engine's relaxed-algebra L-BFGS-B kernel (lbfgsb_fast_oracle.hpp).  m: history size, Mcap: the capacity
// the kernel is built for (5, 8 or — thirty-two lanes per problem — 10), E: coordinates per lane (lanes x E >= n).  Otherwise like oracle_lbfgsb_minimize_batch.
int oracle_lbfgsb_fast_minimize_batch(int objective, const double* params, int n, int m, int Mcap, int E, int64_t B,
                                      const oracle_stop* stop, const double* lower, const double* upper,
                                      const double* x0, double* x_out, double* f_out, double* g_out,
                                      oracle_progress* prog_out, int nthreads, const double* per_problem) {
  const int lanes = (2 * Mcap > 16) ? 32 : 16;
  if (n <= 0 || m <= 0 || m > Mcap || 2 * Mcap > 32 || E <= 0 || lanes * E < n || lanes * E > 1024 || B < 0) return -1;
  auto probe = make_objective(objective, params, n, per_problem);
  if (!probe) return -1;
  const oracle::Stopping st = to_stop(stop);
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    auto fn = make_objective(objective, params, n, per_problem);
    std::vector<double> x(n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t b = 0; b < B; ++b) {
      oracle::LbfgsbFast solver(m, Mcap, E, st);
      if (lower && upper) {
        solver.lower.assign(lower, lower + n);
        solver.upper.assign(upper, upper + n);
      }
      std::memcpy(x.data(), x0 + b * n, sizeof(double) * n);
      fn->set_problem(b);
      oracle::Progress pr;
      oracle::State sol = solver.Minimize(*fn, x, &pr);
      std::memcpy(x_out + b * n, sol.x.data(), sizeof(double) * n);
      f_out[b] = sol.value;
      if (g_out) std::memcpy(g_out + b * n, sol.gradient.data(), sizeof(double) * n);
      if (prog_out) {
        oracle_progress& p = prog_out[b];
        p.status = pr.status;
        p.num_iterations = static_cast<uint32_t>(pr.num_iterations);
        p.nfev = static_cast<uint32_t>(solver.nfev);
        p.sum_k = static_cast<uint32_t>(solver.sum_k);
        p.x_delta = pr.x_delta;
        p.f_delta = pr.f_delta;
        p.gradient_norm = pr.gradient_norm;
      }
    }
  }
  return 0;
}

// v = {stx, fx, dx, sty, fy, dy, stp} in/out; returns cstep's return value.
int oracle_cstep(double* v, double fp, double dp, int* brackt, double stpmin, double stpmax,
                 int* info) {
  bool b = *brackt != 0;
  const int rc = oracle::MoreThuente::cstep(v[0], v[1], v[2], v[3], v[4], v[5], v[6], fp, dp, b,
                                           stpmin, stpmax, *info);
  *brackt = b ? 1 : 0;
  return rc;
}

// One objective evaluation (for objective-level parity tests).
double oracle_eval(int objective, const double* params, int n, int reduction, int width,
                   const double* x, double* g, const double* per_problem) {
  auto fn = make_objective(objective, params, n, per_problem);
  oracle::Reducer red;
  red.fma_group = reduction >> 8;  // butterfly_fma = 1 | (E << 8)
  reduction &= 0xff;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  return fn->eval(x, g, n, red);
}

// Dense BFGS (solver/bfgs.h), same contract as oracle_lbfgs_minimize_batch without m.
int oracle_bfgs_minimize_batch(int objective, const double* params, int n, int64_t B, const oracle_stop* stop,
                               int reduction, int width, const double* x0, double* x_out, double* f_out,
                               double* g_out, oracle_progress* prog_out, int nthreads, const double* per_problem,
                               int linesearch) {
  if (n <= 0 || n > 1024 || B < 0) return -1;
  if (reduction == 1 && (width < n || width > 1024 || (width & (width - 1)))) return -1;
  auto probe = make_objective(objective, params, n, per_problem);
  if (!probe) return -1;
  const oracle::Stopping st = to_stop(stop);
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel num_threads(nthreads)
#endif
  {
    auto fn = make_objective(objective, params, n, per_problem);
    oracle::Bfgs solver(st, red);
    solver.linesearch = linesearch;
    std::vector<double> x(n);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t b = 0; b < B; ++b) {
      fn->set_problem(b);
      std::copy(x0 + b * n, x0 + (b + 1) * n, x.begin());
      oracle::Progress pr;
      const oracle::State sol = solver.Minimize(*fn, x, &pr);
      std::copy(sol.x.begin(), sol.x.end(), x_out + b * n);
      f_out[b] = sol.value;
      if (g_out) std::copy(sol.gradient.begin(), sol.gradient.end(), g_out + b * n);
      if (prog_out) {
        prog_out[b].status = static_cast<int32_t>(pr.status);
        prog_out[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
        prog_out[b].nfev = static_cast<uint32_t>(solver.nfev);
        prog_out[b].sum_k = 0;
        prog_out[b].x_delta = pr.x_delta;
        prog_out[b].f_delta = pr.f_delta;
        prog_out[b].gradient_norm = pr.gradient_norm;
      }
    }
  }
  return 0;
}

// One HagerZhang::Search per row (twin of ref_hz_search in ref_capi.cpp); nfev_out may be null.
int oracle_hz_search(int objective, const double* params, int n, int64_t B, int reduction, int width,
                     const double* x, const double* s, const double* alpha_init, double* x_out, double* f_out,
                     double* g_out, double* alpha_out, uint64_t* nfev_out) {
  if (n <= 0 || n > 1024 || B < 0) return -1;
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  auto fn = make_objective(objective, params, n, nullptr);
  if (!fn) return -1;
  for (int64_t b = 0; b < B; ++b) {
    oracle::State st;
    st.x.assign(x + b * n, x + (b + 1) * n);
    st.gradient.assign(n, 0.0);
    st.value = fn->eval(st.x.data(), st.gradient.data(), n, red);
    const std::vector<double> dir(s + b * n, s + (b + 1) * n);
    double alpha = alpha_init[b];
    uint64_t nfev = 0;
    oracle::HagerZhang::hzls(*fn, red, &st.x, &st.value, &st.gradient, &alpha, dir, &nfev);
    for (int i = 0; i < n; ++i) {
      x_out[b * n + i] = st.x[i];
      g_out[b * n + i] = st.gradient[i];
    }
    f_out[b] = st.value;
    alpha_out[b] = alpha;
    if (nfev_out) nfev_out[b] = nfev;
  }
  return 0;
}

// Hessian diagonal of the ridge objective (constant), n doubles.
int oracle_ridge_hessian_diagonal(const double* params, int n, double* out) {
  auto fn = make_objective(2, params, n, nullptr);
  const std::vector<double> d = static_cast<oracle::SquaredErrorRidge*>(fn.get())->hessian_diagonal(n);
  for (int j = 0; j < n; ++j) out[j] = d[j];
  return 0;
}

int oracle_num_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

// ---------------------------------------------------------------------------------------------
// Augmented Lagrangian (auglag_oracle.hpp); twin of ref_auglag_minimize_batch in ref_auglag_capi.cpp.
struct oracle_al_config {
  double penalty_growth_factor, violation_shrink_ratio;
  int32_t auto_scale_initial_penalty;
  double penalty_auto_objective_scale, penalty_auto_min, penalty_auto_max;
  int32_t warmup_max_inner_iterations;
  double warmup_inner_gradient_tolerance, multiplier_max;
  uint64_t outer_num_iterations;
  double constraint_threshold, kkt_stationarity_threshold;
  int32_t loop;  // (device execution mode of the product's struct; not used here)
};
struct oracle_al_progress {
  int32_t status;
  uint32_t num_iterations;
  double x_delta, f_delta, gradient_norm;
  uint64_t inner_iterations, nfev, sum_k;
};

// The user blob of the problems built next (mi355_al_problem::user_params): kept until replaced.
static std::vector<double> g_al_user_params;

// Problem from the C-ABI arrays: kinds / coef per table row, forms / ks per term, parts[t] primitives per term
// (null = one each).
static oracle::ConstrainedProblem build_problem(int n, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms,
                                                const double* ks, const double* coef, const int32_t* parts) {
  oracle::ConstrainedProblem prob;
  int row = 0;
  auto make = [&](int t) {
    oracle::Term term;
    term.form = forms[t];
    term.k = ks[t];
    const int signed_count = parts ? parts[t] : 1;
    term.product = signed_count < 0;            // parts[t] = -2: the product F1 * F2 (ProdExpression)
    const int count = signed_count < 0 ? -signed_count : signed_count;
    for (int r = 0; r < count; ++r, ++row) {
      oracle::Primitive p;
      p.kind = kinds[row];
      p.coef.assign(coef + static_cast<size_t>(row) * (n + 1), coef + static_cast<size_t>(row + 1) * (n + 1));
      p.user = g_al_user_params.empty() ? nullptr : g_al_user_params.data();
      term.parts.push_back(p);
    }
    return term;
  };
  prob.objective = make(0);
  for (int t = 1; t <= n_eq; ++t) prob.equality.push_back(make(t));
  for (int t = 1 + n_eq; t <= n_eq + n_ineq; ++t) prob.inequality.push_back(make(t));
  return prob;
}

// ks_batch (null, or [B][1 + n_eq + n_ineq]): row b replaces the constants k of the problem's terms.
static void set_constants(oracle::ConstrainedProblem* prob, const double* ks_batch, int64_t b) {
  if (!ks_batch) return;
  const size_t T = 1 + prob->equality.size() + prob->inequality.size();
  const double* row = ks_batch + static_cast<size_t>(b) * T;
  prob->objective.k = row[0];
  for (size_t i = 0; i < prob->equality.size(); ++i) prob->equality[i].k = row[1 + i];
  for (size_t i = 0; i < prob->inequality.size(); ++i) prob->inequality[i].k = row[1 + prob->equality.size() + i];
}

int oracle_auglag_set_user_params(const double* params, int64_t count) {
  if (count < 0 || (count > 0 && !params)) return -1;
  g_al_user_params.assign(params, params + count);
  return 0;
}

// One composite evaluation per row (for the assembly tests): value and gradient of
// ToAugmentedLagrangian(prob, (lambda, mu), penalty) at x[b].
int oracle_auglag_eval(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms,
                       const double* ks, const double* coef, int reduction, int width, const double* x,
                       const double* lambda, const double* mu, const double* penalty, double* f_out,
                       double* g_out, const double* ks_batch, const int32_t* parts) {
  if (n <= 0 || n > 1024) return -1;
  oracle::ConstrainedProblem prob = build_problem(n, n_eq, n_ineq, kinds, forms, ks, coef, parts);
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  for (int64_t b = 0; b < B; ++b) {
    set_constants(&prob, ks_batch, b);
    oracle::AugLagComposite c;
    c.prob = &prob;
    c.lambda.assign(lambda + b * n_eq, lambda + (b + 1) * n_eq);
    c.mu.assign(mu + b * n_ineq, mu + (b + 1) * n_ineq);
    c.rho = penalty[b];
    f_out[b] = c.eval(x + b * n, g_out + b * n, n, red);
  }
  return 0;
}

// Lbfgs::Minimize on the composite with fixed (lambda, mu, penalty) per row: what one augmented-Lagrangian step
// hands to its inner solver (twin of the engine's MI355_OBJ_AL_COMPOSITE objective).
int oracle_auglag_composite_minimize(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds,
                                     const int32_t* forms, const double* ks, const double* coef,
                                     const oracle_stop* stop, int m, int reduction, int width, const double* x0,
                                     const double* lambda, const double* mu, const double* penalty, double* x_out,
                                     double* f_out, double* g_out, oracle_progress* prog_out, int linesearch,
                                     const int32_t* parts) {
  if (n <= 0 || n > 1024 || B < 0) return -1;
  oracle::ConstrainedProblem prob = build_problem(n, n_eq, n_ineq, kinds, forms, ks, coef, parts);
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  for (int64_t b = 0; b < B; ++b) {
    oracle::AugLagComposite c;
    c.prob = &prob;
    c.lambda.assign(lambda + b * n_eq, lambda + (b + 1) * n_eq);
    c.mu.assign(mu + b * n_ineq, mu + (b + 1) * n_ineq);
    c.rho = penalty[b];
    oracle::Lbfgs solver(m, to_stop(stop), red);
    solver.linesearch = linesearch;
    oracle::Progress pr;
    const oracle::State sol = solver.Minimize(c, std::vector<double>(x0 + b * n, x0 + (b + 1) * n), &pr);
    std::copy(sol.x.begin(), sol.x.end(), x_out + b * n);
    f_out[b] = sol.value;
    if (g_out) std::copy(sol.gradient.begin(), sol.gradient.end(), g_out + b * n);
    if (prog_out) {
      prog_out[b].status = static_cast<int32_t>(pr.status);
      prog_out[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog_out[b].nfev = static_cast<uint32_t>(solver.nfev);
      prog_out[b].sum_k = static_cast<uint32_t>(solver.sum_k);
      prog_out[b].x_delta = pr.x_delta;
      prog_out[b].f_delta = pr.f_delta;
      prog_out[b].gradient_norm = pr.gradient_norm;
    }
  }
  return 0;
}

int oracle_auglag_minimize_batch(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms,
                                 const double* ks, const double* coef, const oracle_al_config* cfg,
                                 const oracle_stop* inner_stop, int m, int reduction, int width, double* x,
                                 double* lambda, double* mu, double* penalty, double* violation, double* kkt,
                                 oracle_al_progress* prog, int nthreads, int linesearch, const double* ks_batch,
                                 const int32_t* parts) {
  if (n <= 0 || n > 1024 || B < 0 || n_eq < 0 || n_ineq < 0) return -1;
  if (reduction == 1 && (width < n || width > 1024 || (width & (width - 1)))) return -1;
  oracle::ConstrainedProblem prob = build_problem(n, n_eq, n_ineq, kinds, forms, ks, coef, parts);
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  oracle::AugLagConfig config;
  config.penalty_growth_factor = cfg->penalty_growth_factor;
  config.violation_shrink_ratio = cfg->violation_shrink_ratio;
  config.auto_scale_initial_penalty = cfg->auto_scale_initial_penalty != 0;
  config.penalty_auto_objective_scale = cfg->penalty_auto_objective_scale;
  config.penalty_auto_min = cfg->penalty_auto_min;
  config.penalty_auto_max = cfg->penalty_auto_max;
  config.warmup_max_inner_iterations = cfg->warmup_max_inner_iterations;
  config.warmup_inner_gradient_tolerance = cfg->warmup_inner_gradient_tolerance;
  config.multiplier_max = cfg->multiplier_max;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
  for (int64_t b = 0; b < B; ++b) {
    oracle::ConstrainedProblem own = prob;  // (per-thread copy: the constants may differ per problem)
    set_constants(&own, ks_batch, b);
    oracle::Lbfgs inner(m, to_stop(inner_stop), red);
    inner.linesearch = linesearch;
    oracle::AugmentedLagrangian solver(&own, inner, red);
    solver.config = config;
    solver.stopping_progress.num_iterations = cfg->outer_num_iterations;
    solver.stopping_progress.constraint_threshold = cfg->constraint_threshold;
    solver.stopping_progress.kkt_stationarity_threshold = cfg->kkt_stationarity_threshold;
    oracle::AugLagState state;
    state.x.assign(x + b * n, x + (b + 1) * n);
    state.lambda.assign(lambda + b * n_eq, lambda + (b + 1) * n_eq);
    state.mu.assign(mu + b * n_ineq, mu + (b + 1) * n_ineq);
    state.penalty = penalty[b];
    state.max_violation = violation[b];  // in/out: the incoming state's value feeds the first penalty-growth test
    oracle::AugLagProgress pr;
    const oracle::AugLagState sol = solver.Minimize(state, &pr);
    std::copy(sol.x.begin(), sol.x.end(), x + b * n);
    std::copy(sol.lambda.begin(), sol.lambda.end(), lambda + b * n_eq);
    std::copy(sol.mu.begin(), sol.mu.end(), mu + b * n_ineq);
    penalty[b] = sol.penalty;
    violation[b] = sol.max_violation;
    kkt[b] = sol.max_lagrangian_gradient;
    if (prog) {
      prog[b].status = static_cast<int32_t>(pr.status);
      prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog[b].x_delta = pr.x_delta;
      prog[b].f_delta = pr.f_delta;
      prog[b].gradient_norm = pr.gradient_norm;
      prog[b].inner_iterations = pr.inner_iterations;
      prog[b].nfev = pr.nfev;
      prog[b].sum_k = pr.sum_k;
    }
  }
  return 0;
}

// Lbfgsb as the inner solver (lower / upper: n doubles or both null = SetBounds never called).
int oracle_auglag_box_minimize_batch(int n, int64_t B, int n_eq, int n_ineq, const int32_t* kinds, const int32_t* forms,
                                 const double* ks, const double* coef, const oracle_al_config* cfg,
                                 const oracle_stop* inner_stop, int m, int reduction, int width, double* x,
                                 double* lambda, double* mu, double* penalty, double* violation, double* kkt,
                                 oracle_al_progress* prog, int nthreads, int linesearch, const double* ks_batch,
                                     const double* lower, const double* upper, int std_sort_order,
                                     const int32_t* parts) {
  if (n <= 0 || n > 1024 || B < 0 || n_eq < 0 || n_ineq < 0) return -1;
  if (reduction == 1 && (width < n || width > 1024 || (width & (width - 1)))) return -1;
  oracle::ConstrainedProblem prob = build_problem(n, n_eq, n_ineq, kinds, forms, ks, coef, parts);
  oracle::Reducer red;
  red.kind = reduction ? oracle::Reduction::Butterfly : oracle::Reduction::Sequential;
  red.width = width;
  oracle::AugLagConfig config;
  config.penalty_growth_factor = cfg->penalty_growth_factor;
  config.violation_shrink_ratio = cfg->violation_shrink_ratio;
  config.auto_scale_initial_penalty = cfg->auto_scale_initial_penalty != 0;
  config.penalty_auto_objective_scale = cfg->penalty_auto_objective_scale;
  config.penalty_auto_min = cfg->penalty_auto_min;
  config.penalty_auto_max = cfg->penalty_auto_max;
  config.warmup_max_inner_iterations = cfg->warmup_max_inner_iterations;
  config.warmup_inner_gradient_tolerance = cfg->warmup_inner_gradient_tolerance;
  config.multiplier_max = cfg->multiplier_max;
#ifdef _OPENMP
  if (nthreads <= 0) nthreads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads)
#endif
  for (int64_t b = 0; b < B; ++b) {
    oracle::ConstrainedProblem own = prob;  // (per-thread copy: the constants may differ per problem)
    set_constants(&own, ks_batch, b);
    oracle::Lbfgsb inner(m, to_stop(inner_stop), red);
    inner.linesearch = linesearch;
    inner.std_sort_order = std_sort_order != 0;
    if (lower) {  // SetBounds
      inner.lower.assign(lower, lower + n);
      inner.upper.assign(upper, upper + n);
    }
    oracle::AugmentedLagrangianT<oracle::Lbfgsb> solver(&own, inner, red);
    solver.config = config;
    solver.stopping_progress.num_iterations = cfg->outer_num_iterations;
    solver.stopping_progress.constraint_threshold = cfg->constraint_threshold;
    solver.stopping_progress.kkt_stationarity_threshold = cfg->kkt_stationarity_threshold;
    oracle::AugLagState state;
    state.x.assign(x + b * n, x + (b + 1) * n);
    state.lambda.assign(lambda + b * n_eq, lambda + (b + 1) * n_eq);
    state.mu.assign(mu + b * n_ineq, mu + (b + 1) * n_ineq);
    state.penalty = penalty[b];
    state.max_violation = violation[b];  // in/out: the incoming state's value feeds the first penalty-growth test
    oracle::AugLagProgress pr;
    const oracle::AugLagState sol = solver.Minimize(state, &pr);
    std::copy(sol.x.begin(), sol.x.end(), x + b * n);
    std::copy(sol.lambda.begin(), sol.lambda.end(), lambda + b * n_eq);
    std::copy(sol.mu.begin(), sol.mu.end(), mu + b * n_ineq);
    penalty[b] = sol.penalty;
    violation[b] = sol.max_violation;
    kkt[b] = sol.max_lagrangian_gradient;
    if (prog) {
      prog[b].status = static_cast<int32_t>(pr.status);
      prog[b].num_iterations = static_cast<uint32_t>(pr.num_iterations);
      prog[b].x_delta = pr.x_delta;
      prog[b].f_delta = pr.f_delta;
      prog[b].gradient_norm = pr.gradient_norm;
      prog[b].inner_iterations = pr.inner_iterations;
      prog[b].nfev = pr.nfev;
      prog[b].sum_k = pr.sum_k;
    }
  }
  return 0;
}

}  // extern "C"
