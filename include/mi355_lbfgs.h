/* mi355_lbfgs.h — C-ABI of the MI355X (gfx950) batched L-BFGS engine.
 *
 * This is the drop-in boundary for the hot path of PatWie/CppNumericalSolvers
 * (cppoptlib 2.0.0):
 *
 *     Solver::Minimize            include/cppoptlib/solver/solver.h:181-224
 *       -> Lbfgs::OptimizationStep   include/cppoptlib/solver/lbfgs.h:89-303
 *       -> MoreThuente::Search/cvsrch/cstep
 *                                 include/cppoptlib/linesearch/more_thuente.h:120-407
 *       -> objective              function_base.h:94-126 (FunctionCRTP::operator())
 *       -> Progress::Update       include/cppoptlib/solver/progress.h:153-327
 *
 * The reference runs that chain for ONE problem on one CPU thread.  The entry
 * points below run it for a batch of B independent problems on one GPU; the
 * whole solve (all iterations, all line-search trials) happens inside one
 * fused HIP kernel, one problem per wavefront segment.
 *
 * Plain C: pointers, sizes, PODs.  No C++ / torch / Eigen types cross this
 * boundary and nothing throws across it.  All functions return 0 on success
 * or a negative mi355_status; mi355_lbfgs_last_error() describes the failure.
 *
 * Layouts: batch-major, row per problem:  x[B][n], g[B][n] (double),
 * f[B] (double), progress[B] (mi355_lbfgs_progress).
 */
#ifndef MI355_LBFGS_H_
#define MI355_LBFGS_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_LBFGS_ABI_VERSION 9

/* Error codes (return values). */
enum mi355_status {
  MI355_OK = 0,
  MI355_ERR_INVALID_ARGUMENT = -1, /* bad desc / null pointer / unsupported n, m */
  MI355_ERR_HIP = -2,              /* a HIP runtime call failed */
  MI355_ERR_NO_DEVICE = -3,        /* no gfx950 device / device index out of range */
  MI355_ERR_UNSUPPORTED = -4       /* objective / line search id not built in */
};

/* Per-problem solver status == cppoptlib::solver::Status
 * (solver/progress.h:37-47), same numeric values. */
enum mi355_solver_status {
  MI355_STATUS_NOT_STARTED = -1,
  MI355_STATUS_CONTINUE = 0,
  MI355_STATUS_ITERATION_LIMIT = 1,
  MI355_STATUS_X_DELTA_VIOLATION = 2,
  MI355_STATUS_F_DELTA_VIOLATION = 3,
  MI355_STATUS_GRADIENT_NORM_VIOLATION = 4,
  MI355_STATUS_HESSIAN_CONDITION_VIOLATION = 5,
  MI355_STATUS_FINISHED = 6
};

/* Device objective functors (replace the host FunctionCRTP functor,
 * function_base.h:94-126).  Shared parameters are passed as a blob of doubles. */
enum mi355_objective {
  /* chained Rosenbrock-N; equals src/test/verify.cc:58-69 at n = 2. params: none */
  MI355_OBJ_ROSENBROCK = 0,
  /* f(x) = sum_i a_i x_i^2 + c (README.md:21-28 quick start is a = {5,100}, c = 5).
   * params: a[0..n), c   (n + 1 doubles) */
  MI355_OBJ_DIAG_QUADRATIC = 1,
  /* ridge least squares f(x) = ||A x - y_b||^2 + lambda ||x||^2 with a shared A (rows x n) and
   * one right-hand side per problem: README.md:122-167 `SquaredError(A, y) + lambda * L2Reg(n)`
   * (First-mode branches of function_expressions.h:115-124, :229-236).
   * params: rows, lambda, A[rows][n] row major (2 + rows*n doubles), rows <= MI355_LBFGS_MAX_ROWS;
   * per_problem_data: y[B][per_problem_stride], per_problem_stride >= rows */
  MI355_OBJ_SQUARED_ERROR_RIDGE = 2,
  /* The same function and parameters as MI355_OBJ_SQUARED_ERROR_RIDGE with the two matrix-vector
   * products of every evaluation on the matrix cores (v_mfma_f64_16x16x4_f64): sixteen problems are
   * evaluated together, A x and A^T r accumulate as ascending fused-multiply-add chains (id 2 keeps
   * them as multiply-then-add sums, bit-identical to the README functors), everything else is the
   * same arithmetic.  Results agree with id 2 and with the reference within the 1e-6 tolerance.
   * n <= 64, rows <= 128, m <= 10, More-Thuente; solve entry points only. */
  MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA = 3,
  /* function::ToAugmentedLagrangian(problem, multipliers, penalty) (function_penalty.h:239-246) as an objective
   * of its own: what the reference hands to the inner solver of an augmented-Lagrangian step, and what a
   * penalty-method experiment minimises directly.  The problem is described by terms (see mi355_al_problem).
   * params: n_eq, n_ineq, rows, then per term (parts, form, k), then per row (kind, coefficient row [n + 1]):
   * 3 + 3 (1 + n_eq + n_ineq) + rows (n + 2) doubles; per_problem_data: rows (lambda[n_eq], mu[n_ineq], penalty),
   * per_problem_stride = n_eq + n_ineq + 1, or twice that with one constant k per term appended to every row.
   * Lbfgs solve entry points, m <= 10, either line search. */
  MI355_OBJ_AL_COMPOSITE = 4,
  /* The same function and parameters as MI355_OBJ_SQUARED_ERROR_RIDGE in NORMAL-EQUATION form: A is shared by the
   * batch, so f(x) = x^T G x - 2 c_b^T x + y_b^T y_b with one Gram matrix G = A^T A + lambda I (built once per launch,
   * held in LDS) and, per problem, c_b = A^T y_b and y_b^T y_b — a batched GEMM run once per problem on the matrix
   * cores (v_mfma_f64_16x16x4_f64) before the solve.  An evaluation is then n^2 multiply-adds (t = G x, grad =
   * 2 (t - c_b), f = x . (t - 2 c_b) + y_b^T y_b) instead of 2 rows n, without any coupling between problems: the
   * solve runs in the ordinary persistent Lbfgs kernel.  Algebraically the same function with different rounding (the
   * reference forms r = A x - y_b in every evaluation); x*, f* within 1e-6 of the reference while
   * cond(A^T A + lambda I) <~ 3e2 (tests/test_relaxed_envelope.py; beyond, the reference's own gradient test stops
   * 1e-6 ... 0.6 from the minimiser and the two forms agree in f* and in satisfying that test, not in x*).  G itself is
   * built on the matrix cores once per matrix; it is shared in LDS by the wavefronts of a workgroup up to n = 128 and
   * streamed through L2 up to n = 256.  Fused arithmetic only (MI355_ARITH_DEFAULT / MI355_ARITH_FMA),
   * n <= MI355_LBFGS_MAX_N, rows <= MI355_LBFGS_GRAM_MAX_ROWS, More-Thuente, mi355_lbfgs_minimize_batch[_host]. */
  MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM = 5,
  /* The ridge objective with ONE MATRIX PER PROBLEM: what a program computes that builds `SquaredError(A_b, y_b) + lambda *
   * L2Reg(n)` once per data set (README.md:126-160) and minimises each with Lbfgs.  params: rows, lambda (2 doubles);
   * per_problem_data row b = A_b (rows x n, row major) followed by y_b (per_problem_stride >= rows * n + rows).
   * Normal-equation form per problem: a pre-pass on the matrix cores writes G_b = A_b^T A_b + lambda I, c_b, y_b . y_b to a
   * per-problem row in device memory (P^2 + P + 2 doubles, P = next power of two >= max(n, 8): 33 KB at n = 64), the solve
   * streams its G_b on every evaluation; A_b is read once.  Same envelope as MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM.  Fused
   * arithmetic only, First mode, More-Thuente, n <= MI355_LBFGS_MAX_N, rows <= MI355_LBFGS_GRAM_MAX_ROWS,
   * mi355_lbfgs_minimize_batch[_host] and mi355_lbfgs_eval_batch. */
  MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM = 6,
  /* Ids from here on are USER objectives: device functors supplied as a header and compiled into a build of the
   * library by `cppnumericalsolvers_amd._build.build(user_objectives=[...])` (INTEGRATION.md section "user
   * objectives").  params / per_problem_data are handed to the functor's load() / begin_problem() untouched.  Lbfgs and
   * Bfgs solves (either line search), Lbfgsb solve (the shapes the build asked for; default m <= 5, n <= 64, More-Thuente)
   * and evaluation entry points. */
  MI355_OBJ_USER_FIRST = 100
};

enum mi355_linesearch {
  MI355_LS_MORE_THUENTE = 0, /* linesearch/more_thuente.h (the Lbfgs default, lbfgs.h:41) */
  MI355_LS_HAGER_ZHANG = 1   /* linesearch/hager_zhang.h, the alternative LineSearch template argument */
};

/* Stopping criteria: the fields of cppoptlib::solver::Progress that the
 * stopping test reads (solver/progress.h:87-136), flattened to a POD.
 * mi355_lbfgs_default_stop() fills the reference presets. */
typedef struct mi355_lbfgs_stop {
  uint64_t num_iterations;        /* 0 disables; stops when iterations > limit (strict) */
  double x_delta;                 /* ||x+ - x||_inf threshold, 0 disables */
  int32_t x_delta_violations;     /* consecutive violations needed */
  double f_delta;                 /* |f+ - f| threshold, 0 disables */
  int32_t f_delta_violations;
  int32_t f_delta_relative;       /* scale f_delta by max(|f+|,|f|,1) */
  double gradient_norm;           /* ||g||_inf threshold, 0 disables */
  int32_t gradient_norm_relative; /* scale by max(1, ||x||_inf) */
  int32_t past;                   /* plateau window, 0 disables, <= MI355_LBFGS_MAX_PAST */
  double past_delta;
} mi355_lbfgs_stop;

#define MI355_LBFGS_MAX_PAST 8
#define MI355_LBFGS_MAX_N 256   /* largest dimension of the wavefront-resident kernels (every solver of this header) */
/* Above it mi355_lbfgs_minimize_batch[_host] runs a problem on a WORKGROUP with its vectors and correction ring in an HBM
 * workspace (csrc/lbfgs_wide_kernel.hpp): Lbfgs<F, m, LineSearch> (either line search), First mode or hessian_from_functor, exact arithmetic,
 * Rosenbrock / DiagQuadratic and user objectives built with a functor for this regime, any n up to this bound (the
 * reference is dynamic in n). */
#define MI355_LBFGS_WIDE_MAX_N 16777216
#define MI355_LBFGS_MAX_M 32    /* largest history size */
#define MI355_LBFGS_MAX_ROWS 128 /* largest residual count of MI355_OBJ_SQUARED_ERROR_RIDGE / _MFMA */
#define MI355_LBFGS_GRAM_MAX_ROWS 4096 /* ... of MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM (any n <= MI355_LBFGS_MAX_N) */

/* Per-problem result == the observable fields of Progress after Minimize
 * (solver/progress.h:87-127) + nfev / sum_k accounting the reference lacks. */
typedef struct mi355_lbfgs_progress {
  int32_t status;          /* mi355_solver_status */
  uint32_t num_iterations; /* outer iterations taken */
  uint32_t nfev;           /* objective evaluations (initial one included) */
  uint32_t sum_k;          /* sum over iterations of stored (s,y) pairs used */
  double x_delta;          /* last ||x+ - x||_inf */
  double f_delta;          /* last |f+ - f| */
  double gradient_norm;    /* last ||g||_inf */
} mi355_lbfgs_progress;

/* Arithmetic of the L-BFGS solve kernels.
 * MI355_ARITH_EXACT  no fused multiply-add anywhere: every a*b+c is a rounded product and a rounded sum in the
 *                    operation order of the reference's scalar code; bit-identical to the CPU twin the tests
 *                    keep (pairwise summation trees) and, up to the summation tree of the inner products, to the reference built without
 *                    contraction.  The pinning mode, and what MI355_ARITH_DEFAULT means for every solver except
 *                    mi355_lbfgs_minimize_batch with the More-Thuente line search on an objective with a fused form.
 * MI355_ARITH_FMA    inner products, the axpys of the two-loop recursion, the line search's trial point and the
 *                    objective use fused multiply-adds (what an optimising build of the reference does to its own
 *                    loops: GCC contracts by default).  10-20 % fewer VALU instructions per iteration.  Results agree
 *                    with MI355_ARITH_EXACT and with the reference-order solve within the 1e-6 tolerance of the
 *                    north star (not bit for bit: the rounding of d differs from the first iteration on), and stay
 *                    bit-identical to a CPU twin that fuses the same operations.  Lbfgs + More-Thuente on the Rosenbrock and
 *                    DiagQuadratic objectives, the solver side of the matrix-core ridge objective (and user objectives that
 *                    define eval_fma); MI355_ERR_UNSUPPORTED elsewhere.
 *                    On the mi355_lbfgsb_* entry points the same value selects the RELAXED-ALGEBRA kernels
 *                    (csrc/lbfgsb_fast_kernel.hpp): besides the fused multiply-adds, the 2m x 2m algebra of the compact
 *                    representation is re-derived — ring history in a fixed [Y | S] layout, MM (lbfgsb.h:227-234) factored
 *                    without pivoting, M^-1 c / M^-1 p of the breakpoint loop by linearity (one solve per breakpoint,
 *                    :388-390), and v of :486-500 as ONE elimination with K = MM - WZ WZ^T / theta assembled over the active
 *                    coordinates.  Algebraically the reference's iteration; x*, f* within 1e-6 of the reference binary,
 *                    bit-identical to its own CPU twin (kept with the tests); 2.4 x the
 *                    throughput of the reference-order build on configs[4].  Built for the More-Thuente line search on
 *                    Rosenbrock / DiagQuadratic (and user functors with an eval_fma), m <= 10 (n <= 64; m = 9, 10 on
 *                    thirty-two lanes per problem), m <= 5 (n <= 128);
 *                    it is what MI355_ARITH_DEFAULT selects there.  MI355_ARITH_EXACT keeps the reference's operation order. */
/* Envelope of the relaxed L-BFGS-B default: on the DiagQuadratic objective (whose spectrum the library can read off its
 * parameters) MI355_ARITH_DEFAULT selects the relaxed-algebra kernel only while max|a_i| <= this x min|a_i| — the range in
 * which it is pinned to 1e-6 of the reference binary (tests/test_relaxed_envelope.py; at a spread of 1e6 the two are
 * 1.8e-6 apart, the reference itself being 2e-6 from the minimiser).  Beyond it the default is the reference-order
 * kernel; MI355_ARITH_FMA still forces the relaxed one. */
#define MI355_LBFGSB_RELAXED_MAX_SPREAD 1.0e4
/* The normal-equation forms of the ridge objective (ids 5, 6) are pinned to 1e-6 of the reference while a rigorous bound of
 * cond(A^T A + lambda I) — Gershgorin row sums of G over lambda — stays below this.  The Python driver
 * (`gram="auto"`) and the C++ classes (own-matrix batches under MI355_ARITH_DEFAULT, a sample of the batch) hold their
 * inputs against it; the C entry points take the objective id the caller names. */
#define MI355_RIDGE_GRAM_MAX_CONDITION_BOUND 3.0e2

enum mi355_arithmetic {
  MI355_ARITH_DEFAULT = 0, /* the library's choice: MI355_ARITH_FMA where it is built (and, for mi355_lbfgsb_* on
                              DiagQuadratic, inside MI355_LBFGSB_RELAXED_MAX_SPREAD), else MI355_ARITH_EXACT */
  MI355_ARITH_EXACT = 1,
  MI355_ARITH_FMA = 2
};

/* Opt-in per-iteration trace of chosen problems: what Solver::step_callback_ of the reference observes
 * (solver/solver.h:197, :222), recorded on the device and read back after the solve.  Traced problem i keeps the
 * records of its last `capacity` iterations in a ring: iteration t (1-based, the t-th Progress::Update) is record
 * (t - 1) % capacity of row i; written[i] counts the iterations recorded (it exceeds capacity when the ring wrapped).
 * x / g, when given, hold the iterate and its gradient after the same iterations.  Tracing costs one uniform branch
 * per iteration when off.  Lbfgs / Bfgs / Lbfgsb solve entry points (not the matrix-core ridge objective, not the
 * augmented-Lagrangian loop). */
#define MI355_LBFGS_MAX_TRACED 64
typedef struct mi355_lbfgs_trace_record {
  uint32_t num_iterations; /* the iteration this record describes (Progress::num_iterations after its Update) */
  int32_t status;          /* mi355_solver_status after its Update */
  double value;            /* f at the new iterate */
  double x_delta, f_delta, gradient_norm; /* solver/progress.h:188-195 */
} mi355_lbfgs_trace_record;
typedef struct mi355_lbfgs_trace {
  int32_t count;                     /* traced problems, 1..MI355_LBFGS_MAX_TRACED */
  int32_t capacity;                  /* records per traced problem, >= 1 */
  const int64_t* problems;           /* HOST [count]: batch indices of the traced problems (distinct) */
  mi355_lbfgs_trace_record* records; /* DEVICE [count][capacity] */
  double* x;                         /* DEVICE [count][capacity][n], or NULL */
  double* g;                         /* DEVICE [count][capacity][n], or NULL */
  uint32_t* written;                 /* DEVICE [count]; zeroed by the library before the launch */
} mi355_lbfgs_trace;

/* One batched solve.  Replaces the template parameters and ctor arguments of
 * cppoptlib::solver::Lbfgs<FunctionType, m, LineSearch> (lbfgs.h:40-45). */
typedef struct mi355_lbfgs_desc {
  int32_t objective;            /* mi355_objective */
  int32_t linesearch;           /* mi355_linesearch */
  int32_t n;                    /* problem dimension, 1..MI355_LBFGS_MAX_N (Lbfgs: ..MI355_LBFGS_WIDE_MAX_N) */
  int32_t m;                    /* history size (lbfgs.h:40 default 10), 1..MI355_LBFGS_MAX_M */
  const double* objective_params; /* HOST pointer, n_params doubles (may be NULL if 0) */
  int32_t n_params;
  /* Per-problem objective data, row b belongs to problem b (NULL if the objective has none).
   * A DEVICE pointer for mi355_lbfgs_minimize_batch / _eval_batch, a HOST pointer for
   * mi355_lbfgs_minimize_batch_host. */
  const double* per_problem_data;
  int32_t per_problem_stride;   /* doubles per row */
  /* Mapping of one problem onto a wavefront: lanes per problem (a power of two
   * 8..64; 64 = one problem per wavefront) and elements per lane (1,2,4).
   * 0/0 lets the library choose.  Results do not depend on this choice. */
  int32_t lanes_per_problem;
  int32_t elems_per_lane;
  /* Where the (s, y) history lives: MI355_HISTORY_AUTO lets the library choose,
   * MI355_HISTORY_LDS keeps both halves in LDS, MI355_HISTORY_Y_IN_REGISTERS keeps the
   * y half in registers (kernels of 5, 6 and 10 columns serve m <= 10 with elems_per_lane >= 2; other shapes
   * fall back to LDS).  Results do not depend on this choice either. */
  int32_t history_placement;
  int32_t arithmetic;           /* mi355_arithmetic; 0 = library default */
  /* Second-mode functions whose Hessian is NOT constant: 1 = the diagonal preconditioner is rebuilt at every iterate
   * from the device functor's own hess_diag (diag H at the current x), as the reference re-evaluates
   * function(x, &g, &H) in every step (lbfgs.h:129-138).  hessian_diagonal must then be NULL.  Built for
   * mi355_lbfgs_minimize_batch on objectives whose functor has a hess_diag: Rosenbrock, and user functors that define
   * one; the history is kept in LDS (history_placement is ignored).  0 = First mode, or the constant diagonal below.
   * The condition_hessian stopping test in this mode (hessian_condition_stop > 0; hessian_condition is ignored): the solve
   * kernel builds H(x) of every new iterate in LDS from the functor's hess_full and evaluates ||H||_F ||H^-1||_F itself
   * (LU with partial pivoting, csrc/hessian_condition_device.hpp) — Lbfgs at n <= 64, either line search, at most two
   * coordinates per lane (the library's choice when lanes_per_problem is 0); refused (MI355_ERR_UNSUPPORTED) for larger
   * n, for mi355_bfgs_minimize_batch and for functors without a hess_full. */
  int32_t hessian_from_functor;
  /* Second-mode functions (lbfgs.h:116-139, :177-179): HOST pointer to the n diagonal entries
   * H_jj of the (constant) Hessian.  When non-NULL the two-loop recursion is centred on
   * diag(1 / (|H_jj| + eps)) instead of the scalar s.y / y.y, exactly like the reference's
   * `if constexpr (Differentiability == Second)` branch; NULL = First-mode path.  The
   * reference re-evaluates the Hessian every iteration; this array serves constant Hessians (quadratic
   * objectives such as the README ridge example), hessian_from_functor above the others; the condition_hessian
   * stopping test is driven by hessian_condition / hessian_condition_stop below. */
  const double* hessian_diagonal;
  const mi355_lbfgs_trace* trace; /* NULL = no trace */
  /* Second-mode functions only (hessian_diagonal != NULL): the condition_hessian stopping test of Progress::Update
   * (solver/progress.h:203-210, :318-325).  hessian_condition = ||H||_F ||H^-1||_F of the (constant) Hessian, e.g. from
   * mi355_lbfgs_hessian_condition(); hessian_condition_stop = stopping_progress.condition_hessian (0 = off, the
   * reference's default).  When the test is on and the condition exceeds the threshold, a solve stops with
   * MI355_STATUS_HESSIAN_CONDITION_VIOLATION at the first iteration no earlier test stops — the reference evaluates the
   * (constant) condition number in every Update and tests it last. */
  double hessian_condition;
  double hessian_condition_stop;
  mi355_lbfgs_stop stop;
} mi355_lbfgs_desc;

enum mi355_history_placement {
  MI355_HISTORY_AUTO = 0,
  MI355_HISTORY_LDS = 1,
  MI355_HISTORY_Y_IN_REGISTERS = 2
};

/* One context per device and per stream of solves: it owns the work-queue head, scratch buffers and timing
 * events of the solve in flight, so solves issued through one context must be ordered on one stream (or
 * separated by a synchronisation); use several contexts for concurrent streams.  Not thread-safe. */
typedef struct mi355_lbfgs_ctx mi355_lbfgs_ctx;

/* ---- lifetime ----------------------------------------------------------- */
int mi355_lbfgs_abi_version(void);
/* Creates a context on HIP device `device`.  Fails (MI355_ERR_NO_DEVICE) when no
 * GPU is visible: there is no CPU fallback. */
int mi355_lbfgs_create(int device, mi355_lbfgs_ctx** out);
void mi355_lbfgs_destroy(mi355_lbfgs_ctx* ctx);
/* Thread-local description of the last failure on this thread ("" if none). */
const char* mi355_lbfgs_last_error(void);

/* preset 0: DefaultStoppingSolverProgress (solver/progress.h:353-431)
 * preset 1: ConservativeStoppingSolverProgress (solver/progress.h:456-464)
 * preset 2: what a default-constructed Lbfgsb uses (solver/lbfgsb.h:84-87): preset 0 with
 *           f_delta = 2.22e-9, f_delta_relative = 1 */
int mi355_lbfgs_default_stop(int preset, mi355_lbfgs_stop* out);

/* ---- the hot path -------------------------------------------------------- */
/* Batched Lbfgs::Minimize.  All array pointers are DEVICE pointers on ctx's
 * device; g_out and progress_out may be NULL.  Asynchronous on `stream`
 * (a hipStream_t, NULL = default stream); the caller synchronises.
 * Replaces: solver.Minimize(f, FunctionState(x0)) — solver/solver.h:181-224. */
int mi355_lbfgs_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B,
                               const double* x0, double* x_out, double* f_out, double* g_out,
                               mi355_lbfgs_progress* progress_out, void* stream);

/* Same with HOST pointers (pageable or pinned), synchronous.  The batch goes through pinned staging and device
 * buffers the context owns (no allocation per call once warm): parallel host copies, asynchronous H2D / solve / D2H
 * on the context's own streams; batches beyond a staging slot (256 MiB) are solved in chunks, chunk c + 1 being
 * staged and solved while chunk c travels back.  desc->per_problem_data and the array pointers of desc->trace are
 * HOST pointers here. */
int mi355_lbfgs_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B,
                                    const double* x0, double* x_out, double* f_out, double* g_out,
                                    mi355_lbfgs_progress* progress_out);

/* Batched Lbfgsb::Minimize — box-constrained L-BFGS-B (solver/lbfgsb.h:141-292, Cauchy point
 * :318-430, subspace minimisation :459-515).  `lower` / `upper`: n doubles each, shared by the
 * batch (SetBounds, lbfgsb.h:89-93), or both NULL for the reference's default unbounded box
 * (:124-129).  desc->stop.gradient_norm is the PROJECTED-gradient tolerance, an absolute
 * sup-norm test on the iterate the last step started from (:165-166, :280-283).
 * Built for desc->m <= 10 (5 is the reference default, lbfgsb.h:44) on the Rosenbrock and DiagQuadratic objectives up to
 * n = 256 with the More-Thuente search, with Hager-Zhang up to n = 256 for m <= 5 and up to n = 128 for m = 6..10, and on
 * the SquaredErrorRidge objective (n <= 64, More-Thuente); user objectives: the shapes their build asked for (default
 * m <= 5, n <= 64, More-Thuente).  Other shapes return MI355_ERR_UNSUPPORTED.
 * desc->lanes_per_problem / elems_per_lane / history_placement must be 0.
 * Device pointers, asynchronous on `stream`. */
int mi355_lbfgsb_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, const double* lower,
                                const double* upper, int64_t B, const double* x0, double* x_out,
                                double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                                void* stream);
/* Same with HOST pointers (bounds included), synchronous.  NaN bounds are refused (MI355_ERR_INVALID_ARGUMENT): they
 * make the breakpoint order undefined in the reference as well (std::sort over NaN keys). */
int mi355_lbfgsb_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, const double* lower,
                                     const double* upper, int64_t B, const double* x0, double* x_out,
                                     double* f_out, double* g_out, mi355_lbfgs_progress* progress_out);

/* Dense BFGS: replaces cppoptlib::solver::Bfgs<FunctionType, LineSearch>::Minimize (solver/bfgs.h:65-137 under
 * Solver::Minimize, solver/solver.h:181-224) for B problems at once — same driver, line searches
 * (desc->linesearch) and stopping tests as mi355_lbfgs_minimize_batch, with an explicit n x n inverse-
 * Hessian approximation per problem (in LDS) instead of the (s, y) history.  desc->m, history_placement and
 * hessian_diagonal are not used.  Mapping fields: 0 x 0 = the library's choice (since round 6 one coordinate — one column
 * of H — per lane at the padded widths 32 and 64: H's LDS footprint caps the problems in flight, so a problem's O(n^2)
 * work goes to as many lanes as it has columns), or a built split of the padded width P = 8 / 16 / 32 / 64:
 * lanes x elems in 8x{1,2,4}, 16x{2,4}, 32x{1,2}, 64x1 with lanes * elems == P (others: MI355_ERR_INVALID_ARGUMENT).
 * Results do not depend on the split.  n <= 64; Rosenbrock / DiagQuadratic / user objectives (those: the packed splits
 * 8x{1,2,4}, 16x4 their generated units hold).  progress.sum_k is 0. */
int mi355_bfgs_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                              double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                              void* stream);
int mi355_bfgs_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                   double* x_out, double* f_out, double* g_out,
                                   mi355_lbfgs_progress* progress_out);

/* Duration in ms of the most recent solve kernel on this context, measured with
 * HIP events recorded on the launch stream; blocks until that kernel finished. */
int mi355_lbfgs_last_kernel_ms(mi355_lbfgs_ctx* ctx, float* ms);
/* Launch geometry actually used by the most recent solve (for reports). */
int mi355_lbfgs_last_launch(mi355_lbfgs_ctx* ctx, int32_t* lanes_per_problem,
                            int32_t* elems_per_lane, int32_t* blocks, int32_t* threads,
                            int32_t* lds_bytes, int32_t* y_columns_in_registers);
/* Host helper for bindings: condition_out = ||H||_F * ||H^-1||_F of the symmetric n x n matrix `hessian` (HOST, n*n
 * doubles) — progress.h:208 `current_hessian.norm() * current_hessian.inverse().norm()`: Frobenius norms, the inverse
 * by LU with partial pivoting and column-by-column solves.  A singular matrix gives inf or NaN, as in the reference. */
int mi355_lbfgs_hessian_condition(const double* hessian, int32_t n, double* condition_out);
/* The mi355_arithmetic (MI355_ARITH_EXACT or MI355_ARITH_FMA) the most recent solve on this context ran with. */
int mi355_lbfgs_last_arithmetic(mi355_lbfgs_ctx* ctx, int32_t* arithmetic);

/* One Hager-Zhang line search per problem: replaces HagerZhang<F, Ord>::Search, State overload
 * (linesearch/hager_zhang.h:100-116), i.e. hzls (:282-548) from x[b] along direction[b] with the
 * initial step alpha_init[b].  Outputs the accepted point with its value and gradient (the start
 * state when the search fails) and the step width (0 on failure; alpha_init unchanged when the
 * direction is not a descent direction, :302).  nfev_out (objective evaluations, start point not
 * counted) and g_out may be NULL.  desc->m / stop / linesearch are ignored.  Device pointers + stream,
 * or host pointers (synchronous; objectives without per-problem data). */
int mi355_lbfgs_hz_search_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x,
                                const double* direction, const double* alpha_init, double* x_out, double* f_out,
                                double* g_out, double* alpha_out, uint32_t* nfev_out, void* stream);
int mi355_lbfgs_hz_search_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x,
                               const double* direction, const double* alpha_init, double* x_out, double* f_out,
                               double* g_out, double* alpha_out, uint32_t* nfev_out);

/* ---- more than one GPU ------------------------------------------------------- */
/* A group: one context per entry of `devices` (an entry may repeat a device: two contexts then share it) and an RCCL
 * communicator over the distinct devices (ncclCommInitAll; librccl.so is loaded at run time).  The batch shards
 * trivially — member s owns the contiguous range [B s / G, B (s + 1) / G), no reference code couples two problems —
 * and the only collective of the path is the all-reduce of the 3-word convergence record.
 * Testing on a one-GPU box: with MI355_GROUP_DRY_RUN_RANKS=1 in the environment at creation every member is a RANK of its
 * own even when members share a device, and the all-reduce among the ranks runs as a host-side sum instead of
 * ncclAllReduce — the D > 1 code paths of the group (per-rank records and flag buffers, the agreement check) run end to
 * end without a second GPU; RCCL itself is then not involved. */
typedef struct mi355_lbfgs_group mi355_lbfgs_group;
int mi355_lbfgs_group_create(const int* devices, int n_devices, mi355_lbfgs_group** out);
void mi355_lbfgs_group_destroy(mi355_lbfgs_group* group);
int mi355_lbfgs_group_size(const mi355_lbfgs_group* group);
/* Member `index`'s context (owned by the group): for device-resident shards, drive it with the entry points above. */
mi355_lbfgs_ctx* mi355_lbfgs_group_context(mi355_lbfgs_group* group, int index);
/* Batched Lbfgs::Minimize over the whole group, HOST arrays (as mi355_lbfgs_minimize_batch_host): one host thread per
 * member stages, solves and un-stages its shard on its own context, then the devices all-reduce
 * flag_out = {problems, unconverged (status IterationLimit / Continue / NotStarted), iterations} with ncclAllReduce
 * (ncclUint64, ncclSum) — `unconverged == 0` is the global stop flag.  flag_out / g_out / progress_out may be NULL. */
int mi355_lbfgs_group_minimize_batch_host(mi355_lbfgs_group* group, const mi355_lbfgs_desc* desc, int64_t B,
                                          const double* x0, double* x_out, double* f_out, double* g_out,
                                          mi355_lbfgs_progress* progress_out, uint64_t* flag_out /*[3]*/);
/* The same for Lbfgsb (solver/lbfgsb.h:247-292 overrides the same Minimize; lower / upper: n HOST doubles each, or both
 * NULL for the reference's default box) and for the dense Bfgs. */
int mi355_lbfgsb_group_minimize_batch_host(mi355_lbfgs_group* group, const mi355_lbfgs_desc* desc, const double* lower,
                                           const double* upper, int64_t B, const double* x0, double* x_out,
                                           double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                                           uint64_t* flag_out /*[3]*/);
int mi355_bfgs_group_minimize_batch_host(mi355_lbfgs_group* group, const mi355_lbfgs_desc* desc, int64_t B,
                                         const double* x0, double* x_out, double* f_out, double* g_out,
                                         mi355_lbfgs_progress* progress_out, uint64_t* flag_out /*[3]*/);
/* DEVICE-RESIDENT shards ("per-GPU device buffers"): every array argument is an array of
 * mi355_lbfgs_group_size() per-member pointers; member s solves counts[s] problems whose x0[s] / x_out[s] / f_out[s] /
 * progress_out[s] (required: the convergence record is counted from it) / g_out[s] / per_problem[s] (optional: the
 * array or the entry may be NULL) live on member s's device.  Each member's solve and the kernel that counts its record
 * run on that member's own stream; only the 24-byte record crosses PCIe, and the devices all-reduce it (ncclAllReduce)
 * into flag_out.  Synchronous: returns when every member has finished.  desc->per_problem_data is ignored. */
int mi355_lbfgs_group_minimize_batch(mi355_lbfgs_group* group, const mi355_lbfgs_desc* desc, const int64_t* counts,
                                     const double* const* x0, double* const* x_out, double* const* f_out,
                                     double* const* g_out, mi355_lbfgs_progress* const* progress_out,
                                     const double* const* per_problem, uint64_t* flag_out /*[3]*/);
/* ... and Lbfgsb: lower / upper are arrays of per-member DEVICE pointers (n doubles each), or both NULL. */
int mi355_lbfgsb_group_minimize_batch(mi355_lbfgs_group* group, const mi355_lbfgs_desc* desc,
                                      const double* const* lower, const double* const* upper, const int64_t* counts,
                                      const double* const* x0, double* const* x_out, double* const* f_out,
                                      double* const* g_out, mi355_lbfgs_progress* const* progress_out,
                                      const double* const* per_problem, uint64_t* flag_out /*[3]*/);
/* The collective alone, for shards that stay in HBM: progress_dev[s] (DEVICE array on member s's device, counts[s]
 * records; NULL when counts[s] is 0) is counted by a small kernel on its own device and the records are all-reduced.
 * The call first waits for each member device to be idle (hipDeviceSynchronize), so solves the caller enqueued on ANY
 * stream of those devices before the call are complete when their progress records are counted. */
int mi355_lbfgs_group_allreduce_flags(mi355_lbfgs_group* group, const mi355_lbfgs_progress* const* progress_dev,
                                      const int64_t* counts, uint64_t* flag_out /*[3]*/);

/* ---- synthetic workload + self checks (used by bench / tests) ------------- */
/* Fills x0[B][n] on the device with the seeded benchmark start points:
 * kind 0 ("std"): base_i + 0.1*(2u-1), base = (-1.2, 1, -1.2, 1, ...);
 * kind 1 ("u2"):  -2 + 4u;  u = uniform01(splitmix64(seed ^ (first+b)*n+i)). */
int mi355_lbfgs_fill_x0(mi355_lbfgs_ctx* ctx, int32_t kind, uint64_t seed, int64_t first_problem,
                        int64_t B, int32_t n, double* x0, void* stream);
/* One objective evaluation per problem (value + gradient), device pointers. */
int mi355_lbfgs_eval_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B,
                           const double* x, double* f_out, double* g_out, void* stream);
/* Device MoreThuente::cstep on `count` independent 13-value records
 * {stx,fx,dx,sty,fy,dy,stp,fp,dp,brackt,stpmin,stpmax,info} (in/out, doubles);
 * ret_out[count] receives cstep's return value.  Device pointers. */
int mi355_lbfgs_cstep_batch(mi355_lbfgs_ctx* ctx, int64_t count, double* records, int32_t* ret_out,
                            void* stream);
/* Same with HOST pointers (copies in/out, synchronous): lets host-side unit tests in the
 * style of the reference's src/test/cstep_test.cc drive the device cstep. */
int mi355_lbfgs_cstep_host(mi355_lbfgs_ctx* ctx, int64_t count, double* records, int32_t* ret_out);
/* Cross-lane self test: writes 10 x 64 int32 source-lane maps of the DPP /
 * permlane primitives the kernels use, then 64 doubles of sqrt/div probes. */
int mi355_lbfgs_selftest(mi355_lbfgs_ctx* ctx, int32_t* lane_maps /*[14][64] device*/,
                         const double* probe_in /*[64] device*/, double* probe_out /*[128] device*/,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Augmented Lagrangian.
 *
 * Replaces cppoptlib::solver::AugmentedLagrangian<ConstrainedOptimizationProblem, Lbfgs<...>>::Minimize
 * (solver/augmented_lagrangian.h:216-560: OptimizationStep, ComputeAutoScaledPenalty, ConfigureInnerSubproblem,
 * the multiplier clamps, ComputeLagrangianGradientKktNorm, the best-iterate filter and the Minimize wrapper),
 * function::ToAugmentedLagrangian (function_penalty.h:97-246) and the constrained branch of Progress::Update
 * (solver/progress.h:162-252) for a batch of B problems that share the problem definition and differ in the
 * start (x, lambda, mu, penalty).
 *
 * The reference composes arbitrary host functors; the device evaluates TERMS from a closed menu plus the user
 * functors a build of the library was given (MI355_AL_TERM_USER).  Term 0 is the
 * objective, terms 1..n_eq the equalities c(x) = 0, the next n_ineq the inequalities g(x) >= 0.  Term t is a
 * primitive (or a sum of primitives, see `parts`) — row r is kinds[r] over coef[r*(n+1) .. r*(n+1)+n] — combined
 * with the constant ks[t] as forms[t] says: the expression a reference user writes as `F`, `F - k` or `k - F`
 * (function_expressions.h:497-518; src/examples/constrained_simple2.cc:56-62 is `circle - 2.0`, `2.0 - circle`). */
#define MI355_AL_MAX_CONSTRAINTS 4 /* per kind, as TERMS of the table; affine constraints beyond that: families */
#define MI355_AL_MAX_FAMILY 256    /* family constraints of one problem (mi355_auglag_family_capacity(n) <= this) */
#define MI355_AL_MAX_ROWS 16       /* primitives in the table (a term may be a sum of several) */
#define MI355_AL_PARTS_PRODUCT (-2) /* mi355_al_problem.parts[t]: term t is the product of the next two primitives */

typedef enum mi355_al_term_kind {
  MI355_AL_TERM_ROSENBROCK = 0,     /* chained Rosenbrock (as MI355_OBJ_ROSENBROCK)                     */
  MI355_AL_TERM_DIAG_QUADRATIC = 1, /* sum_i (a_i x_i) x_i + c, gradient (2 a_i) x_i;  row = a[n], c     */
  MI355_AL_TERM_LINEAR = 2,         /* a.dot(x), gradient a;                            row = a[n]        */
  MI355_AL_TERM_SQUARED_NORM = 3,   /* x.squaredNorm(), gradient 2 x                                      */
  /* One residual of a least-squares function (ABI 9): r = a.dot(x) - c, value r * r, gradient (2 r) a.  A sum of them
   * (`parts`) is ||A x - y||^2 — the objective of src/examples/linear_regression.cc:14-39 as the objective TERM of its
   * augmented-Lagrangian half (:86-104). */
  MI355_AL_TERM_SQUARED_AFFINE = 4, /* (a.dot(x) - c)^2, gradient (2 (a.dot(x) - c)) a;    row = a[n], c     */
  /* kinds[r] >= MI355_AL_TERM_USER (= MI355_OBJ_USER_FIRST): the objective id of a USER device functor compiled into
   * this build of the library as a term (build option al_term): value and gradient come from the
   * functor's eval, its parameters are the row's n + 1 coefficients (or mi355_al_problem.user_params, see there).  This
   * is how the reference's non-convex tests
   * (src/test/augmented_lagrangian_test.cc:945-1150: HS024, HS029) are written for the device.  A library without
   * that functor answers MI355_ERR_UNSUPPORTED. */
  MI355_AL_TERM_USER = 100
} mi355_al_term_kind;

typedef enum mi355_al_term_form {
  MI355_AL_FORM_PLAIN = 0,         /* F        */
  MI355_AL_FORM_VALUE_MINUS_K = 1, /* F - k    */
  MI355_AL_FORM_K_MINUS_VALUE = 2  /* k - F    */
} mi355_al_term_form;

/* ConstrainedOptimizationProblem (function_problem.h:44-74); all pointers are HOST memory, copied per call. */
typedef struct mi355_al_problem {
  int32_t n;             /* dimension, 1..MI355_LBFGS_MAX_N */
  int32_t n_eq, n_ineq;  /* 0..MI355_AL_MAX_CONSTRAINTS each; terms = 1 + n_eq + n_ineq */
  const int32_t* kinds;  /* [rows] mi355_al_term_kind of every primitive */
  const int32_t* forms;  /* [terms] mi355_al_term_form */
  const double* ks;      /* [terms] */
  const double* coef;    /* [rows][n + 1] */
  /* [terms] number of primitives summed into each term — `F1 + F2 + ...`, the reference's AddExpression
   * (function_expressions.h:91-143: value fx_f + fx_g, gradient grad_f + grad_g, left to right) — whose rows follow
   * those of the previous term; NULL = one primitive per term (rows = terms).  rows <= MI355_AL_MAX_ROWS.
   * parts[t] = MI355_AL_PARTS_PRODUCT makes term t the PRODUCT of its two rows — `F1 * F2`, the reference's
   * ProdExpression (function_expressions.h:260-315: value fx * gx, gradient gx * grad_f + fx * grad_g). */
  const int32_t* parts;
  /* Parameters of the problem's USER term functors that ask for them (ABI 7): a functor that declares
   * `kTermParamsFromProblem` receives this blob in load() instead of its row's coefficients — the SAME blob it takes
   * as an objective through mi355_lbfgs_desc.objective_params, so one functor serves both as `Lbfgsb<F>`'s objective
   * and as a term here (the dense dual SVM of src/examples/svm_dual_al.cc:36-60: [n, Q]).  HOST memory, copied per
   * call like the rest of the description; NULL / 0 = none. */
  const double* user_params;
  int64_t user_params_count;
  /* Constraint FAMILIES (ABI 8): beyond the table's at most MI355_AL_MAX_CONSTRAINTS terms per kind, a problem may carry
   * n_family_eq equalities and n_family_ineq inequalities given as MATRICES — row i of family_eq / family_ineq is
   * (a_i[0..n), k_i) and stands for the affine constraint  c_i(x) = a_i . x - k_i  (= 0, resp. >= 0): what a reference
   * user writes as `LinearFunctor(a_i) - k_i` (function_expressions.h:497-518) and pushes, one per data point, into the
   * constraint vectors of a ConstrainedOptimizationProblem (function_problem.h:57-84; src/examples/svm_primal_al.cc:139-147
   * builds 2 x 100 of them).  In the problem's constraint order the family rows FOLLOW the table's terms of their kind:
   * lambda is [B][n_eq + n_family_eq], mu is [B][n_ineq + n_family_ineq].  A family constraint is evaluated as the
   * ascending multiply-then-add chain of the reference's `a.dot(x)` and every sum over constraints runs in ascending
   * order, so the family part of the composite is the reference's own arithmetic under every policy.
   * n_family_eq + n_family_ineq <= mi355_auglag_family_capacity(n) (four constraints per lane of the problem's
   * segment: 32 for n <= 16, 64 for n <= 32, 128 for n <= 64, 256 above); closed-menu terms only (no user functors),
   * Lbfgs inner solver with the More-Thuente line search, fused loop; term_constants must be NULL.  HOST memory, copied per
   * call; NULL / 0 = none. */
  int32_t n_family_eq, n_family_ineq;
  const double* family_eq;    /* [n_family_eq][n + 1] */
  const double* family_ineq;  /* [n_family_ineq][n + 1] */
} mi355_al_problem;
/* Largest n_family_eq + n_family_ineq a problem of dimension n may carry (0 when n is out of range). */
int32_t mi355_auglag_family_capacity(int32_t n);

/* AugmentedLagrangianConfig (augmented_lagrangian.h:64-196) and the stopping fields the constrained
 * Progress::Update reads (progress.h:112-126, :212-252). */
typedef struct mi355_al_config {
  double penalty_growth_factor;
  double violation_shrink_ratio;
  int32_t auto_scale_initial_penalty;
  double penalty_auto_objective_scale;
  double penalty_auto_min;
  double penalty_auto_max;
  int32_t warmup_max_inner_iterations;
  double warmup_inner_gradient_tolerance;
  double multiplier_max;
  uint64_t outer_num_iterations;      /* stopping_progress.num_iterations of the outer solver */
  double constraint_threshold;
  double kkt_stationarity_threshold;
  /* How the outer loop is run on the device (results are identical): MI355_AL_LOOP_AUTO, or one of
   * MI355_AL_LOOP_FUSED    the whole loop of a problem inside the persistent L-BFGS kernel — one launch per batch, no
   *                        host round trip, asynchronous; an outer step runs on the lanes of ONE problem, so it pays
   *                        when a wavefront holds few problems (auto: n > 16, and always with the Lbfgsb inner
   *                        solver, whose kernel holds four)
   * MI355_AL_LOOP_LOCKSTEP one launch of the inner solver + one of an outer-step kernel per outer iteration over the
   *                        problems still active, with a count read back every few iterations */
  int32_t loop;
} mi355_al_config;

enum mi355_al_loop { MI355_AL_LOOP_AUTO = 0, MI355_AL_LOOP_FUSED = 1, MI355_AL_LOOP_LOCKSTEP = 2 };

typedef struct mi355_al_progress {
  int32_t status;           /* mi355_solver_status of the outer loop */
  uint32_t num_iterations;  /* outer iterations */
  double x_delta, f_delta, gradient_norm; /* of the last outer step, on the composite (progress.h:188-196) */
  uint64_t inner_iterations; /* accounting: L-BFGS iterations, evaluations and stored (s, y) pairs used, summed */
  uint64_t nfev;             /* over the inner solves */
  uint64_t sum_k;
} mi355_al_progress;

/* The reference defaults: AugmentedLagrangianConfig{} and DefaultStoppingSolverProgress. */
int mi355_auglag_default_config(mi355_al_config* out);

/* One batched constrained solve.  x [B][n], lambda [B][n_eq], mu [B][n_ineq], penalty [B] are DEVICE arrays,
 * read as the initial AugmentedLagrangeState and overwritten with the returned one (the best iterate seen,
 * augmented_lagrangian.h Minimize); kkt [B] receives max_lagrangian_gradient; violation [B] is IN/OUT like the rest
 * of the state: the incoming max_violation (0 for a freshly constructed state) is the "previous violation" of the first
 * outer step's penalty-growth test (augmented_lagrangian.h:435), the returned one belongs to the returned state.
 * lambda / mu may be null when n_eq / n_ineq is 0; progress may be null.  term_constants is null, or a DEVICE array
 * [B][1 + n_eq + n_ineq] that gives every problem of the batch its own constants k (row b replaces problem->ks:
 * B different problems of one shape — e.g. per-problem right-hand sides — instead of B starts of one problem).
 * The inner solver is
 * Lbfgs<FunctionExpr, m, LineSearch> with `inner_stop` as its stopping_progress: m <= 10, linesearch a
 * mi355_linesearch (More-Thuente is the reference default).
 * With the fused loop (config->loop, the default above n = 16) the call is one asynchronous kernel launch on
 * `stream`; with the lock-step loop it returns after the last outer iteration (it reads a counter back every few). */
int mi355_auglag_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem, const mi355_al_config* config,
                                const mi355_lbfgs_stop* inner_stop, int32_t m, int32_t linesearch, int64_t B,
                                const double* term_constants, double* x, double* lambda, double* mu, double* penalty,
                                double* violation, double* kkt, mi355_al_progress* progress, void* stream);
/* Same with HOST arrays (staged through device memory). */
int mi355_auglag_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                     const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                     int32_t linesearch, int64_t B, const double* term_constants, double* x,
                                     double* lambda, double* mu, double* penalty, double* violation, double* kkt,
                                     mi355_al_progress* progress);
/* The same with Lbfgsb<FunctionExpr, m, LineSearch> as the inner solver (box constraints on x handled by the inner
 * solver, general constraints by the outer loop; src/test/augmented_lagrangian_test.cc:1017-1060, :1198-1275):
 * lower / upper are HOST arrays of n doubles (Lbfgsb::SetBounds, lbfgsb.h:89-93) or both NULL (the solver's default
 * box).  With bounds set, max_lagrangian_gradient is the projected norm Lbfgsb::ProjectedGradientInfNorm
 * (lbfgsb.h:105-118), as the reference's HasProjectedGradientInfNorm branch computes it.  m <= 5; n <= 64 with either
 * line search, 64 < n <= 128 (thirty-two lanes per problem; src/examples/svm_dual_al.cc has 100 variables) with
 * More-Thuente.  The loop runs fused inside the L-BFGS-B kernel unless config->loop asks for the lock-step form. */
int mi355_auglag_box_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                    const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                    int32_t linesearch, const double* lower, const double* upper, int64_t B,
                                    const double* term_constants, double* x, double* lambda, double* mu,
                                    double* penalty, double* violation, double* kkt, mi355_al_progress* progress,
                                    void* stream);
int mi355_auglag_box_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                         const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                         int32_t linesearch, const double* lower, const double* upper, int64_t B,
                                         const double* term_constants, double* x, double* lambda, double* mu,
                                         double* penalty, double* violation, double* kkt, mi355_al_progress* progress);
/* Value and gradient of ToAugmentedLagrangian(problem, (lambda, mu), penalty) at every row of x; HOST arrays. */
int mi355_auglag_eval_batch_host(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem, int64_t B,
                                 const double* term_constants, const double* x, const double* lambda, const double* mu,
                                 const double* penalty, double* f_out, double* g_out);

#ifdef __cplusplus
}
#endif
#endif /* MI355_LBFGS_H_ */
