// auglag.hip — augmented-Lagrangian entry points of the C-ABI (include/mi355_lbfgs.h) and their kernels.
//
// Host side of one batched solve (reference: AugmentedLagrangian::Minimize over Solver::Minimize,
// solver/augmented_lagrangian.h, solver/solver.h:181-224), in one of two forms (mi355_al_config.loop):
//
//   fused      pack (lambda, mu, penalty) -> per-problem rows
//              lbfgs_solve_kernel / lbfgsb_solve_kernel <AugLagObjective, ..., AugLagOuterLoop>: every problem's whole
//              outer loop, one launch
//              unpack                                                         (auglag_fused.hip; asynchronous)
//
//   lock-step  pack;  outer kernel, phase 0  (auto-scaled initial penalty)
//              repeat   inner:  lbfgs_solve_kernel / lbfgsb_solve_kernel <AugLagObjective> over the problems still
//                               active (a compacted index list)
//                       outer:  auglag_outer_kernel, phase 1  (multipliers, KKT norm, best iterate, penalty, status;
//                               appends the problems that continue to the next list)
//                       read back the number of problems still active (every iteration at first, then every fourth)
//              unpack
//
// Everything between pack and unpack lives in one grow-only device workspace owned by the context.
#define MI355_DISPATCH_TU
#include "auglag_launch.hpp"

namespace mi355 {
namespace {

// The library's own kernels (closed menu of term kinds); the fused halves are compiled in auglag_fused.hip.
const AlLaunchers kBuiltin = {&AlLaunchTable<BuiltinTermsFor>::inner,          &AlLaunchTable<BuiltinTermsFor>::inner_box,
                              &AlLaunchTable<BuiltinTermsFor>::composite_eval, &AlLaunchTable<BuiltinTermsFor>::outer,
                              &auglag_launch_fused,                            &auglag_launch_fused_box};

// The table a build with user term functors registered (at most one: the generated unit holds all of them).
struct UserAl {
  bool set = false;
  AlLaunchers launchers{};
  std::vector<int> ids;
  std::vector<int> blob_ids;  // of these, the functors that read mi355_al_problem::user_params
};
UserAl& user_al() {
  static UserAl u;
  return u;
}
bool is_user_term(int kind) {
  const UserAl& u = user_al();
  if (!u.set) return false;
  for (int id : u.ids)
    if (id == kind) return true;
  return false;
}

// (lambda, mu, penalty) <-> rows of `stride` doubles: the table's multipliers, rho, optionally the problem's own term
// constants, and at the END of the row the multipliers of the family constraints (lambda [B][n_eq + f_eq] and
// mu [B][n_ineq + f_ineq] hold the table's first, then the family's)
__global__ void pack_multipliers(const double* lambda, const double* mu, const double* penalty,
                                 const double* term_constants, double* mult, long long B, int n_eq, int n_ineq,
                                 int stride, int f_eq, int f_ineq) {
  const long long b = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int le = n_eq + f_eq, li = n_ineq + f_ineq, fam0 = stride - f_eq - f_ineq;
  for (int i = 0; i < n_eq; ++i) mult[b * stride + i] = lambda[b * le + i];
  for (int i = 0; i < n_ineq; ++i) mult[b * stride + n_eq + i] = mu[b * li + i];
  mult[b * stride + n_eq + n_ineq] = penalty[b];
  const int T = 1 + n_eq + n_ineq;
  if (term_constants)
    for (int t = 0; t < T; ++t) mult[b * stride + n_eq + n_ineq + 1 + t] = term_constants[b * T + t];
  for (int i = 0; i < f_eq; ++i) mult[b * stride + fam0 + i] = lambda[b * le + n_eq + i];
  for (int i = 0; i < f_ineq; ++i) mult[b * stride + fam0 + f_eq + i] = mu[b * li + n_ineq + i];
}
__global__ void unpack_multipliers(double* lambda, double* mu, double* penalty, const double* mult, long long B,
                                   int n_eq, int n_ineq, int stride, int f_eq, int f_ineq) {
  const long long b = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int le = n_eq + f_eq, li = n_ineq + f_ineq, fam0 = stride - f_eq - f_ineq;
  for (int i = 0; i < n_eq; ++i) lambda[b * le + i] = mult[b * stride + i];
  for (int i = 0; i < n_ineq; ++i) mu[b * li + i] = mult[b * stride + n_eq + i];
  penalty[b] = mult[b * stride + n_eq + n_ineq];
  for (int i = 0; i < f_eq; ++i) lambda[b * le + n_eq + i] = mult[b * stride + fam0 + i];
  for (int i = 0; i < f_ineq; ++i) mu[b * li + n_ineq + i] = mult[b * stride + fam0 + f_eq + i];
}

static_assert(sizeof(mi355_al_problem) == 96, "mi355_al_problem layout (capi.AlProblem mirrors it)");

int family_count(const mi355_al_problem* p) { return p->n_family_eq + p->n_family_ineq; }

int problem_rows(const mi355_al_problem* p) {
  const int T = 1 + p->n_eq + p->n_ineq;
  if (!p->parts) return T;
  int rows = 0;
  for (int t = 0; t < T; ++t) rows += (p->parts[t] == MI355_AL_PARTS_PRODUCT) ? 2 : p->parts[t];
  return rows;
}

int validate_problem(const mi355_al_problem* p) {
  if (!p || !p->kinds || !p->forms || !p->ks || !p->coef)
    return fail(MI355_ERR_INVALID_ARGUMENT, "null problem description");
  if (p->n < 1 || p->n > MI355_LBFGS_MAX_N) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range");
  if (p->n_eq < 0 || p->n_eq > MI355_AL_MAX_CONSTRAINTS || p->n_ineq < 0 || p->n_ineq > MI355_AL_MAX_CONSTRAINTS)
    return fail(MI355_ERR_INVALID_ARGUMENT, "at most MI355_AL_MAX_CONSTRAINTS equalities and inequalities");
  const int T = 1 + p->n_eq + p->n_ineq;
  for (int t = 0; t < T; ++t) {
    if (p->parts && p->parts[t] != MI355_AL_PARTS_PRODUCT && (p->parts[t] < 1 || p->parts[t] > MI355_AL_MAX_ROWS))
      return fail(MI355_ERR_INVALID_ARGUMENT, "a term is the sum of at least one primitive, or MI355_AL_PARTS_PRODUCT "
                                              "(the product of two)");
    if (p->forms[t] < MI355_AL_FORM_PLAIN || p->forms[t] > MI355_AL_FORM_K_MINUS_VALUE)
      return fail(MI355_ERR_UNSUPPORTED, "unknown term form");
  }
  const int rows = problem_rows(p);
  if (rows > MI355_AL_MAX_ROWS) return fail(MI355_ERR_INVALID_ARGUMENT, "more than MI355_AL_MAX_ROWS primitives");
  if (p->n_family_eq < 0 || p->n_family_ineq < 0 || (p->n_family_eq > 0 && !p->family_eq) ||
      (p->n_family_ineq > 0 && !p->family_ineq))
    return fail(MI355_ERR_INVALID_ARGUMENT, "constraint families: a negative count, or a count without its matrix");
  if (p->n_family_eq > MI355_AL_MAX_FAMILY || p->n_family_ineq > MI355_AL_MAX_FAMILY ||   // (each first: the sum may not wrap)
      family_count(p) > mi355_auglag_family_capacity(p->n))
    return fail(MI355_ERR_UNSUPPORTED, "more family constraints than mi355_auglag_family_capacity(n) (four per lane of "
                                       "the problem's segment, at most MI355_AL_MAX_FAMILY)");
  if (p->user_params_count < 0 || p->user_params_count > (1LL << 27) || (p->user_params_count > 0 && !p->user_params))
    return fail(MI355_ERR_INVALID_ARGUMENT, "user_params: a count without a pointer, or more than 2^27 doubles");
  for (int r = 0; r < rows; ++r) {
    const int kind = p->kinds[r];
    if (kind >= MI355_AL_TERM_USER) {
      if (!is_user_term(kind))
        return fail(MI355_ERR_UNSUPPORTED, "term kind >= MI355_AL_TERM_USER: no such term functor is compiled into this library");
      for (int id : user_al().blob_ids)
        if (id == kind && p->user_params_count <= 0)
          return fail(MI355_ERR_INVALID_ARGUMENT, "this user term functor takes its parameters from mi355_al_problem.user_params, which is empty");
    } else if (kind < MI355_AL_TERM_ROSENBROCK || kind > MI355_AL_TERM_SQUARED_AFFINE) {
      return fail(MI355_ERR_UNSUPPORTED, "unknown term kind");
    }
    if (kind >= MI355_AL_TERM_USER && family_count(p) > 0)
      return fail(MI355_ERR_UNSUPPORTED, "constraint families are built with the closed term menu (no user term functors)");
  }
  return MI355_OK;
}

// The kernels that evaluate this problem's terms: the library's own, or those built with its user term functors.
const AlLaunchers& launchers_of(const mi355_al_problem* p) {
  const int rows = problem_rows(p);
  for (int r = 0; r < rows; ++r)
    if (p->kinds[r] >= MI355_AL_TERM_USER) return user_al().launchers;
  return kBuiltin;
}

// Term table in the layout AugLagObjective<W, E>::fill_shared copies into LDS.
int upload_terms(mi355_lbfgs_ctx* ctx, const mi355_al_problem* p, const Mapping& mp, hipStream_t stream) {
  const int P = mp.W * mp.E, pitch = P + 1, T = 1 + p->n_eq + p->n_ineq, n = p->n, rows = problem_rows(p);
  std::vector<double>& h = ctx->params_host;
  // header, sixteen rows, then the problem's user_params (AugLagObjective::user: at shared_lds_doubles() of the mapping)
  const size_t table = static_cast<size_t>(kAlHeader) + static_cast<size_t>(kAlMaxRows) * pitch +
                       (static_cast<size_t>(kAlMaxRows) * pitch) % 2;
  const size_t user_count = p->user_params ? static_cast<size_t>(p->user_params_count) : 0;
  // ... then, for a problem with constraint families, the family block: k[FC], A[FC][P] row-major, A^T[P][FC]
  // (FC = al_family_capacity(W); rows past the problem's count and coordinates past n are zeros)
  const int fam = family_count(p), FC = al_family_capacity(mp.W);
  const size_t fam_off = (table + user_count + 1) / 2 * 2;
  const size_t fam_doubles = fam > 0 ? static_cast<size_t>(FC) * (1 + 2 * static_cast<size_t>(P)) : 0;
  h.assign(fam_off + fam_doubles + 1, 0.0);
  for (size_t i = 0; i < user_count; ++i) h[table + i] = p->user_params[i];
  h[0] = p->n_eq;
  h[1] = p->n_ineq;
  h[kAlFamBase] = p->n_family_eq;
  h[kAlFamBase + 1] = p->n_family_ineq;
  h[kAlFamBase + 2] = static_cast<double>(fam_off);
  if (fam > 0) {
    double* fk = h.data() + fam_off;
    double* fa = fk + FC;
    double* fat = fa + static_cast<size_t>(FC) * P;
    for (int i = 0; i < fam; ++i) {
      const double* src = (i < p->n_family_eq) ? p->family_eq + static_cast<size_t>(i) * (n + 1)
                                               : p->family_ineq + static_cast<size_t>(i - p->n_family_eq) * (n + 1);
      fk[i] = src[n];
      for (int j = 0; j < n; ++j) {
        fa[static_cast<size_t>(i) * P + j] = src[j];
        fat[static_cast<size_t>(j) * FC + i] = src[j];
      }
    }
  }
  int first = 0;
  for (int t = 0; t < T; ++t) {
    const int parts = p->parts ? p->parts[t] : 1;
    h[kAlTermBase + 4 * t] = first;
    h[kAlTermBase + 4 * t + 1] = parts;
    h[kAlTermBase + 4 * t + 2] = p->forms[t];
    h[kAlTermBase + 4 * t + 3] = p->ks[t];
    first += (parts == MI355_AL_PARTS_PRODUCT) ? 2 : parts;
  }
  for (int r = 0; r < rows; ++r) {
    h[kAlRowBase + r] = p->kinds[r];
    double* row = h.data() + kAlHeader + static_cast<size_t>(r) * pitch;
    if (p->kinds[r] >= MI355_AL_TERM_USER) {  // a user functor's parameters: the n + 1 coefficients as given
      for (int j = 0; j <= n; ++j) row[j] = p->coef[static_cast<size_t>(r) * (n + 1) + j];
      continue;
    }
    for (int j = 0; j < n; ++j) row[j] = p->coef[static_cast<size_t>(r) * (n + 1) + j];
    row[P] = p->coef[static_cast<size_t>(r) * (n + 1) + n];
  }
  ctx->params_resident.clear();  // (the blob of the objective entry points is overwritten)
  if (h.size() > ctx->params_cap) {
    if (ctx->params_dev) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(ctx->params_dev));
    }
    ctx->params_dev = nullptr;
    ctx->params_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->params_dev), h.size() * sizeof(double)));
    ctx->params_cap = h.size();
  }
  // (a context moved to another stream between calls: the new stream waits for the last launch that read the blob —
  //  as upload_params / upload_precond do, engine_internal.hpp wait_for_last_solve; round-5 advisor finding)
  HIP_TRY(mi355::wait_for_last_solve(ctx, ctx->params_stream, stream));
  ctx->params_stream = stream;
  HIP_TRY(hipMemcpyAsync(ctx->params_dev, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, stream));
  return MI355_OK;
}

// Carves the workspace; every array is 256-byte aligned.
struct Workspace {
  char* base = nullptr;
  size_t used = 0;
  template <class T>
  T* take(size_t count) {
    T* p = reinterpret_cast<T*>(base ? base + used : nullptr);
    used += (count * sizeof(T) + 255) / 256 * 256;
    return p;
  }
};

struct Arrays {
  double *x_inner, *mult, *inner_f, *best_x, *best_mult, *best_scalars;
  mi355_lbfgs_progress* inner_progress;
  mi355_al_progress* progress;
  unsigned char *active, *autoscaled;
  unsigned int* remaining;
  int* map[2];  // compacted index lists, ping-pong: an outer step reads the list its inner solve used, writes the next
  double* bounds;  // lower[n], upper[n] of an Lbfgsb inner solver
};

Arrays carve(Workspace& ws, long long B, int n, int stride) {
  Arrays a;
  const size_t b = static_cast<size_t>(B);
  a.x_inner = ws.take<double>(b * n);
  a.mult = ws.take<double>(b * stride);
  a.inner_f = ws.take<double>(b);
  a.best_x = ws.take<double>(b * n);
  a.best_mult = ws.take<double>(b * stride);
  a.best_scalars = ws.take<double>(b * 4);
  a.inner_progress = ws.take<mi355_lbfgs_progress>(b);
  a.progress = ws.take<mi355_al_progress>(b);
  a.active = ws.take<unsigned char>(b);
  a.autoscaled = ws.take<unsigned char>(b);
  a.remaining = ws.take<unsigned int>(64);  // ring of per-iteration counters (kRing)
  a.map[0] = ws.take<int>(b);
  a.map[1] = ws.take<int>(b);
  a.bounds = ws.take<double>(2 * MI355_LBFGS_MAX_N);
  return a;
}

int ensure_workspace(mi355_lbfgs_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->al_workspace_cap) return MI355_OK;
  if (ctx->al_workspace) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipFree(ctx->al_workspace));
  }
  ctx->al_workspace = nullptr;
  ctx->al_workspace_cap = 0;
  HIP_TRY(hipMalloc(&ctx->al_workspace, bytes));
  ctx->al_workspace_cap = bytes;
  return MI355_OK;
}

}  // namespace

void register_user_al_terms(const AlLaunchers& launchers, const int* ids, int count, const int* blob_ids, int blob_count) {
  UserAl& u = user_al();
  u.set = true;
  u.launchers = launchers;
  u.ids.assign(ids, ids + count);
  u.blob_ids.assign(blob_ids, blob_ids + blob_count);
}

int auglag_composite_minimize(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                              double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                              hipStream_t stream) {
  // desc was validated by the caller (mi355_lbfgs.hip): objective_params = n_eq, n_ineq, rows, then per term
  // (parts, form, k), then per row (kind, coefficient row [n + 1])
  if (desc->m > 10) return fail(MI355_ERR_UNSUPPORTED, "the composite objective is built for history sizes 1..10");
  if (desc->lanes_per_problem != 0 || desc->elems_per_lane != 0 || desc->hessian_diagonal != nullptr)
    return fail(MI355_ERR_INVALID_ARGUMENT, "composite objective: leave the mapping fields and hessian_diagonal unset");
  const int n = desc->n, n_eq = static_cast<int>(desc->objective_params[0]);
  const int n_ineq = static_cast<int>(desc->objective_params[1]), T = 1 + n_eq + n_ineq;
  const int rows = static_cast<int>(desc->objective_params[2]);
  std::vector<int32_t> kinds(rows), forms(T), parts(T);
  std::vector<double> ks(T), coef(static_cast<size_t>(rows) * (n + 1));
  const double* term_rec = desc->objective_params + 3;
  for (int t = 0; t < T; ++t) {
    parts[t] = static_cast<int32_t>(term_rec[3 * t]);
    forms[t] = static_cast<int32_t>(term_rec[3 * t + 1]);
    ks[t] = term_rec[3 * t + 2];
  }
  const double* row_rec = term_rec + 3 * T;
  for (int r = 0; r < rows; ++r) {
    const double* row = row_rec + static_cast<size_t>(r) * (n + 2);
    kinds[r] = static_cast<int32_t>(row[0]);
    for (int j = 0; j <= n; ++j) coef[static_cast<size_t>(r) * (n + 1) + j] = row[1 + j];
  }
  mi355_al_problem p;
  p.n = n;
  p.n_eq = n_eq;
  p.n_ineq = n_ineq;
  p.kinds = kinds.data();
  p.forms = forms.data();
  p.ks = ks.data();
  p.coef = coef.data();
  p.parts = parts.data();
  p.user_params = nullptr;  // (a composite objective names menu terms and coefficient-row functors only)
  p.user_params_count = 0;
  p.n_family_eq = p.n_family_ineq = 0;  // (nor constraint families)
  p.family_eq = p.family_ineq = nullptr;
  int rc = validate_problem(&p);
  if (rc != MI355_OK) return rc;
  if (problem_rows(&p) != rows) return fail(MI355_ERR_INVALID_ARGUMENT, "composite objective: rows != sum of parts");
  Mapping mp;
  if (!al_mapping(n, &mp)) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range");
  rc = upload_terms(ctx, &p, mp, stream);
  if (rc != MI355_OK) return rc;
  SolveArgs sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.x0 = x0;
  sa.x_out = x_out;
  sa.f_out = f_out;
  sa.g_out = g_out;
  sa.progress_out = progress_out;
  sa.obj_params = ctx->params_dev;
  sa.per_problem = desc->per_problem_data;
  sa.per_problem_stride = desc->per_problem_stride;
  sa.B = B;
  sa.n = n;
  sa.m = desc->m;
  sa.stop = desc->stop;
  return launchers_of(&p).inner(ctx, mp, desc->linesearch, sa, stream);
}

}  // namespace mi355

using namespace mi355;

extern "C" {

int mi355_auglag_default_config(mi355_al_config* out) {
  if (!out) return fail(MI355_ERR_INVALID_ARGUMENT, "null out pointer");
  out->penalty_growth_factor = 10.0;       // augmented_lagrangian.h, AugmentedLagrangianConfig
  out->violation_shrink_ratio = 0.25;
  out->auto_scale_initial_penalty = 1;
  out->penalty_auto_objective_scale = 10.0;
  out->penalty_auto_min = 1e-8;
  out->penalty_auto_max = 1e8;
  out->warmup_max_inner_iterations = 10;
  out->warmup_inner_gradient_tolerance = 1e-2;
  out->multiplier_max = 1e20;
  out->outer_num_iterations = 10000;       // DefaultStoppingSolverProgress, progress.h:353
  out->constraint_threshold = 1e-5;        // progress.h:378 / :416
  out->kkt_stationarity_threshold = 1e-4;  // progress.h:126
  out->loop = MI355_AL_LOOP_AUTO;
  return MI355_OK;
}

// box: the inner solver is Lbfgsb (lower / upper: HOST arrays of n doubles, or both null = its default box)
static int auglag_minimize_impl(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem, const mi355_al_config* config,
                                const mi355_lbfgs_stop* inner_stop, int32_t m, int32_t linesearch, bool box,
                                const double* lower, const double* upper, int64_t B, const double* term_constants,
                                double* x, double* lambda, double* mu, double* penalty, double* violation, double* kkt,
                                mi355_al_progress* progress, void* stream_) {
  if (!ctx) return fail(MI355_ERR_INVALID_ARGUMENT, "null context");
  int rc = validate_problem(problem);
  if (rc != MI355_OK) return rc;
  if (!config || !inner_stop) return fail(MI355_ERR_INVALID_ARGUMENT, "null config / inner_stop");
  if (B < 0 || B > 0x7fffffffLL) return fail(MI355_ERR_INVALID_ARGUMENT, "batch size out of range");
  if (B == 0) return MI355_OK;
  const int f_eq = problem->n_family_eq, f_ineq = problem->n_family_ineq, fam = f_eq + f_ineq;
  if (!x || !penalty || !violation || !kkt || (problem->n_eq + f_eq > 0 && !lambda) ||
      (problem->n_ineq + f_ineq > 0 && !mu))
    return fail(MI355_ERR_INVALID_ARGUMENT, "null state array");
  if (fam > 0) {   // the family kernels: Lbfgs + More-Thuente, fused loop (csrc/auglag_family.hip)
    if (box) return fail(MI355_ERR_UNSUPPORTED, "constraint families are built for the Lbfgs inner solver");
    if (linesearch != MI355_LS_MORE_THUENTE)
      return fail(MI355_ERR_UNSUPPORTED, "constraint families are built with the More-Thuente line search");
    if (term_constants) return fail(MI355_ERR_UNSUPPORTED, "constraint families: term_constants must be NULL");
    if (config && config->loop == MI355_AL_LOOP_LOCKSTEP)
      return fail(MI355_ERR_UNSUPPORTED, "constraint families run in the fused loop");
  }
  if (!box && (m < 1 || m > 10)) return fail(MI355_ERR_UNSUPPORTED, "the inner L-BFGS is built for history sizes 1..10");
  if (box && (m < 1 || m > 5)) return fail(MI355_ERR_UNSUPPORTED, "the inner L-BFGS-B is built for history sizes 1..5");
  if (box && problem->n > kAlBoxMaxN) return fail(MI355_ERR_UNSUPPORTED, "the inner L-BFGS-B is built for n <= 128");
  if (box && problem->n > 64 && linesearch == MI355_LS_HAGER_ZHANG)
    return fail(MI355_ERR_UNSUPPORTED, "the inner L-BFGS-B for 64 < n <= 128 is built with the More-Thuente line search");
  if ((lower == nullptr) != (upper == nullptr))
    return fail(MI355_ERR_INVALID_ARGUMENT, "lower and upper must both be given or both be NULL");
  for (int j = 0; lower && j < problem->n; ++j)  // (undefined breakpoint order in the reference, see mi355_lbfgs.hip)
    if (lower[j] != lower[j] || upper[j] != upper[j]) return fail(MI355_ERR_INVALID_ARGUMENT, "NaN bound");
  if (linesearch != MI355_LS_MORE_THUENTE && linesearch != MI355_LS_HAGER_ZHANG)
    return fail(MI355_ERR_UNSUPPORTED, "unknown line search id (More-Thuente = 0, Hager-Zhang = 1)");
  if (inner_stop->past > MI355_LBFGS_MAX_PAST) return fail(MI355_ERR_INVALID_ARGUMENT, "inner_stop.past too large");
  if (config->loop < MI355_AL_LOOP_AUTO || config->loop > MI355_AL_LOOP_LOCKSTEP)  // before anything is enqueued
    return fail(MI355_ERR_INVALID_ARGUMENT, "config.loop must be a mi355_al_loop");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MI355_ENTER_DEVICE(ctx);
  const AlLaunchers& launch = launchers_of(problem);
  Mapping mp;
  if (!(box ? al_box_mapping(problem->n, &mp) : al_mapping(problem->n, &mp)))
    return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range");
  const int n = problem->n, n_eq = problem->n_eq, n_ineq = problem->n_ineq;
  // per-problem rows: (lambda, mu, rho), followed by the problem's own term constants when the batch has them
  const int stride = n_eq + n_ineq + 1 + (term_constants ? 1 + n_eq + n_ineq : 0) + fam;
  rc = upload_terms(ctx, problem, mp, stream);
  if (rc != MI355_OK) return rc;
  Workspace sizing;
  carve(sizing, B, n, stride);
  rc = ensure_workspace(ctx, sizing.used);
  if (rc != MI355_OK) return rc;
  Workspace ws;
  ws.base = static_cast<char*>(ctx->al_workspace);
  const Arrays arr = carve(ws, B, n, stride);
  const size_t b = static_cast<size_t>(B);
  const unsigned grid = static_cast<unsigned>((B + 255) / 256);

  hipLaunchKernelGGL(pack_multipliers, dim3(grid), dim3(256), 0, stream, lambda, mu, penalty, term_constants, arr.mult,
                     B, n_eq, n_ineq, stride, f_eq, f_ineq);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemsetAsync(arr.active, 1, b, stream));
  HIP_TRY(hipMemsetAsync(arr.autoscaled, 0, b, stream));
  HIP_TRY(hipMemsetAsync(arr.best_scalars, 0, b * 4 * sizeof(double), stream));
  HIP_TRY(hipMemsetAsync(arr.progress, 0, b * sizeof(mi355_al_progress), stream));
  // `violation` is IN/OUT: the first outer step reads the incoming state's max_violation as the previous violation
  // of its penalty-growth test (augmented_lagrangian.h:435) — 0 for a freshly constructed state, the returned value
  // for a state fed back into Minimize.  max_lagrangian_gradient is only written.

  if (box) {  // SetBounds, or the default box lowest() .. max() (lbfgsb.h:124-129)
    std::vector<double>& h = ctx->bounds_host;
    h.assign(2 * static_cast<size_t>(n), 0.0);
    for (int j = 0; j < n; ++j) {
      h[j] = lower ? lower[j] : -1.7976931348623157e308;
      h[n + j] = upper ? upper[j] : 1.7976931348623157e308;
    }
    HIP_TRY(hipMemcpyAsync(arr.bounds, h.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice, stream));
  }

  AugLagOuterArgs oa;
  std::memset(&oa, 0, sizeof(oa));
  oa.x = x;
  oa.x_inner = arr.x_inner;
  oa.mult = arr.mult;
  oa.violation = violation;
  oa.kkt = kkt;
  oa.active = arr.active;
  oa.autoscaled = arr.autoscaled;
  oa.progress = arr.progress;
  oa.inner_progress = arr.inner_progress;
  oa.best_x = arr.best_x;
  oa.best_mult = arr.best_mult;
  oa.best_scalars = arr.best_scalars;
  oa.remaining = arr.remaining;
  oa.next_map = arr.map[0];
  oa.obj_params = ctx->params_dev;
  // the projected norm applies when bounds were set on the solver (bounds_initialized_, lbfgsb.h:107-109)
  oa.lower = (box && lower) ? arr.bounds : nullptr;
  oa.upper = (box && lower) ? arr.bounds + n : nullptr;
  oa.config = *config;
  oa.B = B;
  oa.n = n;
  oa.stride = stride;
  // auto: fused unless a wavefront holds eight problems (an outer step runs on the lanes of ONE problem while the
  // other segments wait; measured: 1.44x faster than lock-step at four problems per wavefront, 2.2x at two, 0.86x at
  // eight — profiles/r1_auglag.txt)
  const bool fused = fam > 0 || config->loop == MI355_AL_LOOP_FUSED || (config->loop == MI355_AL_LOOP_AUTO && mp.W >= 16);
  if (fused) {
    // the whole loop in one launch of the persistent inner-solver kernel (asynchronous on `stream`)
    SolveArgs fa;
    std::memset(&fa, 0, sizeof(fa));
    fa.x0 = x;                       // the state's x, also oa.x: a segment re-reads what it wrote
    fa.x_out = arr.x_inner;          // (not written in this mode)
    fa.f_out = arr.inner_f;
    fa.obj_params = ctx->params_dev;
    fa.per_problem = arr.mult;
    fa.per_problem_stride = stride;
    fa.B = B;
    fa.n = n;
    fa.m = m;
    fa.stop = *inner_stop;           // ConfigureInnerSubproblem: f_delta = 0; the warm-up is per problem, on the device
    fa.stop.f_delta = 0.0;
    oa.phase = 1;
    if (box) {
      LbfgsbArgs ba;
      ba.s = fa;
      ba.lower = arr.bounds;
      ba.upper = arr.bounds + n;
      rc = launch.fused_box(ctx, mp, linesearch, ba, oa, stream);
    } else if (fam > 0) {
      rc = auglag_launch_fused_family(ctx, mp, fa, oa, stream);
    } else {
      rc = launch.fused(ctx, mp, linesearch, fa, oa, stream);
    }
    if (rc != MI355_OK) return rc;
    hipLaunchKernelGGL(unpack_multipliers, dim3(grid), dim3(256), 0, stream, lambda, mu, penalty, arr.mult, B, n_eq,
                       n_ineq, stride, f_eq, f_ineq);
    HIP_TRY(hipGetLastError());
    if (progress)
      HIP_TRY(hipMemcpyAsync(progress, arr.progress, b * sizeof(mi355_al_progress), hipMemcpyDeviceToDevice, stream));
    return MI355_OK;
  }

  oa.phase = 0;  // lock-step loop: auto-scaled initial penalties first (the fused kernel does that when it fetches)
  rc = launch.outer(mp, oa, stream);
  if (rc != MI355_OK) return rc;
  oa.phase = 1;

  SolveArgs sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.x0 = x;
  sa.x_out = arr.x_inner;
  sa.f_out = arr.inner_f;
  sa.progress_out = arr.inner_progress;
  sa.obj_params = ctx->params_dev;
  sa.per_problem = arr.mult;
  sa.per_problem_stride = stride;
  sa.B = B;  // first outer iteration: every problem, in place; later: the compacted list of the active ones
  sa.n = n;
  sa.m = m;
  const bool has_general_constraints = n_eq + n_ineq > 0;
  // Every problem is on the same outer iteration, and one that stops never restarts: the loop ends when an outer
  // step reports nobody left (each problem's own num_iterations test bounds it).  The outer step of iteration k
  // counts the problems that continue into counter k (a ring, zeroed a round at a time) and lists them; iteration
  // k + 1 reads both on the device, so a chain of iterations needs no host round trip.  The first iterations, where
  // most problems stop, are read back one at a time (the count sizes the next grid); after that four per read-back,
  // launched with the last known count as the grid bound — iterations past the end of the work are empty launches.
  constexpr int kRing = 64;
  constexpr unsigned int kFusedTail = 256;  // problems; hand-over threshold to the fused kernel
  unsigned int remaining = static_cast<unsigned int>(B);
  for (uint64_t outer = 1; remaining != 0;) {
    const int chain = (outer <= 4) ? 1 : 4;
    for (int c = 0; c < chain; ++c, ++outer) {
      const int slot = static_cast<int>(outer % kRing);
      // zeroed half a ring at a time, so that the counter the previous iteration wrote (the other half at the
      // boundary) survives until this iteration's kernels have read it
      if (outer == 1) {
        HIP_TRY(hipMemsetAsync(arr.remaining, 0, kRing * sizeof(unsigned int), stream));
      } else if (slot % (kRing / 2) == 0) {
        HIP_TRY(hipMemsetAsync(arr.remaining + slot, 0, (kRing / 2) * sizeof(unsigned int), stream));
      }
      sa.stop = *inner_stop;            // ConfigureInnerSubproblem
      sa.stop.f_delta = 0.0;
      if (outer == 1 && has_general_constraints && config->warmup_max_inner_iterations > 0) {
        sa.stop.num_iterations = static_cast<uint64_t>(config->warmup_max_inner_iterations);
        sa.stop.gradient_norm = config->warmup_inner_gradient_tolerance;
      }
      if (box) {
        LbfgsbArgs ba;
        ba.s = sa;
        ba.lower = arr.bounds;
        ba.upper = arr.bounds + n;
        rc = launch.inner_box(ctx, mp, linesearch, ba, stream);
      } else {
        rc = launch.inner(ctx, mp, linesearch, sa, stream);
      }
      if (rc != MI355_OK) return rc;
      oa.remaining = arr.remaining + slot;
      rc = launch.outer(mp, oa, stream);
      if (rc != MI355_OK) return rc;
      // the next iteration works on the list this one wrote
      sa.problem_map = oa.next_map;
      sa.count_dev = oa.remaining;
      oa.cur_map = oa.next_map;
      oa.count_dev = oa.remaining;
      oa.next_map = (oa.next_map == arr.map[0]) ? arr.map[1] : arr.map[0];
    }
    HIP_TRY(hipMemcpyAsync(&remaining, oa.count_dev, sizeof(remaining), hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    sa.B = remaining;  // grid bound of the next chain
    oa.B = remaining;
    // Once only stragglers are left (a few wavefronts' worth), their remaining outer iterations run in the fused
    // kernel: no launches or read-backs per iteration, and with so few problems an outer step no longer makes the
    // other segments of a wavefront wait.
    if (!box && remaining != 0 && remaining <= kFusedTail && config->loop == MI355_AL_LOOP_AUTO) {
      SolveArgs fa = sa;
      fa.stop = *inner_stop;
      fa.stop.f_delta = 0.0;
      fa.x_out = arr.x_inner;
      fa.progress_out = nullptr;
      rc = launch.fused(ctx, mp, linesearch, fa, oa, stream);
      if (rc != MI355_OK) return rc;
      break;
    }
    if (config->outer_num_iterations == 0 && outer >= 1000000)
      return fail(MI355_ERR_INVALID_ARGUMENT, "outer loop without an iteration limit did not stop");
  }
  hipLaunchKernelGGL(unpack_multipliers, dim3(grid), dim3(256), 0, stream, lambda, mu, penalty, arr.mult, B, n_eq,
                     n_ineq, stride, 0, 0);
  HIP_TRY(hipGetLastError());
  if (progress)
    HIP_TRY(hipMemcpyAsync(progress, arr.progress, b * sizeof(mi355_al_progress), hipMemcpyDeviceToDevice, stream));
  return MI355_OK;
}

int32_t mi355_auglag_family_capacity(int32_t n) {
  Mapping mp;
  if (n < 1 || n > MI355_LBFGS_MAX_N || !al_mapping(n, &mp)) return 0;
  return al_family_capacity(mp.W);
}

int mi355_auglag_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem, const mi355_al_config* config,
                                const mi355_lbfgs_stop* inner_stop, int32_t m, int32_t linesearch, int64_t B,
                                const double* term_constants, double* x, double* lambda, double* mu, double* penalty,
                                double* violation, double* kkt, mi355_al_progress* progress, void* stream) {
  return auglag_minimize_impl(ctx, problem, config, inner_stop, m, linesearch, false, nullptr, nullptr, B, term_constants,
                              x, lambda, mu, penalty, violation, kkt, progress, stream);
}

int mi355_auglag_box_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                    const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                    int32_t linesearch, const double* lower, const double* upper, int64_t B,
                                    const double* term_constants, double* x, double* lambda, double* mu,
                                    double* penalty, double* violation, double* kkt, mi355_al_progress* progress,
                                    void* stream) {
  return auglag_minimize_impl(ctx, problem, config, inner_stop, m, linesearch, true, lower, upper, B, term_constants, x,
                              lambda, mu, penalty, violation, kkt, progress, stream);
}

static int auglag_minimize_host_impl(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                     const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                     int32_t linesearch, bool box, const double* lower, const double* upper, int64_t B,
                                     const double* term_constants, double* x, double* lambda, double* mu,
                                     double* penalty, double* violation, double* kkt, mi355_al_progress* progress);

int mi355_auglag_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                     const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                     int32_t linesearch, int64_t B, const double* term_constants, double* x,
                                     double* lambda, double* mu, double* penalty, double* violation, double* kkt,
                                     mi355_al_progress* progress) {
  return auglag_minimize_host_impl(ctx, problem, config, inner_stop, m, linesearch, false, nullptr, nullptr, B,
                                   term_constants, x, lambda, mu, penalty, violation, kkt, progress);
}

int mi355_auglag_box_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                         const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                         int32_t linesearch, const double* lower, const double* upper, int64_t B,
                                         const double* term_constants, double* x, double* lambda, double* mu,
                                         double* penalty, double* violation, double* kkt, mi355_al_progress* progress) {
  return auglag_minimize_host_impl(ctx, problem, config, inner_stop, m, linesearch, true, lower, upper, B,
                                   term_constants, x, lambda, mu, penalty, violation, kkt, progress);
}

static int auglag_minimize_host_impl(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem,
                                     const mi355_al_config* config, const mi355_lbfgs_stop* inner_stop, int32_t m,
                                     int32_t linesearch, bool box, const double* lower, const double* upper, int64_t B,
                                     const double* term_constants, double* x, double* lambda, double* mu,
                                     double* penalty, double* violation, double* kkt, mi355_al_progress* progress) {
  if (!ctx) return fail(MI355_ERR_INVALID_ARGUMENT, "null context");
  int rc = validate_problem(problem);
  if (rc != MI355_OK) return rc;
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  if (B == 0) return MI355_OK;
  if (!x || !penalty || !violation || !kkt) return fail(MI355_ERR_INVALID_ARGUMENT, "null state array");
  MI355_ENTER_DEVICE(ctx);
  const size_t b = static_cast<size_t>(B), n = static_cast<size_t>(problem->n);
  // (lambda and mu hold the table's multipliers, then the families')
  const size_t ne = static_cast<size_t>(problem->n_eq + problem->n_family_eq);
  const size_t ni = static_cast<size_t>(problem->n_ineq + problem->n_family_ineq);
  const size_t nk = term_constants ? 1 + static_cast<size_t>(problem->n_eq) + static_cast<size_t>(problem->n_ineq) : 0;
  const size_t doubles = b * (n + ne + ni + 3 + nk);
  char* dev = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), doubles * sizeof(double) + b * sizeof(mi355_al_progress)));
  double* dx = reinterpret_cast<double*>(dev);
  double* dl = dx + b * n;
  double* dm = dl + b * ne;
  double* dp = dm + b * ni;
  double* dv = dp + b;
  double* dk = dv + b;
  double* dtc = dk + b;
  mi355_al_progress* dprog = reinterpret_cast<mi355_al_progress*>(dtc + b * nk);
  auto cleanup = [&](int code) {
    (void)hipFree(dev);
    return code;
  };
  auto up = [&](double* d, const double* h, size_t count) {
    return count == 0 ? hipSuccess : hipMemcpy(d, h, count * sizeof(double), hipMemcpyHostToDevice);
  };
  auto down = [&](double* h, const double* d, size_t count) {
    return count == 0 ? hipSuccess : hipMemcpy(h, d, count * sizeof(double), hipMemcpyDeviceToHost);
  };
  if ((ne > 0 && !lambda) || (ni > 0 && !mu)) return cleanup(fail(MI355_ERR_INVALID_ARGUMENT, "null multiplier array"));
  if (up(dx, x, b * n) != hipSuccess || up(dl, lambda, b * ne) != hipSuccess || up(dm, mu, b * ni) != hipSuccess ||
      up(dp, penalty, b) != hipSuccess || up(dv, violation, b) != hipSuccess ||
      up(dtc, term_constants, b * nk) != hipSuccess)
    return cleanup(fail(MI355_ERR_HIP, "host to device copy failed"));
  rc = auglag_minimize_impl(ctx, problem, config, inner_stop, m, linesearch, box, lower, upper, B, nk ? dtc : nullptr, dx,
                            dl, dm, dp, dv, dk, dprog, nullptr);
  if (rc != MI355_OK) return cleanup(rc);
  if (hipDeviceSynchronize() != hipSuccess) return cleanup(fail(MI355_ERR_HIP, "augmented-Lagrangian kernels failed"));
  if (down(x, dx, b * n) != hipSuccess || down(lambda, dl, b * ne) != hipSuccess || down(mu, dm, b * ni) != hipSuccess ||
      down(penalty, dp, b) != hipSuccess || down(violation, dv, b) != hipSuccess || down(kkt, dk, b) != hipSuccess)
    return cleanup(fail(MI355_ERR_HIP, "device to host copy failed"));
  if (progress && hipMemcpy(progress, dprog, b * sizeof(mi355_al_progress), hipMemcpyDeviceToHost) != hipSuccess)
    return cleanup(fail(MI355_ERR_HIP, "device to host copy failed"));
  return cleanup(MI355_OK);
}

int mi355_auglag_eval_batch_host(mi355_lbfgs_ctx* ctx, const mi355_al_problem* problem, int64_t B,
                                 const double* term_constants, const double* x, const double* lambda, const double* mu,
                                 const double* penalty, double* f_out, double* g_out) {
  if (!ctx) return fail(MI355_ERR_INVALID_ARGUMENT, "null context");
  int rc = validate_problem(problem);
  if (rc != MI355_OK) return rc;
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  if (B == 0) return MI355_OK;
  if (!x || !penalty || !f_out || !g_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null array");
  MI355_ENTER_DEVICE(ctx);
  Mapping mp;
  if (!al_mapping(problem->n, &mp)) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range");
  const int n = problem->n, n_eq = problem->n_eq, n_ineq = problem->n_ineq, T = 1 + n_eq + n_ineq;
  const int f_eq = problem->n_family_eq, f_ineq = problem->n_family_ineq, fam = f_eq + f_ineq;
  if (fam > 0 && term_constants) return fail(MI355_ERR_UNSUPPORTED, "constraint families: term_constants must be NULL");
  const int stride = n_eq + n_ineq + 1 + (term_constants ? T : 0) + fam;
  if ((n_eq + f_eq > 0 && !lambda) || (n_ineq + f_ineq > 0 && !mu))
    return fail(MI355_ERR_INVALID_ARGUMENT, "null multiplier array");
  rc = upload_terms(ctx, problem, mp, nullptr);
  if (rc != MI355_OK) return rc;
  const size_t b = static_cast<size_t>(B);
  std::vector<double> rows(b * stride);
  for (size_t i = 0; i < b; ++i) {
    const size_t le = static_cast<size_t>(n_eq + f_eq), li = static_cast<size_t>(n_ineq + f_ineq);
    for (int c = 0; c < n_eq; ++c) rows[i * stride + c] = lambda[i * le + c];
    for (int c = 0; c < n_ineq; ++c) rows[i * stride + n_eq + c] = mu[i * li + c];
    rows[i * stride + n_eq + n_ineq] = penalty[i];
    if (term_constants)
      for (int t = 0; t < T; ++t) rows[i * stride + n_eq + n_ineq + 1 + t] = term_constants[i * T + t];
    for (int c = 0; c < f_eq; ++c) rows[i * stride + (stride - fam) + c] = lambda[i * le + n_eq + c];
    for (int c = 0; c < f_ineq; ++c) rows[i * stride + (stride - fam) + f_eq + c] = mu[i * li + n_ineq + c];
  }
  double* dev = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev), (b * (2 * n + 1) + rows.size()) * sizeof(double)));
  double* dx = dev;
  double* dg = dx + b * n;
  double* df = dg + b * n;
  double* dm = df + b;
  auto cleanup = [&](int code) {
    (void)hipFree(dev);
    return code;
  };
  if (hipMemcpy(dx, x, b * n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(dm, rows.data(), rows.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(MI355_ERR_HIP, "host to device copy failed"));
  SolveArgs sa;
  std::memset(&sa, 0, sizeof(sa));
  sa.x0 = dx;
  sa.f_out = df;
  sa.g_out = dg;
  sa.obj_params = ctx->params_dev;
  sa.per_problem = dm;
  sa.per_problem_stride = stride;
  sa.B = B;
  sa.n = n;
  sa.m = 1;
  rc = (fam > 0) ? auglag_launch_family_eval(mp, sa, nullptr) : launchers_of(problem).composite_eval(mp, sa, nullptr);
  if (rc != MI355_OK) return cleanup(rc);
  if (hipDeviceSynchronize() != hipSuccess) return cleanup(fail(MI355_ERR_HIP, "evaluation kernel failed"));
  if (hipMemcpy(f_out, df, b * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(g_out, dg, b * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
    return cleanup(fail(MI355_ERR_HIP, "device to host copy failed"));
  return cleanup(MI355_OK);
}

}  // extern "C"
