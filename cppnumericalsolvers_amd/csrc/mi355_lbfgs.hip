// mi355_lbfgs.hip — implementation of the C-ABI in include/mi355_lbfgs.h:
// argument validation, (W, E) mapping choice, kernel dispatch, device scratch.
// gfx950 only; no CPU fallback anywhere in this file.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "engine_internal.hpp"
#include "lbfgs_wide_kernel.hpp"
#include "ridge_mfma_kernel.hpp"  // layout constants of the joint-evaluation ridge kernel (not instantiated here)

// ABI layout guards (mirrored by cppnumericalsolvers_amd/capi.py and the C++ host header).
static_assert(sizeof(mi355_lbfgs_stop) == 64, "mi355_lbfgs_stop layout");
static_assert(sizeof(mi355_lbfgs_progress) == 40, "mi355_lbfgs_progress layout");

namespace {
thread_local std::string g_last_error;
}  // namespace

namespace mi355 {
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

// ---- user objectives: registered by the static initialisers of generated translation units -------------
namespace {
struct UserEntry {
  UserDispatchFn fn[4] = {nullptr, nullptr, nullptr, nullptr};  // lanes per problem 8, 16, 32, 64
  UserLbfgsbFn lbfgsb = nullptr;
  std::string name;
};
std::vector<std::pair<int, UserEntry>>& user_table() {
  static std::vector<std::pair<int, UserEntry>> table;  // (function-local: initialisation order of the units is free)
  return table;
}
int w_slot(int W) { return W == 8 ? 0 : W == 16 ? 1 : W == 32 ? 2 : W == 64 ? 3 : -1; }
}  // namespace
void register_user_objective(int objective_id, int W, UserDispatchFn fn, const char* name) {
  if (objective_id < MI355_OBJ_USER_FIRST || w_slot(W) < 0) return;
  for (auto& e : user_table()) {
    if (e.first == objective_id) {
      e.second.fn[w_slot(W)] = fn;
      return;
    }
  }
  UserEntry u;
  u.fn[w_slot(W)] = fn;
  u.name = name ? name : "";
  user_table().emplace_back(objective_id, u);
}
void register_user_lbfgsb(int objective_id, UserLbfgsbFn fn) {
  if (objective_id < MI355_OBJ_USER_FIRST) return;
  for (auto& e : user_table()) {
    if (e.first == objective_id) {
      e.second.lbfgsb = fn;
      return;
    }
  }
  UserEntry u;
  u.lbfgsb = fn;
  user_table().emplace_back(objective_id, u);
}
static const UserEntry* find_user_objective(int objective_id) {
  for (auto& e : user_table())
    if (e.first == objective_id) return &e.second;
  return nullptr;
}
}  // namespace mi355

namespace {

using namespace mi355;

// History sizes served by a register-history kernel: the variants are built for 5, 6 and 10 columns and
// take any m up to their size (the arrays are chronological, newest last, so a shorter history just
// leaves the oldest columns unused).
bool has_register_history_variant(int m) { return m >= 1 && m <= 10; }

int choose_mapping(int objective, int n, int m, bool allow_register_history, int& W, int& E) {
  // Default mapping, from the measured sweeps (profiles/r1_mapping_sweep.txt).  Pad n to the
  // power of two P >= 8 and pack several problems into a wavefront so that every butterfly
  // instruction serves all of them: four coordinates per lane (W = P/4) when the y half of the
  // history can live in registers, otherwise two (W = P/2) — with both ring halves in LDS the
  // wider packing leaves too few wavefronts per SIMD (LDS capacity / ring size).
  int P = 8;
  while (P < n) P <<= 1;
  const bool reg = allow_register_history && has_register_history_variant(m);
  if (P <= 8) {
    W = 8;
    E = 1;
  } else if (P == 16) {
    W = 8;
    E = 2;
  } else if (P <= 128) {
    E = reg ? 4 : 2;
    // the ridge objective keeps 128/W residual rows per lane in registers: stay at two
    // coordinates per lane so its matrix-vector loops are not starved of registers
    if (objective == MI355_OBJ_SQUARED_ERROR_RIDGE) E = 2;
    W = P / E;
  } else {
    W = 64;
    E = 4;
  }
  return 0;
}

// Default mapping of the dense-BFGS kernels for the padded width P (8, 16, 32, 64): one coordinate — one column of H —
// per lane at P = 32 and 64 (rounds 3-5 packed 4 per lane: eight / four problems and 70 / 130 KB of LDS per wavefront,
// half / a quarter of a wavefront per SIMD); an explicit lanes_per_problem x elems_per_lane selects any built split.
// User objectives keep the packed splits (8 lanes, 16 at P = 64): those are the ones their generated units hold.
void bfgs_default_mapping(int P, bool packed, int& W, int& E) {
  W = packed ? ((P == 64) ? 16 : 8) : ((P <= 16) ? 8 : P);
  E = P / W;
}

// Default mapping of `Lbfgs<F, m, HagerZhang>` in the fused arithmetic: two coordinates per lane (the register-history
// kernels of that line search are built — and spill-free — for E = 2: 1.07x / 1.72x the exact LDS-ring kernel at
// n = 32 / 64, profiles/r6_ab_hz.txt), one below nine coordinates, four only where 64 lanes x 2 do not cover n.
void hz_fused_mapping(int n, int& W, int& E) {
  int P = 8;
  while (P < n) P <<= 1;
  if (P <= 8) {
    W = 8;
    E = 1;
  } else if (P <= 128) {
    W = P / 2;
    E = 2;
  } else {
    W = 64;
    E = 4;
  }
}

bool valid_mapping(int n, int W, int E) {
  if (E == 8) return (W == 4 || W == 8) && n <= W * E;  // the wide-lane kernels (engine_internal.hpp, launch_solve_e8)
  const bool wok = (W == 8 || W == 16 || W == 32 || W == 64);
  const bool eok = (E == 1 || E == 2 || E == 4);
  return wok && eok && n <= W * E;
}

int dispatch(mi355_lbfgs_ctx* ctx, int W, int E, int objective, int mr, const SolveArgs& args,
             hipStream_t stream, bool eval_only) {
  if (objective >= MI355_OBJ_USER_FIRST) {
    const UserEntry* u = find_user_objective(objective);
    if (!u) return fail(MI355_ERR_UNSUPPORTED, "no user objective with this id is compiled into this library");
    const int slot = w_slot(W);
    if (slot < 0 || !u->fn[slot]) return fail(MI355_ERR_UNSUPPORTED, "user objective: this lanes_per_problem is not built");
    return u->fn[slot](ctx, E, mr, args, stream, eval_only);
  }
  switch (W) {
    case 4: return dispatch_w4(ctx, E, objective, mr, args, stream, eval_only);
    case 8: return dispatch_w8(ctx, E, objective, mr, args, stream, eval_only);
    case 16: return dispatch_w16(ctx, E, objective, mr, args, stream, eval_only);
    case 32: return dispatch_w32(ctx, E, objective, mr, args, stream, eval_only);
    case 64: return dispatch_w64(ctx, E, objective, mr, args, stream, eval_only);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "lanes_per_problem must be 8, 16, 32 or 64");
}

int n_params_expected(const mi355_lbfgs_desc* desc) {
  if (desc->objective >= MI355_OBJ_USER_FIRST)  // the blob of a user objective is opaque to the library
    return (find_user_objective(desc->objective) && desc->n_params >= 0) ? desc->n_params : -1;
  switch (desc->objective) {
    case MI355_OBJ_ROSENBROCK: return 0;
    case MI355_OBJ_DIAG_QUADRATIC: return desc->n + 1;
    case MI355_OBJ_SQUARED_ERROR_RIDGE:
    case MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA:
    case MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM: {
      if (!desc->objective_params || desc->n_params < 2) return -2;
      const double rows = desc->objective_params[0];
      const int max_rows = desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM ? MI355_LBFGS_GRAM_MAX_ROWS : MI355_LBFGS_MAX_ROWS;
      if (!(rows >= 1 && rows <= max_rows) || rows != static_cast<int>(rows)) return -2;
      return 2 + static_cast<int>(rows) * desc->n;
    }
    case MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM: {   // rows, lambda; the matrices travel with the per-problem rows
      if (!desc->objective_params || desc->n_params < 2) return -2;
      const double rows = desc->objective_params[0];
      if (!(rows >= 1 && rows <= MI355_LBFGS_GRAM_MAX_ROWS) || rows != static_cast<int>(rows)) return -2;
      return 2;
    }
    case MI355_OBJ_AL_COMPOSITE: {
      if (!desc->objective_params || desc->n_params < 3) return -3;
      const double ne = desc->objective_params[0], ni = desc->objective_params[1], rows = desc->objective_params[2];
      if (!(ne >= 0 && ne <= MI355_AL_MAX_CONSTRAINTS && ni >= 0 && ni <= MI355_AL_MAX_CONSTRAINTS) ||
          ne != static_cast<int>(ne) || ni != static_cast<int>(ni) ||
          !(rows >= 1 + ne + ni && rows <= MI355_AL_MAX_ROWS) || rows != static_cast<int>(rows))
        return -3;
      return 3 + 3 * (1 + static_cast<int>(ne) + static_cast<int>(ni)) + static_cast<int>(rows) * (desc->n + 2);
    }
  }
  return -1;
}

int validate(const mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, long long B, bool allow_wide = false) {
  if (!ctx) return fail(MI355_ERR_INVALID_ARGUMENT, "null context");
  if (!desc) return fail(MI355_ERR_INVALID_ARGUMENT, "null desc");
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  // (dimensions above MI355_LBFGS_MAX_N exist for the Lbfgs solve entry points only: minimize_batch_impl)
  if (desc->n < 1 || desc->n > (allow_wide ? MI355_LBFGS_WIDE_MAX_N : MI355_LBFGS_MAX_N))
    return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range [1, MI355_LBFGS_MAX_N]");
  if (desc->m < 1 || desc->m > MI355_LBFGS_MAX_M)
    return fail(MI355_ERR_INVALID_ARGUMENT, "m out of range [1, MI355_LBFGS_MAX_M]");
  if (desc->linesearch != MI355_LS_MORE_THUENTE && desc->linesearch != MI355_LS_HAGER_ZHANG)
    return fail(MI355_ERR_UNSUPPORTED, "unknown line search id (More-Thuente = 0, Hager-Zhang = 1)");
  const int np = n_params_expected(desc);
  if (np == -2) return fail(MI355_ERR_INVALID_ARGUMENT, "ridge objective: params must start with rows in [1, 128]");
  if (np == -3)
    return fail(MI355_ERR_INVALID_ARGUMENT, "composite objective: params must start with n_eq, n_ineq in [0, MI355_AL_MAX_CONSTRAINTS] and the row count");
  if (np < 0) return fail(MI355_ERR_UNSUPPORTED, "unknown objective id");
  if (desc->objective == MI355_OBJ_AL_COMPOSITE) {
    const int stride = static_cast<int>(desc->objective_params[0]) + static_cast<int>(desc->objective_params[1]) + 1;
    if (!desc->per_problem_data || (desc->per_problem_stride != stride && desc->per_problem_stride != 2 * stride))
      return fail(MI355_ERR_INVALID_ARGUMENT,
                  "composite objective: per_problem_data holds rows (lambda, mu, penalty) of n_eq + n_ineq + 1 doubles, "
                  "optionally followed by one constant per term (1 + n_eq + n_ineq more)");
  }
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE || desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA ||
      desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM) {
    if (!desc->per_problem_data) return fail(MI355_ERR_INVALID_ARGUMENT, "ridge objective: per_problem_data (y) is null");
    if (desc->per_problem_stride < static_cast<int>(desc->objective_params[0]))
      return fail(MI355_ERR_INVALID_ARGUMENT, "ridge objective: per_problem_stride < rows");
  }
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM) {
    if (!desc->per_problem_data)
      return fail(MI355_ERR_INVALID_ARGUMENT, "own-matrix ridge objective: per_problem_data (A_b, y_b) is null");
    const long long rows = static_cast<long long>(desc->objective_params[0]);
    if (desc->per_problem_stride < rows * desc->n + rows)
      return fail(MI355_ERR_INVALID_ARGUMENT, "own-matrix ridge objective: per_problem_stride < rows * n + rows");
  }
  if (desc->n_params != np) return fail(MI355_ERR_INVALID_ARGUMENT, "n_params does not match objective");
  if (np > 0 && !desc->objective_params)
    return fail(MI355_ERR_INVALID_ARGUMENT, "objective_params is null");
  if (desc->arithmetic < MI355_ARITH_DEFAULT || desc->arithmetic > MI355_ARITH_FMA)
    return fail(MI355_ERR_INVALID_ARGUMENT, "arithmetic must be a mi355_arithmetic");
  if (desc->hessian_from_functor != 0 && desc->hessian_from_functor != 1)
    return fail(MI355_ERR_INVALID_ARGUMENT, "hessian_from_functor must be 0 or 1");
  if (desc->hessian_from_functor && desc->hessian_diagonal != nullptr)
    return fail(MI355_ERR_INVALID_ARGUMENT, "hessian_from_functor: the diagonal comes from the functor (hessian_diagonal NULL)");
  if (desc->hessian_from_functor && desc->hessian_condition_stop != 0.0 &&
      (!(desc->hessian_condition_stop > 0.0) || desc->n > 64))
    return fail(MI355_ERR_UNSUPPORTED,
                "hessian_from_functor: the condition_hessian test (hessian_condition_stop > 0) runs on the device for "
                "Lbfgs at n <= 64");
  if (desc->history_placement < 0 || desc->history_placement > 2)
    return fail(MI355_ERR_INVALID_ARGUMENT, "history_placement must be 0 (auto), 1 (LDS) or 2 (y in registers)");
  if (desc->stop.past < 0 || desc->stop.past > MI355_LBFGS_MAX_PAST)
    return fail(MI355_ERR_INVALID_ARGUMENT, "stop.past out of range [0, MI355_LBFGS_MAX_PAST]");
  if (desc->stop.x_delta_violations < 0 || desc->stop.f_delta_violations < 0)
    return fail(MI355_ERR_INVALID_ARGUMENT, "negative violation count");
  if (desc->per_problem_data != nullptr && desc->per_problem_stride < 0)
    return fail(MI355_ERR_INVALID_ARGUMENT, "negative per_problem_stride");
  if (desc->hessian_condition_stop != 0.0 && desc->hessian_diagonal == nullptr && !desc->hessian_from_functor)
    return fail(MI355_ERR_INVALID_ARGUMENT,
                "hessian_condition_stop is the condition_hessian test of Second-mode functions: it needs hessian_diagonal");
  if (desc->hessian_condition_stop < 0.0 || desc->hessian_condition_stop != desc->hessian_condition_stop)
    return fail(MI355_ERR_INVALID_ARGUMENT, "hessian_condition_stop must be >= 0");
  return MI355_OK;
}

int upload_params(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int W, int E, hipStream_t stream) {
  const double* src = desc->objective_params;
  size_t np = static_cast<size_t>(desc->n_params);
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA) {
    // device layout: rows, lambda, the LDS image of A (A[i][j] at i * kJointPitchA + j, zero padded to 128 x 65)
    const int rows = static_cast<int>(desc->objective_params[0]);
    const int n = desc->n;
    std::vector<double>& h = ctx->params_host;
    h.assign(2 + static_cast<size_t>(kJointRows) * kJointPitchA, 0.0);
    h[0] = rows;
    h[1] = desc->objective_params[1];
    const double* a = desc->objective_params + 2;
    for (int i = 0; i < rows; ++i)
      for (int j = 0; j < n; ++j) h[2 + static_cast<size_t>(i) * kJointPitchA + j] = a[static_cast<size_t>(i) * n + j];
    src = h.data();
    np = h.size();
  } else if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE) {
    // device layout: rows, lambda, AT[P][129] (zero padded, pitch kRidgePitch), P = W*E
    const int rows = static_cast<int>(desc->objective_params[0]);
    const int n = desc->n, P = W * E;
    std::vector<double>& h = ctx->params_host;
    h.assign(2 + static_cast<size_t>(P) * kRidgePitch + 1, 0.0);
    h[0] = rows;
    h[1] = desc->objective_params[1];
    double* AT = h.data() + 2;
    const double* a = desc->objective_params + 2;
    for (int i = 0; i < rows; ++i)
      for (int j = 0; j < n; ++j) AT[static_cast<size_t>(j) * kRidgePitch + i] = a[static_cast<size_t>(i) * n + j];
    src = h.data();
    np = h.size();
  }
  if (np == 0) return MI355_OK;
  if (ctx->params_dev && ctx->params_resident.size() == np && ctx->params_stream == stream &&
      std::memcmp(ctx->params_resident.data(), src, np * sizeof(double)) == 0)
    return MI355_OK;  // already there, and ordered on this stream (a solve on another stream uploads again)
  ctx->params_resident.clear();
  HIP_TRY(wait_for_last_solve(ctx, ctx->params_stream, stream));
  if (np > ctx->params_cap) {
    if (ctx->params_dev) HIP_TRY(hipFree(ctx->params_dev));
    ctx->params_dev = nullptr;
    ctx->params_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->params_dev), np * sizeof(double)));
    ctx->params_cap = np;
  }
  HIP_TRY(hipMemcpyAsync(ctx->params_dev, src, np * sizeof(double), hipMemcpyHostToDevice, stream));
  ctx->params_resident.assign(src, src + np);
  ctx->params_stream = stream;
  return MI355_OK;
}

// lbfgs.h:126-131: preconditioner_j = 1 / (|H_jj| + eps), IEEE division on the host; *out = null for First mode
int upload_precond(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, hipStream_t stream, const double** out) {
  *out = nullptr;
  if (desc->hessian_diagonal == nullptr) return MI355_OK;
  if (!ctx->precond_dev)
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->precond_dev), MI355_LBFGS_MAX_N * sizeof(double)));
  ctx->precond_host.resize(desc->n);
  for (int j = 0; j < desc->n; ++j) {
    const double h = desc->hessian_diagonal[j];
    if (!(h == h)) return fail(MI355_ERR_INVALID_ARGUMENT, "hessian_diagonal holds a NaN");
    ctx->precond_host[j] = 1.0 / (std::fabs(h) + 2.220446049250313e-16);
  }
  if (ctx->precond_resident != ctx->precond_host || ctx->precond_stream != stream) {
    HIP_TRY(wait_for_last_solve(ctx, ctx->precond_stream, stream));
    ctx->precond_stream = stream;
    ctx->precond_resident.clear();
    HIP_TRY(hipMemcpyAsync(ctx->precond_dev, ctx->precond_host.data(), desc->n * sizeof(double),
                           hipMemcpyHostToDevice, stream));
    ctx->precond_resident = ctx->precond_host;
  }
  *out = ctx->precond_dev;
  return MI355_OK;
}

// ---------------------------------------------------------------------------
// helper kernels
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

__global__ void fill_x0_kernel(int kind, unsigned long long seed, long long first, long long B, int n,
                               double* x0) {
  const long long total = B * n;
  for (long long t = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; t < total;
       t += static_cast<long long>(gridDim.x) * blockDim.x) {
    const long long b = t / n;
    const int i = static_cast<int>(t - b * n);
    const unsigned long long ctr = static_cast<unsigned long long>((first + b) * n + i);
    const unsigned long long h = splitmix64(seed ^ ctr);
    const double u = static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0);  // [0,1)
    double v;
    if (kind == 0) {
      const double base = (i & 1) ? 1.0 : -1.2;
      v = base + 0.1 * (2.0 * u - 1.0);
    } else {
      v = -2.0 + 4.0 * u;
    }
    x0[t] = v;
  }
}

__global__ void cstep_kernel(long long count, double* rec, int* ret) {
  const long long t = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (t >= count) return;
  double* r = rec + t * 13;
  StepInterval iv;
  iv.stx = r[0]; iv.fx = r[1]; iv.dx = r[2]; iv.sty = r[3]; iv.fy = r[4]; iv.dy = r[5]; iv.stp = r[6];
  const double fp = r[7], dp = r[8];
  iv.brackt = r[9] != 0.0;
  const double stpmin = r[10], stpmax = r[11];
  iv.info = static_cast<int>(r[12]);
  iv.rc = 0;
  iv = mt_cstep(iv, fp, dp, stpmin, stpmax);
  r[0] = iv.stx; r[1] = iv.fx; r[2] = iv.dx; r[3] = iv.sty; r[4] = iv.fy; r[5] = iv.dy; r[6] = iv.stp;
  r[9] = iv.brackt ? 1.0 : 0.0;
  r[12] = static_cast<double>(iv.info);
  const int rc = iv.rc;
  ret[t] = rc;
}

// Source-lane maps of the cross-lane primitives: out[p][lane] = lane whose
// value the primitive delivers.  For the swap levels, out holds the PARTNER
// lane (the two result registers are {own row-pair copy, partner copy}).
__global__ void selftest_kernel(int* maps, const double* probe_in, double* probe_out) {
  const int lane = threadIdx.x;
  const double v = static_cast<double>(lane);
  maps[0 * 64 + lane] = static_cast<int>(dpp_mov<kQuadXor1>(v));
  maps[1 * 64 + lane] = static_cast<int>(dpp_mov<kQuadXor2>(v));
  maps[2 * 64 + lane] = static_cast<int>(dpp_mov<kRowHalfMirror>(v));
  maps[3 * 64 + lane] = static_cast<int>(dpp_mov<kRowMirror>(v));
  maps[4 * 64 + lane] = static_cast<int>(add_xor16(v) - v);  // partner lane
  maps[5 * 64 + lane] = static_cast<int>(add_xor32(v) - v);
  maps[6 * 64 + lane] = static_cast<int>(from_next_lane(v));
  maps[7 * 64 + lane] = static_cast<int>(from_prev_lane(v));
  maps[8 * 64 + lane] = static_cast<int>(row_bcast<16, 3>(v));
  maps[9 * 64 + lane] = static_cast<int>(row_bcast<16, 11>(v));
  // 32-lane segments (L-BFGS-B with m = 9, 10): broadcasts and the minimum across the two rows of a segment
  maps[10 * 64 + lane] = static_cast<int>(row_bcast<32, 3>(v));
  maps[11 * 64 + lane] = static_cast<int>(row_bcast<32, 19>(v));
  maps[12 * 64 + lane] = static_cast<int>(xchg16(v));
  maps[13 * 64 + lane] = row_min_i<32>(lane);
  // arithmetic probes: sqrt and division must be correctly rounded (compared
  // against the host's IEEE results by the test).
  const double p = probe_in[lane];
  probe_out[lane] = __builtin_sqrt(p);
  probe_out[64 + lane] = 1.0 / p;
}

}  // namespace

// ---------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------
extern "C" {

int mi355_lbfgs_abi_version(void) { return MI355_LBFGS_ABI_VERSION; }

const char* mi355_lbfgs_last_error(void) { return g_last_error.c_str(); }

int mi355_lbfgs_create(int device, mi355_lbfgs_ctx** out) {
  if (!out) return fail(MI355_ERR_INVALID_ARGUMENT, "null out pointer");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(MI355_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
  if (device < 0 || device >= count) return fail(MI355_ERR_NO_DEVICE, "device index out of range");
  mi355::DeviceGuard device_guard(device);  // the caller's current device is restored on return
  HIP_TRY(device_guard.status());
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MI355_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", built for gfx950 only");
  auto* ctx = new mi355_lbfgs_ctx();
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  // Plateau rings of the persistent kernels (MI355_LBFGS_MAX_PAST doubles per resident wavefront segment), sized
  // once for the fullest grid the chip can hold — 32 wavefronts per CU, eight segments each — so that no launch
  // ever has to grow it (an entry point that is asynchronous on its stream must not synchronise the device).
  ctx->scratch_cap = static_cast<size_t>(ctx->num_cus) * 32 * 8 * MI355_LBFGS_MAX_PAST;
  if (hipEventCreate(&ctx->ev_start) != hipSuccess || hipEventCreate(&ctx->ev_stop) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&ctx->scratch_dev), ctx->scratch_cap * sizeof(double)) != hipSuccess ||
      hipMalloc(reinterpret_cast<void**>(&ctx->queue_dev), kQueueWords * sizeof(unsigned long long)) != hipSuccess) {
    mi355_lbfgs_destroy(ctx);
    return fail(MI355_ERR_HIP, "context allocation failed");
  }
  // experiment knobs: read here, once, and announced — never a getenv on the launch path, never silent
  if (const char* dbg = std::getenv("MI355_DEBUG_SOLVE_WAVES")) ctx->debug_waves = std::atoi(dbg);
  if (const char* dbg = std::getenv("MI355_DEBUG_SOLVE_BLOCKS")) ctx->debug_blocks = std::atoll(dbg);
  if (ctx->debug_waves > 0 || ctx->debug_blocks > 0)
    std::fprintf(stderr, "mi355_lbfgs: EXPERIMENT knobs active on this context (MI355_DEBUG_SOLVE_WAVES=%d, "
                         "MI355_DEBUG_SOLVE_BLOCKS=%lld): the resident grid is capped; results are unchanged\n",
                 ctx->debug_waves, ctx->debug_blocks);
  *out = ctx;
  return MI355_OK;
}

void mi355_lbfgs_destroy(mi355_lbfgs_ctx* ctx) {
  if (!ctx) return;
  mi355::DeviceGuard device_guard(ctx->device);
  if (ctx->params_dev) (void)hipFree(ctx->params_dev);
  if (ctx->wide_ws) (void)hipFree(ctx->wide_ws);
  if (ctx->queue_dev) (void)hipFree(ctx->queue_dev);
  if (ctx->bounds_dev) (void)hipFree(ctx->bounds_dev);
  if (ctx->precond_dev) (void)hipFree(ctx->precond_dev);
  if (ctx->scratch_dev) (void)hipFree(ctx->scratch_dev);
  if (ctx->profile_dev) (void)hipFree(ctx->profile_dev);
  if (ctx->al_workspace) (void)hipFree(ctx->al_workspace);
  if (ctx->gram_params_dev) (void)hipFree(ctx->gram_params_dev);
  if (ctx->gram_rows_dev) (void)hipFree(ctx->gram_rows_dev);
  mi355::destroy_host_pipeline(ctx);
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  delete ctx;
}

int mi355_lbfgs_default_stop(int preset, mi355_lbfgs_stop* out) {
  if (!out) return fail(MI355_ERR_INVALID_ARGUMENT, "null out pointer");
  if (preset < 0 || preset > 2) return fail(MI355_ERR_INVALID_ARGUMENT, "preset must be 0, 1 or 2");
  // DefaultStoppingSolverProgress, solver/progress.h:353-431
  out->num_iterations = 10000;
  out->x_delta = 1e-9;
  out->x_delta_violations = 1;
  out->f_delta = 0.0;
  out->f_delta_violations = 1;
  out->f_delta_relative = 0;
  out->gradient_norm = 1e-5;
  out->gradient_norm_relative = 1;
  out->past = 3;
  out->past_delta = 1e-6;
  if (preset == 1) {  // ConservativeStoppingSolverProgress, solver/progress.h:456-464
    out->gradient_norm = 5e-6;
    out->past = 5;
    out->past_delta = 1e-10;
  }
  if (preset == 2) {  // Lbfgsb default constructor, solver/lbfgsb.h:84-87
    out->f_delta = 2.22e-9;
    out->f_delta_relative = 1;
  }
  return MI355_OK;
}

}  // extern "C"

// Lbfgs (dense_bfgs = false) and Bfgs (true) share the driver, the objectives, the line searches and
// the kernel; they differ in how the search direction is built.
static int minimize_batch_impl(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                               double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                               void* stream_, bool dense_bfgs) {
  int rc = validate(ctx, desc, B, /*allow_wide=*/!dense_bfgs);
  if (rc != MI355_OK) return rc;
  if (B == 0) return MI355_OK;
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MI355_ENTER_DEVICE(ctx);
  if (desc->n > MI355_LBFGS_MAX_N) {
    // one problem per workgroup, state in an HBM workspace (lbfgs_wide_kernel.hpp): Lbfgs<F, m, MoreThuente>, First
    // mode, exact arithmetic
    if (desc->objective != MI355_OBJ_ROSENBROCK && desc->objective != MI355_OBJ_DIAG_QUADRATIC &&
        desc->objective < MI355_OBJ_USER_FIRST)   // (a user objective needs a functor for this regime: dispatch_wide)
      return fail(MI355_ERR_UNSUPPORTED, "n > MI355_LBFGS_MAX_N is built for the Rosenbrock and DiagQuadratic objectives");
    if (desc->arithmetic == MI355_ARITH_FMA)
      return fail(MI355_ERR_UNSUPPORTED, "n > MI355_LBFGS_MAX_N is built in the exact arithmetic");
    if (desc->hessian_diagonal != nullptr)
      return fail(MI355_ERR_UNSUPPORTED, "n > MI355_LBFGS_MAX_N: Second-mode functions through hessian_from_functor only");
    if (desc->trace != nullptr) return fail(MI355_ERR_UNSUPPORTED, "n > MI355_LBFGS_MAX_N: no per-iteration trace");
    if (desc->lanes_per_problem != 0 || desc->elems_per_lane != 0)
      return fail(MI355_ERR_INVALID_ARGUMENT, "n > MI355_LBFGS_MAX_N: leave the mapping fields 0");
    rc = upload_params(ctx, desc, 0, 0, stream);
    if (rc != MI355_OK) return rc;
    WideArgs wa;
    std::memset(&wa, 0, sizeof(wa));
    wa.x0 = x0;
    wa.x_out = x_out;
    wa.f_out = f_out;
    wa.g_out = g_out;
    wa.progress_out = progress_out;
    wa.obj_params = ctx->params_dev;
    wa.B = B;
    wa.n = desc->n;
    wa.m = desc->m;
    wa.stop = desc->stop;
    wa.linesearch = desc->linesearch;
    wa.hess_from_functor = desc->hessian_from_functor;
    return dispatch_wide(ctx, desc->objective, wa, stream);
  }
  if (dense_bfgs) {
    if (desc->n > 64) return fail(MI355_ERR_UNSUPPORTED, "dense BFGS is built for n <= 64 (H lives in LDS)");
    if (desc->objective != MI355_OBJ_ROSENBROCK && desc->objective != MI355_OBJ_DIAG_QUADRATIC &&
        desc->objective < MI355_OBJ_USER_FIRST)
      return fail(MI355_ERR_UNSUPPORTED, "dense BFGS is built for the Rosenbrock, DiagQuadratic and user objectives");
    if (desc->hessian_diagonal != nullptr || desc->hessian_from_functor)
      return fail(MI355_ERR_INVALID_ARGUMENT, "Bfgs takes no Hessian diagonal (solver/bfgs.h uses first-order information only)");
    if (desc->lanes_per_problem != 0 || desc->elems_per_lane != 0) {
      // an explicit mapping must cover exactly the padded width (H is (W * E)^2 doubles) with a built shape
      int P = 8;
      while (P < desc->n) P <<= 1;
      const int W = desc->lanes_per_problem, E = desc->elems_per_lane;
      const bool packed_only = desc->objective >= MI355_OBJ_USER_FIRST;   // (what a user objective's units hold)
      const bool built = (W == 8 && (E == 1 || E == 2 || E == 4)) || (W == 16 && E == 4) ||
                         (!packed_only && ((W == 16 && E == 2) || (W == 32 && (E == 1 || E == 2)) || (W == 64 && E == 1)));
      if (!built || W * E != P)
        return fail(MI355_ERR_INVALID_ARGUMENT,
                    "dense BFGS: lanes_per_problem x elems_per_lane must equal the padded width (8, 16, 32 or 64) in one of "
                    "the built shapes 8x{1,2,4}, 16x{2,4}, 32x{1,2}, 64x1 (0 x 0: the library's choice)");
    }
  }
  // arithmetic policy: the fused kernels are built for Lbfgs + More-Thuente on objectives with an eval_fma
  const bool user_objective = desc->objective >= MI355_OBJ_USER_FIRST;
  if (desc->hessian_from_functor && desc->objective != MI355_OBJ_ROSENBROCK && !user_objective)
    return fail(MI355_ERR_UNSUPPORTED,
                "hessian_from_functor is built for objectives whose device functor has a hess_diag: Rosenbrock and user "
                "functors that define one (constant Hessians: hessian_diagonal)");
  // (a user objective takes the fused kernels only when asked to: MI355_ARITH_FMA is refused by the launch if its
  //  functor has no eval_fma)
  // (Hager-Zhang, round 6: the fused search on the two built-in objectives with an eval_fma, First mode)
  const bool hz_fma_built = !dense_bfgs && desc->linesearch == MI355_LS_HAGER_ZHANG && desc->n <= MI355_LBFGS_MAX_N &&
                            (desc->objective == MI355_OBJ_ROSENBROCK || desc->objective == MI355_OBJ_DIAG_QUADRATIC) &&
                            desc->hessian_diagonal == nullptr && !desc->hessian_from_functor && desc->m <= 32;
  const bool fma_built = hz_fma_built || (!dense_bfgs && desc->linesearch == MI355_LS_MORE_THUENTE &&
                         (desc->objective == MI355_OBJ_ROSENBROCK || desc->objective == MI355_OBJ_DIAG_QUADRATIC ||
                          desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA ||   // (the solver side of that kernel)
                          desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM ||
                          desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM ||
                          (user_objective && desc->arithmetic == MI355_ARITH_FMA)));
  if (desc->arithmetic == MI355_ARITH_FMA && !fma_built)
    return fail(MI355_ERR_UNSUPPORTED,
                "MI355_ARITH_FMA is built for mi355_lbfgs_minimize_batch with the More-Thuente line search on the "
                "Rosenbrock, DiagQuadratic and matrix-core ridge objectives, and with the Hager-Zhang line search on "
                "Rosenbrock and DiagQuadratic (First mode, n <= 256)");
  const bool use_fma = fma_built && desc->arithmetic != MI355_ARITH_EXACT;
  if (desc->trace != nullptr &&
      (desc->objective == MI355_OBJ_AL_COMPOSITE || desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA))
    return fail(MI355_ERR_UNSUPPORTED, "the per-iteration trace is built for the Lbfgs / Bfgs / Lbfgsb solve kernels "
                                       "(not the matrix-core ridge kernel or the composite objective)");
  if (desc->objective == MI355_OBJ_AL_COMPOSITE) {
    if (dense_bfgs) return fail(MI355_ERR_UNSUPPORTED, "the composite objective is built for Lbfgs");
    return auglag_composite_minimize(ctx, desc, B, x0, x_out, f_out, g_out, progress_out, stream);
  }
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA) {
    if (desc->n > kJointCols) return fail(MI355_ERR_UNSUPPORTED, "the matrix-core ridge kernel is built for n <= 64");
    if (desc->m > 10) return fail(MI355_ERR_UNSUPPORTED, "the matrix-core ridge kernel is built for m <= 10");
    if (desc->linesearch != MI355_LS_MORE_THUENTE)
      return fail(MI355_ERR_UNSUPPORTED, "the matrix-core ridge kernel is built with the More-Thuente line search");
    if ((desc->lanes_per_problem != 0 || desc->elems_per_lane != 0) &&
        !(desc->lanes_per_problem == 32 && desc->elems_per_lane == 2) &&
        !(desc->lanes_per_problem == 16 && desc->elems_per_lane == 4))
      return fail(MI355_ERR_INVALID_ARGUMENT,
                  "the matrix-core ridge kernel maps a problem on 32 lanes x 2 elements (default) or 16 lanes x 4");
    // default: eight wavefronts x two problems; 16 x 4 (four wavefronts x four problems, one per SIMD) measured
    // 7 % slower (profiles/r2_ab_ridge_mapping.txt) and is kept selectable
    const int ridge_lanes = desc->lanes_per_problem == 16 ? 16 : 32;
    rc = upload_params(ctx, desc, 32, 2, stream);
    if (rc != MI355_OK) return rc;
    SolveArgs margs;
    std::memset(&margs, 0, sizeof(margs));
    margs.x0 = x0;
    margs.x_out = x_out;
    margs.f_out = f_out;
    margs.g_out = g_out;
    margs.progress_out = progress_out;
    margs.obj_params = ctx->params_dev;
    margs.per_problem = desc->per_problem_data;
    margs.per_problem_stride = desc->per_problem_stride;
    margs.B = B;
    margs.n = desc->n;
    margs.m = desc->m;
    margs.stop = desc->stop;
    margs.hessian_condition_fires =
        (desc->hessian_condition_stop > 0.0 && desc->hessian_condition > desc->hessian_condition_stop) ? 1 : 0;
    rc = upload_precond(ctx, desc, stream, &margs.precond);
    if (rc != MI355_OK) return rc;
    return launch_ridge_mfma(ctx, margs, stream, ridge_lanes, use_fma);
  }
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM) {
    if (dense_bfgs) return fail(MI355_ERR_UNSUPPORTED, "the normal-equation ridge objective is built for Lbfgs");
    SolveArgs gargs;
    std::memset(&gargs, 0, sizeof(gargs));
    gargs.x0 = x0;
    gargs.x_out = x_out;
    gargs.f_out = f_out;
    gargs.g_out = g_out;
    gargs.progress_out = progress_out;
    gargs.B = B;
    gargs.n = desc->n;
    gargs.m = desc->m;
    gargs.stop = desc->stop;
    gargs.hessian_condition_fires =
        (desc->hessian_condition_stop > 0.0 && desc->hessian_condition > desc->hessian_condition_stop) ? 1 : 0;
    rc = upload_precond(ctx, desc, stream, &gargs.precond);
    if (rc != MI355_OK) return rc;
    rc = setup_trace(ctx, desc, B, stream, gargs);
    if (rc != MI355_OK) return rc;
    return ridge_gram_minimize(ctx, desc, gargs, desc->per_problem_data, desc->per_problem_stride, stream, false);
  }
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM) {
    if (dense_bfgs) return fail(MI355_ERR_UNSUPPORTED, "the own-matrix ridge objective is built for Lbfgs");
    SolveArgs gargs;
    std::memset(&gargs, 0, sizeof(gargs));
    gargs.x0 = x0;
    gargs.x_out = x_out;
    gargs.f_out = f_out;
    gargs.g_out = g_out;
    gargs.progress_out = progress_out;
    gargs.B = B;
    gargs.n = desc->n;
    gargs.m = desc->m;
    gargs.stop = desc->stop;
    rc = setup_trace(ctx, desc, B, stream, gargs);
    if (rc != MI355_OK) return rc;
    return ridge_gram_own_minimize(ctx, desc, gargs, desc->per_problem_data, desc->per_problem_stride, stream, false);
  }
  int W = desc->lanes_per_problem, E = desc->elems_per_lane;
  if (dense_bfgs && W == 0 && E == 0) {
    // Bfgs<F, LineSearch> (bfgs.h:39-41): H is P x P doubles per problem in LDS, which — not the register file — caps the
    // problems in flight per CU; the O(n^2) work of a problem is spread over as many lanes as the LDS-bound residency
    // leaves wavefront slots for (profiles/r6_ab_bfgs_mapping.txt).  Any split of the padded width returns the same bits.
    int P = 8;
    while (P < desc->n) P <<= 1;
    bfgs_default_mapping(P, /*packed=*/desc->objective >= MI355_OBJ_USER_FIRST, W, E);
  } else if (W == 0 && E == 0 && use_fma && hz_fma_built) {
    hz_fused_mapping(desc->n, W, E);
  } else if (W == 0 && E == 0) {
    choose_mapping(desc->objective, desc->n, desc->m, desc->history_placement != MI355_HISTORY_LDS, W, E);
    // the condition_hessian test of a non-constant Hessian keeps an n x n matrix per resident problem in LDS and its
    // kernels are built for at most two coordinates per lane: two problems per wavefront at n > 32 (four would not fit)
    if (desc->hessian_from_functor && desc->hessian_condition_stop > 0.0 && E == 4) {
      E = 2;
      W *= 2;
    }
  } else if (!valid_mapping(desc->n, W, E)) {
    return fail(MI355_ERR_INVALID_ARGUMENT,
                "lanes_per_problem x elems_per_lane must be {8,16,32,64} x {1,2,4} (or {4,8} x 8) and cover n");
  }
  rc = upload_params(ctx, desc, W, E, stream);
  if (rc != MI355_OK) return rc;
  SolveArgs args;
  std::memset(&args, 0, sizeof(args));
  args.x0 = x0;
  args.x_out = x_out;
  args.f_out = f_out;
  args.g_out = g_out;
  args.progress_out = progress_out;
  args.obj_params = ctx->params_dev;
  args.per_problem = desc->per_problem_data;
  args.per_problem_stride = desc->per_problem_stride;
  rc = upload_precond(ctx, desc, stream, &args.precond);
  if (rc != MI355_OK) return rc;
  args.B = B;
  args.n = desc->n;
  args.m = desc->m;
  args.stop = desc->stop;
  rc = setup_trace(ctx, desc, B, stream, args);
  if (rc != MI355_OK) return rc;
  // progress.h:318-325: `condition_hessian > stop.condition_hessian` (a NaN condition never fires, as there)
  args.hessian_condition_fires =
      (desc->hessian_condition_stop > 0.0 && desc->hessian_condition > desc->hessian_condition_stop) ? 1 : 0;
  // y half of the history in registers (0 = library default: yes when a variant exists)
  int mr = (desc->history_placement == MI355_HISTORY_LDS || desc->hessian_from_functor) ? 0 : desc->m;
  args.hess_from_functor = desc->hessian_from_functor;   // (refused by the launch if the functor has no hess_diag)
  if (desc->hessian_from_functor) {
    // the Hessian changes with x: the kernel evaluates the condition number itself (hessian_condition_device.hpp)
    args.hessian_condition_fires = 0;
    args.hessian_condition_stop = desc->hessian_condition_stop;
  }
  if (desc->linesearch == MI355_LS_HAGER_ZHANG)            // Lbfgs<F, m, HagerZhang> (lbfgs.h:41)
    mr = (use_fma && hz_fma_built) ? kMrHzFusedBase - ((desc->history_placement == MI355_HISTORY_LDS) ? 0 : (desc->m <= 10 ? desc->m : 0))
                                   : -1;
  if (dense_bfgs) {                                        // Bfgs<F, LineSearch> (bfgs.h:39-41)
    mr = (desc->linesearch == MI355_LS_HAGER_ZHANG) ? -3 : -2;
  }
  if (use_fma && mr >= 0) mr |= kArithFmaBit;   // (the Hager-Zhang codes carry the policy themselves)
  return dispatch(ctx, W, E, desc->objective, mr, args, stream, /*eval_only=*/false);
}

extern "C" {

int mi355_lbfgs_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B,
                               const double* x0, double* x_out, double* f_out, double* g_out,
                               mi355_lbfgs_progress* progress_out, void* stream_) {
  return minimize_batch_impl(ctx, desc, B, x0, x_out, f_out, g_out, progress_out, stream_, false);
}

int mi355_bfgs_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B,
                              const double* x0, double* x_out, double* f_out, double* g_out,
                              mi355_lbfgs_progress* progress_out, void* stream_) {
  return minimize_batch_impl(ctx, desc, B, x0, x_out, f_out, g_out, progress_out, stream_, true);
}

}  // extern "C"

// device-pointer solve for host_pipeline.hip (solver 0 = Lbfgs, 1 = dense Bfgs)
int mi355_minimize_batch_device(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                                void* stream, int solver) {
  return minimize_batch_impl(ctx, desc, B, x0, x_out, f_out, g_out, progress_out, stream, solver == 1);
}

namespace {

int ensure_bounds(mi355_lbfgs_ctx* ctx, size_t doubles) {
  if (doubles <= ctx->bounds_cap) return MI355_OK;
  if (ctx->bounds_dev) HIP_TRY(hipFree(ctx->bounds_dev));
  ctx->bounds_dev = nullptr;
  ctx->bounds_cap = 0;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->bounds_dev), doubles * sizeof(double)));
  ctx->bounds_cap = doubles;
  return MI355_OK;
}

}  // namespace

extern "C" int mi355_lbfgsb_minimize_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, const double* lower,
                                           const double* upper, int64_t B, const double* x0, double* x_out,
                                           double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                                           void* stream_) {
  int rc = validate(ctx, desc, B);
  if (rc != MI355_OK) return rc;
  if (desc->m > 10) return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B is built for m <= 10 (5 is the reference default)");
  if (desc->n > 256) return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B is built for n <= 256");
  if (desc->hessian_diagonal != nullptr || desc->hessian_from_functor)
    return fail(MI355_ERR_UNSUPPORTED, "Lbfgsb has no preconditioned (Second-mode) path (lbfgsb.h:48-49)");
  if (desc->lanes_per_problem != 0 || desc->elems_per_lane != 0 || desc->history_placement != 0)
    return fail(MI355_ERR_INVALID_ARGUMENT, "L-BFGS-B chooses its own mapping: leave the mapping fields 0");
  if ((lower == nullptr) != (upper == nullptr))
    return fail(MI355_ERR_INVALID_ARGUMENT, "lower and upper must both be given or both be NULL");
  if (B == 0) return MI355_OK;
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MI355_ENTER_DEVICE(ctx);
  const int n = desc->n;
  // 32 lanes per problem: m = 9, 10 (2M = 20 rows of the compact representation take two DPP rows), n > 128, and
  // m = 6..10 above n = 64 (dispatch_lbfgsb_w32 / dispatch_lbfgsb_caps_*); sixteen otherwise
  const bool two_rows = desc->m > 8 || n > 128 || (desc->m > 5 && n > 64);
  const int E = two_rows ? ((n <= 32) ? 1 : ((n <= 64) ? 2 : ((n <= 128) ? 4 : 8)))
                         : ((n <= 16) ? 1 : ((n <= 32) ? 2 : ((n <= 64) ? 4 : 8)));
  // arithmetic policy: the relaxed-algebra kernels (lbfgsb_fast_kernel.hpp) are built for 16 lanes per problem with
  // the More-Thuente line search on the objectives that have a fused form — m <= 8 up to n = 64, m <= 5 up to n = 128.
  // A user objective takes them only when asked to (MI355_ARITH_FMA; refused by the launch if the functor has no
  // eval_fma), as for Lbfgs.
  const bool user_objective = desc->objective >= MI355_OBJ_USER_FIRST;
  const bool fast_shape = two_rows ? (desc->m > 8 && n <= 64 && !user_objective)      // m = 9, 10 on thirty-two lanes
                                   : desc->m <= (n <= 64 ? 8 : 5);
  const bool fast_built = fast_shape && desc->linesearch == MI355_LS_MORE_THUENTE &&
                          (desc->objective == MI355_OBJ_ROSENBROCK || desc->objective == MI355_OBJ_DIAG_QUADRATIC ||
                           (user_objective && desc->arithmetic == MI355_ARITH_FMA));
  if (desc->arithmetic == MI355_ARITH_FMA && !fast_built)
    return fail(MI355_ERR_UNSUPPORTED,
                "MI355_ARITH_FMA (relaxed algebra) for L-BFGS-B is built for the More-Thuente line search on the Rosenbrock "
                "/ DiagQuadratic objectives and user functors with an eval_fma: m <= 10 (n <= 64), m <= 5 (n <= 128)");
  bool use_fast = fast_built && desc->arithmetic != MI355_ARITH_EXACT;
  // MI355_ARITH_DEFAULT stays inside the envelope where the relaxed algebra is pinned to 1e-6 of the reference binary
  // (tests/test_relaxed_envelope.py): a diagonal quadratic whose spectrum spreads over more than
  // MI355_LBFGSB_RELAXED_MAX_SPREAD takes the reference-order kernel; MI355_ARITH_FMA still forces the relaxed one.
  if (use_fast && desc->arithmetic == MI355_ARITH_DEFAULT && desc->objective == MI355_OBJ_DIAG_QUADRATIC &&
      desc->objective_params != nullptr && desc->n_params >= n) {
    double lo_a = 1.7976931348623157e308, hi_a = 0.0;
    for (int j = 0; j < n; ++j) {
      const double a = std::fabs(desc->objective_params[j]);
      lo_a = a < lo_a ? a : lo_a;
      hi_a = a > hi_a ? a : hi_a;
    }
    if (!(hi_a <= MI355_LBFGSB_RELAXED_MAX_SPREAD * lo_a)) use_fast = false;
  }
  if (!lower) {  // default box: lowest() .. max()  (lbfgsb.h:124-129)
    rc = ensure_bounds(ctx, 2 * static_cast<size_t>(MI355_LBFGS_MAX_N));
    if (rc != MI355_OK) return rc;
    std::vector<double>& h = ctx->bounds_host;
    h.assign(2 * static_cast<size_t>(n), 0.0);
    for (int j = 0; j < n; ++j) {
      h[j] = -1.7976931348623157e308;
      h[n + j] = 1.7976931348623157e308;
    }
    HIP_TRY(hipMemcpyAsync(ctx->bounds_dev, h.data(), 2 * n * sizeof(double), hipMemcpyHostToDevice, stream));
    lower = ctx->bounds_dev;
    upper = ctx->bounds_dev + n;
  }
  rc = upload_params(ctx, desc, two_rows ? 32 : 16, E, stream);
  if (rc != MI355_OK) return rc;
  LbfgsbArgs args;
  std::memset(&args, 0, sizeof(args));
  args.s.x0 = x0;
  args.s.x_out = x_out;
  args.s.f_out = f_out;
  args.s.g_out = g_out;
  args.s.progress_out = progress_out;
  args.s.obj_params = ctx->params_dev;
  args.s.per_problem = desc->per_problem_data;
  args.s.per_problem_stride = desc->per_problem_stride;
  args.s.precond = nullptr;
  args.s.scratch = nullptr;
  args.s.next_problem = nullptr;
  args.s.B = B;
  args.s.n = n;
  args.s.m = desc->m;
  args.s.stop = desc->stop;
  rc = setup_trace(ctx, desc, B, stream, args.s);
  if (rc != MI355_OK) return rc;
  args.lower = lower;
  args.upper = upper;
  args.relaxed = use_fast ? 1 : 0;
  if (desc->objective >= MI355_OBJ_USER_FIRST) {
    const UserEntry* u = find_user_objective(desc->objective);
    if (!u || !u->lbfgsb)
      return fail(MI355_ERR_UNSUPPORTED, "no user objective with this id is compiled into this library for L-BFGS-B");
    return u->lbfgsb(ctx, two_rows ? 32 : 16, E, desc->linesearch, args, stream);   // (refuses shapes it was not built for)
  }
  // the shapes added in round 3 (history sizes 6..10 above n = 64, under Hager-Zhang and on the ridge objective)
  const bool ridge = desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE;
  const bool hager_zhang = desc->linesearch == MI355_LS_HAGER_ZHANG;
  if (hager_zhang && n > 64)   // round 4: the alternative line search above n = 64
    return dispatch_lbfgsb_caps_e(ctx, two_rows ? 32 : 16, E, desc->objective, args, stream);
  if (desc->m > 5 && hager_zhang)
    return dispatch_lbfgsb_caps_b(ctx, two_rows ? 32 : 16, E, desc->objective, desc->linesearch, args, stream);
  if (desc->m > 5 && (ridge || n > 64))
    return dispatch_lbfgsb_caps_a(ctx, two_rows ? 32 : 16, E, desc->objective, desc->linesearch, args, stream);
  if (use_fast) return dispatch_lbfgsb_fast(ctx, two_rows ? 32 : 16, E, desc->objective, args, stream);
  if (two_rows) return dispatch_lbfgsb_w32(ctx, desc->objective, desc->linesearch, args, stream);
  return dispatch_lbfgsb_e(ctx, E, desc->objective, desc->linesearch, args, stream);
}

extern "C" {

int mi355_lbfgs_last_kernel_ms(mi355_lbfgs_ctx* ctx, float* ms) {
  if (!ctx || !ms) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  if (!ctx->timed) return fail(MI355_ERR_INVALID_ARGUMENT, "no solve has been launched on this context");
  HIP_TRY(hipEventSynchronize(ctx->ev_stop));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev_start, ctx->ev_stop));
  return MI355_OK;
}

int mi355_lbfgs_last_launch(mi355_lbfgs_ctx* ctx, int32_t* lanes_per_problem, int32_t* elems_per_lane,
                            int32_t* blocks, int32_t* threads, int32_t* lds_bytes,
                            int32_t* y_columns_in_registers) {
  if (!ctx) return fail(MI355_ERR_INVALID_ARGUMENT, "null context");
  if (lanes_per_problem) *lanes_per_problem = ctx->last_W;
  if (elems_per_lane) *elems_per_lane = ctx->last_E;
  if (blocks) *blocks = ctx->last_blocks;
  if (threads) *threads = ctx->last_threads;
  if (lds_bytes) *lds_bytes = ctx->last_lds;
  if (y_columns_in_registers) *y_columns_in_registers = ctx->last_mr;
  return MI355_OK;
}

int mi355_lbfgs_hessian_condition(const double* hessian, int32_t n, double* condition_out) {
  if (!hessian || !condition_out || n < 1 || n > MI355_LBFGS_MAX_N) return fail(MI355_ERR_INVALID_ARGUMENT, "bad argument");
  const size_t nn = static_cast<size_t>(n);
  std::vector<double> lu(hessian, hessian + nn * nn), inv(nn * nn, 0.0), col(nn);
  std::vector<int> piv(nn);
  auto at = [&](std::vector<double>& m, int i, int j) -> double& { return m[static_cast<size_t>(i) * nn + j]; };
  for (int k = 0; k < n; ++k) {  // right-looking LU, partial (first maximum) pivoting
    int p = k;
    double best = std::fabs(at(lu, k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(at(lu, i, k)) > best) {
        best = std::fabs(at(lu, i, k));
        p = i;
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
  for (int c = 0; c < n; ++c) {  // inverse, one unit vector at a time
    std::fill(col.begin(), col.end(), 0.0);
    col[static_cast<size_t>(c)] = 1.0;
    for (int k = 0; k < n; ++k) std::swap(col[static_cast<size_t>(k)], col[static_cast<size_t>(piv[static_cast<size_t>(k)])]);
    for (int j = 0; j < n; ++j)
      for (int i = j + 1; i < n; ++i) col[static_cast<size_t>(i)] -= col[static_cast<size_t>(j)] * at(lu, i, j);
    for (int j = n - 1; j >= 0; --j) {
      col[static_cast<size_t>(j)] = col[static_cast<size_t>(j)] / at(lu, j, j);
      for (int i = 0; i < j; ++i) col[static_cast<size_t>(i)] -= col[static_cast<size_t>(j)] * at(lu, i, j);
    }
    for (int i = 0; i < n; ++i) at(inv, i, c) = col[static_cast<size_t>(i)];
  }
  // Frobenius norms summed in COLUMN-major order — the storage order the reference's `current_hessian.norm()` /
  // `.inverse().norm()` walk (progress.h:203-210) — so that a computed inverse that is not exactly symmetric gives the
  // same chain of additions as there
  double sh = 0.0, si = 0.0;
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      const double h = hessian[static_cast<size_t>(i) * nn + j], v = at(inv, i, j);
      sh += h * h;
      si += v * v;
    }
  *condition_out = std::sqrt(sh) * std::sqrt(si);
  return MI355_OK;
}

int mi355_lbfgs_last_arithmetic(mi355_lbfgs_ctx* ctx, int32_t* arithmetic) {
  if (!ctx || !arithmetic) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  *arithmetic = ctx->last_arith ? ctx->last_arith : MI355_ARITH_EXACT;
  return MI355_OK;
}

#if defined(MI355_LBFGSB_PHASE_TIMING) || defined(MI355_LBFGS_PHASE_TIMING)
// profiling builds only (not part of include/mi355_lbfgs.h): per-phase cycle sums of the last L-BFGS-B launch
int mi355_lbfgsb_phase_cycles(mi355_lbfgs_ctx* ctx, unsigned long long* out16) {
  if (!ctx || !out16 || !ctx->profile_dev) return fail(MI355_ERR_INVALID_ARGUMENT, "no phase counters");
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out16, ctx->profile_dev, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return MI355_OK;
}
#endif

int mi355_lbfgs_fill_x0(mi355_lbfgs_ctx* ctx, int32_t kind, uint64_t seed, int64_t first_problem,
                        int64_t B, int32_t n, double* x0, void* stream_) {
  if (!ctx || !x0) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  if (kind != 0 && kind != 1) return fail(MI355_ERR_INVALID_ARGUMENT, "kind must be 0 or 1");
  if (B < 0 || n < 1) return fail(MI355_ERR_INVALID_ARGUMENT, "bad size");
  if (B == 0) return MI355_OK;
  MI355_ENTER_DEVICE(ctx);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const long long total = static_cast<long long>(B) * n;
  const int threads = 256;
  long long blocks = (total + threads - 1) / threads;
  if (blocks > 256LL * 32) blocks = 256LL * 32;
  hipLaunchKernelGGL(fill_x0_kernel, dim3(static_cast<unsigned>(blocks)), dim3(threads), 0, stream, kind,
                     static_cast<unsigned long long>(seed), static_cast<long long>(first_problem),
                     static_cast<long long>(B), n, x0);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

int mi355_lbfgs_eval_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x,
                           double* f_out, double* g_out, void* stream_) {
  int rc = validate(ctx, desc, B);
  if (rc != MI355_OK) return rc;
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA)
    return fail(MI355_ERR_UNSUPPORTED, "the matrix-core ridge objective has solve entry points only");
  if (B == 0) return MI355_OK;
  if (!x || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x / f_out");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MI355_ENTER_DEVICE(ctx);
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM) {
    SolveArgs gargs;
    std::memset(&gargs, 0, sizeof(gargs));
    gargs.x0 = x;
    gargs.f_out = f_out;
    gargs.g_out = g_out;
    gargs.B = B;
    gargs.n = desc->n;
    gargs.m = desc->m;
    gargs.stop = desc->stop;
    return ridge_gram_minimize(ctx, desc, gargs, desc->per_problem_data, desc->per_problem_stride, stream, true);
  }
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM) {
    SolveArgs gargs;
    std::memset(&gargs, 0, sizeof(gargs));
    gargs.x0 = x;
    gargs.f_out = f_out;
    gargs.g_out = g_out;
    gargs.B = B;
    gargs.n = desc->n;
    gargs.m = desc->m;
    gargs.stop = desc->stop;
    return ridge_gram_own_minimize(ctx, desc, gargs, desc->per_problem_data, desc->per_problem_stride, stream, true);
  }
  int W = desc->lanes_per_problem, E = desc->elems_per_lane;
  if (W == 0 && E == 0) {
    choose_mapping(desc->objective, desc->n, desc->m, false, W, E);
  } else if (!valid_mapping(desc->n, W, E)) {
    return fail(MI355_ERR_INVALID_ARGUMENT, "invalid lanes_per_problem / elems_per_lane");
  }
  rc = upload_params(ctx, desc, W, E, stream);
  if (rc != MI355_OK) return rc;
  SolveArgs args;
  std::memset(&args, 0, sizeof(args));
  args.per_problem = desc->per_problem_data;
  args.per_problem_stride = desc->per_problem_stride;
  args.precond = nullptr;
  args.scratch = nullptr;
  args.x0 = x;
  args.f_out = f_out;
  args.g_out = g_out;
  args.obj_params = ctx->params_dev;
  args.B = B;
  args.n = desc->n;
  args.m = desc->m;
  return dispatch(ctx, W, E, desc->objective, desc->arithmetic == MI355_ARITH_FMA ? kArithFmaBit : 0, args, stream,
                  /*eval_only=*/true);
}

int mi355_lbfgs_hz_search_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x,
                                const double* direction, const double* alpha_init, double* x_out, double* f_out,
                                double* g_out, double* alpha_out, uint32_t* nfev_out, void* stream_) {
  int rc = validate(ctx, desc, B);
  if (rc != MI355_OK) return rc;
  if (desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA || desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM ||
      desc->objective == MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM)
    return fail(MI355_ERR_UNSUPPORTED, "the matrix-core and normal-equation ridge objectives have solve entry points only");
  if (B == 0) return MI355_OK;
  if (!x || !direction || !alpha_init || !x_out || !f_out || !alpha_out)
    return fail(MI355_ERR_INVALID_ARGUMENT, "null x / direction / alpha_init / x_out / f_out / alpha_out");
  if (desc->arithmetic == MI355_ARITH_FMA)
    return fail(MI355_ERR_UNSUPPORTED, "the Hager-Zhang search is built with the exact arithmetic only");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  MI355_ENTER_DEVICE(ctx);
  int W = desc->lanes_per_problem, E = desc->elems_per_lane;
  if (W == 0 && E == 0) {
    choose_mapping(desc->objective, desc->n, desc->m, false, W, E);
  } else if (!valid_mapping(desc->n, W, E)) {
    return fail(MI355_ERR_INVALID_ARGUMENT, "invalid lanes_per_problem / elems_per_lane");
  }
  rc = upload_params(ctx, desc, W, E, stream);
  if (rc != MI355_OK) return rc;
  SolveArgs args;
  std::memset(&args, 0, sizeof(args));
  args.per_problem = desc->per_problem_data;
  args.per_problem_stride = desc->per_problem_stride;
  args.x0 = x;
  args.x_out = x_out;
  args.f_out = f_out;
  args.g_out = g_out;
  args.obj_params = ctx->params_dev;
  args.B = B;
  args.n = desc->n;
  args.m = desc->m;
  args.ls_direction = direction;
  args.ls_alpha_init = alpha_init;
  args.ls_alpha_out = alpha_out;
  args.ls_nfev_out = nfev_out;
  return dispatch(ctx, W, E, desc->objective, 0, args, stream, /*eval_only=*/true);
}

int mi355_lbfgs_hz_search_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x,
                               const double* direction, const double* alpha_init, double* x_out, double* f_out,
                               double* g_out, double* alpha_out, uint32_t* nfev_out) {
  int rc = validate(ctx, desc, B);
  if (rc != MI355_OK) return rc;
  if (B == 0) return MI355_OK;
  if (!x || !direction || !alpha_init || !x_out || !f_out || !alpha_out)
    return fail(MI355_ERR_INVALID_ARGUMENT, "null x / direction / alpha_init / x_out / f_out / alpha_out");
  if (desc->per_problem_data != nullptr)
    return fail(MI355_ERR_UNSUPPORTED, "the host-pointer line search takes objectives without per-problem data");
  MI355_ENTER_DEVICE(ctx);
  const size_t vec = static_cast<size_t>(B) * desc->n * sizeof(double), sc = static_cast<size_t>(B) * sizeof(double);
  char* buf = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), 4 * vec + 3 * sc + static_cast<size_t>(B) * sizeof(uint32_t)));
  double* d_x = reinterpret_cast<double*>(buf);
  double* d_s = reinterpret_cast<double*>(buf + vec);
  double* d_xo = reinterpret_cast<double*>(buf + 2 * vec);
  double* d_go = reinterpret_cast<double*>(buf + 3 * vec);
  double* d_a0 = reinterpret_cast<double*>(buf + 4 * vec);
  double* d_f = reinterpret_cast<double*>(buf + 4 * vec + sc);
  double* d_a = reinterpret_cast<double*>(buf + 4 * vec + 2 * sc);
  uint32_t* d_nf = reinterpret_cast<uint32_t*>(buf + 4 * vec + 3 * sc);
  hipError_t e = hipMemcpy(d_x, x, vec, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_s, direction, vec, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_a0, alpha_init, sc, hipMemcpyHostToDevice);
  rc = MI355_OK;
  if (e == hipSuccess) {
    rc = mi355_lbfgs_hz_search_batch(ctx, desc, B, d_x, d_s, d_a0, d_xo, d_f, d_go, d_a, d_nf, nullptr);
    if (rc == MI355_OK) e = hipDeviceSynchronize();
  }
  if (rc == MI355_OK && e == hipSuccess) e = hipMemcpy(x_out, d_xo, vec, hipMemcpyDeviceToHost);
  if (rc == MI355_OK && e == hipSuccess && g_out) e = hipMemcpy(g_out, d_go, vec, hipMemcpyDeviceToHost);
  if (rc == MI355_OK && e == hipSuccess) e = hipMemcpy(f_out, d_f, sc, hipMemcpyDeviceToHost);
  if (rc == MI355_OK && e == hipSuccess) e = hipMemcpy(alpha_out, d_a, sc, hipMemcpyDeviceToHost);
  if (rc == MI355_OK && e == hipSuccess && nfev_out)
    e = hipMemcpy(nfev_out, d_nf, static_cast<size_t>(B) * sizeof(uint32_t), hipMemcpyDeviceToHost);
  (void)hipFree(buf);
  if (rc != MI355_OK) return rc;
  if (e != hipSuccess) return fail(MI355_ERR_HIP, hipGetErrorString(e));
  return MI355_OK;
}

int mi355_lbfgs_cstep_batch(mi355_lbfgs_ctx* ctx, int64_t count, double* records, int32_t* ret_out,
                            void* stream_) {
  if (!ctx || !records || !ret_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  if (count < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative count");
  if (count == 0) return MI355_OK;
  MI355_ENTER_DEVICE(ctx);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int threads = 64;
  const long long blocks = (count + threads - 1) / threads;
  hipLaunchKernelGGL(cstep_kernel, dim3(static_cast<unsigned>(blocks)), dim3(threads), 0, stream,
                     static_cast<long long>(count), records, ret_out);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

int mi355_lbfgs_cstep_host(mi355_lbfgs_ctx* ctx, int64_t count, double* records, int32_t* ret_out) {
  if (!ctx || !records || !ret_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  if (count < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative count");
  if (count == 0) return MI355_OK;
  MI355_ENTER_DEVICE(ctx);
  const size_t rb = static_cast<size_t>(count) * 13 * sizeof(double);
  const size_t ib = static_cast<size_t>(count) * sizeof(int32_t);
  char* buf = nullptr;
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&buf), rb + ib));
  hipError_t e = hipMemcpy(buf, records, rb, hipMemcpyHostToDevice);
  int rc = MI355_OK;
  if (e == hipSuccess) {
    rc = mi355_lbfgs_cstep_batch(ctx, count, reinterpret_cast<double*>(buf),
                                 reinterpret_cast<int32_t*>(buf + rb), nullptr);
    if (rc == MI355_OK) e = hipDeviceSynchronize();
    if (rc == MI355_OK && e == hipSuccess) e = hipMemcpy(records, buf, rb, hipMemcpyDeviceToHost);
    if (rc == MI355_OK && e == hipSuccess) e = hipMemcpy(ret_out, buf + rb, ib, hipMemcpyDeviceToHost);
  }
  (void)hipFree(buf);
  if (rc != MI355_OK) return rc;
  if (e != hipSuccess) return fail(MI355_ERR_HIP, std::string("cstep host: ") + hipGetErrorString(e));
  return MI355_OK;
}

int mi355_lbfgs_selftest(mi355_lbfgs_ctx* ctx, int32_t* lane_maps, const double* probe_in,
                         double* probe_out, void* stream_) {
  if (!ctx || !lane_maps || !probe_in || !probe_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  MI355_ENTER_DEVICE(ctx);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, lane_maps, probe_in, probe_out);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

}  // extern "C"
