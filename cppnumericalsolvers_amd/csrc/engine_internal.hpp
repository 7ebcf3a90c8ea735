// engine_internal.hpp — shared by the translation units of libmi355_lbfgs.so.
//
// The library is built from several .hip files compiled in parallel (cppnumericalsolvers_amd/_build.py):
// mi355_lbfgs.hip holds the C-ABI (validation, mapping choice, scratch), dispatch_w{8,16,32,64}.hip hold
// the solve / evaluation / line-search kernels of one lanes-per-problem value each, dispatch_lbfgsb.hip
// the L-BFGS-B kernels.  Kernels are instantiated implicitly by the launch templates below, inside the
// translation unit that calls dispatch_e<W> / dispatch_lbfgsb<E>.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi355_lbfgs.h"
#include "lbfgs_kernel.hpp"
#include "lbfgsb_kernel.hpp"

constexpr int kQueueWords = 4;  // work-queue head (+ spare words), zeroed before every launch

struct mi355_lbfgs_ctx {
  int device = 0;
  int num_cus = 0;
  double* params_dev = nullptr;  // objective parameter blob
  size_t params_cap = 0;         // doubles
  std::vector<double> params_host;  // staging for blobs the library re-lays out (kept alive for async copies)
  // what params_dev / precond_dev currently hold (empty = unknown): an upload of identical contents is skipped, so the
  // chunks of a pipelined host batch do not re-copy the objective's blob from pageable memory -- a host-synchronous
  // copy that would keep chunk c + 1 from being enqueued while chunk c runs
  std::vector<double> params_resident, precond_resident;
  std::vector<double> bounds_host;  // default box of the L-BFGS-B entry point
  unsigned long long* queue_dev = nullptr;  // work-queue head of the persistent solve kernel
  double* bounds_dev = nullptr;             // default (unbounded) box / staging for host-pointer bounds
  size_t bounds_cap = 0;                    // doubles
  unsigned long long* profile_dev = nullptr;  // phase counters of the profiling builds
  double* scratch_dev = nullptr;            // plateau rings of the scalars_in_registers() kernels
  size_t scratch_cap = 0;                   // doubles
  double* precond_dev = nullptr;            // Second-mode diagonal preconditioner, MI355_LBFGS_MAX_N doubles
  std::vector<double> precond_host;
  void* wide_ws = nullptr;                  // per-workgroup state of the n > MI355_LBFGS_MAX_N kernel, grows only
  size_t wide_ws_cap = 0;                   // bytes
  size_t device_total_bytes = 0;            // hipMemGetInfo, cached by the first launch of that kernel
  void* al_workspace = nullptr;             // augmented-Lagrangian state arrays (auglag.hip), grows only
  size_t al_workspace_cap = 0;              // bytes
  mi355::TraceArgs* trace_dev = nullptr;    // the active trace's description (mi355_lbfgs_trace), device copy
  mi355::TraceArgs trace_host;
  // host-pointer entry points (host_pipeline.hip): pinned staging and device buffers, two slots, grow-only
  struct HostStage {
    char* pinned = nullptr;    // hipHostMalloc: [inputs | outputs] of one chunk
    char* device = nullptr;    // hipMalloc, same layout
    size_t cap = 0;            // bytes of each
    hipEvent_t in_ready = nullptr, solved = nullptr, out_ready = nullptr;
    hipEvent_t piece_ready[8] = {};  // the output rows travel back in pieces (see unstage in host_pipeline.hip)
  } stage[2];
  hipStream_t stream_in = nullptr, stream_solve = nullptr, stream_out = nullptr;
  unsigned long long* flags_dev = nullptr;  // [3] convergence record of the last sharded solve (host_pipeline.hip)
  // normal-equation ridge objective (dispatch_ridge_gram.hip): shared blob [rows, lambda, G, A padded] and the
  // per-problem rows (c_b, yy_b) the pre-pass writes; grow-only.  gram_key: the parameters G was built from
  double* gram_params_dev = nullptr;
  size_t gram_params_cap = 0;  // doubles
  double* gram_rows_dev = nullptr;
  size_t gram_rows_cap = 0;    // doubles
  std::vector<double> gram_host, gram_key;
  int gram_key_n = 0;
  hipStream_t gram_stream = nullptr, params_stream = nullptr, precond_stream = nullptr;  // streams the cached blobs are ordered on
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool ev_start_armed = false;  // the caller recorded ev_start before its own preparatory kernels: launch_solve keeps it
  bool timed = false;
  int last_W = 0, last_E = 0, last_blocks = 0, last_threads = 0, last_lds = 0, last_mr = 0, last_arith = 0;
  // experiment knobs, read ONCE from the environment when the context is created (mi355_lbfgs_create prints a notice
  // when one is set; 0 = not set): MI355_DEBUG_SOLVE_WAVES caps the wavefronts of a workgroup, MI355_DEBUG_SOLVE_BLOCKS
  // the resident grid; neither changes a result (scripts/ and profiles/ say where they were used)
  int debug_waves = 0;
  long long debug_blocks = 0;
};

namespace mi355 {

// sets the thread's last-error text (mi355_lbfgs_last_error) and returns `code`
int fail(int code, const std::string& msg);

#define HIP_TRY(expr)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (expr);                                                                 \
    if (e_ != hipSuccess)                                                                   \
      return ::mi355::fail(MI355_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Makes ctx's device current for the duration of an entry point and restores the caller's device on the
// way out: a multi-GPU host process (one context per device, possibly torch beside it) keeps its own
// notion of the current device.
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&previous_) != hipSuccess) previous_ = -1;
    status_ = (previous_ == device) ? hipSuccess : hipSetDevice(device);
    switched_ = (status_ == hipSuccess && previous_ != device);
  }
  ~DeviceGuard() {
    if (switched_ && previous_ >= 0) (void)hipSetDevice(previous_);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
  hipError_t status() const { return status_; }

 private:
  int previous_ = -1;
  hipError_t status_ = hipSuccess;
  bool switched_ = false;
};
#define MI355_ENTER_DEVICE(ctx)                         \
  ::mi355::DeviceGuard device_guard_((ctx)->device);    \
  HIP_TRY(device_guard_.status())

// one function per lanes-per-problem value, each in its own translation unit.  `mr`: history placement / solver
// variant (see launch_solve_mr); bit 8 of it (kArithFmaBit) selects the fused arithmetic policy.
constexpr int kArithFmaBit = 256;
// ... and codes at or below kMrHzFusedBase select Lbfgs<F, m, HagerZhang> in the FUSED arithmetic with the y half of the
// history in registers (round 6): code = kMrHzFusedBase - m, m = 0 keeping both halves in the LDS ring.  (-1 stays the
// exact-arithmetic LDS-ring kernel of rounds 2-5.)
constexpr int kMrHzFusedBase = -1000;
int dispatch_w8(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                bool eval_only);
int dispatch_w4(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                bool eval_only);  // four lanes x eight coordinates (n <= 32) only
int dispatch_w16(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                 bool eval_only);
int dispatch_w32(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                 bool eval_only);
int dispatch_w64(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                 bool eval_only);
int dispatch_lbfgsb_e4(mi355_lbfgs_ctx* ctx, int E, int objective, int linesearch, const LbfgsbArgs& args,
                       hipStream_t stream);
// L-BFGS-B with 32 lanes per problem (m = 9, 10), dispatch_lbfgsb_w32.hip
int dispatch_lbfgsb_w32(mi355_lbfgs_ctx* ctx, int objective, int linesearch, const LbfgsbArgs& args, hipStream_t stream);
// Lbfgs for n > MI355_LBFGS_MAX_N: one problem per workgroup, state in HBM (dispatch_wide.hip)
struct WideArgs;
int dispatch_wide(mi355_lbfgs_ctx* ctx, int objective, const WideArgs& args, hipStream_t stream);
// history sizes 6..10 above n = 64 and on the ridge objective (More-Thuente) / under Hager-Zhang (n <= 64):
// dispatch_lbfgsb_caps_a.hip, dispatch_lbfgsb_caps_b.hip
int dispatch_lbfgsb_caps_a(mi355_lbfgs_ctx* ctx, int W, int E, int objective, int linesearch, const LbfgsbArgs& args,
                           hipStream_t stream);
int dispatch_lbfgsb_caps_b(mi355_lbfgs_ctx* ctx, int W, int E, int objective, int linesearch, const LbfgsbArgs& args,
                           hipStream_t stream);
int dispatch_lbfgsb_caps_b32(mi355_lbfgs_ctx* ctx, int E, int objective, const LbfgsbArgs& args, hipStream_t stream);
// Hager-Zhang above n = 64 (dispatch_lbfgsb_caps_e.hip: m <= 5, n <= 256; dispatch_lbfgsb_caps_f.hip: m = 6..10, n <= 128)
int dispatch_lbfgsb_caps_e(mi355_lbfgs_ctx* ctx, int W, int E, int objective, const LbfgsbArgs& args, hipStream_t stream);
int dispatch_lbfgsb_caps_f(mi355_lbfgs_ctx* ctx, int objective, const LbfgsbArgs& args, hipStream_t stream);
int dispatch_lbfgsb_caps_ridge(mi355_lbfgs_ctx* ctx, int W, int E, const LbfgsbArgs& args, hipStream_t stream);
int dispatch_lbfgsb_e(mi355_lbfgs_ctx* ctx, int E, int objective, int linesearch, const LbfgsbArgs& args,
                      hipStream_t stream);
// L-BFGS-B under the relaxed-algebra policy (lbfgsb_fast_kernel.hpp), dispatch_lbfgsb_fast.hip: 16 lanes per problem,
// More-Thuente, Rosenbrock / DiagQuadratic, m <= 8 (n <= 64) or m <= 5 (n <= 128); 32 lanes for m = 9, 10 (n <= 64)
int dispatch_lbfgsb_fast(mi355_lbfgs_ctx* ctx, int W, int E, int objective, const LbfgsbArgs& args, hipStream_t stream);
int dispatch_lbfgsb_fast_w32(mi355_lbfgs_ctx* ctx, int E, int objective, const LbfgsbArgs& args, hipStream_t stream);
// ridge objective on the matrix cores (ridge_mfma_kernel.hpp): workgroups of sixteen problem slots
int launch_ridge_mfma(mi355_lbfgs_ctx* ctx, SolveArgs args, hipStream_t stream, int lanes, bool fma);

// MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM (dispatch_ridge_gram.hip): Gram matrix, matrix-core pre-pass for c_b = A^T y_b, then
// the ordinary Lbfgs kernel.  `args`: everything but obj_params / per_problem
int ridge_gram_minimize(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, SolveArgs args, const double* y_dev,
                        int y_stride, hipStream_t stream, bool eval_only);
int ridge_gram_launch_wide(mi355_lbfgs_ctx* ctx, int P, int m, const SolveArgs& args, hipStream_t stream);
// MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM (dispatch_ridge_gram_own.hip): one matrix per problem
int ridge_gram_own_minimize(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, SolveArgs args, const double* data_dev,
                            int data_stride, hipStream_t stream, bool eval_only);
int ridge_gram_own_prepass(const double* data, long long data_stride, int rows, int n, int P, double lambda, long long B,
                           double* out, hipStream_t stream);

// MI355_OBJ_AL_COMPOSITE: one Lbfgs solve per row on ToAugmentedLagrangian(problem, (lambda, mu), penalty) (auglag.hip)
int auglag_composite_minimize(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                              double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                              hipStream_t stream);

// User objectives (objective ids >= MI355_OBJ_USER_FIRST): a generated translation unit per lanes-per-problem value
// (cppnumericalsolvers_amd/_build.py, build(user_objectives=...)) instantiates the solve / evaluation kernels for
// the user's device functor and registers its dispatch function here from a static initialiser.
using UserDispatchFn = int (*)(mi355_lbfgs_ctx* ctx, int E, int mr, const SolveArgs& args, hipStream_t stream,
                               bool eval_only);
void register_user_objective(int objective_id, int W, UserDispatchFn fn, const char* name);
struct UserObjectiveRegistration {
  UserObjectiveRegistration(int objective_id, int W, UserDispatchFn fn, const char* name) {
    register_user_objective(objective_id, W, fn, name);
  }
};

// ... and the box-constrained solver on a user objective (registered by the 16-lane unit of the objective)
// (W: lanes per problem, 16 or 32; E: coordinates per lane; args.relaxed selects the relaxed-algebra kernel.  The
// generated unit holds the kernels of the shapes its `lbfgsb` key asked for and refuses the others.)
using UserLbfgsbFn = int (*)(mi355_lbfgs_ctx* ctx, int W, int E, int linesearch, const LbfgsbArgs& args, hipStream_t stream);
void register_user_lbfgsb(int objective_id, UserLbfgsbFn fn);
struct UserLbfgsbRegistration {
  UserLbfgsbRegistration(int objective_id, UserLbfgsbFn fn) { register_user_lbfgsb(objective_id, fn); }
};

// desc->trace (device array pointers) -> the trace fields of SolveArgs; uploads the problem list, zeroes `written`
int setup_trace(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, hipStream_t stream, SolveArgs& args);
// frees what host_pipeline.hip hangs on a context
void destroy_host_pipeline(mi355_lbfgs_ctx* ctx);

// profiling builds (-DMI355_LBFGS_PHASE_TIMING / -DMI355_LBFGSB_PHASE_TIMING): 16 zeroed cycle counters
inline hipError_t profile_counters(mi355_lbfgs_ctx* ctx, hipStream_t stream, unsigned long long** out) {
  if (!ctx->profile_dev) {
    const hipError_t e = hipMalloc(reinterpret_cast<void**>(&ctx->profile_dev), 16 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
  }
  *out = ctx->profile_dev;
  return hipMemsetAsync(ctx->profile_dev, 0, 16 * sizeof(unsigned long long), stream);
}

// A cached device blob (objective parameters, preconditioner, Gram block) is about to be overwritten for a solve on
// `stream`, having last been used by a solve ordered on `previous`: the context's contract is one stream of solves per
// context, but a caller that does move a context to another stream must not have its in-flight solve's parameters
// overwritten — the new stream waits for the last launch recorded on the context (round-4 advisor finding).
inline hipError_t wait_for_last_solve(mi355_lbfgs_ctx* ctx, hipStream_t previous, hipStream_t stream) {
  if (previous == stream || !ctx->timed) return hipSuccess;
  return hipStreamWaitEvent(stream, ctx->ev_stop, 0);
}

#ifdef MI355_DISPATCH_TU  // the launch templates are only needed where kernels are instantiated

template <int W, int E, class Obj, int MR, int LS = MI355_LS_MORE_THUENTE, int ALG = kAlgLbfgs, class OUTER = NoOuterLoop,
          class AR = ArithExact>
int launch_solve(mi355_lbfgs_ctx* ctx, SolveArgs args, hipStream_t stream, const typename OUTER::Args& outer_args = {}) {
  constexpr int kSegs = kWave / W;
  constexpr int kLdsLimit = 160 * 1024;
  constexpr bool kBfgs = (ALG == kAlgBfgs);
  constexpr bool kRegScalars = scalars_in_registers(E, MR, Obj::kLdsDoubles);
  if (args.hess_from_functor && (MR != 0 || kBfgs || !HasHessDiag<Obj>::value))
    return fail(MI355_ERR_UNSUPPORTED,
                "hessian_from_functor: this objective's device functor has no hess_diag (built in: Rosenbrock), or the "
                "shape has no kernel with the history in LDS");
  // condition_hessian of a non-constant Hessian: H, the column buffers and the pivots of every resident problem live behind
  // the wavefronts' ordinary regions (hessian_condition_device.hpp)
  int condition_doubles = 0;
  if (args.hessian_condition_stop > 0.0) {
    if (!(HasHessFull<Obj>::value && MR == 0 && !kBfgs && !OUTER::kEnabled) || args.n > kHessianConditionMaxN)
      return fail(MI355_ERR_UNSUPPORTED,
                  "hessian_condition_stop with hessian_from_functor: Lbfgs at n <= 64 on a functor with a hess_full "
                  "(built in: Rosenbrock)");
    condition_doubles = hessian_condition_lds_doubles(args.n, W);
  }
  const int lds_wave = kSegs * ((kBfgs ? bfgs_lds_doubles_per_problem(W * E, Obj::kLdsDoubles)
                                       : lds_doubles_per_problem(args.m, W * E, MR > 0, Obj::kLdsDoubles, kRegScalars)) +
                                condition_doubles) *
                       static_cast<int>(sizeof(double));
  const int lds_shared = Obj::shared_lds_doubles() * static_cast<int>(sizeof(double));
  // Wavefronts per workgroup: 1, unless the objective keeps read-only data in LDS that the
  // wavefronts of a workgroup share (then as many as fit next to it, at most 8).
  int waves = 1;
  if (lds_shared > 0) {
    waves = (kLdsLimit - lds_shared) / lds_wave;
    if (waves > solve_max_waves<W, E, OUTER, Obj>()) waves = solve_max_waves<W, E, OUTER, Obj>();
  }
  if (ctx->debug_waves >= 1 && ctx->debug_waves < waves) waves = ctx->debug_waves;   // (experiments only, see the context)
  if (waves < 1 || lds_shared + lds_wave > kLdsLimit)
    return fail(MI355_ERR_INVALID_ARGUMENT,
                "history / objective data do not fit LDS: reduce m or lanes_per_problem x elems_per_lane");
  const int lds = lds_shared + waves * lds_wave;
  const long long segs_per_block = static_cast<long long>(kSegs) * waves;
  const long long blocks_needed = (args.B + segs_per_block - 1) / segs_per_block;
  auto kern = lbfgs_solve_kernel<W, E, Obj, MR, LS, ALG, OUTER, AR>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  // Persistent grid: as many workgroups as the chip holds at once (bounded by LDS and
  // VGPRs); the segments pull problems from the queue.
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kWave * waves, lds));
  if (per_cu < 1) per_cu = 1;
  long long blocks_ll = static_cast<long long>(per_cu) * ctx->num_cus;
  if (ctx->debug_blocks >= 1 && ctx->debug_blocks < blocks_ll) blocks_ll = ctx->debug_blocks;
  if (blocks_ll > blocks_needed) blocks_ll = blocks_needed;
  args.next_problem = ctx->queue_dev;
  args.scratch = nullptr;
#ifdef MI355_LBFGS_PHASE_TIMING
  HIP_TRY(profile_counters(ctx, stream, &args.profile));
#endif
  if constexpr (kRegScalars || kBfgs) {
    // plateau rings: MAX_PAST doubles per resident segment
    const size_t need = static_cast<size_t>(blocks_ll) * waves * kSegs * MI355_LBFGS_MAX_PAST;
    if (need > ctx->scratch_cap)  // (sized in mi355_lbfgs_create for the fullest resident grid)
      return fail(MI355_ERR_INVALID_ARGUMENT, "resident grid larger than the context's plateau-ring scratch");
    args.scratch = ctx->scratch_dev;
  }
  HIP_TRY(hipMemsetAsync(ctx->queue_dev, 0, kQueueWords * sizeof(unsigned long long), stream));
  if (!ctx->ev_start_armed) HIP_TRY(hipEventRecord(ctx->ev_start, stream));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWave * waves), lds, stream, args, outer_args);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev_stop, stream));
  ctx->timed = true;
  ctx->last_W = W;
  ctx->last_E = E;
  ctx->last_blocks = static_cast<int>(blocks_ll);
  ctx->last_threads = kWave * waves;
  ctx->last_lds = lds;
  ctx->last_mr = MR;
  ctx->last_arith = AR::kFma ? MI355_ARITH_FMA : MI355_ARITH_EXACT;
  return MI355_OK;
}

// does the objective offer a fused form (eval_fma)?
template <class Obj, class = void>
struct HasFusedEval : std::false_type {};
template <class Obj>
struct HasFusedEval<Obj, std::void_t<decltype(&Obj::template eval_fma<8, 1>)>> : std::true_type {};

// Largest register-history variant (y columns in registers) an objective takes in a mapping.  The reference-order ridge
// objective holds 128 / W residual rows per lane in registers: at four coordinates per lane (and at W = 8 with ten columns)
// its register-history kernels spilled 44-316 B of scratch (profiles/r4_kernel_resources.txt) — those shapes take the
// LDS-ring kernel (MR = 0: same arithmetic, same bits, no scratch) or the six-column variant instead.
template <class Obj, int W, int E>
struct RegisterHistoryMax : std::integral_constant<int, 10> {};
template <int W, int E>
struct RegisterHistoryMax<SquaredErrorRidgeObjective<W, E>, W, E>
    : std::integral_constant<int, (E >= 4) ? 0 : ((W == 8) ? 6 : 10)> {};

// History sizes with a register-resident-y kernel variant (lbfgs_kernel.hpp, MR > 0).
template <int W, int E, class Obj>
int launch_solve_mr(mi355_lbfgs_ctx* ctx, int mr, const SolveArgs& args, hipStream_t stream) {
  static_assert(true, "keep in sync with has_register_history_variant()");
  if (mr >= 0 && (mr & kArithFmaBit)) {  // fused arithmetic: Lbfgs + More-Thuente, objectives with an eval_fma
    mr &= ~kArithFmaBit;
    if constexpr (HasFusedEval<Obj>::value) {
      using NO = NoOuterLoop;
      constexpr int MT = MI355_LS_MORE_THUENTE;
      if constexpr (E >= 2) {
        if (mr >= 1 && mr <= 5) return launch_solve<W, E, Obj, 5, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
        if (mr == 6) return launch_solve<W, E, Obj, 6, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
        if (mr >= 7 && mr <= 10) return launch_solve<W, E, Obj, 10, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
      }
      return launch_solve<W, E, Obj, 0, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
    }
  }
  if (mr <= kMrHzFusedBase) {  // Lbfgs<F, m, HagerZhang>, fused arithmetic, register history for m <= 10 (six / ten columns)
    const int m_hist = kMrHzFusedBase - mr;
    if constexpr (HasFusedEval<Obj>::value) {
      using NO = NoOuterLoop;
      constexpr int HZ = MI355_LS_HAGER_ZHANG;
      // register history at TWO coordinates per lane only: the search's segment-uniform state (nine samples of the
      // bracketing state machine) next to four coordinates' worth of y columns spills 36-104 VGPRs and runs 1.5x SLOWER
      // than the LDS ring (profiles/r6_ab_hz.txt), so E = 4 keeps the ring and the entry point maps Hager-Zhang solves
      // onto two coordinates per lane (mi355_lbfgs.hip, hz_fused_mapping)
      if constexpr (E == 2) {
        if (m_hist >= 1 && m_hist <= 6) return launch_solve<W, E, Obj, 6, HZ, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
        if (m_hist >= 7 && m_hist <= 10) return launch_solve<W, E, Obj, 10, HZ, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
      }
      return launch_solve<W, E, Obj, 0, HZ, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
    }
  }
  // mr < 0 selects the other solver variants: -1 Lbfgs with the Hager-Zhang line search (LDS-ring history
  // only), -2 / -3 dense BFGS with the More-Thuente / Hager-Zhang line search
  if (mr == -2 || mr == -3) {
    // H is (W*E)^2 doubles per problem in LDS: built for the padded widths 8, 16, 32, 64 (n <= 64) — the packed mappings
    // of the Lbfgs kernels and, since round 6, the WIDE ones the entry point picks for 32 and 64 (one or two coordinates
    // per lane): H's footprint, not the register file, caps the problems in flight per CU, so a problem's O(n^2) work is
    // spread over as many lanes as it has columns (profiles/r6_ab_bfgs_mapping.txt)
    if constexpr (Obj::shared_lds_doubles() == 0 &&
                  ((W == 8 && (E == 1 || E == 2 || E == 4)) || (W == 16 && (E == 2 || E == 4)) ||
                   (W == 32 && (E == 1 || E == 2)) || (W == 64 && E == 1))) {
      return mr == -2 ? launch_solve<W, E, Obj, 0, MI355_LS_MORE_THUENTE, kAlgBfgs>(ctx, args, stream)
                      : launch_solve<W, E, Obj, 0, MI355_LS_HAGER_ZHANG, kAlgBfgs>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "dense BFGS is built for n <= 64 with the Rosenbrock / DiagQuadratic objectives");
    }
  }
  if (mr < 0) return launch_solve<W, E, Obj, 0, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
  if constexpr (E >= 2) {  // the packed mappings are the LDS-capacity-bound ones
    // the smallest built register-history size that holds m pairs (m = 7..9 -> 10, m <= 4 -> 5)
    constexpr int kMax = RegisterHistoryMax<Obj, W, E>::value;
    if constexpr (kMax >= 5)
      if (mr >= 1 && mr <= 5) return launch_solve<W, E, Obj, 5>(ctx, args, stream);
    if constexpr (kMax >= 6)
      if (mr == 6) return launch_solve<W, E, Obj, 6>(ctx, args, stream);
    if constexpr (kMax >= 10)
      if (mr >= 7 && mr <= 10) return launch_solve<W, E, Obj, 10>(ctx, args, stream);
  }
  return launch_solve<W, E, Obj, 0>(ctx, args, stream);
}

// Rosenbrock problems that fill their segment (n == W * E) on the production path — Lbfgs, More-Thuente, y history
// in registers, m = 6..10: the objective variant whose boundary predicates are compile-time constants
// (objectives.hpp; bit-identical, ~3 % fewer instructions per iteration).
template <int W, int E>
int launch_solve_rosenbrock_full(mi355_lbfgs_ctx* ctx, int mr, const SolveArgs& args, hipStream_t stream) {
  using NO = NoOuterLoop;
  using Obj = RosenbrockFullObjective;
  constexpr int MT = MI355_LS_MORE_THUENTE;
  const bool fma = (mr & kArithFmaBit) != 0;
  mr &= ~kArithFmaBit;
  if (fma)
    return mr == 6 ? launch_solve<W, E, Obj, 6, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream)
                   : launch_solve<W, E, Obj, 10, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
  return mr == 6 ? launch_solve<W, E, Obj, 6>(ctx, args, stream) : launch_solve<W, E, Obj, 10>(ctx, args, stream);
}

// Solve kernels of a user objective: Lbfgs with the More-Thuente line search — y history in registers for m <= 10
// when a lane holds at least two coordinates (6- and 10-column variants), LDS ring otherwise; the fused arithmetic
// when the functor defines eval_fma — and, like the built-in objectives, Lbfgs with the Hager-Zhang line search and
// dense Bfgs with either line search.
template <int W, int E, class Obj>
int launch_solve_user(mi355_lbfgs_ctx* ctx, int mr, const SolveArgs& args, hipStream_t stream) {
  // mr < 0: -1 Lbfgs with the Hager-Zhang line search (LDS-ring history), -2 / -3 dense BFGS with the More-Thuente /
  // Hager-Zhang line search (n <= 64, as for the built-in objectives)
  if (mr == -2 || mr == -3) {
    if constexpr (Obj::shared_lds_doubles() == 0 &&
                  ((W == 8 && (E == 1 || E == 2 || E == 4)) || (W == 16 && E == 4))) {
      return mr == -2 ? launch_solve<W, E, Obj, 0, MI355_LS_MORE_THUENTE, kAlgBfgs>(ctx, args, stream)
                      : launch_solve<W, E, Obj, 0, MI355_LS_HAGER_ZHANG, kAlgBfgs>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "dense BFGS is built for n <= 64 on objectives without workgroup-shared LDS data");
    }
  }
  if (mr < 0) return launch_solve<W, E, Obj, 0, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
  using NO = NoOuterLoop;
  constexpr int MT = MI355_LS_MORE_THUENTE;
  const bool fma = (mr & kArithFmaBit) != 0;
  mr &= ~kArithFmaBit;
  if (fma) {
    if constexpr (HasFusedEval<Obj>::value) {
      if constexpr (E >= 2) {
        if (mr >= 1 && mr <= 6) return launch_solve<W, E, Obj, 6, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
        if (mr >= 7 && mr <= 10) return launch_solve<W, E, Obj, 10, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
      }
      return launch_solve<W, E, Obj, 0, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
    }
  }
  if constexpr (E >= 2) {
    if (mr >= 1 && mr <= 6) return launch_solve<W, E, Obj, 6>(ctx, args, stream);
    if (mr >= 7 && mr <= 10) return launch_solve<W, E, Obj, 10>(ctx, args, stream);
  }
  return launch_solve<W, E, Obj, 0>(ctx, args, stream);
}

template <int W, int E, class Obj, bool HZ_SEARCH = false, class AR = ArithExact>
int launch_eval(const SolveArgs& args, hipStream_t stream) {
  constexpr int kSegs = kWave / W;
  const long long blocks_ll = (args.B + kSegs - 1) / kSegs;
  const int lds = (Obj::shared_lds_doubles() + kSegs * (Obj::kLdsDoubles > 0 ? Obj::kLdsDoubles : 1)) *
                  static_cast<int>(sizeof(double));
  if (lds > 160 * 1024)
    return fail(MI355_ERR_INVALID_ARGUMENT, "objective data does not fit LDS with this lanes_per_problem x elems_per_lane");
  auto kern = HZ_SEARCH ? hz_search_kernel<W, E, Obj> : eval_kernel<W, E, Obj, AR>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWave), lds, stream, args);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

// eval_only: false = solve, true = one-shot kernel: objective evaluation, or (args.ls_direction set)
// one Hager-Zhang search per problem
template <int W, int E, class Obj>
int launch_oneshot(const SolveArgs& args, hipStream_t stream, int mr) {
  if (mr & kArithFmaBit) {  // one evaluation under the fused arithmetic policy
    if constexpr (HasFusedEval<Obj>::value) {
      return launch_eval<W, E, Obj, false, ArithFma>(args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
    }
  }
  return args.ls_direction ? launch_eval<W, E, Obj, true>(args, stream) : launch_eval<W, E, Obj, false>(args, stream);
}

template <int W, int E>
int dispatch_objective(mi355_lbfgs_ctx* ctx, int objective, int mr, const SolveArgs& args,
                       hipStream_t stream, bool eval_only) {
  switch (objective) {
    case MI355_OBJ_ROSENBROCK:
      if constexpr (E == 4 && (W == 8 || W == 16 || W == 32)) {
        if (!eval_only && args.n == W * E && mr >= 0 && (mr & ~kArithFmaBit) >= 6 && (mr & ~kArithFmaBit) <= 10)
          return launch_solve_rosenbrock_full<W, E>(ctx, mr, args, stream);
      }
      if (!eval_only && args.hess_from_functor && args.hessian_condition_stop > 0.0) {
        // the condition_hessian test of a non-constant Hessian: the kernels that carry the LU (RosenbrockConditionObjective)
        if constexpr (E <= 2) {
          if (mr == 0) return launch_solve<W, E, RosenbrockConditionObjective, 0>(ctx, args, stream);
          if (mr == kArithFmaBit)
            return launch_solve<W, E, RosenbrockConditionObjective, 0, MI355_LS_MORE_THUENTE, kAlgLbfgs, NoOuterLoop, ArithFma>(
                ctx, args, stream);
          if (mr == -1) return launch_solve<W, E, RosenbrockConditionObjective, 0, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
        }
        return fail(MI355_ERR_UNSUPPORTED,
                    "hessian_condition_stop with hessian_from_functor: Lbfgs (either line search), one or two coordinates "
                    "per lane");
      }
      return eval_only ? launch_oneshot<W, E, RosenbrockObjective>(args, stream, mr)
                       : launch_solve_mr<W, E, RosenbrockObjective>(ctx, mr, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC:
      return eval_only ? launch_oneshot<W, E, DiagQuadraticObjective<E>>(args, stream, mr)
                       : launch_solve_mr<W, E, DiagQuadraticObjective<E>>(ctx, mr, args, stream);
    case MI355_OBJ_SQUARED_ERROR_RIDGE:
      return eval_only ? launch_oneshot<W, E, SquaredErrorRidgeObjective<W, E>>(args, stream, mr)
                       : launch_solve_mr<W, E, SquaredErrorRidgeObjective<W, E>>(ctx, mr, args, stream);
    default:
      return fail(MI355_ERR_UNSUPPORTED, "unknown objective id");
  }
}

template <int W, class Obj1, class Obj2, class Obj4>
int dispatch_user(mi355_lbfgs_ctx* ctx, int E, int mr, const SolveArgs& args, hipStream_t stream, bool eval_only) {
  if (eval_only && args.ls_direction) return fail(MI355_ERR_UNSUPPORTED, "user objectives have no stand-alone line search entry");
  switch (E) {
    case 1: return eval_only ? launch_oneshot<W, 1, Obj1>(args, stream, mr) : launch_solve_user<W, 1, Obj1>(ctx, mr, args, stream);
    case 2: return eval_only ? launch_oneshot<W, 2, Obj2>(args, stream, mr) : launch_solve_user<W, 2, Obj2>(ctx, mr, args, stream);
    case 4: return eval_only ? launch_oneshot<W, 4, Obj4>(args, stream, mr) : launch_solve_user<W, 4, Obj4>(ctx, mr, args, stream);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "elems_per_lane must be 1, 2 or 4");
}

// Eight coordinates per lane (W = 4: n <= 32, W = 8: n <= 64): twice the problems per wavefront at one wavefront per
// SIMD (the kernel needs ~330 registers).  Built for the objectives without LDS data, More-Thuente, y history in
// registers (m <= 10).
template <int W, class Obj>
int launch_solve_e8(mi355_lbfgs_ctx* ctx, int mr, const SolveArgs& args, hipStream_t stream) {
  if (mr < 0) return fail(MI355_ERR_UNSUPPORTED, "eight coordinates per lane: Lbfgs with the More-Thuente line search only");
  const bool fma = (mr & kArithFmaBit) != 0;
  mr &= ~kArithFmaBit;
  using NO = NoOuterLoop;
  constexpr int MT = MI355_LS_MORE_THUENTE;
  if (mr < 1 || mr > 10)
    return fail(MI355_ERR_UNSUPPORTED, "eight coordinates per lane: y history in registers, m <= 10");
  if (fma) {
    if (mr <= 6) return launch_solve<W, 8, Obj, 6, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
    return launch_solve<W, 8, Obj, 10, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
  }
  if (mr <= 6) return launch_solve<W, 8, Obj, 6>(ctx, args, stream);
  return launch_solve<W, 8, Obj, 10>(ctx, args, stream);
}
template <int W>
int dispatch_e8(mi355_lbfgs_ctx* ctx, int objective, int mr, const SolveArgs& args, hipStream_t stream, bool eval_only) {
  switch (objective) {
    case MI355_OBJ_ROSENBROCK:
      return eval_only ? launch_oneshot<W, 8, RosenbrockObjective>(args, stream, mr)
                       : launch_solve_e8<W, RosenbrockObjective>(ctx, mr, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC:
      return eval_only ? launch_oneshot<W, 8, DiagQuadraticObjective<8>>(args, stream, mr)
                       : launch_solve_e8<W, DiagQuadraticObjective<8>>(ctx, mr, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "eight coordinates per lane are built for the Rosenbrock and DiagQuadratic objectives");
}

template <int W>
int dispatch_e(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args,
               hipStream_t stream, bool eval_only) {
  if constexpr (W == 8) {
    if (E == 8) return dispatch_e8<8>(ctx, objective, mr, args, stream, eval_only);
  }
  switch (E) {
    case 1: return dispatch_objective<W, 1>(ctx, objective, mr, args, stream, eval_only);
    case 2: return dispatch_objective<W, 2>(ctx, objective, mr, args, stream, eval_only);
    case 4: return dispatch_objective<W, 4>(ctx, objective, mr, args, stream, eval_only);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "elems_per_lane must be 1, 2 or 4");
}


template <int E, class Obj, int M, int LS = MI355_LS_MORE_THUENTE, class OUTER = NoOuterLoop, int W = 16>
int launch_lbfgsb(mi355_lbfgs_ctx* ctx, LbfgsbArgs args, hipStream_t stream, const typename OUTER::Args& outer_args = {}) {
  constexpr int kSegs = kWave / W;
  const int lds = (Obj::shared_lds_doubles() + kSegs * lbfgsb_lds_doubles_per_problem<M>(W * E, Obj::kLdsDoubles)) *
                  static_cast<int>(sizeof(double));
  auto kern = lbfgsb_solve_kernel<E, Obj, M, LS, OUTER, W>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kWave, lds));
  if (per_cu < 1) per_cu = 1;
  const long long blocks_needed = (args.s.B + kSegs - 1) / kSegs;
  long long blocks_ll = static_cast<long long>(per_cu) * ctx->num_cus;
  if (blocks_ll > blocks_needed) blocks_ll = blocks_needed;
  args.s.next_problem = ctx->queue_dev;
#ifdef MI355_LBFGSB_PHASE_TIMING
  HIP_TRY(profile_counters(ctx, stream, &args.s.profile));
#endif
  HIP_TRY(hipMemsetAsync(ctx->queue_dev, 0, kQueueWords * sizeof(unsigned long long), stream));
  HIP_TRY(hipEventRecord(ctx->ev_start, stream));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWave), lds, stream, args, outer_args);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev_stop, stream));
  ctx->timed = true;
  ctx->last_W = W;
  ctx->last_E = E;
  ctx->last_blocks = static_cast<int>(blocks_ll);
  ctx->last_threads = kWave;
  ctx->last_lds = lds;
  ctx->last_mr = 0;
  ctx->last_arith = MI355_ARITH_EXACT;
  return MI355_OK;
}

// History sizes: the kernel is built for M = 5 columns (the reference default, lbfgsb.h:44; serves m <= 5) and for
// M = 8 (m = 6..8: 2M = 16 rows of the compact representation, one per lane of the segment's DPP row — the widest
// the row-per-lane algebra goes).  The Hager-Zhang variants and the ridge objective are built for M = 5.
template <int E, class Obj>
int dispatch_lbfgsb_m(mi355_lbfgs_ctx* ctx, int linesearch, const LbfgsbArgs& args, hipStream_t stream) {
  if (args.s.m > 5) {
    if (linesearch == MI355_LS_HAGER_ZHANG)
      return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with the Hager-Zhang line search is built for m <= 5");
    if constexpr (Obj::shared_lds_doubles() == 0) {
      return launch_lbfgsb<E, Obj, 8>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B on this objective is built for m <= 5");
    }
  }
  if (linesearch == MI355_LS_HAGER_ZHANG) {
    if constexpr (Obj::shared_lds_doubles() == 0) {
      return launch_lbfgsb<E, Obj, 5, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
    } else {
      return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B on this objective is built with the More-Thuente line search");
    }
  }
  return launch_lbfgsb<E, Obj, 5>(ctx, args, stream);
}

// 64 < n <= 128: eight coordinates per lane of the 16-lane segment (one wavefront per SIMD: the kernel needs more than
// 256 registers).  Rosenbrock / DiagQuadratic, More-Thuente, m <= 5.
#ifdef MI355_DISPATCH_LBFGSB_TU  // (a plain function: its kernels would be instantiated by every unit that sees it)
inline int dispatch_lbfgsb_wide(mi355_lbfgs_ctx* ctx, int objective, int linesearch, const LbfgsbArgs& args,
                                hipStream_t stream) {
  if (linesearch != MI355_LS_MORE_THUENTE || args.s.m > 5)
    return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B for 64 < n <= 128 is built for m <= 5 with the More-Thuente line search");
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return launch_lbfgsb<8, RosenbrockObjective, 5>(ctx, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC: return launch_lbfgsb<8, DiagQuadraticObjective<8>, 5>(ctx, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B for 64 < n <= 128 is built for the Rosenbrock and DiagQuadratic objectives");
}
#endif

// m = 9, 10: 2M = 20 rows of the compact representation do not fit a DPP row, so a problem takes 32 lanes (two rows;
// two problems per wavefront, one wavefront per SIMD).  Rosenbrock / DiagQuadratic, More-Thuente, n <= 64.
// The same width with eight coordinates per lane serves 128 < n <= 256 (m <= 5).
// Declared above; defined in dispatch_lbfgsb_w32.hip.
#ifdef MI355_DISPATCH_LBFGSB_W32_TU
int dispatch_lbfgsb_w32(mi355_lbfgs_ctx* ctx, int objective, int linesearch, const LbfgsbArgs& args, hipStream_t stream) {
  if (linesearch != MI355_LS_MORE_THUENTE)
    return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with 32 lanes per problem is built with the More-Thuente line search");
  if (args.s.m <= 5 && args.s.n > 128 && args.s.n <= 256) {  // 128 < n <= 256: eight coordinates per lane, m <= 5
    switch (objective) {
      case MI355_OBJ_ROSENBROCK:
        return launch_lbfgsb<8, RosenbrockObjective, 5, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream);
      case MI355_OBJ_DIAG_QUADRATIC:
        return launch_lbfgsb<8, DiagQuadraticObjective<8>, 5, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream);
    }
    return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B for n > 128 is built for the Rosenbrock and DiagQuadratic objectives");
  }
  if (args.s.n > 64)
    return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B for m = 9, 10 is built for n <= 64");
  const bool one = args.s.n <= 32;
  switch (objective) {
    case MI355_OBJ_ROSENBROCK:
      return one ? launch_lbfgsb<1, RosenbrockObjective, 10, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream)
                 : launch_lbfgsb<2, RosenbrockObjective, 10, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC:
      return one ? launch_lbfgsb<1, DiagQuadraticObjective<1>, 10, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream)
                 : launch_lbfgsb<2, DiagQuadraticObjective<2>, 10, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B for m = 9, 10 is built for the Rosenbrock and DiagQuadratic objectives");
}
#endif

template <int E>
int dispatch_lbfgsb(mi355_lbfgs_ctx* ctx, int objective, int linesearch, const LbfgsbArgs& args, hipStream_t stream) {
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return dispatch_lbfgsb_m<E, RosenbrockObjective>(ctx, linesearch, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC: return dispatch_lbfgsb_m<E, DiagQuadraticObjective<E>>(ctx, linesearch, args, stream);
    // Lbfgsb on a regression objective (src/examples/linear_regression.cc:58-74): the ridge functor with its matrix
    // in workgroup-shared LDS, one wavefront (four problems) per workgroup
    case MI355_OBJ_SQUARED_ERROR_RIDGE:
      return dispatch_lbfgsb_m<E, SquaredErrorRidgeObjective<16, E>>(ctx, linesearch, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B is built for the Rosenbrock, DiagQuadratic and SquaredErrorRidge objectives");
}

#endif  // MI355_DISPATCH_TU

}  // namespace mi355
