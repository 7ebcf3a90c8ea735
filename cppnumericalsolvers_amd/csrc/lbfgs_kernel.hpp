// lbfgs_kernel.hpp — the whole L-BFGS solve of one problem on one wavefront
// segment, fused into a single kernel (no per-iteration launches).
//
// Device counterpart of
//   Solver::Minimize            solver/solver.h:181-224   (driver loop)
//   Lbfgs::InitializeSolver     solver/lbfgs.h:72-87
//   Lbfgs::OptimizationStep     solver/lbfgs.h:89-303     (two-loop, s/y ring, gamma)
//   Progress::Update            solver/progress.h:153-327 (stopping tests)
// with MoreThuente from more_thuente_device.hpp.
//
// Mapping.  A problem of dimension n <= W*E is owned by a segment of W
// consecutive lanes; lane `sl` keeps coordinates j = sl*E+e (e < E) of x, g,
// d, ... in registers.  The s half of the (s, y) history ring lives in LDS as
// S[slot][sl][e] — every lane only ever touches its own column, so the accesses
// need no barrier.  They are NOT bank-conflict-free at E = 4: a lane's column is
// 32 bytes and is moved as ds_read_b128 / ds_write_b128 pairs, i.e. 16-byte
// accesses at a 32-byte lane stride — the 16 lanes a b128 pass serves cover the
// 64 four-byte banks twice, a 2-way conflict (SQ_LDS_BANK_CONFLICT = 5.5 % of
// SQ_WAVE_CYCLES on configs[1], 7.4 % on configs[2], 4 x SQ_ACTIVE_INST_LDS in
// both: profiles/r5_pmc.txt).  A swizzled layout that removes it measured
// 0 ... -2 % (profiles/r2_ab_fma_swizzle.txt: the kernels are VALU-issue bound and
// the extra address arithmetic costs what the conflict did), so it stays.  The y
// half sits in registers for the built history sizes (MR > 0, chronological,
// shifted on every accepted pair) and in LDS next to S otherwise (MR = 0).
// 1/(s.y) and the alpha_i of the two-loop recursion are register arrays where
// the register budget allows (scalars_in_registers()), per-segment LDS scalars
// otherwise.  x0 is read from and x*, g*, f*, progress are written to
// batch-major HBM arrays exactly once per solve; nothing else touches HBM
// (the plateau ring of stop.past > 0 is an L2-resident scratch slot).
//
// A workgroup is one wavefront (64/W segments), or several independent
// wavefronts when the objective shares read-only data in LDS.  The grid is
// sized to what the chip can hold (persistent wavefronts); every segment pulls
// the index of an unsolved problem from a global atomic counter, solves it,
// writes the result and pulls the next one, re-initialising in place while the
// other segments of its wavefront keep iterating.  That absorbs the spread of
// iteration counts between problems both across wavefronts and between the
// segments of one (DESIGN.md section 3.8 for what it cannot absorb).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "../../include/mi355_lbfgs.h"
#include "hager_zhang_device.hpp"
#include "more_thuente_device.hpp"
#include "objectives.hpp"
#include "wave_primitives.hpp"
#include "hessian_condition_device.hpp"

namespace mi355 {

// Profiling build (-DMI355_LBFGS_PHASE_TIMING): wavefronts accumulate s_memtime deltas per phase of
// the iteration into args.profile[0..7]; scripts/lbfgs_phases.py prints the table.
#ifdef MI355_LBFGS_PHASE_TIMING
#define MI355_LPHASE(i)                                             \
  do {                                                              \
    const unsigned long long now_ = __builtin_readcyclecounter();   \
    lphase_cycles[lphase_cur] += now_ - lphase_t0;                  \
    lphase_t0 = now_;                                               \
    lphase_cur = (i);                                               \
  } while (0)
#else
#define MI355_LPHASE(i) do { } while (0)
#endif

struct TraceArgs {                    // device-resident description of an active trace
  long long problems[MI355_LBFGS_MAX_TRACED];  // problem indices
  mi355_lbfgs_trace_record* records;  // [count][capacity]
  double* x;                          // [count][capacity][n] or null
  double* g;
  unsigned* written;                  // [count]
  int count;
  int capacity;
};

// Does the objective functor offer diag H(x)?  (optional member: template <int W, int E> void hess_diag(x, h, n, sl))
template <class Obj, class = void>
struct HasHessDiag : std::false_type {};
template <class Obj>
struct HasHessDiag<Obj, std::void_t<decltype(&Obj::template hess_diag<8, 1>)>> : std::true_type {};

// ... and the full Hessian?  (optional member: template <int W, int E> void hess_full(x, Hm, n, sl), column major in LDS)
template <class Obj, class = void>
struct HasHessFull : std::false_type {};
template <class Obj>
struct HasHessFull<Obj, std::void_t<decltype(&Obj::template hess_full<8, 1>)>> : std::true_type {};

struct SolveArgs {
  const double* x0;
  double* x_out;
  double* f_out;
  double* g_out;                      // may be null
  mi355_lbfgs_progress* progress_out; // may be null
  const double* obj_params;           // device
  const double* per_problem;          // device, [B][per_problem_stride] (objective specific, may be null)
  int per_problem_stride;
  const unsigned int* count_dev;      // device or null: the number of queue positions, read at run time instead of B
                                      // (B then only sizes the grid) — lets a chain of launches follow a count that
                                      // an earlier kernel of the chain produced, without a host round trip
  const int* problem_map;             // device, [B] or null: queue position q works on problem problem_map[q]
                                      // (the augmented-Lagrangian loop solves the compacted list of problems that
                                      // are still active; the other rows of every array are left untouched)
  // Second-mode functions (lbfgs.h:116-139): device pointer to n doubles 1/(|H_jj| + eps), the
  // constant diagonal preconditioner that replaces scaling_factor_ at :177-181; null = First mode.
  const double* precond;
  // Second-mode functions whose Hessian is NOT constant (round 3): 1 = the preconditioner is rebuilt at every iterate
  // from the objective functor's hess_diag (diag H at the current x), as lbfgs.h:129-138 re-evaluates
  // function(x, &g, &H); precond is null then.  Kernels with the history in LDS (MR == 0) only.
  int hess_from_functor;
  double* scratch;                    // global scratch: plateau rings of the scalars_in_registers() kernels
  // stand-alone line search (hz_search_kernel): direction s[B][n], initial steps, outputs
  const double* ls_direction;
  const double* ls_alpha_init;
  double* ls_alpha_out;
  unsigned* ls_nfev_out;              // may be null
  // opt-in per-iteration trace of chosen problems (mi355_lbfgs_trace): null when off.  Behind ONE pointer (a record in
  // device memory) rather than seven kernel arguments: the solve kernels are short of scalar registers, and every
  // argument that stays live across the iteration loop is spilled to vector-register lanes and read back
  const struct TraceArgs* trace;
  // Second-mode condition_hessian stopping test (progress.h:318-325): the Hessian is constant, so whether the test
  // fires is decided on the host; it is the LAST test of Progress::Update
  int hessian_condition_fires;
  // ... and with hess_from_functor the Hessian changes with x: > 0 = the threshold itself; the kernel evaluates
  // ||H(x)|| ||H(x)^-1|| after every iteration (hessian_condition_device.hpp; n <= 64, functors with a hess_full)
  double hessian_condition_stop;
  unsigned long long* profile;        // 16 cycle counters (profiling builds only, else null)
  unsigned long long* next_problem;   // device work-queue head, zeroed before every launch
  long long B;
  int n;
  int m;
  mi355_lbfgs_stop stop;
};

// LDS scalars written by lane 0 of a segment are read by its other lanes.  All
// lanes belong to one wavefront and DS operations of a wavefront execute in
// program order, so no hardware barrier is needed; this only stops the compiler
// from moving LDS accesses across the hand-off.
__device__ __forceinline__ void segment_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// rho and the alpha of the running two-loop recursion in registers instead of LDS: 4 MR VGPRs, taken
// where that does not push the kernel over an occupancy step: not for two elements per lane at m = 10
// (157 -> 190 would lose the third wavefront per SIMD); the four-elements-per-lane kernels and the ridge
// kernels (objective_scratch > 0) are above 168 either way (m = 10, E = 4: 246 of 256, no spills).
// These kernels also keep the plateau ring of the stopping test (progress.h:139-140, only touched
// when stop.past > 0) in global scratch, so that the s ring is ALL a problem holds in LDS — n = 64,
// m = 10: 4 x 5120 B per wavefront, eight wavefronts per CU.
__host__ __device__ constexpr bool scalars_in_registers(int E, int MR, int objective_scratch) {
  return MR > 0 && (E >= 4 || MR <= 6 || objective_scratch > 0);
}

// Doubles of LDS one problem needs.  y_in_registers: only the S half of the ring is in LDS.
__host__ __device__ inline int lds_doubles_per_problem(int m, int WE, bool y_in_registers,
                                                       int objective_scratch, bool scalars_in_regs) {
  return (y_in_registers ? 1 : 2) * m * WE + (scalars_in_regs ? 0 : 2 * m + MI355_LBFGS_MAX_PAST) +
         objective_scratch;
}

// Trace hook, called by the solve kernels after every Progress::Update (what step_callback_ of the reference sees
// before the next step, solver/solver.h:197, and after the loop, :222).  When tracing is off this is one uniform
// branch on a kernel argument.  Whether `prob` is traced is looked up again every iteration (a scan of at most
// MI355_LBFGS_MAX_TRACED indices) instead of being carried in a register: the packed kernels have none to spare, and
// a traced launch is a debugging run.
template <int E>
__device__ __forceinline__ void trace_iteration(const SolveArgs& a, long long prob, int n, int sl, unsigned num_iterations,
                                                int status, double f, double x_delta, double f_delta, double gradient_norm,
                                                const double (&x)[E], const double (&g)[E]) {
  if (a.trace == nullptr) return;
  const TraceArgs& t = *a.trace;
  int slot = -1;
  for (int i = 0; i < t.count; ++i)
    if (t.problems[i] == prob) slot = i;
  if (slot < 0) return;
  const size_t rec = static_cast<size_t>(slot) * t.capacity + (num_iterations - 1u) % static_cast<unsigned>(t.capacity);
  if (sl == 0) {
    mi355_lbfgs_trace_record r;
    r.num_iterations = num_iterations;
    r.status = status;
    r.value = f;
    r.x_delta = x_delta;
    r.f_delta = f_delta;
    r.gradient_norm = gradient_norm;
    t.records[rec] = r;
    t.written[slot] = num_iterations;
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    if (j < n) {
      if (t.x) t.x[rec * n + j] = x[e];
      if (t.g) t.g[rec * n + j] = g[e];
    }
  }
}

// The kernel's own by-value SolveArgs, re-read from the kernarg segment at the point of use.  The solve kernels are
// short of scalar registers: every kernel argument that stays live across the iteration loop is spilled to lanes of
// a vector register and read back with v_readlane (a VALU slot plus hazard wait states).  The pointers a problem needs
// once — its start point and per-problem data at the fetch, the result arrays at the end — are therefore loaded
// where they are used (scalar loads, scalar-cache hits) instead of being carried.  The empty asm makes the base
// address opaque so that the loads are not hoisted back out of the cold blocks.
typedef const __attribute__((address_space(4))) SolveArgs* KernargSolveArgs;
__device__ __forceinline__ KernargSolveArgs cold_args() {
  unsigned long long p = reinterpret_cast<unsigned long long>(__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(p));
  return reinterpret_cast<KernargSolveArgs>(p);
}

constexpr int kAlgLbfgs = 0, kAlgBfgs = 1;
// LDS doubles of one dense-BFGS problem: H and three staging vectors (matvec input, s, H y)
__host__ __device__ constexpr int bfgs_lds_doubles_per_problem(int WE, int objective_scratch) {
  return WE * WE + 3 * WE + objective_scratch;
}

// MR = 0: both halves of the (s, y) ring in LDS, any history size m (runtime).
// MR > 0: requires m <= MR.  The y half lives in registers, in chronological order
//   (newest at index MR-1, shifted on every accepted pair) so that the fully unrolled
//   two-loop recursion indexes it statically; only the s half stays in LDS.  LDS is what
//   caps the number of problems in flight per CU (160 KiB / ring size), so halving the
//   footprint doubles the wavefronts per SIMD for the packed mappings.
// A workgroup is `blockDim.x / 64` independent wavefronts.  They only share the objective's
// read-only LDS region (Obj::shared_lds_doubles(), e.g. the ridge objective's matrix A), filled
// cooperatively before the first problem is pulled; after that single barrier the wavefronts
// never synchronise again.  Objectives without shared data run one wavefront per workgroup.
// LS: the LineSearch template argument of the reference's Lbfgs (lbfgs.h:41): MI355_LS_MORE_THUENTE or
// MI355_LS_HAGER_ZHANG.
// ALG: kAlgLbfgs (solver/lbfgs.h) or kAlgBfgs — dense BFGS (solver/bfgs.h:65-137): the same driver, line
// searches and stopping tests around an explicit inverse-Hessian approximation H (WE x WE doubles in
// LDS per problem; MR must be 0).  H stays bitwise symmetric under the update (:128-130) — both cross
// terms s_i Hy_j + Hy_i s_j are the same two products — so every lane reads and writes "its" rows
// through the columns H[j][i], i = its own coordinates: consecutive lanes touch consecutive doubles.
// H's LDS footprint, not the register file, caps the problems in flight per CU, so the entry point spreads a
// problem of padded width 32 / 64 over 32 / 64 lanes (E = 1): 1.8x / 1.5x the throughput of the packed 8 x 4 /
// 16 x 4 mappings, same bits (profiles/r6_ab_bfgs_mapping.txt).  Tried there and not kept: the lane's column of
// H in registers instead of LDS (64 / 128 VGPRs) — slower at 32 (27.4 vs 21.7 ms: two wavefronts per SIMD and a
// guarded, fully unrolled j loop), +15 % at 64 but with 179 spilled VGPRs.
// OUTER: NoOuterLoop, or a policy that turns every queue entry into a LOOP of solves (the augmented-Lagrangian outer
// iteration, csrc/auglag_device.hpp): `begin` runs when a problem is fetched, `step` when a solve stops and either
// asks for the next solve from the current point (returns true; it may retarget the two stopping fields that differ
// between solves) or finishes the problem after writing its own results.
struct NoOuterLoop {
  static constexpr bool kEnabled = false;
  struct Args {};
};

// Wavefronts a workgroup of the solve kernel may hold (its launch bound; launch_solve stays inside it): eight, one for the
// wide-lane kernels (E = 8), and four for a fused outer loop at one problem per wavefront with four coordinates per lane
// (the augmented-Lagrangian kernels of 128 < n <= 256).  Those hold 40-56 KB of LDS per wavefront — LDS, not registers,
// already limits them to one wavefront per SIMD — so the smaller bound costs no occupancy and hands the allocator the
// whole 512-entry register file instead of 56-288 B of scratch (profiles/r5_ab_spills.txt).
// (an objective says `static constexpr bool kLargeFootprint = true` when that holds for its plain solves too: the
// augmented-Lagrangian composite, whose lock-step inner solves run without an outer loop)
template <class Obj, class = void>
struct LargeFootprint : std::false_type {};
template <class Obj>
struct LargeFootprint<Obj, std::void_t<decltype(Obj::kLargeFootprint)>> : std::integral_constant<bool, Obj::kLargeFootprint> {};
template <int W, int E, class OUTER, class Obj>
__host__ __device__ constexpr int solve_max_waves() {
  return (E >= 8) ? 1 : (((OUTER::kEnabled || LargeFootprint<Obj>::value) && W == 64 && E == 4) ? 4 : 8);
}

// AR: arithmetic policy (wave_primitives.hpp): ArithExact, or ArithFma (Lbfgs with either line search; round 6: Hager-Zhang too).
template <int W, int E, class Obj, int MR, int LS = MI355_LS_MORE_THUENTE, int ALG = 0, class OUTER = NoOuterLoop,
          class AR = ArithExact>
// (Forcing 3 waves/SIMD on the E = 4, MR = 6 variant via launch bounds costs 48 B/lane of scratch
// and 15 % of throughput — measured — so the allocator is left alone.)
__global__ __launch_bounds__((64 * solve_max_waves<W, E, OUTER, Obj>())) void lbfgs_solve_kernel(const SolveArgs a, const typename OUTER::Args oa) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int WE = W * E;
  constexpr int kSegs = kWave / W;
  constexpr double eps = 2.220446049250313e-16;  // numeric_limits<double>::epsilon()

  const int lane = threadIdx.x & (kWave - 1);
  const int wave_in_block = threadIdx.x / kWave;
  const int seg = lane / W;
  const int sl = lane % W;
  long long prob = 0;
  bool need_fetch = true;  // segment-uniform: this segment has no problem and must pull one

  const int n = a.n;
  const int m = a.m;
  const long long queue_length = a.count_dev ? static_cast<long long>(*a.count_dev) : a.B;
  // the two stopping fields an outer loop changes between the solves of one problem (uniform otherwise)
  [[maybe_unused]] unsigned long long stop_num_iterations = a.stop.num_iterations;
  [[maybe_unused]] double stop_gradient_norm = a.stop.gradient_norm;
  double* const lds_shared = lds;  // objective's read-only region, common to the workgroup
  constexpr bool kBfgs = (ALG == kAlgBfgs);
  static_assert(!kBfgs || MR == 0, "dense BFGS keeps no (s, y) history");
  static_assert(!AR::kFma || !kBfgs, "the fused arithmetic is built for Lbfgs (either line search), not for dense BFGS");
  constexpr bool kRegScalars = scalars_in_registers(E, MR, Obj::kLdsDoubles);
  constexpr bool kGlobalPast = kRegScalars || kBfgs;  // plateau ring in global scratch
  const int lds_problem = kBfgs ? bfgs_lds_doubles_per_problem(WE, Obj::kLdsDoubles)
                                : lds_doubles_per_problem(m, WE, MR > 0, Obj::kLdsDoubles, kRegScalars);
  double* const lds_wave = lds + Obj::shared_lds_doubles() + wave_in_block * (kSegs * lds_problem);
  double* const S = lds_wave + seg * lds_problem;
  double* const Y = S + m * WE;          // (unused when the y half is register resident)
  double* const rho_mem = (MR > 0) ? Y : Y + m * WE;  // 1/(s_i.y_i) per stored pair (0 = skip, see below)
  // Register-resident y history, chronological (newest at index MR - 1).
  // rho, chronological like Yr (scalars_in_registers())
  [[maybe_unused]] double Rr[MR > 0 ? MR : 1];
  double Yr[MR > 0 ? MR : 1][E];
  double* const alpha_mem = rho_mem + m;
  // plateau ring (progress.h:139-140): LDS, or one MAX_PAST slot per resident segment in global scratch
  double* const past_f =
      kGlobalPast ? a.scratch + ((static_cast<size_t>(blockIdx.x) * (blockDim.x / kWave) + wave_in_block) * kSegs + seg) *
                                 MI355_LBFGS_MAX_PAST
                  : alpha_mem + m;

  Obj obj;
  obj.load(a.obj_params, n, sl, S + lds_problem - Obj::kLdsDoubles, lds_shared);
  if constexpr (Obj::shared_lds_doubles() > 0) {
    obj.fill_shared(lds_shared, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
    __syncthreads();
  }

  // ---- dense BFGS: H[j][i] at Hm[j * WE + i]; staging vectors for the broadcast reads ----------------
  [[maybe_unused]] double* const Hm = S;
  [[maybe_unused]] double* const vbuf = S + WE * WE;
  [[maybe_unused]] double* const sbuf = vbuf + WE;
  [[maybe_unused]] double* const hybuf = sbuf + WE;
  [[maybe_unused]] bool fresh_h = true;  // fresh_inverse_hessian_ (bfgs.h:145-147)
  [[maybe_unused]] auto bfgs_identity = [&]() {
    for (int j = 0; j < WE; ++j) {
#pragma unroll
      for (int e = 0; e < E; ++e) Hm[j * WE + sl * E + e] = (j == sl * E + e) ? 1.0 : 0.0;
    }
    segment_lds_fence();
  };
  // out_i = ((H_i0 v_0 + H_i1 v_1) + ...): the reference's matrix * vector, row i read as column i
  [[maybe_unused]] auto bfgs_matvec = [&](const double (&v)[E], double (&out)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) vbuf[sl * E + e] = v[e];
    segment_lds_fence();
    const double* const col = Hm + sl * E;
    {
      const double v0 = vbuf[0];
#pragma unroll
      for (int e = 0; e < E; ++e) out[e] = col[e] * v0;
    }
#pragma unroll 4
    for (int j = 1; j < a.n; ++j) {
      const double vj = vbuf[j];
#pragma unroll
      for (int e = 0; e < E; ++e) out[e] = out[e] + col[j * WE + e] * vj;
    }
    segment_lds_fence();
  };

  double x[E], g[E];
  double f = 0.0;
  unsigned nfev = 0, sum_k = 0;
  int mem_count = 0, mem_pos = 0;
  double scaling_factor = 1.0;
  unsigned num_iterations = 0;
  int x_delta_violations = 0, f_delta_violations = 0;
  double x_delta = 0.0, f_delta = 0.0, gradient_norm = 0.0;
  int status = MI355_STATUS_NOT_STARTED;
  bool past_init = false;
  int past_pos = 0;
  // Running upper bound on ||x||_inf (exact value at the start of a solve, then
  // grown by every step's ||x+ - x||_inf with a rounding margin).  Two tests of the
  // reference only need ||x|| when they are about to fire; the bound proves the
  // common "cannot fire" case without a reduction or a square root (see below).
  double xinf_bound = 0.0;
  const double n_as_double = static_cast<double>(a.n);
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = g[e] = 0.0;

#ifdef MI355_LBFGS_PHASE_TIMING
  unsigned long long lphase_cycles[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) lphase_cycles[i] = 0;
  unsigned long long lphase_t0 = __builtin_readcyclecounter();
  int lphase_cur = 8;  // kernel prologue
#endif
  // Solver::Minimize prologue from the point in x: evaluate (solver.h:189-192), reset the solver and its Progress
  [[maybe_unused]] double f_start = 0.0;  // that first evaluation (an outer loop reports against it)
  auto reset_solver = [&]() {
    f_start = f;
    nfev = 1;
    sum_k = 0;
    // ---- Lbfgs::InitializeSolver (lbfgs.h:72-87) ----------------------------
    mem_count = 0;
    mem_pos = 0;
    scaling_factor = 1.0;
    if constexpr (kBfgs) {  // Bfgs::InitializeSolver (bfgs.h:65-71)
      bfgs_identity();
      fresh_h = true;
    }
    // ---- Progress (progress.h:82-140) ---------------------------------------
    num_iterations = 0;
    x_delta_violations = 0;
    f_delta_violations = 0;
    x_delta = f_delta = gradient_norm = 0.0;
    status = MI355_STATUS_NOT_STARTED;
    past_init = false;
    past_pos = 0;
    xinf_bound = seg_amax<W, E>(x);
  };
  auto start_solve = [&]() {
    f = obj_eval<W, E, AR>(obj, x, g, n, sl);
    reset_solver();
  };
  // Staged refill (A/B build -DMI355_FETCH_DELAY=<passes>; default 0 = fetch and wait).  A segment that has finished
  // its problem claims the next one and issues the loads of its start point into staging registers, then sits out
  // that many passes of the wavefront's loop while the loads land, so that the other segments of the wavefront keep
  // iterating instead of stalling on one segment's HBM round trip (the profiling build charges 7.6 % of a
  // wavefront's resident time on the config-2 batch to that wait).  Measured with 1 and 3 passes: no gain on any
  // workload (profiles/r2_ab_staged_fetch.txt) — while a wavefront waits, the other wavefront of its SIMD has the
  // issue slots to itself and runs ~1.5x faster, so the wait was not lost time.  Kept as a switch, off.
#ifndef MI355_FETCH_DELAY
#define MI355_FETCH_DELAY 0
#endif
  [[maybe_unused]] int fetch_wait = 0;  // > 0: start point in flight
  [[maybe_unused]] double xn[E];
#pragma unroll
  for (int e = 0; e < E; ++e) xn[e] = 0.0;
  while (true) {
    MI355_LPHASE(0);  // loop top
    [[maybe_unused]] const bool others_running = __builtin_amdgcn_ballot_w64(!need_fetch) != 0;
    if (need_fetch) {
      if (MI355_FETCH_DELAY == 0 || fetch_wait == 0) {
        // ---- next unsolved problem from the queue ------------------------------
        MI355_LPHASE(7);  // the queue's atomic
        unsigned long long nxt = 0;
        if (sl == 0) nxt = atomicAdd(a.next_problem, 1ULL);
        const unsigned lo = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt & 0xffffffffULL)));
        const unsigned hi = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt >> 32)));
        prob = static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
        MI355_LPHASE(9);  // start point from HBM
        if (prob >= queue_length) break;  // queue drained: this segment is done
        const KernargSolveArgs ca = cold_args();
        {
          const int* const map = ca->problem_map;
          if (map != nullptr) prob = map[prob];
        }
        // ---- the start point ----------------------------------------------------
        {
          const double* const x0p = ca->x0;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            const int j = sl * E + e;
            xn[e] = (j < n) ? x0p[prob * n + j] : 0.0;
          }
        }
        obj.begin_problem(ca->per_problem, prob, ca->per_problem_stride, sl);
        if constexpr (MI355_FETCH_DELAY > 0) {
          fetch_wait = MI355_FETCH_DELAY;
          if (others_running) continue;
        }
      } else if (--fetch_wait > 0 && others_running) {
        continue;
      }
      fetch_wait = 0;
      need_fetch = false;
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = xn[e];
      if constexpr (OUTER::kEnabled) OUTER::begin(obj, oa, a, prob, x, sl, stop_num_iterations, stop_gradient_norm);
#ifdef MI355_LBFGS_PHASE_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // charge the wait for x0 to the load, not to the evaluation
#endif
      MI355_LPHASE(10);  // first evaluation + solver reset
      start_solve();
    }


    MI355_LPHASE(1);  // two-loop recursion
    // ======================= Lbfgs::OptimizationStep ========================
    // relative_eps = eps * max(1, ||x||_2) (:93-95) is only read by the descent test
    // (:214-215); it is evaluated there, and only when the bound cannot settle the test.
    double d[E];
#pragma unroll
    for (int e = 0; e < E; ++e) d[e] = g[e];  // :145
    if constexpr (kBfgs) bfgs_matvec(g, d);   // bfgs.h:81: search_direction = -H g, carried as d = H g
    const int k = mem_count;
    sum_k += k;
    if constexpr (!kBfgs) {

    // The ring is walked by slot: chronological position i lives in slot i while
    // the ring is filling and in slot (mem_pos + i) mod m once it is full
    // (:158-162).  rho_mem[slot] holds 1/(s.y), or 0 for a pair the reference
    // skips (|s.y| < eps, :165/:189): with rho = 0 both loop bodies leave d
    // unchanged (alpha = beta = 0), which is the same as skipping them.
    // Both loops are unrolled by two over a pair of register sets (A, B): while
    // the butterfly of one pair runs, the other set is being filled from LDS.
    // The prefetch is unconditional — past the last pair it reads a valid but
    // unused slot — so the loop body has no data-dependent branch.
    const bool full = (mem_count >= m);
    const double* const Sl = S + sl * E;
    [[maybe_unused]] const double* const Yl = Y + sl * E;
    auto prev_slot = [&](int slot) { return (slot == 0) ? m - 1 : slot - 1; };
    auto next_slot = [&](int slot) { return (slot + 1 == m) ? 0 : slot + 1; };
    if constexpr (MR == 0) {
      auto load_pair = [&](int slot, double (&sv)[E], double (&yv)[E], double& rho) {
  #pragma unroll
        for (int e = 0; e < E; ++e) {
          sv[e] = Sl[slot * WE + e];
          yv[e] = Yl[slot * WE + e];
        }
        rho = rho_mem[slot];
      };
      // first loop, newest -> oldest (:157-171)
      if (k > 0) {
        auto body = [&](const double (&sv)[E], const double (&yv)[E], double rho, int i) {
          const double alpha = rho * seg_dot<W, E, AR>(sv, d);
          if (sl == 0) alpha_mem[i] = alpha;  // read back by the whole segment in loop 2
  #pragma unroll
          for (int e = 0; e < E; ++e) d[e] = AR::nmadd(alpha, yv[e], d[e]);
        };
        int slot = full ? prev_slot(mem_pos) : k - 1;
        double sa[E], ya[E], ra, sb[E], yb[E], rb;
        load_pair(slot, sa, ya, ra);
        int i = k - 1;
        while (true) {
          slot = prev_slot(slot);
          load_pair(slot, sb, yb, rb);
          body(sa, ya, ra, i);
          if (i == 0) break;
          --i;
          slot = prev_slot(slot);
          load_pair(slot, sa, ya, ra);
          body(sb, yb, rb, i);
          if (i == 0) break;
          --i;
        }
      }
        // Second mode, non-constant Hessian: M^-1 = 1 / (|diag H(x)| + eps) at the current iterate (:129-134), applied
      // to the centre of the recursion in place of scaling_factor_
      [[maybe_unused]] double pre_x[E];
      [[maybe_unused]] bool from_functor = false;
      if constexpr (HasHessDiag<Obj>::value) {
        if (a.hess_from_functor) {
          from_functor = true;
          double hd[E];
          obj.template hess_diag<W, E>(x, hd, n, sl);
#pragma unroll
          for (int e = 0; e < E; ++e) pre_x[e] = 1.0 / (__builtin_fabs(hd[e]) + 2.220446049250313e-16);
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        if constexpr (HasHessDiag<Obj>::value) {
          if (from_functor) {
            d[e] = ((j < n) ? pre_x[e] : 0.0) * d[e];
            continue;
          }
        }
        d[e] = (a.precond != nullptr) ? ((j < n) ? a.precond[j] : 0.0) * d[e]   // :177-179
                                      : d[e] * scaling_factor;                  // :181
      }
      segment_lds_fence();
      // second loop, oldest -> newest (:185-196)
      if (k > 0) {
        auto body = [&](const double (&sv)[E], const double (&yv)[E], double rho, double al) {
          const double beta = rho * seg_dot<W, E, AR>(yv, d);
          const double c = al - beta;
  #pragma unroll
          for (int e = 0; e < E; ++e) d[e] = AR::madd(sv[e], c, d[e]);
        };
        int slot = full ? mem_pos : 0;
        double sa[E], ya[E], ra, ala, sb[E], yb[E], rb, alb;
        load_pair(slot, sa, ya, ra);
        ala = alpha_mem[0];
        int i = 0;
        while (true) {
          slot = next_slot(slot);
          load_pair(slot, sb, yb, rb);
          alb = alpha_mem[i + 1];  // i + 1 <= m: stays inside this problem's LDS block
          body(sa, ya, ra, ala);
          if (++i == k) break;
          slot = next_slot(slot);
          load_pair(slot, sa, ya, ra);
          ala = alpha_mem[i + 1];
          body(sb, yb, rb, alb);
          if (++i == k) break;
        }
      }
    } else {
      // ---- y history in registers: position t = 0 is the newest pair ----------
      // The loops are fully unrolled, so the two s buffers alternate by a compile-time index (no
      // copies), and — where the register budget allows (kRegScalars) — rho and alpha are register
      // arrays instead of LDS round trips through lane 0 of the segment.
      auto load_s = [&](int slot, double (&sv)[E]) {
#pragma unroll
        for (int e = 0; e < E; ++e) sv[e] = Sl[slot * WE + e];
      };
      [[maybe_unused]] double al[MR > 0 ? MR : 1];   // alpha by t (kRegScalars)
      double sbuf[2][E], rbuf[2];
      // first loop, newest -> oldest (:157-171); alpha is indexed by t
      if (k > 0) {
        int slot = full ? prev_slot(mem_pos) : k - 1;
        load_s(slot, sbuf[0]);
        if constexpr (!kRegScalars) rbuf[0] = rho_mem[slot];
#pragma unroll
        for (int t = 0; t < MR; ++t) {
          if (t < k) {
            slot = prev_slot(slot);
            load_s(slot, sbuf[(t + 1) & 1]);  // prefetch the next (older) pair; past the last one it is unused
            if constexpr (!kRegScalars) rbuf[(t + 1) & 1] = rho_mem[slot];
            const double rho = kRegScalars ? Rr[MR - 1 - t] : rbuf[t & 1];
            const double alpha = rho * seg_dot<W, E, AR>(sbuf[t & 1], d);
            if constexpr (kRegScalars) {
              al[t] = alpha;
            } else {
              if (sl == 0) alpha_mem[t] = alpha;
            }
            const double (&ycol)[E] = Yr[MR - 1 - t];
#pragma unroll
            for (int e = 0; e < E; ++e) d[e] = AR::nmadd(alpha, ycol[e], d[e]);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        d[e] = (a.precond != nullptr) ? ((j < n) ? a.precond[j] : 0.0) * d[e]   // :177-179
                                      : d[e] * scaling_factor;                  // :181
      }
      if constexpr (!kRegScalars) segment_lds_fence();
      // second loop, oldest -> newest (:185-196).  With a full history (the steady state) the first
      // step is t = MR - 1 and the buffers alternate statically; while the history fills, the walk
      // starts at the runtime position k - 1 and the next pair is copied into place instead.
      if (k > 0) {
        int slot = full ? mem_pos : 0;
        auto second_loop = [&](auto full_history) {
          constexpr bool kStatic = decltype(full_history)::value;
          load_s(slot, sbuf[0]);
          double ala = 0.0;
          if constexpr (!kRegScalars) {
            rbuf[0] = rho_mem[slot];
            ala = alpha_mem[k - 1];
          }
#pragma unroll
          for (int t = MR - 1; t >= 0; --t) {
            if (kStatic || t < k) {
              constexpr int kZero = 0;
              const int cur = kStatic ? ((MR - 1 - t) & 1) : kZero;
              const int nxt = kStatic ? ((MR - t) & 1) : 1;
              slot = next_slot(slot);
              load_s(slot, sbuf[nxt]);
              double aln = 0.0;
              if constexpr (!kRegScalars) {
                rbuf[nxt] = rho_mem[slot];
                aln = alpha_mem[t > 0 ? t - 1 : 0];
              }
              const double rho = kRegScalars ? Rr[MR - 1 - t] : rbuf[cur];
              const double alt = kRegScalars ? al[t] : ala;
              const double (&ycol)[E] = Yr[MR - 1 - t];
              const double beta = rho * seg_dot<W, E, AR>(ycol, d);
              const double c = alt - beta;
#pragma unroll
              for (int e = 0; e < E; ++e) d[e] = AR::madd(sbuf[cur][e], c, d[e]);
              if constexpr (!kStatic) {
#pragma unroll
                for (int e = 0; e < E; ++e) sbuf[0][e] = sbuf[1][e];
                rbuf[0] = rbuf[1];
              }
              ala = aln;
            }
          }
        };
        if (k == MR) second_loop(std::true_type{}); else second_loop(std::false_type{});
      }
    }
    }  // !kBfgs

    MI355_LPHASE(2);  // descent test, initial step
    const double descent_direction = -seg_dot<W, E, AR>(g, d);  // :199
    // cvsrch's dginit = g.s with s = -d (more_thuente.h:151) is the same number:
    // every product and every partial sum is the exact negation.
    double dginit = descent_direction;
    double alpha_init = 1.0;                                 // :207-213
    if constexpr (kBfgs) {
      // bfgs.h:87-106.  phi = g . search_direction is the same number as descent_direction.
      if ((descent_direction > 0.0) || (descent_direction != descent_direction)) {
        bfgs_identity();
#pragma unroll
        for (int e = 0; e < E; ++e) d[e] = g[e];             // search_direction = -g
        fresh_h = true;
        dginit = -seg_dot<W, E, AR>(g, g);                       // what the line search computes as g . s
      }
      if (fresh_h) {
        const double dn = __builtin_sqrt(seg_dot<W, E, AR>(d, d));
        alpha_init = (dn > eps) ? 1.0 / dn : 1.0;
      }
    } else if (mem_count == 0) {
      const double dn = __builtin_sqrt(seg_dot<W, E, AR>(d, d));
      alpha_init = (dn > eps) ? 1.0 / dn : 1.0;
    }
    // :214-224  fallback iff !isfinite(descent) || descent > -eps * relative_eps.
    // ||x||_2 <= n * ||x||_inf <= n * xinf_bound, so -eps*(eps*max(1, n*bound)) is a lower
    // bound of the threshold: a finite descent at or below it can never trigger.
    bool invalid_direction;
    if constexpr (kBfgs) {
      invalid_direction = false;
    } else
    if (__builtin_isfinite(descent_direction) &&
        descent_direction <= -eps * (eps * dmax(1.0, n_as_double * xinf_bound))) {
      invalid_direction = false;
    } else {
      const double relative_eps = eps * dmax(1.0, __builtin_sqrt(seg_dot<W, E, AR>(x, x)));  // :93-95
      invalid_direction = !__builtin_isfinite(descent_direction) || descent_direction > -eps * relative_eps;
    }
    if (invalid_direction) {
#pragma unroll
      for (int e = 0; e < E; ++e) d[e] = -g[e];
      mem_count = 0;
      mem_pos = 0;
      const double gg = seg_dot<W, E, AR>(g, g);
      const double gn = __builtin_sqrt(gg);
      alpha_init = (gn > eps) ? 1.0 / gn : 1.0;
      dginit = gg;  // s = -d = g: the line search sees g.g >= 0 and returns at once (quirk Q1)
    }

    // line search along -d (:231-232); keep the current state for s, y and
    // for the non-finite bail-out (:239-241).
    MI355_LPHASE(3);  // line search
    double xp[E], gp[E];
    const double fprev = f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      xp[e] = x[e];
      gp[e] = g[e];
    }
    [[maybe_unused]] bool ls_failed = false;
    if constexpr (LS == MI355_LS_HAGER_ZHANG) {
      double stp = alpha_init;
      nfev += hz_search<W, E, AR>(obj, x, f, g, stp, d, dginit, n, sl, ls_failed);
    } else {
      nfev += mt_cvsrch<W, E, Obj, AR>(obj, x, f, g, alpha_init, d, dginit, n, sl);
    }
    if constexpr (LS == MI355_LS_HAGER_ZHANG) {
      if (ls_failed) {  // hzls returned -1: the State overload hands back the start state (hager_zhang.h:100-116)
        f = fprev;
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = gp[e];
      }
    }

    MI355_LPHASE(4);  // s, y, curvature test, history push, scaling
    double sv[E], yv[E];
    double s_inf = 0.0;  // ||x+ - x||_inf: Progress::Update's x_delta (:190), formed here because the curvature test uses it
    if (!kBfgs && !__builtin_isfinite(f)) {  // Lbfgs only: return current (:239-241)
      f = fprev;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        x[e] = xp[e];
        g[e] = gp[e];
        sv[e] = 0.0;  // x_delta below sees next == current
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        sv[e] = x[e] - xp[e];  // :248
        yv[e] = g[e] - gp[e];  // :249
      }
      s_inf = seg_amax<W, E>(sv);
      const double sy = seg_dot<W, E, AR>(sv, yv);   // :265
      const double yy = seg_dot<W, E, AR>(yv, yv);   // :290 (== grad_diff.norm()^2 of :266)
      // :266-267  accept iff sy > eps*||s||*||y||.  sy <= 0 can never pass (the threshold
      // is >= 0).  sy^2 > 4 eps^2 ss yy implies sy > 2 eps sqrt(ss yy) > threshold (the
      // rounding of the two sides is ~1e-16 relative against a factor 2 of slack), so the
      // two square roots are only evaluated for nearly orthogonal or tiny pairs — and since
      // ss = ||s||^2 <= n ||s||_inf^2, the bound settles the common case without reducing ss
      // at all (the decision is the reference's either way; ss is formed when the bound cannot tell).
      bool accept = false;
      if (sy > 0.0) {
        const double ss_bound = ((n_as_double * s_inf) * s_inf) * (1.0 + 1e-9);
        const double rhs_bound = ((4.0 * eps * eps) * ss_bound) * yy;
        if (rhs_bound >= 1e-290 && sy * sy > rhs_bound) {
          accept = true;
        } else {
          const double ss = seg_dot<W, E, AR>(sv, sv);
          const double rhs = ((4.0 * eps * eps) * ss) * yy;
          if (rhs >= 1e-290 && sy * sy > rhs) {
            accept = true;
          } else {
            const double sy_threshold = eps * __builtin_sqrt(ss) * __builtin_sqrt(yy);  // :266
            accept = sy > sy_threshold;
          }
        }
      }
      if constexpr (kBfgs) {
        if (accept) {                            // bfgs.h:124-132 (same acceptance test as lbfgs.h:266)
          const double rho = 1.0 / sy;
          double Hy[E];
          bfgs_matvec(yv, Hy);
          const double yHy = seg_dot<W, E, AR>(yv, Hy);
          const double c = rho * (rho * yHy + 1.0);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            sbuf[sl * E + e] = sv[e];
            hybuf[sl * E + e] = Hy[e];
          }
          segment_lds_fence();
          double* const col = Hm + sl * E;
          for (int j = 0; j < n; ++j) {
            const double sj = sbuf[j], hyj = hybuf[j];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const double h = col[j * WE + e];
              col[j * WE + e] = (h - rho * (sv[e] * hyj + Hy[e] * sj)) + c * (sv[e] * sj);
            }
          }
          segment_lds_fence();
          fresh_h = false;
        }
      } else {
      if (accept) {                              // :267-280
        int slot;
        if (mem_count < m) {
          slot = mem_count;
          mem_count++;
        } else {
          slot = mem_pos;
          mem_pos = (mem_pos + 1 == m) ? 0 : mem_pos + 1;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) S[slot * WE + sl * E + e] = sv[e];
        if constexpr (MR == 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) Y[slot * WE + sl * E + e] = yv[e];
        } else {
          // chronological register history: drop the oldest, append the newest
#pragma unroll
          for (int i = 0; i + 1 < MR; ++i) {
#pragma unroll
            for (int e = 0; e < E; ++e) Yr[i][e] = Yr[i + 1][e];
          }
#pragma unroll
          for (int e = 0; e < E; ++e) Yr[MR - 1][e] = yv[e];
        }
        const double rho_new = (__builtin_fabs(sy) < eps) ? 0.0 : 1.0 / sy;
        if constexpr (kRegScalars) {
#pragma unroll
          for (int i = 0; i + 1 < MR; ++i) Rr[i] = Rr[i + 1];
          Rr[MR - 1] = rho_new;
        } else {
          if (sl == 0) rho_mem[slot] = rho_new;
        }
        segment_lds_fence();
      }
      if (yy > eps) {                            // :289-298
        const double temp_scaling = sy / yy;
        if (__builtin_isfinite(temp_scaling) && __builtin_fabs(temp_scaling) <= 1e7) {
          scaling_factor = dmax(temp_scaling, eps);
        }
      }
      }  // !kBfgs
    }

    MI355_LPHASE(5);  // Progress::Update
    // ========================== Progress::Update ============================
    num_iterations++;                                    // :188
    f_delta = __builtin_fabs(f - fprev);                 // :189
    x_delta = s_inf;                                     // :190
    gradient_norm = seg_amax<W, E>(g);                   // :195
    xinf_bound = (xinf_bound + x_delta) * (1.0 + 4.0 * eps);  // |x+_j| <= |x_j| + |x+_j - x_j|
    const mi355_lbfgs_stop& st = a.stop;
    status = MI355_STATUS_CONTINUE;
    bool decided = false;
    if ((stop_num_iterations > 0) && (num_iterations > stop_num_iterations)) {  // :212-216
      status = MI355_STATUS_ITERATION_LIMIT;
      decided = true;
    }
    if (!decided) {                                      // :254-262
      if ((st.x_delta > 0) && (x_delta < st.x_delta)) {
        x_delta_violations++;
        if (x_delta_violations >= st.x_delta_violations) {
          status = MI355_STATUS_X_DELTA_VIOLATION;
          decided = true;
        }
      } else {
        x_delta_violations = 0;
      }
    }
    if (!decided) {                                      // :263-277
      const double fscale =
          st.f_delta_relative ? dmax(dmax(__builtin_fabs(f), __builtin_fabs(fprev)), 1.0) : 1.0;
      if ((st.f_delta > 0) && (f_delta < st.f_delta * fscale)) {
        f_delta_violations++;
        if (f_delta_violations >= st.f_delta_violations) {
          status = MI355_STATUS_F_DELTA_VIOLATION;
          decided = true;
        }
      } else {
        f_delta_violations = 0;
      }
    }
    if (!decided && st.past > 0) {                       // :280-298
      const int p = st.past;
      if (!past_init) {
        if (sl < p) past_f[sl] = f;                      // ring lazily filled with current f
        past_init = true;
        past_pos = 0;
        segment_lds_fence();
      }
      if (static_cast<int>(num_iterations) > p) {
        const double pf = past_f[past_pos];
        const double rate = __builtin_fabs(pf - f) / dmax(1.0, __builtin_fabs(f));
        if (rate < st.past_delta) {
          status = MI355_STATUS_F_DELTA_VIOLATION;
          decided = true;
        }
      }
      if (!decided) {
        if (sl == 0) past_f[past_pos] = f;
        segment_lds_fence();
        past_pos = (past_pos + 1 == p) ? 0 : past_pos + 1;
      }
    }
    if (!decided && stop_gradient_norm > 0) {            // :299-317
      if (st.gradient_norm_relative) {
        // scale = max(1, ||x||_inf) <= max(1, bound): if even the bound's threshold is not
        // reached the test cannot fire and ||x||_inf need not be computed.
        if (gradient_norm < stop_gradient_norm * dmax(1.0, xinf_bound)) {
          const double xinf = seg_amax<W, E>(x);
          xinf_bound = xinf;
          if (gradient_norm < stop_gradient_norm * dmax(1.0, xinf)) status = MI355_STATUS_GRADIENT_NORM_VIOLATION;
        }
      } else if (gradient_norm < stop_gradient_norm) {
        status = MI355_STATUS_GRADIENT_NORM_VIOLATION;
      }
    }
    if (status == MI355_STATUS_CONTINUE && a.hessian_condition_fires)   // :318-325 (Second mode)
      status = MI355_STATUS_HESSIAN_CONDITION_VIOLATION;
    if constexpr (HasHessFull<Obj>::value && MR == 0 && !kBfgs && !OUTER::kEnabled) {
      // the same test for a Hessian that is not constant: H(x) of the new iterate, every iteration (:203-210)
      const double condition_stop = cold_args()->hessian_condition_stop;
      if (condition_stop > 0.0 && status == MI355_STATUS_CONTINUE) {
        double* const hc = lds + Obj::shared_lds_doubles() + static_cast<int>(blockDim.x / kWave) * (kSegs * lds_problem) +
                           ((wave_in_block * kSegs + seg) * hessian_condition_lds_doubles(n, W));
        obj.template hess_full<W, E>(x, hc, n, sl);
        const double condition = seg_hessian_condition<W>(hc, hc + n * n, reinterpret_cast<int*>(hc + n * n + W * (n + 1)), n, sl);
        if (condition > condition_stop) status = MI355_STATUS_HESSIAN_CONDITION_VIOLATION;
      }
    }
    MI355_LPHASE(6);  // results / refill
    if constexpr (!OUTER::kEnabled)
      trace_iteration<E>(a, prob, n, sl, num_iterations, status, f, x_delta, f_delta, gradient_norm, x, g);
    if constexpr (OUTER::kEnabled) {
      if (status != MI355_STATUS_CONTINUE) {
        // the outer loop takes the solve's result: either another solve from here, or the problem is finished
        // (it has written its own results)
        if (OUTER::step(obj, oa, a, prob, x, num_iterations, nfev, sum_k, sl, stop_num_iterations, stop_gradient_norm,
                        f_start, f, g)) {
          reset_solver();  // f and g are already the next solve's first evaluation
        } else {
          need_fetch = true;
        }
      }
    } else if (status != MI355_STATUS_CONTINUE) {
      // ---- results of this problem (solver.h:223) ---------------------------
      const KernargSolveArgs ca = cold_args();
      double* const x_out = ca->x_out;
      double* const g_out = ca->g_out;
      double* const f_out = ca->f_out;
      mi355_lbfgs_progress* const progress_out = ca->progress_out;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        if (j < n) {
          x_out[prob * n + j] = x[e];
          if (g_out) g_out[prob * n + j] = g[e];
        }
      }
      if (sl == 0) {
        f_out[prob] = f;
        if (progress_out) {
          mi355_lbfgs_progress pr;
          pr.status = status;
          pr.num_iterations = num_iterations;
          pr.nfev = nfev;
          pr.sum_k = sum_k;
          pr.x_delta = x_delta;
          pr.f_delta = f_delta;
          pr.gradient_norm = gradient_norm;
          progress_out[prob] = pr;
        }
      }
      need_fetch = true;
    }
  }
#ifdef MI355_LBFGS_PHASE_TIMING
  MI355_LPHASE(0);
  if (lane == 0 && a.profile != nullptr) {
#pragma unroll
    for (int i = 0; i < 12; ++i) atomicAdd(a.profile + i, lphase_cycles[i]);
  }
#endif
}

// One objective evaluation per problem (parity tests of the device functors).
template <int W, int E, class Obj, class AR = ArithExact>
__global__ __launch_bounds__(64) void eval_kernel(const SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int kSegs = kWave / W;
  const int lane = threadIdx.x;
  const int seg = lane / W;
  const int sl = lane % W;
  const long long prob = static_cast<long long>(blockIdx.x) * kSegs + seg;
  const int n = a.n;
  Obj obj;
  obj.load(a.obj_params, n, sl,
           lds + Obj::shared_lds_doubles() + seg * (Obj::kLdsDoubles > 0 ? Obj::kLdsDoubles : 1), lds);
  if constexpr (Obj::shared_lds_doubles() > 0) {
    obj.fill_shared(lds, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
    __syncthreads();
  }
  if (prob >= a.B) return;
  double x[E], g[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    x[e] = (j < n) ? a.x0[prob * n + j] : 0.0;
  }
  obj.begin_problem(a.per_problem, prob, a.per_problem_stride, sl);
  const double f = obj_eval<W, E, AR>(obj, x, g, n, sl);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    if (j < n && a.g_out) a.g_out[prob * n + j] = g[e];
  }
  if (sl == 0) a.f_out[prob] = f;
}

// One HagerZhang::Search (State overload, hager_zhang.h:100-116) per problem: from x0[b] along
// ls_direction[b] with the initial step ls_alpha_init[b].  Test / host-API hook, the counterpart
// of mi355_lbfgs_cstep_batch for this line search.
template <int W, int E, class Obj>
__global__ __launch_bounds__(64) void hz_search_kernel(const SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int kSegs = kWave / W;
  const int lane = threadIdx.x;
  const int seg = lane / W;
  const int sl = lane % W;
  const long long prob = static_cast<long long>(blockIdx.x) * kSegs + seg;
  const int n = a.n;
  Obj obj;
  obj.load(a.obj_params, n, sl,
           lds + Obj::shared_lds_doubles() + seg * (Obj::kLdsDoubles > 0 ? Obj::kLdsDoubles : 1), lds);
  if constexpr (Obj::shared_lds_doubles() > 0) {
    obj.fill_shared(lds, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
    __syncthreads();
  }
  if (prob >= a.B) return;
  double x[E], g[E], d[E], x0[E], g0[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    x[e] = (j < n) ? a.x0[prob * n + j] : 0.0;
    d[e] = (j < n) ? -a.ls_direction[prob * n + j] : 0.0;
  }
  obj.begin_problem(a.per_problem, prob, a.per_problem_stride, sl);
  double f = obj.template eval<W, E>(x, g, n, sl);
  const double f0 = f;
#pragma unroll
  for (int e = 0; e < E; ++e) {
    x0[e] = x[e];
    g0[e] = g[e];
  }
  double stp = a.ls_alpha_init[prob];
  bool failed = false;
  const double dginit = -seg_dot<W, E>(g, d);  // g.s
  const int nfev = hz_search<W, E>(obj, x, f, g, stp, d, dginit, n, sl, failed);
  if (failed) {
    f = f0;
    if (nfev == 0) stp = a.ls_alpha_init[prob];  // not a descent direction: the step is left alone (:302)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      x[e] = x0[e];
      g[e] = g0[e];
    }
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    if (j < n) {
      a.x_out[prob * n + j] = x[e];
      if (a.g_out) a.g_out[prob * n + j] = g[e];
    }
  }
  if (sl == 0) {
    a.f_out[prob] = f;
    a.ls_alpha_out[prob] = stp;
    if (a.ls_nfev_out) a.ls_nfev_out[prob] = static_cast<unsigned>(nfev);
  }
}

}  // namespace mi355
