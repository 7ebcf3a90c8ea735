// ridge_mfma_kernel.hpp — L-BFGS on the ridge objective with the two matrix-vector products of every
// objective evaluation on the matrix cores (BASELINE config 4: "objective GEMV on MFMA").
//
// f(x) = ||A x - y_b||^2 + lambda ||x||^2 (README.md:122-167) costs 2 * rows * n multiply-adds per
// evaluation — 3/4 of the solve time on the VALU.  `v_mfma_f64_16x16x4_f64` multiplies a 16 x 4 tile of
// A by the trial points of SIXTEEN problems at once, at four times the VALU's multiply-then-add rate
// and with A fetched once per tile instead of once per problem.  That needs sixteen problems to be
// evaluated together, so this kernel inverts the structure of lbfgs_solve_kernel: the unit of
// scheduling is the workgroup — eight wavefronts, two problems each (W = 32 lanes, E = 2), sixteen
// problem slots — and its loop makes one JOINT objective evaluation per pass:
//
//   every segment publishes the point it needs evaluated (x0 of a fresh problem, or the current
//   line-search trial point) in LDS                                                         barrier
//   r = A X - Y:  wavefront w computes residual rows 16w .. 16w+15 of all 16 problems,
//                 16 MFMAs over the n = 64 columns                                          barrier
//   G = A^T R:    wavefront w computes gradient coordinates 16t .. 16t+15 (t = w mod 4) over the
//                 row half w / 4, 16 MFMAs; the two halves are added at the pick-up             barrier
//   every segment picks up its f and g and advances its own scalar state — Moré–Thuente step
//   selection, or the end of the iteration (history update, stopping tests, results and refill from the
//   work queue) followed by the next two-loop recursion — until it needs the next evaluation.
//
// So a solve takes one pass per objective evaluation instead of one per iteration, and the line search
// is a state machine around the single evaluation site, like csrc/hager_zhang_device.hpp.  Both halves
// of the (s, y) history, rho and alpha live in registers (chronological, shifted on a push); LDS holds
// A (pitch 65: both fragment shapes are conflict-free), the exchange tiles X, R, G and nothing else.
//
// Arithmetic contract (objective id MI355_OBJ_SQUARED_ERROR_RIDGE_MFMA).  The MFMA accumulates
// D = C + sum_k A_k B_k as a chain of fused multiply-adds in k order (checked bitwise on gfx950:
// scripts/microbench/mfma_f64_probe.hip), so with the tiles walked in natural order
//   r_i = fma(A_i,n-1, x_n-1, ... fma(A_i1, x_1, fma(A_i0, x_0, 0))) - y_i
//   g_j = 2 * (fma(A_63,j, r_63, ... fma(A_0j, r_0, 0)) + fma(A_127,j, r_127, ... fma(A_64,j, r_64, 0))) + lambda * (2 x_j)
// and ||r||^2, ||x||^2 and every other reduction on the same pairwise trees as the other kernels.  That
// is the reference's objective up to the rounding of the two products (the VALU objective id 2 keeps
// them as the multiply-then-add sums that are bit-identical to the README functors); x*, f* agree
// with the reference-order solve within the 1e-6 tolerance, and the oracle twin
// (SquaredErrorRidge::fma_chains) reproduces this kernel bit for bit.
#pragma once
#include "lbfgs_kernel.hpp"

namespace mi355 {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kJointSlots = 16;     // problems evaluated together = N of the MFMA tile
constexpr int kJointWaves = 8;      // two slots per wavefront (W = 32, E = 2); the W = 16, E = 4 variant runs 4
constexpr int kJointRows = 128;     // = MI355_LBFGS_MAX_ROWS
constexpr int kJointCols = 64;
constexpr int kJointPitchA = kJointCols + 1;
constexpr int kJointPitchX = kJointCols + 1;
constexpr int kJointPitchR = kJointRows + 1;

// A, the exchange tiles X, G, R, the right-hand sides Y of the sixteen slots, the slots' problem
// indices, and MR doubles of private scratch per lane
// (the alpha of the running two-loop recursion: a register array would push the kernel into spills)
// (`waves` = wavefronts per workgroup; the four-wavefront variant keeps alpha in registers: scratch_per_lane = 0)
__host__ __device__ constexpr int ridge_mfma_lds_doubles(int MR, int waves = kJointWaves, bool alpha_in_lds = true) {
  return kJointRows * kJointPitchA + 2 * kJointSlots * kJointPitchX + 2 * kJointSlots * kJointPitchR + kJointSlots +
         (alpha_in_lds ? MR * waves * kWave : 0);
}

// params (device): rows, lambda, then the LDS image of A: A[i][j] at i * kJointPitchA + j, zero padded to
// 128 x 65.  per_problem: y[B][stride].  Requires n <= 64, rows <= 128, m <= MR.
//
// Two mappings of the sixteen slots onto wavefronts (results are bit-identical: every reduction is the canonical
// pairwise tree, the matrix products are the same chains):
//   W = 32, E = 2   eight wavefronts x two problems, two wavefronts per SIMD (round 1)
//   W = 16, E = 4   four wavefronts x four problems, ONE wavefront per SIMD with the whole (s, y) history, rho and
//                   alpha in registers (256 VGPRs + 118 AGPRs): every butterfly and every scalar instruction serves
//                   four problems instead of two and a reduction has four levels, all DPP (no permlane swap); each
//                   wavefront owns two residual row tiles (two independent MFMA accumulator chains) and one gradient
//                   tile, so all four take part in both matrix phases.  Measured 7 % SLOWER than W = 32 (27.3 vs
//                   25.6 ms on the config-4 batch, 24.9 vs 23.2 ms after the later changes; profiles/r2_ab_ridge_mapping.txt): with a single wavefront per
//                   SIMD nothing hides the LDS latency of the fragment loads or the dependent-issue stalls, which
//                   costs more than the halved instruction count saves.  Selectable (lanes_per_problem = 16), not
//                   the default.
// AR: ArithExact, or ArithFma — the solver's inner products, the two-loop axpys and the trial point fused as in
// lbfgs_solve_kernel (twin: the butterfly_fma policy with fma_group = E on the same objective); the objective's own
// arithmetic above is the same under both.
template <int MR, int W = 32, int E = 2, class AR = ArithExact>
__global__ __launch_bounds__((kWave / W) == 2 ? 512 : 256) void ridge_mfma_solve_kernel(const SolveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  static_assert((W == 32 && E == 2) || (W == 16 && E == 4), "mapping");
  constexpr int kSegsPerWave = kWave / W;                 // 2 or 4 slots per wavefront
  constexpr int kWaves = kJointSlots / kSegsPerWave;      // 8 or 4 wavefronts per workgroup
  constexpr int RPL = kJointRows / W;                     // residual rows per lane of a segment (4 or 8)
  constexpr bool kAlphaInLds = (W == 32);
  constexpr double eps = 2.220446049250313e-16;
  double* const A_lds = lds;
  double* const X_lds = A_lds + kJointRows * kJointPitchA;
  double* const G_lds = X_lds + kJointSlots * kJointPitchX;
  double* const R_lds = G_lds + kJointSlots * kJointPitchX;
  double* const Y_lds = R_lds + kJointSlots * kJointPitchR;   // y of the problem in each slot (zero when idle)
  // alpha_t of the two-loop recursion: segment-uniform, but every lane keeps its own copy at
  // al_lds[t * 512 + tid] (conflict-free, no hand-off between lanes, so no fence)
  int* const alive_flags = reinterpret_cast<int*>(Y_lds + kJointSlots * kJointPitchR);   // one per wavefront (the kJointSlots doubles)
  [[maybe_unused]] double* const al_lds = Y_lds + kJointSlots * kJointPitchR + kJointSlots + threadIdx.x;
  [[maybe_unused]] double al_reg[kAlphaInLds ? 1 : MR];

  const int tid = static_cast<int>(threadIdx.x);
  const int lane = tid & (kWave - 1);
  const int wave = tid / kWave;
  const int seg = lane / W;
  const int sl = lane % W;
  const int slot = kSegsPerWave * wave + seg;
  const int n = a.n;
  const int m = a.m;
  const int rows = static_cast<int>(a.obj_params[0]);
  const double lambda = a.obj_params[1];
  for (int t = tid; t < kJointRows * kJointPitchA; t += static_cast<int>(blockDim.x)) A_lds[t] = a.obj_params[2 + t];
  __syncthreads();

  // plateau ring of the stopping test (progress.h:139-140): global scratch, one slot per problem slot
  double* const past_f = a.scratch + (static_cast<size_t>(blockIdx.x) * kJointSlots + slot) * MI355_LBFGS_MAX_PAST;
  double* const xrow = X_lds + slot * kJointPitchX + sl * E;
  const double* const grow = G_lds + slot * kJointPitchX + sl * E;
  const double* const rrow = R_lds + slot * kJointPitchR + sl * RPL;
  double* const yrow = Y_lds + slot * kJointPitchR + sl * RPL;

  // ---- per-problem state (segment-uniform scalars, lane-distributed vectors) -------------------
  long long prob = 0;
  bool has_problem = false, drained = false, fresh = false;
  double x[E], g[E], d[E], wa[E], gp[E];
  double Sr[MR][E], Yr[MR][E], Rr[MR];  // chronological history, newest at index MR - 1
  double f = 0.0, fprev = 0.0;
  unsigned nfev = 0, sum_k = 0;
  int mem_count = 0;
  double scaling_factor = 1.0;
  unsigned num_iterations = 0;
  int x_delta_violations = 0, f_delta_violations = 0;
  double x_delta = 0.0, f_delta = 0.0, gradient_norm = 0.0;
  bool past_init = false;
  int past_pos = 0;
  double xinf_bound = 0.0;
  const double n_as_double = static_cast<double>(n);
  // Moré–Thuente state (more_thuente.h:137-256) between evaluations
  constexpr double xtol = 1e-15, ftol = 1e-4, gtol = 0.9, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4.0;
  constexpr int maxfev = 20;
  double stx = 0, fx = 0, dgx = 0, sty = 0, fy = 0, dgy = 0, stp = 0, stmin = 0, stmax = 0;
  double width = 0, width1 = 0, finit = 0, dginit = 0, dgtest = 0;
  bool brackt = false, stage1 = true;
  int ls_nfev = 0, infoc = 1;
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = g[e] = d[e] = wa[e] = gp[e] = 0.0;

  // cvsrch loop head (:179-193): interval bounds, clamp, fall back to the best step
  auto ls_prepare = [&]() {
    if (brackt) {
      stmin = dmin(stx, sty);
      stmax = dmax(stx, sty);
    } else {
      stmin = stx;
      stmax = stp + xtrapf * (stp - stx);
    }
    stp = dclamp(stp, stpmin, stpmax);
    if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (ls_nfev >= maxfev - 1) || (infoc == 0) ||
        (brackt && ((stmax - stmin) <= (xtol * stmax)))) {
      stp = stx;
    }
  };

#ifdef MI355_LBFGS_PHASE_TIMING
  unsigned long long lphase_cycles[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) lphase_cycles[i] = 0;
  unsigned long long lphase_t0 = __builtin_readcyclecounter();
  int lphase_cur = 0;
#endif
  while (true) {
    MI355_LPHASE(0);  // refill from the work queue (atomic + x0, y loads)
    // ---- (1) an empty slot pulls the next unsolved problem from the queue ----------------------
    if (!has_problem && !drained) {
      unsigned long long nxt = 0;
      if (sl == 0) nxt = atomicAdd(a.next_problem, 1ULL);
      const unsigned lo = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt & 0xffffffffULL)));
      const unsigned hi = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt >> 32)));
      prob = static_cast<long long>((static_cast<unsigned long long>(hi) << 32) | lo);
      if (prob >= a.B) {
        drained = true;
#pragma unroll
        for (int q = 0; q < RPL; ++q) yrow[q] = 0.0;  // an idle slot evaluates x = 0 against y = 0
      } else {
        has_problem = true;
        fresh = true;
        const KernargSolveArgs ca = cold_args();   // once-per-problem pointers are read where they are used
        const double* const x0p = ca->x0;
        const double* const ypp = ca->per_problem;
        const int ystride = ca->per_problem_stride;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = sl * E + e;
          x[e] = (j < n) ? x0p[prob * n + j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < RPL; ++q) {  // this slot's right-hand side, read by the matrix phase of every pass
          const int row = RPL * sl + q;
          yrow[q] = (row < rows) ? ypp[prob * ystride + row] : 0.0;
        }
      }
    }
    // ---- (2) publish the point to evaluate ------------------------------------------------------
    MI355_LPHASE(7);  // publish + wait for the other wavefronts (barrier A)
    // x0 of a fresh problem, or the line-search trial point wa + stp * s (re-formed after the matrix
    // phase instead of being kept in registers across it)
    auto trial_point = [&](double (&xt)[E]) {
#pragma unroll
      for (int e = 0; e < E; ++e) xt[e] = !has_problem ? 0.0 : (fresh ? x[e] : AR::nmadd(stp, d[e], wa[e]));
    };
    {
      double xt[E];
      trial_point(xt);
#pragma unroll
      for (int e = 0; e < E; ++e) xrow[e] = xt[e];
    }
    // barrier A, and "does any slot still hold a problem": one flag per wavefront in LDS, one barrier (the library's
    // workgroup-wide OR costs two barriers and a reduction through LDS per pass)
    {
      const bool wave_alive = __builtin_amdgcn_ballot_w64(has_problem) != 0;
      if (lane == 0) alive_flags[wave] = wave_alive ? 1 : 0;
      __syncthreads();
      int alive = 0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) alive |= alive_flags[w];
      if (alive == 0) break;  // every slot idle and the queue drained
    }

    MI355_LPHASE(1);  // r = A X - Y
    // ---- (3) r = A X - Y: this wavefront's 16 residual rows of all 16 problems ------------------
    {
      const int p = lane & 15, kq = lane >> 4;
      constexpr int kTiles = (kJointRows / 16) / kWaves;   // residual row tiles per wavefront: 1 or 2
      const double* const bf = X_lds + p * kJointPitchX + kq;
      double yv[kTiles][4];
      v4d acc[kTiles];
#pragma unroll
      for (int t = 0; t < kTiles; ++t) {
        const int row0 = 16 * (kTiles * wave + t);
#pragma unroll
        for (int r = 0; r < 4; ++r) yv[t][r] = Y_lds[p * kJointPitchR + row0 + kq + 4 * r];
        acc[t] = v4d{0.0, 0.0, 0.0, 0.0};
      }
#pragma unroll
      for (int kb = 0; kb < kJointCols / 4; ++kb) {
        const double b = bf[4 * kb];
#pragma unroll
        for (int t = 0; t < kTiles; ++t)   // independent accumulator chains: the matrix pipe issues back to back
          acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_lds[(16 * (kTiles * wave + t) + p) * kJointPitchA + kq + 4 * kb], b,
                                                        acc[t], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) R_lds[p * kJointPitchR + 16 * (kTiles * wave + t) + kq + 4 * r] = acc[t][r] - yv[t][r];
    }
    MI355_LPHASE(2);  // barrier B
    __syncthreads();
    MI355_LPHASE(3);  // G = A^T R (wavefronts 0..3)
    // ---- (4) G = A^T R: gradient coordinates 16t .. 16t+15 of all 16 problems --------------------
    // The 128 rows are summed as TWO chains of 64 (rows 0..63 and 64..127) that are added at the pick-up.  With eight
    // wavefronts every one of them takes a (tile, half) pair — 16 MFMAs each instead of 32 on four wavefronts while
    // the other four wait at the barrier; the upper halves land in the X tile, which is dead once r = A X - Y is
    // done.  With four wavefronts each runs its tile's two chains as independent accumulators.
    {
      const int p = lane & 15, kq = lane >> 4;
      constexpr int kHalfSteps = kJointRows / 8;   // MFMA steps (4 rows each) per half
      if constexpr (kWaves == 8) {
        const int tile = wave & 3, half = wave >> 2;
        const double* const af = A_lds + (kq + 4 * kHalfSteps * half) * kJointPitchA + 16 * tile + p;
        const double* const bf = R_lds + p * kJointPitchR + kq + 4 * kHalfSteps * half;
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
        for (int kb = 0; kb < kHalfSteps; ++kb)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[4 * kb * kJointPitchA], bf[4 * kb], acc, 0, 0, 0);
        double* const out = half ? X_lds : G_lds;
#pragma unroll
        for (int r = 0; r < 4; ++r) out[p * kJointPitchX + 16 * tile + kq + 4 * r] = acc[r];
      } else {
        const double* const af = A_lds + kq * kJointPitchA + 16 * wave + p;
        const double* const bf = R_lds + p * kJointPitchR + kq;
        v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
        for (int kb = 0; kb < kHalfSteps; ++kb) {
          acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[4 * kb * kJointPitchA], bf[4 * kb], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(af[4 * (kb + kHalfSteps) * kJointPitchA], bf[4 * (kb + kHalfSteps)], acc1,
                                                      0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          G_lds[p * kJointPitchX + 16 * wave + kq + 4 * r] = acc0[r];
          X_lds[p * kJointPitchX + 16 * wave + kq + 4 * r] = acc1[r];
        }
      }
    }
    // ||r||^2 and ||x||^2 only need R: done here so that the wavefronts without a gradient tile overlap
    // them with the second matrix phase
    double f1, xx;
    {
      double xt[E];
      trial_point(xt);
      double rr[RPL];
#pragma unroll
      for (int q = 0; q < RPL; ++q) {
        const double r = rrow[q];
        rr[q] = r * r;
      }
      f1 = seg_sum<W>(lane_tree_sum<RPL>(rr));
      xx = seg_dot<W, E, AR>(xt, xt);
    }
    MI355_LPHASE(4);  // barrier C
    __syncthreads();
    MI355_LPHASE(5);  // f, g pick-up + line-search logic
    if (!has_problem) continue;

    // ---- (5) this segment's value and gradient at xt ----------------------------------------------
    {
      double xt[E];
      trial_point(xt);
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        g[e] = (j < n) ? 2.0 * (grow[e] + xrow[e]) + lambda * (2.0 * xt[e]) : 0.0;   // the two row halves of A^T r
      }
      f = f1 + lambda * xx;
    }

    bool start_iteration = false;
    if (fresh) {
      // ---- Solver::Minimize prologue (solver.h:189-192), InitializeSolver, Progress -----------
      fresh = false;
      nfev = 1;
      sum_k = 0;
      mem_count = 0;
      scaling_factor = 1.0;
      num_iterations = 0;
      x_delta_violations = 0;
      f_delta_violations = 0;
      x_delta = f_delta = gradient_norm = 0.0;
      past_init = false;
      past_pos = 0;
      xinf_bound = seg_amax<W, E>(x);
      start_iteration = true;
    } else {
      // ---- cvsrch after an evaluation (:196-252) --------------------------------------------------
      ls_nfev++;
      const double dg = -seg_dot<W, E, AR>(g, d);  // g.s
      const double ftest1 = finit + stp * dgtest;
      int info = 0;
      if ((brackt & ((stp <= stmin) | (stp >= stmax))) | (infoc == 0)) info = 6;
      if ((stp == stpmax) & (f <= ftest1) & (dg <= dgtest)) info = 5;
      if ((stp == stpmin) & ((f > ftest1) | (dg >= dgtest))) info = 4;
      if (ls_nfev >= maxfev) info = 3;
      if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
      if ((f <= ftest1) & (__builtin_fabs(dg) <= gtol * (-dginit))) info = 1;
      if (info == 0) {
        if (stage1 & (f <= ftest1) & (dg >= dmin(ftol, gtol) * dginit)) stage1 = false;
        const bool modified = stage1 & (f <= fx) & (f > ftest1);
        StepInterval iv;
        iv.stx = stx; iv.sty = sty; iv.stp = stp; iv.brackt = brackt; iv.info = infoc; iv.rc = 0;
        iv.fx = modified ? fx - stx * dgtest : fx;
        iv.fy = modified ? fy - sty * dgtest : fy;
        iv.dx = modified ? dgx - dgtest : dgx;
        iv.dy = modified ? dgy - dgtest : dgy;
        const double fm = modified ? f - stp * dgtest : f;
        const double dgm = modified ? dg - dgtest : dg;
        iv = mt_cstep(iv, fm, dgm, stmin, stmax);
        stx = iv.stx; sty = iv.sty; stp = iv.stp; brackt = iv.brackt; infoc = iv.info;
        fx = modified ? iv.fx + stx * dgtest : iv.fx;
        fy = modified ? iv.fy + sty * dgtest : iv.fy;
        dgx = modified ? iv.dx + dgtest : iv.dx;
        dgy = modified ? iv.dy + dgtest : iv.dy;
        if (brackt) {
          if (__builtin_fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
          width1 = width;
          width = __builtin_fabs(sty - stx);
        }
        ls_prepare();
        continue;  // next pass evaluates wa - stp * d
      }
      // line search finished: the accepted point is the one just evaluated
      nfev += static_cast<unsigned>(ls_nfev);
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = AR::nmadd(stp, d[e], wa[e]);
    }

    MI355_LPHASE(6);  // end of iteration (update, stopping tests, results) + two-loop + search set-up
    // ============ from here: end of an iteration (unless fresh), then the start of the next ============
    bool finish_iteration = !start_iteration;
    while (true) {
      if (finish_iteration) {
        // ---- rest of OptimizationStep (lbfgs.h:239-298); wa / gp hold the iterate the step started from
        double sv[E], yv[E];
        double s_inf = 0.0;  // ||x+ - x||_inf (Progress::Update's x_delta; the curvature test uses it first)
        if (!__builtin_isfinite(f)) {  // return current (:239-241)
          f = fprev;
#pragma unroll
          for (int e = 0; e < E; ++e) {
            x[e] = wa[e];
            g[e] = gp[e];
            sv[e] = 0.0;
          }
        } else {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            sv[e] = x[e] - wa[e];  // :248
            yv[e] = g[e] - gp[e];  // :249
          }
          s_inf = seg_amax<W, E>(sv);
          const double sy = seg_dot<W, E, AR>(sv, yv);   // :265
          const double yy = seg_dot<W, E, AR>(yv, yv);   // :290
          bool accept = false;                       // :266-267, see lbfgs_kernel.hpp (ss only when its bound cannot tell)
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
                accept = sy > eps * __builtin_sqrt(ss) * __builtin_sqrt(yy);
              }
            }
          }
          if (accept) {                              // :267-280, chronological registers
            if (mem_count < m) mem_count++;
#pragma unroll
            for (int i = 0; i + 1 < MR; ++i) {
#pragma unroll
              for (int e = 0; e < E; ++e) {
                Sr[i][e] = Sr[i + 1][e];
                Yr[i][e] = Yr[i + 1][e];
              }
              Rr[i] = Rr[i + 1];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
              Sr[MR - 1][e] = sv[e];
              Yr[MR - 1][e] = yv[e];
            }
            Rr[MR - 1] = (__builtin_fabs(sy) < eps) ? 0.0 : 1.0 / sy;  // 0 = the pair both loops skip (:165/:189)
          }
          if (yy > eps) {                            // :289-298
            const double temp_scaling = sy / yy;
            if (__builtin_isfinite(temp_scaling) && __builtin_fabs(temp_scaling) <= 1e7)
              scaling_factor = dmax(temp_scaling, eps);
          }
        }
        // ---- Progress::Update (progress.h:153-327) ------------------------------------------------
        num_iterations++;
        f_delta = __builtin_fabs(f - fprev);
        x_delta = s_inf;
        gradient_norm = seg_amax<W, E>(g);
        xinf_bound = (xinf_bound + x_delta) * (1.0 + 4.0 * eps);
        const mi355_lbfgs_stop& st = a.stop;
        int status = MI355_STATUS_CONTINUE;
        bool decided = false;
        if ((st.num_iterations > 0) && (num_iterations > st.num_iterations)) {
          status = MI355_STATUS_ITERATION_LIMIT;
          decided = true;
        }
        if (!decided) {
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
        if (!decided) {
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
        if (!decided && st.past > 0) {
          const int pw = st.past;
          if (!past_init) {
            if (sl < pw) past_f[sl] = f;
            past_init = true;
            past_pos = 0;
            segment_lds_fence();
          }
          if (static_cast<int>(num_iterations) > pw) {
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
            past_pos = (past_pos + 1 == pw) ? 0 : past_pos + 1;
          }
        }
        if (!decided && st.gradient_norm > 0) {
          if (st.gradient_norm_relative) {
            if (gradient_norm < st.gradient_norm * dmax(1.0, xinf_bound)) {
              const double xinf = seg_amax<W, E>(x);
              xinf_bound = xinf;
              if (gradient_norm < st.gradient_norm * dmax(1.0, xinf)) status = MI355_STATUS_GRADIENT_NORM_VIOLATION;
            }
          } else if (gradient_norm < st.gradient_norm) {
            status = MI355_STATUS_GRADIENT_NORM_VIOLATION;
          }
        }
        if (status == MI355_STATUS_CONTINUE && a.hessian_condition_fires)   // progress.h:318-325 (Second mode)
          status = MI355_STATUS_HESSIAN_CONDITION_VIOLATION;
        if (status != MI355_STATUS_CONTINUE) {
          // ---- results of this problem (solver.h:223); the slot refills at the top of the next pass
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
          has_problem = false;
          break;
        }
      }
      finish_iteration = true;

      // ======================= Lbfgs::OptimizationStep, first half (lbfgs.h:89-232) ===================
#pragma unroll
      for (int e = 0; e < E; ++e) d[e] = g[e];  // :145
      const int k = mem_count;
      sum_k += static_cast<unsigned>(k);
      {
#pragma unroll
        for (int t = 0; t < MR; ++t) {            // newest -> oldest (:157-171)
          if (t < k) {
            const double alpha = Rr[MR - 1 - t] * seg_dot<W, E, AR>(Sr[MR - 1 - t], d);
            if constexpr (kAlphaInLds) {
              al_lds[t * (kWaves * kWave)] = alpha;
            } else {
              al_reg[t] = alpha;
            }
#pragma unroll
            for (int e = 0; e < E; ++e) d[e] = AR::nmadd(alpha, Yr[MR - 1 - t][e], d[e]);
          }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = sl * E + e;
          d[e] = (a.precond != nullptr) ? ((j < n) ? a.precond[j] : 0.0) * d[e]   // :177-179
                                        : d[e] * scaling_factor;                  // :181
        }
#pragma unroll
        for (int t = MR - 1; t >= 0; --t) {       // oldest -> newest (:185-196)
          if (t < k) {
            const double beta = Rr[MR - 1 - t] * seg_dot<W, E, AR>(Yr[MR - 1 - t], d);
            double alt;
            if constexpr (kAlphaInLds) {
              alt = al_lds[t * (kWaves * kWave)];
            } else {
              alt = al_reg[t];
            }
            const double c = alt - beta;
#pragma unroll
            for (int e = 0; e < E; ++e) d[e] = AR::madd(Sr[MR - 1 - t][e], c, d[e]);
          }
        }
      }
      const double descent_direction = -seg_dot<W, E, AR>(g, d);  // :199
      dginit = descent_direction;                              // = g.s with s = -d, bit for bit
      double alpha_init = 1.0;                                 // :207-213
      if (mem_count == 0) {
        const double dn = __builtin_sqrt(seg_dot<W, E, AR>(d, d));
        alpha_init = (dn > eps) ? 1.0 / dn : 1.0;
      }
      bool invalid_direction;                                  // :214-224, see lbfgs_kernel.hpp
      if (__builtin_isfinite(descent_direction) &&
          descent_direction <= -eps * (eps * dmax(1.0, n_as_double * xinf_bound))) {
        invalid_direction = false;
      } else {
        const double relative_eps = eps * dmax(1.0, __builtin_sqrt(seg_dot<W, E, AR>(x, x)));
        invalid_direction = !__builtin_isfinite(descent_direction) || descent_direction > -eps * relative_eps;
      }
      if (invalid_direction) {
#pragma unroll
        for (int e = 0; e < E; ++e) d[e] = -g[e];
        mem_count = 0;
        const double gg = seg_dot<W, E, AR>(g, g);
        const double gn = __builtin_sqrt(gg);
        alpha_init = (gn > eps) ? 1.0 / gn : 1.0;
        dginit = gg;  // s = -d = g: not a descent direction, the search returns at once (quirk Q1)
      }
      // the iterate this step starts from (for s, y and the non-finite bail-out)
      fprev = f;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        wa[e] = x[e];
        gp[e] = g[e];
      }
      if (dginit >= 0.0) continue;  // cvsrch :152-156: x, f, g untouched — finish the iteration right away
      // cvsrch set-up (:158-177)
      brackt = false;
      stage1 = true;
      finit = f;
      dgtest = ftol * dginit;
      width = stpmax - stpmin;
      width1 = 2.0 * width;
      stx = 0.0; fx = finit; dgx = dginit;
      sty = 0.0; fy = finit; dgy = dginit;
      stp = alpha_init;
      ls_nfev = 0;
      infoc = 1;
      ls_prepare();
      break;  // next pass evaluates wa - stp * d
    }
  }
#ifdef MI355_LBFGS_PHASE_TIMING
  MI355_LPHASE(0);
  if (lane == 0 && a.profile != nullptr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) atomicAdd(a.profile + i, lphase_cycles[i]);
  }
#endif
}

}  // namespace mi355
