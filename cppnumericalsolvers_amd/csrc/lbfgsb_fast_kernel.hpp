// lbfgsb_fast_kernel.hpp — box-constrained L-BFGS-B under the RELAXED-ALGEBRA policy (mi355_lbfgs_desc.arithmetic =
// MI355_ARITH_FMA on the mi355_lbfgsb_* entry points; lbfgsb_kernel.hpp is the reference-order build).
//
// The iteration is the reference's (include/cppoptlib/solver/lbfgsb.h: OptimizationStep :141-238, Minimize :247-292,
// GetGeneralizedCauchyPoint :318-430, SubspaceMinimization :459-515, FindAlpha :435-457, SolveM :311-316); the small
// dense algebra of the compact representation is re-derived for a 16-lane wavefront segment:
//   * fixed layout W = [Y_0..Y_{M-1} | S_0..S_{M-1}] over a RING of m slots (no shifting, :216-217), unused slots are
//     zero columns and identity rows of the 2M x 2M matrices; theta multiplies the S half where it is consumed;
//   * lane a < 2M owns column a of W and row a of every 2M x 2M matrix.  A length-n inner product against column a
//     (p = W^T d, WZ r, the S^T Y / S^T S / Y^T Y entries of a new pair) is a chain of fused multiply-adds that lane
//     a runs over its column in LDS against the vector staged in LDS (four interleaved partial chains) — no
//     cross-lane reduction at all; products W * (2M-vector) are chains over the 2M columns by the lane that owns
//     the coordinates;
//   * MM = [[-D, L^T], [L, theta S^T S]] (:227-232) is factored WITHOUT pivoting (Y block diagonal and negative, its
//     Schur complement positive definite — the factorisation of the original Fortran code), reciprocal pivots;
//   * M^-1 c and M^-1 p in the breakpoint loop follow from linearity, so a breakpoint costs one distributed solve
//     (M^-1 W.row(b)) instead of three and the subspace step needs none for M^-1 c;
//   * v of :486-500 is ONE elimination with K = MM - theta^-1 WZ WZ^T, assembled as K0 + theta^-1 W_A^T W_A over the
//     ACTIVE coordinates, K0 = MM - theta^-1 W^T W = [[-D - Y^T Y / theta, -R^T], [-R, 0]] cached per history update.
// Every operation is restated by the CPU twin oracle/lbfgsb_fast_oracle.hpp (tests only), which the kernel equals
// bit for bit; the policy is accepted against the reference binary at 1e-6 on x* and f*.
#pragma once
#include "lbfgsb_kernel.hpp"

namespace mi355 {

// Four and more coordinates per lane: the iterate at the start of the step (x_delta), the point / gradient the line
// search starts from (the new pair) and the box are kept in LDS instead of 10 E registers — with them in registers the
// E = 4 kernels spilled 56 ... 124 bytes per lane at their 256-register budget.  Same values, same arithmetic.
__host__ __device__ constexpr bool lbfgsb_fast_staged(int P) { return P >= 64; }
// lanes per problem: the 2M rows of the compact representation take one lane each — one DPP row of sixteen up to M = 8,
// two rows (thirty-two lanes) for M = 9, 10
__host__ __device__ constexpr int lbfgsb_fast_lanes(int M) { return (2 * M > 16) ? 32 : 16; }
template <int M>
__host__ __device__ constexpr int lbfgsb_fast_lds_doubles_per_problem(int P, int objective_scratch) {
  // history [2M][P + 2], S^T Y / S^T S / Y^T Y (padded to an even count), K0 [2M][2M], an n-vector and a 2M-vector
  // of staging, the plateau ring, the cached reciprocals 1 / (s_a . y_a) of the ring slots (padded to 16)
  // (a 2M-vector of staging and the pivot reciprocals: one entry per lane of the segment — 16 lanes, or 32 for 2M > 16)
  return 2 * M * (P + 2) + ((3 * M * M + 1) & ~1) + 4 * M * M + P + lbfgsb_fast_lanes(M) + MI355_LBFGS_MAX_PAST +
         lbfgsb_fast_lanes(M) + (lbfgsb_fast_staged(P) ? 3 * P : 0) + objective_scratch;
}
// read-only LDS shared by the segments of a workgroup after their per-problem areas: the box [lower | upper]
__host__ __device__ constexpr int lbfgsb_fast_shared_tail_doubles(int P) { return lbfgsb_fast_staged(P) ? 2 * P : 0; }

struct d2 {
  double x, y;
};
__device__ __forceinline__ d2 ld2(const double* p) {  // one ds_read_b128 (all staging arrays are 16-byte aligned)
  typedef double v2 __attribute__((ext_vector_type(2)));
  const v2 v = *reinterpret_cast<const v2*>(p);
  return d2{v.x, v.y};
}

// sum_i col[i] * vec[i] over the P padded coordinates: four interleaved fused chains, added pairwise
template <int P>
__device__ __forceinline__ double chain4(const double* col, const double* vec) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
  for (int i = 0; i < P; i += 4) {
    const d2 c01 = ld2(col + i), c23 = ld2(col + i + 2);
    const d2 v01 = ld2(vec + i), v23 = ld2(vec + i + 2);
    a0 = __builtin_fma(c01.x, v01.x, a0);
    a1 = __builtin_fma(c01.y, v01.y, a1);
    a2 = __builtin_fma(c23.x, v23.x, a2);
    a3 = __builtin_fma(c23.y, v23.y, a3);
  }
  return (a0 + a1) + (a2 + a3);
}
// the same column against two vectors
template <int P>
__device__ __forceinline__ void chain4x2(const double* col, const double* va, const double* vb, double& ra, double& rb) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
#pragma unroll
  for (int i = 0; i < P; i += 4) {
    const d2 c01 = ld2(col + i), c23 = ld2(col + i + 2);
    const d2 u01 = ld2(va + i), u23 = ld2(va + i + 2);
    const d2 w01 = ld2(vb + i), w23 = ld2(vb + i + 2);
    a0 = __builtin_fma(c01.x, u01.x, a0);
    a1 = __builtin_fma(c01.y, u01.y, a1);
    a2 = __builtin_fma(c23.x, u23.x, a2);
    a3 = __builtin_fma(c23.y, u23.y, a3);
    b0 = __builtin_fma(c01.x, w01.x, b0);
    b1 = __builtin_fma(c01.y, w01.y, b1);
    b2 = __builtin_fma(c23.x, w23.x, b2);
    b3 = __builtin_fma(c23.y, w23.y, b3);
  }
  ra = (a0 + a1) + (a2 + a3);
  rb = (b0 + b1) + (b2 + b3);
}

// The lane predicates of the elimination steps (sl > kk, sl == kk, sl < j) are loop invariants of the whole kernel; left
// alone the compiler keeps all ~30 of them in scalar-register pairs across the iteration loop, which is more than a
// wavefront has, and spills them to vector-register lanes.  Made opaque, each step recomputes its predicate with one
// v_cmp instead.
#ifndef MI355_LBFGSB_FAST_OPAQUE_LANE
#define MI355_LBFGSB_FAST_OPAQUE_LANE 1
#endif
__device__ __forceinline__ int step_lane(int sl) {
#if MI355_LBFGSB_FAST_OPAQUE_LANE
  asm volatile("" : "+v"(sl));
#endif
  return sl;
}

// Unpivoted LU of MM, row `sl` per lane.  In: row = the lane's row of MM.  Out: the factors in the form the solves
// use — lz[j] = L(sl, j) below the diagonal and ZERO elsewhere (j = 0..K2-2), uz[j - M] = U(sl, j) above the diagonal
// and zero elsewhere (j = M..K2-1; columns j < M of U are zero above the diagonal because the Y block of MM is
// diagonal, which is also why the pivot row's columns kk+1..M-1 are skipped for kk < M), dinv = 1 / U(sl, sl).  With
// the zeros in place a substitution step is "broadcast, one fused multiply-add" on every lane: no lane predicate, no
// exec-mask juggling, no scalar registers held for the masks.
// ypinv[kk], kk < M: the reciprocal of pivot kk, known without a division — the Y block of MM is diagonal, so the pivot
// of step kk < M is MM(kk, kk) itself: -(s_kk . y_kk) for a slot in use (its reciprocal is cached when the pair
// enters the ring: IEEE division commutes with negation, the bits are those of 1 / MM(kk, kk)) and 1 for an empty one.
template <int K2, int M, int W = 16>
__device__ __forceinline__ void fast_factor_mm(double (&row)[K2], double (&lz)[K2 - 1], double (&uz)[K2 - M], double& dinv,
                                               int sl_in, const double* ypinv) {
  static_for<0, K2>([&](auto ic) {
    constexpr int kk = decltype(ic)::value;
    const int sl = step_lane(sl_in);
    double rinv;
    if constexpr (kk < M) {
      rinv = ypinv[kk];
    } else {
      rinv = 1.0 / row_bcast<W, kk>(row[kk]);
    }
    dinv = (sl == kk) ? rinv : dinv;
    const double mz = (sl > kk) ? row[kk] * rinv : 0.0;
    if constexpr (kk < K2 - 1) lz[kk] = mz;
    constexpr int b0 = (kk < M) ? M : kk + 1;
#pragma unroll
    for (int b = b0; b < K2; ++b) row[b] = __builtin_fma(-mz, row_bcast<W, kk>(row[b]), row[b]);
  });
#pragma unroll
  for (int j = M; j < K2; ++j) uz[j - M] = (step_lane(sl_in) < j) ? row[j] : 0.0;
}
// x := MM^-1 x for a distributed vector (lane a holds x_a)
template <int K2, int M, int W = 16>
__device__ __forceinline__ double fast_solve_mm(const double (&lz)[K2 - 1], const double (&uz)[K2 - M], double dinv,
                                                double x) {
  static_for<0, K2 - 1>([&](auto ic) {
    constexpr int j = decltype(ic)::value;
    x = __builtin_fma(-row_bcast<W, j>(x), lz[j], x);
  });
  static_for<0, K2 - M>([&](auto ic) {
    constexpr int j = K2 - 1 - decltype(ic)::value;
    x = __builtin_fma(-row_bcast<W, j>(x * dinv), uz[j - M], x);
  });
  return x * dinv;
}
// K v = rhs: unpivoted elimination with the right-hand side riding along, then back substitution
template <int K2, int W = 16>
__device__ __forceinline__ double fast_solve_k(double (&row)[K2], double rv, int sl_in) {
  double dinv = 1.0;
  static_for<0, K2>([&](auto ic) {
    constexpr int kk = decltype(ic)::value;
    const int sl = step_lane(sl_in);
    const double rinv = 1.0 / row_bcast<W, kk>(row[kk]);
    dinv = (sl == kk) ? rinv : dinv;
    const double mz = (sl > kk) ? row[kk] * rinv : 0.0;
#pragma unroll
    for (int b = kk + 1; b < K2; ++b) row[b] = __builtin_fma(-mz, row_bcast<W, kk>(row[b]), row[b]);
    rv = __builtin_fma(-mz, row_bcast<W, kk>(rv), rv);
  });
  static_for<0, K2 - 1>([&](auto ic) {
    constexpr int j = K2 - 1 - decltype(ic)::value;
    const int sl = step_lane(sl_in);
    const double yj = row_bcast<W, j>(rv * dinv);
    rv = (sl < j) ? __builtin_fma(-yj, row[j], rv) : rv;
  });
  return rv * dinv;
}

template <int E, class Obj, int M, int W = lbfgsb_fast_lanes(M)>
__global__ __launch_bounds__(64, (M > 5 || E >= 8) ? 1 : 2) void lbfgsb_fast_kernel(const LbfgsbArgs args) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  static_assert(W == lbfgsb_fast_lanes(M), "segment width follows from the capacity");
  constexpr int P = W * E;
  constexpr int K2 = 2 * M;
  constexpr int PITCH = P + 2;  // lane a reads column a two doubles at a time: a 16-byte bank offset per lane
  static_assert(K2 <= W, "the 2M rows of the compact representation must fit the lanes of one segment");
  using AR = ArithFma;
  constexpr double kMax = 1.7976931348623157e308;
  const SolveArgs& a = args.s;

  const int lane = threadIdx.x & (kWave - 1);
  const int seg = lane / W;
  const int sl = lane % W;
  const int n = a.n;
  const int mcap = a.m;
  const bool row_lane = sl < K2;
  const int ra = row_lane ? sl : K2 - 1;  // idle lanes (2M < 16) shadow the last row; their results are masked

  double* const base = lds + Obj::shared_lds_doubles() + seg * lbfgsb_fast_lds_doubles_per_problem<M>(P, Obj::kLdsDoubles);
  double* const Wc = base;                                // [K2][PITCH]: Y slots, then S slots
  double* const Amat = Wc + K2 * PITCH;                   // [M][M]  s_i . y_j
  double* const SSmat = Amat + M * M;                     //         s_i . s_j
  double* const YYmat = SSmat + M * M;                    //         y_i . y_j
  double* const K0m = Amat + ((3 * M * M + 1) & ~1);      // [K2][K2] row major
  double* const vbuf = K0m + K2 * K2;                     // [P]
  double* const ubuf = vbuf + P;                          // [W]
  double* const past_f = ubuf + W;
  double* const ypinv = past_f + MI355_LBFGS_MAX_PAST;    // [W]: pivot reciprocals of the Y block (fast_factor_mm)
  constexpr bool kStaged = lbfgsb_fast_staged(P);
  double* const stage = ypinv + W;                        // kStaged: [3][P] = x at the start of the step, xcur, gcur
  const double* const mycol = Wc + ra * PITCH;

  Obj obj;
  obj.load(a.obj_params, n, sl, stage + (kStaged ? 3 * P : 0), lds);
  if constexpr (Obj::shared_lds_doubles() > 0) {
    obj.fill_shared(lds, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
    __syncthreads();
  }
  const long long queue_length = a.count_dev ? static_cast<long long>(*a.count_dev) : a.B;
  const unsigned long long stop_num_iterations = a.stop.num_iterations;
  const double stop_gradient_norm = a.stop.gradient_norm;

  // the box: registers, or (kStaged) one LDS copy per workgroup behind the per-problem areas
  double lo_reg[kStaged ? 1 : E], hi_reg[kStaged ? 1 : E];
  double* const box = lds + Obj::shared_lds_doubles() +
                      (kWave / W) * lbfgsb_fast_lds_doubles_per_problem<M>(P, Obj::kLdsDoubles);
  if constexpr (kStaged) {
    for (int j = static_cast<int>(threadIdx.x); j < P; j += static_cast<int>(blockDim.x)) {
      box[j] = (j < n) ? args.lower[j] : 0.0;
      box[P + j] = (j < n) ? args.upper[j] : 0.0;
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      lo_reg[e] = (j < n) ? args.lower[j] : 0.0;
      hi_reg[e] = (j < n) ? args.upper[j] : 0.0;
    }
  }
  auto lo_at = [&](int e) -> double {
    if constexpr (kStaged) return box[sl * E + e];
    else return lo_reg[e];
  };
  auto hi_at = [&](int e) -> double {
    if constexpr (kStaged) return box[P + sl * E + e];
    else return hi_reg[e];
  };
  auto clip = [&](const double (&v)[E], double (&out)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = dmax(dmin(v[e], hi_at(e)), lo_at(e));
  };
  auto differs = [&](const double (&u)[E], const double (&v)[E]) {
    int dflag = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) dflag |= (sl * E + e < n && u[e] != v[e]) ? 1 : 0;
    return seg_max<W>(static_cast<double>(dflag)) != 0.0;
  };
  // W bits of a wavefront-wide vote: the lanes of the caller's segment
  auto seg_bits = [&](bool pred) {
    const unsigned long long bal = __ballot(pred);
    return static_cast<unsigned>(bal >> (seg * W)) & ((W == 32) ? 0xffffffffu : 0xffffu);
  };

  long long prob = 0;
  bool need_fetch = true;
  double x[E], g[E];
  double f = 0.0;
  unsigned nfev = 0, sum_k = 0;
  int k = 0, head = 0;
  double theta = 1.0, theta_inverse = 1.0, ws = 1.0;  // ws: the lane's scale of W (1 for a Y column, theta for an S column)
  double mm_lz[K2 - 1], mm_uz[K2 - M];  // the factors of MM in solve form (fast_factor_mm)
  double mm_dinv = 1.0;
  double last_pg = 0.0;
  unsigned num_iterations = 0;
  int x_delta_violations = 0, f_delta_violations = 0;
  double x_delta = 0.0, f_delta = 0.0, gradient_norm = 0.0;
  int status = MI355_STATUS_NOT_STARTED;
  bool past_init = false;
  int past_pos = 0;
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = g[e] = 0.0;
#pragma unroll
  for (int j = 0; j < K2 - 1; ++j) mm_lz[j] = 0.0;
#pragma unroll
  for (int j = 0; j < K2 - M; ++j) mm_uz[j] = 0.0;

  // out_j = sum_a W(j, a) ubuf[a] for the lane's coordinates (raw columns: the scale of the S half is in ubuf)
  auto w_times_ubuf = [&](double (&out)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = 0.0;
    static_for<0, K2>([&](auto ic) {
      constexpr int col = decltype(ic)::value;
      const double ua = ubuf[col];
#pragma unroll
      for (int e = 0; e < E; ++e) out[e] = __builtin_fma(Wc[col * PITCH + sl * E + e], ua, out[e]);
      // (kStaged: at most four columns of reads in flight — left alone the scheduler hoists all 2M E of them, which is
      // the register peak of the four-coordinates-per-lane kernels)
      if constexpr (kStaged && (col % 4 == 3)) __builtin_amdgcn_sched_barrier(0);
    });
  };

#ifdef MI355_LBFGSB_PHASE_TIMING
  unsigned long long phase_cycles[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) phase_cycles[i] = 0;
  unsigned long long phase_t0 = __builtin_readcyclecounter();
  int phase_cur = 0;
#endif
  while (true) {
    MI355_PHASE(0);  // fetch / prologue
    if (need_fetch) {
      unsigned long long nxt = 0;
      if (sl == 0) nxt = atomicAdd(a.next_problem, 1ULL);
      const unsigned lo32 = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt & 0xffffffffULL)));
      const unsigned hi32 = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt >> 32)));
      prob = static_cast<long long>((static_cast<unsigned long long>(hi32) << 32) | lo32);
      if (prob >= queue_length) break;
      const KernargSolveArgs ca = cold_args();
      {
        const int* const map = ca->problem_map;
        if (map != nullptr) prob = map[prob];
      }
      need_fetch = false;
      {
        const double* const x0p = ca->x0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = sl * E + e;
          x[e] = (j < n) ? x0p[prob * n + j] : 0.0;
        }
      }
      obj.begin_problem(ca->per_problem, prob, ca->per_problem_stride, sl);
      // InitializeSolver (:120-139): empty ring (zero columns), identity factors
      static_for<0, K2>([&](auto ic) {
        constexpr int col = decltype(ic)::value;
#pragma unroll
        for (int e = 0; e < E; ++e) Wc[col * PITCH + sl * E + e] = 0.0;
      });
#pragma unroll
      for (int r = 0; r < (3 * M * M + W - 1) / W; ++r)
        if (sl + r * W < 3 * M * M) Amat[sl + r * W] = 0.0;
      {
        // (opaque lane index: otherwise the 2M select results are hoisted out of the problem loop as loop invariants
        // and held — or spilled — for the whole kernel; this path runs once per problem)
        const int slo = step_lane(sl);
#pragma unroll
        for (int b = 0; b < K2; ++b)
          if (slo < K2) K0m[slo * K2 + b] = (b == slo) ? 1.0 : 0.0;
      }
#pragma unroll
      for (int j = 0; j < K2 - 1; ++j) mm_lz[j] = 0.0;   // MM = identity
#pragma unroll
      for (int j = 0; j < K2 - M; ++j) mm_uz[j] = 0.0;
      ypinv[sl] = 1.0;
      segment_lds_fence();
      mm_dinv = 1.0;
      f = obj.template eval_fma<W, E>(x, g, n, sl);            // Minimize prologue (:253)
      nfev = 1;
      sum_k = 0;
      k = 0;
      head = 0;
      theta = theta_inverse = ws = 1.0;
      last_pg = 0.0;
      num_iterations = 0;
      x_delta_violations = 0;
      f_delta_violations = 0;
      x_delta = f_delta = gradient_norm = 0.0;
      status = MI355_STATUS_NOT_STARTED;
      past_init = false;
      past_pos = 0;
    }

    // ============================ OptimizationStep (:141-238) ===========================
    MI355_PHASE(1);  // clip + projected gradient
    double xs[kStaged ? 1 : E];
    const double f_state = f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if constexpr (kStaged) stage[sl * E + e] = x[e];   // (written and read back by the same lane: no fence)
      else xs[e] = x[e];
    }
    {
      double xc0[E];
      clip(x, xc0);                                                   // :148
      if (differs(xc0, x)) {                                          // :151-153
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = xc0[e];
        f = obj.template eval_fma<W, E>(x, g, n, sl);
        nfev++;
      }
    }
    sum_k += k;
    {  // projected gradient sup-norm (:105-118, :165-166)
      double t[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        double gj = g[e];
        if (x[e] <= lo_at(e) && gj > 0) gj = 0.0;
        if (x[e] >= hi_at(e) && gj < 0) gj = 0.0;
        t[e] = (sl * E + e < n) ? __builtin_fabs(gj) : 0.0;
      }
      last_pg = seg_max<W>(lane_max<E>(t));
    }

    // ---- generalized Cauchy point (:318-430) -------------------------------------------
    MI355_PHASE(2);  // Cauchy point: breakpoints, p = W^T d, first solve
    double xc[E], d[E], tb[E];
    bool pending[E];
    double Mc = 0.0;  // M^-1 c, distributed (c itself is never needed)
    {
      int npos = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        d[e] = -g[e];
        double tmp = kMax;
        if (g[e] != 0) {
          tmp = (x[e] - ((g[e] < 0) ? hi_at(e) : lo_at(e))) / g[e];   // one quotient: the bound is selected first
          if (tmp == 0) d[e] = 0;
        }
        tb[e] = tmp;
        xc[e] = x[e];
        pending[e] = (j < n) && (tmp > 0);
        npos += pending[e] ? 1 : 0;
        if (j >= n) d[e] = 0.0;
      }
      const bool any_positive = seg_max<W>(static_cast<double>(npos)) > 0.0;
#pragma unroll
      for (int e = 0; e < E; ++e) vbuf[sl * E + e] = d[e];
      segment_lds_fence();
      double p_vec = row_lane ? ws * chain4<P>(mycol, vbuf) : 0.0;     // p = W^T d (:353)
      segment_lds_fence();
      double f_prime = -seg_dot<W, E, AR>(d, d);                       // :357
      double Mp = fast_solve_mm<K2, M, W>(mm_lz, mm_uz, mm_dinv, p_vec);
      const double pMp = seg_sum<W>(p_vec * Mp);
      double f_doubleprime = (-theta) * f_prime - pMp;                // :361-362
      f_doubleprime = dmax(1e-12, f_doubleprime);
      const double f_dp_orig = f_doubleprime;
      double dt_min = -f_prime / f_doubleprime;
      double t_old = 0.0;

      auto select_min = [&](const bool (&cand)[E], int& b_out, double& t_out) {
        double bt = kMax;
        int bj = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = sl * E + e;
          if (cand[e] && (tb[e] < bt || (tb[e] == bt && j < bj))) {
            bt = tb[e];
            bj = j;
          }
        }
        const double tmin = row_min_d<W>(bj == 0x7fffffff ? kMax : bt);
        const int jmin = row_min_i<W>((bj != 0x7fffffff && bt == tmin) ? bj : 0x7fffffff);
        b_out = jmin;
        t_out = tmin;
      };
      int b = 0;
      double t = 0.0;
      int remaining;
      if (any_positive) {
        select_min(pending, b, t);
        remaining = static_cast<int>(seg_sum<W>(static_cast<double>(npos)));
      } else {
        // all t <= 0: the reference lands on the LAST sorted entry (:370-375): max (t, index)
        double bt = -kMax;
        int bj = -1;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = sl * E + e;
          if (j < n && (tb[e] > bt || (tb[e] == bt && j > bj))) {
            bt = tb[e];
            bj = j;
          }
        }
        const double tmax = seg_max<W>(bj < 0 ? -kMax : bt);
        const int jmax = -row_min_i<W>((bj >= 0 && bt == tmax) ? -bj : 0x7fffffff);
        b = jmax;
        t = tmax;
        remaining = 1;
#pragma unroll
        for (int e = 0; e < E; ++e) pending[e] = (sl * E + e == b);
      }
      double dt = t;
      MI355_PHASE(3);  // Cauchy point: breakpoint loop
      while ((dt_min >= dt) && (remaining > 0)) {                     // :382-412
        const int owner = b / E, be = b % E;
        double gsel = 0.0, dsel = 0.0, xsel = 0.0, losel = 0.0, hisel = 0.0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (e == be) {
            gsel = g[e];
            dsel = d[e];
            xsel = x[e];
            losel = lo_at(e);
            hisel = hi_at(e);
          }
        }
        // the bound the coordinate lands on and its displacement are computed where the coordinate lives (every lane
        // does it for its own be-th coordinate; the owner's is the one that counts): two broadcasts instead of five
        const double xcsel = (dsel > 0) ? hisel : ((dsel < 0) ? losel : xsel);
        const double gb = row_bcast_dyn<W>(gsel, owner);
        const double zb = row_bcast_dyn<W>(xcsel - xsel, owner);
        Mc = __builtin_fma(dt, Mp, Mc);                               // M^-1 (c + dt p)
        const double wbt = row_lane ? ws * mycol[b] : 0.0;            // W.row(b): lane a holds W(b, a)
        const double Mw = fast_solve_mm<K2, M, W>(mm_lz, mm_uz, mm_dinv, wbt);
        const double s1 = seg_sum<W>((gb * wbt) * Mc);
        const double s2 = seg_sum<W>(wbt * Mp);
        const double s3 = seg_sum<W>(((gb * gb) * wbt) * Mw);
        f_prime += ((dt * f_doubleprime + gb * gb) + (theta * gb) * zb) - s1;        // :396-397
        f_doubleprime += ((((-1.0) * theta) * gb) * gb - 2.0 * (gb * s2)) - s3;      // :398-400
        f_doubleprime = dmax(1e-12 * f_dp_orig, f_doubleprime);
        p_vec = __builtin_fma(gb, wbt, p_vec);
        Mp = __builtin_fma(gb, Mw, Mp);
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (sl * E + e == b) {
            xc[e] = xcsel;
            d[e] = 0.0;
            pending[e] = false;
          }
        }
        dt_min = -f_prime / f_doubleprime;
        t_old = t;
        remaining--;
        if (remaining > 0) {
          select_min(pending, b, t);
          dt = t - t_old;
        }
      }
      dt_min = dmax(dt_min, 0.0);
      t_old += dt_min;
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (pending[e]) xc[e] = __builtin_fma(t_old, d[e], x[e]);      // :424-427
      Mc = __builtin_fma(dt_min, Mp, Mc);                             // :429, through M^-1
    }

    // ---- subspace minimisation (:459-515) -----------------------------------------------
    MI355_PHASE(4);  // subspace: r, WZ r
    double smin[E];
    bool do_line_search;
    {
      bool is_free[E];
      int nfree = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        is_free[e] = (sl * E + e < n) && (xc[e] != hi_at(e)) && (xc[e] != lo_at(e));
        nfree += is_free[e] ? 1 : 0;
        smin[e] = xc[e];
      }
      do_line_search = seg_max<W>(static_cast<double>(nfree)) > 0.0;
      if (do_line_search) {
        double rr[E];
        ubuf[sl] = ws * Mc;
        segment_lds_fence();
        {
          double wmc[E];
          w_times_ubuf(wmc);
#pragma unroll
          for (int e = 0; e < E; ++e) {
            rr[e] = __builtin_fma(theta, xc[e] - x[e], g[e]) - wmc[e];                // :480
            vbuf[sl * E + e] = is_free[e] ? rr[e] : 0.0;
          }
        }
        segment_lds_fence();
        const double wzr = row_lane ? ws * chain4<P>(mycol, vbuf) : 0.0;              // WZ r (:485)
        MI355_PHASE(5);  // subspace: K = K0 + theta^-1 W_A^T W_A
        double krow[K2];
#pragma unroll
        for (int bq = 0; bq < K2; bq += 2) {
          const d2 v = ld2(K0m + ra * K2 + bq);
          krow[bq] = v.x;
          krow[bq + 1] = v.y;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          unsigned act = seg_bits((sl * E + e < n) && !is_free[e]);
          while (act != 0) {
            const int l = __builtin_ctz(act);
            act &= act - 1;
            const int j = l * E + e;
            double wrow[K2];  // W.row(j), raw: ten broadcast reads in flight before the first use
#pragma unroll
            for (int bb = 0; bb < K2; ++bb) wrow[bb] = Wc[bb * PITCH + j];
            const double ta = theta_inverse * (ws * mycol[j]);
            const double tbv = ta * theta;
#pragma unroll
            for (int bb = 0; bb < K2; ++bb) krow[bb] = __builtin_fma(bb < M ? ta : tbv, wrow[bb], krow[bb]);
          }
        }
        MI355_PHASE(6);  // subspace: v = K^-1 WZ r
        const double v = fast_solve_k<K2, W>(krow, wzr, sl);
        MI355_PHASE(7);  // subspace: du, alpha*
        const double ti2 = theta_inverse * theta_inverse;
        segment_lds_fence();
        ubuf[sl] = ti2 * (ws * v);
        segment_lds_fence();
        double du[E];
        {
          double wv[E];
          w_times_ubuf(wv);
#pragma unroll
          for (int e = 0; e < E; ++e) du[e] = __builtin_fma(-theta_inverse, rr[e], -wv[e]);   // :503-504
        }
        segment_lds_fence();
        double amin = 1.0;                                            // FindAlpha (:435-457)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          // the ratio matters only where the step overshoots its bound (ratio < 1): elsewhere min(1, ratio) = 1 whatever
          // the quotient rounds to, so the division runs only on wavefronts that hold such a coordinate
          const double room = ((du[e] > 0) ? hi_at(e) : lo_at(e)) - xc[e];
          const bool overshoot = (du[e] > 0) ? (room < du[e]) : (room > du[e]);
          if (is_free[e] && !(__builtin_fabs(du[e]) < 1e-7) && overshoot) {
            const double cand = room / du[e];   // one quotient: the bound is selected first
            amin = dmin(amin, cand);
          }
        }
        const double alphastar = row_min_d<W>(amin);
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (is_free[e]) smin[e] = __builtin_fma(alphastar, du[e], xc[e]);           // :508-514
      }
    }

    // ---- line search / evaluation (:181-203) ------------------------------------------
    MI355_PHASE(8);  // line search
    double xcur[kStaged ? 1 : E], gcur[kStaged ? 1 : E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if constexpr (kStaged) {
        stage[P + sl * E + e] = x[e];
        stage[2 * P + sl * E + e] = g[e];
      } else {
        xcur[e] = x[e];
        gcur[e] = g[e];
      }
    }
    if (do_line_search) {
      double dneg[E];
#pragma unroll
      for (int e = 0; e < E; ++e) dneg[e] = -(smin[e] - x[e]);
      const double dginit = -seg_dot<W, E, AR>(g, dneg);
      nfev += mt_cvsrch<W, E, Obj, AR>(obj, x, f, g, 1.0, dneg, dginit, n, sl);
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = smin[e];
      f = obj.template eval_fma<W, E>(x, g, n, sl);
      nfev++;
    }
    {
      double xcl[E];
      clip(x, xcl);                                                   // :199-203
      if (differs(xcl, x)) {
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = xcl[e];
        f = obj.template eval_fma<W, E>(x, g, n, sl);
        nfev++;
      }
    }

    // ---- history / compact representation update (:206-235) ---------------------------
    MI355_PHASE(9);  // history: ring slot, S^T Y / S^T S / Y^T Y entries
    {
      double ny[E], ns[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        if constexpr (kStaged) {
          ny[e] = g[e] - stage[2 * P + sl * E + e];
          ns[e] = x[e] - stage[P + sl * E + e];
        } else {
          ny[e] = g[e] - gcur[e];
          ns[e] = x[e] - xcur[e];
        }
      }
      const double sTy = seg_dot<W, E, AR>(ns, ny);
      const double yTy = seg_dot<W, E, AR>(ny, ny);
      if (sTy > 1e-7 * yTy) {                                         // :211
        int slot;
        if (k < mcap) {
          slot = k++;
        } else {
          slot = head;
          head = (head + 1 == mcap) ? 0 : head + 1;
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          Wc[slot * PITCH + sl * E + e] = ny[e];
          Wc[(M + slot) * PITCH + sl * E + e] = ns[e];
        }
        segment_lds_fence();
        theta = yTy / sTy;                                            // :222-223
        theta_inverse = 1.0 / theta;
        ws = (sl < M) ? 1.0 : theta;
        {
          double r1, r2;  // column . s_new, column . y_new
          chain4x2<P>(mycol, Wc + (M + slot) * PITCH, Wc + slot * PITCH, r1, r2);
          if (sl < M) {
            Amat[slot * M + sl] = r1;                                 // s_new . Y_a
            if (sl == slot) ypinv[slot] = -(1.0 / r1);                // 1 / MM(slot, slot), see fast_factor_mm
            YYmat[sl * M + slot] = r2;
            YYmat[slot * M + sl] = r2;
          } else if (row_lane) {
            const int i = sl - M;
            SSmat[i * M + slot] = r1;
            SSmat[slot * M + i] = r1;
            Amat[i * M + slot] = r2;                                  // S_a . y_new
          }
        }
        segment_lds_fence();
        MI355_PHASE(10);  // MM, K0 assembly + LU
        {
          // rank of a slot in the ring (0 = oldest); valid slots are 0..k-1
          auto rank = [&](int s) { return s - head + ((s < head) ? mcap : 0); };
          double mm_row[K2];
          if (sl < M) {  // row of a Y slot
            const int as = sl;
            const bool va = as < k;
            const int rka = rank(as);
            const double aaa = Amat[as * M + as], yaa = YYmat[as * M + as];
#pragma unroll
            for (int bq = 0; bq < M; ++bq) {
              mm_row[bq] = (bq == as) ? (va ? -aaa : 1.0) : 0.0;
              const double k0 = (-theta_inverse) * YYmat[as * M + bq];
              K0m[as * K2 + bq] = (bq == as) ? (va ? __builtin_fma(-theta_inverse, yaa, -aaa) : 1.0) : k0;
            }
#pragma unroll
            for (int i = 0; i < M; ++i) {
              const bool both = (i < k) && va;
              const double aia = Amat[i * M + as];
              const int rki = rank(i);
              mm_row[M + i] = (both && rki > rka) ? aia : 0.0;
              K0m[as * K2 + M + i] = (both && rki <= rka) ? -aia : 0.0;
            }
          } else {  // row of an S slot (idle lanes shadow the last one)
            const int i = step_lane(ra) - M;   // (opaque: keeps the (i == j) unit entries out of the loop invariants)
            const bool vi = i < k;
            const int rki = rank(i);
#pragma unroll
            for (int bq = 0; bq < M; ++bq) {
              const bool both = vi && (bq < k);
              const double aib = Amat[i * M + bq];
              const int rkb = rank(bq);
              const double mmv = (both && rki > rkb) ? aib : 0.0;
              mm_row[bq] = row_lane ? mmv : 0.0;
              K0m[ra * K2 + bq] = (both && rki <= rkb) ? -aib : 0.0;
            }
#pragma unroll
            for (int j = 0; j < M; ++j) {
              const double mmv = (vi && (j < k)) ? SSmat[i * M + j] * theta : ((i == j) ? 1.0 : 0.0);
              mm_row[M + j] = row_lane ? mmv : 0.0;
              K0m[ra * K2 + M + j] = (i == j && !vi) ? 1.0 : 0.0;
            }
          }
          fast_factor_mm<K2, M, W>(mm_row, mm_lz, mm_uz, mm_dinv, sl, ypinv);
        }
        segment_lds_fence();
      }
    }

    // ================== Progress::Update (progress.h:153-327), gradient test off ==========
    MI355_PHASE(11);  // Progress::Update + results
    num_iterations++;
    f_delta = __builtin_fabs(f - f_state);
    {
      double dx[E];
#pragma unroll
      for (int e = 0; e < E; ++e) dx[e] = x[e] - (kStaged ? stage[sl * E + e] : xs[kStaged ? 0 : e]);
      x_delta = seg_amax<W, E>(dx);
    }
    gradient_norm = seg_amax<W, E>(g);
    const mi355_lbfgs_stop& st = a.stop;
    status = MI355_STATUS_CONTINUE;
    bool decided = false;
    if ((stop_num_iterations > 0) && (num_iterations > stop_num_iterations)) {
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
          st.f_delta_relative ? dmax(dmax(__builtin_fabs(f), __builtin_fabs(f_state)), 1.0) : 1.0;
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
    // projected-gradient stop (:280-283): overrides whatever Update decided (quirk Q10)
    if ((stop_gradient_norm > 0) && (last_pg < stop_gradient_norm)) status = MI355_STATUS_GRADIENT_NORM_VIOLATION;

    trace_iteration<E>(a, prob, n, sl, num_iterations, status, f, x_delta, f_delta, gradient_norm, x, g);
    if (status != MI355_STATUS_CONTINUE) {
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
#ifdef MI355_LBFGSB_PHASE_TIMING
  MI355_PHASE(0);
  if (lane == 0 && a.profile != nullptr) {
#pragma unroll
    for (int i = 0; i < 16; ++i) atomicAdd(a.profile + i, phase_cycles[i]);
  }
#endif
}

}  // namespace mi355
