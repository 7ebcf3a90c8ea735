// lbfgsb_kernel.hpp — box-constrained L-BFGS-B, one problem per 16-lane wavefront segment.
//
// Device counterpart of the reference's include/cppoptlib/solver/lbfgsb.h
//   Lbfgsb::Minimize                   :247-292   (driver loop, projected-gradient stop)
//   Lbfgsb::OptimizationStep           :141-238   (clip, Cauchy point, subspace step, line search,
//                                                  history + W / MM / LU rebuild)
//   GetGeneralizedCauchyPoint          :318-430
//   SubspaceMinimization / FindAlpha   :459-515 / :435-457
//   SolveM                             :311-316
// with MoreThuente from more_thuente_device.hpp and Progress::Update (solver/progress.h:153-327).
//
// Mapping.  A segment is one DPP row (W = 16 lanes); lane `sl` owns coordinates sl*E..sl*E+E-1.
// Besides the coordinate vectors there is the small dense algebra of the compact representation
// (2k x 2k, k <= M pairs).  It is distributed over the same lanes: lane a owns ROW a of the LU
// factors of MM (and of N = I - M^-1 WZ WZ^T/theta) in registers and element a of every
// 2k-vector (p, c, W.row(b), M^-1 c, ...).  Row pivoting, elimination and the column-oriented
// substitutions broadcast one lane's value to the segment with `v_mov_b32_dpp row_newbcast:j`
// (static j, the loops over the 2M <= 16 rows are fully unrolled), so a solve is ~2k dependent
// steps of a few instructions instead of a scalar O(k^2) loop.  The history (chronological,
// shifted like the reference's leftCols/rightCols) and the k x k matrices S^T Y, S^T S live in
// LDS; S^T Y / S^T S are updated incrementally (their old entries would be recomputed to the
// same bits).  Breakpoints are visited in (t, index) order by repeated segment arg-min instead of
// a sort (the reference's std::sort leaves exact ties unspecified).
#pragma once
#include <stdint.h>

#include <type_traits>

#include "../../include/mi355_lbfgs.h"
#include "lbfgs_kernel.hpp"
#include "more_thuente_device.hpp"
#include "objectives.hpp"
#include "wave_primitives.hpp"

namespace mi355 {

// Profiling build (-DMI355_LBFGSB_PHASE_TIMING): every wavefront accumulates s_memtime deltas per
// phase of the iteration and adds them to args.s.profile[0..15] when it exits;
// scripts/lbfgsb_phases.py prints the breakdown.  Compiled out otherwise.
#ifdef MI355_LBFGSB_PHASE_TIMING
#define MI355_PHASE(i)                                        \
  do {                                                        \
    const unsigned long long now_ = __builtin_readcyclecounter(); \
    phase_cycles[phase_cur] += now_ - phase_t0;               \
    phase_t0 = now_;                                          \
    phase_cur = (i);                                          \
  } while (0)
#else
#define MI355_PHASE(i) do { } while (0)
#endif

struct LbfgsbArgs {
  SolveArgs s;            // shared fields (x0, outputs, objective, stop, queue, B, n; s.m = history size)
  const double* lower;    // device, n doubles (shared by the batch)
  const double* upper;
  int relaxed;            // host side only: the relaxed-algebra kernel (lbfgsb_fast_kernel.hpp) was selected
};

constexpr int kRowNewBcast = 0x150;  // DPP: lane N of each 16-lane row to the whole row
constexpr int kRowRor8 = 0x128;      // DPP row_ror:8: lane l <- lane l^8 of its 16-lane row

// lane l <- lane l^4 (ds_swizzle bit mode: and 0x1f, or 0, xor 4; no LDS memory is touched)
__device__ __forceinline__ double swizzle_xor4(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x101F);
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x101F);
  return __hiloint2double(hi, lo);
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0..N-1 (DPP controls are immediates)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// value of v in lane J of the caller's W-lane segment.  W = 16: one DPP row broadcast.  W = 32: lane J & 15 of the
// caller's own 16-lane row, or the same thing fetched from the segment's other row (v_permlane16_swap).
template <int W, int J>
__device__ __forceinline__ double row_bcast(double v) {
  static_assert((W == 16 || W == 32) && J < W, "segment width");
  const double own = dpp_mov<kRowNewBcast + (J & 15)>(v);
  if constexpr (W == 16) {
    return own;
  } else {
    const double other = xchg16(own);
    return (((__lane_id() >> 4) & 1) == (J >> 4)) ? own : other;
  }
}
// value of v in lane `src` (segment-uniform, runtime) of the caller's W-lane segment
template <int W>
__device__ __forceinline__ double row_bcast_dyn(double v, int src) {
  const int lane = __lane_id();
  const int addr = ((lane & ~(W - 1)) | src) << 2;
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_ds_bpermute(addr, lo);
  hi = __builtin_amdgcn_ds_bpermute(addr, hi);
  return __hiloint2double(hi, lo);
}
template <int W>
__device__ __forceinline__ int row_bcast_dyn_i(int v, int src) {
  const int addr = ((__lane_id() & ~(W - 1)) | src) << 2;
  return __builtin_amdgcn_ds_bpermute(addr, v);
}
template <int W>
__device__ __forceinline__ int row_min_i(int v) {
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, kQuadXor1, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, kQuadXor2, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, kRowHalfMirror, 0xF, 0xF, false));
  v = min(v, __builtin_amdgcn_update_dpp(0x7fffffff, v, kRowMirror, 0xF, 0xF, false));
  if constexpr (W == 32) {
    auto r = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(v), static_cast<unsigned>(v), false, false);
    v = min(static_cast<int>(r[0]), static_cast<int>(r[1]));
  }
  return v;
}
// (inputs: breakpoints t > 0, step ratios > 0, or the +max sentinel — ordered and non-zero, so one v_min_f64 per level
// returns the bits of the compare-and-select form at a third of the instructions)
template <int W>
__device__ __forceinline__ double row_min_d(double v) {
  v = vmin(v, dpp_mov<kQuadXor1>(v));
  v = vmin(v, dpp_mov<kQuadXor2>(v));
  v = vmin(v, dpp_mov<kRowHalfMirror>(v));
  v = vmin(v, dpp_mov<kRowMirror>(v));
  if constexpr (W == 32) v = vmin(v, xchg16(v));
  return v;
}

// The same minimum in compare-and-select form (std::min at every level): what the reference-order (bit-pinning) kernel
// below uses, so that its behaviour on unordered inputs — a NaN breakpoint or step ratio from a NaN gradient — is the
// oracle twin's by construction and not by the argument about input ranges the one-instruction form above rests on
// (round-4 advisor finding).  The relaxed kernel (lbfgsb_fast_kernel.hpp) keeps v_min_f64.
template <int W>
__device__ __forceinline__ double row_min_select(double v) {
  v = dmin(v, dpp_mov<kQuadXor1>(v));
  v = dmin(v, dpp_mov<kQuadXor2>(v));
  v = dmin(v, dpp_mov<kRowHalfMirror>(v));
  v = dmin(v, dpp_mov<kRowMirror>(v));
  if constexpr (W == 32) v = dmin(v, xchg16(v));
  return v;
}

// NV independent sums over the W lanes of a segment in one transposed butterfly: lane sl returns
// sum_lanes v[sl] (lanes >= NV return an unused value).  Every level halves the number of values a
// lane carries instead of reducing each value on all lanes: level 1 pairs lanes (l, l^1) and lane
// l keeps the values whose index has bit 0 equal to bit 0 of l, and so on for bits 1..3 (and bit 4
// of a 32-lane segment).  Each value is summed over exactly the pairwise tree seg_sum<W> uses
// ((l, l^1), then quads, ...), so the results are bit-identical to NV separate seg_sum calls at a
// fraction of the instructions (NV = 10, W = 16: 17 exchanges instead of 40).
template <int NV, int W = 16>
__device__ __forceinline__ double row_transpose_sum(const double (&v)[NV], int sl) {
  static_assert((W == 16 || W == 32) && NV >= 1 && NV <= W, "NV");
  constexpr int N1 = (NV + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2;
  const bool b0 = (sl & 1) != 0, b1 = (sl & 2) != 0, b2 = (sl & 4) != 0, b3 = (sl & 8) != 0;
  double w[N1], z[N2], y[N3], u[N4];
#pragma unroll
  for (int i = 0; i < N1; ++i) {
    const double lo = v[2 * i], hi = (2 * i + 1 < NV) ? v[(2 * i + 1 < NV) ? 2 * i + 1 : 0] : 0.0;
    w[i] = (b0 ? hi : lo) + dpp_mov<kQuadXor1>(b0 ? lo : hi);
  }
#pragma unroll
  for (int i = 0; i < N2; ++i) {
    const double lo = w[2 * i], hi = (2 * i + 1 < N1) ? w[(2 * i + 1 < N1) ? 2 * i + 1 : 0] : 0.0;
    z[i] = (b1 ? hi : lo) + dpp_mov<kQuadXor2>(b1 ? lo : hi);
  }
#pragma unroll
  for (int i = 0; i < N3; ++i) {
    const double lo = z[2 * i], hi = (2 * i + 1 < N2) ? z[(2 * i + 1 < N2) ? 2 * i + 1 : 0] : 0.0;
    y[i] = (b2 ? hi : lo) + swizzle_xor4(b2 ? lo : hi);
  }
#pragma unroll
  for (int i = 0; i < N4; ++i) {
    const double lo = y[2 * i], hi = (2 * i + 1 < N3) ? y[(2 * i + 1 < N3) ? 2 * i + 1 : 0] : 0.0;
    u[i] = (b3 ? hi : lo) + dpp_mov<kRowRor8>(b3 ? lo : hi);
  }
  if constexpr (W == 16) {
    return u[0];
  } else {
    const bool b4 = (sl & 16) != 0;
    const double lo = u[0], hi = (N4 > 1) ? u[N4 > 1 ? 1 : 0] : 0.0;
    return (b4 ? hi : lo) + xchg16(b4 ? lo : hi);
  }
}

// Distributed K2 x K2 LU (row `sl` per lane) with first-maximum row pivoting; mirrors
// oracle SmallLU::factor / the eigen_shim PartialPivLU.  `piv`: lane k holds the pivot row of step k.
// `perm`: the row interchanges composed into one gather — after the factorisation lane a of a
// permuted right-hand side takes element perm_a of the original (what applying the interchanges
// one after the other, as PartialPivLU::solve does, arrives at).
template <int K2, int W = 16>
__device__ __forceinline__ void lu_factor(double (&row)[K2], int& perm, int k2, int sl) {
  perm = sl;
  static_for<0, K2>([&](auto ic) {
    constexpr int kk = decltype(ic)::value;
    if (kk < k2) {
      const double cand = (sl >= kk && sl < k2) ? __builtin_fabs(row[kk]) : -1.0;
      const double best = seg_max<W>(cand);
      const int p = row_min_i<W>((cand == best) ? sl : 0x7fffffff);
      if (best != 0.0) {
        if (p != kk) {  // rows kk and p change places: one gather with a per-lane source
          const int src = (sl == kk) ? p : ((sl == p) ? kk : sl);
#pragma unroll
          for (int j = 0; j < K2; ++j) row[j] = row_bcast_dyn<W>(row[j], src);
          perm = row_bcast_dyn_i<W>(perm, src);
        }
        const double pivot = row_bcast<W, kk>(row[kk]);
        if (sl > kk && sl < k2) row[kk] = row[kk] / pivot;
      }
#pragma unroll
      for (int j = kk + 1; j < K2; ++j) {
        const double ukj = row_bcast<W, kk>(row[j]);
        if (j < k2 && sl > kk && sl < k2) row[j] = row[j] - row[kk] * ukj;
      }
    }
  });
}

// x := LU^-1 x for a distributed vector (lane a holds x_a); column-oriented substitutions.
template <int K2, int W = 16>
__device__ __forceinline__ double lu_solve(const double (&row)[K2], int perm, int k2, int sl, double x) {
  x = row_bcast_dyn<W>(x, perm);  // all row interchanges at once (perm_a = a outside the factored block)
  static_for<0, K2>([&](auto ic) {  // unit lower triangle, column oriented
    constexpr int j = decltype(ic)::value;
    if (j < k2) {
      const double xj = row_bcast<W, j>(x);
      if (sl > j && sl < k2) x = x - xj * row[j];
    }
  });
  static_for<0, K2>([&](auto ic) {  // upper triangle, column oriented, last column first
    constexpr int j = K2 - 1 - decltype(ic)::value;
    if (j < k2) {
      if (sl == j) x = x / row[j];
      const double xj = row_bcast<W, j>(x);
      if (sl < j) x = x - xj * row[j];
    }
  });
  return x;
}

// The same solve for several right-hand sides at once, one per lane.  `lu` is the factorisation
// copied to LDS (row major, pitch K2), `perm` the composed interchanges, `cols` holds column c at
// cols[c*K2 .. c*K2+k2).  Lane c < ncols solves column c in place with exactly the operation order
// of lu_solve (column-oriented substitutions); all lanes of a segment read the same factor entry
// (an LDS broadcast), so ncols solves cost about what one distributed solve costs.  The first
// `ncomplement` columns are replaced by e_c - x instead (the N = I - M^-1 N step, :489-495).
// The lane index made opaque to the optimiser: expressions of it (the unit entries (i == sl) of the complement below, the
// (row, column) of a lane's entry of N) are loop invariants of the whole kernel; left alone the compiler computes them all
// before the iteration loop and — the register file being full — SPILLS them (12 VGPRs of scratch in the configs[4] kernel,
// profiles/r4_kernel_resources.txt).  Recomputed where used they are a handful of integer instructions per iteration.
__device__ __forceinline__ int opaque_lane(int sl) {
  asm volatile("" : "+v"(sl));
  return sl;
}

template <int K2>
__device__ __forceinline__ void lu_solve_columns(const double* lu, const int* perm, double* cols, int k2,
                                                 int ncols, int sl_in, int ncomplement) {
  const int sl = opaque_lane(sl_in);
  const bool complement = sl < ncomplement;
  if (sl < ncols) {
    double* const col = cols + sl * K2;
    double x[K2];
#pragma unroll
    for (int i = 0; i < K2; ++i) x[i] = (i < k2) ? col[perm[i]] : 0.0;
#pragma unroll
    for (int j = 0; j < K2; ++j) {  // unit lower triangle
      if (j + 1 < k2) {
#pragma unroll
        for (int i = j + 1; i < K2; ++i)
          if (i < k2) x[i] = x[i] - x[j] * lu[i * K2 + j];
      }
    }
#pragma unroll
    for (int jj = 0; jj < K2; ++jj) {  // upper triangle, last column first
      const int j = K2 - 1 - jj;
      if (j < k2) {
        x[j] = x[j] / lu[j * K2 + j];
#pragma unroll
        for (int i = 0; i < j; ++i) x[i] = x[i] - x[j] * lu[i * K2 + j];
      }
    }
#pragma unroll
    for (int i = 0; i < K2; ++i)
      if (i < k2) col[i] = complement ? ((i == sl) ? 1.0 : 0.0) - x[i] : x[i];
  }
}

// ascending sum over lanes 0..k2-1 of a distributed vector:  ((t0 + t1) + t2) + ...
template <int K2, int W = 16>
__device__ __forceinline__ double row_seq_sum(double t, int k2) {
  double s = row_bcast<W, 0>(t);
  static_for<1, K2>([&](auto ic) {
    constexpr int a = decltype(ic)::value;
    const double ta = row_bcast<W, a>(t);
    if (a < k2) s = s + ta;
  });
  return (k2 > 0) ? s : 0.0;
}

template <int M>
__host__ __device__ inline int lbfgsb_lds_doubles_per_problem(int P, int objective_scratch) {
  // history (Y, S), S^T Y, S^T S, the K2 x K2 scratch N, the LU of MM + its permutation, plateau ring
  // (N has one column more than rows: the right-hand side WZ r rides along with the column solves)
  return 2 * M * P + 2 * M * M + (4 * M * M + 2 * M) + (4 * M * M + 2 * M) + MI355_LBFGS_MAX_PAST + objective_scratch;
}

// LS: the LineSearch template argument of the reference's Lbfgsb (lbfgsb.h:45)
// OUTER: as for lbfgs_solve_kernel (NoOuterLoop, or the augmented-Lagrangian loop around the solves of a problem)
// W: lanes per problem.  16 (one DPP row) serves 2M <= 16 rows of the compact representation; 32 lanes serve M = 10
// (the broadcasts and the last butterfly level cross the two rows of the segment with v_permlane16_swap).
template <int E, class Obj, int M, int LS = MI355_LS_MORE_THUENTE, class OUTER = NoOuterLoop, int W = 16>
// (two wavefronts per SIMD asked for where the kernel fits 256 registers — one or two coordinates per lane, m <= 5:
// without the hint the allocator hoists the
// per-column scale / index of W out of the iteration and ends at 272)
__global__ __launch_bounds__(64, (E >= 4 || M > 5 || W > 16) ? 1 : 2) void lbfgsb_solve_kernel(const LbfgsbArgs args,
                                                                                                  const typename OUTER::Args oa) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constexpr int P = W * E;
  constexpr int K2 = 2 * M;
  [[maybe_unused]] constexpr int kSegs = kWave / W;
  static_assert(K2 <= W, "the 2M rows of the compact representation must fit the lanes of one segment");
  constexpr double kMax = 1.7976931348623157e308;
  const SolveArgs& a = args.s;

  const int lane = threadIdx.x & (kWave - 1);
  const int seg = lane / W;
  const int sl = lane % W;
  const int n = a.n;
  const int mcap = a.m;  // history size of this solve, 1..M (the template argument m of the reference's Lbfgsb)

  // (the objective's read-only region, if any, comes first: one wavefront per workgroup, so it is per wavefront here)
  double* const base = lds + Obj::shared_lds_doubles() + seg * lbfgsb_lds_doubles_per_problem<M>(P, Obj::kLdsDoubles);
  double* const Yh = base;                  // [M][P] chronological (oldest first)
  double* const Sh = Yh + M * P;
  double* const Amat = Sh + M * P;          // S^T Y, column major, stride M
  double* const SSmat = Amat + M * M;       // S^T S
  double* const Nmat = SSmat + M * M;       // K2 x K2 scratch, column major, stride K2
  double* const LUm = Nmat + K2 * K2 + K2;  // LU of MM, row major, pitch K2 (copy of mm_row for the column solves)
  int* const permL = reinterpret_cast<int*>(LUm + K2 * K2);  // its composed interchanges (K2 ints in K2 doubles)
  double* const past_f = LUm + K2 * K2 + K2;

  Obj obj;
  obj.load(a.obj_params, n, sl, past_f + MI355_LBFGS_MAX_PAST, lds);
  if constexpr (Obj::shared_lds_doubles() > 0) {
    obj.fill_shared(lds, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
    __syncthreads();
  }
  const long long queue_length = a.count_dev ? static_cast<long long>(*a.count_dev) : a.B;
  // the two stopping fields an outer loop changes between the solves of one problem (uniform otherwise)
  [[maybe_unused]] unsigned long long stop_num_iterations = a.stop.num_iterations;
  [[maybe_unused]] double stop_gradient_norm = a.stop.gradient_norm;

  double lo[E], hi[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    lo[e] = (j < n) ? args.lower[j] : 0.0;
    hi[e] = (j < n) ? args.upper[j] : 0.0;
  }
  auto clip = [&](const double (&v)[E], double (&out)[E]) {  // cwiseMin(upper).cwiseMax(lower)
#pragma unroll
    for (int e = 0; e < E; ++e) out[e] = dmax(dmin(v[e], hi[e]), lo[e]);
  };
  auto differs = [&](const double (&u)[E], const double (&v)[E]) {
    int dflag = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) dflag |= (sl * E + e < n && u[e] != v[e]) ? 1 : 0;
    return seg_max<W>(static_cast<double>(dflag)) != 0.0;
  };

  long long prob = 0;
  bool need_fetch = true;
  double x[E], g[E];
  double f = 0.0;
  unsigned nfev = 0, sum_k = 0;
  int k = 0;
  double theta = 1.0;
  double mm_row[K2];
  int mm_perm = 0;  // composed row interchanges of that factorisation (lane a: source row of row a)
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
  for (int j = 0; j < K2; ++j) mm_row[j] = 0.0;

  auto Wval = [&](int col, int coord) {  // W = [Y, theta*S]  (:224-226)
#ifdef MI355_LBFGSB_WVAL_BRANCHY
    return (col < k) ? Yh[col * P + coord] : theta * Sh[(col - k) * P + coord];
#else
    // one load from a selected column and one multiply (by 1 for a Y column: exact) instead of both loads and a
    // select of the results; Sh = Yh + M * P, so the column is a single index
    const bool is_y = col < k;
    const int column = is_y ? col : (M + col - k);
    const double scale = is_y ? 1.0 : theta;
    return scale * Yh[column * P + coord];
#endif
  };
  auto solveM = [&](double v, int k2) {  // :311-316
    return (k2 == 0) ? v : lu_solve<K2, W>(mm_row, mm_perm, k2, sl, v);
  };

#ifdef MI355_LBFGSB_PHASE_TIMING
  unsigned long long phase_cycles[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) phase_cycles[i] = 0;
  unsigned long long phase_t0 = __builtin_readcyclecounter();
  int phase_cur = 0;
#endif
  // Minimize prologue (:253) from the point in x + InitializeSolver (:120-139)
  [[maybe_unused]] double f_start = 0.0;  // the first evaluation of the current solve (an outer loop reports against it)
  auto reset_solver = [&]() {
    f_start = f;
    nfev = 1;
    sum_k = 0;
    k = 0;
    theta = 1.0;
    last_pg = 0.0;
    num_iterations = 0;
    x_delta_violations = 0;
    f_delta_violations = 0;
    x_delta = f_delta = gradient_norm = 0.0;
    status = MI355_STATUS_NOT_STARTED;
    past_init = false;
    past_pos = 0;
  };
  auto start_solve = [&]() {
    f = obj.template eval<W, E>(x, g, n, sl);
    reset_solver();
  };
  while (true) {
    MI355_PHASE(0);  // fetch / prologue
    if (need_fetch) {
      unsigned long long nxt = 0;
      if (sl == 0) nxt = atomicAdd(a.next_problem, 1ULL);
      const unsigned lo32 = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt & 0xffffffffULL)));
      const unsigned hi32 = static_cast<unsigned>(seg_bcast_first<W>(static_cast<int>(nxt >> 32)));
      prob = static_cast<long long>((static_cast<unsigned long long>(hi32) << 32) | lo32);
      if (prob >= queue_length) break;
      const KernargSolveArgs ca = cold_args();  // (LbfgsbArgs starts with its SolveArgs; see lbfgs_kernel.hpp)
      {
        const int* const map = ca->problem_map;
        if (map != nullptr) prob = map[prob];
      }
      need_fetch = false;
      // ---- the start point ------------------------------------------------------------
      {
        const double* const x0p = ca->x0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const int j = sl * E + e;
          x[e] = (j < n) ? x0p[prob * n + j] : 0.0;
        }
      }
      obj.begin_problem(ca->per_problem, prob, ca->per_problem_stride, sl);
      if constexpr (OUTER::kEnabled) OUTER::begin(obj, oa, a, prob, x, sl, stop_num_iterations, stop_gradient_norm);
      start_solve();
    }

    // ============================ OptimizationStep (:141-238) ===========================
    MI355_PHASE(1);  // clip + projected gradient
    double xs[E];  // state x at entry (Progress::Update compares against it)
    const double f_state = f;
#pragma unroll
    for (int e = 0; e < E; ++e) xs[e] = x[e];
    {
      double xc0[E];
      clip(x, xc0);                                                   // :148
      if (differs(xc0, x)) {                                          // :151-153
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = xc0[e];
        f = obj.template eval<W, E>(x, g, n, sl);
        nfev++;
      }
    }
    const int k2 = 2 * k;
    sum_k += k;
    {  // projected gradient sup-norm (:105-118, :165-166)
      double t[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        double gj = g[e];
        if (x[e] <= lo[e] && gj > 0) gj = 0.0;
        if (x[e] >= hi[e] && gj < 0) gj = 0.0;
        t[e] = (sl * E + e < n) ? __builtin_fabs(gj) : 0.0;
      }
      last_pg = seg_max<W>(lane_max<E>(t));
    }

    // ---- generalized Cauchy point (:318-430) -------------------------------------------
    MI355_PHASE(2);  // Cauchy point: breakpoints, p = W^T d, first solve
    double xc[E], d[E], tb[E];
    bool pending[E];
    double c_vec = 0.0, p_vec = 0.0;  // distributed 2k-vectors
    {
      int npos = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        d[e] = -g[e];
        double tmp = kMax;
        if (g[e] != 0) {
          tmp = (g[e] < 0) ? (x[e] - hi[e]) / g[e] : (x[e] - lo[e]) / g[e];
          if (tmp == 0) d[e] = 0;
        }
        tb[e] = tmp;
        xc[e] = x[e];
        pending[e] = (j < n) && (tmp > 0);
        npos += pending[e] ? 1 : 0;
        if (j >= n) d[e] = 0.0;
      }
      const bool any_positive = seg_max<W>(static_cast<double>(npos)) > 0.0;
      {                                                               // p = W^T d (:353)
        double part[K2];  // all 2k dot products share one transposed butterfly
#pragma unroll
        for (int col = 0; col < K2; ++col) {
          double t[E];
#pragma unroll
          for (int e = 0; e < E; ++e) t[e] = (col < k2) ? Wval(col, sl * E + e) * d[e] : 0.0;
          part[col] = lane_tree_sum<E>(t);
        }
        const double val = row_transpose_sum<K2, W>(part, sl);
        p_vec = (sl < k2) ? val : 0.0;
      }
      double f_prime = -seg_dot<W, E>(d, d);                           // :357
      const double Mp0 = solveM(p_vec, k2);
      const double pMp = row_seq_sum<K2, W>(p_vec * Mp0, k2);
      double f_doubleprime = (-theta) * f_prime - pMp;                // :361-362
      f_doubleprime = dmax(1e-12, f_doubleprime);
      const double f_dp_orig = f_doubleprime;
      double dt_min = -f_prime / f_doubleprime;
      double t_old = 0.0;

      // breakpoint selection by (t, index) key among a candidate set
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
        // lanes without a candidate must not win: key (kMax, INT_MAX)
        const double tmin = row_min_select<W>(bj == 0x7fffffff ? kMax : bt);
        const int jmin = row_min_i<W>((bj != 0x7fffffff && bt == tmin) ? bj : 0x7fffffff);
        b_out = jmin;
        t_out = tmin;
      };
      int b = 0;
      double t = 0.0;
      int remaining;  // entries at sorted positions >= i
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
            losel = lo[e];
            hisel = hi[e];
          }
        }
        const double gb = row_bcast_dyn<W>(gsel, owner);
        const double db = row_bcast_dyn<W>(dsel, owner);
        const double xb = row_bcast_dyn<W>(xsel, owner);
        const double lob = row_bcast_dyn<W>(losel, owner);
        const double hib = row_bcast_dyn<W>(hisel, owner);
        double xcb = xb;
        if (db > 0)
          xcb = hib;
        else if (db < 0)
          xcb = lob;
        const double zb = xcb - xb;
        c_vec = c_vec + dt * p_vec;
        const double wbt = (sl < k2) ? Wval(sl, b) : 0.0;            // W.row(b): lane a reads W(b, a)
        // M^-1 c, M^-1 p, M^-1 W.row(b): three right-hand sides, solved side by side (lanes 0..2)
        double Mc = c_vec, Mp = p_vec, Mwbt = wbt;
        if (k2 > 0) {
          if (sl < k2) {
            Nmat[0 * K2 + sl] = c_vec;
            Nmat[1 * K2 + sl] = p_vec;
            Nmat[2 * K2 + sl] = wbt;
          }
          segment_lds_fence();
          lu_solve_columns<K2>(LUm, permL, Nmat, k2, 3, sl, 0);
          segment_lds_fence();
          Mc = (sl < k2) ? Nmat[0 * K2 + sl] : 0.0;
          Mp = (sl < k2) ? Nmat[1 * K2 + sl] : 0.0;
          Mwbt = (sl < k2) ? Nmat[2 * K2 + sl] : 0.0;
          segment_lds_fence();
        }
        const double s1 = row_seq_sum<K2, W>((gb * wbt) * Mc, k2);
        const double s2 = row_seq_sum<K2, W>(wbt * Mp, k2);
        const double s3 = row_seq_sum<K2, W>(((gb * gb) * wbt) * Mwbt, k2);
        f_prime += ((dt * f_doubleprime + gb * gb) + (theta * gb) * zb) - s1;        // :396-397
        f_doubleprime += ((((-1.0) * theta) * gb) * gb - 2.0 * (gb * s2)) - s3;      // :398-400
        f_doubleprime = dmax(1e-12 * f_dp_orig, f_doubleprime);
        p_vec = p_vec + gb * wbt;
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (sl * E + e == b) {
            xc[e] = xcb;
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
        if (pending[e]) xc[e] = x[e] + t_old * d[e];                  // :424-427
      c_vec = c_vec + dt_min * p_vec;                                 // :429
    }

    // ---- subspace minimisation (:459-515) -----------------------------------------------
    MI355_PHASE(4);  // subspace: M^-1 c, r, WZ r, M^-1 (WZ r)
    double smin[E];
    bool do_line_search;
    {
      bool is_free[E];
      int nfree = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        is_free[e] = (sl * E + e < n) && (xc[e] != hi[e]) && (xc[e] != lo[e]);
        nfree += is_free[e] ? 1 : 0;
        smin[e] = xc[e];
      }
      do_line_search = seg_max<W>(static_cast<double>(nfree)) > 0.0;
      if (do_line_search) {
        const double theta_inverse = 1.0 / theta;
        const double Mc = solveM(c_vec, k2);
        double rr[E];
        {
          double wmc[E];
#pragma unroll
          for (int e = 0; e < E; ++e) wmc[e] = 0.0;
          static_for<0, K2>([&](auto ic) {
            constexpr int col = decltype(ic)::value;
            const double mca = row_bcast<W, col>(Mc);
            if (col < k2) {
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const double term = Wval(col, sl * E + e) * mca;
                wmc[e] = (col == 0) ? term : wmc[e] + term;
              }
            }
          });
#pragma unroll
          for (int e = 0; e < E; ++e) rr[e] = (g[e] + theta * (xc[e] - x[e])) - wmc[e];   // :480
        }
        double wzr;
        {                                                             // WZ * r (:485)
          double part[K2];
#pragma unroll
          for (int col = 0; col < K2; ++col) {
            double t[E];
#pragma unroll
            for (int e = 0; e < E; ++e) t[e] = (col < k2 && is_free[e]) ? Wval(col, sl * E + e) * rr[e] : 0.0;
            part[col] = lane_tree_sum<E>(t);
          }
          const double val = row_transpose_sum<K2, W>(part, sl);
          wzr = (sl < k2) ? val : 0.0;
        }
        // v = M^-1 (WZ r) (:486).  Where a lane of the segment is left over (2k < 16) the solve rides along with the
        // column solves of N = I - M^-1 N below as one more column — same factors, same operation order as lu_solve.
        constexpr bool kRideAlong = (K2 < W);
        double v = (kRideAlong && k2 > 0) ? 0.0 : solveM(wzr, k2);
        // N = theta^-1 WZ WZ^T (:487), then N = I - M^-1 N (:489-495), built in LDS
        MI355_PHASE(5);  // subspace: WZ WZ^T
        {
          // Entry (ar, bc) = sum over the free coordinates of (theta^-1 W(:, ar)) * W(:, bc).  Both factors are
          // zeroed (+0) outside the free set once, instead of selecting every product: +0 * +0 is the +0 the
          // reference's zero rows of Z contribute.  The (2k)^2 sums are taken sixteen at a time (entry idx =
          // bc * K2 + ar of the column-major N lands in lane idx mod 16), so a full transposed butterfly serves
          // sixteen entries — 7 butterflies for the 100 entries of m = 5 instead of 10 partly filled ones.
          double Um[K2][E];
#pragma unroll
          for (int ar = 0; ar < K2; ++ar) {
#pragma unroll
            for (int e = 0; e < E; ++e)
              Um[ar][e] = (ar < k2 && is_free[e]) ? theta_inverse * Wval(ar, sl * E + e) : 0.0;
          }
          // (eight coordinates per lane: one column of N per butterfly, to stay inside the register file)
          constexpr int kTotal = K2 * K2;
          constexpr int kBatch = (E >= 8) ? K2 : W;
          constexpr int kBatches = (kTotal + kBatch - 1) / kBatch;
          static_for<0, kBatches>([&](auto ib) {
            constexpr int first = decltype(ib)::value * kBatch;
            constexpr int cnt = (kTotal - first < kBatch) ? (kTotal - first) : kBatch;
            constexpr int bc_lo = first / K2, bc_hi = (first + cnt - 1) / K2;
            if (bc_lo < k2) {
              double wbm[bc_hi - bc_lo + 1][E];
#pragma unroll
              for (int c = 0; c <= bc_hi - bc_lo; ++c) {
                const int col = (bc_lo + c < k2) ? bc_lo + c : 0;
#pragma unroll
                for (int e = 0; e < E; ++e)
                  wbm[c][e] = (is_free[e] && bc_lo + c < k2) ? Wval(col, sl * E + e) : 0.0;
              }
              double part[cnt];
#pragma unroll
              for (int i = 0; i < cnt; ++i) {
                const int bc = (first + i) / K2, ar = (first + i) % K2;
                double t[E];
#pragma unroll
                for (int e = 0; e < E; ++e) t[e] = Um[ar][e] * wbm[bc - bc_lo][e];
                part[i] = lane_tree_sum<E>(t);
              }
              const double val = row_transpose_sum<cnt, W>(part, sl);
              const int slo = opaque_lane(sl);
              const int idx = first + slo;
              if (slo < cnt && (idx % K2) < k2 && (idx / K2) < k2) Nmat[idx] = val;
            }
          });
        }
        if (kRideAlong && k2 > 0 && sl < k2) Nmat[k2 * K2 + sl] = wzr;
        segment_lds_fence();
        MI355_PHASE(6);  // subspace: N = I - M^-1 N, LU(N), v
        if (k2 > 0)  // N = I - M^-1 N, column per lane (+ the ride-along right-hand side)
          lu_solve_columns<K2>(LUm, permL, Nmat, k2, kRideAlong ? k2 + 1 : k2, sl, k2);
        segment_lds_fence();
        if (k2 > 0) {                                                 // :498-500
          double nrow[K2];
          int nperm = 0;
#pragma unroll
          for (int j = 0; j < K2; ++j) nrow[j] = (sl < k2 && j < k2) ? Nmat[j * K2 + sl] : 0.0;
          if constexpr (kRideAlong) v = (sl < k2) ? Nmat[k2 * K2 + sl] : 0.0;
          lu_factor<K2, W>(nrow, nperm, k2, sl);
          v = lu_solve<K2, W>(nrow, nperm, k2, sl, v);
        }
        MI355_PHASE(7);  // subspace: du, alpha*
        const double ti2 = theta_inverse * theta_inverse;
        double du[E];
        {
          double wv[E];
#pragma unroll
          for (int e = 0; e < E; ++e) wv[e] = 0.0;
          static_for<0, K2>([&](auto ic) {
            constexpr int col = decltype(ic)::value;
            const double va = row_bcast<W, col>(v);
            if (col < k2) {
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const double term = (ti2 * Wval(col, sl * E + e)) * va;
                wv[e] = (col == 0) ? term : wv[e] + term;
              }
            }
          });
#pragma unroll
          for (int e = 0; e < E; ++e) du[e] = (-theta_inverse) * rr[e] - wv[e];   // :503-504
        }
        double amin = 1.0;                                            // FindAlpha (:435-457)
#pragma unroll
        for (int e = 0; e < E; ++e) {
          if (is_free[e] && !(__builtin_fabs(du[e]) < 1e-7)) {
            const double cand = (du[e] > 0) ? (hi[e] - xc[e]) / du[e] : (lo[e] - xc[e]) / du[e];
            amin = dmin(amin, cand);
          }
        }
        const double alphastar = row_min_select<W>(amin);
#pragma unroll
        for (int e = 0; e < E; ++e)
          if (is_free[e]) smin[e] = smin[e] + alphastar * du[e];      // :508-514
      }
    }

    // ---- line search / evaluation (:181-203) ------------------------------------------
    MI355_PHASE(8);  // line search
    double xcur[E], gcur[E];
    const double fcur = f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      xcur[e] = x[e];
      gcur[e] = g[e];
    }
    if (do_line_search) {
      double dneg[E];  // negated direction (mt_cvsrch runs along -dneg)
#pragma unroll
      for (int e = 0; e < E; ++e) dneg[e] = -(smin[e] - x[e]);
      const double dginit = -seg_dot<W, E>(g, dneg);
      if constexpr (LS == MI355_LS_HAGER_ZHANG) {
        double stp = 1.0;
        bool ls_failed = false;
        nfev += hz_search<W, E>(obj, x, f, g, stp, dneg, dginit, n, sl, ls_failed);
        if (ls_failed) {  // hzls returned -1: the State overload hands back the start state
          f = fcur;
#pragma unroll
          for (int e = 0; e < E; ++e) g[e] = gcur[e];
        }
      } else {
        nfev += mt_cvsrch<W, E>(obj, x, f, g, 1.0, dneg, dginit, n, sl);
      }
    } else {
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = smin[e];
      f = obj.template eval<W, E>(x, g, n, sl);
      nfev++;
    }
    {
      double xcl[E];
      clip(x, xcl);                                                   // :199-203
      if (differs(xcl, x)) {
#pragma unroll
        for (int e = 0; e < E; ++e) x[e] = xcl[e];
        f = obj.template eval<W, E>(x, g, n, sl);
        nfev++;
      }
    }

    // ---- history / compact representation update (:206-235) ---------------------------
    MI355_PHASE(9);  // history: shift, S^T Y / S^T S entries
    {
      double ny[E], ns[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        ny[e] = g[e] - gcur[e];
        ns[e] = x[e] - xcur[e];
      }
      const double sTy = seg_dot<W, E>(ns, ny);
      const double yTy = seg_dot<W, E>(ny, ny);
      if (sTy > 1e-7 * yTy) {                                         // :211
        if (k < mcap) {
          k++;
        } else {                                                      // shift left (:216-217)
          for (int col = 0; col + 1 < mcap; ++col) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
              Yh[col * P + sl * E + e] = Yh[(col + 1) * P + sl * E + e];
              Sh[col * P + sl * E + e] = Sh[(col + 1) * P + sl * E + e];
            }
          }
          // S^T Y and S^T S lose their first row and column
          double ta[(M * M + W - 1) / W], ts[(M * M + W - 1) / W];
#pragma unroll
          for (int r = 0; r < (M * M + W - 1) / W; ++r) {
            const int idx = sl + r * W;
            const int ia = idx % M, ib = idx / M;
            const bool ok = idx < M * M && ia + 1 < mcap && ib + 1 < mcap;
            ta[r] = ok ? Amat[(ib + 1) * M + ia + 1] : 0.0;
            ts[r] = ok ? SSmat[(ib + 1) * M + ia + 1] : 0.0;
          }
          segment_lds_fence();
#pragma unroll
          for (int r = 0; r < (M * M + W - 1) / W; ++r) {
            const int idx = sl + r * W;
            const int ia = idx % M, ib = idx / M;
            if (idx < M * M && ia + 1 < mcap && ib + 1 < mcap) {
              Amat[ib * M + ia] = ta[r];
              SSmat[ib * M + ia] = ts[r];
            }
          }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) {
          Yh[(k - 1) * P + sl * E + e] = ny[e];
          Sh[(k - 1) * P + sl * E + e] = ns[e];
        }
        segment_lds_fence();
        theta = yTy / sTy;                                            // :222-223 (y.s == s.y bit for bit)
        // new column / row of S^T Y, new row+column of S^T S (older entries are unchanged)
        {
          // value 3*col + {0,1,2}: S_col.y_new, s_new.Y_col, S_col.s_new — one transposed butterfly
          double part[3 * M];
#pragma unroll
          for (int col = 0; col < M; ++col) {
            double t0[E], t1[E], t2[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
              const double sc = (col < k) ? Sh[col * P + sl * E + e] : 0.0;
              const double yc = (col < k) ? Yh[col * P + sl * E + e] : 0.0;
              t0[e] = sc * ny[e];
              t1[e] = ns[e] * yc;
              t2[e] = sc * ns[e];
            }
            part[3 * col + 0] = lane_tree_sum<E>(t0);
            part[3 * col + 1] = lane_tree_sum<E>(t1);
            part[3 * col + 2] = lane_tree_sum<E>(t2);
          }
          auto store = [&](int idx, double val) {
            const int col = idx / 3, which = idx - 3 * col;
            if (idx < 3 * M && col < k) {
              if (which == 0) Amat[(k - 1) * M + col] = val;          // A(col, k-1) = S_col . y_new
              if (which == 1) Amat[col * M + (k - 1)] = val;          // A(k-1, col) = s_new . Y_col
              if (which == 2) {                                       // SS(col, k-1) = SS(k-1, col)
                SSmat[(k - 1) * M + col] = val;
                SSmat[col * M + (k - 1)] = val;
              }
            }
          };
          if constexpr (3 * M <= W) {
            store(sl, row_transpose_sum<3 * M, W>(part, sl));
          } else {  // more sums than lanes (M = 8: 24): two transposed butterflies, the same pairwise trees
            static_assert(3 * M <= 2 * W, "history size");
            double p0[W], p1[3 * M - W];
#pragma unroll
            for (int i = 0; i < W; ++i) p0[i] = part[i];
#pragma unroll
            for (int i = 0; i < 3 * M - W; ++i) p1[i] = part[W + i];
            const double v0 = row_transpose_sum<W, W>(p0, sl);
            const double v1 = row_transpose_sum<3 * M - W, W>(p1, sl);
            store(sl, v0);
            store(W + sl, v1);
          }
        }
        segment_lds_fence();
        // MM = [[-diag(A), L^T], [L, theta*S^T S]] (:227-232): lane i assembles row i, then LU (:234)
        MI355_PHASE(10);  // MM assembly + LU
        // (measured: a branch-free form — one selected LDS index, scale and zero mask per entry — is 2 % slower, the
        // four blocks are taken by disjoint lane ranges and each pass is short)
        const int kk2 = 2 * k;
#pragma unroll
        for (int j = 0; j < K2; ++j) {
          double val = 0.0;
          if (sl < kk2 && j < kk2) {
            const int i = sl;
            if (i < k && j < k) {
              val = (i == j) ? -1 * Amat[i * M + i] : 0.0;
            } else if (i < k && j >= k) {
              const int jj = j - k;  // L^T(i, jj) = L(jj, i)
              val = (jj > i) ? Amat[i * M + jj] : 0.0;
            } else if (i >= k && j < k) {
              const int ii = i - k;  // L(ii, j)
              val = (ii > j) ? Amat[j * M + ii] : 0.0;
            } else {
              val = SSmat[(j - k) * M + (i - k)] * theta;
            }
          }
          mm_row[j] = val;
        }
        lu_factor<K2, W>(mm_row, mm_perm, kk2, sl);
        if (sl < kk2) {  // LDS copy for the side-by-side column solves
#pragma unroll
          for (int j = 0; j < K2; ++j) LUm[sl * K2 + j] = mm_row[j];
          permL[sl] = mm_perm;
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
      for (int e = 0; e < E; ++e) dx[e] = x[e] - xs[e];
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

    if constexpr (!OUTER::kEnabled)
      trace_iteration<E>(a, prob, n, sl, num_iterations, status, f, x_delta, f_delta, gradient_norm, x, g);
    if constexpr (OUTER::kEnabled) {
      if (status != MI355_STATUS_CONTINUE) {
        if (OUTER::step(obj, oa, a, prob, x, num_iterations, nfev, sum_k, sl, stop_num_iterations, stop_gradient_norm,
                        f_start, f, g)) {
          reset_solver();  // f and g are already the next solve's first evaluation
        } else {
          need_fetch = true;
        }
      }
    } else if (status != MI355_STATUS_CONTINUE) {
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
