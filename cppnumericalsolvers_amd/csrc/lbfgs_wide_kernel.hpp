// lbfgs_wide_kernel.hpp — Lbfgs<F, m, MoreThuente>::Minimize for problems LARGER than a wavefront can hold (n > 256).
//
// The reference is dynamic in n (function.h: FunctionXd; src/examples/svm_primal_lbfgs.cc runs at any dimension); the
// kernels of lbfgs_kernel.hpp keep a problem's vectors in the registers of one wavefront segment and its history in
// LDS, which ends at 64 lanes x 4 coordinates.  Beyond that a problem is owned by a WORKGROUP of four wavefronts and
// its state lives in HBM (an L2 / Infinity-Cache resident workspace per resident workgroup):
//
//   x, g (current)  xn, gn (the line search's trial point = the next iterate)  d (two-loop result)
//   S[m][n], Y[m][n] (the correction ring of solver/lbfgs.h:248-280)
//
// Thread t owns the coordinates j = t, t + T, t + 2T, ... (coalesced; T = 256 threads, 1024 for n >= 32768), so every element-wise update is private to a
// thread and the only communication is (i) the reductions and (ii) whatever an objective needs from other coordinates
// (the neighbours x[j +- 1] of the chained Rosenbrock function, all of x for a user's matrix-vector product), read back
// from memory after a workgroup barrier.  This is the regime the state-streaming model of SURVEY section 8d describes:
// the vectors really move, and memory bandwidth bounds the kernel.  Three storage forms execute the same operations in
// the same order (WideVec below; lbfgs_wide_dispatch.hpp picks by measured speed): everything in registers but the ring
// (n <= 512), the workspace with the direction in LDS (n <= 4096), the workspace alone.
//
// Arithmetic: the exact policy only (separate multiplies and adds, -ffp-contract=off).  Summation order (the CPU twin
// used by the tests restates it as its `strided` reduction policy, width 256): thread t adds its own terms in
// ascending order onto 0.0, then the 256 partial sums go through the pairwise tree (xor-butterfly inside
// a wavefront, (w0 + w1) + (w2 + w3) across the four).  Everything else — two-loop recursion :145-196, descent test
// and fallback :199-224, More-Thuente cvsrch / cstep (more_thuente.h:137-407), s / y / curvature test / ring / gamma
// :248-298, Progress::Update (progress.h:153-327) — is the reference's sequence of operations.
#pragma once
#include <type_traits>
#include "lbfgs_kernel.hpp"
#include "more_thuente_device.hpp"
#include "hager_zhang_device.hpp"

namespace mi355 {

// Threads per workgroup: 256 (four wavefronts), or 1024 (sixteen) for n >= kWideBigN -- the same sixteen wavefronts per CU
// either way, but a quarter of the resident workspaces and four times the threads on a problem when the batch is small.
// The summation order follows (thread t of T adds j = t, t + T, ...; then the pairwise tree over T partial sums), so the
// choice depends on n alone.  Device code reads the count from blockDim.x (wide_threads()).
constexpr int kWideThreads = 256;
constexpr int kWideThreadsBig = 1024;
constexpr int kWideBigN = 32768;
constexpr int kWideMaxWaves = kWideThreadsBig / kWave;
__device__ __forceinline__ int wide_threads() { return static_cast<int>(blockDim.x); }
// pairwise tree over the wavefronts' partial results (4 or 16 of them)
template <class Op>
__device__ __forceinline__ double wide_tree(const double* r, Op op) {
  if (blockDim.x == kWideThreads) return op(op(r[0], r[1]), op(r[2], r[3]));
  const double t0 = op(op(r[0], r[1]), op(r[2], r[3])), t1 = op(op(r[4], r[5]), op(r[6], r[7]));
  const double t2 = op(op(r[8], r[9]), op(r[10], r[11])), t3 = op(op(r[12], r[13]), op(r[14], r[15]));
  return op(op(t0, t1), op(t2, t3));
}
constexpr int kWideMaxM = 32;  // MI355_LBFGS_MAX_M

struct WideArgs {
  const double* x0;      // [B][n]
  double* x_out;         // [B][n]
  double* f_out;         // [B]
  double* g_out;         // [B][n] or null
  mi355_lbfgs_progress* progress_out;  // [B] or null
  const double* obj_params;
  double* workspace;     // gridDim.x * ws_stride doubles
  long long ws_stride;   // wide_ws_doubles(n, m, E)
  unsigned long long* next_problem;
  long long B;
  int n, m;
  int d_in_lds;          // 1: the launch carries n doubles of dynamic LDS for the direction (memory form, moderate n)
  int linesearch;        // mi355_linesearch
  int hess_from_functor; // Second mode, non-constant Hessian: the preconditioner from the functor's hess_diag at every iterate
  mi355_lbfgs_stop stop;
};

// ---- workgroup reductions: all 256 threads call, all receive the result -------------------------------------------
__device__ __forceinline__ double wide_sum(double partial, double* red) {
  const double s = seg_sum<kWave>(partial);
  __syncthreads();  // the previous reduction's readers are done with red[]
  if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = s;
  __syncthreads();
  return wide_tree(red, [](double a, double b) { return a + b; });
}
// K sums and L maxima with one barrier pair (red: (K + L) * kWideMaxWaves doubles)
template <int K, int L>
__device__ __forceinline__ void wide_reduce(double (&sums)[K], double (&maxs)[L > 0 ? L : 1], double* red) {
  double ws[K], wm[L > 0 ? L : 1];
#pragma unroll
  for (int q = 0; q < K; ++q) ws[q] = seg_sum<kWave>(sums[q]);
#pragma unroll
  for (int q = 0; q < L; ++q) wm[q] = seg_max<kWave>(maxs[q]);
  __syncthreads();
  if ((threadIdx.x & (kWave - 1)) == 0) {
    const int w = threadIdx.x / kWave;
#pragma unroll
    for (int q = 0; q < K; ++q) red[q * kWideMaxWaves + w] = ws[q];
#pragma unroll
    for (int q = 0; q < L; ++q) red[(K + q) * kWideMaxWaves + w] = wm[q];
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < K; ++q) sums[q] = wide_tree(red + q * kWideMaxWaves, [](double a, double b) { return a + b; });
#pragma unroll
  for (int q = 0; q < L; ++q) maxs[q] = wide_tree(red + (K + q) * kWideMaxWaves, [](double a, double b) { return vmax(a, b); });
}
// ---- where a problem-sized vector lives ----------------------------------------------------------------------------
// E == 0: in the workgroup's HBM workspace, any n.  E > 0: in registers, E coordinates per thread (n <= 256 E): x, g, the
// trial point, its gradient and the direction never touch memory then, only the correction ring does -- less than half the
// traffic of the memory form.  Used for E = 2 (n <= 512): beyond that the register count costs more occupancy than the
// traffic is worth, and the memory form with its direction in LDS is faster (dispatch_wide.hip has the numbers).  The two forms execute the same operations in the same
// order (thread t still adds its terms for j = t, t + 256, ... in ascending order), so they share one twin.
template <int E>
struct WideVec {
  double* mem;
  double reg[E > 0 ? E : 1];
  __device__ __forceinline__ double& at(int j, int e) {
    if constexpr (E > 0) {
      (void)j;
      return reg[e];
    } else {
      (void)e;
      return mem[j];
    }
  }
  __device__ __forceinline__ double get(int j, int e) const {
    if constexpr (E > 0) {
      (void)j;
      return reg[e];
    } else {
      (void)e;
      return mem[j];
    }
  }
};
// body(j, e) for every coordinate j = tid + 256 e < n of the calling thread, ascending
template <int E, class F>
__device__ __forceinline__ void wide_for(int n, F&& body) {
  if constexpr (E > 0) {
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = static_cast<int>(threadIdx.x) + kWideThreads * e;
      if (j < n) body(j, e);
    }
  } else {
    for (int j = threadIdx.x; j < n; j += wide_threads()) body(j, 0);
  }
}

// ---- objectives: value (workgroup uniform) and gradient ----------------------------------------------------------------
// x must be visible to the whole workgroup (the caller has passed a barrier since it was written); g[j] is written by
// the thread that owns j.
// x: the point; g: receives the gradient; dir != null: *gd_out = g . dir in the same pass and the same reduction.
// xmem: n doubles of workspace through which a register-resident x reaches the neighbouring threads (chained objective).
struct RosenbrockWide {
  __device__ __forceinline__ void load(const double*, int) {}
  template <int E>
  __device__ __forceinline__ double eval(WideVec<E>& x, WideVec<E>& g, int n, double* red, double* xmem,
                                         const WideVec<E>* dir = nullptr, double* gd_out = nullptr) const {
    const double* xs = x.mem;
    if constexpr (E > 0) {
      __syncthreads();  // the previous evaluation's neighbour reads are done
      wide_for<E>(n, [&](int j, int e) { xmem[j] = x.reg[e]; });
      xs = xmem;
    }
    __syncthreads();    // x is visible to the workgroup
    double acc = 0.0, gd = 0.0;
    wide_for<E>(n, [&](int j, int e) {
      const double xj = x.get(j, e);
      const bool has_a = (j + 1 < n), has_b = (j > 0);
      double a = 0.0, b = 0.0;
      if (has_a) {
        const double t1 = 1.0 - xj;
        const double t2 = xs[j + 1] - xj * xj;
        acc = acc + (t1 * t1 + (100.0 * t2) * t2);
        a = -2.0 * (1.0 - xj) + (200.0 * t2) * (-2.0 * xj);
      }
      if (has_b) {
        const double xm = xs[j - 1];
        b = 200.0 * (xj - xm * xm);
      }
      const double gj = (has_a && has_b) ? (a + b) : (has_a ? a : b);
      g.at(j, e) = gj;
      if (dir) gd = gd + gj * dir->get(j, e);
    });
    double sums[2] = {acc, gd}, none[1] = {0.0};
    wide_reduce<2, 0>(sums, none, red);
    if (gd_out) *gd_out = sums[1];
    return sums[0];
  }
  // H_jj at the point x (in memory, visible to the workgroup), as RosenbrockObjectiveT::hess_diag
  __device__ __forceinline__ double hess_diag(const double* x, int j, int n) const {
    const bool has_a = (j + 1 < n), has_b = (j > 0);
    const double a = has_a ? ((1200.0 * x[j]) * x[j] - 400.0 * x[j + 1]) + 2.0 : 0.0;
    return (has_a && has_b) ? (a + 200.0) : (has_a ? a : (has_b ? 200.0 : 0.0));
  }
};
// Does a workgroup functor offer diag H(x)?  (optional member: double hess_diag(const double* x_in_memory, int j, int n))
template <class Obj, class = void>
struct HasWideHessDiag : std::false_type {};
template <class Obj>
struct HasWideHessDiag<Obj, std::void_t<decltype(&Obj::hess_diag)>> : std::true_type {};

struct DiagQuadraticWide {
  const double* a_;
  double c_;
  __device__ __forceinline__ void load(const double* params, int n) {
    a_ = params;
    c_ = params[n];
  }
  template <int E>
  __device__ __forceinline__ double eval(WideVec<E>& x, WideVec<E>& g, int n, double* red, double*,
                                         const WideVec<E>* dir = nullptr, double* gd_out = nullptr) const {
    double acc = 0.0, gd = 0.0;
    wide_for<E>(n, [&](int j, int e) {
      const double aj = a_[j], xj = x.get(j, e);
      acc = acc + (aj * xj) * xj;
      const double gj = (2.0 * aj) * xj;
      g.at(j, e) = gj;
      if (dir) gd = gd + gj * dir->get(j, e);
    });
    double sums[2] = {acc, gd}, none[1] = {0.0};
    wide_reduce<2, 0>(sums, none, red);
    if (gd_out) *gd_out = sums[1];
    return sums[0] + c_;
  }
};

// Workspace per resident workgroup (doubles, np = n rounded up to even):
//   E == 0:  x | g | xn | gn | d | S[m] | Y[m]        (5 + 2m) np
//   E  > 0:  xmem | S[m] | Y[m]                        (1 + 2m) np
__host__ __device__ inline long long wide_ws_doubles(int n, int m, int E) {
  const long long np = (static_cast<long long>(n) + 1) & ~1LL;
  return ((E > 0 ? 1 : 5) + 2LL * m) * np;
}

template <class Obj, int E, int LS = MI355_LS_MORE_THUENTE, int T = kWideThreads>
__global__ __launch_bounds__(T) void lbfgs_wide_kernel(const WideArgs a) {
  static_assert(E == 0 || T == kWideThreads, "the register form is built for 256 threads");
  __shared__ double red[8 * kWideMaxWaves];
  __shared__ double sy_mem[kWideMaxM];     // s_i . y_i of the stored pairs, by ring slot
  __shared__ double alpha_mem[kWideMaxM];  // alpha by chronological index
  __shared__ double past_f[MI355_LBFGS_MAX_PAST];
  __shared__ long long fetched;
  constexpr double eps = 2.220446049250313e-16;
  const int tid = threadIdx.x;
  const int n = a.n, m = a.m;
  const long long np = (static_cast<long long>(n) + 1) & ~1LL;  // vectors start on 16-byte boundaries
  double* const ws = a.workspace + static_cast<long long>(blockIdx.x) * a.ws_stride;
  WideVec<E> xc, gc, xn, gn, d;   // current.x, current.gradient, next.x (trial point), next.gradient, direction
  double* xmem = ws;              // E > 0: where a register-resident x meets its neighbours
  double* S;
  if constexpr (E > 0) {
    xc.mem = gc.mem = xn.mem = gn.mem = d.mem = nullptr;
    S = ws + np;
  } else {
    xc.mem = ws;
    gc.mem = ws + np;
    xn.mem = ws + 2 * np;
    gn.mem = ws + 3 * np;
    // the direction is the busiest vector (read and written by every step of the two-loop recursion): when the launch
    // gave the workgroup n doubles of LDS it lives there, which removes half of the memory form's traffic
    extern __shared__ double lds_direction[];
    d.mem = a.d_in_lds ? lds_direction : ws + 4 * np;
    S = ws + 5 * np;
  }
  double* const Y = S + static_cast<long long>(m) * np;
  Obj obj;
  obj.load(a.obj_params, n);

  while (true) {
    __syncthreads();
    if (tid == 0) fetched = static_cast<long long>(atomicAdd(a.next_problem, 1ULL));
    __syncthreads();
    const long long prob = fetched;
    if (prob >= a.B) break;

    // ---- Solver::Minimize, solver.h:189-194: evaluate the start, InitializeSolver ------------------------------
    wide_for<E>(n, [&](int j, int e) { xc.at(j, e) = a.x0[prob * n + j]; });
    double f = obj.template eval<E>(xc, gc, n, red, xmem);
    // carried with the current iterate: x . x (relative_eps, :93-95), max |g_j| and max |x_j| (progress.h:195, :301)
    double xx_cur, gmax_cur, xmax_cur;
    {
      double sums[1] = {0.0}, maxs[2] = {0.0, 0.0};
      wide_for<E>(n, [&](int j, int e) {
        const double xj = xc.get(j, e);
        sums[0] = sums[0] + xj * xj;
        const double ta = __builtin_fabs(gc.get(j, e)), tb = __builtin_fabs(xj);
        if (maxs[0] < ta) maxs[0] = ta;
        if (maxs[1] < tb) maxs[1] = tb;
      });
      wide_reduce<1, 2>(sums, maxs, red);
      xx_cur = sums[0];
      gmax_cur = maxs[0];
      xmax_cur = maxs[1];
    }
    unsigned nfev = 1, sum_k = 0;
    int mem_count = 0, mem_pos = 0;
    double scaling_factor = 1.0;
    unsigned long long num_iterations = 0;
    int x_delta_violations = 0, f_delta_violations = 0;
    double x_delta = 0.0, f_delta = 0.0, gradient_norm = 0.0;
    int status = MI355_STATUS_CONTINUE;
    bool past_init = false;
    int past_pos = 0;

    do {
      // ================= Lbfgs::OptimizationStep, lbfgs.h:89-303 ==================================================
      __syncthreads();  // sy_mem of the pair stored at the end of the previous step (written by thread 0)
      const double relative_eps = eps * dmax(1.0, __builtin_sqrt(xx_cur));                 // :93-95
      const int k = mem_count;
      sum_k += static_cast<unsigned>(k);
      // The two-loop recursion :145-196 with every element-wise update fused with the inner product that follows it:
      // one sweep per step.  Pairs with |s.y| < eps are skipped (:165, :189).
      auto slot_of = [&](int i) { return (mem_count < m) ? i : ((mem_pos + i) % m); };
      auto active = [&](int i) { return !(__builtin_fabs(sy_mem[slot_of(i)]) < eps); };
      auto next_down = [&](int i) { do { --i; } while (i >= 0 && !active(i)); return i; };
      auto next_up = [&](int i) { do { ++i; } while (i < k && !active(i)); return i; };
      const int first = next_up(-1);   // oldest active pair (k: none)
      double gd, dd = 0.0;
      // the factor at the centre of the recursion: scaling_factor_ (:181), or for a Second-mode function
      // M^-1_jj = 1 / (|H_jj(x)| + eps) rebuilt from the functor at this iterate (:129-134, :177-179)
      [[maybe_unused]] const double* xs_hess = xc.mem;
      if constexpr (HasWideHessDiag<Obj>::value && E > 0) {
        if (a.hess_from_functor) {   // (xmem holds the last trial point, which need not be the current iterate)
          __syncthreads();
          wide_for<E>(n, [&](int j, int e) { xmem[j] = xc.reg[e]; });
          __syncthreads();
          xs_hess = xmem;
        }
      }
      auto centre = [&](int j) -> double {
        if constexpr (HasWideHessDiag<Obj>::value) {
          if (a.hess_from_functor) return 1.0 / (__builtin_fabs(obj.hess_diag(xs_hess, j, n)) + eps);
        }
        (void)j;
        return scaling_factor;
      };
      if (first >= k) {
        // no usable pair: d = g * scaling_factor_, with g.d and (for alpha_init) d.d on the way
        double sums[2] = {0.0, 0.0}, none[1] = {0.0};
        wide_for<E>(n, [&](int j, int e) {
          const double gj = gc.get(j, e);
          const double dj = centre(j) * gj;
          d.at(j, e) = dj;
          sums[0] = sums[0] + gj * dj;
          sums[1] = sums[1] + dj * dj;
        });
        wide_reduce<2, 0>(sums, none, red);
        gd = sums[0];
        dd = sums[1];
      } else {
        // first loop, newest -> oldest
        int i = next_down(k);
        double acc = 0.0;
        {
          const double* const s = S + static_cast<long long>(slot_of(i)) * np;
          wide_for<E>(n, [&](int j, int e) {
            const double gj = gc.get(j, e);
            d.at(j, e) = gj;                                                                 // :145
            acc = acc + s[j] * gj;
          });
        }
        while (true) {
          const int idx = slot_of(i);
          const double alpha = (1.0 / sy_mem[idx]) * wide_sum(acc, red);                     // rho * s.d
          if (tid == 0) alpha_mem[i] = alpha;
          const double* const y = Y + static_cast<long long>(idx) * np;
          const int inext = next_down(i);
          acc = 0.0;
          if (inext >= 0) {
            const double* const s = S + static_cast<long long>(slot_of(inext)) * np;
            wide_for<E>(n, [&](int j, int e) {
              const double dj = d.get(j, e) - alpha * y[j];
              d.at(j, e) = dj;
              acc = acc + s[j] * dj;
            });
            i = inext;
          } else {
            // last step of the first loop: the scaling (:181) and the second loop's first inner product ride along
            const double* const y0 = Y + static_cast<long long>(slot_of(first)) * np;
            wide_for<E>(n, [&](int j, int e) {
              const double dj = centre(j) * (d.get(j, e) - alpha * y[j]);
              d.at(j, e) = dj;
              acc = acc + y0[j] * dj;
            });
            break;
          }
        }
        // second loop, oldest -> newest (:185-196)
        i = first;
        while (true) {
          const int idx = slot_of(i);
          const double beta = (1.0 / sy_mem[idx]) * wide_sum(acc, red);                      // rho * y.d
          const double c = alpha_mem[i] - beta;   // (written by thread 0 before at least one barrier pair)
          const double* const s = S + static_cast<long long>(idx) * np;
          const int inext = next_up(i);
          acc = 0.0;
          if (inext < k) {
            const double* const y = Y + static_cast<long long>(slot_of(inext)) * np;
            wide_for<E>(n, [&](int j, int e) {
              const double dj = s[j] * c + d.get(j, e);
              d.at(j, e) = dj;
              acc = acc + y[j] * dj;
            });
            i = inext;
          } else {
            wide_for<E>(n, [&](int j, int e) {
              const double dj = s[j] * c + d.get(j, e);
              d.at(j, e) = dj;
              acc = acc + gc.get(j, e) * dj;
            });
            gd = wide_sum(acc, red);
            break;
          }
        }
      }
      const double descent_direction = -gd;                                                // :199
      double alpha_init = 1.0;                                                             // :207-213
      if (mem_count == 0) {
        const double dn = __builtin_sqrt(dd);
        alpha_init = (dn > eps) ? 1.0 / dn : 1.0;
      }
      double dginit = descent_direction;  // g . (-d), bit for bit
      if (!__builtin_isfinite(descent_direction) || descent_direction > -eps * relative_eps) {  // :214-224
        double gg = 0.0;
        wide_for<E>(n, [&](int j, int e) {
          const double gj = gc.get(j, e);
          d.at(j, e) = -gj;
          gg = gg + gj * gj;
        });
        gg = wide_sum(gg, red);
        mem_count = 0;
        mem_pos = 0;
        const double gnorm = __builtin_sqrt(gg);
        alpha_init = (gnorm > eps) ? 1.0 / gnorm : 1.0;
        dginit = gg;  // g . (-d) = g . g: the products are the same, so is the sum
      }

      // ---- LineSearch::Search along -d (:231-232).  next = (xn, gn, f_next) ----------------------------------------
      double f_next = f;
      if constexpr (LS == MI355_LS_HAGER_ZHANG) {
        // HagerZhang (hager_zhang.h:100-116, :282-548): the scalar state machine of hager_zhang_device.hpp over this
        // kernel's evaluation; a failed search hands back the start state
        double stp = alpha_init, alpha_acc = 0.0;
        bool ls_failed = false;
        const int ls_nfev = hz_search_core(
            [&](double alpha, double& phi, double& dphi) {
              if constexpr (E == 0) __syncthreads();  // every reader of the previous trial point is done
              wide_for<E>(n, [&](int j, int e) { xn.at(j, e) = xc.get(j, e) - alpha * d.get(j, e); });
              double gdn;
              phi = obj.template eval<E>(xn, gn, n, red, xmem, &d, &gdn);
              dphi = -gdn;
            },
            f_next, stp, dginit, ls_failed, alpha_acc);
        nfev += static_cast<unsigned>(ls_nfev);
        if (ls_failed) {
          wide_for<E>(n, [&](int j, int e) {
            xn.at(j, e) = xc.get(j, e);
            gn.at(j, e) = gc.get(j, e);
          });
          f_next = f;
        }
      } else {   // MoreThuente (more_thuente.h:120-256)
        double stp = alpha_init;
        int info = 0, infoc = 1;
        constexpr double xtol = 1e-15, ftol = 1e-4, gtol = 0.9, stpmin = 1e-15, stpmax = 1e15, xtrapf = 4.0;
        constexpr int maxfev = 20;
        int ls_nfev = 0;
        if (dginit >= 0.0) {  // :152-156: nothing evaluated, next = current
          wide_for<E>(n, [&](int j, int e) {
            xn.at(j, e) = xc.get(j, e);
            gn.at(j, e) = gc.get(j, e);
          });
        } else {
          bool brackt = false, stage1 = true;
          const double finit = f;
          const double dgtest = ftol * dginit;
          double width = stpmax - stpmin;
          double width1 = 2.0 * width;
          double stx = 0.0, fx = finit, dgx = dginit;
          double sty = 0.0, fy = finit, dgy = dginit;
          double stmin, stmax;
          while (true) {
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
            if constexpr (E == 0) __syncthreads();  // every reader of the previous trial point is done
            wide_for<E>(n, [&](int j, int e) { xn.at(j, e) = stp * (-d.get(j, e)) + xc.get(j, e); });   // wa + stp * s
            double gdn;
            f_next = obj.template eval<E>(xn, gn, n, red, xmem, &d, &gdn);
            ls_nfev++;
            const double dg = -gdn;                                                          // g . s
            const double ftest1 = finit + stp * dgtest;
            if ((brackt & ((stp <= stmin) | (stp >= stmax))) | (infoc == 0)) info = 6;
            if ((stp == stpmax) & (f_next <= ftest1) & (dg <= dgtest)) info = 5;
            if ((stp == stpmin) & ((f_next > ftest1) | (dg >= dgtest))) info = 4;
            if (ls_nfev >= maxfev) info = 3;
            if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
            if ((f_next <= ftest1) & (__builtin_fabs(dg) <= gtol * (-dginit))) info = 1;
            if (info != 0) break;
            if (stage1 & (f_next <= ftest1) & (dg >= dmin(ftol, gtol) * dginit)) stage1 = false;
            const bool modified = stage1 & (f_next <= fx) & (f_next > ftest1);
            StepInterval iv;
            iv.stx = stx; iv.sty = sty; iv.stp = stp; iv.brackt = brackt; iv.info = infoc; iv.rc = 0;
            iv.fx = modified ? fx - stx * dgtest : fx;
            iv.fy = modified ? fy - sty * dgtest : fy;
            iv.dx = modified ? dgx - dgtest : dgx;
            iv.dy = modified ? dgy - dgtest : dgy;
            const double fm = modified ? f_next - stp * dgtest : f_next;
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
          }
        }
        nfev += static_cast<unsigned>(ls_nfev);
      }

      // ---- :239-298 and the norms of Progress::Update (progress.h:188-195), one pass ----------------------------------
      const double f_prev = f;
      if (__builtin_isfinite(f_next)) {
        // s = next.x - current.x, y = next.gradient - current.gradient
        double sums[4] = {0.0, 0.0, 0.0, 0.0};   // s.y, s.s, y.y, next.x . next.x
        double maxs[3] = {0.0, 0.0, 0.0};        // max |s_j| (x_delta), max |next.g_j|, max |next.x_j|
        wide_for<E>(n, [&](int j, int e) {
          const double xj = xn.get(j, e), gj = gn.get(j, e);
          const double sj = xj - xc.get(j, e), yj = gj - gc.get(j, e);
          sums[0] = sums[0] + sj * yj;
          sums[1] = sums[1] + sj * sj;
          sums[2] = sums[2] + yj * yj;
          sums[3] = sums[3] + xj * xj;
          const double ta = __builtin_fabs(sj), tb = __builtin_fabs(gj), tc = __builtin_fabs(xj);
          if (maxs[0] < ta) maxs[0] = ta;
          if (maxs[1] < tb) maxs[1] = tb;
          if (maxs[2] < tc) maxs[2] = tc;
        });
        wide_reduce<4, 3>(sums, maxs, red);
        const double sy = sums[0], ss = sums[1], yy = sums[2];
        const double sy_threshold = eps * __builtin_sqrt(ss) * __builtin_sqrt(yy);          // :266
        if (sy > sy_threshold) {                                                             // :267-280
          const int slot = (mem_count < m) ? mem_count : mem_pos;
          double* const s = S + static_cast<long long>(slot) * np;
          double* const y = Y + static_cast<long long>(slot) * np;
          wide_for<E>(n, [&](int j, int e) {
            s[j] = xn.get(j, e) - xc.get(j, e);
            y[j] = gn.get(j, e) - gc.get(j, e);
          });
          if (tid == 0) sy_mem[slot] = sy;
          if (mem_count < m) {
            mem_count++;
          } else {
            mem_pos = (mem_pos + 1) % m;
          }
        }
        constexpr double fallback_value = 1e7;                                              // :289-298
        if (yy > eps) {
          const double temp_scaling = sy / yy;   // y.dot(s): the same products as s.dot(y)
          if (__builtin_isfinite(temp_scaling) && __builtin_fabs(temp_scaling) <= fallback_value)
            scaling_factor = dmax(temp_scaling, eps);
        }
        // next becomes current
        if constexpr (E > 0) {
#pragma unroll
          for (int e = 0; e < E; ++e) {
            xc.reg[e] = xn.reg[e];
            gc.reg[e] = gn.reg[e];
          }
        } else {
          double* t = xc.mem; xc.mem = xn.mem; xn.mem = t;
          t = gc.mem; gc.mem = gn.mem; gn.mem = t;
        }
        f = f_next;
        x_delta = maxs[0];
        xx_cur = sums[3];
        gmax_cur = maxs[1];
        xmax_cur = maxs[2];
      } else {
        x_delta = 0.0;   // :239-241: the step returns `current`, so previous == current for the tests below
      }

      // ================= Progress::Update, progress.h:153-327 ====================================================
      num_iterations++;
      f_delta = __builtin_fabs(f - f_prev);
      gradient_norm = gmax_cur;
      status = MI355_STATUS_CONTINUE;
      const mi355_lbfgs_stop& st = a.stop;
      do {
        if ((st.num_iterations > 0) && (num_iterations > st.num_iterations)) {              // :212-216
          status = MI355_STATUS_ITERATION_LIMIT;
          break;
        }
        if ((st.x_delta > 0) && (x_delta < st.x_delta)) {                                    // :254-262
          x_delta_violations++;
          if (x_delta_violations >= st.x_delta_violations) {
            status = MI355_STATUS_X_DELTA_VIOLATION;
            break;
          }
        } else {
          x_delta_violations = 0;
        }
        if ((st.f_delta > 0) &&                                                              // :263-277
            (f_delta < st.f_delta * (st.f_delta_relative
                                         ? dmax(dmax(__builtin_fabs(f), __builtin_fabs(f_prev)), 1.0)
                                         : 1.0))) {
          f_delta_violations++;
          if (f_delta_violations >= st.f_delta_violations) {
            status = MI355_STATUS_F_DELTA_VIOLATION;
            break;
          }
        } else {
          f_delta_violations = 0;
        }
        if (st.past > 0) {                                                                   // :280-298
          const int p = st.past;
          __syncthreads();
          if (!past_init) {
            if (tid < p) past_f[tid] = f;
            past_init = true;
            past_pos = 0;
          }
          __syncthreads();
          bool fired = false;
          if (num_iterations > static_cast<unsigned long long>(p)) {
            const double pf = past_f[past_pos];
            const double rate = __builtin_fabs(pf - f) / dmax(1.0, __builtin_fabs(f));
            fired = rate < st.past_delta;
          }
          if (fired) {
            status = MI355_STATUS_F_DELTA_VIOLATION;
            break;
          }
          __syncthreads();
          if (tid == 0) past_f[past_pos] = f;
          past_pos = (past_pos + 1) % p;
        }
        if (st.gradient_norm > 0) {                                                          // :299-317
          const double scale = st.gradient_norm_relative ? dmax(1.0, xmax_cur) : 1.0;
          if (gradient_norm < st.gradient_norm * scale) {
            status = MI355_STATUS_GRADIENT_NORM_VIOLATION;
            break;
          }
        }
      } while (false);
    } while (status == MI355_STATUS_CONTINUE);

    // ---- results ---------------------------------------------------------------------------------------------------
    wide_for<E>(n, [&](int j, int e) {
      a.x_out[prob * n + j] = xc.get(j, e);
      if (a.g_out) a.g_out[prob * n + j] = gc.get(j, e);
    });
    if (tid == 0) {
      a.f_out[prob] = f;
      if (a.progress_out) {
        mi355_lbfgs_progress pr;
        pr.status = status;
        pr.num_iterations = static_cast<unsigned>(num_iterations);
        pr.nfev = nfev;
        pr.sum_k = sum_k;
        pr.x_delta = x_delta;
        pr.f_delta = f_delta;
        pr.gradient_norm = gradient_norm;
        a.progress_out[prob] = pr;
      }
    }
  }
}

}  // namespace mi355
