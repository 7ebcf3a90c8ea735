// auglag_device.hpp — the augmented-Lagrangian path of the reference on the device.
//
//   cppoptlib::solver::AugmentedLagrangian<Problem, Lbfgs<...>>::Minimize     solver/augmented_lagrangian.h
//     OptimizationStep: ToAugmentedLagrangian(...) -> inner Lbfgs::Minimize -> multiplier / penalty update,
//     KKT norm, best-iterate filter;  Progress::Update (IsConstrained branch, solver/progress.h:162-252)
//
// The reference assembles the composite out of type-erased host functors (function_penalty.h:97-246).  Here
// a constrained problem is a list of TERMS from a closed menu (mi355_al_term_kind) — term 0 the objective, then
// the equalities c(x) = 0, then the inequalities g(x) >= 0; a term is a primitive or a sum of primitives — and
// AugLagObjective evaluates
//   L(x) = f + sum_i lambda_i c_i + sum_i rho (0.5 (c_i c_i)) + sum_j [ (1/(2 rho)) max(0, mu_j - rho g_j)^2 - mu_j^2/(2 rho) ]
// node by node in the order the reference's expression templates would (ConstExpression, AddExpression,
// SubExpression, MulExpression with its c == 0 short circuit, ProdExpression, MaxZeroExpression;
// function_expressions.h:38-388), so the inner solver — the unchanged lbfgs_solve_kernel — sees bit-identical
// values and gradients.  Multipliers and the penalty are per-problem data (lambda, mu, rho), the term table is
// shared by the batch and lives in LDS.
//
// The outer iteration is one launch of auglag_outer_kernel per inner solve; all state stays in HBM, and the
// outer kernel compacts the indices of the problems still active for the next inner solve (SolveArgs::problem_map).
#pragma once
#include <type_traits>

#include "objectives.hpp"

namespace mi355 {

constexpr int kAlMaxC = MI355_AL_MAX_CONSTRAINTS;       // per kind (equalities, inequalities)
constexpr int kAlMaxTerms = 1 + 2 * kAlMaxC;
constexpr int kAlMaxRows = MI355_AL_MAX_ROWS;           // primitives in the table
constexpr int kAlRowDoubles = 2 * kAlMaxC + 2 + kAlMaxTerms + 1;  // lambda, mu, rho, k per term; even
// n_eq, n_ineq, (first row, parts, form, k) per term, kind per row
// ... then the constraint FAMILIES (mi355_al_problem::family_*): count of equalities, of inequalities, offset of the
// family block from the start of the blob, one spare
constexpr int kAlTermBase = 2, kAlRowBase = kAlTermBase + 4 * kAlMaxTerms, kAlFamBase = kAlRowBase + kAlMaxRows,
              kAlHeader = kAlFamBase + 4;
static_assert(kAlHeader % 2 == 0, "coefficient rows stay 16-byte aligned");
// A problem may own up to four family constraints per lane of its segment (256 at one problem per wavefront): every
// lane evaluates ITS constraints as ascending chains, so no cross-lane reduction is spent on a constraint.
constexpr int kAlFamilyPerLane = 4;
__host__ __device__ constexpr int al_family_capacity(int W) { return kAlFamilyPerLane * W; }

// std::max / std::clamp as the reference applies them (NaN falls through the comparisons)
__device__ __forceinline__ double std_max(double a, double b) { return (a < b) ? b : a; }
__device__ __forceinline__ double std_clamp(double v, double lo, double hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

// max_j |a_j| the way the reference's loops compute it (`sup = std::max(sup, std::abs(v))`, and lpNorm<Infinity> over the
// Eigen shim): a NaN entry never wins the comparison, so it is skipped.
template <int W, int E>
__device__ __forceinline__ double seg_amax_skipping_nan(const double (&a)[E]) {
  double t[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const double v = __builtin_fabs(a[e]);
    t[e] = (v != v) ? 0.0 : v;
  }
  return seg_max<W>(lane_max<E>(t));
}

// ---- user functors as terms (mi355_al_term_kind >= MI355_AL_TERM_USER) ----------------------------------------------
// The reference composes ANY functor into a constrained problem (function_problem.h:44-74; its non-convex tests
// HS024 / HS029, src/test/augmented_lagrangian_test.cc:945-1150, are three-line user classes; src/examples/
// svm_dual_al.cc:36-81 is a dense quadratic with a precomputed matrix).  On the device a user term is a functor with the
// interface of csrc/objectives.hpp without a workgroup-shared block; it is compiled into a build of the library together
// with the kernels of this path (_build.build(user_objectives=[dict(..., al_term=True)])), and a row of the term table
// whose kind is the functor's objective id (>= 100) evaluates it: load(params, n, sl, scratch, nullptr), then
// eval<W, E>(x, g, n, sl).  `params` are the row's n + 1 coefficients — or, for a functor that declares
// `static constexpr bool kTermParamsFromProblem = true`, the problem's mi355_al_problem::user_params blob: the same
// parameters the functor takes as an OBJECTIVE through mi355_lbfgs_desc::objective_params, so one functor serves as both.
// `scratch` is per-problem LDS of F::kLdsDoubles doubles (shared by the terms of a problem: an evaluation owns it from
// load to the return of eval).  TermList is the closed set of functors of one library build, tried in order.
template <class F, class = void>
struct TermParamsFromProblem : std::false_type {};
template <class F>
struct TermParamsFromProblem<F, std::void_t<decltype(F::kTermParamsFromProblem)>>
    : std::integral_constant<bool, F::kTermParamsFromProblem> {};

struct NoUserTerms {
  static constexpr int kLdsDoubles = 0;
  template <int W, int E>
  __device__ __forceinline__ static double eval(int, const double*, const double*, double*, const double (&)[E],
                                                double (&g)[E], int, int) {
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = 0.0;
    return 0.0;
  }
};
template <int Id, class F>
struct UserTerm {
  static_assert(Id >= MI355_OBJ_USER_FIRST, "user term ids are user objective ids");
  static_assert(F::shared_lds_doubles() == 0, "a functor used as a term cannot own a workgroup-shared LDS block");
  static constexpr int kId = Id;
  using Functor = F;
};
template <class... Ts>
struct TermList;
template <>
struct TermList<> : NoUserTerms {};
template <class T, class... Rest>
struct TermList<T, Rest...> {
  static constexpr int kLdsDoubles = (T::Functor::kLdsDoubles > TermList<Rest...>::kLdsDoubles)
                                         ? T::Functor::kLdsDoubles
                                         : TermList<Rest...>::kLdsDoubles;
  template <int W, int E>
  __device__ __forceinline__ static double eval(int kind, const double* row, const double* user, double* scratch,
                                                const double (&x)[E], double (&g)[E], int n, int sl) {
    if (kind == T::kId) {
      typename T::Functor f;
      f.load(TermParamsFromProblem<typename T::Functor>::value ? user : row, n, sl, scratch, nullptr);
      f.begin_problem(nullptr, 0, 0, sl);
      return f.template eval<W, E>(x, g, n, sl);
    }
    return TermList<Rest...>::template eval<W, E>(kind, row, user, scratch, x, g, n, sl);
  }
};

// FC: capacity for constraint FAMILIES (0: none — the kernels of the closed term table; al_family_capacity(W): the
// kernels that also take mi355_al_problem::family_eq / family_ineq, csrc/auglag_family.hip).
//
// A family is a matrix: constraint i is the affine function c_i(x) = a_i . x - k_i, what a reference user writes as
// `LinearFunctor(a_i) - k_i` (function_expressions.h:497-518) and pushes into the equality / inequality vector of a
// ConstrainedOptimizationProblem (function_problem.h:57-84) — hundreds of them in src/examples/svm_primal_al.cc:139-147.
// On the device lane l of the problem's segment OWNS constraints l, l + W, l + 2W, l + 3W: it forms their values as
// ascending multiply-then-add chains over x (staged in LDS; the matrix is read TRANSPOSED from global memory, so the
// lanes of a wavefront read consecutive addresses), does the scalar work of its constraints (multiplier, clamp, square)
// and stages the results in LDS; the sums over constraints — the composite's value parts and, per coordinate, its
// gradient parts — are then formed in ASCENDING constraint order, the value parts by every lane redundantly, the gradient
// parts by the lane that owns the coordinate (the matrix read row-major).  Every sum is therefore the reference's own
// chain: the family part of the composite is bit-identical to the reference order under every reduction policy.
template <int W, int E, class Terms = NoUserTerms, int FC = 0>
struct AugLagObjective {
  static constexpr int P = W * E;
  static constexpr int kPitch = P + 1;                   // a[0..P) zero padded, then c
  static constexpr int kFamilyCapacity = FC;
  static constexpr int kFamilyChunk = 8;                  // matrix rows whose loads are in flight together (gradient sums)
  static constexpr int kFamilyValueChunk = 4;             // ... and columns of A^T (values: four constraints per lane each)
  static constexpr bool kLargeFootprint = true;          // (lbfgs_kernel.hpp solve_max_waves)
  static_assert(FC == 0 || FC == al_family_capacity(W), "family capacity is four constraints per lane");
  // family block in the blob (global memory): k[FC], A[FC][P] row-major, A^T[P][FC]; in LDS per problem: x[P], three
  // staged scalars per constraint, the family multipliers and the outer step's two scratch copies of them
  static constexpr int kFamilyLds = (FC > 0) ? P + 6 * FC : 0;
  __host__ __device__ static constexpr long long family_block_doubles() {
    return static_cast<long long>(FC) * (1 + 2 * P);
  }
  // per problem: a row (lambda, mu, rho and, when the batch carries its own term constants, k of every term), and two
  // more rows of scratch for the outer step (the state's multipliers entering and leaving it)
  // (+ the scratch of the library's user term functors, TermList::kLdsDoubles)
  static constexpr int kLdsDoubles = 3 * kAlRowDoubles + Terms::kLdsDoubles + kFamilyLds;
  __host__ __device__ static constexpr int shared_lds_doubles() {
    return kAlHeader + kAlMaxRows * kPitch + (kAlMaxRows * kPitch) % 2;
  }
  const double* params;  // device blob: header, then one coefficient row per term (pitch P + 1), then the user blob
  const double* hdr;     // LDS copy (header and rows)
  const double* user;    // mi355_al_problem::user_params (global memory), behind the table
  double* mult;          // LDS, this problem's lambda[0..n_eq), mu[0..n_ineq), rho [, k[0..terms)]
  double* term_scratch;  // LDS, Terms::kLdsDoubles doubles
  int n_eq, n_ineq;
  int own_k;             // index of k[0] in mult, or -1: the constants of the shared term table apply
  // families (FC > 0)
  int f_eq, f_ineq;      // family equalities (constraints [0, f_eq) of the block), inequalities ([f_eq, f_eq + f_ineq))
  const double* fam_k;   // global: k_i
  const double* fam_a;   // global: A[i][j], pitch P
  const double* fam_at;  // global: A^T[j][i], pitch FC
  double* fx;            // LDS [P]: the point of the evaluation in flight
  double* fs1;           // LDS [FC] x 3: staged per-constraint scalars (fs1, fs1 + FC, fs1 + 2 FC)
  double* fmult;         // LDS [FC] x 3: the family multipliers in use, then the outer step's prev / next copies

  __device__ __forceinline__ void load(const double* p, int, int, double* lds_scratch, double* lds_shared) {
    params = p;
    hdr = lds_shared;
    user = p + shared_lds_doubles();
    mult = lds_scratch;
    term_scratch = lds_scratch + 3 * kAlRowDoubles;
    n_eq = static_cast<int>(p[0]);
    n_ineq = static_cast<int>(p[1]);
    own_k = -1;
    f_eq = f_ineq = 0;
    if constexpr (FC > 0) {
      f_eq = static_cast<int>(p[kAlFamBase]);
      f_ineq = static_cast<int>(p[kAlFamBase + 1]);
      fam_k = p + static_cast<long long>(p[kAlFamBase + 2]);
      fam_a = fam_k + FC;
      fam_at = fam_a + static_cast<long long>(FC) * P;
      fx = term_scratch + Terms::kLdsDoubles;
      fs1 = fx + P;
      fmult = fs1 + 3 * FC;
    }
  }
  __device__ __forceinline__ void fill_shared(double* lds_shared, int tid, int nthreads) const {
    const int last = 1 + n_eq + n_ineq - 1;  // rows in use: through the last term's last primitive
    const int last_parts = static_cast<int>(params[kAlTermBase + 4 * last + 1]);   // (a product term holds two rows)
    const int rows = static_cast<int>(params[kAlTermBase + 4 * last]) + (last_parts == MI355_AL_PARTS_PRODUCT ? 2 : last_parts);
    const int total = kAlHeader + rows * kPitch;
    for (int t = tid; t < total; t += nthreads) lds_shared[t] = params[t];
  }
  // per-problem row: (lambda, mu, rho), optionally followed by one constant k per term
  // (with families the row ends with the family multipliers: lambda of the family equalities, mu of the inequalities)
  __device__ __forceinline__ void begin_problem(const double* per_problem, long long prob, int stride, int sl) {
    const int fam = f_eq + f_ineq, table_stride = stride - fam;
    own_k = (table_stride > n_eq + n_ineq + 1) ? n_eq + n_ineq + 1 : -1;
    for (int i = sl; i < table_stride; i += W) mult[i] = per_problem[prob * stride + i];
    if constexpr (FC > 0)
      for (int i = sl; i < fam; i += W) fmult[i] = per_problem[prob * stride + table_stride + i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // src: a table row (count doubles); fsrc: the family multipliers that go with it (FC > 0)
  __device__ __forceinline__ void set_multipliers(const double* src, int count, int sl, const double* fsrc = nullptr) {
    __builtin_amdgcn_wave_barrier();
    for (int i = sl; i < count; i += W) mult[i] = src[i];
    if constexpr (FC > 0)
      for (int i = sl; i < f_eq + f_ineq; i += W) fmult[i] = fsrc[i];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // Values of the family constraints this lane owns at x: c_i = a_i . x - k_i as the ascending chain the reference's
  // `a.dot(x)` is (first product, then + a_j x_j), i = sl + W q.  Stages x in LDS (every lane reads all of it).
  __device__ __forceinline__ void family_values(const double (&x)[E], int n, int sl, double (&cv)[kAlFamilyPerLane]) const {
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int e = 0; e < E; ++e) fx[sl * E + e] = x[e];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const double x0 = fx[0];
#pragma unroll
    for (int q = 0; q < kAlFamilyPerLane; ++q) cv[q] = fam_at[sl + W * q] * x0;
    // (the matrix comes through L2: every load of kFamilyValueChunk steps is in flight before the first product of the chunk —
    //  a load-wait-multiply loop paid one memory round trip per coordinate, profiles/r5_ab_family_loads.txt; the order of
    //  the arithmetic is unchanged)
    int j = 1;
    for (; j + kFamilyValueChunk <= n; j += kFamilyValueChunk) {
      double a[kFamilyValueChunk][kAlFamilyPerLane], xv[kFamilyValueChunk];
#pragma unroll
      for (int u = 0; u < kFamilyValueChunk; ++u) {
        const double* col = fam_at + static_cast<long long>(j + u) * FC + sl;
#pragma unroll
        for (int q = 0; q < kAlFamilyPerLane; ++q) a[u][q] = col[W * q];
        xv[u] = fx[j + u];
      }
#pragma unroll
      for (int u = 0; u < kFamilyValueChunk; ++u) {
#pragma unroll
        for (int q = 0; q < kAlFamilyPerLane; ++q) cv[q] = cv[q] + a[u][q] * xv[u];
      }
    }
    for (; j < n; ++j) {
      const double xj = fx[j];
      const double* col = fam_at + static_cast<long long>(j) * FC + sl;
#pragma unroll
      for (int q = 0; q < kAlFamilyPerLane; ++q) cv[q] = cv[q] + col[W * q] * xj;
    }
#pragma unroll
    for (int q = 0; q < kAlFamilyPerLane; ++q) cv[q] = cv[q] - fam_k[sl + W * q];
  }

  // Rows [i0, i0 + count) of the row-major family matrix, this lane's E coordinates of each, all loads in flight at once
  // (count <= kFamilyChunk; rows past `last` are not touched).
  __device__ __forceinline__ void family_rows(int i0, int last, int sl, double (&rows)[kFamilyChunk][E]) const {
#pragma unroll
    for (int u = 0; u < kFamilyChunk; ++u) {
      const int i = (i0 + u < last) ? i0 + u : last - 1;
      const double* row = fam_a + static_cast<long long>(i) * P + sl * E;
#pragma unroll
      for (int e = 0; e < E; ++e) rows[u][e] = row[e];
    }
  }

  // Value (segment uniform) and gradient of the primitive in table row r.
  __device__ __forceinline__ double primitive(int r, const double (&x)[E], double (&g)[E], int n, int sl) const {
    const int kind = static_cast<int>(hdr[kAlRowBase + r]);
    const double* row = hdr + kAlHeader + r * kPitch;
    double v;
    if (kind >= MI355_AL_TERM_USER) {
      v = Terms::template eval<W, E>(kind, row, user, term_scratch, x, g, n, sl);
    } else if (kind == MI355_AL_TERM_ROSENBROCK) {
      RosenbrockObjective rb;
      v = rb.template eval<W, E>(x, g, n, sl);
    } else {
      double tt[E];
      if (kind == MI355_AL_TERM_DIAG_QUADRATIC) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double a = row[sl * E + e];
          tt[e] = (a * x[e]) * x[e];
          g[e] = (2.0 * a) * x[e];
        }
      } else if (kind == MI355_AL_TERM_LINEAR || kind == MI355_AL_TERM_SQUARED_AFFINE) {
#pragma unroll
        for (int e = 0; e < E; ++e) {
          const double a = row[sl * E + e];
          tt[e] = a * x[e];
          g[e] = a;
        }
      } else {  // squared norm
#pragma unroll
        for (int e = 0; e < E; ++e) {
          tt[e] = x[e] * x[e];
          g[e] = 2.0 * x[e];
        }
      }
      v = seg_sum<W>(lane_tree_sum<E>(tt));
      if (kind == MI355_AL_TERM_DIAG_QUADRATIC) v = v + row[P];
      if (kind == MI355_AL_TERM_SQUARED_AFFINE) {  // r = a.x - c: value r r, gradient (2 r) a
        const double r = v - row[P];
#pragma unroll
        for (int e = 0; e < E; ++e) g[e] = (2.0 * r) * g[e];
        v = r * r;
      }
    }
    return v;
  }

  // Value and gradient of term t: the sum of its primitives, left to right (AddExpression) — or the product of its two
  // (ProdExpression) — then its form (v, v - k, k - v).
  __device__ __forceinline__ double term(int t, const double (&x)[E], double (&g)[E], int n, int sl) const {
    const int first = static_cast<int>(hdr[kAlTermBase + 4 * t]);
    const int parts = static_cast<int>(hdr[kAlTermBase + 4 * t + 1]);
    const int form = static_cast<int>(hdr[kAlTermBase + 4 * t + 2]);
    const double k = (own_k >= 0) ? mult[own_k + t] : hdr[kAlTermBase + 4 * t + 3];
    double v = primitive(first, x, g, n, sl);
    if (parts == MI355_AL_PARTS_PRODUCT) {  // ProdExpression (function_expressions.h:282-293): (f g)' = g f' + f g'
      double g2[E];
      const double v2 = primitive(first + 1, x, g2, n, sl);
#pragma unroll
      for (int e = 0; e < E; ++e) g[e] = v2 * g[e] + v * g2[e];
      v = v * v2;
    }
    for (int r = 1; r < parts; ++r) {
      double g2[E];
      const double v2 = primitive(first + r, x, g2, n, sl);
      v = v + v2;
#pragma unroll
      for (int e = 0; e < E; ++e) g[e] = g[e] + g2[e];
    }
    if (form == MI355_AL_FORM_VALUE_MINUS_K) {
#pragma unroll
      for (int e = 0; e < E; ++e) g[e] = g[e] - 0.0;
      v = v - k;
    } else if (form == MI355_AL_FORM_K_MINUS_VALUE) {
#pragma unroll
      for (int e = 0; e < E; ++e) g[e] = 0.0 - g[e];
      v = k - v;
    }
    return v;
  }

  // MulExpression (function_expressions.h:203-236)
  __device__ __forceinline__ static double scale(double c, double v, double (&g)[E]) {
    const bool zero = (c == 0.0);
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = zero ? 0.0 : c * g[e];
    return zero ? 0.0 : c * v;
  }
  // ProdExpression of a function with itself (:262-271)
  __device__ __forceinline__ static double square(double v, double (&g)[E]) {
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = v * g[e] + v * g[e];
    return v * v;
  }

  template <int WW, int EE>
  __device__ __forceinline__ double eval(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(WW == W && EE == E, "mapping");
    const double fv = term(0, x, g, n, sl);
    const double rho = mult[n_eq + n_ineq];
    // equalities feed two separately accumulated parts: FormLagrangianPart (function_penalty.h:97-108) and
    // FormPenaltyPart (:115-127); the reference evaluates c twice, to the same bits
    double lv = 0.0, pv = 0.0, lpart[E], ppart[E], cg[E], tg[E];
#pragma unroll
    for (int e = 0; e < E; ++e) lpart[e] = ppart[e] = 0.0;
    for (int c = 0; c < n_eq; ++c) {
      const double cv = term(1 + c, x, cg, n, sl);
#pragma unroll
      for (int e = 0; e < E; ++e) tg[e] = cg[e];
      double tv = scale(mult[c], cv, tg);
      lv = lv + tv;
#pragma unroll
      for (int e = 0; e < E; ++e) lpart[e] = lpart[e] + tg[e];
      tv = square(cv, cg);
      tv = scale(0.5, tv, cg);
      tv = scale(rho, tv, cg);
      pv = pv + tv;
#pragma unroll
      for (int e = 0; e < E; ++e) ppart[e] = ppart[e] + cg[e];
    }
    const double half_inv_rho = 1.0 / (2.0 * rho);   // (used where rho > 0 only)
    const int fam = f_eq + f_ineq;
    if constexpr (FC > 0) {
      if (fam > 0) {
        // the scalar work of the constraints this lane owns, staged for the ordered sums below:
        //   equality i    fs1 = c_i, fs2 = lambda_i c_i (MulExpression), fs3 = rho (0.5 (c_i c_i))
        //   inequality i  fs1 = max(0, mu_i - rho g_i), fs2 = (1 / (2 rho)) fs1^2, fs3 = mu_i^2 / (2 rho)
        double cv[kAlFamilyPerLane];
        family_values(x, n, sl, cv);
        double* const fs2 = fs1 + FC;
        double* const fs3 = fs2 + FC;
#pragma unroll
        for (int q = 0; q < kAlFamilyPerLane; ++q) {
          const int i = sl + W * q;
          const double c = cv[q];
          if (i < f_eq) {
            const double lam = fmult[i];
            fs1[i] = c;
            fs2[i] = (lam == 0.0) ? 0.0 : lam * c;
            double t = c * c;
            t = 0.5 * t;
            fs3[i] = (rho == 0.0) ? 0.0 : rho * t;
          } else if (i < fam) {
            const double m = fmult[i];
            double t = (rho == 0.0) ? 0.0 : rho * c;
            t = m - t;
            t = (t <= 0.0) ? 0.0 : t;
            fs1[i] = t;
            const double sq = t * t;
            fs2[i] = (half_inv_rho == 0.0) ? 0.0 : half_inv_rho * sq;
            fs3[i] = (m * m) * half_inv_rho;
          }
        }
        segment_lds_fence();
        // the family equalities continue both chains of the table's (FormLagrangianPart, FormPenaltyPart)
        for (int i = 0; i < f_eq; ++i) lv = lv + fs2[i];
        for (int i = 0; i < f_eq; ++i) pv = pv + fs3[i];
        for (int i0 = 0; i0 < f_eq; i0 += kFamilyChunk) {
          double rows[kFamilyChunk][E];
          family_rows(i0, f_eq, sl, rows);
#pragma unroll
          for (int u = 0; u < kFamilyChunk; ++u) {
            const int i = i0 + u;
            if (i < f_eq) {
              const double lam = fmult[i], c = fs1[i];
#pragma unroll
              for (int e = 0; e < E; ++e) {
                const double a = rows[u][e] - 0.0;                   // gradient of `F - k`
                lpart[e] = lpart[e] + ((lam == 0.0) ? 0.0 : lam * a);
                double t = c * a + c * a;                            // ProdExpression of c with itself
                t = 0.5 * t;
                t = (rho == 0.0) ? 0.0 : rho * t;
                ppart[e] = ppart[e] + t;
              }
            }
          }
        }
      }
    }
    double value = fv + lv;
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = g[e] + lpart[e];
    value = value + pv;
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = g[e] + ppart[e];
    // FormInequalityPart (:154-194), Powell-Hestenes-Rockafellar; skipped as a whole for rho <= 0
    double iv = 0.0;
#pragma unroll
    for (int e = 0; e < E; ++e) lpart[e] = 0.0;
    if (!(rho <= 0.0)) {
      for (int c = 0; c < n_ineq; ++c) {
        const double m = mult[n_eq + c];
        double tv = term(1 + n_eq + c, x, cg, n, sl);
        tv = scale(rho, tv, cg);
        tv = m - tv;
        const bool clamp = (tv <= 0.0);  // MaxZeroExpression (:347-363)
#pragma unroll
        for (int e = 0; e < E; ++e) cg[e] = clamp ? 0.0 : (0.0 - cg[e]);
        tv = clamp ? 0.0 : tv;
        tv = square(tv, cg);
        tv = scale(half_inv_rho, tv, cg);
        iv = iv + tv;
        iv = iv - (m * m) * half_inv_rho;
#pragma unroll
        for (int e = 0; e < E; ++e) lpart[e] = (lpart[e] + cg[e]) - 0.0;
      }
      if constexpr (FC > 0) {
        const double* const fs2 = fs1 + FC;
        const double* const fs3 = fs2 + FC;
        for (int i = f_eq; i < fam; ++i) {   // the family inequalities continue the table's chain
          iv = iv + fs2[i];
          iv = iv - fs3[i];
        }
        for (int i0 = f_eq; i0 < fam; i0 += kFamilyChunk) {
          double rows[kFamilyChunk][E];
          family_rows(i0, fam, sl, rows);
#pragma unroll
          for (int u = 0; u < kFamilyChunk; ++u) {
            const int i = i0 + u;
            if (i < fam) {
              const double t = fs1[i];
              const bool clamp = (t == 0.0);     // staged max(0, .): zero exactly when MaxZeroExpression clamped
#pragma unroll
              for (int e = 0; e < E; ++e) {
                double c = rows[u][e] - 0.0;
                c = (rho == 0.0) ? 0.0 : rho * c;
                c = clamp ? 0.0 : (0.0 - c);
                c = t * c + t * c;
                c = (half_inv_rho == 0.0) ? 0.0 : half_inv_rho * c;
                lpart[e] = (lpart[e] + c) - 0.0;
              }
            }
          }
        }
      }
    }
    value = value + iv;
    // (padding coordinates j >= n carry zeros through every node — except under a non-finite multiplier or penalty,
    //  where inf * 0 would leave a NaN that the solver's reductions over the padded width must not see)
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = (sl * E + e < n) ? g[e] + lpart[e] : 0.0;
    return value;
  }
};

// State of the outer loop, one row per problem, resident in HBM for the whole solve.
struct AugLagOuterArgs {
  double* x;              // [B][n]  AugmentedLagrangeState::x
  const double* x_inner;  // [B][n]  result of the inner solve
  double* mult;           // [B][stride] lambda, mu, rho  (the inner solve's per-problem data)
  double* violation;      // [B]  max_violation
  double* kkt;            // [B]  max_lagrangian_gradient
  unsigned char* active;  // [B]  1 while Status::Continue
  unsigned char* autoscaled;  // [B] 1 = rho was auto-scaled in this outer iteration (previous state's penalty 0)
  mi355_al_progress* progress;  // [B]
  const mi355_lbfgs_progress* inner_progress;  // [B], of the inner solve just finished
  // best-iterate filter (augmented_lagrangian.h, UpdateBestIterateInPlace)
  double* best_x;         // [B][n]
  double* best_mult;      // [B][stride]
  double* best_scalars;   // [B][4] objective, violation, kkt, recorded
  unsigned int* remaining;  // problems still active after this launch ...
  int* next_map;            // ... and their indices, in arrival order (the next inner solve's problem_map)
  const int* cur_map;       // null: this launch covers problems 0..B-1; else the B problems cur_map[0..B-1]
  const unsigned int* count_dev;  // null, or the run-time length of cur_map (B then only sizes the grid)
  const double* obj_params;
  // box of an Lbfgsb inner solver (device, n doubles each), or null: the KKT norm is then the plain sup norm
  // (augmented_lagrangian.h ComputeLagrangianGradientKktNorm: ProjectedGradientInfNorm when the inner solver has one)
  const double* lower;
  const double* upper;
  mi355_al_config config;
  long long B;
  int n, stride;
  int phase;              // 0: auto-scale the initial penalty; 1: update after an inner solve
};

// ComputeAutoScaledPenalty on the first outer iteration, when the caller's penalty is 0: the new penalty goes into the
// problem's row (global) and into the objective's LDS copy.  obj.begin_problem must have run.
// The outer step is COLD code inside the persistent solve kernel (once per inner solve).  Its per-lane global addresses
// (state rows, best iterate, bounds: base + lane offset) are loop invariants of the kernel; left alone the compiler forms
// them all ahead of the iteration loop and, the register file being full there, spills them (96-288 B of scratch in the
// fused kernels, profiles/r4_kernel_resources.txt).  With the lane index opaque they are formed where they are used.
__device__ __forceinline__ int al_cold_lane(int sl) {
  asm volatile("" : "+v"(sl));
  return sl;
}

template <int W, int E, class Terms, int FC>
__device__ __forceinline__ void al_autoscale(AugLagObjective<W, E, Terms, FC>& obj, const AugLagOuterArgs& a, long long prob,
                                             const double (&xs)[E], int sl_in) {
  const int sl = al_cold_lane(sl_in);
  const int n = a.n, n_eq = obj.n_eq, n_ineq = obj.n_ineq, nm = n_eq + n_ineq;
  const mi355_al_config& cfg = a.config;
  const double penalty = obj.mult[nm];
  if (!(cfg.auto_scale_initial_penalty && penalty == 0.0)) return;
  double g[E];
  double objective_magnitude = __builtin_fabs(obj.term(0, xs, g, n, sl));
  objective_magnitude = std_max(objective_magnitude, 1.0);
  double squared_residual_sum = 0.0;
  for (int c = 0; c < n_eq; ++c) {
    const double value = obj.term(1 + c, xs, g, n, sl);
    squared_residual_sum += 0.5 * value * value;
  }
  const int fam = obj.f_eq + obj.f_ineq;
  if constexpr (FC > 0) {
    if (fam > 0) {   // the values of the family constraints, staged: the sum runs over them in the reference's order
      double cv[kAlFamilyPerLane];
      obj.family_values(xs, n, sl, cv);
#pragma unroll
      for (int q = 0; q < kAlFamilyPerLane; ++q)
        if (sl + W * q < fam) obj.fs1[sl + W * q] = cv[q];
      segment_lds_fence();
      for (int i = 0; i < obj.f_eq; ++i) {
        const double value = obj.fs1[i];
        squared_residual_sum += 0.5 * value * value;
      }
    }
  }
  for (int c = 0; c < n_ineq; ++c) {
    const double value = obj.term(1 + n_eq + c, xs, g, n, sl);
    if (value < 0.0) squared_residual_sum += 0.5 * value * value;
  }
  if constexpr (FC > 0) {
    for (int i = obj.f_eq; i < fam; ++i) {
      const double value = obj.fs1[i];
      if (value < 0.0) squared_residual_sum += 0.5 * value * value;
    }
  }
  const double denom = std_max(squared_residual_sum, 1.0);
  const double rho = std_clamp(cfg.penalty_auto_objective_scale * objective_magnitude / denom, cfg.penalty_auto_min,
                               cfg.penalty_auto_max);
  __builtin_amdgcn_wave_barrier();
  if (sl == 0) {
    obj.mult[nm] = rho;
    a.mult[prob * a.stride + nm] = rho;
    a.autoscaled[prob] = 1;
  }
  segment_lds_fence();
}

// One outer step after an inner solve (AugmentedLagrangian::OptimizationStep past the inner Minimize, then
// Progress::Update): xs is the state's x the inner solve started from, xn its result; obj holds the multipliers the
// inner solve used.  Returns the outer status.  On Continue the state rows (x, multipliers, violation, KKT norm,
// progress, best iterate) are updated and obj holds the next multipliers; otherwise the rows hold the returned state.
// start_value: null, or the composite at xs under the multipliers of the solve that just ran (the solve's own first
// evaluation) — Progress::Update's previous_value unless the penalty was auto-scaled in this step; next_value /
// next_gradient receive the composite at xn under the next multipliers (Progress::Update's current_value), which
// is also the first evaluation of the next inner solve.
template <int W, int E, class Terms, int FC>
__device__ __forceinline__ int al_outer_step(AugLagObjective<W, E, Terms, FC>& obj, const AugLagOuterArgs& a, long long prob,
                                             const double (&xs)[E], const double (&xn)[E], unsigned inner_its,
                                             unsigned inner_nfev, unsigned inner_sum_k, int sl_in,
                                             const double* start_value, double& next_value, double (&next_gradient)[E]) {
  const int sl = al_cold_lane(sl_in);
  const int n = a.n, n_eq = obj.n_eq, n_ineq = obj.n_ineq, nm = n_eq + n_ineq;
  const mi355_al_config& cfg = a.config;
  double* const prevm = obj.mult + kAlRowDoubles;   // the state's multipliers entering the step
  double* const nextm = prevm + kAlRowDoubles;      // ... and leaving it
  const double penalty = obj.mult[nm];
  double g[E], buf[E];
  // every global read of this step is issued up front: the step is a chain of small reductions, and the wavefront
  // fences between them would otherwise serialise the memory latencies
  const double previous_max_violation = a.violation[prob];
  double* const bs = a.best_scalars + prob * 4;
  const double best_objective = bs[0], best_violation = bs[1];
  const bool recorded = bs[3] != 0.0;
  mi355_al_progress pr = a.progress[prob];
  const bool was_autoscaled = a.autoscaled[prob] != 0;
  // ---- multiplier update (OptimizationStep) ---------------------------------------------------------
  for (int i = sl; i <= nm; i += W) prevm[i] = obj.mult[i];
  // families: the multipliers entering / leaving the step, next to the ones in use
  const int fam = obj.f_eq + obj.f_ineq, table_stride = a.stride - fam;
  double* const fprev = (FC > 0) ? obj.fmult + FC : nullptr;
  double* const fnext = (FC > 0) ? obj.fmult + 2 * FC : nullptr;
  double fam_violation = 0.0;
  if constexpr (FC > 0) {
    if (fam > 0) {
      // one evaluation of the constraints this lane owns serves their multiplier updates and the violation; the
      // updated multipliers are staged for the ordered KKT sums below
      double cv[kAlFamilyPerLane];
      obj.family_values(xn, n, sl, cv);
#pragma unroll
      for (int q = 0; q < kAlFamilyPerLane; ++q) {
        const int i = sl + W * q;
        if (i < fam) {
          const double old = obj.fmult[i];
          fprev[i] = old;
          double cand;
          if (i < obj.f_eq) {
            fam_violation = std_max(fam_violation, __builtin_fabs(cv[q]));
            cand = old + penalty * cv[q];
            cand = __builtin_isfinite(cand) ? std_clamp(cand, -cfg.multiplier_max, cfg.multiplier_max) : 0.0;
          } else {
            fam_violation = std_max(fam_violation, std_max(0.0, -cv[q]));
            cand = std_max(0.0, old - penalty * cv[q]);
            cand = __builtin_isfinite(cand) ? std_clamp(cand, 0.0, cfg.multiplier_max) : 0.0;
          }
          fnext[i] = cand;
        }
      }
      fam_violation = seg_max<W>(fam_violation);   // (a NaN value never wins a comparison: order does not matter)
    }
  }
  segment_lds_fence();
  // One evaluation of every constraint serves both its multiplier update and its column of the KKT sum
  // (ComputeLagrangianGradientKktNorm: sum_grad = grad f, += lambda_i grad c_i in order, -= mu_j grad g_j in order,
  // with the UPDATED multipliers; the update of constraint i needs only c_i).  (Every call of term() here and in
  // eval() uses BOTH its value and its gradient.  A second loop that re-evaluated the constraints for their gradients
  // only was miscompiled by hipcc 7.2 — wrong gradient for a two-primitive term whose first primitive is linear, at
  // two coordinates per lane; keeping the dead value alive with an empty asm cured it — so no such call exists.)
  const double objective = obj.term(0, xn, g, n, sl);
  double max_violation = 0.0;
  for (int c = 0; c < n_eq; ++c) {
    const double cv = obj.term(1 + c, xn, buf, n, sl);
    max_violation = std_max(max_violation, __builtin_fabs(cv));
    double cand = prevm[c] + penalty * cv;
    cand = __builtin_isfinite(cand) ? std_clamp(cand, -cfg.multiplier_max, cfg.multiplier_max) : 0.0;
    if (sl == 0) nextm[c] = cand;
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = g[e] + cand * buf[e];
  }
  if constexpr (FC > 0) {   // ... + lambda_i grad c_i over the family equalities, in order
    constexpr int kChunk = AugLagObjective<W, E, Terms, FC>::kFamilyChunk;
    for (int i0 = 0; i0 < obj.f_eq; i0 += kChunk) {
      double rows[kChunk][E];
      obj.family_rows(i0, obj.f_eq, sl, rows);
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if (i0 + u < obj.f_eq) {
          const double cand = fnext[i0 + u];
#pragma unroll
          for (int e = 0; e < E; ++e) g[e] = g[e] + cand * (rows[u][e] - 0.0);
        }
      }
    }
  }
  for (int c = 0; c < n_ineq; ++c) {
    const double cv = obj.term(1 + n_eq + c, xn, buf, n, sl);
    const double violation = std_max(0.0, -cv);
    max_violation = std_max(max_violation, violation);
    double cand = std_max(0.0, prevm[n_eq + c] - penalty * cv);
    cand = __builtin_isfinite(cand) ? std_clamp(cand, 0.0, cfg.multiplier_max) : 0.0;
    if (sl == 0) nextm[n_eq + c] = cand;
#pragma unroll
    for (int e = 0; e < E; ++e) g[e] = g[e] - cand * buf[e];
  }
  if constexpr (FC > 0) {   // ... - mu_j grad g_j over the family inequalities
    constexpr int kChunk = AugLagObjective<W, E, Terms, FC>::kFamilyChunk;
    for (int i0 = obj.f_eq; i0 < fam; i0 += kChunk) {
      double rows[kChunk][E];
      obj.family_rows(i0, fam, sl, rows);
#pragma unroll
      for (int u = 0; u < kChunk; ++u) {
        if (i0 + u < fam) {
          const double cand = fnext[i0 + u];
#pragma unroll
          for (int e = 0; e < E; ++e) g[e] = g[e] - cand * (rows[u][e] - 0.0);
        }
      }
    }
    max_violation = std_max(max_violation, fam_violation);
  }
  segment_lds_fence();
  if (a.lower != nullptr) {  // Lbfgsb::ProjectedGradientInfNorm (lbfgsb.h:105-118)
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      if (j < n) {
        if (xn[e] <= a.lower[j] && g[e] > 0.0) g[e] = 0.0;
        if (xn[e] >= a.upper[j] && g[e] < 0.0) g[e] = 0.0;
      }
    }
  }
  const double kkt = seg_amax_skipping_nan<W, E>(g);
  // ---- UpdateBestIterateInPlace (candidate.penalty is still the pre-growth one) ---------------------
  bool take = false;
  {
    double bad[E];
#pragma unroll
    for (int e = 0; e < E; ++e) bad[e] = __builtin_isfinite(xn[e]) ? 0.0 : 1.0;
    const bool finite = __builtin_isfinite(objective) && __builtin_isfinite(max_violation) &&
                        seg_max<W>(lane_max<E>(bad)) == 0.0;
    constexpr double tol = 1e-5;  // filter_feasibility_tolerance
    const bool cf = max_violation <= tol, bf = best_violation <= tol;
    if (finite) {
      if (!recorded) take = true;
      else if (cf && !bf) take = true;
      else if (!cf && bf) take = false;
      else if (cf && bf) take = objective < best_objective;
      else take = max_violation < best_violation || (max_violation == best_violation && objective < best_objective);
    }
    if (take) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const int j = sl * E + e;
        if (j < n) a.best_x[prob * n + j] = xn[e];
      }
      for (int i = sl; i < nm; i += W) a.best_mult[prob * a.stride + i] = nextm[i];
      if constexpr (FC > 0)
        for (int i = sl; i < fam; i += W) a.best_mult[prob * a.stride + table_stride + i] = fnext[i];
      if (sl == 0) {
        a.best_mult[prob * a.stride + nm] = penalty;
        bs[0] = objective;
        bs[1] = max_violation;
        bs[2] = kkt;
        bs[3] = 1.0;
      }
    }
  }
  // ---- penalty growth -----------------------------------------------------------------------------------
  const bool shrank = max_violation <= cfg.violation_shrink_ratio * previous_max_violation;
  const double next_penalty = shrank ? penalty : penalty * cfg.penalty_growth_factor;
  if (sl == 0) {
    nextm[nm] = next_penalty;
    if (was_autoscaled) prevm[nm] = 0.0;  // the state entering this step still had penalty 0
  }
  segment_lds_fence();
  // ---- Progress::Update, IsConstrained branch (progress.h:162-252) ----------------------------------
  double previous_value;
  if (start_value != nullptr && !was_autoscaled) {
    previous_value = *start_value;  // same function, same point, same arithmetic as the solve's first evaluation
  } else {
    obj.set_multipliers(prevm, nm + 1, sl, fprev);
    previous_value = obj.template eval<W, E>(xs, g, n, sl);
  }
  obj.set_multipliers(nextm, nm + 1, sl, fnext);
  const double current_value = obj.template eval<W, E>(xn, g, n, sl);
  next_value = current_value;
#pragma unroll
  for (int e = 0; e < E; ++e) next_gradient[e] = g[e];
  pr.num_iterations += 1;
  pr.f_delta = __builtin_fabs(current_value - previous_value);
  double dx[E];
#pragma unroll
  for (int e = 0; e < E; ++e) dx[e] = xn[e] - xs[e];
  pr.x_delta = seg_amax_skipping_nan<W, E>(dx);
  pr.gradient_norm = seg_amax_skipping_nan<W, E>(g);
  pr.inner_iterations += inner_its;
  pr.nfev += inner_nfev;
  pr.sum_k += inner_sum_k;
  int status;
  if (cfg.outer_num_iterations > 0 && pr.num_iterations > cfg.outer_num_iterations) {
    status = MI355_STATUS_ITERATION_LIMIT;
  } else if (!__builtin_isfinite(max_violation) || !__builtin_isfinite(kkt)) {
    status = MI355_STATUS_ITERATION_LIMIT;
  } else {
    const bool primal_feasible = __builtin_fabs(max_violation) <= cfg.constraint_threshold;
    const bool kkt_stationary = (cfg.kkt_stationarity_threshold <= 0.0) || (kkt <= cfg.kkt_stationarity_threshold);
    status = (primal_feasible && kkt_stationary) ? MI355_STATUS_FINISHED : MI355_STATUS_CONTINUE;
  }
  pr.status = status;
  // ---- write the next state; a problem that stops hands back its best iterate ------------------------
  // (when the best iterate is the one recorded in this launch it is still in registers; an older one is read
  // back from the arrays a previous launch wrote)
  const bool done = status != MI355_STATUS_CONTINUE;
  const bool use_stored = done && recorded && !take;
  if (done && take && sl == 0) nextm[nm] = penalty;
  segment_lds_fence();
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    if (j < n) a.x[prob * n + j] = use_stored ? a.best_x[prob * n + j] : xn[e];
  }
  for (int i = sl; i <= nm; i += W)
    a.mult[prob * a.stride + i] = use_stored ? a.best_mult[prob * a.stride + i] : nextm[i];
  if constexpr (FC > 0)
    for (int i = sl; i < fam; i += W)
      a.mult[prob * a.stride + table_stride + i] =
          use_stored ? a.best_mult[prob * a.stride + table_stride + i] : fnext[i];
  if (sl == 0) {
    a.violation[prob] = use_stored ? a.best_scalars[prob * 4 + 1] : max_violation;
    a.kkt[prob] = use_stored ? a.best_scalars[prob * 4 + 2] : kkt;
    a.progress[prob] = pr;
    a.autoscaled[prob] = 0;
  }
  return status;
}

// Fused form of the outer loop: the OUTER policy of lbfgs_solve_kernel and lbfgsb_solve_kernel.  A wavefront segment
// keeps its problem through every outer iteration — inner solve, outer step, next inner solve from the same registers —
// so the whole batch is ONE launch with no host round trip, and no iteration waits for the slowest problem of the
// previous one.  SolveArgs::x0 and the outer arguments' x are the same array (the state's x); SolveArgs::stop is the
// inner solver's stopping record after ConfigureInnerSubproblem (f_delta = 0).
template <int W, int E, class Terms = NoUserTerms, int FC = 0>
struct AugLagOuterLoop {
  static constexpr bool kEnabled = true;
  using Args = AugLagOuterArgs;
  using Obj = AugLagObjective<W, E, Terms, FC>;

  // A problem was fetched (x = the state's x, obj.begin_problem done): initial penalty, warm-up stopping test of the
  // first inner solve (ConfigureInnerSubproblem)
  __device__ __forceinline__ static void begin(Obj& obj, const Args& oa, const SolveArgs& a, long long prob,
                                               const double (&x)[E], int sl, unsigned long long& stop_num_iterations,
                                               double& stop_gradient_norm) {
    // (a problem handed over by the lock-step loop after some outer iterations is past both)
    const bool first = oa.progress[prob].num_iterations == 0;
    if (first) al_autoscale(obj, oa, prob, x, sl);
    const bool warmup = first && (obj.n_eq + obj.n_ineq + obj.f_eq + obj.f_ineq > 0) &&
                        oa.config.warmup_max_inner_iterations > 0;
    stop_num_iterations = warmup ? static_cast<unsigned long long>(oa.config.warmup_max_inner_iterations)
                                 : a.stop.num_iterations;
    stop_gradient_norm = warmup ? oa.config.warmup_inner_gradient_tolerance : a.stop.gradient_norm;
  }

  // An inner solve stopped at x.  True: the outer loop continues with another solve from x (the multipliers of the
  // next state are in obj); false: the problem is finished and its rows hold the returned state.
  // f_start: the first evaluation of the solve that just ran; on a true return f and g hold the first evaluation of
  // the next one (the outer step computes it anyway, for Progress::Update).
  __device__ __forceinline__ static bool step(Obj& obj, const Args& oa, const SolveArgs& a, long long prob,
                                              const double (&x)[E], unsigned inner_iterations, unsigned inner_nfev,
                                              unsigned inner_sum_k, int sl, unsigned long long& stop_num_iterations,
                                              double& stop_gradient_norm, double f_start, double& f, double (&g)[E]) {
    // the x this solve started from: every lane re-reads the coordinates it wrote itself (previous step, or the
    // caller's start point)
    double xs[E];
    const int slc = al_cold_lane(sl);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = slc * E + e;
      xs[e] = (j < oa.n) ? oa.x[prob * oa.n + j] : 0.0;
    }
    const int status =
        al_outer_step(obj, oa, prob, xs, x, inner_iterations, inner_nfev, inner_sum_k, sl, &f_start, f, g);
    // The scalars of the state (written by the segment's first lane) are read by all its lanes in the next step: lanes
    // of ONE wavefront, through one L1 — a workgroup-scope fence (the stores and loads drained, no cache maintenance) is
    // what that needs.  (`__threadfence()` stood here until round 5: an agent-scope fence — `buffer_wbl2 sc1` +
    // `buffer_inv sc1`, a write-back and invalidate of the XCD's L2 — per outer step of every problem; measurably
    // harmless at the rates this path reaches, but not what the hand-off asks for.)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    stop_num_iterations = a.stop.num_iterations;
    stop_gradient_norm = a.stop.gradient_norm;
    return status == MI355_STATUS_CONTINUE;
  }
};

// Lock-step form of the outer loop: one launch per outer iteration over the problems still active; phase 0
// auto-scales the initial penalties.
template <int W, int E, class Terms = NoUserTerms>
__global__ __launch_bounds__(256) void auglag_outer_kernel(AugLagOuterArgs a) {
  extern __shared__ double lds[];
  using Obj = AugLagObjective<W, E, Terms>;
  constexpr int kSegs = kWave / W;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave_in_block = threadIdx.x / kWave;
  const int seg = lane / W;
  const int sl = lane % W;
  const int n = a.n;
  Obj obj;
  obj.load(a.obj_params, n, sl, lds + Obj::shared_lds_doubles() + (wave_in_block * kSegs + seg) * Obj::kLdsDoubles, lds);
  obj.fill_shared(lds, static_cast<int>(threadIdx.x), static_cast<int>(blockDim.x));
  __syncthreads();
  const long long slot = (static_cast<long long>(blockIdx.x) * (blockDim.x / kWave) + wave_in_block) * kSegs + seg;
  if (slot >= (a.count_dev ? static_cast<long long>(*a.count_dev) : a.B)) return;
  const long long prob = a.cur_map ? a.cur_map[slot] : slot;
  if (!a.active[prob]) return;
  double xs[E], xn[E];
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int j = sl * E + e;
    xs[e] = (j < n) ? a.x[prob * n + j] : 0.0;
    xn[e] = (j < n && a.phase != 0) ? a.x_inner[prob * n + j] : 0.0;
  }
  obj.begin_problem(a.mult, prob, a.stride, sl);
  if (a.phase == 0) {
    al_autoscale(obj, a, prob, xs, sl);
    return;
  }
  const mi355_lbfgs_progress inner = a.inner_progress[prob];
  double next_value, next_gradient[E];  // (the next launch of the inner kernel evaluates them again)
  const int status = al_outer_step(obj, a, prob, xs, xn, inner.num_iterations, inner.nfev, inner.sum_k, sl,
                                         nullptr, next_value, next_gradient);
  if (sl == 0) {
    const bool done = status != MI355_STATUS_CONTINUE;
    a.active[prob] = done ? 0 : 1;
    if (!done) a.next_map[atomicAdd(a.remaining, 1u)] = static_cast<int>(prob);
  }
}

}  // namespace mi355
