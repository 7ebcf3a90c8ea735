// cppoptlib/mi355/device_twin.h — how a function tells the MI355X engine what to run for it.
//
// The reference hands its solvers an arbitrary host functor (FunctionCRTP::operator(), function_base.h:103-120) and,
// where the static type is in the way, erases it behind FunctionExpr<TScalar, TMode, TDim> (function_base.h:194-260).  A
// GPU engine needs the objective as device code, so every function that reaches a solver carries a DEVICE TWIN: the id of
// a kernel objective (mi355_objective) with its parameter blobs, and / or a term of the augmented-Lagrangian menu
// (mi355_al_term_kind).  There are two ways to state it:
//
//  * the STATIC protocol — members of the function type, read at compile time by Lbfgs<F> / Lbfgsb<F> / Bfgs<F>:
//        static constexpr int kDeviceObjective;         std::vector<double> DeviceParams() const;       (objective)
//        static constexpr int kAlTermKind;              std::vector<double> AlCoefficients(int n) const; (AL primitive)
//    plus the optional members listed with the traits below;
//  * the RUN-TIME record — `TwinRecord`, what the type-erased FunctionExpr stores next to its host clone, so that
//    `Lbfgs<FunctionExprXd>` (src/examples/simple.cc:22,58) works on whatever was assigned to the wrapper.  A function
//    type states it in one line, by returning a record from the builders in cppoptlib/mi355/objectives.h:
//        auto DeviceTwin() const { return cppoptlib::mi355::twin::DiagQuadratic({5, 100}, 5); }
//    Records compose the way the reference's expression templates do (`f - k`, `k - f`, `-1 * f`, `f + g`, `f * g`).
//
// A function without a twin still converts into a FunctionExpr and evaluates on the host; handing it to a solver fails
// loudly (there is no CPU fallback).
#ifndef CPPOPTLIB_MI355_DEVICE_TWIN_H_
#define CPPOPTLIB_MI355_DEVICE_TWIN_H_

#include <array>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "context.h"

namespace cppoptlib::mi355 {

// FNV-1a over the bit patterns of `count` doubles, chained through `seed`: the parameter-blob hash a function type
// with a large blob computes once at construction (`uint64_t DeviceParamsHash() const`), so that a batch of B functions
// is checked for shared parameters with B integer comparisons.
inline uint64_t HashDoubles(const double* data, size_t count, uint64_t seed = 1469598103934665603ull) {
  uint64_t h = seed;
  for (size_t i = 0; i < count; ++i) {
    uint64_t bits;
    std::memcpy(&bits, data + i, sizeof bits);
    h = (h ^ bits) * 1099511628211ull;
    h ^= h >> 29;
  }
  return h;
}

// ---- the static protocol: traits over the function type ------------------------------------------------------------
template <class F, class = void>
struct HasDeviceParamsHash : std::false_type {};
template <class F>
struct HasDeviceParamsHash<F, std::void_t<decltype(std::declval<const F&>().DeviceParamsHash())>> : std::true_type {};

template <class F, class = void>
struct HasDeviceParamsOfDimension : std::false_type {};
template <class F>
struct HasDeviceParamsOfDimension<F, std::void_t<decltype(std::declval<const F&>().DeviceParams(1))>> : std::true_type {};

template <class F, class = void>
struct HasDeviceObjective : std::false_type {};
template <class F>
struct HasDeviceObjective<F, std::void_t<decltype(F::kDeviceObjective),
                                         decltype(std::declval<const F&>().DeviceParams())>>
    : std::true_type {};
template <class F>
struct HasDeviceObjective<F, std::enable_if_t<HasDeviceParamsOfDimension<F>::value, std::void_t<decltype(F::kDeviceObjective)>>>
    : std::true_type {};

// A Second-mode function whose Hessian is not constant says so with `static constexpr bool kDeviceHessianFromFunctor =
// true`: its device functor has a hess_diag and Lbfgs asks the kernel to rebuild the preconditioner at every iterate
// (mi355_lbfgs_desc::hessian_from_functor) instead of uploading DeviceHessianDiagonal() once.
template <class F, class = void>
struct HessianFromFunctor : std::false_type {};
template <class F>
struct HessianFromFunctor<F, std::void_t<decltype(F::kDeviceHessianFromFunctor)>>
    : std::integral_constant<bool, F::kDeviceHessianFromFunctor> {};

template <class F, class = void>
struct HasDeviceHessianDiagonal : std::false_type {};
template <class F>
struct HasDeviceHessianDiagonal<F, std::void_t<decltype(std::declval<const F&>().DeviceHessianDiagonal())>>
    : std::true_type {};

// A function type may name a second device twin that evaluates the same function in a fused / re-associated form
// (`static constexpr int kDeviceObjectiveFused`): the solvers take it when the caller asks for MI355_ARITH_FMA
// (SetArithmetic) and the reference-order twin otherwise.  For the ridge functors that is the normal-equation form
// (MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM: x*, f* within 1e-6 of the reference, ~4 x the throughput of id 2).
template <class F, class = void>
struct FusedDeviceObjective {
  static constexpr int Of(int /*arithmetic*/) { return F::kDeviceObjective; }
};
template <class F>
struct FusedDeviceObjective<F, std::void_t<decltype(F::kDeviceObjectiveFused)>> {
  static constexpr int Of(int arithmetic) {
    return arithmetic == MI355_ARITH_FMA ? F::kDeviceObjectiveFused : F::kDeviceObjective;
  }
};

// Objectives whose device twin needs data per problem (e.g. the right-hand side y) expose
//     std::vector<double> DevicePerProblem() const;
template <class F, class = void>
struct HasPerProblemData : std::false_type {};
template <class F>
struct HasPerProblemData<F, std::void_t<decltype(std::declval<const F&>().DevicePerProblem())>> : std::true_type {};

// Function types with an OWN-PARAMETERS device form (`kDeviceObjectiveOwnMatrix`, `DeviceOwnMatrixParams()`,
// `DeviceOwnMatrixRow()`, `DeviceFingerprint()`): a batch whose functions do NOT share their parameters — a different
// matrix A per problem, what a reference program gets from building `SquaredError(A_b, y_b)` once per data set
// (README.md:126-160) — is solved with every problem's own parameters in its per-problem row.
template <class F, class = void>
struct HasOwnMatrixForm : std::false_type {};
template <class F>
struct HasOwnMatrixForm<F, std::void_t<decltype(F::kDeviceObjectiveOwnMatrix),
                                       decltype(std::declval<const F&>().DeviceOwnMatrixRow()),
                                       decltype(std::declval<const F&>().DeviceFingerprint())>> : std::true_type {};

// AL primitives: `static constexpr int kAlTermKind` + `std::vector<double> AlCoefficients(int n) const` (row [n + 1]).
template <class F, class = void>
struct IsAlPrimitive : std::false_type {};
template <class F>
struct IsAlPrimitive<F, std::void_t<decltype(F::kAlTermKind), decltype(std::declval<const F&>().AlCoefficients(1))>>
    : std::true_type {};

// A USER primitive whose device functor takes a parameter blob of its own (kTermParamsFromProblem: the same blob it takes
// as an objective — the kernel matrix of src/examples/svm_dual_al.cc:45-50) hands it over through AlUserParams().
template <class F, class = void>
struct HasAlUserParams : std::false_type {};
template <class F>
struct HasAlUserParams<F, std::void_t<decltype(std::declval<const F&>().AlUserParams())>> : std::true_type {};

// The one-line hook: `auto DeviceTwin() const` returning a TwinRecord.
template <class F, class = void>
struct HasDeviceTwinHook : std::false_type {};
template <class F>
struct HasDeviceTwinHook<F, std::void_t<decltype(std::declval<const F&>().DeviceTwin())>> : std::true_type {};

// ---- the run-time record ---------------------------------------------------------------------------------------------

// The function as an unconstrained OBJECTIVE of Lbfgs / Lbfgsb / Bfgs (mi355_lbfgs_desc).  Blobs are produced on demand
// by closures over one shared copy of the function, so copies of a record (and of the FunctionExpr holding it) are cheap.
struct TwinObjective {
  bool valid = false;
  int id = -1;        // mi355_objective, the reference's operation order
  int id_fused = -1;  // the fused / re-associated form taken under MI355_ARITH_FMA (-1: the same kernel)
  std::function<std::vector<double>(int n)> params;            // mi355_lbfgs_desc.objective_params
  std::function<std::vector<double>()> per_problem;            // one row of per_problem_data (null: none)
  std::function<std::vector<double>(int n)> hessian_diagonal;  // a CONSTANT Hessian's diagonal (null: none)
  bool hessian_from_functor = false;                           // diag H(x) from the device functor at every iterate
  std::function<uint64_t()> params_hash;                       // hash of everything params() returns (null: none)
  // own-parameters form (HasOwnMatrixForm)
  int id_own_matrix = -1;
  std::function<std::vector<double>()> own_params, own_row;
  std::function<std::array<double, 3>()> own_key;
  std::function<std::array<double, 6>()> fingerprint;
  std::function<double()> condition_bound;

  int Id(int arithmetic) const { return (arithmetic == MI355_ARITH_FMA && id_fused >= 0) ? id_fused : id; }
};

// kinds and coefficient-row builders of the primitives of a term, left to right
struct AlPrimitiveList {
  std::vector<int> kinds;
  std::vector<std::function<std::vector<double>(int)>> rows;
  std::vector<std::function<std::vector<double>()>> user_params;  // of the primitives that have one
};

// The function as a TERM of a constrained problem (mi355_al_problem): a primitive of the device menu, a left-to-right sum
// of primitives (the reference's AddExpression) or the product of two (ProdExpression), then `F`, `F - k` or `k - F`.
struct TwinTerm {
  bool valid = false;
  AlPrimitiveList prims;
  int form = MI355_AL_FORM_PLAIN;
  double k = 0;
  bool product = false;

  // `LinearForm(a)` or `LinearForm(a) - k`: an affine constraint a . x - k.  Constraint vectors longer than the term table
  // holds (MI355_AL_MAX_CONSTRAINTS per kind) travel as a FAMILY — a matrix of such rows (mi355_al_problem.family_*,
  // solver/augmented_lagrangian.h) — which is how src/examples/svm_primal_al.cc:139-147 with its 200 constraints runs.
  bool IsAffineRow() const {
    return valid && !product && prims.kinds.size() == 1 && prims.kinds[0] == MI355_AL_TERM_LINEAR &&
           (form == MI355_AL_FORM_PLAIN || form == MI355_AL_FORM_VALUE_MINUS_K);
  }
  // mi355_al_problem.parts of this term: the number of primitives summed, or MI355_AL_PARTS_PRODUCT
  int parts() const { return product ? MI355_AL_PARTS_PRODUCT : static_cast<int>(prims.kinds.size()); }
  int rows() const { return static_cast<int>(prims.kinds.size()); }
  const std::vector<int>& kinds() const { return prims.kinds; }
  double constant() const { return k; }
  // mi355_al_problem.user_params of this term's primitives (empty: none of them takes a blob)
  std::vector<std::vector<double>> UserParams() const {
    std::vector<std::vector<double>> all;
    for (const auto& blob : prims.user_params) all.push_back(blob());
    return all;
  }
  // the coefficient rows [parts][n + 1] of the C-ABI, concatenated; empty when a primitive was built for another
  // dimension
  std::vector<double> Coefficients(int n) const {
    std::vector<double> all;
    for (const auto& row : prims.rows) {
      const std::vector<double> r = row(n);
      if (static_cast<int>(r.size()) != n + 1) return {};
      all.insert(all.end(), r.begin(), r.end());
    }
    return all;
  }
};

// Least-squares data kept symbolically, so that `LeastSquares + lambda * SquaredNorm` — the README's ridge composition —
// resolves to the ridge kernel whatever the static types of the two operands are.
struct LeastSquaresShape {
  int rows = 0, n = 0;
  std::vector<double> a_row_major, y;
};

struct TwinRecord {
  TwinObjective objective;
  TwinTerm term;
  // why a facet is missing (shown by the solver that needed it)
  std::string why_no_objective = "the function type states no device twin (kDeviceObjective / DeviceParams, or DeviceTwin())";
  std::string why_no_term = "the function type is not a term of the device menu (kAlTermKind / AlCoefficients, or DeviceTwin())";
  // symbolic shapes the composition rules look at
  std::shared_ptr<const LeastSquaresShape> least_squares;  // ||A x - y||^2
  bool squared_norm = false;                               // squared_norm_scale * x.squaredNorm()
  double squared_norm_scale = 1.0;
  bool squared_norm_scaled = false;                        // written `c * SquaredNorm` (a MulExpression), not bare
};

// ---- records of function types that follow the static protocol -----------------------------------------------------
template <class F>
TwinObjective ObjectiveOfStatic(std::shared_ptr<const F> fn) {
  TwinObjective o;
  if constexpr (HasDeviceObjective<F>::value) {
    o.valid = true;
    o.id = F::kDeviceObjective;
    o.id_fused = FusedDeviceObjective<F>::Of(MI355_ARITH_FMA) != F::kDeviceObjective
                     ? FusedDeviceObjective<F>::Of(MI355_ARITH_FMA)
                     : -1;
    o.params = [fn](int n) {
      if constexpr (HasDeviceParamsOfDimension<F>::value) {
        return fn->DeviceParams(n);
      } else {
        (void)n;
        return fn->DeviceParams();
      }
    };
    if constexpr (HasPerProblemData<F>::value) o.per_problem = [fn]() { return fn->DevicePerProblem(); };
    o.hessian_from_functor = HessianFromFunctor<F>::value;
    if constexpr (HasDeviceHessianDiagonal<F>::value)
      o.hessian_diagonal = [fn](int) { return fn->DeviceHessianDiagonal(); };
    if constexpr (HasDeviceParamsHash<F>::value) o.params_hash = [fn]() { return fn->DeviceParamsHash(); };
    if constexpr (HasOwnMatrixForm<F>::value) {
      o.id_own_matrix = F::kDeviceObjectiveOwnMatrix;
      o.own_params = [fn]() { return fn->DeviceOwnMatrixParams(); };
      o.own_row = [fn]() { return fn->DeviceOwnMatrixRow(); };
      o.own_key = [fn]() { return fn->DeviceOwnMatrixKey(); };
      o.fingerprint = [fn]() { return fn->DeviceFingerprint(); };
      o.condition_bound = [fn]() { return fn->NormalEquationConditionBound(); };
    }
  }
  return o;
}

template <class P>
void AppendAlPrimitive(std::shared_ptr<const P> p, AlPrimitiveList* out) {
  out->kinds.push_back(P::kAlTermKind);
  out->rows.push_back([p](int n) { return p->AlCoefficients(n); });
  if constexpr (HasAlUserParams<P>::value) out->user_params.push_back([p]() { return p->AlUserParams(); });
}

// The record of a function object: its DeviceTwin() hook if it has one, else whatever the static protocol states.
template <class F>
TwinRecord RecordOfFunction(const F& f) {
  if constexpr (HasDeviceTwinHook<F>::value) {
    return f.DeviceTwin();
  } else {
    TwinRecord r;
    if constexpr (HasDeviceObjective<F>::value || IsAlPrimitive<F>::value) {
      auto fn = std::make_shared<const F>(f);
      r.objective = ObjectiveOfStatic<F>(fn);
      if constexpr (IsAlPrimitive<F>::value) {
        r.term.valid = true;
        AppendAlPrimitive<F>(fn, &r.term.prims);
      }
    }
    return r;
  }
}

// A composition that is a TERM of the device menu is also an OBJECTIVE: the augmented-Lagrangian composite
// (MI355_OBJ_AL_COMPOSITE) with that term as its objective and no constraints evaluates exactly the term — the rows summed
// (or multiplied) left to right as the reference's AddExpression / ProdExpression do, then `F`, `F - k` or `k - F`.  So
// `FunctionExpr f = DiagQuadratic(a, c) + LinearForm(b); Lbfgs<decltype(f)>` runs on the device like the ridge sum does,
// through the kernel the augmented-Lagrangian solver already uses (Lbfgs, First mode, n <= 256; no new kernel).
// Parameter blob: n_eq = 0, n_ineq = 0, rows, (parts, form, k), then per row (kind, coefficients [n + 1]); the per-problem
// row is the penalty alone (0: there is nothing to penalise).  Terms whose primitives take a user parameter blob keep to
// constrained problems (mi355_al_problem.user_params has no place in an objective's blob).
inline void ObjectiveFromTerm(TwinRecord* r) {
  if (r->objective.valid || !r->term.valid || !r->term.prims.user_params.empty()) return;
  const TwinTerm term = r->term;
  r->objective.valid = true;
  r->objective.id = MI355_OBJ_AL_COMPOSITE;
  r->objective.id_fused = -1;
  r->objective.params = [term](int n) {
    const std::vector<double> coef = term.Coefficients(n);
    if (static_cast<int>(coef.size()) != term.rows() * (n + 1)) Fail("a sum / product of device terms was built for another dimension");
    std::vector<double> p{0.0, 0.0, static_cast<double>(term.rows()), static_cast<double>(term.parts()),
                          static_cast<double>(term.form), term.constant()};
    for (int row = 0; row < term.rows(); ++row) {
      p.push_back(term.kinds()[static_cast<size_t>(row)]);
      p.insert(p.end(), coef.begin() + static_cast<std::ptrdiff_t>(row) * (n + 1),
               coef.begin() + static_cast<std::ptrdiff_t>(row + 1) * (n + 1));
    }
    return p;
  };
  r->objective.per_problem = []() { return std::vector<double>{0.0}; };
}

// ---- composition: what the reference's expression templates do to two functions, done to their records -------------

// `f - k` (kConstantFirst = false) and `k - f` (true): SubExpression with a ConstExpression operand
// (function_expressions.h:497-518).  Only a term keeps a twin; the device has no objective `f - k`.
inline TwinRecord OffsetRecord(const TwinRecord& f, double k, bool constant_first) {
  TwinRecord r;
  r.why_no_objective = "`f - k` / `k - f` has a device twin as a TERM of a constrained problem only";
  if (!f.term.valid) {
    r.why_no_term = f.why_no_term;
  } else if (f.term.form != MI355_AL_FORM_PLAIN) {
    r.why_no_term = "a term takes one constant: `(f - k1) - k2` has no device form (write `f - (k1 + k2)`)";
  } else {
    r.term = f.term;
    r.term.form = constant_first ? MI355_AL_FORM_K_MINUS_VALUE : MI355_AL_FORM_VALUE_MINUS_K;
    r.term.k = k;
  }
  ObjectiveFromTerm(&r);
  return r;
}

// `c * f`: MulExpression (function_expressions.h:200-254).  As a term, c = 1 and c = -1 are exact re-statements in the
// menu's forms: -(f) = 0 - f, -(f - k) = k - f, -(k - f) = f - k, all to the bit (IEEE negation and subtraction are
// symmetric; only the sign of an exact zero can differ).  Other factors have no device form as a term.
inline TwinRecord ScaledRecord(double c, const TwinRecord& f) {
  TwinRecord r;
  r.why_no_objective = "a scalar multiple has a device twin only as `lambda * SquaredNorm` inside a ridge sum, or as +-1 "
                       "times a term";
  if (f.squared_norm && !f.squared_norm_scaled) {
    r.squared_norm = true;
    r.squared_norm_scaled = true;
    r.squared_norm_scale = c;
  }
  if (!f.term.valid) {
    r.why_no_term = f.why_no_term;
  } else if (c == 1.0) {
    r.term = f.term;
  } else if (c == -1.0) {
    r.term = f.term;
    if (f.term.form == MI355_AL_FORM_PLAIN) {
      r.term.form = MI355_AL_FORM_K_MINUS_VALUE;
      r.term.k = 0.0;
    } else if (f.term.form == MI355_AL_FORM_VALUE_MINUS_K) {
      r.term.form = MI355_AL_FORM_K_MINUS_VALUE;
    } else {
      r.term.form = MI355_AL_FORM_VALUE_MINUS_K;
    }
  } else {
    r.why_no_term = "a term scaled by a factor other than 1 or -1 has no device form";
  }
  ObjectiveFromTerm(&r);
  return r;
}

// `f * g`: ProdExpression (function_expressions.h:260-315) of two primitives
inline TwinRecord ProductRecord(const TwinRecord& f, const TwinRecord& g) {
  TwinRecord r;
  r.why_no_objective = "a product of functions has a device twin as a TERM of a constrained problem only";
  auto single = [](const TwinTerm& t) { return t.valid && !t.product && t.form == MI355_AL_FORM_PLAIN && t.rows() == 1; };
  if (single(f.term) && single(g.term)) {
    r.term = f.term;
    r.term.product = true;
    r.term.prims.kinds.push_back(g.term.prims.kinds[0]);
    r.term.prims.rows.push_back(g.term.prims.rows[0]);
    for (const auto& blob : g.term.prims.user_params) r.term.prims.user_params.push_back(blob);
  } else {
    r.why_no_term = "only the product of two PRIMITIVES of the device menu is a term (MI355_AL_PARTS_PRODUCT)";
  }
  ObjectiveFromTerm(&r);
  return r;
}

// `f + g`: AddExpression (function_expressions.h:91-143).  As a term: a left-nested sum of primitives `(P1 + P2) + P3`,
// the order the device sums a term's primitives in.  As an objective: the ridge rule, see SumRecord in objectives.h.
inline TwinTerm SumTerm(const TwinRecord& f, const TwinRecord& g, std::string* why) {
  TwinTerm t;
  auto plain = [](const TwinTerm& s) { return s.valid && !s.product && s.form == MI355_AL_FORM_PLAIN; };
  if (plain(f.term) && plain(g.term) && g.term.rows() == 1) {
    t = f.term;
    t.prims.kinds.push_back(g.term.prims.kinds[0]);
    t.prims.rows.push_back(g.term.prims.rows[0]);
    for (const auto& blob : g.term.prims.user_params) t.prims.user_params.push_back(blob);
  } else {
    *why = "only a left-nested sum of PRIMITIVES of the device menu, `(P1 + P2) + P3`, is a term";
  }
  return t;
}

}  // namespace cppoptlib::mi355
#endif  // CPPOPTLIB_MI355_DEVICE_TWIN_H_
