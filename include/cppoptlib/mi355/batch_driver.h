// cppoptlib/mi355/batch_driver.h — what the batched solver classes (Lbfgs, Lbfgsb, Bfgs) share on the host side:
// packing states into the batch-major arrays of the C-ABI, unpacking results, and the callback replay.
//
// Callback contract (reference: solver/solver.h:196-222).  The reference invokes step_callback_ before every
// OptimizationStep and once after the loop.  Here the loop runs inside one kernel, so a solve with a callback is run
// with the engine's per-iteration trace (mi355_lbfgs_trace: value, deltas, gradient norm, status, x and g after every
// iteration) and the callback is REPLAYED from it after the kernel returns: the same sequence of (state, progress)
// pairs in the same order — the evaluated start state with a fresh Progress, the state after every iteration with
// status Continue, the final state with the final status.  A solve without a callback records nothing and evaluates
// nothing on the host.  The trace is a ring of `capacity` iterations (the solver's iteration limit + 1, at most
// kMaxReplayedIterations): a longer solve replays its last `capacity` iterations.
#ifndef CPPOPTLIB_MI355_BATCH_DRIVER_H_
#define CPPOPTLIB_MI355_BATCH_DRIVER_H_

#include <algorithm>
#include <array>
#include <cstdint>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "context.h"
#include "objectives.h"

namespace cppoptlib::mi355 {

constexpr int64_t kMaxReplayedIterations = 1 << 17;

// ---- twin access: one spelling for function types that state their twin statically (members read at compile time) and
// for the type-erased FunctionExpr, whose twin is the run-time record it carries (cppoptlib/mi355/device_twin.h) --------
template <class F>
struct IsTypeErased : cppoptlib::function::IsFunctionExpr<F> {};
// Function types read through a run-time record: the erased wrapper (the record it stores) and static types that state
// their twin with the one-line DeviceTwin() hook only (the record the hook returns; README.md:21-28 `Lbfgs<Quadratic>`).
template <class F>
constexpr bool kUsesRecord = IsTypeErased<F>::value || (HasDeviceTwinHook<F>::value && !HasDeviceObjective<F>::value);
template <class F>
decltype(auto) Record(const F& f) {
  if constexpr (IsTypeErased<F>::value) {
    return (f.device_twin);
  } else {
    return f.DeviceTwin();
  }
}

// What a solver class template requires of its FunctionType at compile time.  For an erased type the check happens when a
// function reaches Minimize (RequireObjective): the static type no longer says what was assigned.
template <class F>
constexpr bool kHasDeviceTwin = HasDeviceObjective<F>::value || kUsesRecord<F>;

template <class F>
void RequireObjective(const F& f, const char* solver) {
  if constexpr (kUsesRecord<F>) {
    if constexpr (IsTypeErased<F>::value)
      if (!f.ptr) Fail(std::string(solver) + ": empty FunctionExpr");
    const auto& record = Record(f);
    if (!record.objective.valid)
      Fail(std::string(solver) + ": the function has no device twin as an objective — " + record.why_no_objective +
           "; the MI355X engine has no CPU fallback");
  } else {
    (void)f;
    (void)solver;
  }
}
// mi355_lbfgs_desc.objective under `arithmetic` (the fused twin under MI355_ARITH_FMA where the function names one)
template <class F>
int ObjectiveId(const F& f, int arithmetic) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.Id(arithmetic);
  } else {
    (void)f;
    return FusedDeviceObjective<F>::Of(arithmetic);
  }
}
// the reference-order twin, whatever the arithmetic (Lbfgsb, Bfgs, stand-alone line searches)
template <class F>
int PlainObjectiveId(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.id;
  } else {
    (void)f;
    return F::kDeviceObjective;
  }
}
// mi355_lbfgs_desc.objective_params (functions whose blob depends on the dimension, e.g. the augmented-Lagrangian
// composite of function_penalty.h, take n)
template <class F>
std::vector<double> ObjectiveParams(const F& f, int n) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.params(n);
  } else if constexpr (HasDeviceParamsOfDimension<F>::value) {
    return f.DeviceParams(n);
  } else {
    (void)n;
    return f.DeviceParams();
  }
}
// one row of per-problem data (empty: the objective has none)
template <class F>
std::vector<double> PerProblemRow(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.per_problem ? Record(f).objective.per_problem() : std::vector<double>();
  } else if constexpr (HasPerProblemData<F>::value) {
    return f.DevicePerProblem();
  } else {
    (void)f;
    return {};
  }
}
// does the type (static) or the wrapped function (erased) carry per-problem data at all?
template <class F>
bool CarriesPerProblemData(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return static_cast<bool>(Record(f).objective.per_problem);
  } else {
    (void)f;
    return HasPerProblemData<F>::value;
  }
}
template <class F>
bool UsesHessianFromFunctor(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.hessian_from_functor;
  } else {
    (void)f;
    return HessianFromFunctor<F>::value;
  }
}
// the diagonal of a CONSTANT Hessian: what the function states, else the diagonal of the host functor's Hessian at the
// origin (which is where the reference's preconditioner reads it too: lbfgs.h:116-139 evaluates H at the current x)
template <class F>
std::vector<double> ConstantHessianDiagonal(const F& f, int n) {
  std::vector<double> d;
  if constexpr (kUsesRecord<F>) {
    if (Record(f).objective.hessian_diagonal) d = Record(f).objective.hessian_diagonal(n);
  } else if constexpr (HasDeviceHessianDiagonal<F>::value) {
    d = f.DeviceHessianDiagonal();
  }
  if (d.empty()) {
    typename F::VectorType zero(n);
    for (int i = 0; i < n; ++i) zero[i] = 0;
    typename F::MatrixType hessian;
    f(zero, nullptr, &hessian);
    d.resize(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) d[static_cast<size_t>(i)] = static_cast<double>(hessian(i, i));
  }
  if (static_cast<int>(d.size()) != n) Fail("the function's Hessian diagonal has another dimension than the start state");
  return d;
}
template <class F>
bool CarriesParamsHash(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return static_cast<bool>(Record(f).objective.params_hash);
  } else {
    (void)f;
    return HasDeviceParamsHash<F>::value;
  }
}
template <class F>
uint64_t ParamsHash(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.params_hash();
  } else if constexpr (HasDeviceParamsHash<F>::value) {
    return f.DeviceParamsHash();
  } else {
    (void)f;
    return 0;
  }
}
// the own-parameters form (a different matrix per function of a batch)
template <class F>
constexpr bool kMayHaveOwnMatrixForm = HasOwnMatrixForm<F>::value || kUsesRecord<F>;
template <class F>
bool CarriesOwnMatrixForm(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.id_own_matrix >= 0;
  } else {
    (void)f;
    return HasOwnMatrixForm<F>::value;
  }
}
template <class F>
int OwnMatrixObjectiveId(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.id_own_matrix;
  } else {
    (void)f;
    return F::kDeviceObjectiveOwnMatrix;
  }
}
template <class F>
std::vector<double> OwnMatrixParams(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.own_params();
  } else {
    return f.DeviceOwnMatrixParams();
  }
}
template <class F>
std::vector<double> OwnMatrixRow(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.own_row();
  } else {
    return f.DeviceOwnMatrixRow();
  }
}
template <class F>
std::array<double, 3> OwnMatrixKey(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.own_key();
  } else {
    return f.DeviceOwnMatrixKey();
  }
}
template <class F>
std::array<double, 6> Fingerprint(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.fingerprint();
  } else {
    return f.DeviceFingerprint();
  }
}
template <class F>
double ConditionBound(const F& f) {
  if constexpr (kUsesRecord<F>) {
    return Record(f).objective.condition_bound();
  } else {
    return f.NormalEquationConditionBound();
  }
}

// states -> x0[B][n]
template <class StateType>
std::vector<double> PackStates(const std::vector<StateType>& states, int n) {
  std::vector<double> x0(states.size() * static_cast<size_t>(n));
  for (size_t b = 0; b < states.size(); ++b) {
    if (static_cast<int>(states[b].x.size()) != n) Fail("MinimizeBatch: mixed dimensions");
    for (int i = 0; i < n; ++i) x0[b * n + i] = states[b].x[i];
  }
  return x0;
}

// Per-problem data rows of the C-ABI (mi355_lbfgs_desc.per_problem_data) for function types that carry them
// (HasPerProblemData: e.g. the ridge objective's right-hand side y).  ONE function object describes one problem, so
// its row is replicated for the B start states; a vector of B function objects (the reference's README builds one
// SquaredError(A, y) per problem, README.md:126-167) gives one row each.  Returns the stride (0: no per-problem data).
template <class FunctionType>
int PackPerProblem(const FunctionType& function, int64_t B, std::vector<double>* rows) {
  rows->clear();
  if constexpr (HasPerProblemData<FunctionType>::value || kUsesRecord<FunctionType>) {
    const std::vector<double> row = PerProblemRow(function);
    rows->reserve(row.size() * static_cast<size_t>(B));
    for (int64_t b = 0; b < B; ++b) rows->insert(rows->end(), row.begin(), row.end());
    return static_cast<int>(row.size());
  }
  return 0;
}
template <class FunctionType>
int PackPerProblem(const std::vector<FunctionType>& functions, std::vector<double>* rows) {
  rows->clear();
  if constexpr (HasPerProblemData<FunctionType>::value || kUsesRecord<FunctionType>) {
    size_t stride = 0;
    for (size_t b = 0; b < functions.size(); ++b) {
      const std::vector<double> row = PerProblemRow(functions[b]);
      if (b == 0) stride = row.size();
      if (row.size() != stride) Fail("MinimizeBatch: functions with per-problem rows of different sizes");
      rows->insert(rows->end(), row.begin(), row.end());
    }
    return static_cast<int>(stride);
  }
  return 0;
}
// Do the functions of the batch share their device parameters?  O(B): the full-blob hash every such function type
// computes at construction (DeviceParamsHash), then one full comparison of the first and last blobs.
template <class FunctionType>
bool SharesDeviceParams(const std::vector<FunctionType>& functions) {
  if (functions.size() < 2) return true;
  if (CarriesParamsHash(functions[0])) {
    const uint64_t first = ParamsHash(functions[0]);
    for (size_t b = 1; b < functions.size(); ++b)
      if (!CarriesParamsHash(functions[b]) || ParamsHash(functions[b]) != first) return false;
  } else {
    const auto first = Fingerprint(functions[0]);
    for (size_t b = 1; b < functions.size(); ++b)
      if (Fingerprint(functions[b]) != first) return false;
  }
  return ObjectiveParams(functions.back(), 0) == ObjectiveParams(functions[0], 0);
}
// The own-matrix kernel takes (rows, lambda) from ONE shared blob: every function of such a batch must agree on them
// (a regularisation sweep — same matrix, different lambda — is not this form: solve it one lambda per batch).
template <class FunctionType>
void CheckOwnMatrixKey(const std::vector<FunctionType>& functions) {
  const auto key = OwnMatrixKey(functions[0]);
  for (size_t b = 1; b < functions.size(); ++b)
    if (!CarriesOwnMatrixForm(functions[b]) || OwnMatrixKey(functions[b]) != key)
      Fail("MinimizeBatch(functions, states): functions with their own matrices must agree on (rows, n, lambda) — the "
           "kernel reads them from one shared blob; solve batches that differ in lambda (a regularisation sweep) one "
           "lambda at a time");
}
template <class FunctionType>
int PackOwnMatrixRows(const std::vector<FunctionType>& functions, std::vector<double>* rows) {
  rows->clear();
  size_t stride = 0;
  for (size_t b = 0; b < functions.size(); ++b) {
    const std::vector<double> row = OwnMatrixRow(functions[b]);
    if (b == 0) {
      stride = row.size();
      rows->reserve(stride * functions.size());
    }
    if (row.size() != stride) Fail("MinimizeBatch: functions of different shapes in one batch");
    rows->insert(rows->end(), row.begin(), row.end());
  }
  return static_cast<int>(stride);
}

// The functions of a batch share one device parameter blob (for the ridge objective: the matrix A and lambda); what
// differs between them is their per-problem row.
template <class FunctionType>
void CheckSharedParams(const std::vector<FunctionType>& functions, int n) {
  if (functions.empty()) return;
  auto params_of = [n](const FunctionType& fn) { return ObjectiveParams(fn, n); };
  // EVERY function is compared with the first (round-4 advisor finding: a sampled check lets a batch through whose
  // functions differ away from the sample and solves it silently with functions[0]'s parameters):
  //  * types with a large blob carry a hash of it computed once at construction (DeviceParamsHash): B comparisons;
  //  * otherwise the blobs themselves are compared — they are small (Rosenbrock: empty; DiagQuadratic: n + 1 doubles),
  //    the same order of work as packing the start states.  A user function type with a LARGE blob and no hash pays
  //    O(B x blob) here; it should add `uint64_t DeviceParamsHash() const` (cppoptlib/mi355/objectives.h HashDoubles).
  const char* what =
      "MinimizeBatch(functions, states): the functions of a batch must share their device parameters "
      "(DeviceParams()); only their per-problem rows (DevicePerProblem()) may differ";
  if constexpr (kUsesRecord<FunctionType>) {  // the functions of an erased batch may wrap different kernels
    const int id = PlainObjectiveId(functions[0]);
    for (size_t b = 1; b < functions.size(); ++b) {
      RequireObjective(functions[b], "MinimizeBatch(functions, states)");
      if (PlainObjectiveId(functions[b]) != id)
        Fail("MinimizeBatch(functions, states): the functions of a batch must wrap the same device objective");
    }
  }
  if (CarriesParamsHash(functions[0])) {
    const uint64_t first = ParamsHash(functions[0]);
    for (size_t b = 1; b < functions.size(); ++b)
      if (!CarriesParamsHash(functions[b]) || ParamsHash(functions[b]) != first) Fail(what);
    if (params_of(functions.back()) != params_of(functions[0])) Fail(what);
  } else {
    const std::vector<double> first = params_of(functions[0]);
    for (size_t b = 1; b < functions.size(); ++b)
      if (params_of(functions[b]) != first) Fail(what);
  }
}

// x[B][n], f[B], g[B][n], progress[B] -> (state, progress) tuples
template <class StateType, class ProgressType, class VectorType>
std::vector<std::tuple<StateType, ProgressType>> UnpackResults(int n, int64_t B, const std::vector<double>& x,
                                                               const std::vector<double>& f, const std::vector<double>& g,
                                                               const std::vector<mi355_lbfgs_progress>& prog) {
  std::vector<std::tuple<StateType, ProgressType>> result;
  result.reserve(static_cast<size_t>(B));
  for (int64_t b = 0; b < B; ++b) {
    VectorType xv(n), gv(n);
    for (int i = 0; i < n; ++i) {
      xv[i] = x[static_cast<size_t>(b) * n + i];
      gv[i] = g[static_cast<size_t>(b) * n + i];
    }
    result.emplace_back(StateType(std::move(xv), f[static_cast<size_t>(b)], std::move(gv)),
                        ProgressType::FromDevice(prog[static_cast<size_t>(b)]));
  }
  return result;
}

// Progress::condition_hessian of a Second-mode function under a solver whose kernel has no use for the Hessian (Bfgs,
// Lbfgsb).  The reference's Progress::Update (solver/progress.h:203-210) recomputes ||H(x)|| ||H(x)^-1|| at every iterate
// for ANY solver when the function is Second mode, so `Bfgs<FunctionExprXd2>` of src/examples/simple.cc:56-57 prints it
// from its callback and returns it.  Here it comes from the HOST functor's Hessian at the shown point (an O(n^3)
// factorisation per record, as in the reference); the STOPPING test on it is built for Lbfgs only and refused elsewhere.
template <class FunctionType, class VectorType>
double HostHessianCondition(const FunctionType& function, const VectorType& x) {
  const int n = static_cast<int>(x.size());
  typename FunctionType::MatrixType hessian;
  function(x, nullptr, &hessian);
  std::vector<double> h(static_cast<size_t>(n) * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) h[static_cast<size_t>(i) * n + j] = static_cast<double>(hessian(i, j));
  double condition = 0;
  Check(mi355_lbfgs_hessian_condition(h.data(), n, &condition), "mi355_lbfgs_hessian_condition");
  return condition;
}

// MinimizeOne for Bfgs / Lbfgsb: with a Second-mode function the replayed records (all but the fresh Progress of the start
// state) and the returned Progress carry condition_hessian; `condition_stop` > 0 is refused.
template <class StateType, class ProgressType, class VectorType, class FunctionType, class Callback, class Run>
std::tuple<StateType, ProgressType> MinimizeOneReportingCondition(const char* solver, const FunctionType& function,
                                                                  const StateType& start, bool has_callback,
                                                                  const Callback& callback, uint64_t iteration_limit,
                                                                  double condition_stop, Run&& run);

// One problem with the reference's callback semantics.  `run(n, B, x0, x, f, g, progress, trace)` performs the solve
// through the solver's host-pointer entry point (trace may be null).
template <class StateType, class ProgressType, class VectorType, class FunctionType, class Callback, class Run>
std::tuple<StateType, ProgressType> MinimizeOne(const FunctionType& function, const StateType& start, bool has_callback,
                                                const Callback& callback, uint64_t iteration_limit, Run&& run) {
  const int n = static_cast<int>(start.x.size());
  std::vector<double> x0(static_cast<size_t>(n)), x(x0.size()), g(x0.size()), f(1);
  for (int i = 0; i < n; ++i) x0[static_cast<size_t>(i)] = start.x[i];
  std::vector<mi355_lbfgs_progress> prog(1);
  if (!has_callback) {
    run(n, int64_t{1}, x0.data(), x.data(), f.data(), g.data(), prog.data(), static_cast<const mi355_lbfgs_trace*>(nullptr));
    return UnpackResults<StateType, ProgressType, VectorType>(n, 1, x, f, g, prog)[0];
  }
  const int64_t capacity =
      (iteration_limit > 0 && static_cast<int64_t>(iteration_limit) + 1 < kMaxReplayedIterations)
          ? static_cast<int64_t>(iteration_limit) + 1
          : kMaxReplayedIterations;
  std::vector<mi355_lbfgs_trace_record> records(static_cast<size_t>(capacity));
  std::vector<double> tx(static_cast<size_t>(capacity) * n), tg(tx.size());
  uint32_t written = 0;
  const int64_t problem0 = 0;
  mi355_lbfgs_trace trace;
  trace.count = 1;
  trace.capacity = static_cast<int32_t>(capacity);
  trace.problems = &problem0;
  trace.records = records.data();
  trace.x = tx.data();
  trace.g = tg.data();
  trace.written = &written;
  run(n, int64_t{1}, x0.data(), x.data(), f.data(), g.data(), prog.data(), &trace);
  // replay: solver.h:189-192 evaluates the start point on the host, :197 shows it with a fresh Progress
  callback(function, StateType(function, start.x), ProgressType());
  const int64_t kept = std::min<int64_t>(written, capacity);
  for (int64_t t = static_cast<int64_t>(written) - kept + 1; t <= static_cast<int64_t>(written); ++t) {
    const size_t r = static_cast<size_t>((t - 1) % capacity);
    VectorType xv(n), gv(n);
    for (int i = 0; i < n; ++i) {
      xv[i] = tx[r * n + i];
      gv[i] = tg[r * n + i];
    }
    mi355_lbfgs_progress p{};
    p.status = records[r].status;
    p.num_iterations = records[r].num_iterations;
    p.x_delta = records[r].x_delta;
    p.f_delta = records[r].f_delta;
    p.gradient_norm = records[r].gradient_norm;
    if (t == static_cast<int64_t>(written)) {  // the call after the loop (:222) carries the final accounting
      p.nfev = prog[0].nfev;
      p.sum_k = prog[0].sum_k;
    }
    callback(function, StateType(std::move(xv), records[r].value, std::move(gv)), ProgressType::FromDevice(p));
  }
  return UnpackResults<StateType, ProgressType, VectorType>(n, 1, x, f, g, prog)[0];
}

template <class StateType, class ProgressType, class VectorType, class FunctionType, class Callback, class Run>
std::tuple<StateType, ProgressType> MinimizeOneReportingCondition(const char* solver, const FunctionType& function,
                                                                  const StateType& start, bool has_callback,
                                                                  const Callback& callback, uint64_t iteration_limit,
                                                                  double condition_stop, Run&& run) {
  if constexpr (FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second) {
    if (condition_stop > 0)
      Fail(std::string(solver) + ": the condition_hessian stopping test (progress.h:318-325) is built for Lbfgs on the device; "
           "this solver reports the quantity but cannot stop on it (no CPU fallback)");
    using Scalar = typename FunctionType::ScalarType;
    auto replay = [&callback](const FunctionType& fn, const StateType& state, const ProgressType& progress) {
      ProgressType shown = progress;
      if (progress.num_iterations > 0) shown.condition_hessian = static_cast<Scalar>(HostHessianCondition(fn, state.x));
      callback(fn, state, shown);
    };
    auto out = MinimizeOne<StateType, ProgressType, VectorType>(function, start, has_callback, replay, iteration_limit,
                                                                 std::forward<Run>(run));
    if (std::get<1>(out).num_iterations > 0)
      std::get<1>(out).condition_hessian = static_cast<Scalar>(HostHessianCondition(function, std::get<0>(out).x));
    return out;
  } else {
    (void)solver;
    (void)condition_stop;
    return MinimizeOne<StateType, ProgressType, VectorType>(function, start, has_callback, callback, iteration_limit,
                                                            std::forward<Run>(run));
  }
}

// RAII owner of a device group (mi355_lbfgs_group): one engine context per listed device + an RCCL communicator.
class DeviceGroup {
 public:
  explicit DeviceGroup(const std::vector<int>& devices) {
    Check(mi355_lbfgs_group_create(devices.data(), static_cast<int>(devices.size()), &group_), "mi355_lbfgs_group_create");
  }
  ~DeviceGroup() { mi355_lbfgs_group_destroy(group_); }
  DeviceGroup(const DeviceGroup&) = delete;
  DeviceGroup& operator=(const DeviceGroup&) = delete;
  mi355_lbfgs_group* get() const { return group_; }
  int size() const { return mi355_lbfgs_group_size(group_); }

 private:
  mi355_lbfgs_group* group_ = nullptr;
};

// The all-reduced convergence record of a sharded solve.
struct GlobalFlag {
  uint64_t total = 0;        // problems solved by all members
  uint64_t unconverged = 0;  // stopped on the iteration limit (or never started)
  uint64_t iterations = 0;   // sum of outer iterations
  bool all_converged() const { return unconverged == 0; }
};

}  // namespace cppoptlib::mi355
#endif  // CPPOPTLIB_MI355_BATCH_DRIVER_H_
