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
#include <cstdint>
#include <tuple>
#include <utility>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "context.h"
#include "objectives.h"

namespace cppoptlib::mi355 {

constexpr int64_t kMaxReplayedIterations = 1 << 17;

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
  if constexpr (HasPerProblemData<FunctionType>::value) {
    const std::vector<double> row = function.DevicePerProblem();
    rows->reserve(row.size() * static_cast<size_t>(B));
    for (int64_t b = 0; b < B; ++b) rows->insert(rows->end(), row.begin(), row.end());
    return static_cast<int>(row.size());
  }
  return 0;
}
template <class FunctionType>
int PackPerProblem(const std::vector<FunctionType>& functions, std::vector<double>* rows) {
  rows->clear();
  if constexpr (HasPerProblemData<FunctionType>::value) {
    size_t stride = 0;
    for (size_t b = 0; b < functions.size(); ++b) {
      const std::vector<double> row = functions[b].DevicePerProblem();
      if (b == 0) stride = row.size();
      if (row.size() != stride) Fail("MinimizeBatch: functions with per-problem rows of different sizes");
      rows->insert(rows->end(), row.begin(), row.end());
    }
    return static_cast<int>(stride);
  }
  return 0;
}
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

// Do the functions of the batch share their device parameters?  O(B): the full-blob hash every such function type
// computes at construction (DeviceParamsHash), then one full comparison of the first and last blobs.
template <class FunctionType>
bool SharesDeviceParams(const std::vector<FunctionType>& functions) {
  if (functions.size() < 2) return true;
  if constexpr (HasDeviceParamsHash<FunctionType>::value) {
    const uint64_t first = functions[0].DeviceParamsHash();
    for (size_t b = 1; b < functions.size(); ++b)
      if (functions[b].DeviceParamsHash() != first) return false;
  } else {
    const auto first = functions[0].DeviceFingerprint();
    for (size_t b = 1; b < functions.size(); ++b)
      if (functions[b].DeviceFingerprint() != first) return false;
  }
  return functions.back().DeviceParams() == functions[0].DeviceParams();
}
// The own-matrix kernel takes (rows, lambda) from ONE shared blob: every function of such a batch must agree on them
// (a regularisation sweep — same matrix, different lambda — is not this form: solve it one lambda per batch).
template <class FunctionType>
void CheckOwnMatrixKey(const std::vector<FunctionType>& functions) {
  const auto key = functions[0].DeviceOwnMatrixKey();
  for (size_t b = 1; b < functions.size(); ++b)
    if (functions[b].DeviceOwnMatrixKey() != key)
      Fail("MinimizeBatch(functions, states): functions with their own matrices must agree on (rows, n, lambda) — the "
           "kernel reads them from one shared blob; solve batches that differ in lambda (a regularisation sweep) one "
           "lambda at a time");
}
template <class FunctionType>
int PackOwnMatrixRows(const std::vector<FunctionType>& functions, std::vector<double>* rows) {
  rows->clear();
  size_t stride = 0;
  for (size_t b = 0; b < functions.size(); ++b) {
    const std::vector<double> row = functions[b].DeviceOwnMatrixRow();
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
  auto params_of = [n](const FunctionType& fn) {
    if constexpr (HasDeviceParamsOfDimension<FunctionType>::value) {
      return fn.DeviceParams(n);
    } else {
      (void)n;
      return fn.DeviceParams();
    }
  };
  // EVERY function is compared with the first (round-4 advisor finding: a sampled check lets a batch through whose
  // functions differ away from the sample and solves it silently with functions[0]'s parameters):
  //  * types with a large blob carry a hash of it computed once at construction (DeviceParamsHash): B comparisons;
  //  * otherwise the blobs themselves are compared — they are small (Rosenbrock: empty; DiagQuadratic: n + 1 doubles),
  //    the same order of work as packing the start states.  A user function type with a LARGE blob and no hash pays
  //    O(B x blob) here; it should add `uint64_t DeviceParamsHash() const` (cppoptlib/mi355/objectives.h HashDoubles).
  const char* what =
      "MinimizeBatch(functions, states): the functions of a batch must share their device parameters "
      "(DeviceParams()); only their per-problem rows (DevicePerProblem()) may differ";
  if constexpr (HasDeviceParamsHash<FunctionType>::value) {
    const uint64_t first = functions[0].DeviceParamsHash();
    for (size_t b = 1; b < functions.size(); ++b)
      if (functions[b].DeviceParamsHash() != first) Fail(what);
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
