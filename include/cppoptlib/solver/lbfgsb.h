// cppoptlib/solver/lbfgsb.h — box-constrained L-BFGS-B on the MI355X, cppoptlib-shaped.
//
// Drop-in for the reference's solver/lbfgsb.h: `Lbfgsb<FunctionType, m = 5, LineSearch>` (:44-49),
// `SetBounds(lower, upper)` (:89-93), `Minimize` (:247-292; stops on the PROJECTED gradient norm),
// a default constructor that adds the relative f-delta test (:84-87) and the inherited constructor
// taking an explicit `Progress` (:74).  The solve runs in `lbfgsb_solve_kernel` behind
// `mi355_lbfgsb_minimize_batch_host` (include/mi355_lbfgs.h).  New: MinimizeBatch.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_LBFGSB_H_
#define INCLUDE_CPPOPTLIB_SOLVER_LBFGSB_H_

#include <memory>
#include <tuple>
#include <utility>
#include <vector>

#include "../../mi355_lbfgs.h"
#include "../linesearch/more_thuente.h"
#include "../mi355/batch_driver.h"
#include "../mi355/context.h"
#include "solver.h"

namespace cppoptlib::solver {

template <typename FunctionType, int m = 5, template <class, int> class LineSearch = linesearch::MoreThuente>
class Lbfgsb : public Solver<FunctionType, cppoptlib::function::FunctionState<typename FunctionType::ScalarType,
                                                                              FunctionType::Dimension>> {
  static_assert(FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::First ||
                    FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second,
                "L-BFGS-B only supports first- or second-order differentiable functions");
  static_assert(std::is_floating_point<typename FunctionType::ScalarType>::value,
                "ScalarType must be float or double (the MI355X engine computes in fp64 either way: a float function type is "
                "widened at the boundary and its results are rounded back, see INTEGRATION.md)");
  static_assert(cppoptlib::mi355::kHasDeviceTwin<FunctionType>,
                "FunctionType has no device twin (see cppoptlib/mi355/objectives.h); no CPU fallback");
  static_assert(m >= 1 && m <= 10, "the device L-BFGS-B kernel is built for m <= 10 (5 is the reference default)");

 public:
  using StateType =
      cppoptlib::function::FunctionState<typename FunctionType::ScalarType, FunctionType::Dimension>;
  using Superclass = Solver<FunctionType, StateType>;
  using ProgressType = typename Superclass::ProgressType;
  using ScalarType = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  using MatrixType = typename FunctionType::MatrixType;

  using Superclass::Superclass;
  Lbfgsb() : Superclass() {  // lbfgsb.h:84-87
    this->stopping_progress.f_delta = ScalarType(2.22e-9);
    this->stopping_progress.f_delta_relative = true;
  }

  static constexpr int kHistorySize = m;
  static constexpr int kLineSearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;
  // the box as host arrays (empty until SetBounds): what an outer solver hands to the C-ABI together with this
  // solver's stopping_progress (solver/augmented_lagrangian.h)
  const std::vector<double>& LowerBound() const { return lower_; }
  const std::vector<double>& UpperBound() const { return upper_; }

  // Sup-norm of the box-projected gradient (reference lbfgsb.h:105-118): |g_j| with the coordinates dropped where x_j sits on
  // an active bound and g_j points out of the box — the convergence measure of a box-constrained solve, and what an outer
  // augmented-Lagrangian loop reads for KKT stationarity at a box-constrained inner optimum.  Without SetBounds the box is
  // unbounded and this is the plain sup-norm of the gradient.  A host-side helper: the device solve computes the same
  // quantity for its own stopping test.
  ScalarType ProjectedGradientInfNorm(const VectorType& x, const VectorType& gradient) const {
    ScalarType norm = ScalarType(0);
    const bool boxed = !lower_.empty();
    for (std::ptrdiff_t j = 0; j < static_cast<std::ptrdiff_t>(x.size()); ++j) {
      ScalarType gj = gradient[j];
      if (boxed) {
        if (x[j] <= static_cast<ScalarType>(lower_[static_cast<size_t>(j)]) && gj > 0) gj = ScalarType(0);
        if (x[j] >= static_cast<ScalarType>(upper_[static_cast<size_t>(j)]) && gj < 0) gj = ScalarType(0);
      }
      const ScalarType a = gj < 0 ? -gj : gj;
      if (norm < a) norm = a;     // std::max(norm, |g_j|): a NaN entry leaves the running maximum as it is
    }
    return norm;
  }

  void SetBounds(const VectorType& lower_bound, const VectorType& upper_bound) {
    lower_.assign(static_cast<size_t>(lower_bound.size()), 0.0);
    upper_.assign(static_cast<size_t>(upper_bound.size()), 0.0);
    for (size_t i = 0; i < lower_.size(); ++i) {
      lower_[i] = lower_bound[static_cast<std::ptrdiff_t>(i)];
      upper_[i] = upper_bound[static_cast<std::ptrdiff_t>(i)];
      // (the order of NaN breakpoints in the Cauchy search is undefined in the reference too; the C-ABI refuses them)
      if (lower_[i] != lower_[i] || upper_[i] != upper_[i]) cppoptlib::mi355::Fail("SetBounds: NaN bound");
    }
  }
  void SetContext(std::shared_ptr<cppoptlib::mi355::Context> ctx) { ctx_ = std::move(ctx); }
  // mi355_arithmetic: MI355_ARITH_DEFAULT (the relaxed-algebra kernels where they are built), MI355_ARITH_EXACT (the
  // reference's operation order, the bit-pinning build) or MI355_ARITH_FMA.
  void SetArithmetic(int arithmetic) { arithmetic_ = arithmetic; }

  // With a callback set the solve is traced on the device and the callback replayed afterwards
  // (cppoptlib/mi355/batch_driver.h); without one nothing is evaluated on the host.
  std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                               const StateType& function_state) override {
    // (a Second-mode function: the reference's Progress::Update computes condition_hessian under every solver —
    //  reported here from the host functor's Hessian, batch_driver.h)
    return cppoptlib::mi355::MinimizeOneReportingCondition<StateType, ProgressType, VectorType>(
        "Lbfgsb", function, function_state, this->HasCallback(), this->step_callback_,
        static_cast<uint64_t>(this->stopping_progress.num_iterations),
        static_cast<double>(this->stopping_progress.condition_hessian),
        [&](int n, int64_t B, const double* x0, double* x, double* f, double* g, mi355_lbfgs_progress* prog,
            const mi355_lbfgs_trace* trace) { MinimizeBatchRaw(function, n, B, x0, x, f, g, prog, trace); });
  }

  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const FunctionType& function,
                                                                 const std::vector<StateType>& states) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    MinimizeBatchRaw(function, n, B, x0.data(), x.data(), f.data(), g.data(), prog.data());
    return cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
  }

  // One function object per problem (as Lbfgs::MinimizeBatch(functions, states)): shared device parameters, one
  // per-problem row each — e.g. B regression problems with their own right-hand sides in one box
  // (src/examples/linear_regression.cc runs Lbfgsb on such a function).
  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const std::vector<FunctionType>& functions,
                                                                 const std::vector<StateType>& states) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (functions.size() != states.size()) cppoptlib::mi355::Fail("MinimizeBatch: one function per start state");
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    for (const FunctionType& fn : functions) cppoptlib::mi355::RequireObjective(fn, "Lbfgsb::MinimizeBatch");
    cppoptlib::mi355::CheckSharedParams(functions, n);
    if (!lower_.empty() && static_cast<int>(lower_.size()) != n) cppoptlib::mi355::Fail("SetBounds: dimension mismatch");
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    std::vector<double> params, rows;
    mi355_lbfgs_desc d = Desc(functions[0], n, &params, nullptr);
    d.per_problem_stride = cppoptlib::mi355::PackPerProblem(functions, &rows);
    d.per_problem_data = rows.empty() ? nullptr : rows.data();
    cppoptlib::mi355::Check(
        mi355_lbfgsb_minimize_batch_host(ctx_->get(), &d, lower_.empty() ? nullptr : lower_.data(),
                                         upper_.empty() ? nullptr : upper_.data(), B, x0.data(), x.data(), f.data(),
                                         g.data(), prog.data()),
        "mi355_lbfgsb_minimize_batch_host");
    return cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
  }

  // The batch over a device group (mi355_lbfgsb_group_minimize_batch_host): contiguous shards, one per member, each
  // solved on its own GPU by its own host thread; `flag` receives the RCCL all-reduced convergence record.
  std::vector<std::tuple<StateType, ProgressType>> ShardedMinimizeBatch(const FunctionType& function,
                                                                        const std::vector<StateType>& states,
                                                                        cppoptlib::mi355::DeviceGroup& group,
                                                                        cppoptlib::mi355::GlobalFlag* flag = nullptr) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    if (!lower_.empty() && static_cast<int>(lower_.size()) != n) cppoptlib::mi355::Fail("SetBounds: dimension mismatch");
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    std::vector<double> params, rows;
    mi355_lbfgs_desc d = Desc(function, n, &params, nullptr);
    d.per_problem_stride = cppoptlib::mi355::PackPerProblem(function, B, &rows);
    d.per_problem_data = rows.empty() ? nullptr : rows.data();
    uint64_t record[3] = {0, 0, 0};
    cppoptlib::mi355::Check(
        mi355_lbfgsb_group_minimize_batch_host(group.get(), &d, lower_.empty() ? nullptr : lower_.data(),
                                               upper_.empty() ? nullptr : upper_.data(), B, x0.data(), x.data(),
                                               f.data(), g.data(), prog.data(), record),
        "mi355_lbfgsb_group_minimize_batch_host");
    if (flag) {
      flag->total = record[0];
      flag->unconverged = record[1];
      flag->iterations = record[2];
    }
    return cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
  }

  // Batch-major HOST arrays in and out (mi355_lbfgsb_minimize_batch_host).
  void MinimizeBatchRaw(const FunctionType& function, int n, int64_t B, const double* x0, double* x, double* f,
                        double* g, mi355_lbfgs_progress* progress, const mi355_lbfgs_trace* trace = nullptr) {
    if (!lower_.empty() && static_cast<int>(lower_.size()) != n) cppoptlib::mi355::Fail("SetBounds: dimension mismatch");
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    std::vector<double> params, rows;
    mi355_lbfgs_desc d = Desc(function, n, &params, trace);
    // (a function with a per-problem row — the ridge / regression objective — describes ONE problem: its row is
    //  replicated for the B start states)
    d.per_problem_stride = cppoptlib::mi355::PackPerProblem(function, B, &rows);
    d.per_problem_data = rows.empty() ? nullptr : rows.data();
    cppoptlib::mi355::Check(
        mi355_lbfgsb_minimize_batch_host(ctx_->get(), &d, lower_.empty() ? nullptr : lower_.data(),
                                         upper_.empty() ? nullptr : upper_.data(), B, x0, x, f, g, progress),
        "mi355_lbfgsb_minimize_batch_host");
  }

  // Device-resident batch (every array pointer is DEVICE memory on the context's device, bounds included — n doubles
  // each, or both null for the default box); asynchronous on `stream` (a hipStream_t, null = default stream).
  void MinimizeBatchDevice(const FunctionType& function, int n, int64_t B, const double* lower_dev,
                           const double* upper_dev, const double* x0_dev, double* x_dev, double* f_dev, double* g_dev,
                           mi355_lbfgs_progress* progress_dev, void* stream = nullptr,
                           const double* per_problem_dev = nullptr, int per_problem_stride = 0) {
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    std::vector<double> params;
    mi355_lbfgs_desc d = Desc(function, n, &params, nullptr);
    d.per_problem_data = per_problem_dev;
    d.per_problem_stride = per_problem_stride;
    cppoptlib::mi355::Check(mi355_lbfgsb_minimize_batch(ctx_->get(), &d, lower_dev, upper_dev, B, x0_dev, x_dev, f_dev,
                                                        g_dev, progress_dev, stream),
                            "mi355_lbfgsb_minimize_batch");
  }

 private:
  mi355_lbfgs_desc Desc(const FunctionType& function, int n, std::vector<double>* params,
                        const mi355_lbfgs_trace* trace) const {
    cppoptlib::mi355::RequireObjective(function, "Lbfgsb");
    *params = cppoptlib::mi355::ObjectiveParams(function, n);
    mi355_lbfgs_desc d{};
    d.objective = cppoptlib::mi355::PlainObjectiveId(function);
    d.linesearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;
    d.n = n;
    d.m = m;
    d.objective_params = params->empty() ? nullptr : params->data();
    d.n_params = static_cast<int32_t>(params->size());
    d.history_placement = MI355_HISTORY_AUTO;
    d.arithmetic = arithmetic_;
    d.hessian_diagonal = nullptr;  // lbfgsb.h:48-49 of the reference: second-order information is never used
    d.trace = trace;
    d.stop = this->stopping_progress.ToDeviceStop();
    return d;
  }

  std::vector<double> lower_, upper_;
  int arithmetic_ = MI355_ARITH_DEFAULT;
  std::shared_ptr<cppoptlib::mi355::Context> ctx_;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_LBFGSB_H_
