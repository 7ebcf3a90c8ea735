// cppoptlib/solver/lbfgs.h — L-BFGS on the MI355X, cppoptlib-shaped.
//
// Drop-in for the reference's solver/lbfgs.h: same class template
//   Lbfgs<FunctionType, m = 10, LineSearch = linesearch::MoreThuente>   (lbfgs.h:40-45)
// same StateType / ProgressType aliases, same Minimize signature and return
// type.  The solve itself — two-loop recursion with the (s, y) ring
// (lbfgs.h:145-196, :248-298), Moré–Thuente line search, stopping tests — runs
// in one HIP kernel behind the C-ABI of include/mi355_lbfgs.h.
//
// New here: MinimizeBatch — the reference is one-problem-at-a-time; a GPU wants
// thousands of independent problems per launch.
#ifndef INCLUDE_CPPOPTLIB_SOLVER_LBFGS_H_
#define INCLUDE_CPPOPTLIB_SOLVER_LBFGS_H_

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

template <typename FunctionType, int m = 10, template <class, int> class LineSearch = linesearch::MoreThuente>
class Lbfgs : public Solver<FunctionType, cppoptlib::function::FunctionState<typename FunctionType::ScalarType,
                                                                             FunctionType::Dimension>> {
  static_assert(FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::First ||
                    FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second,
                "L-BFGS only supports first- or second-order differentiable functions");
  static_assert(std::is_floating_point<typename FunctionType::ScalarType>::value,
                "ScalarType must be float or double (the MI355X engine computes in fp64 either way: a float function type is "
                "widened at the boundary and its results are rounded back, see INTEGRATION.md)");
  static_assert(cppoptlib::mi355::kHasDeviceTwin<FunctionType>,
                "FunctionType has no device twin (kDeviceObjective / DeviceParams, see "
                "cppoptlib/mi355/objectives.h); the MI355X engine has no CPU fallback");
  static_assert(m >= 1 && m <= MI355_LBFGS_MAX_M, "history size m out of range");

 public:
  using StateType =
      cppoptlib::function::FunctionState<typename FunctionType::ScalarType, FunctionType::Dimension>;
  using Superclass = Solver<FunctionType, StateType>;
  using ProgressType = typename Superclass::ProgressType;
  using ScalarType = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  using MatrixType = typename FunctionType::MatrixType;

  static constexpr int kHistorySize = m;
  static constexpr int kLineSearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;

  using Superclass::Superclass;

  // Engine context (device 0 by default); set before the first Minimize to pick a GPU.
  void SetContext(std::shared_ptr<cppoptlib::mi355::Context> ctx) { ctx_ = std::move(ctx); }
  // mi355_arithmetic: MI355_ARITH_DEFAULT (the fused production kernels where they exist), MI355_ARITH_EXACT (the
  // bit-pinning build) or MI355_ARITH_FMA.
  void SetArithmetic(int arithmetic) { arithmetic_ = arithmetic; }

  // Lbfgs::Minimize of the reference (solver/solver.h:181-224): only `.x` of the incoming state is used, value and
  // gradient are evaluated at x0 (:189-192).  With a callback set the solve is traced on the device and the callback
  // replayed (cppoptlib/mi355/batch_driver.h); without one nothing is evaluated on the host.
  std::tuple<StateType, ProgressType> Minimize(const FunctionType& function,
                                               const StateType& function_state) override {
    cppoptlib::mi355::RequireObjective(function, "Lbfgs");
    // (Second mode: every Update of the reference recomputes Progress::condition_hessian, progress.h:203-210, so the
    //  replayed records carry it too — the constant of this solve, or ||H(x)|| ||H(x)^-1|| of the host functor at the
    //  replayed point; the record of the start state is a fresh Progress, as there)
    auto replay = [this](const FunctionType& fn, const StateType& state, const ProgressType& progress) {
      if constexpr (FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second) {
        ProgressType shown = progress;
        if (progress.num_iterations > 0) {
          double condition = hessian_condition_;
          if (cppoptlib::mi355::UsesHessianFromFunctor(fn)) {
            const int n = static_cast<int>(state.x.size());
            cppoptlib::mi355::Check(mi355_lbfgs_hessian_condition(HostHessian(fn, state.x, n).data(), n, &condition),
                                    "mi355_lbfgs_hessian_condition");
          }
          shown.condition_hessian = static_cast<ScalarType>(condition);
        }
        this->step_callback_(fn, state, shown);
      } else {
        this->step_callback_(fn, state, progress);
      }
    };
    auto out = cppoptlib::mi355::MinimizeOne<StateType, ProgressType, VectorType>(
        function, function_state, this->HasCallback(), replay,
        static_cast<uint64_t>(this->stopping_progress.num_iterations),
        [&](int n, int64_t B, const double* x0, double* x, double* f, double* g, mi355_lbfgs_progress* prog,
            const mi355_lbfgs_trace* trace) { MinimizeBatchRaw(function, n, B, x0, x, f, g, prog, trace); });
    ReportHessianCondition(function, &out, /*single_problem=*/true);
    return out;
  }

  // Solves every start state independently in one kernel launch.
  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const FunctionType& function,
                                                                 const std::vector<StateType>& states) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    MinimizeBatchRaw(function, n, B, x0.data(), x.data(), f.data(), g.data(), prog.data());
    auto out = cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
    for (auto& r : out) ReportHessianCondition(function, &r, /*single_problem=*/false);
    return out;
  }

  // One function object PER problem (the reference's README builds `SquaredError(A, y)` once per right-hand side,
  // README.md:126-167): functions[b] is minimised from states[b].  The functions share their device parameters (for the
  // ridge objective the matrix and lambda) and differ in their per-problem rows, which are packed one per problem
  // instead of replicating a single row as MinimizeBatch(function, states) does.
  std::vector<std::tuple<StateType, ProgressType>> MinimizeBatch(const std::vector<FunctionType>& functions,
                                                                 const std::vector<StateType>& states) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (functions.size() != states.size()) cppoptlib::mi355::Fail("MinimizeBatch: one function per start state");
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    for (const FunctionType& fn : functions) cppoptlib::mi355::RequireObjective(fn, "Lbfgs::MinimizeBatch");
    bool own_matrices = false;
    if constexpr (cppoptlib::mi355::kMayHaveOwnMatrixForm<FunctionType> &&
                  FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::First) {
      own_matrices = cppoptlib::mi355::CarriesOwnMatrixForm(functions[0]) &&
                     !cppoptlib::mi355::SharesDeviceParams(functions);
      if (own_matrices) {   // (all refusals before anything touches the device)
        // This is the ONLY device form of such a batch, so the switch to it is unconditional under MI355_ARITH_DEFAULT
        // / MI355_ARITH_FMA, and:
        //  * SetArithmetic(MI355_ARITH_EXACT) is refused here (the C-ABI refuses it too) instead of being dropped;
        //  * (rows, n, lambda) must agree over the batch (the kernel reads them from one blob);
        //  * the form is pinned to 1e-6 of the reference while cond(A^T A + lambda I) <~ 3e2; under
        //    MI355_ARITH_DEFAULT a sample of the batch (first, last, every ceil(B/16)-th function) is held against
        //    that bound (Gershgorin, rigorous) and the batch refused when the sample exceeds it — say
        //    SetArithmetic(MI355_ARITH_FMA) to take the form regardless.
        if (arithmetic_ == MI355_ARITH_EXACT)
          cppoptlib::mi355::Fail("MinimizeBatch(functions, states): functions with their own matrices run the normal-equation "
                                 "form (fused arithmetic); MI355_ARITH_EXACT is not available for such a batch");
        cppoptlib::mi355::CheckOwnMatrixKey(functions);
        if (arithmetic_ == MI355_ARITH_DEFAULT) {
          const size_t step = (functions.size() + 15) / 16;
          auto within = [&](size_t b) {
            if (cppoptlib::mi355::ConditionBound(functions[b]) > MI355_RIDGE_GRAM_MAX_CONDITION_BOUND)
              cppoptlib::mi355::Fail("MinimizeBatch(functions, states): cond(A^T A + lambda I) of a sampled function exceeds "
                                     "the envelope the normal-equation form is pinned in (MI355_RIDGE_GRAM_MAX_CONDITION_"
                                     "BOUND); SetArithmetic(MI355_ARITH_FMA) takes it regardless");
          };
          for (size_t b = 0; b < functions.size(); b += step) within(b);
          within(functions.size() - 1);
        }
      }
    }
    if (!own_matrices) cppoptlib::mi355::CheckSharedParams(functions, n);
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    DescStorage st;
    FillDesc(functions[0], n, B, /*host_per_problem=*/false, &st);
    if constexpr (cppoptlib::mi355::kMayHaveOwnMatrixForm<FunctionType>) {
      if (own_matrices) {
        // a different matrix per function (README.md:126-160 built once per data set): every problem's own parameters
        // travel in its per-problem row; the normal-equation form per problem (fused arithmetic, More-Thuente).
        st.d.objective = cppoptlib::mi355::OwnMatrixObjectiveId(functions[0]);
        st.params = cppoptlib::mi355::OwnMatrixParams(functions[0]);
        st.d.objective_params = st.params.data();
        st.d.n_params = static_cast<int32_t>(st.params.size());
        st.d.arithmetic = MI355_ARITH_DEFAULT;
        st.d.per_problem_stride = cppoptlib::mi355::PackOwnMatrixRows(functions, &st.per_problem);
      }
    }
    if (!own_matrices) st.d.per_problem_stride = cppoptlib::mi355::PackPerProblem(functions, &st.per_problem);
    st.d.per_problem_data = st.per_problem.empty() ? nullptr : st.per_problem.data();
    cppoptlib::mi355::Check(mi355_lbfgs_minimize_batch_host(ctx_->get(), &st.d, B, x0.data(), x.data(), f.data(), g.data(),
                                                            prog.data()),
                            "mi355_lbfgs_minimize_batch_host");
    auto out = cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
    for (size_t i = 0; i < out.size(); ++i) ReportHessianCondition(functions[i], &out[i], /*single_problem=*/false);
    return out;
  }

  // The same over a device group: the batch is cut into contiguous shards, one per member, each solved on its own
  // GPU by its own host thread; `flag` receives the RCCL all-reduced convergence record.
  std::vector<std::tuple<StateType, ProgressType>> ShardedMinimizeBatch(const FunctionType& function,
                                                                        const std::vector<StateType>& states,
                                                                        cppoptlib::mi355::DeviceGroup& group,
                                                                        cppoptlib::mi355::GlobalFlag* flag = nullptr) {
    const int64_t B = static_cast<int64_t>(states.size());
    if (B == 0) return {};
    const int n = static_cast<int>(states[0].x.size());
    const std::vector<double> x0 = cppoptlib::mi355::PackStates(states, n);
    std::vector<double> x(x0.size()), g(x0.size()), f(static_cast<size_t>(B));
    std::vector<mi355_lbfgs_progress> prog(static_cast<size_t>(B));
    DescStorage st;
    FillDesc(function, n, B, /*host_per_problem=*/true, &st);
    uint64_t record[3] = {0, 0, 0};
    cppoptlib::mi355::Check(mi355_lbfgs_group_minimize_batch_host(group.get(), &st.d, B, x0.data(), x.data(), f.data(),
                                                                  g.data(), prog.data(), record),
                            "mi355_lbfgs_group_minimize_batch_host");
    if (flag) {
      flag->total = record[0];
      flag->unconverged = record[1];
      flag->iterations = record[2];
    }
    auto out = cppoptlib::mi355::UnpackResults<StateType, ProgressType, VectorType>(n, B, x, f, g, prog);
    for (auto& r : out) ReportHessianCondition(function, &r, /*single_problem=*/false);
    return out;
  }

  // Batch-major HOST arrays in and out: x0[B][n] -> x[B][n], f[B], g[B][n], progress[B].  Pinned staging, persistent
  // device buffers and chunked overlap live behind the C entry point (mi355_lbfgs_minimize_batch_host).
  void MinimizeBatchRaw(const FunctionType& function, int n, int64_t B, const double* x0, double* x,
                        double* f, double* g, mi355_lbfgs_progress* progress, const mi355_lbfgs_trace* trace = nullptr) {
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    DescStorage st;
    FillDesc(function, n, B, /*host_per_problem=*/true, &st);
    st.d.trace = trace;
    cppoptlib::mi355::Check(mi355_lbfgs_minimize_batch_host(ctx_->get(), &st.d, B, x0, x, f, g, progress),
                            "mi355_lbfgs_minimize_batch_host");
  }

  // Device-resident batch: every array pointer is DEVICE memory on the context's device (x0_dev[B][n] -> x_dev[B][n],
  // f_dev[B], g_dev / progress_dev may be null); asynchronous on `stream` (a hipStream_t; null = the default stream) —
  // the caller synchronises.  This is the path the headline throughput is quoted on: nothing crosses PCIe.
  // Functions with per-problem data (e.g. the ridge objective's right-hand sides) pass them as a device array
  // per_problem_dev[B][per_problem_stride].
  void MinimizeBatchDevice(const FunctionType& function, int n, int64_t B, const double* x0_dev, double* x_dev,
                           double* f_dev, double* g_dev, mi355_lbfgs_progress* progress_dev, void* stream = nullptr,
                           const double* per_problem_dev = nullptr, int per_problem_stride = 0) {
    if (!ctx_) ctx_ = cppoptlib::mi355::Context::Default();
    DescStorage st;
    FillDesc(function, n, B, /*host_per_problem=*/false, &st);
    st.d.per_problem_data = per_problem_dev;
    st.d.per_problem_stride = per_problem_stride;
    cppoptlib::mi355::Check(
        mi355_lbfgs_minimize_batch(ctx_->get(), &st.d, B, x0_dev, x_dev, f_dev, g_dev, progress_dev, stream),
        "mi355_lbfgs_minimize_batch");
  }

 private:
  struct DescStorage {  // the desc and the host arrays it points into
    mi355_lbfgs_desc d{};
    std::vector<double> params, per_problem, hessian_diagonal;
  };
  void FillDesc(const FunctionType& function, int n, int64_t B, bool host_per_problem, DescStorage* st) const {
    cppoptlib::mi355::RequireObjective(function, "Lbfgs");
    st->params = cppoptlib::mi355::ObjectiveParams(function, n);
    mi355_lbfgs_desc& d = st->d;
    d.objective = cppoptlib::mi355::ObjectiveId(function, arithmetic_);
    d.linesearch = LineSearch<FunctionType, 1>::kDeviceLineSearch;
    d.n = n;
    d.m = m;
    d.objective_params = st->params.empty() ? nullptr : st->params.data();
    d.n_params = static_cast<int32_t>(st->params.size());
    d.arithmetic = arithmetic_;
    // objectives with per-problem data: this function object describes ONE problem, so its data row is replicated
    // for every start state of the batch
    if (host_per_problem) {
      d.per_problem_stride = cppoptlib::mi355::PackPerProblem(function, B, &st->per_problem);
      d.per_problem_data = st->per_problem.empty() ? nullptr : st->per_problem.data();
    }
    d.history_placement = MI355_HISTORY_AUTO;
    // lbfgs.h:116-139 of the reference: Second-mode functions get the diagonal preconditioner
    if constexpr (FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second) {
      if (cppoptlib::mi355::UsesHessianFromFunctor(function)) {
        // a non-constant Hessian: the device functor's hess_diag supplies diag H(x) at every iterate
        d.hessian_from_functor = 1;
        // progress.h:203-210, :318-325: ||H(x)|| ||H(x)^-1|| of every iterate against the threshold — evaluated by the
        // solve kernel from the functor's hess_full (n <= 64; the library refuses what it has no kernel for)
        d.hessian_condition_stop = static_cast<double>(this->stopping_progress.condition_hessian);
      } else {
        st->hessian_diagonal = cppoptlib::mi355::ConstantHessianDiagonal(function, n);
        d.hessian_diagonal = st->hessian_diagonal.data();
        // progress.h:203-210: condition_hessian = ||H|| ||H^-1|| of the (constant) Hessian, tested last in every Update
        VectorType zero(n);
        for (int i = 0; i < n; ++i) zero[i] = 0;
        cppoptlib::mi355::Check(mi355_lbfgs_hessian_condition(HostHessian(function, zero, n).data(), n, &d.hessian_condition),
                                "mi355_lbfgs_hessian_condition");
        d.hessian_condition_stop = static_cast<double>(this->stopping_progress.condition_hessian);
        hessian_condition_ = d.hessian_condition;
      }
    }
    d.stop = this->stopping_progress.ToDeviceStop();
  }

  // H(x) of the host functor, row major
  static std::vector<double> HostHessian(const FunctionType& function, const VectorType& x, int n) {
    MatrixType hessian;
    function(x, nullptr, &hessian);
    std::vector<double> h(static_cast<size_t>(n) * n);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < n; ++j) h[static_cast<size_t>(i) * n + j] = static_cast<double>(hessian(i, j));
    return h;
  }

  // Progress::condition_hessian of a returned (state, progress) pair (progress.h:203-210 of the reference: ||H|| ||H^-1||
  // at the current x, recomputed in every Update; what a caller sees is the value at the returned x).  A constant Hessian
  // has one value for the whole batch (computed in FillDesc, also the stopping test's input).  A function whose Hessian
  // is evaluated on the device (kDeviceHessianFromFunctor) gets it from the HOST functor's Hessian at the returned point
  // — the same number the reference's last Update produced — which is an O(n^3) LU per problem on one host thread: the
  // one-problem Minimize of the reference's API always pays it, MinimizeBatch only when the caller asked for the quantity
  // (stopping_progress.condition_hessian > 0); otherwise the field of a batch result stays 0.
  void ReportHessianCondition(const FunctionType& function, std::tuple<StateType, ProgressType>* result,
                              bool single_problem) const {
    if constexpr (FunctionType::Differentiability == cppoptlib::function::DifferentiabilityMode::Second) {
      double condition = hessian_condition_;
      if (cppoptlib::mi355::UsesHessianFromFunctor(function)) {
        condition = 0;
        if (single_problem || this->stopping_progress.condition_hessian > 0) {
          const StateType& state = std::get<0>(*result);
          const int n = static_cast<int>(state.x.size());
          cppoptlib::mi355::Check(mi355_lbfgs_hessian_condition(HostHessian(function, state.x, n).data(), n, &condition),
                                  "mi355_lbfgs_hessian_condition");
        }
      }
      std::get<1>(*result).condition_hessian = static_cast<ScalarType>(condition);
    } else {
      (void)function;
      (void)result;
      (void)single_problem;
    }
  }

  int arithmetic_ = MI355_ARITH_DEFAULT;
  mutable double hessian_condition_ = 0;  // of the last Second-mode solve: reported in the returned Progress
  std::shared_ptr<cppoptlib::mi355::Context> ctx_;
};

}  // namespace cppoptlib::solver
#endif  // INCLUDE_CPPOPTLIB_SOLVER_LBFGS_H_
