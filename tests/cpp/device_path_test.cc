// The parts of the boundary the reference does not have, driven from plain C++17:
//   * the callback replay: step_callback_ sees what the reference's would (solver/solver.h:196-222);
//   * MinimizeBatchDevice: device-resident arrays + a stream, nothing crosses PCIe;
//   * the host-pointer pipeline in its chunked form (MI355_HOST_STAGE_BYTES makes a small batch span chunks);
//   * ShardedMinimizeBatch over a device group (two contexts on device 0 here) with the RCCL all-reduced record.
// Device memory is handled through four runtime entry points declared by hand (this file is compiled by g++ without
// the HIP headers, like every user of the drop-in headers would be).
#include <cstdlib>
#include <sstream>
#include <string>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

extern "C" {
int hipMalloc(void** ptr, size_t bytes);
int hipFree(void* ptr);
int hipMemcpy(void* dst, const void* src, size_t bytes, int kind);  // 1 = host to device, 2 = device to host
int hipDeviceSynchronize(void);
}

using Function = cppoptlib::function::Rosenbrock<>;
using Solver = cppoptlib::solver::Lbfgs<Function, 6>;
using State = Solver::StateType;

static std::vector<State> Starts(int B, int n) {
  std::vector<State> starts;
  for (int b = 0; b < B; ++b) {
    Function::VectorType x(n);
    for (int i = 0; i < n; ++i) x[i] = (i % 2 ? 1.0 : -1.2) + 0.0007 * ((b * 37 + i * 11) % 101 - 50);
    starts.emplace_back(x);
  }
  return starts;
}

int main() {
  Function f;
  // ---- callback replay ---------------------------------------------------------------------------------
  {
    Solver solver;
    struct Seen { size_t it; double value, x_delta, gnorm; cppoptlib::solver::Status status; double x0; };
    std::vector<Seen> seen;
    solver.SetCallback([&](const Function&, const State& s, const Solver::ProgressType& p) {
      seen.push_back({p.num_iterations, s.value, double(p.x_delta), double(p.gradient_norm), p.status, s.x[0]});
    });
    Function::VectorType x(2);
    x[0] = -1.2; x[1] = 1.0;
    auto [sol, st] = solver.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_EQ(seen.size(), size_t(st.num_iterations) + 1);
    EXPECT_NEAR(seen.front().value, 24.2, 1e-12);                       // f(-1.2, 1), evaluated like solver.h:189-192
    EXPECT_EQ(seen.front().it, size_t(0));
    EXPECT_TRUE(seen.front().status == cppoptlib::solver::Status::NotStarted);
    for (size_t k = 1; k < seen.size(); ++k) {
      EXPECT_EQ(seen[k].it, k);                                         // one call per iteration, in order
      if (k + 1 < seen.size()) EXPECT_TRUE(seen[k].status == cppoptlib::solver::Status::Continue);
      EXPECT_TRUE(seen[k].value <= seen[k - 1].value);                  // More-Thuente: sufficient decrease
    }
    EXPECT_TRUE(seen.back().status == st.status);
    EXPECT_EQ(seen.back().value, sol.value);
    EXPECT_EQ(seen.back().x0, sol.x[0]);
    EXPECT_EQ(seen.back().x_delta, double(st.x_delta));
    // the traced solve is the same solve: identical to one without a callback
    Solver plain;
    auto [sol2, st2] = plain.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_EQ(sol2.value, sol.value);
    EXPECT_EQ(st2.num_iterations, st.num_iterations);
    // PrintProgressCallback prints the reference's block per call (solver.h:57-139): a header line, Value, X, Gradient,
    // Gradient Norm, X Delta, F Delta and a closing rule — eight lines for a First-mode function
    std::ostringstream os;
    Solver printing;
    printing.SetCallback(cppoptlib::solver::PrintProgressCallback<Function, State>(os));
    printing.Minimize(f, cppoptlib::function::FunctionState(x));
    size_t lines = 0, blocks = 0;
    const std::string printed = os.str();
    for (char c : printed) lines += (c == '\n');
    for (size_t at = printed.find("--- Iteration:"); at != std::string::npos; at = printed.find("--- Iteration:", at + 1)) ++blocks;
    EXPECT_EQ(blocks, size_t(st.num_iterations) + 1);
    EXPECT_EQ(lines, 8 * blocks);
    EXPECT_TRUE(printed.find("  Gradient Norm:") != std::string::npos && printed.find("  F Delta:") != std::string::npos);
    // an iteration limit shorter than the solve: the replay ends on IterationLimit
    Solver limited;
    limited.stopping_progress.num_iterations = 4;
    int calls = 0;
    cppoptlib::solver::Status last = cppoptlib::solver::Status::NotStarted;
    limited.SetCallback([&](const Function&, const State&, const Solver::ProgressType& p) { ++calls; last = p.status; });
    limited.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_EQ(calls, 6);   // 5 iterations (strict '>', progress.h:212-216) + the start state
    EXPECT_TRUE(last == cppoptlib::solver::Status::IterationLimit);
    // Lbfgsb replays the same way
    cppoptlib::solver::Lbfgsb<Function> boxed;
    int bcalls = 0;
    boxed.SetCallback([&](const Function&, const State&, const cppoptlib::solver::Lbfgsb<Function>::ProgressType&) { ++bcalls; });
    auto [bs, bp] = boxed.Minimize(f, cppoptlib::function::FunctionState(x));
    EXPECT_EQ(bcalls, int(bp.num_iterations) + 1);
  }
  // ---- device-resident batch -----------------------------------------------------------------------------
  const int B = 3000, n = 32;
  const std::vector<State> starts = Starts(B, n);
  Solver solver;
  solver.stopping_progress.x_delta = 1e-11;
  solver.stopping_progress.past = 0;
  solver.stopping_progress.gradient_norm = 1e-8;
  const auto host = solver.MinimizeBatch(f, starts);
  {
    std::vector<double> x0(size_t(B) * n), x(x0.size()), g(x0.size()), fv(B);
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < n; ++i) x0[size_t(b) * n + i] = starts[b].x[i];
    std::vector<mi355_lbfgs_progress> prog(B);
    void *d_x0 = nullptr, *d_x = nullptr, *d_g = nullptr, *d_f = nullptr, *d_p = nullptr;
    EXPECT_EQ(hipMalloc(&d_x0, x0.size() * 8), 0);
    EXPECT_EQ(hipMalloc(&d_x, x0.size() * 8), 0);
    EXPECT_EQ(hipMalloc(&d_g, x0.size() * 8), 0);
    EXPECT_EQ(hipMalloc(&d_f, size_t(B) * 8), 0);
    EXPECT_EQ(hipMalloc(&d_p, size_t(B) * sizeof(mi355_lbfgs_progress)), 0);
    EXPECT_EQ(hipMemcpy(d_x0, x0.data(), x0.size() * 8, 1), 0);
    solver.MinimizeBatchDevice(f, n, B, static_cast<const double*>(d_x0), static_cast<double*>(d_x),
                               static_cast<double*>(d_f), static_cast<double*>(d_g),
                               static_cast<mi355_lbfgs_progress*>(d_p), /*stream=*/nullptr);
    EXPECT_EQ(hipDeviceSynchronize(), 0);
    EXPECT_EQ(hipMemcpy(x.data(), d_x, x.size() * 8, 2), 0);
    EXPECT_EQ(hipMemcpy(fv.data(), d_f, size_t(B) * 8, 2), 0);
    EXPECT_EQ(hipMemcpy(prog.data(), d_p, size_t(B) * sizeof(mi355_lbfgs_progress), 2), 0);
    int mismatches = 0;
    for (int b = 0; b < B; ++b) {
      if (fv[b] != std::get<0>(host[b]).value || prog[b].num_iterations != std::get<1>(host[b]).num_iterations) ++mismatches;
      for (int i = 0; i < n; ++i) mismatches += (x[size_t(b) * n + i] != std::get<0>(host[b]).x[i]);
    }
    EXPECT_EQ(mismatches, 0);   // the host entry point is the same kernel behind a staging pipeline
    hipFree(d_x0); hipFree(d_x); hipFree(d_g); hipFree(d_f); hipFree(d_p);
  }
  // ---- chunked host pipeline: 3000 problems through slots of 256 KiB --------------------------------------
  {
    setenv("MI355_HOST_STAGE_BYTES", "262144", 1);
    const auto chunked = solver.MinimizeBatch(f, starts);
    unsetenv("MI355_HOST_STAGE_BYTES");
    int mismatches = 0;
    for (int b = 0; b < B; ++b) {
      mismatches += (std::get<0>(chunked[b]).value != std::get<0>(host[b]).value);
      mismatches += (std::get<1>(chunked[b]).num_iterations != std::get<1>(host[b]).num_iterations);
      for (int i = 0; i < n; ++i) mismatches += (std::get<0>(chunked[b]).x[i] != std::get<0>(host[b]).x[i]);
      for (int i = 0; i < n; ++i) mismatches += (std::get<0>(chunked[b]).gradient[i] != std::get<0>(host[b]).gradient[i]);
    }
    EXPECT_EQ(mismatches, 0);
  }
  // ---- device group: two contexts on device 0, contiguous shards, RCCL all-reduce of the record -----------
  {
    cppoptlib::mi355::DeviceGroup group({0, 0});
    EXPECT_EQ(group.size(), 2);
    cppoptlib::mi355::GlobalFlag flag;
    const auto sharded = solver.ShardedMinimizeBatch(f, starts, group, &flag);
    EXPECT_EQ(flag.total, uint64_t(B));
    EXPECT_EQ(flag.unconverged, uint64_t(0));
    EXPECT_TRUE(flag.all_converged());
    uint64_t iterations = 0;
    int mismatches = 0;
    for (int b = 0; b < B; ++b) {
      iterations += std::get<1>(host[b]).num_iterations;
      mismatches += (std::get<0>(sharded[b]).value != std::get<0>(host[b]).value);
      for (int i = 0; i < n; ++i) mismatches += (std::get<0>(sharded[b]).x[i] != std::get<0>(host[b]).x[i]);
    }
    EXPECT_EQ(mismatches, 0);          // sharding does not change a single bit
    EXPECT_EQ(flag.iterations, iterations);
    // a limit nobody can meet: the record says so
    Solver limited = solver;
    limited.stopping_progress.num_iterations = 3;
    limited.ShardedMinimizeBatch(f, starts, group, &flag);
    EXPECT_EQ(flag.unconverged, uint64_t(B));
    EXPECT_TRUE(!flag.all_converged());
  }
  TEST_MAIN_END();
}
