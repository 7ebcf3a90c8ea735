// The host glue under load — what `make -C tests/cpp asan` / `tsan` run with the sanitized build of the library
// (the reference has no threads; the drop-in's host side does — worker threads for the pageable
// <-> pinned copies, three streams and two staging slots per context, one host thread per member of a device group):
//   * the host-pointer entry in its chunked, double-buffered form (MI355_HOST_STAGE_BYTES = 70,000: dozens of chunks),
//     with and without a trace;
//   * a device group of three contexts on device 0: host Lbfgs / Lbfgsb solves, the device-resident sharded solve;
//   * three host threads, each with its own context and solver, solving concurrently (distinct contexts may be used
//     from distinct threads: include/mi355_lbfgs.h).
// Every result is compared with the plain single-context solve: the pipeline must not change a bit.
#include <cstdlib>
#include <thread>
#include <vector>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using Function = cppoptlib::function::Rosenbrock<>;
using Solver = cppoptlib::solver::Lbfgs<Function, 6>;
using Boxed = cppoptlib::solver::Lbfgsb<Function>;
using State = Solver::StateType;

static std::vector<State> Starts(int B, int n, int salt) {
  std::vector<State> starts;
  for (int b = 0; b < B; ++b) {
    Function::VectorType x(n);
    for (int i = 0; i < n; ++i) x[i] = (i % 2 ? 1.0 : -1.2) + 0.0007 * ((b * 37 + i * 11 + salt) % 101 - 50);
    starts.emplace_back(x);
  }
  return starts;
}

template <class R>
static int Mismatches(const R& a, const R& b, int n) {
  int bad = (a.size() != b.size());
  int shown = 0;
  for (size_t k = 0; k < a.size() && k < b.size(); ++k) {
    int here = (std::get<0>(a[k]).value != std::get<0>(b[k]).value);
    here += (std::get<1>(a[k]).num_iterations != std::get<1>(b[k]).num_iterations);
    for (int i = 0; i < n; ++i) here += (std::get<0>(a[k]).x[i] != std::get<0>(b[k]).x[i]);
    if (here && shown++ < 6)
      std::printf("  problem %zu differs: value %.17g vs %.17g, iterations %zu vs %zu, status %d vs %d\n", k,
                  std::get<0>(a[k]).value, std::get<0>(b[k]).value, size_t(std::get<1>(a[k]).num_iterations),
                  size_t(std::get<1>(b[k]).num_iterations), int(std::get<1>(a[k]).status), int(std::get<1>(b[k]).status));
    bad += here;
  }
  if (bad) std::printf("  %d mismatching entries in %zu problems\n", bad, a.size());
  std::fflush(stdout);
  return bad;
}

int main() {
  Function f;
  const int n = 16, B = 1500;
  const auto starts = Starts(B, n, 0);
  Solver solver;
  const auto plain = solver.MinimizeBatch(f, starts);

  // ---- chunked host pipeline ---------------------------------------------------------------------------------
  setenv("MI355_HOST_STAGE_BYTES", "70000", 1);
  for (int rep = 0; rep < 3; ++rep) EXPECT_EQ(Mismatches(solver.MinimizeBatch(f, starts), plain, n), 0);
  {
    Solver traced = solver;
    size_t calls = 0;
    traced.SetCallback([&](const Function&, const State&, const Solver::ProgressType&) { ++calls; });
    auto [s, p] = traced.Minimize(f, starts[7]);
    EXPECT_EQ(calls, size_t(p.num_iterations) + 1);
    EXPECT_EQ(s.value, std::get<0>(plain[7]).value);
  }

  // ---- device group, three contexts on one device -------------------------------------------------------------
  {
    cppoptlib::mi355::DeviceGroup group({0, 0, 0});
    cppoptlib::mi355::GlobalFlag flag;
    for (int rep = 0; rep < 2; ++rep) {
      EXPECT_EQ(Mismatches(solver.ShardedMinimizeBatch(f, starts, group, &flag), plain, n), 0);
      EXPECT_EQ(flag.total, uint64_t(B));
      EXPECT_TRUE(flag.all_converged());
    }
    Boxed boxed;
    Function::VectorType lo(n), hi(n);
    for (int i = 0; i < n; ++i) {
      lo[i] = -1.5;
      hi[i] = 0.8;
    }
    boxed.SetBounds(lo, hi);
    const auto bplain = boxed.MinimizeBatch(f, starts);
    EXPECT_EQ(Mismatches(boxed.ShardedMinimizeBatch(f, starts, group, &flag), bplain, n), 0);
    EXPECT_EQ(flag.total, uint64_t(B));
  }
  unsetenv("MI355_HOST_STAGE_BYTES");

  // ---- three threads, three contexts -----------------------------------------------------------------------------
  {
    std::vector<int> bad(3, -1);
    std::vector<std::thread> threads;
    for (int t = 0; t < 3; ++t)
      threads.emplace_back([&, t]() {
        Solver mine;
        mine.SetContext(std::make_shared<cppoptlib::mi355::Context>(0));
        const auto st = Starts(400 + 50 * t, n, 0);
        int b = 0;
        for (int rep = 0; rep < 3; ++rep) {
          const auto r = mine.MinimizeBatch(f, st);
          for (size_t k = 0; k < r.size(); ++k) b += (std::get<0>(r[k]).value != std::get<0>(plain[k]).value);
        }
        bad[t] = b;
      });
    for (auto& th : threads) th.join();
    for (int t = 0; t < 3; ++t) EXPECT_EQ(bad[t], 0);
  }
  TEST_MAIN_END();
}
