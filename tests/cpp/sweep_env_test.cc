// -DCPPOPT_SWEEP: the reference's parameter-sweep build reads five fields of the default stopping preset from the
// environment (progress.h:359-381 of the reference: CPPOPT_X_DELTA, CPPOPT_X_DELTA_VIOL, CPPOPT_GRAD_NORM, CPPOPT_PAST,
// CPPOPT_PAST_DELTA).  Host only: the preset is a plain value; a solver constructed afterwards carries it to the device
// in its mi355_lbfgs_stop.
#ifndef CPPOPT_SWEEP
#error "build this test with -DCPPOPT_SWEEP"
#endif
#include <cstdlib>

#include "cppoptlib/function.h"
#include "cppoptlib/solver/lbfgs.h"
#include "cppoptlib/solver/lbfgsb.h"
#include "mini_test.h"

using namespace cppoptlib::function;

int main() {
  using F = Rosenbrock<>;
  using State = FunctionState<double, F::Dimension>;
  for (const char* v : {"CPPOPT_X_DELTA", "CPPOPT_X_DELTA_VIOL", "CPPOPT_GRAD_NORM", "CPPOPT_PAST", "CPPOPT_PAST_DELTA"}) unsetenv(v);
  {  // nothing set: the preset of progress.h:353-431
    const auto p = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
    EXPECT_EQ(p.num_iterations, size_t(10000));
    EXPECT_EQ(p.x_delta, 1e-9);
    EXPECT_EQ(p.x_delta_violations, 1);
    EXPECT_EQ(p.gradient_norm, 1e-5);
    EXPECT_EQ(p.past, 3);
    EXPECT_EQ(p.past_delta, 1e-6);
    EXPECT_EQ(p.f_delta, 0.0);
    EXPECT_EQ(p.constraint_threshold, 1e-5);
  }
  setenv("CPPOPT_X_DELTA", "2.5e-12", 1);
  setenv("CPPOPT_X_DELTA_VIOL", "4", 1);
  setenv("CPPOPT_GRAD_NORM", "3e-7", 1);
  setenv("CPPOPT_PAST", "7", 1);
  setenv("CPPOPT_PAST_DELTA", "1e-11", 1);
  {
    const auto p = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
    EXPECT_EQ(p.x_delta, 2.5e-12);
    EXPECT_EQ(p.x_delta_violations, 4);
    EXPECT_EQ(p.gradient_norm, 3e-7);
    EXPECT_EQ(p.past, 7);
    EXPECT_EQ(p.past_delta, 1e-11);
    EXPECT_EQ(p.num_iterations, size_t(10000));   // not a sweep knob
    EXPECT_EQ(p.f_delta, 0.0);
    // the conservative preset starts from the default one and overrides gradient_norm / past / past_delta (:456-464)
    const auto c = cppoptlib::solver::ConservativeStoppingSolverProgress<F, State>();
    EXPECT_EQ(c.x_delta, 2.5e-12);
    EXPECT_EQ(c.x_delta_violations, 4);
    EXPECT_EQ(c.gradient_norm, 5e-6);
    EXPECT_EQ(c.past, 5);
    // a solver constructed now carries the swept preset (solver.h: stopping_progress = DefaultStoppingSolverProgress)
    cppoptlib::solver::Lbfgs<F> solver;
    EXPECT_EQ(solver.stopping_progress.gradient_norm, 3e-7);
    EXPECT_EQ(solver.stopping_progress.past, 7);
    const mi355_lbfgs_stop dev = solver.stopping_progress.ToDeviceStop();
    EXPECT_EQ(dev.x_delta, 2.5e-12);
    EXPECT_EQ(dev.x_delta_violations, 4);
    EXPECT_EQ(dev.gradient_norm, 3e-7);
    EXPECT_EQ(dev.past, 7);
    EXPECT_EQ(dev.past_delta, 1e-11);
    // Lbfgsb re-enables its relative f-delta test on top of the swept preset (lbfgsb.h:84-87)
    cppoptlib::solver::Lbfgsb<F> box;
    EXPECT_EQ(box.stopping_progress.x_delta, 2.5e-12);
    EXPECT_EQ(box.stopping_progress.f_delta, 2.22e-9);
  }
  // only some set: the others keep the preset
  unsetenv("CPPOPT_PAST");
  unsetenv("CPPOPT_X_DELTA");
  {
    const auto p = cppoptlib::solver::DefaultStoppingSolverProgress<F, State>();
    EXPECT_EQ(p.past, 3);
    EXPECT_EQ(p.x_delta, 1e-9);
    EXPECT_EQ(p.gradient_norm, 3e-7);
  }
  TEST_MAIN_END();
}
