// The reference's src/test/cstep_test.cc (seven known-answer tests, :54-204) against the
// DEVICE cstep, through the same static call the reference test uses:
//   MoreThuente<FunctionType, 1>::cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax, info)
#include "cppoptlib/function.h"
#include "cppoptlib/linesearch/more_thuente.h"
#include "mini_test.h"

using LineSearch = cppoptlib::solver::linesearch::MoreThuente<cppoptlib::function::Rosenbrock<>, 1>;

static int CallCstep(double& stx, double& fx, double& dx, double& sty, double& fy, double& dy, double& stp,
                     double fp, double dp, bool& brackt, double stpmin, double stpmax, int& info) {
  return LineSearch::cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax, info);
}

int main() {
  {  // Case 1: quadratic model recovers the minimiser (:54-72)
    double stx = 0, fx = 0, dx = -1, sty = 0, fy = 0, dy = 0, stp = 3;
    bool brackt = false;
    int info = 0;
    EXPECT_EQ(0, CallCstep(stx, fx, dx, sty, fy, dy, stp, 1.5, 2.0, brackt, 0.0, 10.0, info));
    EXPECT_EQ(1, info);
    EXPECT_TRUE(brackt);
    EXPECT_NEAR(1.0, stp, 1e-12);
    EXPECT_EQ(0.0, stx);
    EXPECT_EQ(3.0, sty);
    EXPECT_EQ(1.5, fy);
    EXPECT_EQ(2.0, dy);
  }
  {  // Case 2: derivative sign flip (:81-100)
    double stx = 0, fx = 2, dx = -2, sty = 0, fy = 0, dy = 0, stp = 3;
    bool brackt = false;
    int info = 0;
    EXPECT_EQ(0, CallCstep(stx, fx, dx, sty, fy, dy, stp, 0.5, 1.0, brackt, 0.0, 10.0, info));
    EXPECT_EQ(2, info);
    EXPECT_TRUE(brackt);
    EXPECT_NEAR(2.0, stp, 1e-12);
    EXPECT_EQ(3.0, stx);
    EXPECT_EQ(0.5, fx);
    EXPECT_EQ(1.0, dx);
    EXPECT_EQ(0.0, sty);
    EXPECT_EQ(2.0, fy);
    EXPECT_EQ(-2.0, dy);
  }
  {  // Case 3: not bracketed, advances toward the minimiser (:109-127)
    double stx = 0, fx = 8, dx = -4, sty = 0, fy = 0, dy = 0, stp = 1;
    bool brackt = false;
    int info = 0;
    EXPECT_EQ(0, CallCstep(stx, fx, dx, sty, fy, dy, stp, 4.5, -3.0, brackt, 0.0, 20.0, info));
    EXPECT_EQ(3, info);
    EXPECT_TRUE(!brackt);
    EXPECT_TRUE(stp > 1.0);
    EXPECT_TRUE(stp <= 20.0);
    EXPECT_EQ(1.0, stx);
    EXPECT_EQ(4.5, fx);
    EXPECT_EQ(-3.0, dx);
  }
  {  // Case 4: extrapolates to stpmax (:138-151)
    double stx = 0, fx = 5, dx = -1, sty = 0, fy = 0, dy = 0, stp = 1;
    bool brackt = false;
    int info = 0;
    EXPECT_EQ(0, CallCstep(stx, fx, dx, sty, fy, dy, stp, 3.99, -1.03, brackt, 0.0, 50.0, info));
    EXPECT_EQ(4, info);
    EXPECT_TRUE(!brackt);
    EXPECT_EQ(50.0, stp);
  }
  {  // Clamp (:155-168)
    double stx = 0, fx = 0, dx = -1, sty = 0, fy = 0, dy = 0, stp = 3;
    bool brackt = false;
    int info = 0;
    EXPECT_EQ(0, CallCstep(stx, fx, dx, sty, fy, dy, stp, 1.5, 2.0, brackt, 0.1, 0.75, info));
    EXPECT_TRUE(stp >= 0.1 && stp <= 0.75);
  }
  {  // 0.66 safeguard (:175-191)
    double stx = 0, fx = 0, dx = -1, sty = 1, fy = 0.5, dy = 1.5, stp = 0.99;
    bool brackt = true;
    int info = 0;
    EXPECT_EQ(0, CallCstep(stx, fx, dx, sty, fy, dy, stp, 0.49, 1.4, brackt, 0.0, 2.0, info));
    EXPECT_EQ(1, info);
    EXPECT_TRUE(brackt);
    EXPECT_TRUE(stp >= 0.0 && stp <= 0.66 + 1e-12);
  }
  {  // Rejects non-descent input (:196-204)
    double stx = 0, fx = 0, dx = 1, sty = 0, fy = 0, dy = 0, stp = 1;
    bool brackt = false;
    int info = 0;
    EXPECT_EQ(-1, CallCstep(stx, fx, dx, sty, fy, dy, stp, 0.5, 0.5, brackt, 0.0, 10.0, info));
  }
  TEST_MAIN_END();
}
