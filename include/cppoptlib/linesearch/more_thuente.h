// cppoptlib/linesearch/more_thuente.h — Moré–Thuente line search (device).
//
// The reference's MoreThuente<FunctionType, Ord> (linesearch/more_thuente.h)
// is the default `LineSearch` template argument of Lbfgs (solver/lbfgs.h:41).
// Here the class is the tag that selects the device line search
// (csrc/more_thuente_device.hpp); `cstep` — the piece the reference unit-tests
// directly (src/test/cstep_test.cc) — is callable from the host and executes
// the DEVICE implementation on one record.
#ifndef INCLUDE_CPPOPTLIB_LINESEARCH_MORE_THUENTE_H_
#define INCLUDE_CPPOPTLIB_LINESEARCH_MORE_THUENTE_H_

#include "../../mi355_lbfgs.h"
#include "../mi355/context.h"

namespace cppoptlib::solver::linesearch {

template <typename FunctionType, int Ord>
class MoreThuente {
 public:
  using ScalarType = typename FunctionType::ScalarType;
  static constexpr int kDeviceLineSearch = MI355_LS_MORE_THUENTE;

  // Same parameter list as the reference's cstep (more_thuente.h:261-267).
  static int cstep(ScalarType& stx, ScalarType& fx, ScalarType& dx, ScalarType& sty, ScalarType& fy,
                   ScalarType& dy, ScalarType& stp, ScalarType& fp, ScalarType& dp, bool& brackt,
                   ScalarType& stpmin, ScalarType& stpmax, int& info) {
    double rec[13] = {double(stx), double(fx),         double(dx),     double(sty),    double(fy),
                      double(dy),  double(stp),        double(fp),     double(dp),     brackt ? 1.0 : 0.0,
                      double(stpmin), double(stpmax),  double(info)};
    int32_t rc = 0;
    auto ctx = cppoptlib::mi355::Context::Default();
    cppoptlib::mi355::Check(mi355_lbfgs_cstep_host(ctx->get(), 1, rec, &rc), "mi355_lbfgs_cstep_host");
    stx = ScalarType(rec[0]); fx = ScalarType(rec[1]); dx = ScalarType(rec[2]);
    sty = ScalarType(rec[3]); fy = ScalarType(rec[4]); dy = ScalarType(rec[5]);
    stp = ScalarType(rec[6]);
    brackt = rec[9] != 0.0;
    info = static_cast<int>(rec[12]);
    return rc;
  }
};

}  // namespace cppoptlib::solver::linesearch
#endif  // INCLUDE_CPPOPTLIB_LINESEARCH_MORE_THUENTE_H_
