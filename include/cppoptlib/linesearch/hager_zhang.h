// cppoptlib/linesearch/hager_zhang.h — Hager–Zhang line search (device).
//
// The reference's HagerZhang<FunctionType, Ord> (linesearch/hager_zhang.h) is the alternative
// `LineSearch` template argument: `Lbfgs<F, m, HagerZhang>` (solver/lbfgs.h:41, hager_zhang.h:39-42).
// Here the class is the tag that selects the device line search (csrc/hager_zhang_device.hpp) inside
// the batched solve, and its three `Search` overloads — the public surface of the reference class
// (:63-116) — run ONE search on the device through mi355_lbfgs_hz_search_host, so the 1-D
// known-answer tests of the reference (src/test/hager_zhang_test.cc) read the same here.
#ifndef INCLUDE_CPPOPTLIB_LINESEARCH_HAGER_ZHANG_H_
#define INCLUDE_CPPOPTLIB_LINESEARCH_HAGER_ZHANG_H_

#include <vector>

#include "../../mi355_lbfgs.h"
#include "../mi355/context.h"
#include "../mi355/batch_driver.h"

namespace cppoptlib::solver::linesearch {

template <typename FunctionType, int Ord>
class HagerZhang {
 public:
  using ScalarType = typename FunctionType::ScalarType;
  using VectorType = typename FunctionType::VectorType;
  static constexpr int kDeviceLineSearch = MI355_LS_HAGER_ZHANG;
  static_assert(cppoptlib::mi355::kHasDeviceTwin<FunctionType>,
                "HagerZhang runs on the MI355X: the function type needs a device twin (kDeviceObjective / DeviceParams)");

  // step width only (hager_zhang.h:63-74)
  static ScalarType Search(const VectorType& x, const VectorType& search_direction, const FunctionType& function,
                           const ScalarType alpha_init = 1.0) {
    VectorType xo, go;
    ScalarType fo = 0;
    return Run(x, search_direction, function, alpha_init, &xo, &fo, &go);
  }

  // cached (f0, g0) overload (:79-95); f0 / g0 are recomputed on the device (same bits)
  static ScalarType Search(const VectorType& x, ScalarType /*f0*/, const VectorType& /*g0*/,
                           const VectorType& search_direction, const FunctionType& function, ScalarType alpha_init,
                           VectorType* x_out, ScalarType* f_out, VectorType* g_out) {
    VectorType xo, go;
    ScalarType fo = 0;
    const ScalarType alpha = Run(x, search_direction, function, alpha_init, &xo, &fo, &go);
    if (x_out) *x_out = xo;
    if (f_out) *f_out = fo;
    if (g_out) *g_out = go;
    return alpha;
  }

  // fully evaluated state in, fully evaluated state out (:100-116)
  template <class State>
  static State Search(const State& start, const VectorType& search_direction, const FunctionType& function,
                      const ScalarType alpha_init = ScalarType(1), ScalarType* alpha_out = nullptr) {
    VectorType xo, go;
    ScalarType fo = 0;
    const ScalarType alpha = Run(start.x, search_direction, function, alpha_init, &xo, &fo, &go);
    if (alpha_out) *alpha_out = alpha;
    return State(std::move(xo), fo, std::move(go));
  }

 private:
  static ScalarType Run(const VectorType& x, const VectorType& s, const FunctionType& function, ScalarType alpha_init,
                        VectorType* x_out, ScalarType* f_out, VectorType* g_out) {
    static_assert(!cppoptlib::mi355::HasPerProblemData<FunctionType>::value,
                  "stand-alone line searches take objectives without per-problem data");
    const int n = static_cast<int>(x.size());
    cppoptlib::mi355::RequireObjective(function, "HagerZhang::Search");
    if (cppoptlib::mi355::CarriesPerProblemData(function))
      cppoptlib::mi355::Fail("HagerZhang::Search: stand-alone line searches take objectives without per-problem data");
    const std::vector<double> params = cppoptlib::mi355::ObjectiveParams(function, n);
    mi355_lbfgs_desc d{};
    d.objective = cppoptlib::mi355::PlainObjectiveId(function);
    d.linesearch = MI355_LS_HAGER_ZHANG;
    d.n = n;
    d.m = 1;
    d.objective_params = params.empty() ? nullptr : params.data();
    d.n_params = static_cast<int32_t>(params.size());
    mi355_lbfgs_default_stop(0, &d.stop);
    std::vector<double> xin(static_cast<size_t>(n)), sin(xin.size()), xo(xin.size()), go(xin.size());
    for (int i = 0; i < n; ++i) {
      xin[static_cast<size_t>(i)] = x[i];
      sin[static_cast<size_t>(i)] = s[i];
    }
    double a0 = alpha_init, fo = 0, alpha = 0;
    auto ctx = cppoptlib::mi355::Context::Default();
    cppoptlib::mi355::Check(mi355_lbfgs_hz_search_host(ctx->get(), &d, 1, xin.data(), sin.data(), &a0, xo.data(), &fo,
                                                       go.data(), &alpha, nullptr),
                            "mi355_lbfgs_hz_search_host");
    *x_out = VectorType(n);
    *g_out = VectorType(n);
    for (int i = 0; i < n; ++i) {
      (*x_out)[i] = xo[static_cast<size_t>(i)];
      (*g_out)[i] = go[static_cast<size_t>(i)];
    }
    *f_out = fo;
    return alpha;
  }
};

}  // namespace cppoptlib::solver::linesearch
#endif  // INCLUDE_CPPOPTLIB_LINESEARCH_HAGER_ZHANG_H_
