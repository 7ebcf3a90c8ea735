// dispatch_lbfgsb_caps_f.hip — Lbfgsb<F, m, HagerZhang>, m = 6..10, 64 < n <= 128: thirty-two lanes x four coordinates
// (see dispatch_lbfgsb_caps_e.hip; its own unit so that the two compile side by side).
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {

int dispatch_lbfgsb_caps_f(mi355_lbfgs_ctx* ctx, int objective, const LbfgsbArgs& args, hipStream_t stream) {
  constexpr int HZ = MI355_LS_HAGER_ZHANG;
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return launch_lbfgsb<4, RosenbrockObjective, 10, HZ, NoOuterLoop, 32>(ctx, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC:
      return launch_lbfgsb<4, DiagQuadraticObjective<4>, 10, HZ, NoOuterLoop, 32>(ctx, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED,
              "L-BFGS-B with the Hager-Zhang line search is built for the Rosenbrock and DiagQuadratic objectives");
}

}  // namespace mi355
