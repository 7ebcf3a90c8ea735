// dispatch_lbfgsb_fast_w32.hip — the relaxed-algebra L-BFGS-B kernels for history sizes 9 and 10 (round 4): 2M = 20 rows
// of the compact representation take two DPP rows, so a problem is owned by thirty-two lanes (two problems per wavefront);
// n <= 64 with one or two coordinates per lane.  Same kernel source as the sixteen-lane ones (lbfgsb_fast_kernel.hpp).
#include "lbfgsb_fast_dispatch.hpp"

namespace mi355 {

int dispatch_lbfgsb_fast_w32(mi355_lbfgs_ctx* ctx, int E, int objective, const LbfgsbArgs& args, hipStream_t stream) {
  switch (objective) {
    case MI355_OBJ_ROSENBROCK:
      if (E == 1) return launch_lbfgsb_fast_w32<1, RosenbrockObjective>(ctx, args, stream);
      if (E == 2) return launch_lbfgsb_fast_w32<2, RosenbrockObjective>(ctx, args, stream);
      break;
    case MI355_OBJ_DIAG_QUADRATIC:
      if (E == 1) return launch_lbfgsb_fast_w32<1, DiagQuadraticObjective<1>>(ctx, args, stream);
      if (E == 2) return launch_lbfgsb_fast_w32<2, DiagQuadraticObjective<2>>(ctx, args, stream);
      break;
    default:
      return fail(MI355_ERR_UNSUPPORTED, "relaxed-algebra L-BFGS-B is built for the Rosenbrock and DiagQuadratic objectives");
  }
  return fail(MI355_ERR_UNSUPPORTED, "relaxed-algebra L-BFGS-B with m = 9, 10 is built for n <= 64");
}

}  // namespace mi355
