// dispatch_lbfgsb_caps_a.hip — L-BFGS-B beyond the first set of shapes (round 3): history sizes 6..10 up to n = 256
// (32 lanes per problem, four / eight coordinates per lane) and on the ridge objective (m = 6..8 with sixteen lanes,
// m = 9, 10 with thirty-two).  The reference's Lbfgsb<F, m> is a template over any m (solver/lbfgsb.h:44) and
// dynamic in n; these are the remaining instantiations of lbfgsb_solve_kernel, in their own unit so that they compile
// next to the others.
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {

int dispatch_lbfgsb_caps_a(mi355_lbfgs_ctx* ctx, int W, int E, int objective, int linesearch, const LbfgsbArgs& args,
                           hipStream_t stream) {
  if (linesearch != MI355_LS_MORE_THUENTE)
    return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with the Hager-Zhang line search is built for n <= 64 on the Rosenbrock and DiagQuadratic objectives");
  constexpr int MT = MI355_LS_MORE_THUENTE;
  if (objective == MI355_OBJ_SQUARED_ERROR_RIDGE)   // (its kernels compile in their own unit)
    return dispatch_lbfgsb_caps_ridge(ctx, W, E, args, stream);
  if (W != 32) return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS-B kernel for this mapping");
  switch (objective) {  // 64 < n <= 256, m = 6..10
    case MI355_OBJ_ROSENBROCK:
      if (E == 4) return launch_lbfgsb<4, RosenbrockObjective, 10, MT, NoOuterLoop, 32>(ctx, args, stream);
      if (E == 8) return launch_lbfgsb<8, RosenbrockObjective, 10, MT, NoOuterLoop, 32>(ctx, args, stream);
      break;
    case MI355_OBJ_DIAG_QUADRATIC:
      if (E == 4) return launch_lbfgsb<4, DiagQuadraticObjective<4>, 10, MT, NoOuterLoop, 32>(ctx, args, stream);
      if (E == 8) return launch_lbfgsb<8, DiagQuadraticObjective<8>, 10, MT, NoOuterLoop, 32>(ctx, args, stream);
      break;
    default:
      return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B for n > 64 is built for the Rosenbrock and DiagQuadratic objectives");
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS-B kernel for this mapping");
}

}  // namespace mi355
