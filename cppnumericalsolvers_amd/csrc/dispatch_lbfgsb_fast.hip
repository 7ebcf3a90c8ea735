// dispatch_lbfgsb_fast.hip — the L-BFGS-B kernels of the relaxed-algebra policy (lbfgsb_fast_kernel.hpp; see
// engine_internal.hpp).
#include "lbfgsb_fast_dispatch.hpp"

namespace mi355 {
namespace {
template <int E>
int by_objective(mi355_lbfgs_ctx* ctx, int objective, const LbfgsbArgs& args, hipStream_t stream) {
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return dispatch_lbfgsb_fast_m<E, RosenbrockObjective>(ctx, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC: return dispatch_lbfgsb_fast_m<E, DiagQuadraticObjective<E>>(ctx, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "relaxed-algebra L-BFGS-B is built for the Rosenbrock and DiagQuadratic objectives");
}
}  // namespace

int dispatch_lbfgsb_fast(mi355_lbfgs_ctx* ctx, int W, int E, int objective, const LbfgsbArgs& args, hipStream_t stream) {
  if (W == 32) return dispatch_lbfgsb_fast_w32(ctx, E, objective, args, stream);   // m = 9, 10: dispatch_lbfgsb_fast_w32.hip
  switch (E) {
    case 1: return by_objective<1>(ctx, objective, args, stream);
    case 2: return by_objective<2>(ctx, objective, args, stream);
    case 4: return by_objective<4>(ctx, objective, args, stream);
    case 8: return by_objective<8>(ctx, objective, args, stream);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "elems_per_lane must be 1, 2, 4 or 8");
}
}  // namespace mi355
