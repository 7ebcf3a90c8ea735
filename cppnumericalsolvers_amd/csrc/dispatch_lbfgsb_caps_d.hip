// dispatch_lbfgsb_caps_d.hip — Lbfgsb<F, m, HagerZhang> for m = 9, 10 (thirty-two lanes per problem, n <= 64); see
// dispatch_lbfgsb_caps_b.hip.
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {

template <class Obj1, class Obj2>
static int hz32(mi355_lbfgs_ctx* ctx, int E, const LbfgsbArgs& args, hipStream_t stream) {
  constexpr int HZ = MI355_LS_HAGER_ZHANG;
  switch (E) {
    case 1: return launch_lbfgsb<1, Obj1, 10, HZ, NoOuterLoop, 32>(ctx, args, stream);
    case 2: return launch_lbfgsb<2, Obj2, 10, HZ, NoOuterLoop, 32>(ctx, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with the Hager-Zhang line search is built for n <= 64");
}

int dispatch_lbfgsb_caps_b32(mi355_lbfgs_ctx* ctx, int E, int objective, const LbfgsbArgs& args, hipStream_t stream) {
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return hz32<RosenbrockObjective, RosenbrockObjective>(ctx, E, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC: return hz32<DiagQuadraticObjective<1>, DiagQuadraticObjective<2>>(ctx, E, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with the Hager-Zhang line search is built for the Rosenbrock and DiagQuadratic objectives");
}

}  // namespace mi355
