// dispatch_lbfgsb_caps_b.hip — Lbfgsb<F, m, HagerZhang> for history sizes 6..10 (n <= 64): m = 6..8 with sixteen lanes
// per problem, m = 9, 10 with thirty-two (see dispatch_lbfgsb_caps_a.hip).
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {

template <class Obj1, class Obj2, class Obj4>
static int hz_by_mapping(mi355_lbfgs_ctx* ctx, int W, int E, const LbfgsbArgs& args, hipStream_t stream) {
  constexpr int HZ = MI355_LS_HAGER_ZHANG;
  if (W == 16) {
    switch (E) {
      case 1: return launch_lbfgsb<1, Obj1, 8, HZ>(ctx, args, stream);
      case 2: return launch_lbfgsb<2, Obj2, 8, HZ>(ctx, args, stream);
      case 4: return launch_lbfgsb<4, Obj4, 8, HZ>(ctx, args, stream);
    }
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with the Hager-Zhang line search is built for n <= 64");
}

int dispatch_lbfgsb_caps_b(mi355_lbfgs_ctx* ctx, int W, int E, int objective, int linesearch, const LbfgsbArgs& args,
                           hipStream_t stream) {
  if (linesearch != MI355_LS_HAGER_ZHANG) return fail(MI355_ERR_INVALID_ARGUMENT, "Hager-Zhang unit");
  if (W == 32) return dispatch_lbfgsb_caps_b32(ctx, E, objective, args, stream);   // m = 9, 10: dispatch_lbfgsb_caps_d.hip
  switch (objective) {
    case MI355_OBJ_ROSENBROCK:
      return hz_by_mapping<RosenbrockObjective, RosenbrockObjective, RosenbrockObjective>(ctx, W, E, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC:
      return hz_by_mapping<DiagQuadraticObjective<1>, DiagQuadraticObjective<2>, DiagQuadraticObjective<4>>(ctx, W, E, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B with the Hager-Zhang line search is built for the Rosenbrock and DiagQuadratic objectives");
}

}  // namespace mi355
