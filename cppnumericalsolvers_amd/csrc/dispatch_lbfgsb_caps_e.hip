// dispatch_lbfgsb_caps_e.hip — Lbfgsb<F, m, HagerZhang> above n = 64 (round 4): the reference's line-search template
// argument is independent of the dimension (solver/lbfgsb.h:44-49, linesearch/hager_zhang.h:39-42).  m <= 5: sixteen
// lanes x eight coordinates up to n = 128, thirty-two lanes x eight up to n = 256; m = 6..10: thirty-two lanes x four
// coordinates up to n = 128 (dispatch_lbfgsb_caps_f.hip).  Rosenbrock and DiagQuadratic; the kernels are lbfgsb_solve_kernel instantiations like the others.
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {

template <class Obj8>
static int hz_above_64(mi355_lbfgs_ctx* ctx, int W, int E, const LbfgsbArgs& args, hipStream_t stream) {
  constexpr int HZ = MI355_LS_HAGER_ZHANG;
  if (args.s.m <= 5) {
    if (W == 16 && E == 8) return launch_lbfgsb<8, Obj8, 5, HZ>(ctx, args, stream);
    if (W == 32 && E == 8) return launch_lbfgsb<8, Obj8, 5, HZ, NoOuterLoop, 32>(ctx, args, stream);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS-B kernel for this mapping");
}

int dispatch_lbfgsb_caps_e(mi355_lbfgs_ctx* ctx, int W, int E, int objective, const LbfgsbArgs& args, hipStream_t stream) {
  if (args.s.m > 5) {
    // m = 6..10: thirty-two lanes x four coordinates (64 < n <= 128; dispatch_lbfgsb_caps_f.hip).  With eight coordinates
    // per lane (n > 128) the twenty history columns, the box and the Hager-Zhang bracket exceed the 512 registers of a
    // lone wavefront by more than 1 KB of scratch per lane: refused rather than shipped.
    if (W == 32 && E == 4) return dispatch_lbfgsb_caps_f(ctx, objective, args, stream);
    return fail(MI355_ERR_UNSUPPORTED,
                "L-BFGS-B with the Hager-Zhang line search and m > 5 is built for n <= 128 (m <= 5: n <= 256)");
  }
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return hz_above_64<RosenbrockObjective>(ctx, W, E, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC: return hz_above_64<DiagQuadraticObjective<8>>(ctx, W, E, args, stream);
  }
  return fail(MI355_ERR_UNSUPPORTED,
              "L-BFGS-B with the Hager-Zhang line search is built for the Rosenbrock and DiagQuadratic objectives");
}

}  // namespace mi355
