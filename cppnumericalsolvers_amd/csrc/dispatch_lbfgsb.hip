// dispatch_lbfgsb.hip — the L-BFGS-B kernels with one or two coordinates per lane (see engine_internal.hpp; four and
// eight coordinates per lane: dispatch_lbfgsb_e4.hip — two units so that the library's cold build is not one long compile).
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {
int dispatch_lbfgsb_e(mi355_lbfgs_ctx* ctx, int E, int objective, int linesearch, const LbfgsbArgs& args,
                      hipStream_t stream) {
  switch (E) {
    case 1: return dispatch_lbfgsb<1>(ctx, objective, linesearch, args, stream);
    case 2: return dispatch_lbfgsb<2>(ctx, objective, linesearch, args, stream);
    case 4:
    case 8: return dispatch_lbfgsb_e4(ctx, E, objective, linesearch, args, stream);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "elems_per_lane must be 1, 2 or 4");
}
}  // namespace mi355
