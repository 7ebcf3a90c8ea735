// dispatch_lbfgsb.hip — the L-BFGS-B kernels (see engine_internal.hpp).
#define MI355_DISPATCH_TU 1
#define MI355_DISPATCH_LBFGSB_TU 1
#include "engine_internal.hpp"

namespace mi355 {
int dispatch_lbfgsb_e(mi355_lbfgs_ctx* ctx, int E, int objective, int linesearch, const LbfgsbArgs& args,
                      hipStream_t stream) {
  switch (E) {
    case 1: return dispatch_lbfgsb<1>(ctx, objective, linesearch, args, stream);
    case 2: return dispatch_lbfgsb<2>(ctx, objective, linesearch, args, stream);
    case 4: return dispatch_lbfgsb<4>(ctx, objective, linesearch, args, stream);
    case 8: return dispatch_lbfgsb_wide(ctx, objective, linesearch, args, stream);  // 64 < n <= 128
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "elems_per_lane must be 1, 2 or 4");
}
}  // namespace mi355
