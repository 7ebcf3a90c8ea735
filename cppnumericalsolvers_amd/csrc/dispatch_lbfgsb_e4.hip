// dispatch_lbfgsb_e4.hip — the L-BFGS-B kernels with four and eight coordinates per lane (see dispatch_lbfgsb.hip).
#define MI355_DISPATCH_TU 1
#define MI355_DISPATCH_LBFGSB_TU 1
#include "engine_internal.hpp"

namespace mi355 {
int dispatch_lbfgsb_e4(mi355_lbfgs_ctx* ctx, int E, int objective, int linesearch, const LbfgsbArgs& args,
                       hipStream_t stream) {
  if (E == 4) return dispatch_lbfgsb<4>(ctx, objective, linesearch, args, stream);
  return dispatch_lbfgsb_wide(ctx, objective, linesearch, args, stream);  // 64 < n <= 128
}
}  // namespace mi355
