// dispatch_w8.hip — kernels with 8 lanes per problem (see engine_internal.hpp).
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {
int dispatch_w8(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                 bool eval_only) {
  return dispatch_e<8>(ctx, E, objective, mr, args, stream, eval_only);
}
// four lanes per problem exist only with eight coordinates per lane (n <= 32: sixteen problems per wavefront)
int dispatch_w4(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                bool eval_only) {
  if (E != 8) return fail(MI355_ERR_INVALID_ARGUMENT, "four lanes per problem take eight coordinates per lane");
  return dispatch_e8<4>(ctx, objective, mr, args, stream, eval_only);
}
}  // namespace mi355
