// dispatch_w64.hip — kernels with 64 lanes per problem (see engine_internal.hpp).
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {
int dispatch_w64(mi355_lbfgs_ctx* ctx, int E, int objective, int mr, const SolveArgs& args, hipStream_t stream,
                 bool eval_only) {
  return dispatch_e<64>(ctx, E, objective, mr, args, stream, eval_only);
}
}  // namespace mi355
