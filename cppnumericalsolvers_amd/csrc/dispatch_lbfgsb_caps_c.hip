// dispatch_lbfgsb_caps_c.hip — L-BFGS-B on the ridge objective with history sizes 6..10 (m = 6..8 with sixteen lanes per
// problem, m = 9, 10 with thirty-two); see dispatch_lbfgsb_caps_a.hip.
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"

namespace mi355 {

int dispatch_lbfgsb_caps_ridge(mi355_lbfgs_ctx* ctx, int W, int E, const LbfgsbArgs& args, hipStream_t stream) {
  constexpr int MT = MI355_LS_MORE_THUENTE;
    if (W == 16) {  // m = 6..8
      switch (E) {
        case 1: return launch_lbfgsb<1, SquaredErrorRidgeObjective<16, 1>, 8>(ctx, args, stream);
        case 2: return launch_lbfgsb<2, SquaredErrorRidgeObjective<16, 2>, 8>(ctx, args, stream);
        case 4: return launch_lbfgsb<4, SquaredErrorRidgeObjective<16, 4>, 8>(ctx, args, stream);
      }
    } else {  // m = 9, 10
      switch (E) {
        case 1: return launch_lbfgsb<1, SquaredErrorRidgeObjective<32, 1>, 10, MT, NoOuterLoop, 32>(ctx, args, stream);
        case 2: return launch_lbfgsb<2, SquaredErrorRidgeObjective<32, 2>, 10, MT, NoOuterLoop, 32>(ctx, args, stream);
      }
    }
    return fail(MI355_ERR_UNSUPPORTED, "L-BFGS-B on the ridge objective is built for n <= 64");
}

}  // namespace mi355
