// auglag_fused.hip — the augmented-Lagrangian outer loop fused into the persistent L-BFGS kernel
// (lbfgs_solve_kernel<..., AugLagOuterLoop>, csrc/auglag_device.hpp): its own translation unit so that the kernels of
// the two forms of the loop compile in parallel.
#define MI355_DISPATCH_TU
#include "auglag_launch.hpp"

namespace mi355 {

int auglag_launch_fused(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const SolveArgs& args,
                        const AugLagOuterArgs& outer, hipStream_t stream) {
  return AlLaunchTable<BuiltinTermsFor>::fused(ctx, mp, linesearch, args, outer, stream);
}

int auglag_launch_fused_box(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const LbfgsbArgs& args,
                            const AugLagOuterArgs& outer, hipStream_t stream) {
  return AlLaunchTable<BuiltinTermsFor>::fused_box(ctx, mp, linesearch, args, outer, stream);
}

}  // namespace mi355
