// auglag_fused.hip — the augmented-Lagrangian outer loop fused into the persistent L-BFGS kernel
// (lbfgs_solve_kernel<..., AugLagOuterLoop>, csrc/auglag_device.hpp): its own translation unit so that the kernels of
// the two forms of the loop compile in parallel.
#define MI355_DISPATCH_TU
#include "auglag_internal.hpp"

namespace mi355 {

// The whole outer loop in the persistent L-BFGS kernel (AugLagOuterLoop): one launch per batch.
int auglag_launch_fused(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const SolveArgs& args,
                 const AugLagOuterArgs& outer, hipStream_t stream) {
  return with_mapping(mp, [&](auto w, auto e) {
    constexpr int W = decltype(w)::value, E = decltype(e)::value;
    if constexpr (W == 16 && E != 2) {  // mappings of the Lbfgsb inner solver only
      return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS kernel for this mapping");
    } else {
      using Obj = AugLagObjective<W, E>;
      using Outer = AugLagOuterLoop<W, E>;
      if (linesearch == MI355_LS_HAGER_ZHANG)
        return launch_solve<W, E, Obj, 0, MI355_LS_HAGER_ZHANG, kAlgLbfgs, Outer>(ctx, args, stream, outer);
      constexpr int MR = (E == 4) ? 0 : 10;
      return launch_solve<W, E, Obj, MR, MI355_LS_MORE_THUENTE, kAlgLbfgs, Outer>(ctx, args, stream, outer);
    }
  });
}

int auglag_launch_fused_box(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const LbfgsbArgs& args,
                            const AugLagOuterArgs& outer, hipStream_t stream) {
  return with_mapping(mp, [&](auto w, auto e) {
    constexpr int W = decltype(w)::value, E = decltype(e)::value;
    if constexpr (W != 16) {
      return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS-B kernel for this mapping");
    } else {
      using Obj = AugLagObjective<16, E>;
      using Outer = AugLagOuterLoop<16, E>;
      if (linesearch == MI355_LS_HAGER_ZHANG)
        return launch_lbfgsb<E, Obj, 5, MI355_LS_HAGER_ZHANG, Outer>(ctx, args, stream, outer);
      return launch_lbfgsb<E, Obj, 5, MI355_LS_MORE_THUENTE, Outer>(ctx, args, stream, outer);
    }
  });
}

}  // namespace mi355
