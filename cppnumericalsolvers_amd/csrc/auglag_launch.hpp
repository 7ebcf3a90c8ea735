// auglag_launch.hpp — the launchers of the augmented-Lagrangian kernels, as templates over the set of user term
// functors compiled in.  AlLaunchTable<BuiltinTermsFor> is the library's own table (the closed menu of
// mi355_al_term_kind; auglag.hip and auglag_fused.hip instantiate its halves); a library built with user term functors
// (_build.build(user_objectives=[dict(..., al_term=True)])) carries a generated unit that instantiates
// AlLaunchTable<UserTermsFor> for the mappings it asked for and registers it (UserAlRegistration): problems whose term
// table names a user functor (kind >= MI355_AL_TERM_USER) are solved by those kernels.
#pragma once
#include "auglag_internal.hpp"

namespace mi355 {

// TermsFor<W, E>: kBuilt (are the kernels of this mapping compiled in this unit?) and type (a TermList)
template <int W, int E>
struct BuiltinTermsFor {
  static constexpr bool kBuilt = true;
  using type = NoUserTerms;
};

template <template <int, int> class TermsFor>
struct AlLaunchTable {
  template <int W, int E>
  using ObjOf = AugLagObjective<W, E, typename TermsFor<W, E>::type>;
  template <int W, int E>
  using OuterOf = AugLagOuterLoop<W, E, typename TermsFor<W, E>::type>;

  static int wide_box_linesearch() {
    return fail(MI355_ERR_UNSUPPORTED, "the L-BFGS-B inner solver for 64 < n <= 128 is built with the More-Thuente line search");
  }
  static int not_built() {
    return fail(MI355_ERR_UNSUPPORTED, "the augmented-Lagrangian kernels with this library's user terms were not built for this dimension");
  }

  // Lbfgs<F, m, LineSearch> on the composite, one launch over the problems still active
  static int inner(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const SolveArgs& args, hipStream_t stream) {
    return with_mapping(mp, [&](auto w, auto e) {
      constexpr int W = decltype(w)::value, E = decltype(e)::value;
      if constexpr (al_box_only_mapping(W, E)) {
        return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS kernel for this mapping");
      } else if constexpr (!TermsFor<W, E>::kBuilt) {
        return not_built();
      } else {
        // Lbfgs<F, m, HagerZhang>: the LDS-ring kernel (as for the other objectives, engine_internal.hpp)
        if (linesearch == MI355_LS_HAGER_ZHANG)
          return launch_solve<W, E, ObjOf<W, E>, 0, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
        // both ring halves in LDS, as in the fused loop below (the register-history kernels of the composite spilled)
        constexpr int MR = 0;
        return launch_solve<W, E, ObjOf<W, E>, MR>(ctx, args, stream);
      }
    });
  }

  // Lbfgsb<F, m <= 5, LineSearch> on the composite (lbfgsb_solve_kernel: sixteen lanes per problem; thirty-two with
  // four coordinates each for 64 < n <= 128, More-Thuente)
  static int inner_box(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const LbfgsbArgs& args,
                       hipStream_t stream) {
    return with_mapping(mp, [&](auto w, auto e) {
      constexpr int W = decltype(w)::value, E = decltype(e)::value;
      if constexpr (W == 32 && E == 4) {
        if constexpr (!TermsFor<W, E>::kBuilt) {
          return not_built();
        } else {
          if (linesearch != MI355_LS_MORE_THUENTE) return wide_box_linesearch();
          return launch_lbfgsb<4, ObjOf<32, 4>, 5, MI355_LS_MORE_THUENTE, NoOuterLoop, 32>(ctx, args, stream);
        }
      } else if constexpr (W != 16) {
        return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS-B kernel for this mapping");
      } else if constexpr (!TermsFor<W, E>::kBuilt) {
        return not_built();
      } else {
        if (linesearch == MI355_LS_HAGER_ZHANG)
          return launch_lbfgsb<E, ObjOf<16, E>, 5, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
        return launch_lbfgsb<E, ObjOf<16, E>, 5>(ctx, args, stream);
      }
    });
  }

  static int composite_eval(const Mapping& mp, const SolveArgs& args, hipStream_t stream) {
    return with_mapping(mp, [&](auto w, auto e) {
      constexpr int W = decltype(w)::value, E = decltype(e)::value;
      if constexpr (!TermsFor<W, E>::kBuilt) {
        return not_built();
      } else {
        using Obj = ObjOf<W, E>;
        constexpr int kSegs = kWave / W;
        const long long blocks = (args.B + kSegs - 1) / kSegs;
        const int lds = (Obj::shared_lds_doubles() + kSegs * Obj::kLdsDoubles) * static_cast<int>(sizeof(double));
        auto kern = eval_kernel<W, E, Obj>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kWave), lds, stream, args);
        HIP_TRY(hipGetLastError());
        return static_cast<int>(MI355_OK);
      }
    });
  }

  static int outer(const Mapping& mp, const AugLagOuterArgs& args, hipStream_t stream) {
    return with_mapping(mp, [&](auto w, auto e) {
      constexpr int W = decltype(w)::value, E = decltype(e)::value;
      if constexpr (!TermsFor<W, E>::kBuilt) {
        return not_built();
      } else {
        using Obj = ObjOf<W, E>;
        constexpr int kSegs = kWave / W, kWaves = 4;
        const int lds = (Obj::shared_lds_doubles() + kWaves * kSegs * Obj::kLdsDoubles) * static_cast<int>(sizeof(double));
        const long long per_block = static_cast<long long>(kSegs) * kWaves;
        const long long blocks = (args.B + per_block - 1) / per_block;
        auto kern = auglag_outer_kernel<W, E, typename TermsFor<W, E>::type>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kWave * kWaves), lds, stream, args);
        HIP_TRY(hipGetLastError());
        return static_cast<int>(MI355_OK);
      }
    });
  }

  // The whole outer loop in the persistent L-BFGS kernel (AugLagOuterLoop): one launch per batch.
  static int fused(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const SolveArgs& args,
                   const AugLagOuterArgs& outer_args, hipStream_t stream) {
    return with_mapping(mp, [&](auto w, auto e) {
      constexpr int W = decltype(w)::value, E = decltype(e)::value;
      if constexpr (al_box_only_mapping(W, E)) {
        return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS kernel for this mapping");
      } else if constexpr (!TermsFor<W, E>::kBuilt) {
        return not_built();
      } else {
        using Obj = ObjOf<W, E>;
        using Outer = OuterOf<W, E>;
        if (linesearch == MI355_LS_HAGER_ZHANG)
          return launch_solve<W, E, Obj, 0, MI355_LS_HAGER_ZHANG, kAlgLbfgs, Outer>(ctx, args, stream, outer_args);
        // both halves of the inner solver's history in the LDS ring (MR = 0): with the y half in registers (MR = 10) the
        // composite's temporaries pushed these kernels to 64-288 B of scratch per lane; the ring kernels have none and
        // are as fast or faster up to n = 64, 5 % slower at n = 100 (profiles/r5_ab_al_ring.txt).  Same arithmetic, same bits.
        constexpr int MR = 0;
        return launch_solve<W, E, Obj, MR, MI355_LS_MORE_THUENTE, kAlgLbfgs, Outer>(ctx, args, stream, outer_args);
      }
    });
  }

  // The same around the L-BFGS-B kernel (Lbfgsb inner solver, sixteen lanes per problem).
  static int fused_box(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const LbfgsbArgs& args,
                       const AugLagOuterArgs& outer_args, hipStream_t stream) {
    return with_mapping(mp, [&](auto w, auto e) {
      constexpr int W = decltype(w)::value, E = decltype(e)::value;
      if constexpr (W == 32 && E == 4) {
        if constexpr (!TermsFor<W, E>::kBuilt) {
          return not_built();
        } else {
          if (linesearch != MI355_LS_MORE_THUENTE) return wide_box_linesearch();
          return launch_lbfgsb<4, ObjOf<32, 4>, 5, MI355_LS_MORE_THUENTE, OuterOf<32, 4>, 32>(ctx, args, stream, outer_args);
        }
      } else if constexpr (W != 16) {
        return fail(MI355_ERR_INVALID_ARGUMENT, "no L-BFGS-B kernel for this mapping");
      } else if constexpr (!TermsFor<W, E>::kBuilt) {
        return not_built();
      } else {
        using Obj = ObjOf<16, E>;
        using Outer = OuterOf<16, E>;
        if (linesearch == MI355_LS_HAGER_ZHANG)
          return launch_lbfgsb<E, Obj, 5, MI355_LS_HAGER_ZHANG, Outer>(ctx, args, stream, outer_args);
        return launch_lbfgsb<E, Obj, 5, MI355_LS_MORE_THUENTE, Outer>(ctx, args, stream, outer_args);
      }
    });
  }

  static AlLaunchers table() { return {&inner, &inner_box, &composite_eval, &outer, &fused, &fused_box}; }
};

}  // namespace mi355
