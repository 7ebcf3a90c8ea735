// auglag_family.hip — the augmented-Lagrangian kernels for problems with constraint FAMILIES
// (mi355_al_problem::family_eq / family_ineq: hundreds of affine constraints given as matrices, as the reference's
// src/examples/svm_primal_al.cc:139-147 builds them one functor at a time; function_problem.h:57-84 takes vectors of any
// length).  Same kernels as auglag_fused.hip — lbfgs_solve_kernel<..., AugLagOuterLoop> — with the objective's family
// capacity switched on (csrc/auglag_device.hpp: four constraints per lane of the problem's segment); their own
// translation unit so that the table-only kernels keep their registers and LDS and the units compile in parallel.
#define MI355_DISPATCH_TU
#include "auglag_launch.hpp"

namespace mi355 {
namespace {

// the mappings of al_mapping (the Lbfgs inner solver)
template <class F>
int with_family_mapping(const Mapping& mp, F&& f) {
  if (mp.W == 8 && mp.E == 1) return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
  if (mp.W == 8 && mp.E == 2) return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
  if (mp.W == 16 && mp.E == 2) return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
  if (mp.W == 32 && mp.E == 2) return f(std::integral_constant<int, 32>{}, std::integral_constant<int, 2>{});
  if (mp.W == 64 && mp.E == 2) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 2>{});
  if (mp.W == 64 && mp.E == 4) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 4>{});
  return fail(MI355_ERR_INVALID_ARGUMENT, "no augmented-Lagrangian family kernel for this mapping");
}

}  // namespace

int auglag_launch_fused_family(mi355_lbfgs_ctx* ctx, const Mapping& mp, const SolveArgs& args,
                               const AugLagOuterArgs& outer_args, hipStream_t stream) {
  return with_family_mapping(mp, [&](auto w, auto e) {
    constexpr int W = decltype(w)::value, E = decltype(e)::value, FC = al_family_capacity(W);
    using Obj = AugLagObjective<W, E, NoUserTerms, FC>;
    using Outer = AugLagOuterLoop<W, E, NoUserTerms, FC>;
    constexpr int MR = 0;   // both history halves in the LDS ring, as the table-only kernels (auglag_launch.hpp)
    return launch_solve<W, E, Obj, MR, MI355_LS_MORE_THUENTE, kAlgLbfgs, Outer>(ctx, args, stream, outer_args);
  });
}

int auglag_launch_family_eval(const Mapping& mp, const SolveArgs& args, hipStream_t stream) {
  return with_family_mapping(mp, [&](auto w, auto e) {
    constexpr int W = decltype(w)::value, E = decltype(e)::value, FC = al_family_capacity(W);
    using Obj = AugLagObjective<W, E, NoUserTerms, FC>;
    constexpr int kSegs = kWave / W;
    const long long blocks = (args.B + kSegs - 1) / kSegs;
    const int lds = (Obj::shared_lds_doubles() + kSegs * Obj::kLdsDoubles) * static_cast<int>(sizeof(double));
    auto kern = eval_kernel<W, E, Obj>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(kWave), lds, stream, args);
    HIP_TRY(hipGetLastError());
    return static_cast<int>(MI355_OK);
  });
}

}  // namespace mi355
