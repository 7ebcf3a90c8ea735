// auglag_internal.hpp — shared by the two translation units of the augmented-Lagrangian path (auglag.hip: C-ABI,
// lock-step loop, Lbfgsb inner solver; auglag_fused.hip: the fused outer loop inside the persistent L-BFGS kernel).
#pragma once
#include <initializer_list>
#include <type_traits>

#include "engine_internal.hpp"

#include "auglag_device.hpp"

namespace mi355 {

struct Mapping {
  int W, E;
};

// One mapping per padded dimension; the inner solver keeps its y history in registers (m <= 10).
inline bool al_mapping(int n, Mapping* out) {
  int P = 8;
  while (P < n) P <<= 1;
  switch (P) {
    case 8: *out = {8, 1}; return true;
    case 16: *out = {8, 2}; return true;
    case 32: *out = {16, 2}; return true;
    case 64: *out = {32, 2}; return true;
    case 128: *out = {64, 2}; return true;
    case 256: *out = {64, 4}; return true;
  }
  return false;
}

// Lbfgsb inner solver: sixteen lanes per problem up to n = 64, thirty-two with four coordinates each up to n = 128
// (the size of the reference's src/examples/svm_dual_al.cc: 100 dual variables).
constexpr int kAlBoxMaxN = 128;
inline bool al_box_mapping(int n, Mapping* out) {
  if (n > kAlBoxMaxN) return false;
  if (n > 64) {
    *out = {32, 4};
    return true;
  }
  *out = {16, (n <= 16) ? 1 : ((n <= 32) ? 2 : 4)};
  return true;
}
// mappings that only the Lbfgsb inner solver uses
constexpr bool al_box_only_mapping(int W, int E) { return (W == 16 && E != 2) || (W == 32 && E == 4); }

template <class F>
int with_mapping(const Mapping& mp, F&& f) {
  if (mp.W == 8 && mp.E == 1) return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{});
  if (mp.W == 8 && mp.E == 2) return f(std::integral_constant<int, 8>{}, std::integral_constant<int, 2>{});
  if (mp.W == 16 && mp.E == 1) return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{});
  if (mp.W == 16 && mp.E == 2) return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 2>{});
  if (mp.W == 16 && mp.E == 4) return f(std::integral_constant<int, 16>{}, std::integral_constant<int, 4>{});
  if (mp.W == 32 && mp.E == 2) return f(std::integral_constant<int, 32>{}, std::integral_constant<int, 2>{});
  if (mp.W == 32 && mp.E == 4) return f(std::integral_constant<int, 32>{}, std::integral_constant<int, 4>{});
  if (mp.W == 64 && mp.E == 2) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 2>{});
  if (mp.W == 64 && mp.E == 4) return f(std::integral_constant<int, 64>{}, std::integral_constant<int, 4>{});
  return fail(MI355_ERR_INVALID_ARGUMENT, "no augmented-Lagrangian kernel for this mapping");
}

// The launchers of one set of kernels (auglag_launch.hpp: AlLaunchTable); every entry dispatches on the mapping.
struct AlLaunchers {
  int (*inner)(mi355_lbfgs_ctx*, const Mapping&, int linesearch, const SolveArgs&, hipStream_t);
  int (*inner_box)(mi355_lbfgs_ctx*, const Mapping&, int linesearch, const LbfgsbArgs&, hipStream_t);
  int (*composite_eval)(const Mapping&, const SolveArgs&, hipStream_t);
  int (*outer)(const Mapping&, const AugLagOuterArgs&, hipStream_t);
  int (*fused)(mi355_lbfgs_ctx*, const Mapping&, int linesearch, const SolveArgs&, const AugLagOuterArgs&, hipStream_t);
  int (*fused_box)(mi355_lbfgs_ctx*, const Mapping&, int linesearch, const LbfgsbArgs&, const AugLagOuterArgs&,
                   hipStream_t);
};

// The fused halves of the library's own table live in their own translation unit (auglag_fused.hip).
int auglag_launch_fused(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const SolveArgs& args,
                        const AugLagOuterArgs& outer, hipStream_t stream);
int auglag_launch_fused_box(mi355_lbfgs_ctx* ctx, const Mapping& mp, int linesearch, const LbfgsbArgs& args,
                            const AugLagOuterArgs& outer, hipStream_t stream);

// Problems with constraint FAMILIES (mi355_al_problem::family_*): their own translation unit (auglag_family.hip) —
// AugLagObjective / AugLagOuterLoop with the family capacity of the mapping, closed term menu, Lbfgs + More-Thuente, the
// fused loop; and the composite evaluation behind mi355_auglag_eval_batch_host.
int auglag_launch_fused_family(mi355_lbfgs_ctx* ctx, const Mapping& mp, const SolveArgs& args, const AugLagOuterArgs& outer,
                               hipStream_t stream);
int auglag_launch_family_eval(const Mapping& mp, const SolveArgs& args, hipStream_t stream);

// A library built with user term functors registers ONE table for all of them (static initialisation of the generated
// unit): the kernels that evaluate term kinds `ids[0..count)` next to the closed menu.
// blob_ids: the ids (or -1) of the functors that take mi355_al_problem::user_params (kTermParamsFromProblem): a problem
// that names one of them without a blob is refused.
void register_user_al_terms(const AlLaunchers& launchers, const int* ids, int count, const int* blob_ids, int blob_count);
struct UserAlRegistration {
  UserAlRegistration(const AlLaunchers& launchers, std::initializer_list<int> ids, std::initializer_list<int> blob_ids = {}) {
    register_user_al_terms(launchers, ids.begin(), static_cast<int>(ids.size()), blob_ids.begin(),
                           static_cast<int>(blob_ids.size()));
  }
};

}  // namespace mi355
