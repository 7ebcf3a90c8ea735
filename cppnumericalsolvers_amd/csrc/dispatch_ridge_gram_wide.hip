// dispatch_ridge_gram_wide.hip — the normal-equation ridge objective above n = 64 (README.md:126-160 takes any A): a problem
// takes a whole wavefront, two coordinates per lane with G (128 KB) shared in LDS by the wavefronts of a workgroup up to
// n = 128, four coordinates per lane with G streamed through L2 up to n = 256.  History sizes up to 10 in the kernel that
// keeps the y half in registers (n <= 128), the others in the LDS-ring kernel.
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"
#include "ridge_gram.hpp"

namespace mi355 {

template <int W, int E>
static int launch_gram_wide(mi355_lbfgs_ctx* ctx, int m, const SolveArgs& args, hipStream_t stream) {
  using Obj = RidgeGramObjective<W, E>;
  constexpr int MT = MI355_LS_MORE_THUENTE;
  if constexpr (E <= 2) {  // (four coordinates per lane: ten y columns in registers + the objective's batch of rows of G
                           //  exceed 256 registers — 164 bytes of scratch per lane — so n > 128 takes the LDS ring)
    if (m <= 10) return launch_solve<W, E, Obj, 10, MT, kAlgLbfgs, NoOuterLoop, ArithFma>(ctx, args, stream);
  }
  return launch_solve<W, E, Obj, 0, MT, kAlgLbfgs, NoOuterLoop, ArithFma>(ctx, args, stream);
}

int ridge_gram_launch_wide(mi355_lbfgs_ctx* ctx, int P, int m, const SolveArgs& args, hipStream_t stream) {
  switch (P) {
    case 128: return launch_gram_wide<64, 2>(ctx, m, args, stream);
    case 256: return launch_gram_wide<64, 4>(ctx, m, args, stream);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "mapping");
}

}  // namespace mi355
