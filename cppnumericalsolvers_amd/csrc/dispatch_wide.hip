// dispatch_wide.hip — Lbfgs for n > MI355_LBFGS_MAX_N: one problem per workgroup, state in an HBM workspace
// (lbfgs_wide_kernel.hpp, lbfgs_wide_dispatch.hpp): the built-in objectives and the table of user functors.
#define MI355_DISPATCH_TU 1
#include "lbfgs_wide_dispatch.hpp"

#include <utility>
#include <vector>

namespace mi355 {
namespace {
std::vector<std::pair<int, UserWideFn>>& user_wide_table() {
  static std::vector<std::pair<int, UserWideFn>> table;
  return table;
}
}  // namespace

void register_user_wide(int objective_id, UserWideFn fn) {
  if (objective_id < MI355_OBJ_USER_FIRST) return;
  for (auto& e : user_wide_table())
    if (e.first == objective_id) {
      e.second = fn;
      return;
    }
  user_wide_table().emplace_back(objective_id, fn);
}

int dispatch_wide(mi355_lbfgs_ctx* ctx, int objective, const WideArgs& args, hipStream_t stream) {
  if (args.m > kWideMaxM) return fail(MI355_ERR_INVALID_ARGUMENT, "m out of range");
  switch (objective) {
    case MI355_OBJ_ROSENBROCK: return dispatch_wide_objective<RosenbrockWide>(ctx, args, stream);
    case MI355_OBJ_DIAG_QUADRATIC: return dispatch_wide_objective<DiagQuadraticWide>(ctx, args, stream);
  }
  for (auto& e : user_wide_table())
    if (e.first == objective) return e.second(ctx, args, stream);
  return fail(MI355_ERR_UNSUPPORTED,
              "n > MI355_LBFGS_MAX_N is built for the Rosenbrock and DiagQuadratic objectives and for user objectives "
              "compiled in with a functor for this regime (wide_type)");
}

}  // namespace mi355
