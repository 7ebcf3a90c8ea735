// dispatch_ridge_gram_own.hip — the ridge objective with ONE MATRIX PER PROBLEM (objective id
// MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM): what a program computes that builds the README objective
// `SquaredError(A_b, y_b) + lambda * L2Reg` once per data set (README.md:126-160) and minimises each with Lbfgs.
// Normal-equation form per problem: a pre-pass (one workgroup per problem, matrix cores) writes G_b = A_b^T A_b + lambda I,
// c_b = A_b^T y_b and y_b . y_b into a per-problem row in HBM (P^2 + P + 2 doubles: 33 KB at n = 64, 8.7 GB for 262 144
// problems — sized for 288 GB), then the ordinary persistent Lbfgs kernel runs with an objective that streams ITS G_b on
// every evaluation.  A_b itself is read exactly once.  Fused arithmetic, More-Thuente, n <= 256, rows <= 4096.
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"
#include "ridge_gram.hpp"

namespace mi355 {
namespace {

template <int W, int E>
int launch_own(mi355_lbfgs_ctx* ctx, int m, const SolveArgs& args, hipStream_t stream) {
  using Obj = RidgeGramObjective<W, E, false, true>;
  constexpr int MT = MI355_LS_MORE_THUENTE;
#ifndef MI355_GRAM_OWN_MR0
  if constexpr (E == 2) {  // (as for the shared matrix: the y half of the history in registers where the budget allows)
    if (m <= 10) return launch_solve<W, E, Obj, 10, MT, kAlgLbfgs, NoOuterLoop, ArithFma>(ctx, args, stream);
  }
#endif
  return launch_solve<W, E, Obj, 0, MT, kAlgLbfgs, NoOuterLoop, ArithFma>(ctx, args, stream);
}

template <int W, int E>
int eval_own(const SolveArgs& args, hipStream_t stream) {
  using Obj = RidgeGramObjective<W, E, false, true>;
  constexpr int kSegs = kWave / W;
  const long long blocks_ll = (args.B + kSegs - 1) / kSegs;
  const int lds = kSegs * Obj::kLdsDoubles * static_cast<int>(sizeof(double));
  auto kern = eval_kernel<W, E, Obj, ArithFma>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWave), lds, stream, args);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

}  // namespace

// data_dev: [B][data_stride] on the device, row b = A_b (rows x n, row major) then y_b.
int ridge_gram_own_minimize(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, SolveArgs args, const double* data_dev,
                            int data_stride, hipStream_t stream, bool eval_only) {
  const int n = desc->n;
  const int rows = static_cast<int>(desc->objective_params[0]);
  const double lambda = desc->objective_params[1];
  if (n > kGramMaxCols || rows > kGramMaxRows)
    return fail(MI355_ERR_UNSUPPORTED, "the own-matrix ridge objective is built for n <= 256, rows <= 4096");
  if (!eval_only && desc->linesearch != MI355_LS_MORE_THUENTE)
    return fail(MI355_ERR_UNSUPPORTED, "the own-matrix ridge objective is built with the More-Thuente line search");
  if (desc->arithmetic == MI355_ARITH_EXACT)
    return fail(MI355_ERR_UNSUPPORTED, "the own-matrix ridge objective is a fused-arithmetic (normal-equation) form");
  if (desc->lanes_per_problem != 0 || desc->elems_per_lane != 0)
    return fail(MI355_ERR_INVALID_ARGUMENT, "the own-matrix ridge objective chooses its own mapping");
  if (desc->hessian_diagonal != nullptr || desc->hessian_from_functor)
    return fail(MI355_ERR_UNSUPPORTED, "the own-matrix ridge objective is First mode (every problem would need its own diagonal)");
  int P = 8;
  while (P < n) P <<= 1;
  const size_t row_doubles = static_cast<size_t>(P) * P + P + 2;
  const size_t need = static_cast<size_t>(args.B) * row_doubles;
  if (need > ctx->gram_rows_cap) {
    if (ctx->gram_rows_dev) HIP_TRY(hipFree(ctx->gram_rows_dev));
    ctx->gram_rows_dev = nullptr;
    ctx->gram_rows_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->gram_rows_dev), need * sizeof(double)));
    ctx->gram_rows_cap = need;
  }
  if (2 > ctx->gram_params_cap) {
    if (ctx->gram_params_dev) HIP_TRY(hipFree(ctx->gram_params_dev));
    ctx->gram_params_dev = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->gram_params_dev), 16 * sizeof(double)));
    ctx->gram_params_cap = 16;
  }
  ctx->gram_key.clear();   // (the shared-matrix cache of this context no longer describes gram_params_dev)
  ctx->gram_key_n = 0;
  struct Disarm {
    mi355_lbfgs_ctx* c;
    ~Disarm() { c->ev_start_armed = false; }
  } disarm{ctx};
  if (!eval_only) {
    HIP_TRY(hipEventRecord(ctx->ev_start, stream));
    ctx->ev_start_armed = true;
  }
  ctx->gram_host.assign(2, 0.0);
  ctx->gram_host[0] = rows;
  ctx->gram_host[1] = lambda;
  HIP_TRY(hipMemcpyAsync(ctx->gram_params_dev, ctx->gram_host.data(), 2 * sizeof(double), hipMemcpyHostToDevice, stream));
  int rc = ridge_gram_own_prepass(data_dev, data_stride, rows, n, P, lambda, args.B, ctx->gram_rows_dev, stream);
  if (rc != MI355_OK) return rc;
  args.obj_params = ctx->gram_params_dev;
  args.per_problem = ctx->gram_rows_dev;
  args.per_problem_stride = static_cast<int>(row_doubles);
  if (eval_only) {
    switch (P) {
      case 8: return eval_own<8, 1>(args, stream);
      case 16: return eval_own<8, 2>(args, stream);
      case 32: return eval_own<16, 2>(args, stream);
      case 64: return eval_own<32, 2>(args, stream);
      case 128: return eval_own<64, 2>(args, stream);
      case 256: return eval_own<64, 4>(args, stream);
    }
    return fail(MI355_ERR_INVALID_ARGUMENT, "mapping");
  }
  switch (P) {
    case 8: return launch_own<8, 1>(ctx, desc->m, args, stream);
    case 16: return launch_own<8, 2>(ctx, desc->m, args, stream);
    case 32: return launch_own<16, 2>(ctx, desc->m, args, stream);
    case 64: return launch_own<32, 2>(ctx, desc->m, args, stream);
    case 128: return launch_own<64, 2>(ctx, desc->m, args, stream);
    case 256: return launch_own<64, 4>(ctx, desc->m, args, stream);
  }
  return fail(MI355_ERR_INVALID_ARGUMENT, "mapping");
}

}  // namespace mi355
