// dispatch_ridge_gram.hip — the ridge objective in normal-equation form (ridge_gram.hpp): Gram matrix and c_b = A^T y_b
// on the matrix cores, then the ordinary persistent Lbfgs kernel on the n x n quadratic.  Shapes up to n = 64 here; the
// solve kernels of 64 < n <= 256 compile in dispatch_ridge_gram_wide.hip.
#define MI355_DISPATCH_TU 1
#define MI355_RIDGE_GRAM_PREPASS_TU 1
#include <cmath>

#include "engine_internal.hpp"
#include "ridge_gram.hpp"

namespace mi355 {
namespace {

template <int W, int E>
int launch_gram(mi355_lbfgs_ctx* ctx, int m, const SolveArgs& args, hipStream_t stream) {
  using Obj = RidgeGramObjective<W, E>;
  using NO = NoOuterLoop;
  constexpr int MT = MI355_LS_MORE_THUENTE;
  if constexpr (E >= 2) {
    if (m <= 5) return launch_solve<W, E, Obj, 5, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
    if (m == 6) return launch_solve<W, E, Obj, 6, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
    if (m <= 10) return launch_solve<W, E, Obj, 10, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
  }
  return launch_solve<W, E, Obj, 0, MT, kAlgLbfgs, NO, ArithFma>(ctx, args, stream);
}

// one evaluation per problem (the library's launch_eval would also instantiate the stand-alone Hager-Zhang search,
// which needs the exact-order eval this functor does not have)
template <int W, int E>
int eval_gram(const SolveArgs& args, hipStream_t stream) {
  using Obj = RidgeGramObjective<W, E>;
  constexpr int kSegs = kWave / W;
  const long long blocks_ll = (args.B + kSegs - 1) / kSegs;
  const int lds = (Obj::shared_lds_doubles() + kSegs * Obj::kLdsDoubles) * static_cast<int>(sizeof(double));
  auto kern = eval_kernel<W, E, Obj, ArithFma>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWave), lds, stream, args);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

}  // namespace

// The own-matrix pre-pass (its kernel lives in this unit with the other two matrix-core kernels)
int ridge_gram_own_prepass(const double* data, long long data_stride, int rows, int n, int P, double lambda, long long B,
                           double* out, hipStream_t stream) {
  hipLaunchKernelGGL(ridge_gram_own_prepass_kernel, dim3(static_cast<unsigned>(B)), dim3(256), 0, stream, data, data_stride,
                     rows, n, P, lambda, B, out);
  HIP_TRY(hipGetLastError());
  return MI355_OK;
}

// args: everything but obj_params / per_problem (filled here).  y_dev: [B][y_stride] on the device.
// eval_only: one evaluation per problem at args.x0 (mi355_lbfgs_eval_batch) instead of a solve.
int ridge_gram_minimize(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, SolveArgs args, const double* y_dev,
                        int y_stride, hipStream_t stream, bool eval_only) {
  const int n = desc->n;
  const int rows = static_cast<int>(desc->objective_params[0]);
  const double lambda = desc->objective_params[1];
  const double* A = desc->objective_params + 2;
  if (n > kGramMaxCols || rows > kGramMaxRows)
    return fail(MI355_ERR_UNSUPPORTED, "the normal-equation ridge objective is built for n <= 256, rows <= 4096");
  if (!eval_only && desc->linesearch != MI355_LS_MORE_THUENTE)
    return fail(MI355_ERR_UNSUPPORTED, "the normal-equation ridge objective is built with the More-Thuente line search");
  if (desc->arithmetic == MI355_ARITH_EXACT)
    return fail(MI355_ERR_UNSUPPORTED, "the normal-equation ridge objective is a fused-arithmetic form (use "
                                       "MI355_OBJ_SQUARED_ERROR_RIDGE for the reference's operation order)");
  if (desc->lanes_per_problem != 0 || desc->elems_per_lane != 0)
    return fail(MI355_ERR_INVALID_ARGUMENT, "the normal-equation ridge objective chooses its own mapping");
  // Mapping: TWO coordinates per lane (one for n <= 8, four for n > 128).  The matrix-vector loop wants a batch of rows
  // of G in flight (ridge_gram.hpp), i.e. registers the four-coordinates-per-lane kernels do not have at m = 10 (255 VGPRs
  // before the objective); with two the kernel sits at ~220, eight wavefronts per CU next to the 32 KB of G at n <= 64.
  int P = 8;
  while (P < n) P <<= 1;
  const int E = (P == 8) ? 1 : ((P == 256) ? 4 : 2);
  const int AC = gram_a_cols(P), rows4 = (rows + 3) & ~3;
  // ---- shared parameters: rows, lambda, G[P][P], A padded to [rows4][AC]; rebuilt only when A / lambda / n change or the
  // solves move to another stream (the cached blob is ordered on the stream it was built on) ----
  const size_t key_len = 2 + static_cast<size_t>(rows) * n;
  const size_t blob = 2 + static_cast<size_t>(P) * P + static_cast<size_t>(rows4) * AC;
  const bool same = ctx->gram_key_n == n && ctx->gram_key.size() == key_len && ctx->gram_stream == stream &&
                    std::memcmp(ctx->gram_key.data(), desc->objective_params, key_len * sizeof(double)) == 0 &&
                    ctx->gram_params_dev != nullptr;
  // timing: the pre-pass (and, on a new matrix, the Gram kernel) belong to the launch — launch_solve keeps this start
  struct Disarm {   // (whatever path leaves this function, the next launch records its own start)
    mi355_lbfgs_ctx* c;
    ~Disarm() { c->ev_start_armed = false; }
  } disarm{ctx};
  if (!eval_only) {
    HIP_TRY(hipEventRecord(ctx->ev_start, stream));
    ctx->ev_start_armed = true;
  }
  if (!same) {
    HIP_TRY(wait_for_last_solve(ctx, ctx->gram_stream, stream));
    std::vector<double>& h = ctx->gram_host;
    h.assign(blob, 0.0);
    h[0] = rows;
    h[1] = lambda;
    double* Apad = h.data() + 2 + static_cast<size_t>(P) * P;
    for (int r = 0; r < rows; ++r)
      for (int j = 0; j < n; ++j) Apad[static_cast<size_t>(r) * AC + j] = A[static_cast<size_t>(r) * n + j];
    if (blob > ctx->gram_params_cap) {
      if (ctx->gram_params_dev) HIP_TRY(hipFree(ctx->gram_params_dev));
      ctx->gram_params_dev = nullptr;
      ctx->gram_params_cap = 0;
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->gram_params_dev), blob * sizeof(double)));
      ctx->gram_params_cap = blob;
    }
    HIP_TRY(hipMemcpyAsync(ctx->gram_params_dev, h.data(), blob * sizeof(double), hipMemcpyHostToDevice, stream));
    // G = A^T A + lambda I on the matrix cores (one wavefront per 16 x 16 tile), written over the zeroed G of the blob
    const int tiles = AC / 16;
    hipLaunchKernelGGL(ridge_gram_matrix_kernel, dim3(static_cast<unsigned>(tiles * tiles)), dim3(64), 0, stream,
                       ctx->gram_params_dev + 2 + static_cast<size_t>(P) * P, rows4, AC, n, P, lambda,
                       ctx->gram_params_dev + 2);
    HIP_TRY(hipGetLastError());
    ctx->gram_key.assign(desc->objective_params, desc->objective_params + key_len);
    ctx->gram_key_n = n;
    ctx->gram_stream = stream;
  }
  // ---- per-problem rows (c_b, yy_b): the batched GEMM on the matrix cores ----
  const size_t need = static_cast<size_t>(args.B) * (P + 2);
  if (need > ctx->gram_rows_cap) {
    if (ctx->gram_rows_dev) HIP_TRY(hipFree(ctx->gram_rows_dev));
    ctx->gram_rows_dev = nullptr;
    ctx->gram_rows_cap = 0;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->gram_rows_dev), need * sizeof(double)));
    ctx->gram_rows_cap = need;
  }
  const double* a_pad_dev = ctx->gram_params_dev + 2 + static_cast<size_t>(P) * P;
  const unsigned blocks = static_cast<unsigned>((args.B + 63) / 64);
  hipLaunchKernelGGL(ridge_gram_prepass_kernel, dim3(blocks), dim3(256), 0, stream, a_pad_dev, y_dev, y_stride, rows,
                     rows4, AC, n, P, static_cast<long long>(args.B), ctx->gram_rows_dev);
  HIP_TRY(hipGetLastError());
  args.obj_params = ctx->gram_params_dev;
  args.per_problem = ctx->gram_rows_dev;
  args.per_problem_stride = P + 2;
  if (eval_only) {
    switch (P) {
      case 8: return eval_gram<8, 1>(args, stream);
      case 16: return eval_gram<8, 2>(args, stream);
      case 32: return eval_gram<16, 2>(args, stream);
      case 64: return eval_gram<32, 2>(args, stream);
      case 128: return eval_gram<64, 2>(args, stream);
      case 256: return eval_gram<64, 4>(args, stream);
    }
    return fail(MI355_ERR_INVALID_ARGUMENT, "mapping");
  }
  int rc = MI355_ERR_INVALID_ARGUMENT;
  switch (P) {
    case 8: rc = launch_gram<8, 1>(ctx, desc->m, args, stream); break;
    case 16: rc = launch_gram<8, 2>(ctx, desc->m, args, stream); break;
    case 32: rc = launch_gram<16, 2>(ctx, desc->m, args, stream); break;
    case 64: rc = launch_gram<32, 2>(ctx, desc->m, args, stream); break;
    default: rc = ridge_gram_launch_wide(ctx, P, desc->m, args, stream); break;   // dispatch_ridge_gram_wide.hip
  }
  return rc;
}

}  // namespace mi355
