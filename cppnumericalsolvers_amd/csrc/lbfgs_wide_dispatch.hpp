// lbfgs_wide_dispatch.hpp — launcher of the workgroup kernel for n > MI355_LBFGS_MAX_N (lbfgs_wide_kernel.hpp), shared by
// the library's own unit (dispatch_wide.hip: Rosenbrock, DiagQuadratic) and the generated units of user objectives that
// bring a functor for this regime (`wide_type` in _build.build(user_objectives=...)).
#pragma once
#include "engine_internal.hpp"

#include "lbfgs_wide_kernel.hpp"

namespace mi355 {


template <class Obj, int E, int LS, int T = kWideThreads>
int launch_wide(mi355_lbfgs_ctx* ctx, WideArgs args, hipStream_t stream) {
  auto kern = lbfgs_wide_kernel<Obj, E, LS, T>;
  // memory form: the direction in LDS while four workgroups per CU still fit (32 KB each)
  int lds_max_n = 4096;
  if (const char* v = std::getenv("MI355_WIDE_LDS_MAX_N")) lds_max_n = std::atoi(v);   // A/B switch (0 = never)
  args.d_in_lds = (E == 0 && args.n <= lds_max_n) ? 1 : 0;
  const int lds = args.d_in_lds ? static_cast<int>(((static_cast<long long>(args.n) + 1) & ~1LL) * sizeof(double)) : 0;
  if (lds > 0)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, T, lds));
  if (per_cu < 1) per_cu = 1;
  if (T == kWideThreads && per_cu > 4) per_cu = 4;  // sixteen wavefronts per CU hide the memory latency; more only enlarge the workspace
  if (T == kWideThreadsBig) per_cu = 1;
  args.ws_stride = wide_ws_doubles(args.n, args.m, E);
  long long blocks = static_cast<long long>(per_cu) * ctx->num_cus;
  if (blocks > args.B) blocks = args.B;
  // the workspace is (5 + 2m) n doubles per RESIDENT workgroup: keep it under a quarter of the device memory
  if (ctx->device_total_bytes == 0) {   // (asked once per context: hipMemGetInfo synchronises)
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    ctx->device_total_bytes = total_b;
  }
  const long long budget = static_cast<long long>(ctx->device_total_bytes / 4);
  const long long per_block = args.ws_stride * static_cast<long long>(sizeof(double));
  if (per_block > budget) return fail(MI355_ERR_HIP, "n too large for the workspace of one problem");
  if (blocks * per_block > budget) blocks = budget / per_block;
  const size_t need = static_cast<size_t>(blocks) * static_cast<size_t>(per_block);
  if (need > ctx->wide_ws_cap) {
    if (ctx->wide_ws) {
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(hipFree(ctx->wide_ws));
    }
    ctx->wide_ws = nullptr;
    ctx->wide_ws_cap = 0;
    HIP_TRY(hipMalloc(&ctx->wide_ws, need));
    ctx->wide_ws_cap = need;
  }
  args.workspace = static_cast<double*>(ctx->wide_ws);
  args.next_problem = ctx->queue_dev;
  HIP_TRY(hipMemsetAsync(ctx->queue_dev, 0, kQueueWords * sizeof(unsigned long long), stream));
  HIP_TRY(hipEventRecord(ctx->ev_start, stream));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks)), dim3(T), lds, stream, args);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev_stop, stream));
  ctx->timed = true;
  ctx->last_W = T;
  ctx->last_E = E;   // coordinates per thread held in registers (0: the vectors live in the workspace)
  ctx->last_blocks = static_cast<int>(blocks);
  ctx->last_threads = T;
  ctx->last_lds = lds;
  ctx->last_mr = 0;
  ctx->last_arith = MI355_ARITH_EXACT;
  return MI355_OK;
}


// The storage form by dimension.
template <class Obj>
int dispatch_wide_objective(mi355_lbfgs_ctx* ctx, const WideArgs& args, hipStream_t stream) {
  // Three forms, one sequence of operations (scripts/wide_bench.py, Rosenbrock, m = 10, 100 iterations, one box):
  //   n <= 512          vectors in registers, two coordinates per thread        15.5 ms vs 18.4 (memory + LDS form)
  //   512 < n <= 4096   vectors in the workspace, the direction in LDS          n = 1024: 31.0 vs 34.3 (registers, E = 4);
  //                                                                             2048: 41.3 vs 55.1, 4096: 82.8 vs 109.5 (plain)
  //   n > 4096          everything in the workspace                             (64 KB of LDS at n = 8192 halves the
  //                                                                              resident workgroups: 133 vs 104 ms)
  // Eight / sixteen coordinates per thread in registers LOSE (256 registers + scratch, one workgroup per CU: 55.8 vs 51.1
  // ms at n = 2048, 171.6 vs 109.9 at 4096): the kernel needs the parallelism more than it needs the traffic.
  const char* force = std::getenv("MI355_WIDE_IN_MEMORY");   // A/B switch: the memory-resident form at every n
  const int n = (force && force[0] == '1') ? (1 << 30) : args.n;
  if (args.hess_from_functor && !HasWideHessDiag<Obj>::value)
    return fail(MI355_ERR_UNSUPPORTED, "hessian_from_functor: this objective's workgroup functor has no hess_diag");
  // n >= kWideBigN: sixteen wavefronts per problem (the summation order has 1024 lanes there: a function of n alone)
  const bool big = args.n >= kWideBigN;
  if (args.linesearch == MI355_LS_HAGER_ZHANG) {   // Lbfgs<F, m, HagerZhang>
    // (no sixteen-wavefront form: at the 128 registers a 1024-thread workgroup leaves per thread the Hager-Zhang state
    //  machine spilled 208-216 bytes per lane; the four-wavefront kernel serves every n, its sums run over 256 lanes)
    if (n <= 512) return launch_wide<Obj, 2, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
    return launch_wide<Obj, 0, MI355_LS_HAGER_ZHANG>(ctx, args, stream);
  }
  if (big) return launch_wide<Obj, 0, MI355_LS_MORE_THUENTE, kWideThreadsBig>(ctx, args, stream);
  if (n <= 512) return launch_wide<Obj, 2, MI355_LS_MORE_THUENTE>(ctx, args, stream);
  return launch_wide<Obj, 0, MI355_LS_MORE_THUENTE>(ctx, args, stream);
}

// A user objective's functor for this regime, registered by its generated unit.
using UserWideFn = int (*)(mi355_lbfgs_ctx* ctx, const WideArgs& args, hipStream_t stream);
void register_user_wide(int objective_id, UserWideFn fn);
struct UserWideRegistration {
  UserWideRegistration(int objective_id, UserWideFn fn) { register_user_wide(objective_id, fn); }
};

}  // namespace mi355
