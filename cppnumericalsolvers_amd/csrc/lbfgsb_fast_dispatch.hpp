// lbfgsb_fast_dispatch.hpp — launch templates of the relaxed-algebra L-BFGS-B kernels (lbfgsb_fast_kernel.hpp).
// Included by dispatch_lbfgsb_fast.hip and by the generated 16-lane unit of a user objective (_build.py), so that the
// other translation units of the library do not depend on the kernel's source.
#pragma once
#define MI355_DISPATCH_TU 1
#include "engine_internal.hpp"
#include "lbfgsb_fast_kernel.hpp"

namespace mi355 {

template <int E, class Obj, int M>
int launch_lbfgsb_fast(mi355_lbfgs_ctx* ctx, LbfgsbArgs args, hipStream_t stream) {
  constexpr int W = lbfgsb_fast_lanes(M), kSegs = kWave / W;
  const int lds = (Obj::shared_lds_doubles() + kSegs * lbfgsb_fast_lds_doubles_per_problem<M>(W * E, Obj::kLdsDoubles) +
                   lbfgsb_fast_shared_tail_doubles(W * E)) * static_cast<int>(sizeof(double));
  if (lds > 160 * 1024) return fail(MI355_ERR_INVALID_ARGUMENT, "history / objective data do not fit LDS");
  auto kern = lbfgsb_fast_kernel<E, Obj, M>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kWave, lds));
  if (per_cu < 1) per_cu = 1;
  const long long blocks_needed = (args.s.B + kSegs - 1) / kSegs;
  long long blocks_ll = static_cast<long long>(per_cu) * ctx->num_cus;
  if (blocks_ll > blocks_needed) blocks_ll = blocks_needed;
  args.s.next_problem = ctx->queue_dev;
#ifdef MI355_LBFGSB_PHASE_TIMING
  HIP_TRY(profile_counters(ctx, stream, &args.s.profile));
#endif
  HIP_TRY(hipMemsetAsync(ctx->queue_dev, 0, kQueueWords * sizeof(unsigned long long), stream));
  HIP_TRY(hipEventRecord(ctx->ev_start, stream));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWave), lds, stream, args);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev_stop, stream));
  ctx->timed = true;
  ctx->last_W = W;
  ctx->last_E = E;
  ctx->last_blocks = static_cast<int>(blocks_ll);
  ctx->last_threads = kWave;
  ctx->last_lds = lds;
  ctx->last_mr = 0;
  ctx->last_arith = MI355_ARITH_FMA;
  return MI355_OK;
}

// m = 9, 10: capacity 10 on thirty-two lanes per problem (n <= 64: one or two coordinates per lane)
template <int E, class Obj>
int launch_lbfgsb_fast_w32(mi355_lbfgs_ctx* ctx, const LbfgsbArgs& args, hipStream_t stream) {
  if constexpr (!HasFusedEval<Obj>::value) {
    return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
  } else {
    return launch_lbfgsb_fast<E, Obj, 10>(ctx, args, stream);
  }
}

// the relaxed-algebra kernels of one objective on sixteen lanes: capacity 5 (m <= 5) and, up to four coordinates per
// lane, 8 (m = 6..8)
template <int E, class Obj>
int dispatch_lbfgsb_fast_m(mi355_lbfgs_ctx* ctx, const LbfgsbArgs& args, hipStream_t stream) {
  if constexpr (!HasFusedEval<Obj>::value) {
    return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
  } else {
    if (args.s.m <= 5) return launch_lbfgsb_fast<E, Obj, 5>(ctx, args, stream);
    if constexpr (E <= 4) {
      if (args.s.m <= 8) return launch_lbfgsb_fast<E, Obj, 8>(ctx, args, stream);
    }
    return fail(MI355_ERR_UNSUPPORTED, "relaxed-algebra L-BFGS-B is built for m <= 10 (n <= 64) / m <= 5 (n <= 128)");
  }
}

// One relaxed-algebra kernel of a user objective (generated units, _build.py): refused when the functor has no eval_fma
template <int E, class Obj, int M>
int launch_lbfgsb_fast_user(mi355_lbfgs_ctx* ctx, const LbfgsbArgs& args, hipStream_t stream) {
  if constexpr (!HasFusedEval<Obj>::value) {
    return fail(MI355_ERR_UNSUPPORTED, "this objective has no fused-arithmetic form (MI355_ARITH_FMA)");
  } else {
    return launch_lbfgsb_fast<E, Obj, M>(ctx, args, stream);
  }
}

}  // namespace mi355
