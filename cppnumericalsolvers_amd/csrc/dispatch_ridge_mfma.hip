// dispatch_ridge_mfma.hip — the joint-evaluation ridge kernel (see ridge_mfma_kernel.hpp, engine_internal.hpp).
#include "engine_internal.hpp"
#include "ridge_mfma_kernel.hpp"

namespace mi355 {

template <int W, int E, class AR>
int launch_ridge_mfma_mapping(mi355_lbfgs_ctx* ctx, SolveArgs args, hipStream_t stream) {
  constexpr int MR = 10;
  constexpr int kWaves = kJointSlots / (kWave / W);
  const int lds = ridge_mfma_lds_doubles(MR, kWaves, /*alpha_in_lds=*/W == 32) * static_cast<int>(sizeof(double));
  auto kern = ridge_mfma_solve_kernel<MR, W, E, AR>;
  HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int per_cu = 0;
  HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, kWaves * kWave, lds));
  if (per_cu < 1) per_cu = 1;
  const long long blocks_needed = (args.B + kJointSlots - 1) / kJointSlots;
  long long blocks_ll = static_cast<long long>(per_cu) * ctx->num_cus;
  if (blocks_ll > blocks_needed) blocks_ll = blocks_needed;
  // plateau rings: MAX_PAST doubles per resident problem slot
  const size_t need = static_cast<size_t>(blocks_ll) * kJointSlots * MI355_LBFGS_MAX_PAST;
  if (need > ctx->scratch_cap)  // (sized in mi355_lbfgs_create for the fullest resident grid)
    return fail(MI355_ERR_INVALID_ARGUMENT, "resident grid larger than the context's plateau-ring scratch");
  args.scratch = ctx->scratch_dev;
  args.next_problem = ctx->queue_dev;
#ifdef MI355_LBFGS_PHASE_TIMING
  HIP_TRY(profile_counters(ctx, stream, &args.profile));
#endif
  HIP_TRY(hipMemsetAsync(ctx->queue_dev, 0, kQueueWords * sizeof(unsigned long long), stream));
  HIP_TRY(hipEventRecord(ctx->ev_start, stream));
  hipLaunchKernelGGL(kern, dim3(static_cast<unsigned>(blocks_ll)), dim3(kWaves * kWave), lds, stream, args);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(ctx->ev_stop, stream));
  ctx->timed = true;
  ctx->last_W = W;
  ctx->last_E = E;
  ctx->last_blocks = static_cast<int>(blocks_ll);
  ctx->last_threads = kWaves * kWave;
  ctx->last_lds = lds;
  ctx->last_mr = MR;
  ctx->last_arith = AR::kFma ? MI355_ARITH_FMA : MI355_ARITH_EXACT;
  return MI355_OK;
}

// lanes: 32 (eight wavefronts x two problems, the default) or 16 (four wavefronts x four problems: see
// ridge_mfma_kernel.hpp and profiles/r2_ab_ridge_mapping.txt)
int launch_ridge_mfma(mi355_lbfgs_ctx* ctx, SolveArgs args, hipStream_t stream, int lanes, bool fma) {
  if (fma)
    return lanes == 16 ? launch_ridge_mfma_mapping<16, 4, ArithFma>(ctx, args, stream)
                       : launch_ridge_mfma_mapping<32, 2, ArithFma>(ctx, args, stream);
  return lanes == 16 ? launch_ridge_mfma_mapping<16, 4, ArithExact>(ctx, args, stream)
                     : launch_ridge_mfma_mapping<32, 2, ArithExact>(ctx, args, stream);
}

}  // namespace mi355
