// host_pipeline.hip — everything between HOST arrays and the device-pointer entry points:
//
//   * the per-iteration trace set-up (mi355_lbfgs_trace) shared by the solve entry points;
//   * the host-pointer pipeline: pinned staging + persistent device buffers owned by the context (two slots, grow
//     only — no hipMalloc / hipFree per call), multi-threaded pageable <-> pinned copies, asynchronous H2D / solve /
//     D2H on three streams, and for batches larger than a staging slot a chunked loop in which chunk c + 1 is staged
//     and solved while chunk c travels back;
//   * device groups: one context per device, one host thread per device, contiguous shards of the batch, and the
//     ONE collective of the path — an RCCL all-reduce (over xGMI between GPUs) of the 3-word convergence record
//     [problems, unconverged, iterations].  librccl is loaded at run time (dlopen), so the
//     single-GPU library has no link-time dependency on it.
//
// Replaces nothing in the reference (which is single-problem, single-thread, host only): this is the plumbing a
// batched drop-in needs so that `Lbfgs<F>::MinimizeBatch(function, states)` on host vectors gets the device path.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <numeric>
#include <thread>

#include "engine_internal.hpp"

using namespace mi355;

// the device-pointer entry points this file drives (mi355_lbfgs.hip)
int mi355_minimize_batch_device(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out,
                                void* stream, int solver);

namespace mi355 {

// ---------------------------------------------------------------------------------------------------
// trace set-up (device array pointers in desc->trace)
// ---------------------------------------------------------------------------------------------------
int setup_trace(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, hipStream_t stream, SolveArgs& args) {
  args.trace = nullptr;
  const mi355_lbfgs_trace* t = desc->trace;
  if (t == nullptr) return MI355_OK;
  if (t->count < 1 || t->count > MI355_LBFGS_MAX_TRACED)
    return fail(MI355_ERR_INVALID_ARGUMENT, "trace.count out of range [1, MI355_LBFGS_MAX_TRACED]");
  if (t->capacity < 1) return fail(MI355_ERR_INVALID_ARGUMENT, "trace.capacity must be >= 1");
  if (!t->problems || !t->records || !t->written)
    return fail(MI355_ERR_INVALID_ARGUMENT, "trace.problems / records / written must not be NULL");
  for (int i = 0; i < t->count; ++i) {
    if (t->problems[i] < 0 || t->problems[i] >= B)
      return fail(MI355_ERR_INVALID_ARGUMENT, "trace.problems holds an index outside the batch");
    for (int k = 0; k < i; ++k)
      if (t->problems[k] == t->problems[i]) return fail(MI355_ERR_INVALID_ARGUMENT, "trace.problems holds a duplicate");
    ctx->trace_host.problems[i] = t->problems[i];
  }
  ctx->trace_host.records = t->records;
  ctx->trace_host.x = t->x;
  ctx->trace_host.g = t->g;
  ctx->trace_host.written = t->written;
  ctx->trace_host.count = t->count;
  ctx->trace_host.capacity = t->capacity;
  if (!ctx->trace_dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&ctx->trace_dev), sizeof(TraceArgs)));
  // (a pageable source: the copy is staged before the call returns, so trace_host may be reused by the next call)
  HIP_TRY(hipMemcpyAsync(ctx->trace_dev, &ctx->trace_host, sizeof(TraceArgs), hipMemcpyHostToDevice, stream));
  HIP_TRY(hipMemsetAsync(t->written, 0, t->count * sizeof(unsigned), stream));
  args.trace = ctx->trace_dev;
  return MI355_OK;
}

void destroy_host_pipeline(mi355_lbfgs_ctx* ctx) {
  for (auto& st : ctx->stage) {
    if (st.pinned) (void)hipHostFree(st.pinned);
    if (st.device) (void)hipFree(st.device);
    if (st.in_ready) (void)hipEventDestroy(st.in_ready);
    if (st.solved) (void)hipEventDestroy(st.solved);
    if (st.out_ready) (void)hipEventDestroy(st.out_ready);
    for (hipEvent_t ev : st.piece_ready)
      if (ev) (void)hipEventDestroy(ev);
    st = mi355_lbfgs_ctx::HostStage();
  }
  if (ctx->stream_in) (void)hipStreamDestroy(ctx->stream_in);
  if (ctx->stream_solve) (void)hipStreamDestroy(ctx->stream_solve);
  if (ctx->stream_out) (void)hipStreamDestroy(ctx->stream_out);
  ctx->stream_in = ctx->stream_solve = ctx->stream_out = nullptr;
  if (ctx->trace_dev) (void)hipFree(ctx->trace_dev);
  if (ctx->flags_dev) (void)hipFree(ctx->flags_dev);
  ctx->trace_dev = nullptr;
  ctx->flags_dev = nullptr;
}

}  // namespace mi355

namespace {

constexpr size_t kStageBytesDefault = 256u << 20;  // per slot: larger batches are solved in chunks
// (MI355_HOST_STAGE_BYTES overrides the slot size: the tests use it to drive small batches through the chunked loop)
size_t stage_bytes_max() {
  const char* v = std::getenv("MI355_HOST_STAGE_BYTES");
  if (v && *v) {
    const long long b = std::atoll(v);
    if (b >= (64 << 10)) return static_cast<size_t>(b);
  }
  return kStageBytesDefault;
}
constexpr int kCopyThreads = 4;

// pageable <-> pinned copies on a few host threads (one memcpy stream per thread saturates a fraction of the
// host's memory bandwidth; four are enough to keep up with PCIe gen 5)
void parallel_memcpy(void* dst, const void* src, size_t bytes) {
  if (bytes < (4u << 20)) {
    std::memcpy(dst, src, bytes);
    return;
  }
  std::thread th[kCopyThreads - 1];
  const size_t part = (bytes / kCopyThreads + 63) & ~static_cast<size_t>(63);
  for (int t = 0; t < kCopyThreads; ++t) {
    const size_t off = std::min(bytes, part * t), len = std::min(bytes - off, part);
    auto job = [=]() {
      if (len) std::memcpy(static_cast<char*>(dst) + off, static_cast<const char*>(src) + off, len);
    };
    if (t + 1 < kCopyThreads) {
      th[t] = std::thread(job);
    } else {
      job();
    }
  }
  for (auto& t : th) t.join();
}

struct Layout {  // one chunk of C problems inside a staging slot
  size_t x0, pp, in_bytes;           // inputs
  size_t x, g, f, p, out_bytes;      // outputs (offsets from the start of the output part)
  size_t total;
};
Layout layout(int64_t C, int n, int pp_stride, bool want_g, bool want_p) {
  auto al = [](size_t v) { return (v + 255) & ~static_cast<size_t>(255); };
  Layout L;
  const size_t c = static_cast<size_t>(C);
  L.x0 = 0;
  L.pp = al(c * n * sizeof(double));
  L.in_bytes = al(L.pp + c * pp_stride * sizeof(double));
  L.x = 0;
  L.g = al(c * n * sizeof(double));
  L.f = L.g + (want_g ? al(c * n * sizeof(double)) : 0);
  L.p = L.f + al(c * sizeof(double));
  L.out_bytes = L.p + (want_p ? al(c * sizeof(mi355_lbfgs_progress)) : 0);
  L.total = L.in_bytes + L.out_bytes;
  return L;
}

int ensure_stage(mi355_lbfgs_ctx* ctx, int slot, size_t bytes) {
  auto& st = ctx->stage[slot];
  if (!ctx->stream_in) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_in, hipStreamNonBlocking));
  if (!ctx->stream_solve) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_solve, hipStreamNonBlocking));
  if (!ctx->stream_out) HIP_TRY(hipStreamCreateWithFlags(&ctx->stream_out, hipStreamNonBlocking));
  if (!st.in_ready) {
    HIP_TRY(hipEventCreateWithFlags(&st.in_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&st.solved, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&st.out_ready, hipEventDisableTiming));
    for (hipEvent_t& ev : st.piece_ready) HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  }
  if (bytes <= st.cap) return MI355_OK;
  if (st.pinned) HIP_TRY(hipHostFree(st.pinned));
  if (st.device) HIP_TRY(hipFree(st.device));
  st.pinned = st.device = nullptr;
  st.cap = 0;
  const size_t want = bytes + bytes / 8;  // a little head-room: batches of one application tend to be similar
  HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&st.pinned), want, hipHostMallocDefault));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&st.device), want));
  st.cap = want;
  return MI355_OK;
}

using DeviceSolve = std::function<int(const mi355_lbfgs_desc*, int64_t, const double*, double*, double*, double*,
                                      mi355_lbfgs_progress*, hipStream_t)>;

// host arrays in, host arrays out, through the context's staging slots
int run_host_batch(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0, double* x_out,
                   double* f_out, double* g_out, mi355_lbfgs_progress* progress_out, const DeviceSolve& solve) {
  const int n = desc->n;
  const int pps = desc->per_problem_data ? desc->per_problem_stride : 0;
  const bool want_g = g_out != nullptr, want_p = progress_out != nullptr;
  const size_t per_problem_bytes = layout(1024, n, pps, want_g, want_p).total / 1024 + 1;
  // problems per staging slot: what the byte budget holds — a multiple of 64 where it holds that many, fewer (down to
  // one) for the long vectors of the workgroup kernel (n up to 2^24: a row of x0 + x + g alone is 384 MB there), and a
  // clear error when a single problem exceeds the budget
  int64_t chunk = static_cast<int64_t>(stage_bytes_max() / per_problem_bytes);
  if (chunk < 1)
    return fail(MI355_ERR_INVALID_ARGUMENT,
                "host-pointer entry point: one problem of this dimension needs more staging bytes than "
                "the MI355_HOST_STAGE_BYTES budget allows; raise it or use the device-pointer entry point");
  if (chunk >= 64) chunk &= ~int64_t(63);
  if (chunk > B) chunk = B;
  const int64_t chunks = (B + chunk - 1) / chunk;
  const size_t slot_bytes = layout(chunk, n, pps, want_g, want_p).total;
  int rc = ensure_stage(ctx, 0, slot_bytes);
  if (rc == MI355_OK && chunks > 1) rc = ensure_stage(ctx, 1, slot_bytes);
  if (rc != MI355_OK) return rc;

  // ---- trace with HOST arrays: device copies in index order, so that every chunk's traced problems are a
  // contiguous range of rows; results go back to the caller's order at the end
  const mi355_lbfgs_trace* ht = desc->trace;
  std::vector<int> order;
  char* trace_dev = nullptr;
  size_t t_rec = 0, t_x = 0, t_g = 0, t_w = 0;
  if (ht) {
    if (ht->count < 1 || ht->count > MI355_LBFGS_MAX_TRACED || ht->capacity < 1 || !ht->problems || !ht->records ||
        !ht->written)
      return fail(MI355_ERR_INVALID_ARGUMENT, "trace: count in [1, MI355_LBFGS_MAX_TRACED], capacity >= 1, problems / "
                                              "records / written not NULL");
    order.resize(ht->count);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return ht->problems[a] < ht->problems[b]; });
    for (int i = 0; i < ht->count; ++i) {
      if (ht->problems[i] < 0 || ht->problems[i] >= B)
        return fail(MI355_ERR_INVALID_ARGUMENT, "trace.problems holds an index outside the batch");
      if (i > 0 && ht->problems[order[i]] == ht->problems[order[i - 1]])
        return fail(MI355_ERR_INVALID_ARGUMENT, "trace.problems holds a duplicate");
    }
    const size_t rows = static_cast<size_t>(ht->count) * ht->capacity;
    t_rec = 0;
    t_x = rows * sizeof(mi355_lbfgs_trace_record);
    t_g = t_x + (ht->x ? rows * n * sizeof(double) : 0);
    t_w = t_g + (ht->g ? rows * n * sizeof(double) : 0);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&trace_dev), t_w + ht->count * sizeof(unsigned)));
  }
  auto cleanup = [&](int code) {
    if (trace_dev) (void)hipFree(trace_dev);
    return code;
  };

  hipError_t e = hipSuccess;
  // The rows of a chunk cross PCIe in pieces, so that the host copy of piece k (pageable <-> pinned, which also pays
  // the first-touch page faults of freshly allocated caller arrays) overlaps the DMA of piece k + 1.
  const char* pieces_env = std::getenv("MI355_HOST_PIECES");  // (tuning / A-B runs: force the number of pieces)
  const int pieces_forced = (pieces_env && *pieces_env) ? std::atoi(pieces_env) : 0;
  auto pieces_of = [&](int64_t bc) {
    if (pieces_forced >= 1) return std::min(8, pieces_forced);
    const size_t row_bytes = static_cast<size_t>(bc) * n * sizeof(double);
    int k = static_cast<int>(row_bytes / (6u << 20));
    return std::max(1, std::min(8, k));
  };
  auto piece_range = [](int64_t bc, int pieces, int k, int64_t& r0, int64_t& r1) {
    r0 = bc * k / pieces;
    r1 = bc * (k + 1) / pieces;
  };
  auto unstage = [&](int64_t c) -> hipError_t {  // chunk c: wait for its results, copy them to the caller's arrays
    auto& st = ctx->stage[c & 1];
    const int64_t b0 = c * chunk, bc = std::min(chunk, B - b0);
    const Layout L = layout(chunk, n, pps, want_g, want_p);
    const char* out = st.pinned + L.in_bytes;
    const int pieces = pieces_of(bc);
    for (int k = 0; k < pieces; ++k) {
      int64_t r0, r1;
      piece_range(bc, pieces, k, r0, r1);
      hipError_t err = hipEventSynchronize(st.piece_ready[k]);
      if (err != hipSuccess) return err;
      const size_t off = static_cast<size_t>(r0) * n * sizeof(double), len = static_cast<size_t>(r1 - r0) * n * sizeof(double);
      parallel_memcpy(reinterpret_cast<char*>(x_out + b0 * n) + off, out + L.x + off, len);
      if (want_g) parallel_memcpy(reinterpret_cast<char*>(g_out + b0 * n) + off, out + L.g + off, len);
    }
    hipError_t err = hipEventSynchronize(st.out_ready);
    if (err != hipSuccess) return err;
    std::memcpy(f_out + b0, out + L.f, static_cast<size_t>(bc) * sizeof(double));
    if (want_p) std::memcpy(progress_out + b0, out + L.p, static_cast<size_t>(bc) * sizeof(mi355_lbfgs_progress));
    return hipSuccess;
  };

  int first_traced = 0;
  for (int64_t c = 0; c < chunks && e == hipSuccess; ++c) {
    auto& st = ctx->stage[c & 1];
    const int64_t b0 = c * chunk, bc = std::min(chunk, B - b0);
    const Layout L = layout(chunk, n, pps, want_g, want_p);
    if (c >= 2) {  // the slot is free again once chunk c - 2 has been copied out
      e = unstage(c - 2);
      if (e != hipSuccess) break;
    }
    {
      const int pieces = pieces_of(bc);
      for (int k = 0; k < pieces && e == hipSuccess; ++k) {
        int64_t r0, r1;
        piece_range(bc, pieces, k, r0, r1);
        const size_t off = static_cast<size_t>(r0) * n * sizeof(double), len = static_cast<size_t>(r1 - r0) * n * sizeof(double);
        parallel_memcpy(st.pinned + L.x0 + off, reinterpret_cast<const char*>(x0 + b0 * n) + off, len);
        e = hipMemcpyAsync(st.device + L.x0 + off, st.pinned + L.x0 + off, len, hipMemcpyHostToDevice, ctx->stream_in);
      }
      if (pps && e == hipSuccess) {
        const size_t len = static_cast<size_t>(bc) * pps * sizeof(double);
        parallel_memcpy(st.pinned + L.pp, desc->per_problem_data + b0 * pps, len);
        e = hipMemcpyAsync(st.device + L.pp, st.pinned + L.pp, len, hipMemcpyHostToDevice, ctx->stream_in);
      }
    }
    if (e == hipSuccess) e = hipEventRecord(st.in_ready, ctx->stream_in);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream_solve, st.in_ready, 0);
    if (e != hipSuccess) break;
    mi355_lbfgs_desc dd = *desc;
    dd.per_problem_data = pps ? reinterpret_cast<const double*>(st.device + L.pp) : nullptr;
    mi355_lbfgs_trace dt;
    std::vector<int64_t> local;
    dd.trace = nullptr;
    if (ht) {  // the traced problems of this chunk: rows first_traced .. of the index-ordered device arrays
      int last = first_traced;
      while (last < ht->count && ht->problems[order[last]] < b0 + bc) ++last;
      if (last > first_traced) {
        for (int i = first_traced; i < last; ++i) local.push_back(ht->problems[order[i]] - b0);
        const size_t row0 = static_cast<size_t>(first_traced) * ht->capacity;
        dt.count = last - first_traced;
        dt.capacity = ht->capacity;
        dt.problems = local.data();
        dt.records = reinterpret_cast<mi355_lbfgs_trace_record*>(trace_dev + t_rec) + row0;
        dt.x = ht->x ? reinterpret_cast<double*>(trace_dev + t_x) + row0 * n : nullptr;
        dt.g = ht->g ? reinterpret_cast<double*>(trace_dev + t_g) + row0 * n : nullptr;
        dt.written = reinterpret_cast<unsigned*>(trace_dev + t_w) + first_traced;
        dd.trace = &dt;
        first_traced = last;
      }
    }
    char* out = st.device + L.in_bytes;
    const int rc2 = solve(&dd, bc, reinterpret_cast<const double*>(st.device + L.x0), reinterpret_cast<double*>(out + L.x),
                          reinterpret_cast<double*>(out + L.f), want_g ? reinterpret_cast<double*>(out + L.g) : nullptr,
                          want_p ? reinterpret_cast<mi355_lbfgs_progress*>(out + L.p) : nullptr, ctx->stream_solve);
    if (rc2 != MI355_OK) {
      (void)hipDeviceSynchronize();
      return cleanup(rc2);
    }
    e = hipEventRecord(st.solved, ctx->stream_solve);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream_out, st.solved, 0);
    {
      const int pieces = pieces_of(bc);
      char* pin_out = st.pinned + L.in_bytes;
      for (int k = 0; k < pieces && e == hipSuccess; ++k) {
        int64_t r0, r1;
        piece_range(bc, pieces, k, r0, r1);
        const size_t off = static_cast<size_t>(r0) * n * sizeof(double), len = static_cast<size_t>(r1 - r0) * n * sizeof(double);
        e = hipMemcpyAsync(pin_out + L.x + off, out + L.x + off, len, hipMemcpyDeviceToHost, ctx->stream_out);
        if (e == hipSuccess && want_g)
          e = hipMemcpyAsync(pin_out + L.g + off, out + L.g + off, len, hipMemcpyDeviceToHost, ctx->stream_out);
        if (e == hipSuccess) e = hipEventRecord(st.piece_ready[k], ctx->stream_out);
      }
      if (e == hipSuccess)  // f and the progress records
        e = hipMemcpyAsync(pin_out + L.f, out + L.f, L.out_bytes - L.f, hipMemcpyDeviceToHost, ctx->stream_out);
    }
    if (e == hipSuccess) e = hipEventRecord(st.out_ready, ctx->stream_out);
  }
  for (int64_t c = std::max<int64_t>(0, chunks - 2); c < chunks && e == hipSuccess; ++c) e = unstage(c);
  if (e == hipSuccess && ht) {  // trace rows back to the caller's order
    e = hipStreamSynchronize(ctx->stream_solve);
    const size_t rowb = static_cast<size_t>(ht->capacity);
    for (int i = 0; i < ht->count && e == hipSuccess; ++i) {
      const int u = order[i];
      e = hipMemcpy(ht->records + u * rowb, trace_dev + t_rec + i * rowb * sizeof(mi355_lbfgs_trace_record),
                    rowb * sizeof(mi355_lbfgs_trace_record), hipMemcpyDeviceToHost);
      if (e == hipSuccess && ht->x)
        e = hipMemcpy(ht->x + u * rowb * n, trace_dev + t_x + i * rowb * n * sizeof(double), rowb * n * sizeof(double),
                      hipMemcpyDeviceToHost);
      if (e == hipSuccess && ht->g)
        e = hipMemcpy(ht->g + u * rowb * n, trace_dev + t_g + i * rowb * n * sizeof(double), rowb * n * sizeof(double),
                      hipMemcpyDeviceToHost);
      if (e == hipSuccess)
        e = hipMemcpy(ht->written + u, trace_dev + t_w + i * sizeof(unsigned), sizeof(unsigned), hipMemcpyDeviceToHost);
    }
  }
  if (e != hipSuccess) {
    (void)hipDeviceSynchronize();
    return cleanup(fail(MI355_ERR_HIP, std::string("host batch: ") + hipGetErrorString(e)));
  }
  return cleanup(MI355_OK);
}

// [problems, unconverged (status <= IterationLimit), iterations] of a progress array, added into out[3]
__global__ void count_flags_kernel(const mi355_lbfgs_progress* p, long long B, unsigned long long* out) {
  unsigned long long bad = 0, it = 0;
  for (long long b = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; b < B;
       b += static_cast<long long>(gridDim.x) * blockDim.x) {
    bad += (p[b].status <= MI355_STATUS_ITERATION_LIMIT) ? 1u : 0u;
    it += p[b].num_iterations;
  }
  for (int off = 32; off > 0; off >>= 1) {
    bad += __shfl_down(bad, off, 64);
    it += __shfl_down(it, off, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(out + 1, bad);
    atomicAdd(out + 2, it);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(out + 0, static_cast<unsigned long long>(B));
}

// ---- librccl, loaded on first use ------------------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string error;
  bool ok = false;
  bool load() {
    if (ok) return true;
    if (handle) {  // an earlier attempt found the library but not every symbol: start over
      dlclose(handle);
      handle = nullptr;
    }
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (handle) break;
    }
    if (!handle) {
      error = std::string("librccl.so not found: ") + dlerror();
      return false;
    }
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(handle, "ncclAllReduce"));
    GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
    GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd) {
      error = "librccl.so lacks ncclCommInitAll / ncclAllReduce / ncclGroupStart";
      dlclose(handle);
      handle = nullptr;
      return false;
    }
    ok = true;
    return true;
  }
};
Rccl g_rccl;
constexpr int kNcclUint64 = 5, kNcclSum = 0;  // ncclDataType_t / ncclRedOp_t values of nccl.h (rccl.h)

}  // namespace

struct mi355_lbfgs_group {
  std::vector<mi355_lbfgs_ctx*> ctx;   // one per entry of the device list (an entry may repeat a device)
  std::vector<int> distinct;           // the device of every RANK of the collective (normally: the distinct devices, in
                                       // first-appearance order)
  std::vector<void*> comm;             // one RCCL communicator rank per entry of `distinct`
  std::vector<int> leader;             // index into ctx of the first context of rank d
  std::vector<int> rank_of;            // member -> rank
  // MI355_GROUP_DRY_RUN_RANKS=1 (a one-GPU box cannot give a group two RCCL ranks): every member is a rank of its own even
  // when members share a device, and the all-reduce among the ranks is a HOST-side sum instead of ncclAllReduce.  Every
  // D > 1 code path of the group — per-rank records, one flag buffer per rank, the agreement check over the ranks, the
  // per-rank counting kernels — then runs end to end on one device; RCCL itself is not exercised (it is at D = 1 otherwise).
  bool host_allreduce = false;
};

extern "C" {

int mi355_lbfgs_group_create(const int* devices, int n_devices, mi355_lbfgs_group** out) {
  if (!out) return fail(MI355_ERR_INVALID_ARGUMENT, "null out pointer");
  *out = nullptr;
  if (!devices || n_devices < 1 || n_devices > 64) return fail(MI355_ERR_INVALID_ARGUMENT, "device list of 1..64 entries");
  auto* g = new mi355_lbfgs_group();
  const char* dry = std::getenv("MI355_GROUP_DRY_RUN_RANKS");
  g->host_allreduce = dry && dry[0] == '1';
  if (g->host_allreduce)   // (read once, here, and never silent: this group does NOT use RCCL)
    std::fprintf(stderr, "mi355_lbfgs: MI355_GROUP_DRY_RUN_RANKS=1 — this device group sums its convergence records on the "
                         "HOST instead of ncclAllReduce (one-GPU dry run of the multi-rank paths; not for production)\n");
  for (int i = 0; i < n_devices; ++i) {
    mi355_lbfgs_ctx* c = nullptr;
    const int rc = mi355_lbfgs_create(devices[i], &c);
    if (rc != MI355_OK) {
      mi355_lbfgs_group_destroy(g);
      return rc;
    }
    g->ctx.push_back(c);
    const auto seen = std::find(g->distinct.begin(), g->distinct.end(), devices[i]);
    if (g->host_allreduce || seen == g->distinct.end()) {
      g->rank_of.push_back(static_cast<int>(g->distinct.size()));
      g->distinct.push_back(devices[i]);
      g->leader.push_back(i);
    } else {
      g->rank_of.push_back(static_cast<int>(seen - g->distinct.begin()));
    }
  }
  if (g->host_allreduce) {   // no communicator: group_allreduce sums on the host
    *out = g;
    return MI355_OK;
  }
  if (!g_rccl.load()) {
    mi355_lbfgs_group_destroy(g);
    return fail(MI355_ERR_UNSUPPORTED, g_rccl.error);
  }
  g->comm.assign(g->distinct.size(), nullptr);
  const int nrc = g_rccl.CommInitAll(g->comm.data(), static_cast<int>(g->distinct.size()), g->distinct.data());
  if (nrc != 0) {
    g->comm.clear();
    mi355_lbfgs_group_destroy(g);
    return fail(MI355_ERR_HIP, std::string("ncclCommInitAll: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "failed"));
  }
  *out = g;
  return MI355_OK;
}

void mi355_lbfgs_group_destroy(mi355_lbfgs_group* g) {
  if (!g) return;
  for (void* c : g->comm)
    if (c) (void)g_rccl.CommDestroy(c);
  for (auto* c : g->ctx) mi355_lbfgs_destroy(c);
  delete g;
}

int mi355_lbfgs_group_size(const mi355_lbfgs_group* g) { return g ? static_cast<int>(g->ctx.size()) : 0; }

mi355_lbfgs_ctx* mi355_lbfgs_group_context(mi355_lbfgs_group* g, int index) {
  if (!g || index < 0 || index >= static_cast<int>(g->ctx.size())) return nullptr;
  return g->ctx[index];
}

}  // extern "C"

namespace {

// shard s of a batch of B over G members: the contiguous range [B s / G, B (s + 1) / G)  
inline void shard_range(int64_t B, int s, int G, int64_t& lo, int64_t& hi) {
  lo = B * s / G;
  hi = B * (s + 1) / G;
}

// Sum of the ranks' 3-word records, in place in every rank's flags_dev (enqueued on each leader's stream_solve):
// ncclAllReduce(ncclUint64, ncclSum) over the group's communicator — or, in the dry run of mi355_lbfgs_group above, a
// host-side sum handed back to every rank.
int allreduce_rank_records(mi355_lbfgs_group* g) {
  const int D = static_cast<int>(g->distinct.size());
  if (g->host_allreduce) {
    unsigned long long sum[3] = {0, 0, 0};
    for (int d = 0; d < D; ++d) {
      mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
      mi355::DeviceGuard guard(c->device);
      unsigned long long r[3];
      HIP_TRY(hipMemcpyAsync(r, c->flags_dev, sizeof(r), hipMemcpyDeviceToHost, c->stream_solve));
      HIP_TRY(hipStreamSynchronize(c->stream_solve));
      for (int k = 0; k < 3; ++k) sum[k] += r[k];
    }
    for (int d = 0; d < D; ++d) {
      mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
      mi355::DeviceGuard guard(c->device);
      HIP_TRY(hipMemcpyAsync(c->flags_dev, sum, sizeof(sum), hipMemcpyHostToDevice, c->stream_solve));
      HIP_TRY(hipStreamSynchronize(c->stream_solve));
    }
    return MI355_OK;
  }
  int nrc = g_rccl.GroupStart();
  for (int d = 0; d < D && nrc == 0; ++d) {
    mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
    nrc = g_rccl.AllReduce(c->flags_dev, c->flags_dev, 3, kNcclUint64, kNcclSum, g->comm[d], c->stream_solve);
  }
  const int nrc_end = g_rccl.GroupEnd();
  if (nrc == 0) nrc = nrc_end;
  if (nrc != 0)
    return fail(MI355_ERR_HIP, std::string("ncclAllReduce: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(nrc) : "failed"));
  return MI355_OK;
}

// The one collective of the path: local[d * 3 .. d * 3 + 2] = {problems, unconverged, iterations} of distinct device d
// -> the same global record on every device (ncclAllReduce, ncclUint64, ncclSum, one rank per distinct device)
int group_allreduce(mi355_lbfgs_group* g, const std::vector<unsigned long long>& local, unsigned long long (&result)[3]) {
  const int D = static_cast<int>(g->distinct.size());
  for (int d = 0; d < D; ++d) {
    mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
    mi355::DeviceGuard guard(c->device);
    if (!c->flags_dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->flags_dev), 3 * sizeof(unsigned long long)));
    if (!c->stream_solve) HIP_TRY(hipStreamCreateWithFlags(&c->stream_solve, hipStreamNonBlocking));
    HIP_TRY(hipMemcpyAsync(c->flags_dev, &local[d * 3], 3 * sizeof(unsigned long long), hipMemcpyHostToDevice, c->stream_solve));
  }
  const int rc_reduce = allreduce_rank_records(g);
  if (rc_reduce != MI355_OK) return rc_reduce;
  for (int d = 0; d < D; ++d) {  // every device holds the same global record; read them all, return the first
    mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
    mi355::DeviceGuard guard(c->device);
    unsigned long long r[3];
    HIP_TRY(hipMemcpyAsync(r, c->flags_dev, sizeof(r), hipMemcpyDeviceToHost, c->stream_solve));
    HIP_TRY(hipStreamSynchronize(c->stream_solve));
    if (d == 0) {
      std::memcpy(result, r, sizeof(r));
    } else if (std::memcmp(result, r, sizeof(r)) != 0) {
      return fail(MI355_ERR_HIP, "the devices disagree on the all-reduced convergence record");
    }
  }
  return MI355_OK;
}
int rank_of_member(const mi355_lbfgs_group* g, int member) { return g->rank_of[static_cast<size_t>(member)]; }

// device-pointer solve of one member's shard on `stream`; (context, member index) -> the callable
using MemberSolve = std::function<DeviceSolve(mi355_lbfgs_ctx*, int)>;

// HOST arrays, whole group: one host thread per member stages, solves and un-stages its shard on its own context
int group_minimize(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, int64_t B, const double* x0, double* x_out,
                   double* f_out, double* g_out, mi355_lbfgs_progress* progress_out, uint64_t* flag_out,
                   const MemberSolve& member_solve) {
  if (!g || !desc) return fail(MI355_ERR_INVALID_ARGUMENT, "null group / desc");
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  if (desc->trace) return fail(MI355_ERR_UNSUPPORTED, "the sharded entry point takes no trace (trace one shard through its context)");
  if (B > 0 && (!x0 || !x_out || !f_out)) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out");
  const int G = static_cast<int>(g->ctx.size());
  const int n = desc->n;
  std::vector<int> rcs(G, MI355_OK);
  std::vector<std::string> errs(G);
  std::vector<mi355_lbfgs_progress> local_progress;
  if (!progress_out && B > 0) {  // the convergence record needs the status words
    local_progress.resize(static_cast<size_t>(B));
    progress_out = local_progress.data();
  }
  auto member = [&](int s) {
    int64_t lo, hi;
    shard_range(B, s, G, lo, hi);
    mi355_lbfgs_ctx* c = g->ctx[s];
    mi355::DeviceGuard guard(c->device);
    int rc = MI355_OK;
    if (hi > lo) {
      mi355_lbfgs_desc d = *desc;
      if (d.per_problem_data) d.per_problem_data = desc->per_problem_data + lo * desc->per_problem_stride;
      rc = run_host_batch(c, &d, hi - lo, x0 + lo * n, x_out + lo * n, f_out + lo, g_out ? g_out + lo * n : nullptr,
                          progress_out + lo, member_solve(c, s));
    }
    rcs[s] = rc;
    if (rc != MI355_OK) errs[s] = mi355_lbfgs_last_error();
  };
  std::vector<std::thread> threads;
  for (int s = 1; s < G; ++s) threads.emplace_back(member, s);
  member(0);
  for (auto& t : threads) t.join();
  for (int s = 0; s < G; ++s)
    if (rcs[s] != MI355_OK) return fail(rcs[s], "group member " + std::to_string(s) + ": " + errs[s]);

  const int D = static_cast<int>(g->distinct.size());
  std::vector<unsigned long long> local(static_cast<size_t>(D) * 3, 0ULL);
  for (int s = 0; s < G; ++s) {  // members that share a device are added on the host first
    int64_t lo, hi;
    shard_range(B, s, G, lo, hi);
    const int d = rank_of_member(g, s);
    local[d * 3 + 0] += static_cast<unsigned long long>(hi - lo);
    for (int64_t b = lo; b < hi; ++b) {
      local[d * 3 + 1] += (progress_out[b].status <= MI355_STATUS_ITERATION_LIMIT) ? 1u : 0u;
      local[d * 3 + 2] += progress_out[b].num_iterations;
    }
  }
  unsigned long long result[3] = {0, 0, 0};
  const int rc = group_allreduce(g, local, result);
  if (rc != MI355_OK) return rc;
  if (flag_out) {
    flag_out[0] = result[0];
    flag_out[1] = result[1];
    flag_out[2] = result[2];
  }
  return MI355_OK;
}

// DEVICE arrays, whole group ("per-GPU device buffers"): member s solves counts[s] problems that
// already live on its device — x0[s], x_out[s], f_out[s], progress_out[s] (g_out[s], per_problem[s] optional) — on its
// context's own stream; a small kernel on the SAME stream counts its convergence record, and the records are
// all-reduced.  Nothing but the 24-byte record crosses PCIe.
using MemberLaunch = std::function<int(mi355_lbfgs_ctx*, int, const mi355_lbfgs_desc*, hipStream_t)>;
int group_minimize_device(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, const int64_t* counts,
                          mi355_lbfgs_progress* const* progress_out, uint64_t* flag_out, const MemberLaunch& launch) {
  if (!g || !desc || !counts || !progress_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null group / desc / counts / progress_out");
  if (desc->trace) return fail(MI355_ERR_UNSUPPORTED, "the sharded entry point takes no trace (trace one shard through its context)");
  const int G = static_cast<int>(g->ctx.size());
  const int D = static_cast<int>(g->distinct.size());
  for (int s = 0; s < G; ++s) {
    if (counts[s] < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative shard size");
    if (counts[s] > 0 && !progress_out[s]) return fail(MI355_ERR_INVALID_ARGUMENT, "null progress array for a non-empty shard");
  }
  for (int s = 0; s < G; ++s) {  // enqueue every member's solve + count; the launches are asynchronous
    mi355_lbfgs_ctx* c = g->ctx[s];
    mi355::DeviceGuard guard(c->device);
    if (!c->flags_dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->flags_dev), 3 * sizeof(unsigned long long)));
    if (!c->stream_solve) HIP_TRY(hipStreamCreateWithFlags(&c->stream_solve, hipStreamNonBlocking));
    HIP_TRY(hipMemsetAsync(c->flags_dev, 0, 3 * sizeof(unsigned long long), c->stream_solve));
    if (counts[s] == 0) continue;
    const int rc = launch(c, s, desc, c->stream_solve);
    if (rc != MI355_OK) {
      const std::string err = mi355_lbfgs_last_error();
      for (int t = 0; t <= s; ++t) {
        mi355::DeviceGuard gt(g->ctx[t]->device);
        if (g->ctx[t]->stream_solve) (void)hipStreamSynchronize(g->ctx[t]->stream_solve);
      }
      return fail(rc, "group member " + std::to_string(s) + ": " + err);
    }
    const long long Bs = counts[s];
    const unsigned blocks = static_cast<unsigned>(std::min<long long>((Bs + 255) / 256, 1024));
    hipLaunchKernelGGL(count_flags_kernel, dim3(blocks), dim3(256), 0, c->stream_solve, progress_out[s], Bs, c->flags_dev);
    HIP_TRY(hipGetLastError());
  }
  std::vector<unsigned long long> local(static_cast<size_t>(D) * 3, 0ULL);
  for (int s = 0; s < G; ++s) {
    mi355_lbfgs_ctx* c = g->ctx[s];
    mi355::DeviceGuard guard(c->device);
    unsigned long long r[3];
    HIP_TRY(hipMemcpyAsync(r, c->flags_dev, sizeof(r), hipMemcpyDeviceToHost, c->stream_solve));
    HIP_TRY(hipStreamSynchronize(c->stream_solve));
    const int d = rank_of_member(g, s);
    for (int k = 0; k < 3; ++k) local[d * 3 + k] += r[k];
  }
  unsigned long long result[3] = {0, 0, 0};
  const int rc = group_allreduce(g, local, result);
  if (rc != MI355_OK) return rc;
  if (flag_out) {
    flag_out[0] = result[0];
    flag_out[1] = result[1];
    flag_out[2] = result[2];
  }
  return MI355_OK;
}

// per-member device copies of a host box (freed by the destructor)
struct MemberBounds {
  std::vector<double*> dev;
  std::vector<int> device;
  ~MemberBounds() {
    for (size_t s = 0; s < dev.size(); ++s)
      if (dev[s]) {
        mi355::DeviceGuard guard(device[s]);
        (void)hipFree(dev[s]);
      }
  }
  int upload(mi355_lbfgs_group* g, const double* lower, const double* upper, int n) {
    dev.assign(g->ctx.size(), nullptr);
    device.assign(g->ctx.size(), 0);
    if (!lower) return MI355_OK;
    for (size_t s = 0; s < g->ctx.size(); ++s) {
      device[s] = g->ctx[s]->device;
      mi355::DeviceGuard guard(device[s]);
      HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dev[s]), 2 * static_cast<size_t>(n) * sizeof(double)));
      HIP_TRY(hipMemcpy(dev[s], lower, n * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(dev[s] + n, upper, n * sizeof(double), hipMemcpyHostToDevice));
    }
    return MI355_OK;
  }
};

}  // namespace

namespace {
MemberSolve unconstrained_member(int solver) {  // 0 Lbfgs, 1 dense Bfgs
  return [solver](mi355_lbfgs_ctx* c, int) -> DeviceSolve {
    return [c, solver](const mi355_lbfgs_desc* dd, int64_t bc, const double* a, double* b, double* f, double* gg,
                       mi355_lbfgs_progress* p, hipStream_t st) {
      return mi355_minimize_batch_device(c, dd, bc, a, b, f, gg, p, st, solver);
    };
  };
}
int check_box(const mi355_lbfgs_desc* desc, const double* lower, const double* upper) {
  if (!desc) return fail(MI355_ERR_INVALID_ARGUMENT, "null desc");
  if ((lower == nullptr) != (upper == nullptr))
    return fail(MI355_ERR_INVALID_ARGUMENT, "lower and upper must both be given or both be NULL");
  if (desc->n < 1 || desc->n > MI355_LBFGS_MAX_N) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range [1, MI355_LBFGS_MAX_N]");
  for (int j = 0; lower && j < desc->n; ++j)
    if (lower[j] != lower[j] || upper[j] != upper[j]) return fail(MI355_ERR_INVALID_ARGUMENT, "NaN bound");
  return MI355_OK;
}
}  // namespace

extern "C" {

int mi355_lbfgs_group_minimize_batch_host(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                          double* x_out, double* f_out, double* g_out,
                                          mi355_lbfgs_progress* progress_out, uint64_t* flag_out) {
  return group_minimize(g, desc, B, x0, x_out, f_out, g_out, progress_out, flag_out, unconstrained_member(0));
}

int mi355_bfgs_group_minimize_batch_host(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                         double* x_out, double* f_out, double* g_out,
                                         mi355_lbfgs_progress* progress_out, uint64_t* flag_out) {
  return group_minimize(g, desc, B, x0, x_out, f_out, g_out, progress_out, flag_out, unconstrained_member(1));
}

// Lbfgsb::Minimize (lbfgsb.h:247-292) over the whole group, HOST arrays; lower / upper: n doubles each or both NULL
int mi355_lbfgsb_group_minimize_batch_host(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, const double* lower,
                                           const double* upper, int64_t B, const double* x0, double* x_out, double* f_out,
                                           double* g_out, mi355_lbfgs_progress* progress_out, uint64_t* flag_out) {
  if (!g) return fail(MI355_ERR_INVALID_ARGUMENT, "null group");
  int rc = check_box(desc, lower, upper);
  if (rc != MI355_OK) return rc;
  MemberBounds bounds;
  rc = bounds.upload(g, lower, upper, desc->n);
  if (rc != MI355_OK) return rc;
  return group_minimize(g, desc, B, x0, x_out, f_out, g_out, progress_out, flag_out,
                        [&bounds](mi355_lbfgs_ctx* c, int s) -> DeviceSolve {
                          double* const bd = bounds.dev[s];
                          return [c, bd](const mi355_lbfgs_desc* dd, int64_t bc, const double* a, double* b, double* f,
                                         double* gg, mi355_lbfgs_progress* p, hipStream_t st) {
                            return mi355_lbfgsb_minimize_batch(c, dd, bd, bd ? bd + dd->n : nullptr, bc, a, b, f, gg, p, st);
                          };
                        });
}

// Device-resident shards: every array argument is an array of G per-member DEVICE pointers (member s's arrays live on
// member s's device); per_problem / g_out may be NULL, or hold NULL entries.
int mi355_lbfgs_group_minimize_batch(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, const int64_t* counts,
                                     const double* const* x0, double* const* x_out, double* const* f_out,
                                     double* const* g_out, mi355_lbfgs_progress* const* progress_out,
                                     const double* const* per_problem, uint64_t* flag_out) {
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out pointer arrays");
  return group_minimize_device(g, desc, counts, progress_out, flag_out,
                               [&](mi355_lbfgs_ctx* c, int s, const mi355_lbfgs_desc* d0, hipStream_t st) {
                                 mi355_lbfgs_desc d = *d0;
                                 d.per_problem_data = per_problem ? per_problem[s] : nullptr;
                                 return mi355_minimize_batch_device(c, &d, counts[s], x0[s], x_out[s], f_out[s],
                                                                    g_out ? g_out[s] : nullptr, progress_out[s], st, 0);
                               });
}

// ... and Lbfgsb; lower / upper: per-member DEVICE pointers (n doubles each), or both NULL for the default box
int mi355_lbfgsb_group_minimize_batch(mi355_lbfgs_group* g, const mi355_lbfgs_desc* desc, const double* const* lower,
                                      const double* const* upper, const int64_t* counts, const double* const* x0,
                                      double* const* x_out, double* const* f_out, double* const* g_out,
                                      mi355_lbfgs_progress* const* progress_out, const double* const* per_problem,
                                      uint64_t* flag_out) {
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out pointer arrays");
  if ((lower == nullptr) != (upper == nullptr))
    return fail(MI355_ERR_INVALID_ARGUMENT, "lower and upper must both be given or both be NULL");
  return group_minimize_device(g, desc, counts, progress_out, flag_out,
                               [&](mi355_lbfgs_ctx* c, int s, const mi355_lbfgs_desc* d0, hipStream_t st) {
                                 mi355_lbfgs_desc d = *d0;
                                 d.per_problem_data = per_problem ? per_problem[s] : nullptr;
                                 return mi355_lbfgsb_minimize_batch(c, &d, lower ? lower[s] : nullptr, upper ? upper[s] : nullptr,
                                                                    counts[s], x0[s], x_out[s], f_out[s],
                                                                    g_out ? g_out[s] : nullptr, progress_out[s], st);
                               });
}

// convergence record of a device-resident progress array, all-reduced over the group's devices: the collective
// alone, for callers that keep their shards in HBM (each context's array lives on that context's device)
int mi355_lbfgs_group_allreduce_flags(mi355_lbfgs_group* g, const mi355_lbfgs_progress* const* progress_dev,
                                      const int64_t* counts, uint64_t* flag_out) {
  if (!g || !progress_dev || !counts || !flag_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null argument");
  const int G = static_cast<int>(g->ctx.size());
  const int D = static_cast<int>(g->distinct.size());
  for (int d = 0; d < D; ++d) {
    mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
    mi355::DeviceGuard guard(c->device);
    // The progress arrays were written by solves on streams of the caller's choosing; the counting kernel runs on the
    // context's own non-blocking stream, which nothing orders against those.  Wait for the device: every solve
    // enqueued on it before this call has then finished (mi355_lbfgs_group_minimize_batch keeps solve and count on
    // one stream and needs no such wait).
    HIP_TRY(hipDeviceSynchronize());
    if (!c->flags_dev) HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->flags_dev), 3 * sizeof(unsigned long long)));
    if (!c->stream_solve) HIP_TRY(hipStreamCreateWithFlags(&c->stream_solve, hipStreamNonBlocking));
    HIP_TRY(hipMemsetAsync(c->flags_dev, 0, 3 * sizeof(unsigned long long), c->stream_solve));
    for (int s = 0; s < G; ++s) {
      if (rank_of_member(g, s) != d || counts[s] <= 0) continue;
      if (!progress_dev[s]) return fail(MI355_ERR_INVALID_ARGUMENT, "null progress array for a non-empty shard");
      const long long Bs = counts[s];
      const unsigned blocks = static_cast<unsigned>(std::min<long long>((Bs + 255) / 256, 1024));
      hipLaunchKernelGGL(count_flags_kernel, dim3(blocks), dim3(256), 0, c->stream_solve, progress_dev[s], Bs, c->flags_dev);
      HIP_TRY(hipGetLastError());
    }
  }
  const int rc_reduce = allreduce_rank_records(g);
  if (rc_reduce != MI355_OK) return rc_reduce;
  mi355_lbfgs_ctx* c0 = g->ctx[g->leader[0]];
  mi355::DeviceGuard guard(c0->device);
  unsigned long long r[3];
  HIP_TRY(hipMemcpyAsync(r, c0->flags_dev, sizeof(r), hipMemcpyDeviceToHost, c0->stream_solve));
  HIP_TRY(hipStreamSynchronize(c0->stream_solve));
  for (int d = 1; d < D; ++d) {
    mi355_lbfgs_ctx* c = g->ctx[g->leader[d]];
    mi355::DeviceGuard gd(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream_solve));
  }
  flag_out[0] = r[0];
  flag_out[1] = r[1];
  flag_out[2] = r[2];
  return MI355_OK;
}

// ---- host-pointer entry points of the single-context API (declared in include/mi355_lbfgs.h) -------------
int mi355_lbfgs_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                    double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out) {
  if (!ctx || !desc) return fail(MI355_ERR_INVALID_ARGUMENT, "null context / desc");
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  if (B == 0) return mi355_minimize_batch_device(ctx, desc, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out");
  if (desc->n < 1 || desc->n > MI355_LBFGS_WIDE_MAX_N) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range [1, MI355_LBFGS_WIDE_MAX_N]");
  MI355_ENTER_DEVICE(ctx);
  return run_host_batch(ctx, desc, B, x0, x_out, f_out, g_out, progress_out,
                        [&](const mi355_lbfgs_desc* dd, int64_t bc, const double* a, double* b, double* f, double* gg,
                            mi355_lbfgs_progress* p, hipStream_t st) {
                          return mi355_minimize_batch_device(ctx, dd, bc, a, b, f, gg, p, st, 0);
                        });
}

int mi355_bfgs_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, int64_t B, const double* x0,
                                   double* x_out, double* f_out, double* g_out, mi355_lbfgs_progress* progress_out) {
  if (!ctx || !desc) return fail(MI355_ERR_INVALID_ARGUMENT, "null context / desc");
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  if (B == 0) return mi355_minimize_batch_device(ctx, desc, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1);
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out");
  if (desc->n < 1 || desc->n > MI355_LBFGS_MAX_N) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range [1, MI355_LBFGS_MAX_N]");
  MI355_ENTER_DEVICE(ctx);
  return run_host_batch(ctx, desc, B, x0, x_out, f_out, g_out, progress_out,
                        [&](const mi355_lbfgs_desc* dd, int64_t bc, const double* a, double* b, double* f, double* gg,
                            mi355_lbfgs_progress* p, hipStream_t st) {
                          return mi355_minimize_batch_device(ctx, dd, bc, a, b, f, gg, p, st, 1);
                        });
}

int mi355_lbfgsb_minimize_batch_host(mi355_lbfgs_ctx* ctx, const mi355_lbfgs_desc* desc, const double* lower,
                                     const double* upper, int64_t B, const double* x0, double* x_out, double* f_out,
                                     double* g_out, mi355_lbfgs_progress* progress_out) {
  if (!ctx || !desc) return fail(MI355_ERR_INVALID_ARGUMENT, "null context / desc");
  if (B < 0) return fail(MI355_ERR_INVALID_ARGUMENT, "negative batch size");
  if ((lower == nullptr) != (upper == nullptr))
    return fail(MI355_ERR_INVALID_ARGUMENT, "lower and upper must both be given or both be NULL");
  if (desc->n < 1 || desc->n > MI355_LBFGS_MAX_N) return fail(MI355_ERR_INVALID_ARGUMENT, "n out of range [1, MI355_LBFGS_MAX_N]");
  // a NaN bound makes the breakpoint order of the Cauchy search undefined in the reference as well (std::sort over
  // NaN keys, lbfgsb.h:298-305, :349): refused rather than reproduced
  for (int j = 0; lower && j < desc->n; ++j)
    if (lower[j] != lower[j] || upper[j] != upper[j]) return fail(MI355_ERR_INVALID_ARGUMENT, "NaN bound");
  if (B == 0) return mi355_lbfgsb_minimize_batch(ctx, desc, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
  if (!x0 || !x_out || !f_out) return fail(MI355_ERR_INVALID_ARGUMENT, "null x0 / x_out / f_out");
  MI355_ENTER_DEVICE(ctx);
  double* bounds_dev = nullptr;
  if (lower) {
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&bounds_dev), 2 * static_cast<size_t>(desc->n) * sizeof(double)));
    hipError_t e = hipMemcpy(bounds_dev, lower, desc->n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(bounds_dev + desc->n, upper, desc->n * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      (void)hipFree(bounds_dev);
      return fail(MI355_ERR_HIP, std::string("lbfgsb host batch: ") + hipGetErrorString(e));
    }
  }
  const int rc = run_host_batch(
      ctx, desc, B, x0, x_out, f_out, g_out, progress_out,
      [&](const mi355_lbfgs_desc* dd, int64_t bc, const double* a, double* b, double* f, double* gg, mi355_lbfgs_progress* p,
          hipStream_t st) {
        return mi355_lbfgsb_minimize_batch(ctx, dd, bounds_dev, bounds_dev ? bounds_dev + dd->n : nullptr, bc, a, b, f, gg,
                                           p, st);
      });
  if (bounds_dev) (void)hipFree(bounds_dev);
  return rc;
}

}  // extern "C"
