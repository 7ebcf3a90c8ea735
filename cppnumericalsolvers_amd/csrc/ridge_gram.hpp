// ridge_gram.hpp — the ridge objective  f(x) = ||A x - y_b||^2 + lambda ||x||^2  (README.md:122-167,
// `SquaredError(A, y) + lambda * L2Reg(n)`) in NORMAL-EQUATION form (objective id MI355_OBJ_SQUARED_ERROR_RIDGE_GRAM).
//
// A is shared by the batch, so  f(x) = x^T G x - 2 c_b^T x + y_b^T y_b  with ONE Gram matrix
//   G = A^T A + lambda I        (n x n, computed once per matrix on the matrix cores — ridge_gram_matrix_kernel —
//                                held in LDS up to n = 128, streamed through L2 above)
// and per problem
//   c_b = A^T y_b,  yy_b = y_b . y_b        (a batched GEMM  C[B x n] = Y[B x rows] A[rows x n]).
// The only place where a problem's data meets the matrix is that GEMM; it runs ONCE per problem, on the matrix cores
// (ridge_gram_prepass_kernel: v_mfma_f64_16x16x4_f64, sixteen problems per tile).  After it an evaluation is
//   t = G x;  h = t - c;  grad = 2 h;  f = x . (h - c) + yy
// — n^2 multiply-adds instead of the 2 rows n of the two matrix-vector products (a quarter for A 128 x 64), with no
// cross-problem coupling, so the solve runs in the ordinary persistent lbfgs_solve_kernel (independent wavefronts,
// no workgroup barrier per evaluation, as the matrix-core kernel of objective id 3 needs).
// Algebraically the same function; rounding differs (the reference forms r = A x - y every time), and the twin the
// tests keep restates exactly these operations.  x*, f* stay within the north star's 1e-6 of the reference.
#pragma once
#include "lbfgs_kernel.hpp"

namespace mi355 {

constexpr int kGramMaxRows = 4096;  // = MI355_LBFGS_GRAM_MAX_ROWS
constexpr int kGramMaxCols = 256;   // = MI355_LBFGS_MAX_N
constexpr int kGramLdsMaxCols = 128;  // G lives in LDS up to this padded width (128 x 128 doubles = 128 KB), in L2 above

// columns of the padded copy of A the matrix-core kernels read: P rounded up to a multiple of 16
__host__ __device__ constexpr int gram_a_cols(int P) { return (P + 15) & ~15; }

// Device functor.  params (device): rows, lambda, G[P][P] row major (zero padded; exactly symmetric), P = W * E.
// Per-problem row (written by the pre-pass, stride P + 2): c zero padded to P, then yy.
// G_IN_LDS: the workgroup's wavefronts share one LDS copy of G (P <= 128); otherwise every evaluation streams G from
// memory (512 KB at P = 256: resident in the L2 of the XCD).
// OWN (objective id MI355_OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM): every problem has its own matrix — the pre-pass
// (ridge_gram_own_prepass_kernel) writes G_b behind the problem's (c_b, yy_b), rows of P + 2 + P * P doubles, and an
// evaluation streams ITS G_b (the resident problems' matrices sit in L2 / the Infinity Cache: 32 KB each at n = 64).
template <int W, int E, bool G_IN_LDS = (W * E <= kGramLdsMaxCols), bool OWN = false>
struct RidgeGramObjective {
  static_assert(!(OWN && G_IN_LDS), "an own matrix is read from memory");
  static constexpr int P = W * E;
  static constexpr int kLdsDoubles = P;  // x staging per problem
  __host__ __device__ static constexpr int shared_lds_doubles() { return G_IN_LDS ? P * P : 0; }
  const double* g_global;
  const double* G;  // LDS (G_IN_LDS) or global
  double* xs;
  const double* row_;  // the problem's pre-pass row: c padded to P, then yy (re-read per evaluation: an L2 hit that the
                       // matrix-vector loop hides, instead of 2 E + 2 registers held across the whole iteration)

  __device__ __forceinline__ void load(const double* params, int, int, double* lds_scratch, double* lds_shared) {
    g_global = params + 2;
    G = G_IN_LDS ? lds_shared : g_global;
    xs = lds_scratch;
  }
  __device__ __forceinline__ void fill_shared(double* lds_shared, int tid, int nthreads) const {
    if constexpr (G_IN_LDS) {
      for (int t = tid; t < P * P; t += nthreads) lds_shared[t] = g_global[t];
    }
  }
  __device__ __forceinline__ void begin_problem(const double* per_problem, long long prob, int stride, int) {
    row_ = per_problem + prob * stride;
    if constexpr (OWN) G = row_ + (P + 2);
  }

  // t_i = sum_j G[j][i] x_j (G is symmetric: lane sl reads its E consecutive entries of row j — consecutive lanes,
  // consecutive addresses), one fused chain per coordinate, ascending j over the padded width.
  template <int WW, int EE>
  __device__ __forceinline__ double eval_fma(const double (&x)[EE], double (&g)[EE], int n, int sl) const {
    static_assert(WW == W && EE == E, "mapping");
    double c[E];
#pragma unroll
    for (int e = 0; e < E; ++e) c[e] = row_[sl * E + e];
    const double yy = row_[P];
#pragma unroll
    for (int e = 0; e < E; ++e) xs[sl * E + e] = x[e];
    segment_lds_fence();
    double t[E];
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = 0.0;
    const double* mine = G + sl * E;
    // Rows of G are fetched kBatch at a time with every read of a batch in flight before the first multiply-add (the
    // scheduling barriers keep the compiler from re-serialising "read, wait, use": left alone it trades the
    // latency of ~3 P / 2 dependent LDS round trips per evaluation for a handful of registers).
#ifndef MI355_GRAM_OWN_BATCH
#define MI355_GRAM_OWN_BATCH 8
#endif
    // (own matrix: the rows come from HBM, not LDS; deeper batches were measured and do not help — the kernel already
    //  streams at 91 % of the achievable HBM rate, profiles/r4_ab_own_matrix.txt; the macro is the A/B switch)
    // (E = 4 is the P = 256 mapping: G comes through L2 and the history ring leaves ONE wavefront per SIMD, so the loads in
    //  flight per wavefront are the memory-level parallelism of the whole CU: eight rows = sixteen 16-byte loads per lane;
    //  with four the evaluation ran at 21 TB/s of L2 reads against the 39 TB/s the L1s can take in)
#ifndef MI355_GRAM_WIDE_BATCH
#define MI355_GRAM_WIDE_BATCH 8
#endif
    constexpr int kBatch = (E >= 4) ? MI355_GRAM_WIDE_BATCH : ((OWN && P >= 16) ? (MI355_GRAM_OWN_BATCH < P ? MI355_GRAM_OWN_BATCH : P) : 8);
    static_assert(P % kBatch == 0, "padded width");
    // rows j >= n of G are zero and so is x_j there: fma(0, 0, t) = t (t is never -0: it starts at +0), so the chain stops
    // at the last batch that holds a real row — n = 200 reads 200 of the 256 padded rows
    const int n_rows = (n + kBatch - 1) & ~(kBatch - 1);
#pragma unroll 1
    for (int j0 = 0; j0 < n_rows; j0 += kBatch) {
      double gb[kBatch][E], xb[kBatch];
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
        xb[q] = xs[j0 + q];
#pragma unroll
        for (int e = 0; e < E; ++e) gb[q][e] = mine[(j0 + q) * P + e];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < kBatch; ++q) {
#pragma unroll
        for (int e = 0; e < E; ++e) t[e] = __builtin_fma(gb[q][e], xb[q], t[e]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    segment_lds_fence();
    double u[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const double h = t[e] - c[e];
      g[e] = (sl * E + e < n) ? 2.0 * h : 0.0;
      u[e] = h - c[e];
    }
    return seg_dot<W, E, ArithFma>(x, u) + yy;
  }
};

#ifdef MI355_RIDGE_GRAM_PREPASS_TU  // the two matrix-core kernels are plain functions: defined in ONE unit (dispatch_ridge_gram.hip)
// Operand layout of v_mfma_f64_16x16x4_f64 (scripts/microbench/mfma_f64_probe.hip): A[i][k] in lane i + 16 k, B[k][j] in
// lane j + 16 k, D row (lane >> 4) + 4 reg, column lane & 15; D = A B + C is, per entry, the chain of fused multiply-adds
// in k order starting from C.
typedef double gram_v4d __attribute__((ext_vector_type(4)));

// G = A^T A + lambda I on the matrix cores: one wavefront per 16 x 16 tile of G, the rows of A walked in ascending order,
// so entry (i, j) is the ascending fused chain  acc = fma(A[r][i], A[r][j], acc)  from 0 — the products commute, so G is
// symmetric to the bit — with lambda added to the diagonal entries i < n afterwards.  a_pad: A zero padded to
// [rows4][AC] row major (rows4 a multiple of 4, AC = gram_a_cols(P)).  G: [P][P], padding zero.
__global__ __launch_bounds__(64) void ridge_gram_matrix_kernel(const double* __restrict__ a_pad, int rows4, int AC, int n,
                                                               int P, double lambda, double* __restrict__ G) {
  const int tiles = AC / 16;
  const int it = static_cast<int>(blockIdx.x) / tiles, jt = static_cast<int>(blockIdx.x) % tiles;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, k = lane >> 4;
  gram_v4d acc = gram_v4d{0.0, 0.0, 0.0, 0.0};
  for (int t = 0; t < rows4 / 4; ++t) {
    const double* row = a_pad + static_cast<long long>(4 * t + k) * AC;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(row[it * 16 + i], row[jt * 16 + i], acc, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int gi = it * 16 + k + 4 * reg, gj = jt * 16 + i;
    if (gi < P && gj < P) G[static_cast<long long>(gi) * P + gj] = (gi == gj && gi < n) ? acc[reg] + lambda : acc[reg];
  }
}

// Pre-pass: out[b] = (c_b zero padded to `cols`, yy_b, one pad word: rows of cols + 2 doubles, 16-byte aligned),
// c_b = A^T y_b as ascending fused chains over the rows (what the MFMA accumulates when the row tiles are walked in
// order), yy_b = y_b . y_b as four interleaved chains (rows r = k mod 4) added pairwise.  One wavefront per 16 problems;
// the column tiles of A are taken four at a time (the right-hand sides are re-read per group: they are L1 hits).
// a_pad: A zero padded to [rows4][AC] row major.
// The rows are walked in chunks of kChunk k-steps with every load of a chunk — kChunk right-hand-side elements and
// kChunk x NT tile elements — issued before the first MFMA (the first form loaded, waited and multiplied once per k-step:
// 1.88 ms for 32 768 right-hand sides of 1000 rows against a matrix-core time of 0.17 ms; profiles/r4_ab_own_prepass.txt).
// Loads are unconditional from clamped addresses; a k-step past the last row multiplies by a zeroed right-hand side
// (fma(0, b, acc) = acc: acc is never -0).  Per tile the MFMA sequence is unchanged: same bits.
#ifndef MI355_GRAM_PREPASS_DIRECT
template <int NT>
__device__ __forceinline__ void gram_prepass_group(const double* __restrict__ a_pad, const double* __restrict__ yrow,
                                                   bool live, int rows, int rows4, int AC, int n, int cols, long long B,
                                                   long long b0, int jg, int i, int k, double* __restrict__ out) {
  constexpr int kChunk = 8;
  const int stride = cols + 2, steps = rows4 / 4;
  gram_v4d acc[NT];
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) acc[jt] = gram_v4d{0.0, 0.0, 0.0, 0.0};
  double sq = 0.0;
#pragma unroll 1
  for (int t0 = 0; t0 < steps; t0 += kChunk) {
    double av[kChunk], bv[kChunk][NT];
#pragma unroll
    for (int q = 0; q < kChunk; ++q) {
      const int r = 4 * (t0 + q) + k;
      av[q] = yrow[r < rows ? r : rows - 1];
      const double* arow = a_pad + static_cast<long long>(r < rows4 ? r : rows4 - 1) * AC + jg * 16 + i;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) bv[q][jt] = arow[jt * 16];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < kChunk; ++q) {
      const double a = (live && 4 * (t0 + q) + k < rows) ? av[q] : 0.0;
      sq = __builtin_fma(a, a, sq);
#pragma unroll
      for (int jt = 0; jt < NT; ++jt) acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv[q][jt], acc[jt], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int ip = k + 4 * reg, j = (jg + jt) * 16 + i;
      if (b0 + ip < B && j < cols) out[(b0 + ip) * stride + j] = (j < n) ? acc[jt][reg] : 0.0;
    }
  }
  if (jg == 0) {
    const double yy = add_xor32(add_xor16(sq));  // (s0 + s1) + (s2 + s3) in every lane of the four
    if (k == 0 && live) out[(b0 + i) * stride + cols] = yy;
  }
}

__global__ __launch_bounds__(256) void ridge_gram_prepass_kernel(const double* __restrict__ a_pad,
                                                                 const double* __restrict__ y, int y_stride, int rows,
                                                                 int rows4, int AC, int n, int cols, long long B,
                                                                 double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long b0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * 16;
  if (b0 >= B) return;
  const int i = lane & 15, k = lane >> 4;
  const bool live = b0 + i < B;
  const double* yrow = y + (live ? (b0 + i) : 0) * static_cast<long long>(y_stride);
  const int tiles = AC / 16;
  for (int jg = 0; jg < tiles; jg += 4) {
    switch (tiles - jg) {   // (wavefront-uniform)
      case 1: gram_prepass_group<1>(a_pad, yrow, live, rows, rows4, AC, n, cols, B, b0, jg, i, k, out); break;
      case 2: gram_prepass_group<2>(a_pad, yrow, live, rows, rows4, AC, n, cols, B, b0, jg, i, k, out); break;
      case 3: gram_prepass_group<3>(a_pad, yrow, live, rows, rows4, AC, n, cols, B, b0, jg, i, k, out); break;
      default: gram_prepass_group<4>(a_pad, yrow, live, rows, rows4, AC, n, cols, B, b0, jg, i, k, out); break;
    }
  }
}
#else
// (the first form, kept as the A/B reference)
__global__ __launch_bounds__(256) void ridge_gram_prepass_kernel(const double* __restrict__ a_pad,
                                                                 const double* __restrict__ y, int y_stride, int rows,
                                                                 int rows4, int AC, int n, int cols, long long B,
                                                                 double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const long long b0 = (static_cast<long long>(blockIdx.x) * 4 + wave) * 16;
  if (b0 >= B) return;
  const int i = lane & 15, k = lane >> 4;
  const bool live = b0 + i < B;
  const double* yrow = y + (live ? (b0 + i) : 0) * static_cast<long long>(y_stride);
  const int stride = cols + 2;
  const int tiles = AC / 16;
  for (int jg = 0; jg < tiles; jg += 4) {
    gram_v4d acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = gram_v4d{0.0, 0.0, 0.0, 0.0};
    double sq = 0.0;
    for (int t = 0; t < rows4 / 4; ++t) {
      const int r = 4 * t + k;
      const double a = (live && r < rows) ? yrow[r] : 0.0;
      sq = __builtin_fma(a, a, sq);
#pragma unroll
      for (int jt = 0; jt < 4; ++jt) {
        if (jg + jt < tiles) {   // (wavefront-uniform)
          const double b = a_pad[static_cast<long long>(r) * AC + (jg + jt) * 16 + i];
          acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[jt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int ip = k + 4 * reg, j = (jg + jt) * 16 + i;
        if (jg + jt < tiles && b0 + ip < B && j < cols) out[(b0 + ip) * stride + j] = (j < n) ? acc[jt][reg] : 0.0;
      }
    }
    if (jg == 0) {
      const double yy = add_xor32(add_xor16(sq));  // (s0 + s1) + (s2 + s3) in every lane of the four
      if (k == 0 && live) out[(b0 + i) * stride + cols] = yy;
    }
  }
}
#endif  // MI355_GRAM_PREPASS_DIRECT

// Own-matrix pre-pass: one workgroup (four wavefronts) per problem.  data row b = A_b (rows x n, row major, as the caller
// holds it: no padding) then y_b.  out row b = c_b padded to P, yy_b, one pad word, then G_b [P][P]: the same chains as
// the shared-matrix kernels above — G_b(i, j) = ascending fused chain over the rows from 0 (+ lambda on the diagonal),
// c_b(j) = ascending fused chain of y_r A_rj, yy_b = four interleaved chains (r mod 4) added pairwise.
//
// A wavefront owns 32 x 32 blocks of G_b — 2 x 2 matrix-core tiles, so two A-operand and two B-operand loads feed four
// MFMAs (one load per MFMA instead of two; a diagonal block loads two) — and walks the rows in chunks of 4 kChunk with
// EVERY operand load of a chunk issued before its first MFMA: the first form of this kernel (one tile at a time, load ->
// wait -> MFMA) exposed a full memory round trip per k-step and ran at 1.0 TB/s of reads: 4.25 ms for 65 536 problems of
// 128 x 64, against 2.2 ms for this one (chunks of 8 k-steps; 4 and 16 measured 0.2 / 0.5 ms behind:
// profiles/r4_ab_own_prepass.txt).
// Per tile the MFMA sequence — k-steps in ascending order — is unchanged, so every entry of G_b keeps its bits.
#ifndef MI355_GRAM_OWN_CHUNK
#define MI355_GRAM_OWN_CHUNK 8
#endif
#ifndef MI355_GRAM_OWN_PREPASS_DIRECT
__global__ __launch_bounds__(256) void ridge_gram_own_prepass_kernel(const double* __restrict__ data, long long data_stride,
                                                                     int rows, int n, int P, double lambda, long long B,
                                                                     double* __restrict__ out) {
  constexpr int kChunk = MI355_GRAM_OWN_CHUNK;
  const long long prob = blockIdx.x;
  if (prob >= B) return;
  const double* A = data + prob * data_stride;
  const double* y = A + static_cast<long long>(rows) * n;
  double* row = out + prob * (static_cast<long long>(P) * P + P + 2);
  double* G = row + P + 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, k = lane >> 4;
  const int nb = (P + 31) / 32, steps = ((rows + 3) & ~3) / 4;
  for (int blk = wave; blk < nb * nb; blk += 4) {
    const int bi = blk / nb, bj = blk % nb;
    const bool diag = (bi == bj);
    const int ci0 = bi * 32 + i, ci1 = ci0 + 16, cj0 = bj * 32 + i, cj1 = cj0 + 16;
    const int ki0 = ci0 < n ? ci0 : n - 1, ki1 = ci1 < n ? ci1 : n - 1, kj0 = cj0 < n ? cj0 : n - 1, kj1 = cj1 < n ? cj1 : n - 1;
    gram_v4d acc00 = gram_v4d{0.0, 0.0, 0.0, 0.0}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
#pragma unroll 1
    for (int t0 = 0; t0 < steps; t0 += kChunk) {
      double a0[kChunk], a1[kChunk], b0[kChunk], b1[kChunk];
      // load phase: unconditional loads from clamped addresses (a predicated load costs a branch and a wait each),
      // nothing consumed before the barrier; the zeroing of out-of-range rows / columns happens at the MFMA
#pragma unroll
      for (int q = 0; q < kChunk; ++q) {
        const int r = 4 * (t0 + q) + k;
        const double* ar = A + static_cast<long long>(r < rows ? r : 0) * n;
        a0[q] = ar[ki0];
        a1[q] = ar[ki1];
      }
      if (!diag) {
#pragma unroll
        for (int q = 0; q < kChunk; ++q) {
          const int r = 4 * (t0 + q) + k;
          const double* ar = A + static_cast<long long>(r < rows ? r : 0) * n;
          b0[q] = ar[kj0];
          b1[q] = ar[kj1];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < kChunk; ++q) {
        const bool rv = 4 * (t0 + q) + k < rows;   // (k-steps past the last row multiply zeros: fma(0, 0, acc) = acc, acc is never -0)
        const double x0 = (rv && ci0 < n) ? a0[q] : 0.0, x1 = (rv && ci1 < n) ? a1[q] : 0.0;
        const double y0 = diag ? x0 : ((rv && cj0 < n) ? b0[q] : 0.0), y1 = diag ? x1 : ((rv && cj1 < n) ? b1[q] : 0.0);
        acc00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y0, acc00, 0, 0, 0);
        acc01 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, y1, acc01, 0, 0, 0);
        acc10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y0, acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, y1, acc11, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int gi0 = bi * 32 + k + 4 * reg, gi1 = gi0 + 16, gj0 = bj * 32 + i, gj1 = gj0 + 16;
      if (gi0 < P && gj0 < P) G[static_cast<long long>(gi0) * P + gj0] = (gi0 == gj0 && gi0 < n) ? acc00[reg] + lambda : acc00[reg];
      if (gi0 < P && gj1 < P) G[static_cast<long long>(gi0) * P + gj1] = acc01[reg];   // (the two off-diagonal tiles of a block never touch G's diagonal)
      if (gi1 < P && gj0 < P) G[static_cast<long long>(gi1) * P + gj0] = acc10[reg];
      if (gi1 < P && gj1 < P) G[static_cast<long long>(gi1) * P + gj1] = (gi1 == gj1 && gi1 < n) ? acc11[reg] + lambda : acc11[reg];
    }
  }
  for (int j = threadIdx.x; j < P; j += 256) {
    double acc = 0.0;
    if (j < n) {
      constexpr int kRows = 16;   // loads of sixteen rows in flight, then their chain
      int r = 0;
      for (; r + kRows <= rows; r += kRows) {
        double av[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) av[q] = A[static_cast<long long>(r + q) * n + j];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < kRows; ++q) acc = __builtin_fma(y[r + q], av[q], acc);
      }
      for (; r < rows; ++r) acc = __builtin_fma(y[r], A[static_cast<long long>(r) * n + j], acc);
    }
    row[j] = acc;
  }
  if (threadIdx.x < 64) {   // yy: chain k = lane >> 4 over the rows r = k mod 4, as the shared-matrix pre-pass
    double sq = 0.0;
    for (int t = 0; t < steps; ++t) {
      const int r = 4 * t + k;
      const double a = (r < rows) ? y[r] : 0.0;
      sq = __builtin_fma(a, a, sq);
    }
    const double yy = add_xor32(add_xor16(sq));
    if (lane == 0) {
      row[P] = yy;
      row[P + 1] = 0.0;
    }
  }
}
#else
// (the first form: one tile at a time straight from memory; kept as the A/B reference)
__global__ __launch_bounds__(256) void ridge_gram_own_prepass_kernel(const double* __restrict__ data, long long data_stride,
                                                                     int rows, int n, int P, double lambda, long long B,
                                                                     double* __restrict__ out) {
  const long long prob = blockIdx.x;
  if (prob >= B) return;
  const double* A = data + prob * data_stride;
  const double* y = A + static_cast<long long>(rows) * n;
  double* row = out + prob * (static_cast<long long>(P) * P + P + 2);
  double* G = row + P + 2;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, k = lane >> 4;
  const int tiles = (P + 15) / 16, rows4 = (rows + 3) & ~3;
  for (int tile = wave; tile < tiles * tiles; tile += 4) {
    const int it = tile / tiles, jt = tile % tiles;
    const int ci = it * 16 + i, cj = jt * 16 + i;
    gram_v4d acc = gram_v4d{0.0, 0.0, 0.0, 0.0};
    for (int t = 0; t < rows4 / 4; ++t) {
      const int r = 4 * t + k;
      const double a = (r < rows && ci < n) ? A[static_cast<long long>(r) * n + ci] : 0.0;
      const double b = (r < rows && cj < n) ? A[static_cast<long long>(r) * n + cj] : 0.0;
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int gi = it * 16 + k + 4 * reg, gj = jt * 16 + i;
      if (gi < P && gj < P) G[static_cast<long long>(gi) * P + gj] = (gi == gj && gi < n) ? acc[reg] + lambda : acc[reg];
    }
  }
  for (int j = threadIdx.x; j < P; j += 256) {
    double acc = 0.0;
    if (j < n)
      for (int r = 0; r < rows; ++r) acc = __builtin_fma(y[r], A[static_cast<long long>(r) * n + j], acc);
    row[j] = acc;
  }
  if (threadIdx.x < 64) {   // yy: chain k = lane >> 4 over the rows r = k mod 4, as the shared-matrix pre-pass
    double sq = 0.0;
    for (int t = 0; t < rows4 / 4; ++t) {
      const int r = 4 * t + k;
      const double a = (r < rows) ? y[r] : 0.0;
      sq = __builtin_fma(a, a, sq);
    }
    const double yy = add_xor32(add_xor16(sq));
    if (lane == 0) {
      row[P] = yy;
      row[P + 1] = 0.0;
    }
  }
}
#endif  // MI355_GRAM_OWN_PREPASS_DIRECT
#endif  // MI355_RIDGE_GRAM_PREPASS_TU

}  // namespace mi355
