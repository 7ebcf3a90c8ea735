// wave_primitives.hpp — cross-lane building blocks for gfx950 (CDNA4, wave64).
//
// One L-BFGS problem lives on a *segment* of W consecutive lanes of a
// wavefront (W = 64: one problem per wavefront).  Every n-element inner
// product / norm of the reference (Eigen `.dot`, `.norm`, `.lpNorm<Infinity>`
// at solver/lbfgs.h:95,164,169,188,194,199,209,221,265,266,290,292,
// linesearch/more_thuente.h:151,201, solver/progress.h:190,195,310) becomes an
// xor-butterfly over the W lanes of the segment.
//
// fp64 has no DPP-fused add on gfx950, so one butterfly level is
//   2 x v_mov_b32_dpp (or v_permlane{16,32}_swap_b32) + 1 x v_add_f64.
// Levels:  xor1, xor2   quad_perm            (within a quad)
//          xor4         row_half_mirror      (valid because lanes of a quad
//          xor8         row_mirror            already hold identical values)
//          xor16        v_permlane16_swap    (gfx950 only)
//          xor32        v_permlane32_swap    (gfx950 only)
// __shfl_xor would lower to ds_bpermute_b32 (LDS crossbar, ~10x the latency).
//
// Because IEEE add is commutative, both partners of every exchange compute the
// same bits, so after the last level ALL lanes of the segment hold the same
// value ("segment-uniform" scalars) and the summation tree is the plain
// pairwise tree over the zero-padded power-of-two width.  That tree depends
// only on P = W*E (E = elements per lane, reduced pairwise in-lane first), so
// results are independent of the (W, E) mapping chosen.
#pragma once
#include <hip/hip_runtime.h>

namespace mi355 {

constexpr int kWave = 64;

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // old = 0 + bound_ctrl: lanes whose source is outside the row/wavefront read 0
  // and the compiler needs no copy of the input (v_mov_b32_dpp dst, src directly).
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// DPP controls (gfx9 encoding)
constexpr int kQuadXor1 = 0xB1;        // quad_perm:[1,0,3,2]
constexpr int kQuadXor2 = 0x4E;        // quad_perm:[2,3,0,1]
constexpr int kRowHalfMirror = 0x141;  // lane i <- 7-i within 8
constexpr int kRowMirror = 0x140;      // lane i <- 15-i within 16
constexpr int kWaveShl1 = 0x130;       // lane i <- i+1  (whole wavefront)
constexpr int kWaveShr1 = 0x138;       // lane i <- i-1  (whole wavefront)

// value held by the lane whose index differs in bit 4 (xor 16)
__device__ __forceinline__ double xchg16(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  // r[0]: rows (0,0,2,2) of v; r[1]: rows (1,1,3,3) of v.  The partner value
  // is whichever of the two is not our own row; a+b below never needs to know.
  const int lane = __lane_id();
  const bool odd_row = (lane >> 4) & 1;
  return odd_row ? __hiloint2double(r1[0], r0[0]) : __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ double xchg32(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const int lane = __lane_id();
  const bool upper = lane >= 32;
  return upper ? __hiloint2double(r1[0], r0[0]) : __hiloint2double(r1[1], r0[1]);
}
// a + partner(a) for the two swap levels without the select: the swap leaves
// {lower copy, upper copy} in the two result registers and a+b is commutative.
__device__ __forceinline__ double add_xor16(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ double add_xor32(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(r1[0], r0[0]) + __hiloint2double(r1[1], r0[1]);
}
__device__ __forceinline__ double max_xor16(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto r0 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __builtin_fmax(__hiloint2double(r1[0], r0[0]), __hiloint2double(r1[1], r0[1]));
}
__device__ __forceinline__ double max_xor32(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __builtin_fmax(__hiloint2double(r1[0], r0[0]), __hiloint2double(r1[1], r0[1]));
}

// The xor16 / xor32 partner through the LDS crossbar instead of the VALU:
// v_permlane{16,32}_swap cost ~10 VALU cycles each (scripts/microbench/valu_rates.hip:
// a swap level is ~33 cycles of VALU issue against ~12.5 for a DPP level), while
// ds_swizzle / ds_bpermute are issued on the LDS pipe and only the add stays on the VALU.
// They do not touch LDS memory.  Selected at compile time with MI355_XCHG_VIA_DS.
#ifndef MI355_XCHG_VIA_DS
#define MI355_XCHG_VIA_DS 1
#endif
__device__ __forceinline__ double partner_xor16_ds(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  // bit-mask mode: and_mask 0x1f, or_mask 0, xor_mask 0x10 (swaps the two 16-lane halves of
  // each group of 32 lanes)
  lo = __builtin_amdgcn_ds_swizzle(lo, 0x401F);
  hi = __builtin_amdgcn_ds_swizzle(hi, 0x401F);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double partner_xor32_ds(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  const int addr = (__lane_id() ^ 32) << 2;
  lo = __builtin_amdgcn_ds_bpermute(addr, lo);
  hi = __builtin_amdgcn_ds_bpermute(addr, hi);
  return __hiloint2double(hi, lo);
}

// Sum over the W lanes of the caller's segment; result in every lane.
template <int W>
__device__ __forceinline__ double seg_sum(double v) {
  static_assert(W == 1 || W == 2 || W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "W");
  if constexpr (W >= 2) v = v + dpp_mov<kQuadXor1>(v);
  if constexpr (W >= 4) v = v + dpp_mov<kQuadXor2>(v);
  if constexpr (W >= 8) v = v + dpp_mov<kRowHalfMirror>(v);
  if constexpr (W >= 16) v = v + dpp_mov<kRowMirror>(v);
#if MI355_XCHG_VIA_DS
  // measured (profiles/r1_mapping_sweep.txt): the LDS-crossbar exchange pays at W = 64,
  // where two swap levels per butterfly make the VALU the bottleneck; at W = 32 the single
  // swap level is cheaper than the extra LDS round trip.
  if constexpr (W == 32) v = add_xor16(v);
  if constexpr (W >= 64) v = v + partner_xor16_ds(v);
  if constexpr (W >= 64) v = v + partner_xor32_ds(v);
#else
  if constexpr (W >= 32) v = add_xor16(v);
  if constexpr (W >= 64) v = add_xor32(v);
#endif
  return v;
}
// maxNum of two doubles as ONE v_max_f64.  __builtin_fmax on a value that arrived through the integer DPP moves
// makes the compiler insert a canonicalising v_max_f64 x, x, x first (it cannot prove the bits are not a signalling
// NaN): a quarter of every max-butterfly level.  No signalling NaN can arise on this path — every NaN here is the
// quiet result of an arithmetic instruction — so the bare instruction returns the same bits.
__device__ __forceinline__ double vmax(double a, double b) {
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// minNum as ONE v_min_f64 (same reasoning; on ordered, non-zero inputs the bits of `(b < a) ? b : a`)
__device__ __forceinline__ double vmin(double a, double b) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// Max over the W lanes of the segment (inputs are |.| values, never NaN-ordered).
template <int W>
__device__ __forceinline__ double seg_max(double v) {
  if constexpr (W >= 2) v = vmax(v, dpp_mov<kQuadXor1>(v));
  if constexpr (W >= 4) v = vmax(v, dpp_mov<kQuadXor2>(v));
  if constexpr (W >= 8) v = vmax(v, dpp_mov<kRowHalfMirror>(v));
  if constexpr (W >= 16) v = vmax(v, dpp_mov<kRowMirror>(v));
#if MI355_XCHG_VIA_DS
  if constexpr (W == 32) v = max_xor16(v);
  if constexpr (W >= 64) v = __builtin_fmax(v, partner_xor16_ds(v));
  if constexpr (W >= 64) v = __builtin_fmax(v, partner_xor32_ds(v));
#else
  if constexpr (W >= 32) v = max_xor16(v);
  if constexpr (W >= 64) v = max_xor32(v);
#endif
  return v;
}

// In-lane pairwise tree over E values (E = 1, 2, 4).
template <int E>
__device__ __forceinline__ double lane_tree_sum(const double (&t)[E]) {
  static_assert(E == 1 || E == 2 || E == 4 || E == 8 || E == 16, "E");
  if constexpr (E == 1) {
    return t[0];
  } else {
    double h[E / 2];
#pragma unroll
    for (int i = 0; i < E / 2; ++i) h[i] = t[2 * i] + t[2 * i + 1];
    return lane_tree_sum<E / 2>(h);
  }
}
template <int E>
__device__ __forceinline__ double lane_max(const double (&t)[E]) {
  double m = t[0];
#pragma unroll
  for (int e = 1; e < E; ++e) m = __builtin_fmax(m, t[e]);
  return m;
}

// Arithmetic policies of the L-BFGS solve kernels (mi355_lbfgs_desc.arithmetic).
//   ArithExact  every a*b+c is a rounded product followed by a rounded sum (the library is built with
//               -ffp-contract=off, so the compiler fuses nothing): the operation order of the reference's
//               scalar code, bit-identical to the oracle's butterfly policy.  The pinning mode.
//   ArithFma    the multiply-adds of the inner products' in-lane part, of the two-loop recursion's axpys, of the
//               line search's trial point and of the objective are single v_fma_f64 — what GCC/Clang make of the
//               reference's own loops at -O3 -march=native (contraction is on by default there).  An in-lane
//               inner product becomes ONE chain  fma(a_E-1, b_E-1, ... fma(a_1, b_1, a_0 b_0))  over the lane's E
//               coordinates (E instructions instead of 2E - 1) followed by the same butterfly, so the summation
//               tree now depends on E as well.  The test suite's CPU twin has a policy that restates exactly this
//               (`butterfly_fma`), so the device stays bit-identical to a CPU twin in this mode too;
//               against the reference-order solve results agree within the 1e-6 tolerance.
struct ArithExact {
  static constexpr bool kFma = false;
  static __device__ __forceinline__ double madd(double a, double b, double c) { return a * b + c; }
  static __device__ __forceinline__ double nmadd(double a, double b, double c) { return c - a * b; }
};
struct ArithFma {
  static constexpr bool kFma = true;
  static __device__ __forceinline__ double madd(double a, double b, double c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ double nmadd(double a, double b, double c) { return __builtin_fma(-a, b, c); }
};

// fused in-lane inner product: one chain over up to kFmaGroup = 4 coordinates; a lane with more coordinates (E = 8)
// adds the chains of its groups of four pairwise — the first levels of the same tree the butterfly continues, so
// E = 8 and E = 4 give the same bits
constexpr int kFmaGroup = 4;
template <int E, int E0 = 0, int N = E>
__device__ __forceinline__ double lane_fma_dot(const double (&a)[E], const double (&b)[E]) {
  if constexpr (N <= kFmaGroup) {
    double t = a[E0] * b[E0];
#pragma unroll
    for (int e = 1; e < N; ++e) t = __builtin_fma(a[E0 + e], b[E0 + e], t);
    return t;
  } else {
    return lane_fma_dot<E, E0, N / 2>(a, b) + lane_fma_dot<E, E0 + N / 2, N / 2>(a, b);
  }
}

template <int W, int E, class AR = ArithExact>
__device__ __forceinline__ double seg_dot(const double (&a)[E], const double (&b)[E]) {
  if constexpr (AR::kFma) {
    return seg_sum<W>(lane_fma_dot<E>(a, b));
  } else {
    double t[E];
#pragma unroll
    for (int e = 0; e < E; ++e) t[e] = a[e] * b[e];
    return seg_sum<W>(lane_tree_sum<E>(t));
  }
}

// objective evaluation under an arithmetic policy: objectives that offer a fused form define eval_fma
template <int W, int E, class AR, class Obj>
__device__ __forceinline__ double obj_eval(const Obj& obj, const double (&x)[E], double (&g)[E], int n, int sl) {
  if constexpr (AR::kFma) {
    return obj.template eval_fma<W, E>(x, g, n, sl);
  } else {
    return obj.template eval<W, E>(x, g, n, sl);
  }
}
// max(|a|, |b|) as one instruction (source modifiers), see vmax
__device__ __forceinline__ double vmax_abs(double a, double b) {
  double r;
  asm("v_max_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int W, int E>
__device__ __forceinline__ double seg_amax(const double (&a)[E]) {
  double m;
  if constexpr (E == 1) {
    m = vmax_abs(a[0], a[0]);
  } else {
    m = vmax_abs(a[0], a[1]);
#pragma unroll
    for (int e = 2; e + 1 < E; e += 2) m = vmax(m, vmax_abs(a[e], a[e + 1]));
    if constexpr (E % 2 == 1) m = vmax(m, vmax_abs(a[E - 1], a[E - 1]));
  }
  return seg_max<W>(m);
}

// Value of `v` in the next / previous lane of the wavefront (lane 63 / lane 0
// read 0; callers mask those positions).
__device__ __forceinline__ double from_next_lane(double v) { return dpp_mov<kWaveShl1>(v); }
__device__ __forceinline__ double from_prev_lane(double v) { return dpp_mov<kWaveShr1>(v); }

// Broadcast a 32-bit value from the first lane of the caller's segment.
// Coordinate j of the problem's x on every lane of its segment (for functors whose formula mixes a few named
// coordinates): the owner contributes x_j, everyone else +0.0, through the segment sum -- exactly x_j (a -0.0 arrives
// as +0.0).
template <int W, int E>
__device__ __forceinline__ double seg_coordinate(const double (&x)[E], int j, int sl) {
  double v = 0.0;
#pragma unroll
  for (int e = 0; e < E; ++e) v = (sl * E + e == j) ? x[e] : v;
  return seg_sum<W>(v);
}

template <int W>
__device__ __forceinline__ int seg_bcast_first(int v) {
  if constexpr (W == 64) {
    return __builtin_amdgcn_readfirstlane(v);
  } else {
    const int lane = __lane_id();
    return __shfl(v, lane & ~(W - 1), kWave);
  }
}

}  // namespace mi355
