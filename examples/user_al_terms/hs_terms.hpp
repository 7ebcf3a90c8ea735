// hs_terms.hpp — USER functors as augmented-Lagrangian terms (worked example; MI355_AL_TERM_USER).
//
// The reference's non-convex augmented-Lagrangian tests are written with three-line user classes
// (src/test/augmented_lagrangian_test.cc:945-962 Hs024Objective, :1090-1100 ProductObjective3D, :1103-1113
// Hs029Ellipse) handed to ConstrainedOptimizationProblem.  On the device the same functions are functors with the
// interface of cppnumericalsolvers_amd/csrc/objectives.hpp that own no LDS; a build of the library compiles them in as
// terms,
//     _build.build(output=".../libmi355_lbfgs_hs.so", al_dims=(2,),
//                  user_objectives=[dict(name="hs024_objective", header=<this file>, type="user_examples::Hs024Objective",
//                                        id=100, al_term=True, objective=False), ...])
// and a row of mi355_al_problem::kinds with the functor's id evaluates it (its parameters: the row's n + 1
// coefficients; unused by these three).  A problem is owned by W lanes, lane `sl` holds coordinates j = sl*E + e;
// mi355::seg_coordinate hands a named coordinate to all of them.  Operation order = the reference classes', so that the
// CPU twin (oracle/auglag_oracle.hpp, kinds 100-102) and the reference itself (oracle/ref_auglag_capi.cpp) can be pinned
// against the device bit for bit / to 1e-6.
#pragma once

namespace user_examples {

struct TermBase {  // a term owns no LDS and reads no per-problem data
  static constexpr int kLdsDoubles = 0;
  __host__ __device__ static constexpr int shared_lds_doubles() { return 0; }
  __device__ __forceinline__ void load(const double*, int, int, double*, double*) {}
  __device__ __forceinline__ void begin_problem(const double*, long long, int, int) {}
};

// f(x) = ((x0 - 3)^2 - 9) x1^3 / (27 sqrt 3)                                      (Hock-Schittkowski 24)
struct Hs024Objective : TermBase {
  template <int W, int E>
  __device__ __forceinline__ double eval(const double (&x)[E], double (&g)[E], int, int sl) const {
    const double x0 = mi355::seg_coordinate<W, E>(x, 0, sl), x1 = mi355::seg_coordinate<W, E>(x, 1, sl);
    const double bracket = (x0 - 3.0) * (x0 - 3.0) - 9.0;
    const double scale = 1.0 / (27.0 * 1.7320508075688772);  // 1 / (27 sqrt(3)), sqrt correctly rounded
    const double g0 = (((2.0 * (x0 - 3.0)) * x1) * x1) * x1 * scale;
    const double g1 = (((3.0 * bracket) * x1) * x1) * scale;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      g[e] = (j == 0) ? g0 : ((j == 1) ? g1 : 0.0);
    }
    return (((bracket * x1) * x1) * x1) * scale;
  }
};

// f(x) = -x0 x1                                                                   (the 2-D Hock-Schittkowski 29)
struct ProductObjective : TermBase {
  template <int W, int E>
  __device__ __forceinline__ double eval(const double (&x)[E], double (&g)[E], int, int sl) const {
    const double x0 = mi355::seg_coordinate<W, E>(x, 0, sl), x1 = mi355::seg_coordinate<W, E>(x, 1, sl);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      g[e] = (j == 0) ? -x1 : ((j == 1) ? -x0 : 0.0);
    }
    return (-x0) * x1;
  }
};

// c(x) = 48 - x0^2 - 2 x1^2  (>= 0)
struct Hs029Ellipse : TermBase {
  template <int W, int E>
  __device__ __forceinline__ double eval(const double (&x)[E], double (&g)[E], int, int sl) const {
    const double x0 = mi355::seg_coordinate<W, E>(x, 0, sl), x1 = mi355::seg_coordinate<W, E>(x, 1, sl);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const int j = sl * E + e;
      g[e] = (j == 0) ? -2.0 * x0 : ((j == 1) ? -4.0 * x1 : 0.0);
    }
    return (48.0 - x0 * x0) - (2.0 * x1) * x1;
  }
};

}  // namespace user_examples
