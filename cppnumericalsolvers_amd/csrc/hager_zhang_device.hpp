// hager_zhang_device.hpp — Hager–Zhang line search on a wavefront segment.
//
// Device counterpart of the reference's linesearch/hager_zhang.h (the alternative LineSearch
// template argument of Lbfgs, lbfgs.h:41):
//   Search (State overload)          :100-116
//   SatisfiesWolfe (T1 / T2)         :128-140
//   Secant, PhiDphi                  :143-157
//   Update (U0-U3), Bisect           :163-214
//   Secant2 (S1-S4)                  :218-277
//   hzls (I0, B0-B3, main loop, L2)  :282-548
//
// The reference is written as nested calls that evaluate phi in nine places.  On a wavefront the
// segments of different problems are at different places of that control flow, and every
// evaluation site would run once per pass for whichever segments sit there.  Here the search is a
// state machine instead: one loop, ONE evaluation site (trial point, objective, directional
// derivative — the lane-parallel part), after which each segment advances its own scalar state
// (`pc`) until it needs the next sample or is done.  All bookkeeping is segment-uniform and in
// registers: the reference's growing sample history is only ever read through the current
// bracket ends, the last pushed sample and the one before it, so those are what is kept; the
// two decisions that compare history indices (Secant2 :257-259) compare sample serial numbers.
// The best-so-far point (:319-332) is remembered as (alpha, phi) and re-evaluated if it is ever
// returned — same operands, same bits — instead of holding two more vectors per lane.
#pragma once
#include "more_thuente_device.hpp"
#include "wave_primitives.hpp"

namespace mi355 {

struct HzSample {
  double alpha, phi, dphi;
  int id;  // position in the reference's history vector
};

// The search itself over an evaluator `evaluate(alpha, phi, dphi)` (phi = f(x + alpha s), dphi = g(x + alpha s).s): all the
// rest of hzls is scalar.  f carries phi(0) in and the accepted value out; alpha_out is the accepted step (the LAST call of
// `evaluate` was made at it).  Shared by the wavefront-segment form below and the workgroup kernel (lbfgs_wide_kernel.hpp).
template <class Evaluate>
__device__ __forceinline__ int hz_search_core(Evaluate&& evaluate, double& f, double& stp, const double dginit,
                                              bool& failed, double& alpha_out) {
  const double alpha_init = stp;
  constexpr double delta = 1.0 / 10.0, sigma = 9.0 / 10.0;              // :286-287
  constexpr double epsilon_k = 1e-6, gamma = 0.66, rho = 5.0, psi3 = 0.1;  // :288-291
  constexpr int maxlinesearch = 50, iterfinitemax = 60;
  constexpr double eps = 2.220446049250313e-16;

  failed = false;
  const double phi_0 = f, dphi_0 = dginit;
  if (dphi_0 >= 0.0) {  // :302 (x, f, g untouched)
    failed = true;
    return 0;
  }
  const double phi_lim = phi_0 + epsilon_k * __builtin_fabs(phi_0);

  auto wolfe = [&](const HzSample& c) {
    const bool w1 = (delta * dphi_0 >= (c.phi - phi_0) / c.alpha) && (c.dphi >= sigma * dphi_0);
    const bool w2 = ((2 * delta - 1) * dphi_0 >= c.dphi) && (c.dphi >= sigma * dphi_0) && (c.phi <= phi_lim);
    return w1 || w2;
  };
  auto secant = [](const HzSample& p, const HzSample& q) {
    return (p.alpha * q.dphi - q.alpha * p.dphi) / (q.dphi - p.dphi);
  };
  auto finite = [](const HzSample& r) { return __builtin_isfinite(r.phi) && __builtin_isfinite(r.dphi); };

  // where the scalar control flow of a segment currently is
  enum Pc : int {
    kEval,            // needs phi(alpha_eval); `stage` says which sample it is
    kBracketTop,      // head of the bracket loop (:371)
    kMainTop,         // head of the shrinking loop (:458)
    kBisectTop,       // head of Bisect's loop (:193)
    kAfterUpdate,     // an Update / Bisect finished; `ret` says who asked
    kAfterSecant2,    // :493-497
    kFinishBest,      // return the best-seen sample (:443-454, :537-547)
    kDone
  };
  enum Stage : int { kInit, kExpand, kBisect, kSecantC, kSecantC2, kMid, kCollapse, kBest };
  enum Ret : int { kRetB2, kRetS2, kRetS4, kRetL2 };

  const HzSample origin{0.0, phi_0, dphi_0, 0};
  HzSample last = origin, prev = origin;   // last pushed sample and the one pushed before it
  HzSample A = origin, B = origin;         // current bracket (history[ia], history[ib])
  HzSample a0 = origin, b0 = origin;       // the bracket Secant2 started from
  HzSample ba = origin, bb = origin;       // Bisect's running bracket
  HzSample cs = origin;                    // Secant2's first sample
  int count = 1;                           // history.size()
  bool hit = false;                        // result flag of the Update / Bisect in flight
  double best_alpha = 0.0, best_phi = phi_0;
  double c = alpha_init;                   // :336-337
  if (!(c > 0.0)) c = 1.0;
  int iter = 1, iterfinite = 0;
  int pc = kEval, stage = kInit, ret = kRetB2;
  double alpha_eval = c;
  alpha_out = 0.0;
  double phi_out = phi_0;
  int nfev = 0;

  auto push = [&](HzSample& r) {
    r.id = count++;
    prev = last;
    last = r;
  };
  auto note_best = [&](const HzSample& r) {
    if (r.alpha > 0.0 && r.phi < best_phi) {
      best_alpha = r.alpha;
      best_phi = r.phi;
    }
  };
  // accept the most recently evaluated point (its gradient is already in g)
  auto finish_last = [&](const HzSample& w) {
    alpha_out = w.alpha;
    phi_out = w.phi;
    pc = kDone;
  };
  // Update (:163-182): leaves the new bracket in (A, B) and `hit`, or starts a Bisect
  auto update = [&](const HzSample& ua, const HzSample& ub, const HzSample& uc, int who) {
    ret = who;
    hit = false;
    pc = kAfterUpdate;
    if (uc.alpha < ua.alpha || uc.alpha > ub.alpha) {  // U0
      A = ua;
      B = ub;
    } else if (uc.dphi >= 0.0) {                       // U1
      A = ua;
      B = uc;
    } else if (uc.phi <= phi_lim) {                    // U2
      A = uc;
      B = ub;
    } else {                                           // U3
      ba = ua;
      bb = uc;
      pc = kBisectTop;
    }
  };

  while (true) {
    if (pc == kEval) {
      // ---- the one evaluation site: phi(alpha) = f(x + alpha s), dphi = g(x + alpha s).s (:150-157)
      double dphi_eval;
      evaluate(alpha_eval, f, dphi_eval);
      if (stage != kBest) nfev++;  // re-forming the remembered best point is not an evaluation of the algorithm
      HzSample r{alpha_eval, f, dphi_eval, 0};

      if (stage == kInit) {                                            // :339-365
        if (!finite(r)) {
          if (iterfinite < iterfinitemax) {
            c *= psi3;
            alpha_eval = c;
            ++iterfinite;
          } else {
            failed = true;
            pc = kDone;
          }
        } else {
          push(r);
          note_best(r);
          if (wolfe(r)) finish_last(r); else pc = kBracketTop;
        }
      } else if (stage == kExpand) {                                   // B3 (:403-440)
        if (!finite(r)) {
          if (iterfinite < iterfinitemax) {
            c = (last.alpha + c) / 2.0;
            alpha_eval = c;
            ++iterfinite;
          } else {
            pc = kFinishBest;
          }
        } else {
          push(r);
          note_best(r);
          if (wolfe(r)) {
            finish_last(r);
          } else {
            ++iter;
            pc = kBracketTop;
          }
        }
      } else if (stage == kBisect) {                                   // :196-212
        push(r);
        if (wolfe(r)) {
          A = ba;
          B = r;
          hit = true;
          pc = kAfterUpdate;
        } else if (r.dphi >= 0.0) {
          A = ba;
          B = r;
          hit = false;
          pc = kAfterUpdate;
        } else {
          if (r.phi <= phi_lim) ba = r; else bb = r;
          pc = kBisectTop;
        }
      } else if (stage == kSecantC) {                                  // S1-S2 (:229-244)
        push(r);
        cs = r;
        if (wolfe(r)) finish_last(r); else update(a0, b0, r, kRetS2);
      } else if (stage == kSecantC2) {                                 // S3-S4 (:261-274)
        push(r);
        if (wolfe(r)) finish_last(r); else update(A, B, r, kRetS4);
      } else if (stage == kMid) {                                      // L2 (:499-525)
        push(r);
        note_best(r);
        if (wolfe(r)) finish_last(r); else update(A, B, r, kRetL2);
      } else if (stage == kCollapse) {                                 // :465-471
        finish_last(r);
      } else {                                                         // kBest: the remembered best sample
        alpha_out = best_alpha;
        phi_out = best_phi;
        pc = kDone;
      }
    }

    // ---- scalar control flow until the next sample is needed ------------------------------------
    while (pc != kEval && pc != kDone) {
      if (pc == kBracketTop) {                                         // :371-441
        if (iter >= maxlinesearch) {
          pc = kFinishBest;                                            // not bracketed (:443-454)
        } else if (last.dphi >= 0.0) {                                 // B1: b = last; a = latest feasible before it,
          B = last;                                                    // which is always the sample pushed before
          A = prev;
          ++iter;
          pc = kMainTop;
        } else if (last.phi > phi_lim) {                               // B2
          ba = origin;
          bb = last;
          ret = kRetB2;
          pc = kBisectTop;
        } else {                                                       // B3
          c *= rho;
          alpha_eval = c;
          iterfinite = 0;
          stage = kExpand;
          pc = kEval;
        }
      } else if (pc == kBisectTop) {                                   // :193
        if (bb.alpha - ba.alpha > eps * bb.alpha) {
          alpha_eval = (ba.alpha + bb.alpha) / 2.0;
          stage = kBisect;
          pc = kEval;
        } else {
          A = ba;
          B = bb;
          hit = false;
          pc = kAfterUpdate;
        }
      } else if (pc == kAfterUpdate) {
        if (hit) {                                                     // the hit is the last evaluated sample
          finish_last(B);
        } else if (ret == kRetB2) {                                    // :395-397
          ++iter;
          pc = kMainTop;
        } else if (ret == kRetS2) {                                    // S3 (:247-262)
          const bool moved_b = (B.id == cs.id), moved_a = (A.id == cs.id);
          double c2 = cs.alpha;
          if (moved_b)
            c2 = secant(b0, B);
          else if (moved_a)
            c2 = secant(a0, A);
          if ((moved_a || moved_b) && A.alpha <= c2 && c2 <= B.alpha) {
            alpha_eval = c2;
            stage = kSecantC2;
            pc = kEval;
          } else {
            pc = kAfterSecant2;
          }
        } else if (ret == kRetS4) {
          pc = kAfterSecant2;
        } else {                                                       // kRetL2 (:527-534)
          ++iter;
          pc = kMainTop;
        }
      } else if (pc == kAfterSecant2) {                                // :493-500
        if (B.alpha - A.alpha < gamma * (b0.alpha - a0.alpha)) {
          ++iter;
          pc = kMainTop;
        } else {
          alpha_eval = (A.alpha + B.alpha) / 2.0;
          stage = kMid;
          pc = kEval;
        }
      } else if (pc == kMainTop) {                                     // :458-476
        if (iter >= maxlinesearch) {
          pc = kFinishBest;
        } else if (B.alpha - A.alpha <= eps * B.alpha) {
          if (A.alpha > 0.0) {
            alpha_eval = A.alpha;
            stage = kCollapse;
            pc = kEval;
          } else {
            pc = kFinishBest;
          }
        } else {                                                       // Secant2, S1 (:224-229)
          a0 = A;
          b0 = B;
          double cc = secant(a0, b0);
          if (!__builtin_isfinite(cc)) cc = (a0.alpha + b0.alpha) / 2.0;
          alpha_eval = cc;
          stage = kSecantC;
          pc = kEval;
        }
      } else {                                                         // kFinishBest
        if (best_alpha > 0.0) {
          alpha_eval = best_alpha;
          stage = kBest;
          pc = kEval;
        } else {
          failed = true;
          pc = kDone;
        }
      }
    }
    if (pc == kDone) break;
  }

  stp = failed ? 0.0 : alpha_out;
  if (!failed) f = phi_out;
  return nfev;
}

// Returns the number of objective evaluations.  On success x, f, g hold the accepted point (f, g as
// evaluated there) and stp the accepted step; `failed` reports hzls' -1 exits, for which the caller
// keeps its start state (x is untouched, f and g are not) and stp is 0 — or unchanged when the
// direction is not a descent direction (:302).  d is the NEGATED direction (the search runs along
// s = -d), dginit = g.s at the start, stp carries the initial trial step in.
// AR: the arithmetic policy of the evaluations (wave_primitives.hpp).  Under ArithFma the objective and the directional
// derivative are the fused forms (eval_fma, one fma chain per lane in the inner product); the trial point stays the
// unfused `x0 - alpha d` of the exact build — what the CPU twin's Hager-Zhang evaluation computes under its butterfly_fma
// policy (the test suite's restatement of the reference), so the fused search has a bit-identical CPU twin as well.
template <int W, int E, class AR = ArithExact, class Obj>
__device__ __forceinline__ int hz_search(const Obj& obj, double (&x)[E], double& f, double (&g)[E],
                                         double& stp, const double (&d)[E], const double dginit,
                                         int n, int sl, bool& failed) {
  double wa[E];
#pragma unroll
  for (int e = 0; e < E; ++e) wa[e] = x[e];
  double alpha_acc = 0.0;
  const int nfev = hz_search_core(
      [&](double alpha, double& phi, double& dphi) {
        double xt[E];
#pragma unroll
        for (int e = 0; e < E; ++e) xt[e] = wa[e] - alpha * d[e];
        phi = obj_eval<W, E, AR>(obj, xt, g, n, sl);
        dphi = -seg_dot<W, E, AR>(g, d);
      },
      f, stp, dginit, failed, alpha_acc);
  if (!failed && nfev > 0) {
    // the accepted point, re-formed from the accepted step: same operands, same bits
#pragma unroll
    for (int e = 0; e < E; ++e) x[e] = wa[e] - alpha_acc * d[e];
  }
  return nfev;
}

}  // namespace mi355
