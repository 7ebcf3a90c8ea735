// more_thuente_device.hpp — Moré–Thuente line search on a wavefront segment.
//
// Device counterpart of the reference's linesearch/more_thuente.h:
//   Search (State overload)  :120-135
//   cvsrch                   :137-256
//   cstep                    :261-407
// All step-length bookkeeping (stx, fx, dgx, sty, ... ) is scalar and
// segment-uniform: every lane of the segment carries the same value and takes
// the same branch, so the control flow below is the reference's own, executed
// redundantly per lane; only the trial point x = wa + stp*s, the objective and
// the directional derivative dg = g.s are lane-parallel.
#pragma once
#include "wave_primitives.hpp"

namespace mi355 {

__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; }   // std::min
__device__ __forceinline__ double dmax(double a, double b) { return (a < b) ? b : a; }   // std::max
__device__ __forceinline__ double dclamp(double v, double lo, double hi) {               // std::clamp
  return (v < lo) ? lo : ((hi < v) ? hi : v);
}
__device__ __forceinline__ double max_abs3(double x, double y, double z) {  // more_thuente.h:409-411
  return dmax(__builtin_fabs(x), dmax(__builtin_fabs(y), __builtin_fabs(z)));
}

// Interval state updated by cstep (all by value: keeps everything in registers).
struct StepInterval {
  double stx, fx, dx;  // best step so far, its value and derivative
  double sty, fy, dy;  // other end of the interval of uncertainty
  double stp;          // current / next trial step
  bool brackt;
  int info;
  int rc;              // cstep's return value (0, or -1 on invalid input)
};

#ifdef MI355_CSTEP_SELECT
// more_thuente.h:261-407.  fp, dp are the value/derivative at the trial stp;
// stpmin/stpmax are the bracket bounds computed by cvsrch.  rc = 0, or -1 on
// invalid input (info = 0, nothing else changed).
//
// A/B build -DMI355_CSTEP_SELECT.  The reference's four cases share one shape — a cubic step through (theta, s, gamma, p / q) and a
// secant / quadratic step — and differ in which end of the interval and which derivative enter it.
// The segments of a wavefront take different cases in the same call, and a divergent `if` chain
// executes every case that any segment takes (seven double-precision divisions and a square root
// each).  Here the operands are selected first and the arithmetic is issued once: every operation
// below is the reference's own operation on the reference's own operands for the case the segment
// is in, so the results are the same bits; the values computed for a case that does not use them
// (case 4 without a bracket) are discarded.
__device__ __forceinline__ StepInterval mt_cstep(StepInterval in, const double fp, const double dp,
                                                 const double stpmin, const double stpmax) {
  double stx = in.stx, fx = in.fx, dx = in.dx, sty = in.sty, fy = in.fy, dy = in.dy, stp = in.stp;
  bool brackt = in.brackt;
  if ((brackt && ((stp <= dmin(stx, sty)) || (stp >= dmax(stx, sty)))) ||
      (dx * (stp - stx) >= 0.0) || (stpmax < stpmin)) {
    in.info = 0;
    in.rc = -1;
    return in;
  }
  // dx / |dx| (:291) is +-1 for every finite non-zero dx (dx == 0 left through the test above); an
  // infinite or NaN dx gives NaN, as the quotient does.
  const double unit = (__builtin_fabs(dx) < __builtin_inf()) ? __builtin_copysign(1.0, dx) : __builtin_nan("");
  const double sgnd = dp * unit;
  const bool c1 = fp > fx;                                                   // :294
  const bool c2 = !c1 && (sgnd < 0.0);                                       // :312
  const bool c3 = !c1 && !c2 && (__builtin_fabs(dp) < __builtin_fabs(dx));   // :330
  const bool c4 = !c1 && !c2 && !c3;                                         // :359
  const int info = c1 ? 1 : (c2 ? 2 : (c3 ? 3 : 4));
  const bool bound = c1 | c3;

  // the end of the interval the cubic runs to: (stx, fx, dx) in cases 1-3, (sty, fy, dy) in case 4
  const double sta = c4 ? sty : stx;
  const double da = c4 ? dy : dx;
  const double num = c4 ? (fp - fy) : (fx - fp);      // :296 / :362, in the reference's orientation
  const double den = c4 ? (sty - stp) : (stp - stx);
  const double theta = 3.0 * num / den + da + dp;
  const double s = max_abs3(theta, da, dp);
  double rad = (theta / s) * (theta / s) - (da / s) * (dp / s);
  if (c3) rad = dmax(0.0, rad);                       // :335
  double gamma = s * __builtin_sqrt(rad);
  if (c1 ? (stp < stx) : (stp > sta)) gamma = -gamma;
  const double u = c1 ? dx : dp;
  const double v = c1 ? dp : da;
  const double gm = gamma - u;
  const double p = gm + theta;
  const double q = c3 ? ((gamma + (dx - dp)) + gamma) : ((gm + gamma) + v);
  const double r = p / q;
  const double base = c1 ? stx : stp;
  const double diff = c1 ? (stp - stx) : (sta - stp);
  double stpc = base + r * diff;
  if (c3 & !((r < 0.0) & (gamma != 0.0))) stpc = (stp > stx) ? stpmax : stpmin;   // :342-348
  // quadratic (case 1) / secant (cases 2, 3) step
  const double t1 = (fx - fp) / (stp - stx) + dx;
  double quo = (c1 ? dx : dp) / (c1 ? t1 : (dp - dx));
  if (c1) quo = quo / 2.0;
  const double stpq = base + quo * diff;
  const double ac = __builtin_fabs(stpc - base), aq = __builtin_fabs(stpq - base);
  double stpf;
  if (c1) {
    stpf = (ac < aq) ? stpc : stpc + (stpq - stpc) / 2.0;
  } else if (c2) {
    stpf = (ac > aq) ? stpc : stpq;
  } else if (c3) {
    stpf = brackt ? ((ac < aq) ? stpc : stpq) : ((ac > aq) ? stpc : stpq);
  } else {
    stpf = brackt ? stpc : ((stp > stx) ? stpmax : stpmin);
  }
  brackt = brackt | c1 | c2;
  // Update the interval of uncertainty.
  if (c1) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stpf = dclamp(stpf, stpmin, stpmax);
  stp = stpf;
  if (brackt & bound) {
    if (sty > stx) {
      stp = dmin(stx + 0.66 * (sty - stx), stp);
    } else {
      stp = dmax(stx + 0.66 * (sty - stx), stp);
    }
  }
  StepInterval out;
  out.stx = stx; out.fx = fx; out.dx = dx;
  out.sty = sty; out.fy = fy; out.dy = dy;
  out.stp = stp;
  out.brackt = brackt;
  out.info = info;
  out.rc = 0;
  return out;
}

#else
// The reference's own control flow (the default: measured 1 % faster than the operand-select form above on every
// workload, profiles/r2_ab_cstep_select.txt — a wavefront's segments rarely need different cases in the same call).
// more_thuente.h:261-407.  fp, dp are the value/derivative at the trial stp;
// stpmin/stpmax are the bracket bounds computed by cvsrch.  rc = 0, or -1 on
// invalid input (info = 0, nothing else changed).
__device__ __forceinline__ StepInterval mt_cstep(StepInterval in, const double fp, const double dp,
                                                 const double stpmin, const double stpmax) {
  double stx = in.stx, fx = in.fx, dx = in.dx, sty = in.sty, fy = in.fy, dy = in.dy, stp = in.stp;
  bool brackt = in.brackt;
  int info;
  info = 0;
  bool bound = false;
  if ((brackt && ((stp <= dmin(stx, sty)) || (stp >= dmax(stx, sty)))) ||
      (dx * (stp - stx) >= 0.0) || (stpmax < stpmin)) {
    in.info = 0;
    in.rc = -1;
    return in;
  }
  const double sgnd = dp * (dx / __builtin_fabs(dx));
  double stpf = 0.0, stpc = 0.0, stpq = 0.0;
  if (fp > fx) {  // Case 1: higher function value -> minimum is bracketed.
    info = 1;
    bound = true;
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = max_abs3(theta, dx, dp);
    double gamma = s * __builtin_sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta;
    const double q = ((gamma - dx) + gamma) + dp;
    const double r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (__builtin_fabs(stpc - stx) < __builtin_fabs(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / 2.0;
    brackt = true;
  } else if (sgnd < 0.0) {  // Case 2: derivatives change sign -> bracketed.
    info = 2;
    bound = false;
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = max_abs3(theta, dx, dp);
    double gamma = s * __builtin_sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = ((gamma - dp) + gamma) + dx;
    const double r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (__builtin_fabs(stpc - stp) > __builtin_fabs(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = true;
  } else if (__builtin_fabs(dp) < __builtin_fabs(dx)) {  // Case 3: |derivative| decreases.
    info = 3;
    bound = true;
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double s = max_abs3(theta, dx, dp);
    double gamma = s * __builtin_sqrt(dmax(0.0, (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta;
    const double q = (gamma + (dx - dp)) + gamma;
    const double r = p / q;
    if ((r < 0.0) & (gamma != 0.0)) {
      stpc = stp + r * (stx - stp);
    } else if (stp > stx) {
      stpc = stpmax;
    } else {
      stpc = stpmin;
    }
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      stpf = (__builtin_fabs(stp - stpc) < __builtin_fabs(stp - stpq)) ? stpc : stpq;
    } else {
      stpf = (__builtin_fabs(stp - stpc) > __builtin_fabs(stp - stpq)) ? stpc : stpq;
    }
  } else {  // Case 4: |derivative| does not decrease.
    info = 4;
    bound = false;
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double s = max_abs3(theta, dy, dp);
      double gamma = s * __builtin_sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta;
      const double q = ((gamma - dp) + gamma) + dy;
      const double r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  // Update the interval of uncertainty.
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < 0.0) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stpf = dclamp(stpf, stpmin, stpmax);
  stp = stpf;
  if (brackt & bound) {
    if (sty > stx) {
      stp = dmin(stx + 0.66 * (sty - stx), stp);
    } else {
      stp = dmax(stx + 0.66 * (sty - stx), stp);
    }
  }
  StepInterval out;
  out.stx = stx; out.fx = fx; out.dx = dx;
  out.sty = sty; out.fy = fy; out.dy = dy;
  out.stp = stp;
  out.brackt = brackt;
  out.info = info;
  out.rc = 0;
  return out;
}

#endif

// more_thuente.h:137-256 with the State-overload prologue of :120-135.
// In:  x = start point, f/g = value/gradient there, d = NEGATED search direction
//      (the line search runs along s = -d, lbfgs.h:231-232), stp = initial step,
//      dginit = g.s.   Out: x, f, g at the last evaluated trial.
// Working with d instead of a separate s = -d saves E register pairs and is exact:
// wa + stp*(-d) == wa - stp*d and g.(-d) == -(g.d) bit for bit (negation commutes
// with IEEE rounding).
// Returns the number of objective evaluations performed.
template <int W, int E, class Obj, class AR = ArithExact>
__device__ __forceinline__ int mt_cvsrch(const Obj& obj, double (&x)[E], double& f, double (&g)[E],
                                         double stp, const double (&d)[E], const double dginit,
                                         int n, int sl) {
  int info = 0;
  int infoc = 1;
  constexpr double xtol = 1e-15;
  constexpr double ftol = 1e-4;
  constexpr double gtol = 0.9;
  constexpr double stpmin = 1e-15;
  constexpr double stpmax = 1e15;
  constexpr double xtrapf = 4.0;
  constexpr int maxfev = 20;
  int nfev = 0;

  // dginit = g.s (:151) is supplied by the caller, which already holds it.
  if (dginit >= 0.0) return 0;  // no descent direction: x, f, g untouched (:152-156)

  bool brackt = false;
  bool stage1 = true;
  const double finit = f;
  const double dgtest = ftol * dginit;
  double width = stpmax - stpmin;
  double width1 = 2.0 * width;
  double wa[E];
#pragma unroll
  for (int e = 0; e < E; ++e) wa[e] = x[e];

  double stx = 0.0, fx = finit, dgx = dginit;
  double sty = 0.0, fy = finit, dgy = dginit;
  double stmin, stmax;

  while (true) {
    if (brackt) {
      stmin = dmin(stx, sty);
      stmax = dmax(stx, sty);
    } else {
      stmin = stx;
      stmax = stp + xtrapf * (stp - stx);
    }
    stp = dclamp(stp, stpmin, stpmax);
    if ((brackt && ((stp <= stmin) || (stp >= stmax))) || (nfev >= maxfev - 1) || (infoc == 0) ||
        (brackt && ((stmax - stmin) <= (xtol * stmax)))) {
      stp = stx;
    }
    {
      // The trial point lives only for the evaluation; the accepted one is re-formed after
      // the loop.  That keeps E doubles per lane out of the step-selection arithmetic below,
      // which is where the kernel's register budget (waves per SIMD) is set.
#ifdef MI355_NO_XRECOMP
#pragma unroll
      for (int e = 0; e < E; ++e) x[e] = AR::nmadd(stp, d[e], wa[e]);  // wa + stp * s
      f = obj_eval<W, E, AR>(obj, x, g, n, sl);
#else
      double xt[E];
#pragma unroll
      for (int e = 0; e < E; ++e) xt[e] = AR::nmadd(stp, d[e], wa[e]);  // wa + stp * s
      f = obj_eval<W, E, AR>(obj, xt, g, n, sl);
#endif
    }
    nfev++;
    const double dg = -seg_dot<W, E, AR>(g, d);  // g.s
    const double ftest1 = finit + stp * dgtest;

    if ((brackt & ((stp <= stmin) | (stp >= stmax))) | (infoc == 0)) info = 6;
    if ((stp == stpmax) & (f <= ftest1) & (dg <= dgtest)) info = 5;
    if ((stp == stpmin) & ((f > ftest1) | (dg >= dgtest))) info = 4;
    if (nfev >= maxfev) info = 3;
    if (brackt & (stmax - stmin <= xtol * stmax)) info = 2;
    if ((f <= ftest1) & (__builtin_fabs(dg) <= gtol * (-dginit))) info = 1;
    if (info != 0) break;

    if (stage1 & (f <= ftest1) & (dg >= dmin(ftol, gtol) * dginit)) stage1 = false;

    // :225-244.  During stage 1 the step is computed on the modified function
    // psi(a) = f(a) - f(0) - a*dgtest; both branches call the same cstep, so
    // the arguments are selected first and cstep is instantiated once.
    const bool modified = stage1 & (f <= fx) & (f > ftest1);
    StepInterval iv;
    iv.stx = stx; iv.sty = sty; iv.stp = stp; iv.brackt = brackt; iv.info = infoc; iv.rc = 0;
    iv.fx = modified ? fx - stx * dgtest : fx;
    iv.fy = modified ? fy - sty * dgtest : fy;
    iv.dx = modified ? dgx - dgtest : dgx;
    iv.dy = modified ? dgy - dgtest : dgy;
    const double fm = modified ? f - stp * dgtest : f;
    const double dgm = modified ? dg - dgtest : dg;
    iv = mt_cstep(iv, fm, dgm, stmin, stmax);
    stx = iv.stx; sty = iv.sty; stp = iv.stp; brackt = iv.brackt; infoc = iv.info;
    fx = modified ? iv.fx + stx * dgtest : iv.fx;
    fy = modified ? iv.fy + sty * dgtest : iv.fy;
    dgx = modified ? iv.dx + dgtest : iv.dx;
    dgy = modified ? iv.dy + dgtest : iv.dy;
    if (brackt) {
      if (__builtin_fabs(sty - stx) >= 0.66 * width1) stp = stx + 0.5 * (sty - stx);
      width1 = width;
      width = __builtin_fabs(sty - stx);
    }
  }
#ifndef MI355_NO_XRECOMP
  // the accepted point, re-formed from the accepted step: same operands, same bits
#pragma unroll
  for (int e = 0; e < E; ++e) x[e] = AR::nmadd(stp, d[e], wa[e]);
#endif
  return nfev;
}

}  // namespace mi355
