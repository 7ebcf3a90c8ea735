// ORACLE — TEST INFRASTRUCTURE ONLY (see lbfgs_oracle.hpp).
//
// CPU restatement of the reference's box-constrained solver
//   include/cppoptlib/solver/lbfgsb.h  (Lbfgsb<FunctionType, m = 5, MoreThuente>)
// — OptimizationStep :141-238, Minimize :247-292, GetGeneralizedCauchyPoint :318-430,
// FindAlpha :435-457, SubspaceMinimization :459-515, SolveM :311-316 — with plain arrays.
// The small dense algebra (2k x 2k, k <= m) follows oracle/eigen_shim's evaluation order
// (ascending sums, right-looking LU with first-maximum row pivoting, column-oriented
// substitution), which is what the unmodified reference header computes when compiled over
// that shim; real Eigen may differ in the last ulp inside its product / LU kernels.
// Length-n reductions go through the Reducer policy like in lbfgs_oracle.hpp; with the
// Butterfly policy the free-variable products of SubspaceMinimization are trees over all n
// positions with zeros at the non-free ones (what a wavefront segment computes).
// Breakpoints are ordered by (t, index) — a stable order; the reference's std::sort leaves the
// order of exactly equal breakpoints implementation defined.
#pragma once
#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <vector>

#include "lbfgs_oracle.hpp"

namespace oracle {

// Dense column-major matrix with the shim's LU (oracle/eigen_shim/Eigen/LU).
struct SmallLU {
  int n = 0;
  std::vector<double> lu;  // column major
  std::vector<int> piv;
  bool empty() const { return n == 0; }
  double& at(int i, int j) { return lu[static_cast<size_t>(j) * n + i]; }
  double at(int i, int j) const { return lu[static_cast<size_t>(j) * n + i]; }
  void factor(const std::vector<double>& a, int size) {
    n = size;
    lu = a;
    piv.assign(n, 0);
    for (int k = 0; k < n; ++k) {
      int p = k;
      double best = std::fabs(at(k, k));
      for (int i = k + 1; i < n; ++i) {
        const double v = std::fabs(at(i, k));
        if (v > best) {
          best = v;
          p = i;
        }
      }
      piv[k] = p;
      if (best != 0.0) {
        if (p != k)
          for (int j = 0; j < n; ++j) std::swap(at(k, j), at(p, j));
        for (int i = k + 1; i < n; ++i) at(i, k) = at(i, k) / at(k, k);
      }
      for (int j = k + 1; j < n; ++j)
        for (int i = k + 1; i < n; ++i) at(i, j) = at(i, j) - at(i, k) * at(k, j);
    }
  }
  std::vector<double> solve(const std::vector<double>& b) const {
    std::vector<double> x = b;
    for (int k = 0; k < n; ++k) std::swap(x[k], x[piv[k]]);
    for (int j = 0; j < n; ++j)
      for (int i = j + 1; i < n; ++i) x[i] = x[i] - x[j] * at(i, j);
    for (int j = n - 1; j >= 0; --j) {
      x[j] = x[j] / at(j, j);
      for (int i = 0; i < j; ++i) x[i] = x[i] - x[j] * at(i, j);
    }
    return x;
  }
};

struct Lbfgsb {
  int m = 5;  // lbfgsb.h:44
  Stopping stopping_progress;
  Reducer red;
  std::vector<double> lower, upper;  // SetBounds (:89-93); empty = unbounded (:124-129)
  // Order of exactly equal breakpoints: false = by index (stable; what the device computes),
  // true = whatever libstdc++'s std::sort produces for the reference's call (lbfgsb.h:298-305),
  // used only to compare bit for bit with the reference binary built by the same toolchain.
  bool std_sort_order = false;

  // private state, lbfgsb.h:517-532
  int n_ = 0;
  double theta_ = 1.0;
  int k_ = 0;                        // stored pairs (columns of y_history_/s_history_)
  std::vector<double> Yh_, Sh_;      // n x k, column major, chronological (oldest first)
  SmallLU MM_lu_;
  double last_projected_gradient_norm_ = std::numeric_limits<double>::infinity();
  uint64_t nfev = 0, sum_k = 0;
  int linesearch = 0;  // LineSearch template argument (lbfgsb.h:45): 0 MoreThuente, 1 HagerZhang
  // bench.py's useful-flop model of configs[4] (measurement only: nothing below feeds back into the iteration)
  double model_flops = 0.0;                       // sum of ReferenceStepFlops over the OptimizationSteps of this solve
  uint64_t sum_breakpoints = 0, sum_free = 0;     // breakpoints examined (:382-412), free variables (:462-468), summed
  mutable int last_breakpoints_ = 0, last_free_ = 0;

  // Floating-point operations of ONE OptimizationStep in the algebra lbfgsb.h writes down (a division or a comparison
  // against a bound counts 1; the objective evaluations of the line search are priced separately, from nfev):
  // c = 2k columns of W, nb = breakpoints examined by the Cauchy loop, F = free variables of the subspace step.
  static double ReferenceStepFlops(int n, int c, int nb, int F, bool pair_accepted, int k_after) {
    const double dn = n, dc = c, dF = F;
    double fl = 4.0 * dn;                               // clip (:148), projected-gradient norm (:165)
    fl += 2.0 * dn;                                     // breakpoints: one difference and one quotient each (:334-347)
    fl += 2.0 * dc * dn + 2.0 * dn;                     // p = W^T d (:353), f' = -d.d (:357)
    fl += 2.0 * dc * dc + 2.0 * dc + 4.0;               // SolveM(p), p.(Mp), f'', dt_min (:361-366)
    fl += nb * (6.0 * dc * dc + 10.0 * dc + 20.0);      // c += dt p, three SolveM, three dots, p += g_b w_b, scalars (:382-412)
    fl += 2.0 * dn + 2.0 * dc;                          // drift of the free coordinates (:424-427), c += dt_min p (:429)
    if (F > 0) {
      fl += 2.0 * dc * dc + 2.0 * dc * dn + 3.0 * dn;   // SolveM(c), W (M c), rr (:480)
      fl += 2.0 * dc * dF + 2.0 * dc * dc;              // WZ r, SolveM (:485)
      fl += 2.0 * dc * dc * dF + dc * dc;               // N = WZ WZ^T / theta (:487)
      fl += 2.0 * dc * dc * dc + dc * dc;               // one SolveM per column of N, I - MN (:489-495)
      fl += (2.0 / 3.0) * dc * dc * dc + 2.0 * dc * dc; // N.lu().solve(v) (:499)
      fl += 2.0 * dc * dF + 3.0 * dF;                   // du (:503-504)
      fl += 4.0 * dF;                                   // FindAlpha (:435-457), alpha* du added to the Cauchy point (:508-514)
      fl += dn;                                         // direction = subspace_min - x (:189)
    }
    fl += 6.0 * dn;                                     // new_y, new_s, s.y, y.y (:206-211)
    if (pair_accepted) {
      const double k = k_after;
      fl += 1.0 + k * dn;                               // theta, theta S (:222-226)
      fl += 4.0 * k * k * dn + k * k;                   // S^T Y and S^T S recomputed in full, the latter scaled (:227-232)
      fl += (2.0 / 3.0) * 8.0 * k * k * k;              // MM.lu() (:234)
    }
    return fl;
  }

  explicit Lbfgsb(int m_in = 5, Stopping stop = DefaultStopping(), Reducer r = Reducer{})
      : m(m_in), stopping_progress(stop), red(r) {}

  // reference default constructor (:84-87): f_delta = 2.22e-9, relative
  static Stopping DefaultLbfgsbStopping() {
    Stopping s = DefaultStopping();
    s.f_delta = 2.22e-9;
    s.f_delta_relative = true;
    return s;
  }

  double W(int i, int a) const {  // W = [Y, theta*S]  (:224-226)
    return a < k_ ? Yh_[static_cast<size_t>(a) * n_ + i] : theta_ * Sh_[static_cast<size_t>(a - k_) * n_ + i];
  }
  std::vector<double> SolveM(const std::vector<double>& b) const {  // :311-316
    if (b.empty() || MM_lu_.empty()) return b;
    return MM_lu_.solve(b);
  }
  std::vector<double> clip(const std::vector<double>& x) const {  // cwiseMin(upper).cwiseMax(lower)
    std::vector<double> r(x.size());
    for (size_t j = 0; j < x.size(); ++j) r[j] = std::max(std::min(x[j], upper[j]), lower[j]);
    return r;
  }
  double ProjectedGradientInfNorm(const std::vector<double>& x, const std::vector<double>& g) const {  // :105-118
    double norm = 0.0;
    for (size_t j = 0; j < x.size(); ++j) {  // (x.size(): also callable on a solver that never ran, as the reference's)
      double gj = g[j];
      if (x[j] <= lower[j] && gj > 0) gj = 0.0;
      if (x[j] >= upper[j] && gj < 0) gj = 0.0;
      norm = std::max(norm, std::fabs(gj));
    }
    return norm;
  }

  void InitializeSolver(int n) {  // :120-139
    n_ = n;
    if (lower.empty()) {
      lower.assign(n, std::numeric_limits<double>::lowest());
      upper.assign(n, std::numeric_limits<double>::max());
    }
    theta_ = 1.0;
    k_ = 0;
    Yh_.clear();
    Sh_.clear();
    MM_lu_ = SmallLU();
  }

  // length-n dot over the positions where mask != 0 (all positions if mask == nullptr)
  double dotn(const double* a, const double* b, const unsigned char* mask = nullptr) const {
    double t[1024];
    int cnt = 0;
    if (red.kind == Reduction::Sequential) {
      for (int i = 0; i < n_; ++i)
        if (!mask || mask[i]) t[cnt++] = a[i] * b[i];
      return red.sum(t, cnt);
    }
    for (int i = 0; i < n_; ++i) t[i] = (!mask || mask[i]) ? a[i] * b[i] : 0.0;
    return red.sum(t, n_);
  }
  std::vector<double> Wcol(int a) const {
    std::vector<double> c(n_);
    for (int i = 0; i < n_; ++i) c[i] = W(i, a);
    return c;
  }

  // :318-430
  void GetGeneralizedCauchyPoint(const std::vector<double>& x, const std::vector<double>& g,
                                 std::vector<double>* x_cauchy, std::vector<double>* c) const {
    constexpr double max_value = std::numeric_limits<double>::max();
    constexpr double epsilon = 1e-12;
    const int n = n_, k2 = 2 * k_;
    std::vector<double> t_of(n), d(n);
    for (int j = 0; j < n; ++j) d[j] = -g[j];
    for (int j = 0; j < n; ++j) {                                   // :334-347
      if (g[j] == 0) {
        t_of[j] = max_value;
      } else {
        const double tmp = (g[j] < 0) ? (x[j] - upper[j]) / g[j] : (x[j] - lower[j]) / g[j];
        t_of[j] = tmp;
        if (tmp == 0) d[j] = 0;
      }
    }
    std::vector<int> sorted(n);                                     // :349
    std::iota(sorted.begin(), sorted.end(), 0);
    if (std_sort_order) {
      std::sort(sorted.begin(), sorted.end(), [&](size_t a, size_t b) { return t_of[a] < t_of[b]; });
    } else {
      std::stable_sort(sorted.begin(), sorted.end(), [&](int a, int b) { return t_of[a] < t_of[b]; });
    }
    *x_cauchy = x;
    last_breakpoints_ = 0;
    std::vector<double> p(k2);                                      // p = W^T d  (:353)
    for (int a = 0; a < k2; ++a) {
      const std::vector<double> wc = Wcol(a);
      p[a] = dotn(wc.data(), d.data());
    }
    c->assign(k2, 0.0);
    double f_prime = -dotn(d.data(), d.data());                      // :357
    const std::vector<double> Mp0 = SolveM(p);
    double pMp = 0.0;
    for (int a = 0; a < k2; ++a) pMp = (a == 0) ? p[0] * Mp0[0] : pMp + p[a] * Mp0[a];
    double f_doubleprime = (-theta_) * f_prime - pMp;               // :361-362
    f_doubleprime = std::max(epsilon, f_doubleprime);
    const double f_dp_orig = f_doubleprime;
    double dt_min = -f_prime / f_doubleprime;
    double t_old = 0;
    int i = 0;
    for (int j = 0; j < n; j++) {                                   // :370-375
      i = j;
      if (t_of[sorted[j]] > 0) break;
    }
    int b = sorted[i];
    double t = t_of[b];
    double dt = t;
    while ((dt_min >= dt) && (i < n)) {                             // :382-412
      if (d[b] > 0)
        (*x_cauchy)[b] = upper[b];
      else if (d[b] < 0)
        (*x_cauchy)[b] = lower[b];
      const double zb = (*x_cauchy)[b] - x[b];
      for (int a = 0; a < k2; ++a) (*c)[a] = (*c)[a] + dt * p[a];
      std::vector<double> wbt(k2);
      for (int a = 0; a < k2; ++a) wbt[a] = W(b, a);
      const std::vector<double> Mc = SolveM(*c), Mp = SolveM(p), Mwbt = SolveM(wbt);
      const double gb = g[b];
      double s1 = 0.0, s2 = 0.0, s3 = 0.0;
      for (int a = 0; a < k2; ++a) {
        const double t1 = (gb * wbt[a]) * Mc[a];
        const double t2 = wbt[a] * Mp[a];
        const double t3 = ((gb * gb) * wbt[a]) * Mwbt[a];
        s1 = (a == 0) ? t1 : s1 + t1;
        s2 = (a == 0) ? t2 : s2 + t2;
        s3 = (a == 0) ? t3 : s3 + t3;
      }
      f_prime += ((dt * f_doubleprime + gb * gb) + (theta_ * gb) * zb) - s1;            // :396-397
      f_doubleprime += ((((-1.0) * theta_) * gb) * gb - 2.0 * (gb * s2)) - s3;          // :398-400
      f_doubleprime = std::max(epsilon * f_dp_orig, f_doubleprime);
      for (int a = 0; a < k2; ++a) p[a] = p[a] + gb * wbt[a];
      d[b] = 0;
      dt_min = -f_prime / f_doubleprime;
      t_old = t;
      ++i;
      if (i < n) {
        b = sorted[i];
        t = t_of[b];
        dt = t - t_old;
      }
      ++last_breakpoints_;
    }
    dt_min = std::max(dt_min, 0.0);
    t_old += dt_min;
    for (int j = i; j < n; ++j) {                                   // :424-427
      const int idx = sorted[j];
      (*x_cauchy)[idx] = x[idx] + t_old * d[idx];
    }
    for (int a = 0; a < k2; ++a) (*c)[a] = (*c)[a] + dt_min * p[a];  // :429
  }

  // :459-515 (FindAlpha :435-457 inlined)
  bool SubspaceMinimization(const std::vector<double>& x, const std::vector<double>& g,
                            const std::vector<double>& x_cauchy, const std::vector<double>& c,
                            std::vector<double>* subspace_min) const {
    const int n = n_, k2 = 2 * k_;
    std::vector<unsigned char> is_free(n, 0);
    std::vector<int> free_idx;
    for (int i = 0; i < n; i++)
      if ((x_cauchy[i] != upper[i]) && (x_cauchy[i] != lower[i])) {
        is_free[i] = 1;
        free_idx.push_back(i);
      }
    *subspace_min = x_cauchy;
    last_free_ = static_cast<int>(free_idx.size());
    if (free_idx.empty()) return false;                             // :472-474
    const double theta_inverse = 1 / theta_;
    const std::vector<double> Mc = SolveM(c);
    std::vector<double> rr(n);                                      // :480
    for (int i = 0; i < n; ++i) {
      double wmc = 0.0;
      for (int a = 0; a < k2; ++a) wmc = (a == 0) ? W(i, 0) * Mc[0] : wmc + W(i, a) * Mc[a];
      rr[i] = (g[i] + theta_ * (x_cauchy[i] - x[i])) - wmc;
    }
    std::vector<double> wzr(k2);                                    // WZ * r  (:485)
    for (int a = 0; a < k2; ++a) {
      const std::vector<double> wc = Wcol(a);
      wzr[a] = dotn(wc.data(), rr.data(), is_free.data());
    }
    std::vector<double> v = SolveM(wzr);
    std::vector<double> N(static_cast<size_t>(k2) * k2);             // :487  N = theta^-1 * WZ * WZ^T
    for (int b = 0; b < k2; ++b) {
      const std::vector<double> wb = Wcol(b);
      for (int a = 0; a < k2; ++a) {
        std::vector<double> wa = Wcol(a);
        for (int i = 0; i < n; ++i) wa[i] = theta_inverse * wa[i];
        N[static_cast<size_t>(b) * k2 + a] = dotn(wa.data(), wb.data(), is_free.data());
      }
    }
    if (k2 > 0) {                                                   // :489-495  N = I - M^-1 N
      std::vector<double> MN(N.size());
      for (int col = 0; col < k2; ++col) {
        std::vector<double> ncol(N.begin() + static_cast<size_t>(col) * k2, N.begin() + static_cast<size_t>(col + 1) * k2);
        const std::vector<double> sol = SolveM(ncol);
        std::copy(sol.begin(), sol.end(), MN.begin() + static_cast<size_t>(col) * k2);
      }
      for (int col = 0; col < k2; ++col)
        for (int row = 0; row < k2; ++row)
          N[static_cast<size_t>(col) * k2 + row] = ((row == col) ? 1.0 : 0.0) - MN[static_cast<size_t>(col) * k2 + row];
    }
    if (!v.empty()) {                                               // :498-500
      SmallLU nlu;
      nlu.factor(N, k2);
      v = nlu.solve(v);
    }
    const double ti2 = theta_inverse * theta_inverse;
    double alphastar = 1;                                           // FindAlpha :435-457
    std::vector<double> du(free_idx.size());
    for (size_t f = 0; f < free_idx.size(); ++f) {                  // :503-504
      const int i = free_idx[f];
      double wv = 0.0;
      for (int a = 0; a < k2; ++a) wv = (a == 0) ? (ti2 * W(i, 0)) * v[0] : wv + (ti2 * W(i, a)) * v[a];
      du[f] = (-theta_inverse) * rr[i] - wv;
    }
    for (size_t f = 0; f < free_idx.size(); ++f) {
      const int i = free_idx[f];
      if (std::fabs(du[f]) < 1e-7) {
        continue;
      } else if (du[f] > 0) {
        alphastar = std::min(alphastar, (upper[i] - x_cauchy[i]) / du[f]);
      } else {
        alphastar = std::min(alphastar, (lower[i] - x_cauchy[i]) / du[f]);
      }
    }
    for (size_t f = 0; f < free_idx.size(); ++f) {                  // :508-514
      const int i = free_idx[f];
      (*subspace_min)[i] = (*subspace_min)[i] + alphastar * du[f];
    }
    return true;
  }

  State eval_state(const Objective& function, const std::vector<double>& x) {
    State s;
    s.x = x;
    s.gradient.assign(x.size(), 0.0);
    s.value = function.eval(s.x.data(), s.gradient.data(), static_cast<int>(x.size()), red);
    ++nfev;
    return s;
  }

  // :141-238
  State OptimizationStep(const Objective& function, const State& current) {
    const int n = n_;
    std::vector<double> x = clip(current.x);                        // :148
    double current_value = current.value;
    std::vector<double> current_gradient = current.gradient;
    if (x != current.x) {                                           // :151-153
      State s = eval_state(function, x);
      current_value = s.value;
      current_gradient = s.gradient;
    }
    sum_k += static_cast<uint64_t>(k_);
    last_projected_gradient_norm_ = ProjectedGradientInfNorm(x, current_gradient);  // :165-166
    std::vector<double> cauchy_point, c;
    GetGeneralizedCauchyPoint(x, current_gradient, &cauchy_point, &c);
    std::vector<double> subspace_min;
    const bool do_line_search = SubspaceMinimization(x, current_gradient, cauchy_point, c, &subspace_min);
    State next;
    next.x = x;
    next.value = current_value;
    next.gradient = current_gradient;
    if (do_line_search) {                                           // :186-193
      std::vector<double> direction(n);
      for (int j = 0; j < n; ++j) direction[j] = subspace_min[j] - x[j];
      next = (linesearch == 1) ? HagerZhang::Search(next, direction, function, red, 1.0, &nfev)
                               : MoreThuente::Search(next, direction, function, red, 1.0, &nfev);
    } else {
      next = eval_state(function, subspace_min);
    }
    const std::vector<double> clipped = clip(next.x);              // :199-203
    if (clipped != next.x) next = eval_state(function, clipped);
    std::vector<double> new_y(n), new_s(n);                        // :206-207
    for (int j = 0; j < n; ++j) new_y[j] = next.gradient[j] - current_gradient[j];
    for (int j = 0; j < n; ++j) new_s[j] = next.x[j] - x[j];
    const double sTy = dotn(new_s.data(), new_y.data());
    const double yTy = dotn(new_y.data(), new_y.data());
    sum_breakpoints += static_cast<uint64_t>(last_breakpoints_);
    sum_free += static_cast<uint64_t>(last_free_);
    model_flops += ReferenceStepFlops(n, 2 * k_, last_breakpoints_, last_free_, sTy > 1e-7 * yTy, std::min(k_ + 1, m));
    if (sTy > 1e-7 * yTy) {                                         // :211
      if (k_ < m) {
        Yh_.resize(static_cast<size_t>(k_ + 1) * n);
        Sh_.resize(static_cast<size_t>(k_ + 1) * n);
        k_++;
      } else {                                                      // shift left (:216-217)
        std::copy(Yh_.begin() + n, Yh_.end(), Yh_.begin());
        std::copy(Sh_.begin() + n, Sh_.end(), Sh_.begin());
      }
      std::copy(new_y.begin(), new_y.end(), Yh_.begin() + static_cast<size_t>(k_ - 1) * n);
      std::copy(new_s.begin(), new_s.end(), Sh_.begin() + static_cast<size_t>(k_ - 1) * n);
      theta_ = yTy / dotn(new_y.data(), new_s.data());             // :222-223
      const int k = k_, k2 = 2 * k_;
      std::vector<double> A(static_cast<size_t>(k) * k), SS(static_cast<size_t>(k) * k);
      for (int bcol = 0; bcol < k; ++bcol)
        for (int a = 0; a < k; ++a) {
          A[static_cast<size_t>(bcol) * k + a] = dotn(&Sh_[static_cast<size_t>(a) * n], &Yh_[static_cast<size_t>(bcol) * n]);
          SS[static_cast<size_t>(bcol) * k + a] = dotn(&Sh_[static_cast<size_t>(a) * n], &Sh_[static_cast<size_t>(bcol) * n]);
        }
      std::vector<double> MM(static_cast<size_t>(k2) * k2, 0.0);    // :227-232
      auto mm = [&](int i, int j) -> double& { return MM[static_cast<size_t>(j) * k2 + i]; };
      for (int a = 0; a < k; ++a) mm(a, a) = -1 * A[static_cast<size_t>(a) * k + a];                // D
      for (int j = 0; j < k; ++j)
        for (int i = 0; i < k; ++i) {
          const double l_ij = (i > j) ? A[static_cast<size_t>(j) * k + i] : 0.0;                     // L(i,j)
          mm(k + i, j) = l_ij;                                                                       // L
          mm(j, k + i) = l_ij;                                                                       // L^T
          mm(k + i, k + j) = SS[static_cast<size_t>(j) * k + i] * theta_;                            // (S^T S) theta
        }
      MM_lu_.factor(MM, k2);                                        // :234
    }
    return next;
  }

  // :247-292
  State Minimize(const Objective& function, const std::vector<double>& x0, Progress* progress_out) {
    const int n = static_cast<int>(x0.size());
    Progress solver_state;
    nfev = 0;
    sum_k = 0;
    model_flops = 0.0;
    sum_breakpoints = sum_free = 0;
    State cur = eval_state(function, x0);                           // :253
    Stopping stop = stopping_progress;                              // :258-260
    const double projected_gradient_tolerance = stop.gradient_norm;
    stop.gradient_norm = 0.0;
    InitializeSolver(n);
    do {
      const State prev = cur;
      cur = OptimizationStep(function, prev);
      solver_state.Update(prev, cur, stop);
      if ((projected_gradient_tolerance > 0) && (last_projected_gradient_norm_ < projected_gradient_tolerance)) {
        solver_state.status = GradientNormViolation;               // :280-283 (overrides, quirk Q10)
      }
    } while (solver_state.status == Continue);
    if (progress_out) *progress_out = solver_state;
    return cur;
  }
};

}  // namespace oracle
