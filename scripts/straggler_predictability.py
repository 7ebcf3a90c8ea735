#!/usr/bin/env python3
"""Can the long solves of the configs[1] batch be predicted from their start points?  (CPU only.)

The drain tail of the headline batch is t(fetch of the longest solves) + their length (DESIGN.md section 2 item 6,
profiles/HISTORY.md section 3.8): a longest-first fetch order would hide it IF the stragglers (the 3 % of the problems above
300 iterations; mean 223, maximum 545) could be named before they run.  This script solves a prefix of the bench batch with
the CPU twin of the production kernel (test infrastructure, here as the measuring device), then fits gradient-boosted
trees on the start point and on everything cheap to compute from it (f, log f, |g|_inf, |g|_2, extremes, the trailing
coordinates) and reports how well a held-out third is ranked.

    python scripts/straggler_predictability.py > profiles/r6_straggler_predictability.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from sklearn.ensemble import GradientBoostingClassifier, GradientBoostingRegressor
    from sklearn.metrics import roc_auc_score
    from sklearn.model_selection import train_test_split
    import cppnumericalsolvers_amd as amd
    import oracle_lib as O
    B, n, m = 16384, 32, 6
    x0 = amd.synthetic_x0_host(B, n, "std")
    _, _, _, p = O.minimize_batch("rosenbrock", x0, m=m, stop=O.parity_stop(), nthreads=os.cpu_count() or 8,
                                  reduction="butterfly_fma", width=32, fma_group=4)
    it = p["num_iterations"].astype(float)
    print("first %d problems of the configs[1] batch (Rosenbrock-%d, m = %d, parity stopping), the fused kernel's CPU twin:" % (B, n, m))
    print("  iterations mean %.1f, 99 %% %.0f, max %.0f; above 300 iterations: %.2f %% of the problems" %
          (it.mean(), np.percentile(it, 99), it.max(), 100.0 * (it > 300).mean()))
    a, b = 1 - x0[:, :-1], x0[:, 1:] - x0[:, :-1] ** 2
    f = (a * a + 100 * b * b).sum(1)
    g = np.zeros_like(x0)
    g[:, :-1] += -2 * a - 400 * b * x0[:, :-1]
    g[:, 1:] += 200 * b
    named = [("f(x0)", f), ("log f(x0)", np.log(f)), ("|g(x0)|_inf", np.abs(g).max(1)), ("|g(x0)|_2", np.linalg.norm(g, axis=1)),
             ("min x0", x0.min(1)), ("max x0", x0.max(1)), ("x0[n-1]", x0[:, -1]), ("x0[n-2]", x0[:, -2]), ("x0[0]", x0[:, 0])]
    feats = np.column_stack([x0] + [v for _, v in named])
    y = (it > 300).astype(int)
    Xtr, Xte, ytr, yte, ittr, itte = train_test_split(feats, y, it, test_size=1.0 / 3.0, random_state=0)
    clf = GradientBoostingClassifier(n_estimators=200, max_depth=3, random_state=0).fit(Xtr, ytr)
    reg = GradientBoostingRegressor(n_estimators=200, max_depth=3, random_state=0).fit(Xtr, ittr)
    print("gradient-boosted trees on (x0, %s), held-out third:" % ", ".join(k for k, _ in named))
    print("  ranking the stragglers (> 300 iterations): AUC %.3f   (0.5 = chance)" % roc_auc_score(yte, clf.predict_proba(Xte)[:, 1]))
    print("  predicted vs actual iteration count: correlation %+.3f" % np.corrcoef(reg.predict(Xte), itte)[0, 1])
    print("single quantities against the iteration count (correlation, AUC for > 300):")
    for k, v in named:
        print("  %-12s %+.3f  %.3f" % (k, np.corrcoef(v, it)[0, 1], roc_auc_score(y, v)))
    print("Nothing cheap names a straggler before it runs: a longest-first fetch order has nothing to sort by.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
