#!/usr/bin/env python3
"""Randomised differential run of the FIRST link of the parity chain, on the CPU: the sequential twin (oracle/*.hpp through
oracle/liboracle.so) against the reference binary (oracle/_ref/libref.so: the unmodified reference headers over
oracle/eigen_shim), compared for EQUALITY — x, f, g, status, num_iterations, nfev where the binding counts them, x_delta,
f_delta, gradient_norm of every problem of every trial.

scripts/fuzz_parity.py / fuzz_auglag.py draw the second link (device == butterfly twin) on the GPU; this script draws the
link that needs no GPU, over what the fixed grids of tests/test_oracle.py and tests/test_auglag_oracle.py do not list:
solver (Lbfgs / Lbfgsb / Bfgs, Lbfgs on the shared-matrix ridge in First and Second mode, Lbfgs on the Second-mode Rosenbrock
with the condition_hessian test, the augmented-Lagrangian loop on random term tables and constraint families), n, the
history sizes the reference binary instantiates, both line searches, every stopping field, start points, boxes.

    python scripts/fuzz_oracle_vs_reference.py --trials 2000 --seed 3 > /tmp/fuzz_cpu.jsonl
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

LBFGS_M = {"more_thuente": [1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 16, 20], "hager_zhang": [1, 3, 5, 6, 10]}   # oracle/ref_capi.cpp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=500)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=0.0)
    args = ap.parse_args()
    import oracle_lib as O
    import ref_lib as R
    import auglag_lib as al
    from fuzz_parity import pick_n, random_stop
    if not R.available():
        print(json.dumps({"summary": "oracle/_ref/libref.so is not available"}))
        return 2
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    counts = {"compared": 0, "mismatch": 0}
    per = {}
    PROG = ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm")
    for trial in range(args.trials):
        if args.budget_s and time.time() - t0 > args.budget_s:
            break
        kind = str(rng.choice(["lbfgs", "lbfgs", "lbfgsb", "lbfgsb", "bfgs", "ridge", "second_cond", "auglag", "auglag_family"]))
        ls = str(rng.choice(["more_thuente", "more_thuente", "hager_zhang"]))
        B = int(rng.choice([1, 2, 5, 12, 24]))
        rec = {"trial": trial, "kind": kind, "B": B, "linesearch": ls}
        bad = []

        def compare(o, r, keys=PROG):
            for name, a, b in zip(("x", "f", "g"), o[:3], r[:3]):
                if not np.array_equal(a, b, equal_nan=True):
                    bad.append(name)
            for k in keys:
                if not np.array_equal(o[3][k], r[3][k], equal_nan=True):
                    bad.append(k)

        if kind in ("lbfgs", "bfgs", "lbfgsb"):
            objective = str(rng.choice(["rosenbrock", "diag_quadratic"]))
            n = pick_n(rng, 64 if kind == "bfgs" else (128 if kind == "lbfgsb" else 256))
            stop = random_stop(rng, O, kind == "lbfgsb")
            stop.num_iterations = min(int(stop.num_iterations), 2000)
            params = None
            if objective == "diag_quadratic":
                a = rng.uniform(0.05, 40.0, n)
                params = np.concatenate([a, [float(rng.normal())]])
                x0 = rng.uniform(-3, 3, (B, n))
            else:
                x0 = (np.tile([-1.2, 1.0], n)[:n] + rng.normal(0, 0.3, (B, n))) if rng.random() < 0.5 else rng.uniform(-2, 2, (B, n))
            rec.update(objective=objective, n=n)
            if kind == "lbfgs":
                m = int(rng.choice(LBFGS_M[ls]))
                rec.update(m=m)
                compare(O.minimize_batch(objective, x0, m=m, stop=stop, params=params, linesearch=ls),
                        R.minimize_batch(objective, x0, m=m, stop=stop, params=params, linesearch=ls))
            elif kind == "bfgs":
                compare(O.bfgs_minimize_batch(objective, x0, stop=stop, params=params, linesearch=ls),
                        R.bfgs_minimize_batch(objective, x0, stop=stop, params=params, linesearch=ls))
            else:
                if objective == "diag_quadratic":
                    ls = "more_thuente"
                    m = int(rng.choice([3, 5, 6, 8, 10]))
                else:
                    m = int(rng.choice([3, 5] if ls == "hager_zhang" else [3, 5, 6, 10]))
                lo = hi = None
                if rng.random() < 0.8:
                    centre = np.zeros(n)
                    lo = centre - rng.uniform(0.0, 2.0, n)
                    hi = centre + rng.uniform(0.0, 2.0, n)
                    lo[rng.random(n) < 0.2] = -np.inf
                    hi[rng.random(n) < 0.2] = np.inf
                    pin = rng.random(n) < 0.05
                    hi[pin] = lo[pin] = centre[pin]
                rec.update(m=m, linesearch=ls, boxed=lo is not None)
                compare(O.lbfgsb_minimize_batch(objective, x0, m=m, stop=stop, params=params, lower=lo, upper=hi, std_sort_order=True,
                                                linesearch=ls),
                        R.lbfgsb_minimize_batch(objective, x0, m=m, stop=stop, params=params, lower=lo, upper=hi, linesearch=ls),
                        keys=PROG + ("nfev",))
        elif kind == "ridge":
            n = pick_n(rng, 64)
            rows = int(rng.choice([1, 5, 32, 64, 100, 128]))
            A = rng.normal(size=(rows, n))
            lam = float(rng.choice([0.1, 3.0, 0.0])) if rows >= n else float(rng.choice([0.1, 3.0]))
            Y = rng.normal(size=(B, rows))
            x0 = rng.normal(size=(B, n)) if rng.random() < 0.5 else np.zeros((B, n))
            second = bool(rng.integers(0, 2))
            stop = random_stop(rng, O)
            stop.num_iterations = min(int(stop.num_iterations), 2000)
            rec.update(n=n, rows=rows, lam=lam, second=second)
            compare(O.minimize_batch("squared_error_ridge", x0, m=10, stop=stop, params=O.ridge_params(A, lam), per_problem=Y,
                                     second_mode=second),
                    R.ridge_minimize_batch(A, lam, Y, x0, stop=stop, second_mode=second))
        elif kind == "second_cond":
            n = pick_n(rng, 48)
            m = int(rng.choice([5, 6, 10]))
            threshold = float(rng.choice([0.0, 3e3, 1e5, 1e13]))
            x0 = (np.tile([-1.2, 1.0], n)[:n] + rng.normal(0, 0.3, (B, n))) if rng.random() < 0.5 else rng.uniform(-2, 2, (B, n))
            stop = random_stop(rng, O)
            stop.num_iterations = min(int(stop.num_iterations), 1000)
            rec.update(n=n, m=m, condition_hessian=threshold)
            xr, fr, gr, pr, cr = R.rosenbrock_second_minimize_batch_cond(x0, m=m, stop=stop, condition_hessian=threshold)
            O.lib().oracle_set_condition_hessian_stop(threshold)
            O.lib().oracle_track_hessian_condition(1)
            try:
                o = O.minimize_batch("rosenbrock", x0, m=m, stop=stop, second_mode="functor")
                co = O.hessian_conditions(B)
            finally:
                O.lib().oracle_set_condition_hessian_stop(0.0)
                O.lib().oracle_track_hessian_condition(0)
            compare(o, (xr, fr, gr, pr))
            if not np.array_equal(co, cr, equal_nan=True):
                bad.append("condition_hessian")
        else:
            n = pick_n(rng, 64)
            if kind == "auglag":
                p = al.random_problem(n, rng)
            else:
                f_eq = int(rng.integers(0, min(n, 12))) if rng.random() < 0.6 else 0
                f_ineq = int(rng.integers(0, 60)) if (rng.random() < 0.8 or f_eq == 0) else 0
                if f_eq == 0 and f_ineq == 0:
                    f_ineq = 1
                p = al.random_family_problem(n, f_eq, f_ineq, seed=int(rng.integers(0, 1 << 30)), table=bool(rng.integers(0, 2)))
                rec.update(f_eq=f_eq, f_ineq=f_ineq)
                ls = "more_thuente"
            cfg_kw = dict(outer_num_iterations=int(rng.choice([1, 2, 5, 12])),
                          penalty_growth_factor=float(rng.choice([10.0, 4.0, 1.5])),
                          violation_shrink_ratio=float(rng.choice([0.25, 0.5, 0.9])),
                          auto_scale_initial_penalty=int(rng.integers(0, 2)),
                          warmup_max_inner_iterations=int(rng.choice([0, 3, 10])),
                          constraint_threshold=float(rng.choice([1e-5, 1e-3, 1e-8])),
                          kkt_stationarity_threshold=float(rng.choice([1e-4, 1e-2, 1e-7])),
                          multiplier_max=float(rng.choice([1e20, 5.0])))
            cfg = al.default_config(**cfg_kw)
            pen0 = float(rng.choice([0.0, 1.0, 3.0, 50.0]))
            x0 = rng.uniform(-1, 1, (B, n))
            lam0 = rng.uniform(-1, 1, (B, p.n_eq)) if (p.n_eq and rng.random() < 0.4) else None
            mu0 = rng.uniform(0, 1, (B, p.n_ineq)) if (p.n_ineq and rng.random() < 0.4) else None
            rec.update(n=n, n_eq=p.n_eq, n_ineq=p.n_ineq, linesearch=ls, config=cfg_kw, penalty0=pen0)
            o = al.oracle_minimize(p, x0, lambda0=lam0, mu0=mu0, penalty0=pen0, config=cfg, linesearch=ls)
            r = al.ref_minimize(p, x0, lambda0=lam0, mu0=mu0, penalty0=pen0, config=cfg, linesearch=ls)
            bad += [k for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient")
                    if not np.array_equal(o[k], r[k], equal_nan=True)]
            # (the reference keeps no inner-iteration / evaluation counters: the binding leaves them zero)
            bad += [k for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm")
                    if not np.array_equal(o["progress"][k], r["progress"][k], equal_nan=True)]
        rec["mismatch"] = bad
        counts["compared"] += 1
        counts["problems"] = counts.get("problems", 0) + B
        per[kind] = per.get(kind, 0) + 1
        if bad:
            counts["mismatch"] += 1
        print(json.dumps(rec), flush=True)
    print(json.dumps({"summary": dict(counts, by_kind=per, seed=args.seed, seconds=round(time.time() - t0, 1))}), flush=True)
    return 1 if counts["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main())
