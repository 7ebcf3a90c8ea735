#!/usr/bin/env python3
"""Randomised differential run of the augmented-Lagrangian loop: device solve == butterfly twin, compared for equality.

Like scripts/fuzz_parity.py, for `AugmentedLagrangian<Problem, Lbfgs>` (solver/augmented_lagrangian.h of the reference):
random term tables (tests/auglag_lib.py random_problem: every primitive kind, sums, all three forms), random constraint
families (dense affine rows, with and without table terms ahead of them, up to the capacity of the mapping), random
configuration (auto-scaled / manual initial penalty, growth factor, shrink ratio, warm-up, outer limit, thresholds),
random initial multipliers, both device loops (fused / lock-step; families: fused), both line searches.  Compared: x,
both multiplier vectors, penalty, max_violation, KKT norm, and every field of the progress record.

    python scripts/fuzz_auglag.py --trials 300 --seed 3 > gpurun_out/fuzz_auglag.jsonl
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def padded(n):
    P = 8
    while P < n:
        P <<= 1
    return P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=0.0)
    args = ap.parse_args()
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import BatchedAugmentedLagrangian, ConstrainedProblem, capi
    import auglag_lib as al

    def engine_problem(p):
        mk = lambda t: ConstrainedProblem.term(t["prims"], t["form"], t["k"], product=t.get("product", False))
        return ConstrainedProblem(p.n, mk(p.terms[0]), [mk(t) for t in p.table_eq], [mk(t) for t in p.table_ineq],
                                  family_equality=p.family_equality, family_inequality=p.family_inequality)

    rng = np.random.default_rng(args.seed)
    ctx = amd.Context(0)
    t0 = time.time()
    counts = {"compared": 0, "refused": 0, "mismatch": 0}
    by_kind = {}
    for trial in range(args.trials):
        if args.budget_s and time.time() - t0 > args.budget_s:
            break
        kind = str(rng.choice(["table", "table", "family"]))
        edges = [1, 2, 5, 8, 9, 16, 17, 30, 32, 33, 64, 65, 100, 128, 129, 200, 256]
        n = int(rng.choice(edges)) if rng.random() < 0.5 else int(rng.integers(1, 257))
        B = int(rng.choice([1, 3, 9, 20, 40]))
        ls = "more_thuente"
        loop = "fused"
        rec = {"trial": trial, "kind": kind, "n": n, "B": B}
        try:
            if kind == "table":
                p = al.random_problem(n, rng)
                ls = str(rng.choice(["more_thuente", "more_thuente", "hager_zhang"]))
                loop = str(rng.choice(["fused", "lockstep", "auto"]))
            else:
                cap = int(capi.load().mi355_auglag_family_capacity(C.c_int32(n)))
                f_eq = int(rng.integers(0, max(1, min(cap, 2 * n) + 1))) if rng.random() < 0.6 else 0
                f_ineq = int(rng.integers(0, cap + 1)) if (rng.random() < 0.8 or f_eq == 0) else 0
                if f_eq == 0 and f_ineq == 0:
                    f_ineq = 1
                f_eq = min(f_eq, max(0, n - 1))      # keep the equality rows of a feasible point independent
                p = al.random_family_problem(n, f_eq, f_ineq, seed=int(rng.integers(0, 1 << 30)), table=bool(rng.integers(0, 2)))
                rec.update(f_eq=f_eq, f_ineq=f_ineq, capacity=int(cap))
            cfg_kw = dict(outer_num_iterations=int(rng.choice([1, 2, 5, 12, 25])),
                          penalty_growth_factor=float(rng.choice([10.0, 4.0, 1.5])),
                          violation_shrink_ratio=float(rng.choice([0.25, 0.5, 0.9])),
                          auto_scale_initial_penalty=int(rng.integers(0, 2)),
                          warmup_max_inner_iterations=int(rng.choice([0, 3, 10])),
                          constraint_threshold=float(rng.choice([1e-5, 1e-3, 1e-8])),
                          kkt_stationarity_threshold=float(rng.choice([1e-4, 1e-2, 1e-7])),
                          multiplier_max=float(rng.choice([1e20, 5.0])))
            cfg = al.default_config(**cfg_kw)
            pen0 = float(rng.choice([0.0, 1.0, 3.0, 50.0]))
            x0 = rng.uniform(-1, 1, (B, n)) * float(rng.choice([1.0, 1.0, 2.5]))
            lam0 = rng.uniform(-1, 1, (B, p.n_eq)) if (p.n_eq and rng.random() < 0.4) else None
            mu0 = rng.uniform(0, 1, (B, p.n_ineq)) if (p.n_ineq and rng.random() < 0.4) else None
            rec.update(n_eq=p.n_eq, n_ineq=p.n_ineq, linesearch=ls, loop=loop, config=cfg_kw, penalty0=pen0,
                       multipliers_given=[lam0 is not None, mu0 is not None])
            s = BatchedAugmentedLagrangian(context=ctx, linesearch=ls)
            c = s.default_config()
            for name, _ in cfg._fields_:
                setattr(c, name, getattr(cfg, name))
            c.loop = capi.AL_LOOP[loop]
            s.config = c
            d = s.minimize_host(engine_problem(p), x0, lambda0=lam0, mu0=mu0, penalty0=pen0)
            o = al.oracle_minimize(p, x0, lambda0=lam0, mu0=mu0, penalty0=pen0, config=cfg, reduction="butterfly", width=padded(n),
                                   linesearch=ls)
            bad = [k for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient")
                   if not np.array_equal(d[k], o[k], equal_nan=True)]
            bad += [k for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm", "inner_iterations", "nfev", "sum_k")
                    if not np.array_equal(d["progress"][k], o["progress"][k], equal_nan=True)]
            rec["mismatch"] = bad
            rec["outer_max"] = int(d["progress"]["num_iterations"].max())
            rec["inner_max"] = int(d["progress"]["inner_iterations"].max())
            counts["compared"] += 1
            by_kind[kind] = by_kind.get(kind, 0) + 1
            if bad:
                counts["mismatch"] += 1
                rec["max_abs_dx"] = float(np.nanmax(np.abs(d["x"] - o["x"])))
        except capi.EngineError as e:
            rec["refused"] = "%d: %s" % (e.code, str(e)[:160])
            counts["refused"] += 1
        print(json.dumps(rec), flush=True)
    print(json.dumps({"summary": dict(counts, by_kind=by_kind, seed=args.seed, seconds=round(time.time() - t0, 1))}), flush=True)
    ctx.close()
    return 1 if counts["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main())
