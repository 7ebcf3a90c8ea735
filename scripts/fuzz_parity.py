#!/usr/bin/env python3
"""Randomised differential run: device solve == butterfly twin, bit for bit, on random shapes of the hot path.

The committed GPU tests pin chosen grids (tests/test_gpu_parity.py and friends).  This script draws what no grid lists —
dimension, history size, mapping of a problem onto the wavefront, history placement, line search, arithmetic policy,
every stopping field, batch size, start points, boxes — runs the engine (through the C-ABI, as the tests do) and the CPU
twin of the same summation tree (tests/oracle_lib.py: test infrastructure, here as the checker), and compares x, f, g,
status, iteration and evaluation counts for equality.  A refusal of the library (a shape it has no kernel for) is
counted, never compared.  One JSON line per trial on stdout, a summary at the end; exit code 1 on any mismatch.

    python scripts/fuzz_parity.py --trials 400 --seed 5 > gpurun_out/fuzz.jsonl
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


def pick_n(rng, hi):
    edges = [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256]
    if rng.random() < 0.5:
        return int(rng.choice([e for e in edges if e <= hi]))
    return int(rng.integers(1, hi + 1))


def random_stop(rng, O, lbfgsb=False):
    kw = dict(num_iterations=int(rng.choice([1, 2, 7, 60, 400, 10000])),
              x_delta=float(rng.choice([0.0, 1e-9, 1e-11, 1e-4])),
              x_delta_violations=int(rng.choice([0, 1, 1, 3])),
              f_delta=float(rng.choice([0.0, 0.0, 1e-10, 2.22e-9, 1e-5])),
              f_delta_violations=int(rng.choice([0, 1, 1, 2])),
              f_delta_relative=int(rng.integers(0, 2)),
              gradient_norm=float(rng.choice([0.0, 1e-5, 1e-8, 1e-3])),
              gradient_norm_relative=int(rng.integers(0, 2)),
              past=int(rng.choice([0, 0, 1, 3, 5, 8])),
              past_delta=float(rng.choice([0.0, 1e-6, 1e-10])))
    if lbfgsb and rng.random() < 0.3:
        return O.lbfgsb_default_stop()
    if rng.random() < 0.2:
        return O.parity_stop()
    if rng.random() < 0.2:
        return O.default_stop()
    return O.make_stop(**kw)


def starts(rng, amd, B, n, objective):
    kind = rng.integers(0, 4)
    if objective == "rosenbrock":
        if kind == 0:
            return amd.synthetic_x0_host(B, n, "std", seed=int(rng.integers(1, 1 << 20)))
        if kind == 1:
            return amd.synthetic_x0_host(B, n, "u2", seed=int(rng.integers(1, 1 << 20)))
        if kind == 2:
            return rng.uniform(-1.5, 1.5, (B, n))
        return np.tile([-1.2, 1.0], n)[:n] + rng.normal(0, 0.3, (B, n))
    return rng.uniform(-3, 3, (B, n)) * (10.0 ** rng.integers(-1, 2))


def to_dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def same(dev, ora, keys=("status", "num_iterations", "nfev"), nan_keys_exempt=False):
    """Equality of every output.  nan_keys_exempt (Lbfgsb only): rows whose returned f is NaN in BOTH results are compared on
    status, counts and the NaN-ness of f alone — once an iterate holds NaN the reference sorts its breakpoints with
    std::sort over NaN keys (lbfgsb.h:298-304, :349), which is undefined behaviour (the comparator is no strict weak order),
    so WHICH coordinates the NaNs spread to from there is not a property of the reference the device could share
    (found by seed 63, trial 1028: every stopping threshold zero, iterates underflowing after ~400 iterations)."""
    bad = []
    keep = slice(None)
    if nan_keys_exempt:
        poisoned = np.isnan(dev[1]) & np.isnan(ora[1])
        same.exempt_rows += int(poisoned.sum())
        if poisoned.any():
            keep = ~poisoned
    for name, a, b in zip(("x", "f", "g"), dev[:3], ora[:3]):
        if name == "f":
            if not np.array_equal(a, b, equal_nan=True):
                bad.append(name)
        elif not np.array_equal(a[keep], b[keep], equal_nan=True):
            bad.append(name)
    for k in keys:
        if not np.array_equal(dev[3][k], ora[3][k]):
            bad.append(k)
    return bad


same.exempt_rows = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--solvers", default="lbfgs,lbfgs,lbfgs,lbfgsb,lbfgsb,bfgs,lbfgsb_relaxed,lbfgs_second,ridge_gram,ridge_mfma,lbfgs_wide",
                    help="comma-separated draw list (repeat a name to weight it)")
    ap.add_argument("--budget-s", type=float, default=0.0, help="stop drawing trials after this many seconds (0 = off)")
    ap.add_argument("--dump-dir", default="", help="write the inputs and both results of every mismatching trial there (.npz)")
    args = ap.parse_args()
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    import oracle_lib as O

    def engine_stop(s):
        d = capi.Stop()
        for name, _ in s._fields_:
            setattr(d, name, getattr(s, name))
        return d

    rng = np.random.default_rng(args.seed)
    ctx = amd.Context(0)
    t0 = time.time()
    counts = {"compared": 0, "refused": 0, "mismatch": 0}
    by_solver = {}
    for trial in range(args.trials):
        if args.budget_s and time.time() - t0 > args.budget_s:
            break
        solver = str(rng.choice(args.solvers.split(",")))
        objective = str(rng.choice(["rosenbrock", "rosenbrock", "diag_quadratic", "ridge"]))
        ls = str(rng.choice(["more_thuente", "more_thuente", "hager_zhang"]))
        if solver == "lbfgsb_relaxed":     # the relaxed-algebra kernels: More-Thuente, built-in objectives without row data
            ls = "more_thuente"
            if objective == "ridge":
                objective = "diag_quadratic"
        if solver in ("ridge_gram", "ridge_mfma"):   # the fast forms of the regression objective (configs[3])
            objective, ls = "ridge", "more_thuente"
        if solver == "lbfgs_wide":         # n > 256: the workgroup kernel (csrc/lbfgs_wide_kernel.hpp), exact arithmetic
            if objective == "ridge":
                objective = "rosenbrock"
        if solver == "lbfgs_second":       # Second mode with the Hessian from the functor (+ the condition_hessian test)
            objective = "rosenbrock"
        B = int(rng.choice([1, 2, 7, 33, 64, 129, 300]))
        rec = {"trial": trial, "solver": solver, "objective": objective, "linesearch": ls, "B": B}
        try:
            if solver == "bfgs":
                n = pick_n(rng, 64)
            elif solver == "lbfgs_wide":
                n = int(rng.choice([257, 300, 511, 512, 513, 1000, 2048, 4096, 4097])) if rng.random() < 0.5 else int(rng.integers(257, 3000))
                B = int(rng.choice([1, 3, 9]))
                rec["B"] = B
            elif solver == "ridge_gram":
                n = pick_n(rng, 256)
            elif objective == "ridge":
                n = pick_n(rng, 64)
            else:
                n = pick_n(rng, 256)
            m = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 10])) if solver.startswith("lbfgsb") else int(rng.choice([1, 2, 3, 5, 6, 7, 10, 13, 20, 32]))
            stop_o = random_stop(rng, O, solver.startswith("lbfgsb"))
            rec.update(n=n, m=m, stop={k: getattr(stop_o, k) for k, _ in stop_o._fields_})
            params = None
            per_problem = None
            if objective == "rosenbrock":
                obj, oname = amd.Rosenbrock(), "rosenbrock"
                x0 = starts(rng, amd, B, n, objective)
            elif objective == "diag_quadratic":
                a = rng.uniform(0.05, 40.0, n)
                c = float(rng.normal())
                obj, oname = amd.DiagQuadratic(a, c), "diag_quadratic"
                params = np.concatenate([a, [c]])
                x0 = starts(rng, amd, B, n, objective)
            else:
                rows = int(rng.choice([1, 5, 32, 64, 100, 128]))
                if solver == "ridge_gram" and rng.random() < 0.4:
                    rows = int(rng.choice([129, 300, 1000]))
                A = rng.normal(size=(rows, n))
                lam = float(rng.choice([0.0, 0.1, 3.0]))
                if lam == 0.0 and rows < n:
                    lam = 0.1
                obj, oname = amd.SquaredErrorRidge(A, lam), "squared_error_ridge"
                params = O.ridge_params(A, lam)
                per_problem = rng.normal(size=(B, rows))
                x0 = rng.normal(size=(B, n)) if rng.random() < 0.5 else np.zeros((B, n))
                rec.update(rows=rows, lam=lam)
            pp_dev = to_dev(torch, per_problem) if per_problem is not None else None
            if solver == "lbfgs":
                W = E = 0
                if rng.random() < 0.35:
                    cand = [(w, e) for w in (8, 16, 32, 64) for e in (1, 2, 4) if w * e >= n]
                    W, E = cand[int(rng.integers(0, len(cand)))]
                fused = objective != "ridge" and rng.random() < 0.4     # (round 6: Hager-Zhang has its fused kernels too)
                placement = int(rng.integers(0, 3))
                s = amd.BatchedLbfgs(m=m, stopping_progress=engine_stop(stop_o), context=ctx, lanes_per_problem=W, elems_per_lane=E,
                                     history_placement=placement, linesearch=ls, arithmetic="fma" if fused else "exact")
                x, f, g, p = s.minimize(obj, to_dev(torch, x0), per_problem=pp_dev)
                torch.cuda.synchronize()
                ll = s.last_launch()
                W, E = ll["lanes_per_problem"], ll["elems_per_lane"]
                rec.update(W=W, E=E, placement=placement, fused=bool(fused), y_regs=ll["y_columns_in_registers"])
                ora = O.minimize_batch(oname, x0, m=m, stop=stop_o, params=params, per_problem=per_problem, linesearch=ls,
                                       reduction="butterfly_fma" if fused else "butterfly", width=W * E, fma_group=E if fused else 0)
                keys = ("status", "num_iterations", "nfev", "sum_k")
            elif solver == "lbfgs_wide":
                mm = int(rng.choice([1, 3, 5, 6, 10, 17]))
                stop_o.num_iterations = min(int(stop_o.num_iterations), 300)
                rec.update(m=mm, stop={k: getattr(stop_o, k) for k, _ in stop_o._fields_})
                s = amd.BatchedLbfgs(m=mm, stopping_progress=engine_stop(stop_o), context=ctx, linesearch=ls, arithmetic="exact")
                x, f, g, p = s.minimize(obj, to_dev(torch, x0))
                torch.cuda.synchronize()
                T = s.last_launch()["threads"]
                rec.update(threads=T)
                ora = O.minimize_batch(oname, x0, m=mm, stop=stop_o, params=params, linesearch=ls, reduction="strided", width=T)
                keys = ("status", "num_iterations", "nfev", "sum_k")
            elif solver in ("ridge_gram", "ridge_mfma"):
                second = bool(rng.integers(0, 2))
                mm = min(m, 10)
                rec.update(m=mm, second=second)
                if solver == "ridge_gram":
                    obj = amd.SquaredErrorRidge(A, lam, differentiability="second" if second else "first", gram=True)
                    s = amd.BatchedLbfgs(m=mm, stopping_progress=engine_stop(stop_o), context=ctx, arithmetic="default")
                else:
                    obj = amd.SquaredErrorRidge(A, lam, differentiability="second" if second else "first", matrix_cores=True)
                    s = amd.BatchedLbfgs(m=mm, stopping_progress=engine_stop(stop_o), context=ctx, arithmetic="exact")
                x, f, g, p = s.minimize(obj, to_dev(torch, x0), per_problem=pp_dev)
                torch.cuda.synchronize()
                P = 8
                while P < n:
                    P *= 2
                if solver == "ridge_gram":
                    Eg = 1 if P == 8 else (4 if P == 256 else 2)
                    ora = O.minimize_batch("squared_error_ridge_gram", x0, m=mm, stop=stop_o, params=params, per_problem=per_problem,
                                           reduction="butterfly_fma", width=P, fma_group=Eg, second_mode=second)
                else:
                    ora = O.minimize_batch("squared_error_ridge_mfma", x0, m=mm, stop=stop_o, params=params, per_problem=per_problem,
                                           reduction="butterfly", width=64, second_mode=second)
                keys = ("status", "num_iterations", "nfev", "sum_k")
            elif solver == "lbfgs_second":
                fused = ls == "more_thuente" and rng.random() < 0.4
                threshold = float(rng.choice([0.0, 0.0, 3e3, 1e5, 1e13])) if n <= 64 else 0.0
                s = amd.BatchedLbfgs(m=m, stopping_progress=engine_stop(stop_o), context=ctx, linesearch=ls,
                                     arithmetic="fma" if fused else "exact", condition_hessian=threshold)
                x, f, g, p = s.minimize(amd.Rosenbrock(differentiability="second"), to_dev(torch, x0))
                torch.cuda.synchronize()
                ll = s.last_launch()
                W, E = ll["lanes_per_problem"], ll["elems_per_lane"]
                rec.update(W=W, E=E, fused=bool(fused), condition_hessian=threshold)
                O.lib().oracle_set_condition_hessian_stop(threshold)
                try:
                    ora = O.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, second_mode="functor", linesearch=ls,
                                           reduction="butterfly_fma" if fused else "butterfly", width=W * E, fma_group=E if fused else 0)
                    if threshold > 0:
                        co = O.hessian_conditions(B)
                        rec["condition_margin_min"] = float(np.min(np.abs(co - threshold) / threshold)) if len(co) else None
                finally:
                    O.lib().oracle_set_condition_hessian_stop(0.0)
                keys = ("status", "num_iterations", "nfev", "sum_k")
            elif solver in ("lbfgsb", "lbfgsb_relaxed"):
                relaxed = solver == "lbfgsb_relaxed"
                s = amd.BatchedLbfgsb(m=m, stopping_progress=engine_stop(stop_o), context=ctx, linesearch=ls,
                                      arithmetic="fma" if relaxed else "exact")
                lo = hi = None
                if rng.random() < 0.8:
                    centre = x0.mean(axis=0) if objective != "rosenbrock" else np.zeros(n)
                    lo = centre - rng.uniform(0.0, 2.0, n)
                    hi = centre + rng.uniform(0.0, 2.0, n)
                    lo[rng.random(n) < 0.2] = -np.inf
                    hi[rng.random(n) < 0.2] = np.inf
                    pin = rng.random(n) < 0.05
                    hi[pin] = lo[pin] = centre[pin]
                    s.SetBounds(lo, hi)
                x, f, g, p = s.minimize(obj, to_dev(torch, x0), per_problem=pp_dev)
                torch.cuda.synchronize()
                P = 8
                while P < n:
                    P *= 2
                if relaxed:
                    ora = O.lbfgsb_fast_minimize_batch(oname, x0, m=m, stop=stop_o, params=params, lower=lo, upper=hi)
                else:
                    ora = O.lbfgsb_minimize_batch(oname, x0, m=m, stop=stop_o, params=params, per_problem=per_problem, lower=lo,
                                                  upper=hi, reduction="butterfly", width=P, linesearch=ls)
                keys = ("status", "num_iterations", "nfev")
            else:
                P = 8
                while P < n:
                    P *= 2
                W = E = 0
                if rng.random() < 0.4:     # round 6: an explicit split of the padded width (same bits under every built split)
                    cand = [(w, e) for w in (8, 16, 32, 64) for e in (1, 2, 4) if w * e == P]
                    W, E = cand[int(rng.integers(0, len(cand)))]
                s = amd.BatchedBfgs(stopping_progress=engine_stop(stop_o), context=ctx, linesearch=ls, lanes_per_problem=W,
                                    elems_per_lane=E)
                x, f, g, p = s.minimize(obj, to_dev(torch, x0), per_problem=pp_dev)
                torch.cuda.synchronize()
                ll = s.last_launch()
                rec.update(W=ll["lanes_per_problem"], E=ll["elems_per_lane"], explicit=bool(W))
                ora = O.bfgs_minimize_batch(oname, x0, stop=stop_o, params=params, per_problem=per_problem, reduction="butterfly",
                                            width=P, linesearch=ls)
                keys = ("status", "num_iterations", "nfev")
            dev = (x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p))
            bad = same(dev, ora, keys, nan_keys_exempt=solver.startswith("lbfgsb"))
            rec["iterations_max"] = int(dev[3]["num_iterations"].max())
            rec["mismatch"] = bad
            counts["compared"] += 1
            by_solver[solver] = by_solver.get(solver, 0) + 1
            if bad:
                counts["mismatch"] += 1
                rows_bad = np.nonzero(np.any(dev[0] != ora[0], axis=1) | (dev[3]["num_iterations"] != ora[3]["num_iterations"]))[0]
                rec["first_bad_rows"] = [int(r) for r in rows_bad[:5]]
                rec["max_abs_dx"] = float(np.nanmax(np.abs(dev[0] - ora[0])))
                if args.dump_dir:
                    os.makedirs(args.dump_dir, exist_ok=True)
                    loc = locals()
                    extra = {k: np.asarray(loc[k]) for k in ("params", "per_problem", "lo", "hi", "A") if loc.get(k) is not None}
                    np.savez(os.path.join(args.dump_dir, "fuzz_seed%d_trial%d.npz" % (args.seed, trial)), x0=x0,
                             dev_x=dev[0], dev_f=dev[1], dev_g=dev[2], ora_x=ora[0], ora_f=ora[1], ora_g=ora[2],
                             dev_iters=dev[3]["num_iterations"], ora_iters=ora[3]["num_iterations"],
                             dev_status=dev[3]["status"], ora_status=ora[3]["status"],
                             dev_nfev=dev[3]["nfev"], ora_nfev=ora[3]["nfev"], **extra)
        except capi.EngineError as e:
            rec["refused"] = "%d: %s" % (e.code, str(e)[:160])
            counts["refused"] += 1
        print(json.dumps(rec), flush=True)
    summary = dict(counts, by_solver=by_solver, seed=args.seed, seconds=round(time.time() - t0, 1),
                   lbfgsb_rows_compared_on_status_and_counts_only_after_nan=same.exempt_rows)
    print(json.dumps({"summary": summary}), flush=True)
    ctx.close()
    return 1 if counts["mismatch"] else 0


if __name__ == "__main__":
    sys.exit(main())
