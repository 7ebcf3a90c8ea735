#!/usr/bin/env python3
"""Throughput of the augmented-Lagrangian path on one MI355X, with the CPU oracle
timed beside it and a parity check on a sample.

    python scripts/auglag_bench.py [--batch 65536] [--n 12] [--steps 3] [--cpu-sample 2048]

One step = one mi355_auglag_minimize_batch call: every outer iteration is one launch of the persistent L-BFGS
kernel over the problems still active plus one launch of the outer-step kernel, with a 4-byte read-back.
Prints one JSON line.  (The oracle import is the checker / CPU baseline, as in bench.py.)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--n", type=int, default=12)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=2048)
    ap.add_argument("--outer-limit", type=int, default=40)
    ap.add_argument("--loop", default="auto", choices=["auto", "fused", "lockstep"])
    ap.add_argument("--inner-limit", type=int, default=10000, help="num_iterations of the inner solver's stopping test")
    ap.add_argument("--inner", default="lbfgs", choices=["lbfgs", "lbfgsb"],
                    help="lbfgsb: Lbfgsb<F, 5> as the inner solver with the box [-1, 0.15]^n (n <= 64)")
    ap.add_argument("--svm-primal", action="store_true",
                    help="the shape of the reference's src/examples/svm_primal_al.cc: 105 variables, 200 affine inequality "
                         "constraints as ONE constraint family (mi355_al_problem.family_ineq), penalty 1, starts around 0")
    args = ap.parse_args()
    import torch
    import auglag_lib as al
    from cppnumericalsolvers_amd import BatchedAugmentedLagrangian, ConstrainedProblem, capi

    rng = np.random.default_rng(20260923)
    mk = lambda t: ConstrainedProblem.term(t["prims"], t["form"], t["k"])
    if args.svm_primal:
        p, _, _ = al.svm_primal_al_problem()
        args.n = p.n
        x0 = rng.uniform(-0.5, 0.5, (args.batch, args.n))
        x0[0] = 0.0
    else:
        p = al.quadratic_simplex_problem(args.n, seed=3)
        x0 = rng.uniform(-1, 1, (args.batch, args.n))
    ep = ConstrainedProblem(p.n, mk(p.terms[0]), [mk(t) for t in p.table_eq], [mk(t) for t in p.table_ineq],
                            family_equality=p.family_equality, family_inequality=p.family_inequality)
    cfg = al.default_config(outer_num_iterations=args.outer_limit)
    box = args.inner == "lbfgsb"
    lower, upper = (np.full(args.n, -1.0), np.full(args.n, 0.15)) if box else (None, None)
    s = BatchedAugmentedLagrangian(inner=args.inner, lower=lower, upper=upper)
    s.inner_stopping_progress.num_iterations = args.inner_limit
    import oracle_lib
    ostop = oracle_lib.lbfgsb_default_stop() if box else oracle_lib.default_stop()
    ostop.num_iterations = args.inner_limit
    for name, _ in cfg._fields_:
        setattr(s.config, name, getattr(cfg, name))
    s.config.loop = capi.AL_LOOP[args.loop]
    dev = torch.device("cuda:0")
    x0_dev = torch.from_numpy(x0).to(dev)

    def step():
        x = x0_dev.clone()
        lam = torch.zeros(args.batch, max(p.n_eq, 1), dtype=torch.float64, device=dev)
        mu = torch.zeros(args.batch, max(p.n_ineq, 1), dtype=torch.float64, device=dev)
        pen = torch.full((args.batch,), 1.0 if args.svm_primal else 0.0, dtype=torch.float64, device=dev)
        viol, kkt, prog = s.minimize(ep, x, lam, mu, pen)
        return x, lam, mu, pen, viol, kkt, prog

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    x, lam, mu, pen, viol, kkt, prog = out
    pr = prog.cpu().numpy().view(capi.AL_PROGRESS_DTYPE)

    # CPU oracle on a sample: timing (all host threads) and parity (sequential policy = the reference's arithmetic)
    k = min(args.cpu_sample, args.batch)
    t1 = time.perf_counter()
    o = (al.oracle_box_minimize(p, x0[:k], lower=lower, upper=upper, config=cfg, std_sort_order=False, inner_stop=ostop)
         if box else al.oracle_minimize(p, x0[:k], config=cfg, inner_stop=ostop, penalty0=1.0 if args.svm_primal else 0.0))
    cpu_dt = time.perf_counter() - t1
    dx = np.abs(x.cpu().numpy()[:k] - o["x"]).max()
    same_status = float(np.mean(pr["status"][:k] == o["progress"]["status"]))
    def finish(d):
        d["roofline"]["frac"] = d["roofline"]["achieved"] / d["roofline"]["peak"]
        return d

    print(json.dumps(finish({
        "metric": "augmented-Lagrangian solves/s", "value": args.batch / dt, "unit": "solves/s",
        "ms_per_step": dt * 1e3, "batch": args.batch, "n": args.n, "loop": args.loop, "inner": args.inner, "inner_limit": args.inner_limit, "n_eq": p.n_eq, "n_ineq": p.n_ineq,
        "workload": ("primal soft-margin SVM (svm_primal_al.cc): 105 variables, 200 affine inequality constraints as one "
                     "constraint family, penalty 1; " if args.svm_primal else
                     "min sum a_i x_i^2 + c  s.t.  sum x = 1, x_0 <= 0.2; penalty auto-scaled; ") +
                    ("Lbfgsb<m=5> inner solver, box [-1, 0.15]^n" if box else "Lbfgs<m=10> inner solver"),
        "outer_iterations_mean": float(pr["num_iterations"].mean()), "outer_iterations_max": int(pr["num_iterations"].max()),
        "inner_iterations_mean": float(pr["inner_iterations"].mean()),
        "finished_fraction": float(np.mean(pr["status"] == 6)), "max_violation_max": float(viol.max().item()),
        # state-streaming model of the inner solves (8n(6T + 2 sum_k) bytes per solve) over the WHOLE call,
        # outer-step kernels and read-backs included
        "roofline": {"bound": "hbm", "unit": "GB/s", "peak": 8000.0,
                     "achieved": float(8 * args.n * (6 * pr["inner_iterations"].astype(np.float64).sum()
                                                     + 2 * pr["sum_k"].astype(np.float64).sum()) / dt / 1e9),
                     "note": "algorithmic bytes of the inner L-BFGS iterations / wall time of the call"},
        "cpu_baseline": {"value": k / cpu_dt, "unit": "solves/s", "cores": os.cpu_count(), "kind": "port",
                         "sample": "%d problems of the same batch, oracle/auglag_oracle.hpp, OpenMP" % k},
        "parity": {"max_abs_dx_vs_oracle_sequential": float(dx), "same_status_fraction": same_status},
    })))


if __name__ == "__main__":
    main()
