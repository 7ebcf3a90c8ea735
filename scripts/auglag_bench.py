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
    ap.add_argument("--counters", action="store_true",
                    help="re-run this script under rocprofv3 --pmc (bench.py's passes: FETCH_SIZE, WRITE_SIZE, SQ, fp64 "
                         "instruction classes) and add the binding fraction, VALU-busy and measured HBM traffic to the line; "
                         "also times the reference binary (oracle/_ref/libref.so, its own AugmentedLagrangian + Lbfgs) beside "
                         "the port: the SURVEY 8(f3) row of profiles/README.md's table")
    ap.add_argument("--no-cpu", action="store_true", help="(the counter passes' child runs: GPU only)")
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

    if args.no_cpu:
        print(json.dumps({"value": args.batch / dt, "ms_per_step": dt * 1e3}))
        return
    # CPU oracle on a sample: timing (all host threads) and parity (sequential policy = the reference's arithmetic)
    k = min(args.cpu_sample, args.batch)

    def run_oracle():
        return (al.oracle_box_minimize(p, x0[:k], lower=lower, upper=upper, config=cfg, std_sort_order=False, inner_stop=ostop)
                if box else al.oracle_minimize(p, x0[:k], config=cfg, inner_stop=ostop, penalty0=1.0 if args.svm_primal else 0.0))

    t1 = time.perf_counter()
    o = run_oracle()
    cpu_dt = time.perf_counter() - t1
    if args.counters and cpu_dt < 5.0:     # the table's row: warm-up (the run above) + 3 timed repetitions, median — as bench.py
        reps = []
        for _ in range(3):
            t1 = time.perf_counter()
            run_oracle()
            reps.append(time.perf_counter() - t1)
        cpu_dt = float(np.median(reps))
    dx = np.abs(x.cpu().numpy()[:k] - o["x"]).max()
    same_status = float(np.mean(pr["status"][:k] == o["progress"]["status"]))
    def finish(d):
        d["roofline"]["frac"] = d["roofline"]["achieved"] / d["roofline"]["peak"]
        if args.counters:
            counters_and_reference(d)
        return d

    def counters_and_reference(d):
        """The keys bench.py's line carries (roofline.binding first, roofline_valu, config, cpu_reference), so that
        scripts/profiles_table.py prints this row next to the configs."""
        import bench
        from concurrent.futures import ThreadPoolExecutor
        T = float(pr["inner_iterations"].astype(np.float64).sum())
        sum_k = float(pr["sum_k"].astype(np.float64).sum())
        nfev = float(pr["nfev"].astype(np.float64).sum())
        # useful flops of the inner solves: the Lbfgs model of bench.py with the composite's evaluation cost per coordinate:
        # objective a_i x_i^2 (3) + two affine constraints (a . x and the gradient axpy: 4 each) + the penalty assembly (2)
        c_obj = 3.0 + 4.0 * (p.n_eq + p.n_ineq) + 2.0
        flops = 8.0 * args.n * sum_k + 22.0 * args.n * T + (4.0 + c_obj) * args.n * nfev
        child = ["--batch", str(args.batch), "--n", str(args.n), "--steps", "1", "--warmup", "1", "--outer-limit",
                 str(args.outer_limit), "--loop", args.loop, "--inner-limit", str(args.inner_limit), "--inner", args.inner,
                 "--no-cpu"] + (["--svm-primal"] if args.svm_primal else [])
        lc = bench.live_counters(child, script=os.path.abspath(__file__), calls=2,
                                 kernels=("_solve_kernel", "auglag_outer_kernel"))
        rf = d["roofline"]
        rf.update(bound="hbm-state-streaming-model", kernel="lbfgs_solve_kernel<..., AugLagComposite, OUTER = the fused outer "
                  "loop> (+ auglag_outer_kernel in the lock-step loop)", kernel_ms=dt * 1e3, traffic=lc.get("traffic"),
                  traffic_source=lc.get("traffic_source"),
                  hbm_frac_measured=(lc["traffic"] / dt / 1e9 / bench.HBM_PEAK_GBS) if "traffic" in lc else None)
        for key in ("traffic_error", "sq_error", "flops_error"):
            if key in lc:
                rf[key] = lc[key]
        rv = {"bound": "valu-fp64", "achieved": flops / dt / 1e12, "peak": bench.FP64_VALU_PEAK_TF, "unit": "TFLOP/s",
              "frac": flops / dt / 1e12 / bench.FP64_VALU_PEAK_TF, "frac_of_fma_peak": flops / dt / 1e12 / bench.FP64_VALU_PEAK_TF,
              "useful_flops_per_launch": flops, "executed_flops": lc.get("executed_flops"),
              "frac_executed": (lc["executed_flops"] / dt / 1e12 / bench.FP64_VALU_PEAK_TF) if "executed_flops" in lc else None,
              "valu_busy": lc.get("valu_busy"),
              "note": "useful flops = 8 n sum_k + 22 n T + (4 + c_obj) n nfev over the inner iterations of all outer steps, "
                      "c_obj = %.0f per coordinate (diagonal quadratic + %d affine constraints + penalty assembly); time = wall "
                      "time of the whole call (outer-step work and read-backs included)" % (c_obj, p.n_eq + p.n_ineq)}
        d["roofline_valu"] = rv
        # exact leg: the device against its twin (the oracle in the kernel's own summation order), bit for bit on a subsample
        twin_equal = None
        if not (box or args.svm_primal):
            P = 8
            while P < args.n:
                P *= 2
            kt = min(k, 256)
            ot = al.oracle_minimize(p, x0[:kt], config=cfg, inner_stop=ostop, reduction="butterfly", width=P)
            twin_equal = bool(np.array_equal(x.cpu().numpy()[:kt], ot["x"]) and
                              np.array_equal(pr["inner_iterations"][:kt], ot["progress"]["inner_iterations"]))
        d["config"] = {"workload": "SURVEY 8(f3): %d constrained problems, n = %d: %s" % (args.batch, args.n, d["workload"]),
                       "parity_vs_cpu_sample": {"problems": int(k), "max_abs_dx": float(dx), "tol": 1e-3,
                                                "against": "oracle/auglag_oracle.hpp, sequential policy (== the reference "
                                                           "binary bit for bit, tests/test_auglag_oracle.py); tol = the "
                                                           "reference's own test tolerance for this solver (1e-3 primal: the "
                                                           "outer loop stops at a KKT norm of 1e-4, so two summation orders "
                                                           "end ~1e-4 apart); the exact statement is twin_bitwise_equal",
                                                "twin_bitwise_equal": twin_equal, "twin_problems": 256 if twin_equal is not None else 0}}
        # the reference binary beside the port: its own AugmentedLagrangian<Problem, Lbfgs<FunctionExprXd>> over the Eigen
        # shim (oracle/_ref/libref.so, the PINNED -O2 build: there is no -O3 timing build of this entry point), a thread
        # pool pulling chunks of 16 problems
        if not (box or args.svm_primal):
            try:
                cores = bench.cpu_threads(oracle_lib.lib().oracle_num_threads())
                kr = min(k, max(16 * cores, 256))
                chunks = [(b0, min(kr, b0 + 16)) for b0 in range(0, kr, 16)]
                al.ref_minimize(p, x0[:1], config=cfg, inner_stop=ostop)     # (loads and binds the library once)

                def one(c):
                    return al.ref_minimize(p, x0[c[0]:c[1]], config=cfg, inner_stop=ostop)["x"]

                def run_ref():
                    with ThreadPoolExecutor(max_workers=cores) as pool:
                        return np.concatenate(list(pool.map(one, chunks)))
                med, ts = bench._timed(run_ref)
                xr = run_ref()
                d["cpu_reference"] = {"value": kr / med, "unit": "solves/s", "cores": cores, "kind": "reference-over-shim",
                                      "sample": "first %d problems, the reference's solver/augmented_lagrangian.h + lbfgs.h over "
                                                "oracle/eigen_shim (oracle/_ref/libref.so, -O2 pinned build), %d threads pulling "
                                                "chunks of 16; warm-up + 3 timed repetitions, median %.2f s" % (kr, cores, med),
                                      "max_abs_dx_device_vs_reference": float(np.abs(x.cpu().numpy()[:kr] - xr).max())}
            except Exception as e:  # noqa: BLE001
                d["cpu_reference"] = {"value": None, "error": "%s: %s" % (type(e).__name__, e)}
        res = {"roofline": rf, "roofline_valu": rv}
        bench.physical_roofline(res)
        d["roofline"] = res["roofline"]

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
                         "sample": "%d problems of the same batch, oracle/auglag_oracle.hpp (strict build), OpenMP%s" % (
                             k, "; warm-up + 3 timed repetitions, median" if args.counters else "; one run")},
        "parity": {"max_abs_dx_vs_oracle_sequential": float(dx), "same_status_fraction": same_status},
    })))


if __name__ == "__main__":
    main()
