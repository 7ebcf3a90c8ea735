#!/usr/bin/env python
"""bench.py — L-BFGS solves/sec on batched Rosenbrock-N (BASELINE.json metric).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run with one rank per GPU.  A "step" is one pass of the hot
path over one batch: ONE launch of the fused solve kernel that runs every
L-BFGS iteration of every problem of this rank's shard, followed by the global
stop-flag all-reduce.  Inputs (x0) are resident in HBM before the timed region.

Workload (per GPU, weak scaling): BASELINE.json configs[1] — B = 65,536
Rosenbrock problems of dimension 32, L-BFGS m = 6, fp64, "parity stopping (B)"
(SURVEY.md section 7).  `--workload cfg3` selects the configs[2] shard instead
(131,072 problems of dimension 64, m = 10 per GPU).

Rank 0 prints ONE JSON line with `roofline` (algorithmic bytes of SURVEY.md
section 8d over the HIP-event kernel time) and, at N = 1, `cpu_baseline` (the
CPU oracle timed on this box's cores on a bounded prefix of the same batch).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B per GPU, n, m)
    "cfg2": dict(B=65536, n=32, m=6, desc="configs[1]: 65,536 x Rosenbrock-32, L-BFGS m=6, fp64"),
    "cfg3": dict(B=131072, n=64, m=10, desc="configs[2] shard: 131,072 x Rosenbrock-64, L-BFGS m=10, fp64"),
    "cfg4": dict(B=262144, n=64, m=10, rows=128, lam=0.1,
                 desc="configs[3]: 262,144 x SquaredError ridge (A 128x64 shared, y_b per problem, lambda 0.1, x0 = 0), "
                      "L-BFGS m=10, fp64, objective matrix-vector products on v_mfma_f64_16x16x4_f64"),
    "cfg5": dict(B=262144, n=32, m=5, lower=-1.5, upper=0.8, x0="u2",
                 desc="configs[4]: 262,144 x Rosenbrock-32 in the box [-1.5, 0.8]^32 via Lbfgsb (Cauchy point + subspace "
                      "minimisation), m=5, fp64"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 20260923


def algorithmic_bytes(n, iters, sum_k, rows=0):
    """B_solve = sum_t 8 n (6 + 2 k_t) = 8 n (6 T + 2 sum_k); the ridge
    objective adds 8*rows bytes per iteration (y_b) and A (8*rows*n) once per launch."""
    return 8.0 * n * (6.0 * float(iters) + 2.0 * float(sum_k)) + 8.0 * rows * float(iters) + 8.0 * rows * n


def lbfgsb_tight_stop(stop):
    """Tight stopping for the 1e-6 parity bar of the box-constrained workload (projected-gradient
    tolerance 1e-8 absolute, x_delta 1e-11, no f-delta / plateau test)."""
    stop.num_iterations = 10000
    stop.x_delta = 1e-11
    stop.x_delta_violations = 1
    stop.f_delta = 0.0
    stop.f_delta_relative = 0
    stop.gradient_norm = 1e-8
    stop.past = 0
    return stop


def cpu_baseline(x0_host, n, m, budget_s=12.0, objective="rosenbrock", params=None, per_problem=None, box=None,
                 linesearch="more_thuente"):
    """Time the CPU oracle (port of the reference algorithm) on a bounded prefix."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    stop = oracle_lib.parity_stop()
    cores = oracle_lib.lib().oracle_num_threads()
    probe = min(x0_host.shape[0], 64 * cores)

    def run(count):
        pp = per_problem[:count] if per_problem is not None else None
        if box is not None:
            return oracle_lib.lbfgsb_minimize_batch(objective, x0_host[:count], m=m, nthreads=cores,
                                                    stop=lbfgsb_tight_stop(oracle_lib.default_stop()),
                                                    lower=np.full(n, box[0]), upper=np.full(n, box[1]),
                                                    std_sort_order=True)
        return oracle_lib.minimize_batch(objective, x0_host[:count], m=m, stop=stop, nthreads=cores,
                                         params=params, per_problem=pp, linesearch=linesearch)
    t0 = time.perf_counter()
    run(probe)
    dt = time.perf_counter() - t0
    rate = probe / dt
    sample = int(min(x0_host.shape[0], max(probe, rate * budget_s)))
    t0 = time.perf_counter()
    xs, fs, _, ps = run(sample)
    dt = time.perf_counter() - t0
    return dict(value=sample / dt, unit="solves/s", cores=cores, kind="port",
                sample="first %d problems of the same batch, oracle/lbfgs_oracle.hpp (sequential order), "
                       "OpenMP schedule(dynamic) on %d threads, %.1f s" % (sample, cores, dt)), (xs, fs, ps, sample)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))  # cfg2..cfg5 = BASELINE.json configs[1..4]
    ap.add_argument("--batch", type=int, default=0, help="override problems per GPU")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per problem (0 = library default)")
    ap.add_argument("--elems", type=int, default=0, help="elements per lane (0 = library default)")
    ap.add_argument("--history", type=int, default=0, help="0 auto, 1 LDS ring, 2 y half in registers")
    ap.add_argument("--x0", default="std", choices=["std", "u2"])
    ap.add_argument("--ridge-valu", action="store_true",
                    help="cfg4: the exact-order VALU ridge kernel (objective id 2) instead of the matrix-core one")
    ap.add_argument("--linesearch", default="more_thuente", choices=["more_thuente", "hager_zhang"],
                    help="LineSearch template argument of Lbfgs (the BASELINE configs use the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "RANK" in os.environ  # under torch.distributed.run even at world size 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    Bg, n, m = wl["B"], wl["n"], wl["m"]
    if args.workload == "cfg5":
        solver = amd.BatchedLbfgsb(m=m, stopping_progress=lbfgsb_tight_stop(amd.capi.default_stop("lbfgsb")),
                                   device=local_rank)
        solver.SetBounds(np.full(n, wl["lower"]), np.full(n, wl["upper"]))
    else:
        solver = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop(), device=local_rank,
                                  lanes_per_problem=args.lanes, elems_per_lane=args.elems,
                                  history_placement=args.history, linesearch=args.linesearch)
    B_global = Bg * world
    lo, hi = sharded.shard_range(B_global, rank, world)
    rows = wl.get("rows", 0)
    per_problem = None
    ridge_host = None
    if args.workload == "cfg4":
        A_host, Y_host = amd.synthetic_ridge_host(hi - lo, rows, n, SEED, first_problem=lo)
        ridge_host = (A_host, Y_host)
        obj = amd.SquaredErrorRidge(A_host, wl["lam"], matrix_cores=not args.ridge_valu)
        per_problem = torch.from_numpy(Y_host).to(solver.device)     # resident in HBM
        x0 = torch.zeros(hi - lo, n, dtype=torch.float64, device=solver.device)
    else:
        obj = amd.Rosenbrock()
        x0 = solver.fill_x0(hi - lo, n, wl.get("x0", args.x0), SEED, first_problem=lo)  # resident in HBM
    torch.cuda.synchronize()

    def step():
        x, f, g, prog = solver.minimize(obj, x0, per_problem=per_problem)
        status, iters, nfev, sum_k = sharded.progress_fields_device(prog)
        flag = sharded.allreduce_flag(sharded.local_counts(status, iters))
        return (x, f, g, prog), flag

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, flag = step()
        kernel_ms.append(solver.last_kernel_ms())  # HIP events on the launch stream
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    x, f, g, prog = out
    pn = amd.progress_to_numpy(prog)
    iters_sum, sumk_sum, nfev_sum = int(pn["num_iterations"].sum()), int(pn["sum_k"].sum()), int(pn["nfev"].sum())
    bytes_launch = algorithmic_bytes(n, iters_sum, sumk_sum, rows)
    k_ms = float(np.mean(kernel_ms))
    achieved = bytes_launch / (k_ms * 1e-3) / 1e9
    value = B_global * args.steps / elapsed
    launch = solver.last_launch()

    result = {
        "metric": "L-BFGS solves/sec (batched Rosenbrock-N)",
        "value": value,
        "unit": "solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl["desc"] + ("; Hager-Zhang line search" if args.linesearch == "hager_zhang" else "") +
                        "; x0 '%s' seed %d; parity stopping (B): x_delta=1e-11, "
                        "gradient_norm=1e-8 %s, past=0, 10000 iterations" % (
                            "zero" if rows else wl.get("x0", args.x0), SEED,
                            "absolute on the projected gradient" if args.workload == "cfg5" else "relative"),
            "problems_per_gpu": Bg, "n": n, "m": m, "parallelism": "batch-sharded x%d" % world,
            "lanes_per_problem": launch["lanes_per_problem"], "elems_per_lane": launch["elems_per_lane"],
            "grid_workgroups": launch["blocks"], "threads_per_workgroup": launch["threads"],
            "lds_bytes_per_workgroup": launch["lds_bytes"],
            "y_columns_in_registers": launch["y_columns_in_registers"],
            "mean_iterations": iters_sum / float(len(pn)), "mean_nfev": nfev_sum / float(len(pn)),
            "all_converged": bool(flag.all_converged), "unconverged": int(flag.unconverged),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": None,
            "kernel": ("lbfgsb_solve_kernel<%d,Rosenbrock,5>" % launch["elems_per_lane"]) if args.workload == "cfg5" else
                      "ridge_mfma_solve_kernel<10>" if (rows and not args.ridge_valu) else
                      "lbfgs_solve_kernel<%d,%d,%s,%d>" % (launch["lanes_per_problem"], launch["elems_per_lane"],
                                                           "SquaredErrorRidge" if rows else "Rosenbrock",
                                                           launch["y_columns_in_registers"]),
            "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": bytes_launch,
            "note": "algorithmic bytes = sum_b 8n(6T_b + 2 sum_k_b) (state-streaming model, SURVEY 8d); the fused "
                    "kernel keeps that state in registers/LDS, so achieved may exceed physical HBM bandwidth",
        },
    }
    if rows and not args.ridge_valu:
        # the matrix-core kernel is priced against the dense f64 MFMA peak as well: every objective
        # evaluation is 2 * 2 * rows * n flops on v_mfma_f64_16x16x4_f64 (MI355X_MICROARCH.md: 78.6 TFLOP/s)
        flops = float(nfev_sum) * 4.0 * rows * n
        result["roofline_mfma"] = {
            "bound": "mfma", "achieved": flops / (k_ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
            "frac": flops / (k_ms * 1e-3) / 1e12 / 78.6, "flops_per_launch": flops,
            "note": "objective matrix-vector products only; the rest of the iteration runs on the VALU "
                    "(DESIGN.md section 3.4 for the phase shares)"}
    traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(traffic_file):
        try:
            tr = json.load(open(traffic_file)).get(args.workload)
            if tr:
                result["roofline"]["traffic"] = tr["bytes_per_launch"]
                result["roofline"]["traffic_source"] = tr.get("source", "profiles/")
        except Exception:
            pass

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        x0h = x0.cpu().numpy()
        if ridge_host is not None:
            cb, (xs, fs, ps, sample) = cpu_baseline(
                x0h, n, m, objective="squared_error_ridge",
                params=np.concatenate([[float(rows), wl["lam"]], ridge_host[0].ravel()]), per_problem=ridge_host[1])
        elif args.workload == "cfg5":
            cb, (xs, fs, ps, sample) = cpu_baseline(x0h, n, m, box=(wl["lower"], wl["upper"]))
            cb["sample"] = cb["sample"].replace("lbfgs_oracle.hpp", "lbfgsb_oracle.hpp")
        else:
            cb, (xs, fs, ps, sample) = cpu_baseline(x0h, n, m, linesearch=args.linesearch)
        result["cpu_baseline"] = cb
        xh, fh = x.cpu().numpy()[:sample], f.cpu().numpy()[:sample]
        result["config"]["parity_vs_cpu_sample"] = {
            "problems": int(sample), "max_abs_dx": float(np.max(np.abs(xh - xs))),
            "max_abs_df": float(np.max(np.abs(fh - fs))), "tol": 1e-6}

    if rank == 0 and world == 1 and not args.no_secondary:
        # PCIe-inclusive rate through the host-pointer entry point (pageable host memory); informational
        x0h = x0.cpu().numpy()
        pph = ridge_host[1] if ridge_host is not None else None
        solver.minimize_host(obj, x0h[:1024], per_problem=pph[:1024] if pph is not None else None)
        t0 = time.perf_counter()
        solver.minimize_host(obj, x0h, per_problem=pph)
        dth = time.perf_counter() - t0
        result["config"]["pcie_inclusive_host_entry"] = {"value": x0h.shape[0] / dth, "unit": "solves/s",
                                                         "ms": dth * 1e3}

    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "cfg2":
        # secondary figure: the configs[2] per-GPU shard (n=64, m=10), same protocol, 1 warm + 2 timed
        w3 = WORKLOADS["cfg3"]
        s3 = amd.BatchedLbfgs(m=w3["m"], stopping_progress=amd.parity_stop(), context=solver.ctx)
        x03 = s3.fill_x0(w3["B"], w3["n"], args.x0, SEED)
        s3.minimize(amd.Rosenbrock(), x03)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ms3 = []
        for _ in range(2):
            o3 = s3.minimize(amd.Rosenbrock(), x03)
            ms3.append(s3.last_kernel_ms())
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        p3 = amd.progress_to_numpy(o3[3])
        b3 = algorithmic_bytes(w3["n"], int(p3["num_iterations"].sum()), int(p3["sum_k"].sum()))
        result["config"]["secondary_cfg3_shard"] = {
            "workload": w3["desc"], "value": 2 * w3["B"] / dt3, "unit": "solves/s",
            "kernel_ms": float(np.mean(ms3)), "achieved_GBs": b3 / (np.mean(ms3) * 1e-3) / 1e9,
            "mean_iterations": float(p3["num_iterations"].mean())}

    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "cfg2":
        # secondary figure: constrained solves through the augmented-Lagrangian outer loop (the full
        # line with its CPU baseline and parity check comes from scripts/auglag_bench.py): 16 384 problems of n = 64,
        #   min sum_i a_i x_i^2 + c  s.t.  sum x = 1,  x_0 <= 0.2,  penalty auto-scaled, outer limit 40
        nal, Bal = 64, 16384
        rng = np.random.default_rng(3)
        T = amd.ConstrainedProblem.term
        e0 = np.zeros(nal)
        e0[0] = 1.0
        prob = amd.ConstrainedProblem(nal, T("diag_quadratic", a=rng.uniform(0.5, 4.0, nal), c=0.5),
                                      [T("linear", "value_minus_k", 1.0, a=np.ones(nal))],
                                      [T("linear", "k_minus_value", 0.2, a=e0)])
        al = amd.BatchedAugmentedLagrangian(context=solver.ctx)
        al.config.outer_num_iterations = 40
        dev = solver.device
        xa0 = torch.from_numpy(np.random.default_rng(SEED).uniform(-1, 1, (Bal, nal))).to(dev)

        def al_step():
            xa = xa0.clone()
            lam = torch.zeros(Bal, 1, dtype=torch.float64, device=dev)
            mu = torch.zeros(Bal, 1, dtype=torch.float64, device=dev)
            pen = torch.zeros(Bal, dtype=torch.float64, device=dev)
            return al.minimize(prob, xa, lam, mu, pen)

        al_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            viol, kkt, prog = al_step()
        torch.cuda.synchronize()
        dta = (time.perf_counter() - t0) / 3
        pa = amd.al_progress_to_numpy(prog)
        result["config"]["secondary_augmented_lagrangian"] = {
            "workload": "16,384 constrained problems, n = 64: diagonal quadratic, one equality, one inequality; "
                        "Lbfgs<m=10> inner solver, whole outer loop in one kernel launch",
            "value": Bal / dta, "unit": "solves/s", "ms": dta * 1e3,
            "finished_fraction": float(np.mean(pa["status"] == 6)), "max_violation": float(viol.max().item()),
            "mean_outer_iterations": float(pa["num_iterations"].mean()),
            "mean_inner_iterations": float(pa["inner_iterations"].mean())}

    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
