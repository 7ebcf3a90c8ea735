#!/usr/bin/env python
"""bench.py — L-BFGS solves/sec on batched Rosenbrock-N (BASELINE.json metric).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 launched by
torch.distributed.run with one rank per GPU — or, when no RANK / WORLD_SIZE is in
the environment, the script re-executes ITSELF under torch.distributed.run with N
ranks (launch_plan below) and exits non-zero when fewer than N GPUs are visible:
a line is only printed when N distinct devices solved and N ranks took part in
the all-reduce (`n_gpus`, `multi_gpu.rccl_ranks`).  A "step" is one pass of the hot
path over one batch: ONE launch of the fused solve kernel that runs every
L-BFGS iteration of every problem of this rank's shard, followed by the global
stop-flag all-reduce.  Inputs (x0) are resident in HBM before the timed region.

Workload (per GPU, weak scaling): BASELINE.json configs[1] — B = 65,536
Rosenbrock problems of dimension 32, L-BFGS m = 6, fp64, "parity stopping (B)"
(SURVEY.md section 7).  `--workload cfg3` selects the configs[2] shard
(131,072 problems of dimension 64, m = 10 per GPU), `--workload cfg3full` the
whole configs[2] batch (1,048,576 problems, sharded over the ranks: strong
scaling, the G = 1, 2, 4, 8 rows of BASELINE.md section 4).

Rank 0 prints ONE JSON line — at EVERY N: under `--gpus N > 1` rank 0 still runs the counter passes (on its own device, its
shard as a one-GPU child run), the CPU legs and the parity sample while the other ranks are parked on a host-side (gloo)
group, after all collectives of the run.  Next to the contract's fields it carries
  roofline       the state-streaming byte model of SURVEY.md section 8d over the HIP-event kernel time (a
                 throughput in the units the north star asked for — the fused kernel keeps that state on chip),
                 with `traffic` = HBM bytes per launch MEASURED in this run by two rocprofv3 --pmc child passes, and ONE HOP
                 from there the binding fraction: `frac_physical` = max(measured HBM fraction, issued fp64 lane-flops /
                 78.6 TFLOP/s), `bound_physical` ("hbm" / "valu-fp64"), `useful_frac`, `valu_busy`
  roofline_valu  what physically bounds the kernel: useful fp64 flop/s against the VALU peak (Lbfgsb: the reference's
                 operation count per step, counted by the oracle), and the VALU-busy fraction from a third counter pass
  cpu_baseline   the CPU oracle ("port") on this box's cores: warm-up + 3 timed repetitions, median
  cpu_reference  the reference's own headers (oracle/_ref, over the Eigen shim) under the same protocol
  north_star     BASELINE.json's target workload (1,048,576 x Rosenbrock-64, m = 10) next to `value`: strong-scaled over the
                 ranks of this run (N > 1), the whole batch on the one GPU (N = 1)
The run itself is `run_bench(args, plan, runtime)`; everything that touches the GPU or the process group sits behind the
runtime object (GpuRuntime here), so that tests/test_bench_line_gloo.py can drive the same code at world size 2 over gloo.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B per GPU, n, m)
    "cfg2": dict(B=65536, n=32, m=6, desc="configs[1]: 65,536 x Rosenbrock-32, L-BFGS m=6, fp64"),
    "cfg3": dict(B=131072, n=64, m=10, desc="configs[2] shard: 131,072 x Rosenbrock-64, L-BFGS m=10, fp64"),
    "cfg3full": dict(B=1048576, n=64, m=10, strong=True,
                     desc="configs[2]: 1,048,576 x Rosenbrock-64, L-BFGS m=10, fp64 (whole batch, sharded over the ranks)"),
    "cfg4": dict(B=262144, n=64, m=10, rows=128, lam=0.1,
                 desc="configs[3]: 262,144 x SquaredError ridge (A 128x64 shared, y_b per problem, lambda 0.1, x0 = 0), "
                      "L-BFGS m=10, fp64"),
    # beyond BASELINE.json: the README ridge objective on a matrix larger than configs[3]'s (README.md:126-160 takes any A):
    # normal-equation form with G streamed through L2 (n > 128), one wavefront per problem
    "cfg4big": dict(B=32768, n=200, m=10, rows=1000, lam=0.1,
                    desc="beyond BASELINE.json: 32,768 x SquaredError ridge (A 1000x200 shared, y_b per problem, lambda 0.1, "
                         "x0 = 0), L-BFGS m=10, fp64"),
    # beyond BASELINE.json: the README objective built once PER DATA SET — every problem has its own matrix A_b (17 GB of
    # matrices at the configs[3] batch size; 65,536 problems = 4.3 GB here), normal equations per problem
    "cfg4own": dict(B=65536, n=64, m=10, rows=128, lam=0.1, own=True,
                    desc="beyond BASELINE.json: 65,536 x SquaredError ridge with ONE MATRIX PER PROBLEM (A_b 128x64, y_b, lambda 0.1, "
                         "x0 = 0), L-BFGS m=10, fp64"),
    "cfg5": dict(B=262144, n=32, m=5, lower=-1.5, upper=0.8, x0="u2",
                 desc="configs[4]: 262,144 x Rosenbrock-32 in the box [-1.5, 0.8]^32 via Lbfgsb (Cauchy point + subspace "
                      "minimisation), m=5, fp64"),
    # SURVEY.md section 8(f) rows measured like the configs (round-5 verdict, "Next" 4): the configs[1] batch shape under
    # the Hager-Zhang line search and under dense BFGS, and the configs[3] objective declared Second mode
    "f_hz": dict(B=65536, n=32, m=6, linesearch="hager_zhang",
                 desc="SURVEY 8(f2): 65,536 x Rosenbrock-32, Lbfgs<F, 6, HagerZhang>, fp64"),
    "f_bfgs": dict(B=65536, n=32, m=1, solver="bfgs",
                   desc="SURVEY 8(f4): 65,536 x Rosenbrock-32, dense Bfgs<F, MoreThuente> (explicit 32 x 32 inverse-Hessian "
                        "approximation per problem in LDS), fp64"),
    "f_second": dict(B=65536, n=64, m=10, rows=128, lam=0.1, second=True,
                     desc="SURVEY 8(f1): 65,536 x SquaredError ridge (A 128x64 shared, y_b per problem, lambda 0.1, x0 = 0) declared "
                          "Second mode: Lbfgs m=10 centred on the diagonal preconditioner 1/(|H_jj| + eps) (lbfgs.h:116-139), fp64"),
    # beyond BASELINE.json: a problem larger than a wavefront holds (n > 256) is owned by a workgroup and its vectors and
    # correction ring live in HBM (csrc/lbfgs_wide_kernel.hpp) -- the regime where the state-streaming model is physical
    "wide": dict(B=2048, n=4096, m=10, x0="u2", objective="diag_quadratic",
                 desc="beyond BASELINE.json: 2,048 x DiagQuadratic-4096 (a_i in [0.5, 20], c = 1.5), L-BFGS m=10, fp64, one "
                      "problem per workgroup with its state in HBM"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6300.0  # MI355X_MICROARCH.md: measured-achievable streaming ceiling (SURVEY 8d asks for both)
FP64_VALU_PEAK_TF = 78.6   # 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2 flop (FMA) x 2.4 GHz (public datasheet figure)
SEED = 20260923


def algorithmic_bytes(n, iters, sum_k, rows=0, own_matrices=0):
    """B_solve = sum_t 8 n (6 + 2 k_t) = 8 n (6 T + 2 sum_k); the ridge
    objective adds 8*rows bytes per iteration (y_b) and A (8*rows*n) once per launch — or, with one matrix per problem,
    once per problem."""
    return 8.0 * n * (6.0 * float(iters) + 2.0 * float(sum_k)) + 8.0 * rows * float(iters) + \
        8.0 * rows * n * max(1, int(own_matrices))


def algorithmic_flops(n, iters, sum_k, nfev, rows=0, ridge_form=None):
    """Flops the kernels EXECUTE per launch, priced from the algorithm (the counter pass below measures them): per
    iteration 8 n k for the two-loop recursion (k dot products + k axpys per loop, 2 flops per coordinate each: the
    kernels keep rho_i = 1 / s_i.y_i from the update, where the reference recomputes s_i.y_i in both loops —
    SURVEY 8d's 12 n k prices that recomputation) + 22 n (descent test, s / y, three inner products, two sup norms,
    scaling), and per evaluation (4 + c_obj) n: trial point 2, directional derivative 2, objective c_obj = 15 for
    Rosenbrock, 4 rows for the two matrix-vector products of the ridge objective in the reference's form
    (ridge_form "direct" / "mfma"), 2 n + 4 for the normal-equation form (t = G x, then h, grad, f) whose per-problem
    pre-pass c_b = A^T y_b adds 2 rows n once per solve."""
    if rows and ridge_form == "gram":
        c_obj = 2.0 * n + 4.0
    elif rows:
        c_obj = 4.0 * rows
    else:
        c_obj = 15.0
    return 8.0 * n * float(sum_k) + 22.0 * n * float(iters) + (4.0 + c_obj) * n * float(nfev)


def bfgs_algorithmic_bytes(n, iters):
    """Dense BFGS in the state-streaming convention of SURVEY 8d: per iteration the n x n approximation is read for
    d = -H g, read for H y and read + written by the rank-two update (4 passes), plus the six vectors of the Lbfgs model."""
    return 8.0 * (4.0 * n * n + 6.0 * n) * float(iters)


def bfgs_algorithmic_flops(n, iters, nfev):
    """bfgs.h:81,123-134 per iteration: H g and H y (2 n^2 each), the update H - rho (s Hy^T + Hy s^T) + c s s^T (8 n^2:
    two outer products 1 n^2 each, their sum, the scaling, the subtraction, s s^T, its scaling, the addition) + the 22 n of
    vector work of the Lbfgs model; per evaluation (4 + 15) n as for Lbfgs on Rosenbrock."""
    return (12.0 * n * n + 22.0 * n) * float(iters) + 19.0 * n * float(nfev)


def gram_prepass_flops(n, rows, problems):
    return 2.0 * rows * n * float(problems)


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


def _timed(fn, repetitions=3):
    """warm-up + >= 3 timed repetitions -> (median seconds, all seconds)."""
    fn()
    ts = []
    for _ in range(repetitions):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)), ts


def cpu_threads(omp_default, env=None, affinity=None):
    """Threads of the CPU legs.  torch.distributed.run exports OMP_NUM_THREADS=1 to its ranks unless the caller set it,
    and the CPU legs run on rank 0 while every other rank is parked: under a launcher the legs take the cores this
    process may run on (its affinity mask), not the launcher's per-rank default.  MI355_BENCH_CPU_THREADS overrides."""
    env = os.environ if env is None else env
    if env.get("MI355_BENCH_CPU_THREADS"):
        return max(1, int(env["MI355_BENCH_CPU_THREADS"]))
    if "TORCHELASTIC_RUN_ID" in env or "LOCAL_RANK" in env:
        if affinity is None:
            affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        return max(int(omp_default), int(affinity))
    return int(omp_default)


def cpu_legs(x0_host, n, m, budget_s=None, objective="rosenbrock", params=None, per_problem=None, box=None,
             linesearch="more_thuente", stop=None, solver="lbfgs", second_mode=False):
    """The CPU path beside the GPU number (rank 0, N = 1, a bounded prefix of the same batch):
      parity     one run of the STRICT oracle build (sequential order, no contraction — bit-identical to the
                 reference binary, tests/test_oracle.py) whose results the GPU's are compared with
      port       the same source built here and now `g++ -O3 -march=native -fopenmp`, OpenMP schedule(dynamic)
                 on all host threads: warm-up + 3 timed repetitions, median
      reference  the reference's headers over the Eigen shim (oracle/_ref/libref_o3.so), a thread pool pulling
                 chunks of 32 problems, same protocol
    """
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    stop = stop or oracle_lib.parity_stop()
    cores = cpu_threads(oracle_lib.lib().oracle_num_threads())
    if budget_s is None:   # seconds of CPU time per leg and repetition (the default keeps the whole run within minutes)
        budget_s = float(os.environ.get("MI355_BENCH_CPU_BUDGET_S", "4.0"))

    def run(count, library=None):
        pp = per_problem[:count] if per_problem is not None else None
        if box is not None:
            return oracle_lib.lbfgsb_minimize_batch(objective, x0_host[:count], m=m, nthreads=cores, stop=stop,
                                                    lower=np.full(n, box[0]), upper=np.full(n, box[1]),
                                                    std_sort_order=True, library=library)
        if solver == "bfgs":
            return oracle_lib.bfgs_minimize_batch(objective, x0_host[:count], stop=stop, params=params, nthreads=cores,
                                                  per_problem=pp, linesearch=linesearch, library=library)
        return oracle_lib.minimize_batch(objective, x0_host[:count], m=m, stop=stop, nthreads=cores,
                                         params=params, per_problem=pp, linesearch=linesearch, library=library,
                                         second_mode=second_mode)

    probe = min(x0_host.shape[0], 64 * cores)
    t0 = time.perf_counter()
    run(probe)
    rate = probe / (time.perf_counter() - t0)
    sample = int(min(x0_host.shape[0], max(probe, rate * budget_s)))
    xs, fs, _, ps = run(sample)                                      # parity leg (strict build)
    model_counts = oracle_lib.lbfgsb_last_model_counts() if box is not None else None
    try:
        native = oracle_lib.native_lib()
        build = "g++ -O3 -march=native -fopenmp (built on this box)"
    except Exception as e:                                           # no compiler on the box: time the strict build
        native, build = None, "prebuilt -O2 -ffp-contract=off (native build failed: %s)" % type(e).__name__
    med, ts = _timed(lambda: run(sample, native))
    port = dict(value=sample / med, unit="solves/s", cores=cores, kind="port",
                sample="first %d problems of the same batch, oracle/lbfgs_oracle.hpp%s (sequential order), OpenMP "
                       "schedule(dynamic) on %d threads; warm-up + 3 timed repetitions, median %.2f s" % (
                           sample, " oracle::Bfgs" if solver == "bfgs" else " Second mode" if second_mode else
                           " with the Hager-Zhang search" if linesearch == "hager_zhang" else "", cores, med),
                repetitions_s=[round(t, 4) for t in ts], per_core=sample / med / cores, build=build)
    reference = None
    try:
        import ref_lib
        ridge = objective == "squared_error_ridge" and m == 10 and per_problem is not None
        L = ref_lib.fast_lib() if (objective in ("rosenbrock", "diag_quadratic") or ridge) else None
        if L is not None:
            rsample = max(cores * 32, sample // 2)

            def run_ref():
                if ridge:   # the README functors composed by the reference's expression templates, its own Lbfgs<F, 10>
                    rows = int(params[0])
                    return ref_lib.ridge_minimize_batch_threaded(np.asarray(params[2:]).reshape(rows, n), float(params[1]),
                                                                 per_problem[:rsample], x0_host[:rsample], stop=stop,
                                                                 threads=cores, library=L, second_mode=second_mode)
                return ref_lib.minimize_batch_threaded(
                    objective, x0_host[:rsample], m=m, stop=stop, threads=cores, library=L, params=params,
                    lower=np.full(n, box[0]) if box else None, upper=np.full(n, box[1]) if box else None,
                    solver=solver, linesearch=linesearch)
            rmed, rts = _timed(run_ref)
            reference = dict(value=rsample / rmed, unit="solves/s", cores=cores, kind="reference-over-shim",
                             sample="first %d problems, the reference's own solver/%s.h + %s.h%s compiled "
                                    "over oracle/eigen_shim (oracle/_ref/libref_o3.so), %d "
                                    "threads pulling chunks of 32; warm-up + 3 timed repetitions, median %.2f s" % (
                                        rsample, "bfgs" if solver == "bfgs" else "lbfgsb" if box else "lbfgs", linesearch,
                                        (" on the README ridge functors (function_expressions.h sums)" +
                                         (" declared Second mode" if second_mode else "")) if ridge else "",
                                        cores, rmed),
                             build="g++ -O3 -march=x86-64-v3 (prebuilt where the reference tree is: it does not travel to "
                                   "the GPU box), compiler-default contraction; NOT the port's flags (-march=native on "
                                   "the box): the like-with-like pair is this value and port_at_the_same_flags below",
                             repetitions_s=[round(t, 4) for t in rts], per_core=rsample / rmed / cores)
            try:   # the port once more AT THE REFERENCE LEG'S FLAGS, same sample: the like-with-like pair of CPU figures
                v3 = oracle_lib.v3_lib()
                vmed, vts = _timed(lambda: run(rsample, v3))
                reference["port_at_the_same_flags"] = dict(
                    value=rsample / vmed, unit="solves/s", cores=cores, build="g++ -O3 -march=x86-64-v3 -fopenmp (built on this box)",
                    sample="first %d problems (the reference leg's sample), median %.2f s" % (rsample, vmed),
                    repetitions_s=[round(t, 4) for t in vts], reference_over_port=(rsample / rmed) / (rsample / vmed))
            except Exception as e:
                reference["port_at_the_same_flags"] = dict(value=None, error="%s: %s" % (type(e).__name__, e))
    except Exception as e:  # the checker library is optional on the box
        reference = dict(value=None, error="%s: %s" % (type(e).__name__, e))
    if model_counts is not None:
        port["lbfgsb_model_counts"] = model_counts
    return port, reference, (xs, fs, ps, sample)


# ----------------------------------------------------------------------------------------------------
# live counter passes: bench.py re-runs itself (one launch, no CPU legs) under rocprofv3 --pmc, one
# counter group per pass, and reads the solve kernel's rows
# ----------------------------------------------------------------------------------------------------
PMC_PASSES = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "sq": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU",
           "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
    # fp64 instructions the wavefronts issued, by class (wave-level counts: one per instruction whatever the EXEC mask)
    "flops": ["SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_TRANS_F64",
              "SQ_INSTS_VALU_MFMA_MOPS_F64"],
}
SOLVE_KERNELS = ("_solve_kernel", "lbfgsb_fast_kernel", "lbfgs_wide_kernel")
PREPASS_KERNELS = ("ridge_gram_prepass_kernel", "ridge_gram_matrix_kernel")


def pmc_pass(name, child_args, timeout_s=240, kernels=SOLVE_KERNELS, script=None, calls=None):
    """One rocprofv3 --pmc pass over `python bench.py <child_args>` (or another `script` of this repo); returns {counter:
    mean over the dispatches of the kernels whose name contains one of `kernels`} or raises.  `calls` = N: a call of the
    child's hot path is SEVERAL dispatches (the lock-step augmented-Lagrangian loop): the sum over all matching dispatches
    divided by the child's N calls instead."""
    import csv
    import glob
    import shutil
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        raise RuntimeError("rocprofv3 not found")
    out = tempfile.mkdtemp(prefix="bench_pmc_%s_" % name, dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    # the child is a plain ONE-process run: no rank environment, no launcher bookkeeping, no dry-run request
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "ROLE_WORLD_SIZE",
              "LOCAL_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
              "TORCHELASTIC_MAX_RESTARTS", "TORCHELASTIC_USE_AGENT_STORE", "OMP_NUM_THREADS", DRY_RUN_ENV,
              "MI355_BENCH_SELF_LAUNCHED", "MI355_BENCH_STRONG_ROW"):
        env.pop(k, None)
    cmd = [rocprof, "--pmc"] + PMC_PASSES[name] + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p",
                                                   "--", sys.executable, script or os.path.join(ROOT, "bench.py")] + child_args
    # its own session: on a timeout the WHOLE tree goes (rocprofv3 and the python child under it — a surviving child would
    # keep its context on the GPU under the timed ranks)
    proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                            start_new_session=True)
    try:
        rc = proc.wait(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        proc.wait()
        shutil.rmtree(out, ignore_errors=True)
        raise
    if rc != 0:
        shutil.rmtree(out, ignore_errors=True)
        raise subprocess.CalledProcessError(rc, cmd[:4])
    acc = {}
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if any(t in k for t in kernels):
                acc.setdefault(r["Counter_Name"], {}).setdefault(k, []).append(float(r["Counter_Value"]))
    shutil.rmtree(out, ignore_errors=True)
    if not acc:
        raise RuntimeError("no solve-kernel rows in the counter output")
    if calls:
        return {c: float(sum(np.sum(v) for v in per_kernel.values())) / float(calls) for c, per_kernel in acc.items()}
    # per launch: the mean over a kernel's dispatches, summed over the kernels of one launch (solve + pre-pass)
    return {c: float(sum(np.mean(v) for v in per_kernel.values())) for c, per_kernel in acc.items()}


def live_counters(child_args, script=None, calls=None, kernels=SOLVE_KERNELS):
    """HBM bytes per launch (FETCH_SIZE / WRITE_SIZE, separate passes, corrected as MI355X_MICROARCH.md prescribes:
    both in KiB-units of 64-B fabric requests; FETCH_SIZE doubled for wide coalesced reads on gfx950) and the SQ
    counters behind the VALU-busy fraction."""
    res = {}
    hung = []      # a pass that ran into its time limit: the remaining passes are skipped (a stuck profiler must not cost
                   # the run 4 x the limit — the line then carries the *_error keys instead of the counters)

    def guarded_pass(name, *a, **kw):
        if hung:
            raise RuntimeError("skipped: the %s pass exceeded its time limit" % hung[0])
        try:
            return pmc_pass(name, *a, **kw)
        except subprocess.TimeoutExpired:
            hung.append(name)
            raise
    try:
        fetch = guarded_pass("fetch", child_args, script=script, calls=calls, kernels=kernels)["FETCH_SIZE"]
        write = guarded_pass("write", child_args, script=script, calls=calls, kernels=kernels)["WRITE_SIZE"]
        res["traffic"] = 2.0 * fetch * 1024.0 + write * 1024.0
        res["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE child passes of "
                                 "bench.py (one launch each); read bytes = 2 x FETCH_SIZE KiB (gfx950 wide-read "
                                 "correction, MI355X_MICROARCH.md), write bytes = WRITE_SIZE KiB")
        res["fetch_bytes"], res["write_bytes"] = 2.0 * fetch * 1024.0, write * 1024.0
    except Exception as e:
        res["traffic_error"] = "%s: %s" % (type(e).__name__, e)
    try:
        sq = guarded_pass("sq", child_args, script=script, calls=calls, kernels=kernels)
        res["sq"] = sq
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        res["valu_busy"] = sq["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / (sq["GRBM_GUI_ACTIVE"] / 8.0)
    except Exception as e:
        res["sq_error"] = "%s: %s" % (type(e).__name__, e)
    try:
        fl = guarded_pass("flops", child_args, kernels=kernels + PREPASS_KERNELS, script=script, calls=calls)
        res["flop_insts"] = fl
        # lane-flops ISSUED: an fp64 VALU instruction occupies its SIMD for all 64 lanes whatever the EXEC mask (padding
        # lanes, divergent line searches and idle segments of a tail wavefront are counted — they cost issue slots);
        # FMA = 2 flops per lane; one MFMA "MOPS" unit = 512 flops (rocprofiler-sdk counter_defs.yaml)
        res["executed_flops"] = 64.0 * (2.0 * fl.get("SQ_INSTS_VALU_FMA_F64", 0.0) + fl.get("SQ_INSTS_VALU_ADD_F64", 0.0) +
                                        fl.get("SQ_INSTS_VALU_MUL_F64", 0.0) + fl.get("SQ_INSTS_VALU_TRANS_F64", 0.0)) + \
            512.0 * fl.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
    except Exception as e:
        res["flops_error"] = "%s: %s" % (type(e).__name__, e)
    return res


# ----------------------------------------------------------------------------------------------------
# launch plan: `python bench.py --gpus N` must measure N GPUs or fail — never print an n_gpus: 1 line for N > 1
# ----------------------------------------------------------------------------------------------------
RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE")


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


DRY_RUN_ENV = "MI355_BENCH_SHARED_DEVICE_DRY_RUN"


def dry_run_ranks(env):
    """MI355_BENCH_SHARED_DEVICE_DRY_RUN=N: N ranks that all use device 0 (0 = not a dry run)."""
    try:
        return max(0, int(env.get(DRY_RUN_ENV, "0") or "0"))
    except ValueError:
        return 0


def launch_plan(gpus, launcher, env, argv, visible_gpus, port=None):
    """Decide how this invocation gets its `gpus` ranks (pure function of its arguments; tests/test_bench_launch.py).

      under a launcher   RANK / WORLD_SIZE are in the environment (torch.distributed.run started us, as the driver does
                         for N > 1): run as that rank; WORLD_SIZE must equal --gpus
      self-launch        --gpus N > 1 (or --launcher torchrun) without a rank environment: re-exec under
                         `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`,
                         one rank per GPU, RCCL process group
      in-process         --gpus 1 without a rank environment: the one-GPU line, no process group
    `error` is set (and the process must exit non-zero) when fewer than `gpus` devices are visible or the rank
    environment disagrees with --gpus."""
    plan = {"gpus": gpus, "visible_gpus": visible_gpus, "error": None, "cmd": None, "dry_run": False}
    in_env = [k for k in RANK_ENV if k in env]
    if gpus < 1:
        plan.update(mode="error", error="--gpus must be >= 1")
        return plan
    # ---- the shared-device DRY RUN (round-5 verdict, "Next" 2): MI355_BENCH_SHARED_DEVICE_DRY_RUN=N starts N ranks that
    # ALL use device 0, with gloo as the collective backend (RCCL refuses two ranks on one device) — the real GpuRuntime,
    # real kernels, the rocprofv3 child passes on rank 0 with the other ranks parked, the strong configs[2] row, the line
    # assembly — so that a hang, a port clash, an HSA queue limit or a rocprofv3-under-torchrun problem shows on a one-GPU
    # box.  It is a one-GPU run: the line says "dry_run": true and "n_gpus": 1, and --gpus N > 1 is refused.
    ranks = dry_run_ranks(env)
    if ranks > 0:
        plan["dry_run"] = True
        if gpus != 1:
            plan.update(mode="error", error="%s=%d with --gpus %d: a shared-device dry run measures ONE GPU and is never a "
                                            "--gpus N line" % (DRY_RUN_ENV, ranks, gpus))
            return plan
        if visible_gpus is not None and visible_gpus < 1:
            plan.update(mode="error", error="%s needs one visible GPU" % DRY_RUN_ENV)
            return plan
        if in_env:
            world = int(env.get("WORLD_SIZE", "1"))
            plan.update(mode="rank-of-launcher", world=world, rank=int(env.get("RANK", "0")), local_rank=0)
            if world != ranks:
                plan["error"] = "%s=%d but the launcher's WORLD_SIZE is %d" % (DRY_RUN_ENV, ranks, world)
            return plan
        child = [a for a in argv if a != "--launch-plan"]
        plan.update(mode="self-launch", world=ranks,
                    rank_plan=[{"rank": r, "local_rank": r, "device": "cuda:0"} for r in range(ranks)],
                    cmd=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                         "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
                         os.path.join(ROOT, "bench.py")] + child)
        return plan
    if in_env:
        world = int(env.get("WORLD_SIZE", "1"))
        plan.update(mode="rank-of-launcher", world=world, rank=int(env.get("RANK", "0")),
                    local_rank=int(env.get("LOCAL_RANK", "0")))
        if world != gpus:
            plan["error"] = "--gpus %d but the launcher's WORLD_SIZE is %d" % (gpus, world)
        elif visible_gpus is not None and plan["local_rank"] >= visible_gpus:
            plan["error"] = "rank with LOCAL_RANK=%d but only %d GPU(s) visible" % (plan["local_rank"], visible_gpus)
        return plan
    if launcher == "none" and gpus > 1:
        plan.update(mode="error", error="--launcher none with --gpus %d and no RANK / WORLD_SIZE in the environment: "
                                        "refusing to print a one-GPU line labelled otherwise" % gpus)
        return plan
    if gpus > 1 or launcher == "torchrun":
        child = [a for a in argv if a != "--launch-plan"]
        plan.update(mode="self-launch", world=gpus,
                    rank_plan=[{"rank": r, "local_rank": r, "device": "cuda:%d" % r} for r in range(gpus)],
                    cmd=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
                         "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()),
                         os.path.join(ROOT, "bench.py")] + child)
    else:
        plan.update(mode="in-process", world=1, rank=0, local_rank=0)
    if visible_gpus is not None and visible_gpus < gpus:
        plan["error"] = "--gpus %d but only %d GPU(s) visible on this node" % (gpus, visible_gpus)
    return plan


def _visible_gpus():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


# ----------------------------------------------------------------------------------------------------
# the runtime: everything of a bench run that touches the GPU or the process group, behind one object.
# bench.py only ever constructs GpuRuntime (no CPU fallback: start() exits without an MI355X);
# tests/test_bench_line_gloo.py drives the SAME run_bench() at world size 2 over gloo on a GPU-less box with a
# stand-in it builds itself from the CPU oracle, so that the N > 1 line assembly (who solved, the rank-0-only
# counter / CPU legs with the other ranks parked, the strong-scaled north-star row) is exercised before any
# multi-GPU node is.
# ----------------------------------------------------------------------------------------------------
class GpuRuntime:
    collective_backend = "nccl"           # RCCL on ROCm
    default_cpu_budget_s = 4.0

    def __init__(self, plan):
        self.plan = plan
        self.world, self.rank, self.local_rank = plan["world"], plan["rank"], plan["local_rank"]
        self.use_dist = plan["mode"] == "rank-of-launcher"  # under torch.distributed.run even at world size 1
        self.host_group = None
        self.dry_run = bool(plan.get("dry_run"))
        if self.dry_run:
            # every rank on device 0 (plan["local_rank"] is 0 for all of them); RCCL refuses duplicate devices, so the
            # collectives of the run go over gloo — everything else (kernels, counter passes, parking, line) is the real thing
            self.collective_backend = "gloo"

    def start(self):
        import torch
        import torch.distributed as dist
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        if self.use_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if self.dry_run:
                import datetime
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world,
                                        timeout=datetime.timedelta(minutes=30))
                self.device = torch.device("cpu")     # (bench.py's own small collective tensors; the data stays on cuda:0)
            else:
                dist.init_process_group(self.collective_backend, rank=self.rank, world_size=self.world,
                                        device_id=self.device)
            self._host_group()

    def _host_group(self):
        # a second, host-side (gloo) group: ranks != 0 park on it while rank 0 runs its counter passes and CPU legs
        # (minutes) — no RCCL kernel spins on the idle GPUs meanwhile and no collective watchdog is involved
        import datetime
        import torch.distributed as dist
        if self.world > 1:
            self.host_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(minutes=60))

    def sync(self):
        import torch
        torch.cuda.synchronize()

    def barrier(self):
        import torch.distributed as dist
        self.sync()
        if self.use_dist:
            dist.barrier()
        self.sync()

    def host_barrier(self):
        import torch.distributed as dist
        if self.host_group is not None:
            dist.barrier(group=self.host_group)

    def lbfgs(self, like=None, **kw):
        import cppnumericalsolvers_amd as amd
        if like is not None:
            return amd.BatchedLbfgs(context=like.ctx, **kw)
        return amd.BatchedLbfgs(device=self.local_rank, **kw)

    def lbfgsb(self, **kw):
        import cppnumericalsolvers_amd as amd
        return amd.BatchedLbfgsb(device=self.local_rank, **kw)

    def device_identity(self):
        import torch
        props = torch.cuda.get_device_properties(self.local_rank)
        return "%s/%s" % (os.uname().nodename, getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or
                          "cuda:%d" % self.local_rank)

    def live_counters(self, child_args):
        return live_counters(child_args)

    def finish(self):
        import torch.distributed as dist
        if self.use_dist:
            dist.destroy_process_group()


def parse_args(argv=None):
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
    ap.add_argument("--arithmetic", default="default", choices=["default", "exact", "fma"],
                    help="mi355_arithmetic of the L-BFGS kernels: default = the production (fused) build where it "
                         "exists, exact = the bit-pinning build")
    ap.add_argument("--stop", default="parity", choices=["parity", "variant_a", "default"],
                    help="parity stopping (B) of SURVEY section 7, its variant A (x_delta 1e-9), or the reference's "
                         "default preset")
    ap.add_argument("--ridge-gram", action="store_true",
                    help="cfg4 (the default since round 3): the normal-equation form (objective id 5): Gram matrix + "
                         "c_b = A^T y_b on the matrix cores once per problem, then n^2 multiply-adds per evaluation in "
                         "the ordinary Lbfgs kernel")
    ap.add_argument("--ridge-mfma", action="store_true",
                    help="cfg4: the round-1/2 kernel that evaluates r = A x - y_b on the matrix cores at every evaluation "
                         "(objective id 4)")
    ap.add_argument("--ridge-valu", action="store_true",
                    help="cfg4: the exact-order VALU ridge kernel (objective id 2) instead of the matrix-core one")
    ap.add_argument("--linesearch", default="more_thuente", choices=["more_thuente", "hager_zhang"],
                    help="LineSearch template argument of Lbfgs (the BASELINE configs use the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-counters", action="store_true", help="skip the rocprofv3 --pmc child passes")
    ap.add_argument("--launcher", default="auto", choices=["auto", "torchrun", "none"],
                    help="auto: --gpus N > 1 without RANK / WORLD_SIZE in the environment re-executes itself under "
                         "torch.distributed.run with N ranks; torchrun: do that at N = 1 as well; none: never")
    ap.add_argument("--launch-plan", action="store_true",
                    help="print the launch decision (mode, command, rank plan, visible GPUs) as one JSON line and exit: "
                         "0 when the plan can run here, 3 when it cannot")
    return ap.parse_args(argv)


def main():
    args = parse_args()
    plan = launch_plan(args.gpus, args.launcher, os.environ, sys.argv[1:], _visible_gpus())
    if args.launch_plan:
        print(json.dumps(plan))
        raise SystemExit(3 if plan["error"] else 0)
    if plan["error"]:
        raise SystemExit("bench.py: " + plan["error"])
    if plan["mode"] == "self-launch":
        sys.stdout.flush()
        os.environ["MI355_BENCH_SELF_LAUNCHED"] = "1"
        os.execv(sys.executable, plan["cmd"])
    line = run_bench(args, plan, GpuRuntime(plan))
    if line is not None:
        print(json.dumps(line))


def run_bench(args, plan, rt):
    """One bench run on the runtime `rt` (GpuRuntime here; a test's stand-in in tests/test_bench_line_gloo.py).
    Every rank calls this; rank 0 gets the line (a dict), the others None."""
    args.ridge_gram = not (args.ridge_mfma or args.ridge_valu)   # cfg4: the fastest form that meets the parity bar

    import torch
    import torch.distributed as dist
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import sharded

    world, rank, local_rank = plan["world"], plan["rank"], plan["local_rank"]
    rt.start()
    use_dist = rt.use_dist

    wl = dict(WORKLOADS[args.workload])
    if args.batch:
        wl["B"] = args.batch
    if wl.get("linesearch"):
        args.linesearch = wl["linesearch"]
    dense_bfgs = wl.get("solver") == "bfgs"
    strong = bool(wl.get("strong"))
    Bg, n, m = wl["B"], wl["n"], wl["m"]
    stop_desc = {"parity": "parity stopping (B): x_delta=1e-11, gradient_norm=1e-8 %s, past=0, 10000 iterations",
                 "variant_a": "parity stopping, variant A: x_delta=1e-9, gradient_norm=1e-8 %s, past=0, 10000 iterations",
                 "default": "the reference's default preset (progress.h:353-431: x_delta=1e-9, gradient_norm=1e-5 %s, "
                            "plateau test past=3 / 1e-6); results then compare with another summation order only to "
                            "~1e-3 (SURVEY section 7), the twin stays exact"}[args.stop]

    def engine_stop():
        if args.workload == "cfg5":
            return lbfgsb_tight_stop(amd.capi.default_stop("lbfgsb"))
        s = amd.parity_stop() if args.stop != "default" else amd.capi.default_stop()
        if args.stop == "variant_a":
            s.x_delta = 1e-9
        return s

    if args.workload == "cfg5":
        solver = rt.lbfgsb(m=m, stopping_progress=engine_stop(), arithmetic=args.arithmetic)
        solver.SetBounds(np.full(n, wl["lower"]), np.full(n, wl["upper"]))
    elif dense_bfgs:
        solver = amd.BatchedBfgs(stopping_progress=engine_stop(), device=rt.local_rank, linesearch=args.linesearch)
    else:
        solver = rt.lbfgs(m=m, stopping_progress=engine_stop(),
                          lanes_per_problem=args.lanes, elems_per_lane=args.elems,
                          history_placement=args.history, linesearch=args.linesearch,
                          arithmetic=args.arithmetic)
    B_global = Bg if strong else Bg * world
    lo, hi = sharded.shard_range(B_global, rank, world)
    rows = wl.get("rows", 0)
    per_problem = None
    ridge_host = None
    own_host = None
    if wl.get("own"):
        # one matrix per problem, generated on the device (17 GB at 262,144 problems: not through the host); a bounded
        # prefix goes back to the host for the CPU legs
        gen = torch.Generator(device=solver.device)
        gen.manual_seed(SEED + lo)
        per_problem = torch.randn(hi - lo, rows * n + rows, dtype=torch.float64, device=solver.device, generator=gen)
        per_problem[:, :rows * n] *= 1.0 / np.sqrt(float(rows))
        obj = amd.SquaredErrorRidgePerProblem(rows, wl["lam"])
        x0 = torch.zeros(hi - lo, n, dtype=torch.float64, device=solver.device)
        own_host = per_problem[:min(hi - lo, 16384)].cpu().numpy()
    elif args.workload in ("cfg4", "cfg4big", "f_second"):
        A_host, Y_host = amd.synthetic_ridge_host(hi - lo, rows, n, SEED, first_problem=lo)
        ridge_host = (A_host, Y_host)
        obj = amd.SquaredErrorRidge(A_host, wl["lam"], matrix_cores=not (args.ridge_valu or args.ridge_gram),
                                    gram=args.ridge_gram, differentiability="second" if wl.get("second") else "first")
        per_problem = torch.from_numpy(Y_host).to(solver.device)     # resident in HBM
        x0 = torch.zeros(hi - lo, n, dtype=torch.float64, device=solver.device)
    elif wl.get("objective") == "diag_quadratic":
        dq_params = np.concatenate([0.5 + 19.5 * np.random.default_rng(SEED).uniform(size=n), [1.5]])
        obj = amd.DiagQuadratic(dq_params[:n], dq_params[n])
        x0 = solver.fill_x0(hi - lo, n, wl.get("x0", args.x0), SEED, first_problem=lo)
    else:
        obj = amd.Rosenbrock()
        x0 = solver.fill_x0(hi - lo, n, wl.get("x0", args.x0), SEED, first_problem=lo)  # resident in HBM
    rt.sync()

    def step():
        x, f, g, prog = solver.minimize(obj, x0, per_problem=per_problem)
        status, iters, nfev, sum_k = sharded.progress_fields_device(prog)
        flag = sharded.allreduce_flag(sharded.local_counts(status, iters))
        return (x, f, g, prog), flag

    barrier = rt.barrier

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
        t = torch.tensor([elapsed], dtype=torch.float64, device=rt.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    x, f, g, prog = out
    pn = amd.progress_to_numpy(prog)
    # ---- who actually solved: one record per rank (device identity, problems solved, kernel ms), gathered over RCCL ----
    mine = {"rank": rank, "device": rt.device_identity(), "solved": int(len(pn)), "kernel_ms": float(np.mean(kernel_ms))}
    if use_dist:
        ones = torch.ones(1, dtype=torch.int64, device=rt.device)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)                     # the all-reduced rank count (RCCL)
        rccl_ranks = int(ones.item())
        records = [None] * world
        dist.all_gather_object(records, mine)
    else:
        rccl_ranks, records = 1, [mine]
    solving = [r for r in records if r["solved"] > 0]
    n_gpus_measured = len({r["device"] for r in solving})
    dry_run = bool(plan.get("dry_run"))
    if dry_run:
        # N ranks, ONE device, gloo: the line is a one-GPU dry run and says so; every rank must have solved and reduced
        if n_gpus_measured != 1 or rccl_ranks != world or len(solving) != world or int(flag.total) != B_global:
            raise SystemExit("bench.py: shared-device dry run with %d rank(s): %d device(s) solved, %d rank(s) solved, %d "
                             "in the all-reduce, %d of %d problems in the global record"
                             % (world, n_gpus_measured, len(solving), rccl_ranks, int(flag.total), B_global))
    elif n_gpus_measured != args.gpus or rccl_ranks != args.gpus or int(flag.total) != B_global:
        raise SystemExit("bench.py: --gpus %d but %d distinct device(s) solved, %d rank(s) in the all-reduce, %d of %d "
                         "problems in the global record: refusing to print a mislabelled line"
                         % (args.gpus, n_gpus_measured, rccl_ranks, int(flag.total), B_global))
    iters_sum, sumk_sum, nfev_sum = int(pn["num_iterations"].sum()), int(pn["sum_k"].sum()), int(pn["nfev"].sum())
    bytes_launch = algorithmic_bytes(n, iters_sum, sumk_sum, rows, (hi - lo) if wl.get("own") else 0)
    ridge_form = None if not rows else ("gram" if (args.ridge_gram or wl.get("own")) else "direct" if args.ridge_valu else "mfma")
    flops_launch = algorithmic_flops(n, iters_sum, sumk_sum, nfev_sum, rows, ridge_form)
    if ridge_form == "gram":
        flops_launch += gram_prepass_flops(n, rows, hi - lo)
    if dense_bfgs:
        bytes_launch = bfgs_algorithmic_bytes(n, iters_sum)
        flops_launch = bfgs_algorithmic_flops(n, iters_sum, nfev_sum)
    k_ms = float(np.mean(kernel_ms))
    achieved = bytes_launch / (k_ms * 1e-3) / 1e9
    value = B_global * args.steps / elapsed
    launch = solver.last_launch()
    arith = solver.last_arithmetic()
    kernel_name = ((("lbfgsb_fast_kernel<%d,Rosenbrock,5>" if arith == "fma" else "lbfgsb_solve_kernel<%d,Rosenbrock,5>")
                    % launch["elems_per_lane"]) if args.workload == "cfg5" else
                   "lbfgs_wide_kernel<DiagQuadratic>" if args.workload == "wide" else
                   "lbfgs_solve_kernel<%d,%d,Rosenbrock,0,MoreThuente,kAlgBfgs,ArithExact>" % (
                       launch["lanes_per_problem"], launch["elems_per_lane"]) if dense_bfgs else
                   "ridge_mfma_solve_kernel<10>" if (rows and not (args.ridge_valu or args.ridge_gram)) else
                   "lbfgs_solve_kernel<%d,%d,%s,%d,%s>" % (launch["lanes_per_problem"], launch["elems_per_lane"],
                                                           ("RidgeGram" if args.ridge_gram else "SquaredErrorRidge") if rows
                                                           else "Rosenbrock",
                                                           launch["y_columns_in_registers"],
                                                           "ArithFma" if arith == "fma" else "ArithExact"))

    result = {
        "metric": "L-BFGS solves/sec (batched Rosenbrock-N)",
        "value": value,
        "unit": "solves/s",
        "n_gpus": n_gpus_measured,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": wl["desc"] + ("; Hager-Zhang line search" if args.linesearch == "hager_zhang" else "") +
                        "; x0 '%s' seed %d; " % ("zero" if rows else wl.get("x0", args.x0), SEED) +
                        stop_desc % ("absolute on the projected gradient" if args.workload == "cfg5" else "relative"),
            "problems_per_gpu": hi - lo, "problems_total": B_global, "n": n, "m": m,
            "parallelism": "batch-sharded x%d" % world,
            "arithmetic": arith + (" (relaxed algebra of the compact representation + fused multiply-adds; bit-identical "
                                   "to its CPU twin oracle/lbfgsb_fast_oracle.hpp, within 1e-6 of the reference binary)"
                                   if (arith == "fma" and args.workload == "cfg5") else
                                   " (fused multiply-adds; bit-identical to the oracle's butterfly_fma twin, within 1e-6 "
                                   "of the reference-order solve)" if arith == "fma" else
                                   " (no FMA; bit-identical to the oracle's butterfly twin)"),
            "lanes_per_problem": launch["lanes_per_problem"], "elems_per_lane": launch["elems_per_lane"],
            "grid_workgroups": launch["blocks"], "threads_per_workgroup": launch["threads"],
            "lds_bytes_per_workgroup": launch["lds_bytes"],
            "y_columns_in_registers": launch["y_columns_in_registers"],
            "mean_iterations": iters_sum / float(len(pn)), "mean_nfev": nfev_sum / float(len(pn)),
            "max_iterations": int(pn["num_iterations"].max()),
            "all_converged": bool(flag.all_converged), "unconverged": int(flag.unconverged),
        },
        "roofline": {
            "bound": "hbm-state-streaming-model",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "frac_of_achievable_6p3TBs": achieved / HBM_ACHIEVABLE_GBS,
            "traffic": None,
            "kernel": kernel_name,
            "kernel_ms": k_ms,
            "algorithmic_bytes_per_launch": bytes_launch,
            "model": "STATE-STREAMING MODEL, NOT A BANDWIDTH: algorithmic bytes = sum_b 8n(6T_b + 2 sum_k_b) (SURVEY 8d) "
                     "over the kernel time; the fused kernel keeps that state in registers/LDS, so frac may exceed 1. "
                     "The physical HBM figure is hbm_frac_measured (traffic / kernel time / 8 TB/s); what bounds the "
                     "kernel is VALU issue: see roofline_valu",
        },
        "roofline_valu": {
            "bound": "valu-fp64",
            "achieved": flops_launch / (k_ms * 1e-3) / 1e12,
            "peak": FP64_VALU_PEAK_TF if arith == "fma" else FP64_VALU_PEAK_TF / 2.0,
            "unit": "TFLOP/s",
            "frac": flops_launch / (k_ms * 1e-3) / 1e12 / (FP64_VALU_PEAK_TF if arith == "fma" else FP64_VALU_PEAK_TF / 2.0),
            "frac_of_fma_peak": flops_launch / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF,
            "useful_flops_per_launch": flops_launch,
            "executed_flops": None,
            "frac_executed": None,
            "valu_busy": None,
            "note": "useful flops = what the algorithm needs as the kernels execute it: 8 n sum_k (two-loop with cached "
                    "rho; SURVEY 8d's 12 n sum_k prices the reference's recomputation of s_i.y_i) + 22 n T + (4 + c_obj) n "
                    "nfev, c_obj = 15 Rosenbrock / 4 rows ridge direct / 2 n + 4 ridge normal-equation (+ 2 rows n per "
                    "problem for its pre-pass); peak = 78.6 TFLOP/s fp64 VALU with FMA, 39.3 without (the exact build issues "
                    "separate multiplies and adds); executed_flops = lane-flops ISSUED per launch from a counter pass of "
                    "this run: 64 x (2 FMA_F64 + ADD_F64 + MUL_F64 + TRANS_F64) + 512 x MFMA_MOPS_F64 (wave-level "
                    "instruction counts: idle lanes of padded / divergent / tail wavefronts included), frac_executed = that "
                    "over the kernel time over 78.6; valu_busy = SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 "
                    "XCDs)",
        },
    }
    if rows and not args.ridge_gram:
        result["config"]["ridge_form"] = (
            "objective matrix-vector products r = A x - y_b, A^T r on v_mfma_f64_16x16x4_f64 at every evaluation "
            "(ridge_mfma_solve_kernel)" if not args.ridge_valu else
            "reference-order VALU kernel (objective id 2), bit-identical to the README functors under the reference's Lbfgs")
    if rows and args.ridge_gram:
        result["config"]["ridge_form"] = (
            "normal equations: f = x^T G x - 2 c_b^T x + y_b^T y_b, G = A^T A + lambda I once per matrix (matrix cores; cached "
            "by the context across launches), c_b = A^T y_b and y_b^T y_b once per problem by a batched GEMM on "
            "v_mfma_f64_16x16x4_f64 (inside the timed region AND inside kernel_ms: the HIP events bracket pre-pass + solve "
            "kernel), then n^2 multiply-adds per evaluation; algebraically the "
            "reference's objective, x* / f* within 1e-6 of the reference binary; --ridge-mfma times the round-1/2 kernel "
            "that evaluates r = A x - y_b on the matrix cores at every evaluation, --ridge-valu the exact-order VALU kernel")
    if rows and not (args.ridge_valu or args.ridge_gram):
        # the matrix-core kernel is priced against the dense f64 MFMA peak as well: every objective
        # evaluation is 2 * 2 * rows * n flops on v_mfma_f64_16x16x4_f64 (MI355X_MICROARCH.md: 78.6 TFLOP/s)
        flops = float(nfev_sum) * 4.0 * rows * n
        result["roofline_mfma"] = {
            "bound": "mfma", "achieved": flops / (k_ms * 1e-3) / 1e12, "peak": 78.6, "unit": "TFLOP/s",
            "frac": flops / (k_ms * 1e-3) / 1e12 / 78.6, "flops_per_launch": flops,
            "note": "objective matrix-vector products only; the rest of the iteration runs on the VALU "
                    "(DESIGN.md section 3.4 for the phase shares)"}

    # ---- the north-star row under the driver's own multi-GPU launch: configs[2] at its full size, STRONG-scaled ----
    # (every rank takes part: its contiguous shard, the 3-word all-reduce per step, MAX-over-ranks timing; run BEFORE
    # the rank-0-only legs below so that all collectives of the run are back to back)
    # (MI355_BENCH_STRONG_ROW=1 forces the row at world size 1 too: the one-GPU box's test of this code path)
    if use_dist and (world > 1 or os.environ.get("MI355_BENCH_STRONG_ROW") == "1") and \
            (not args.no_secondary or os.environ.get("MI355_BENCH_STRONG_ROW") == "1") and args.workload == "cfg2":
        strong_row = strong_cfg3full(args, amd, solver, torch, dist, sharded, rank, world, rt)
        if rank == 0:
            result["secondary_cfg3full_strong"] = strong_row

    # ---- counters measured in this run (rank 0, on ITS device; at N > 1 the other ranks are parked on the host-side
    # barrier at the end of this function): HBM traffic and VALU-busy of the same per-GPU launch --------
    if rank == 0 and not args.no_counters:
        # the child is a ONE-GPU run of this rank's shard (its rank environment is stripped in pmc_pass)
        child_batch = (hi - lo) if (world > 1 and strong) else args.batch
        child = ["--workload", args.workload, "--batch", str(child_batch), "--steps", "1", "--warmup", "1",
                 "--arithmetic", args.arithmetic, "--stop", args.stop, "--x0", args.x0, "--linesearch", args.linesearch,
                 "--lanes", str(args.lanes), "--elems", str(args.elems), "--history", str(args.history),
                 "--no-cpu-baseline", "--no-secondary", "--no-counters"] + (["--ridge-valu"] if args.ridge_valu else []) + \
                (["--ridge-mfma"] if args.ridge_mfma else [])
        rt.sync()
        lc = rt.live_counters(child)
        if "traffic" in lc:
            result["roofline"]["traffic"] = lc["traffic"]
            result["roofline"]["traffic_source"] = lc["traffic_source"]
            result["roofline"]["traffic_read_bytes"] = lc["fetch_bytes"]
            result["roofline"]["traffic_write_bytes"] = lc["write_bytes"]
            result["roofline"]["hbm_frac_measured"] = lc["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            result["roofline"]["hbm_frac_measured_of_achievable_6p3TBs"] = lc["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_ACHIEVABLE_GBS
        if "valu_busy" in lc:
            result["roofline_valu"]["valu_busy"] = lc["valu_busy"]
            sq = lc["sq"]
            # wave-level VALU instructions charged to one iteration of one problem (a wavefront instruction serves the
            # 64 / lanes_per_problem problems of its wavefront and is counted once)
            result["roofline_valu"]["valu_wave_instructions_per_problem_iteration"] = sq["SQ_INSTS_VALU"] / max(1, iters_sum)
            result["roofline_valu"]["valu_wave_instructions_per_launch"] = sq["SQ_INSTS_VALU"]
            result["roofline_valu"]["salu_wave_instructions_per_launch"] = sq.get("SQ_INSTS_SALU")
            result["roofline_valu"]["wait_inst_any_over_wave_cycles"] = sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"]
        if "executed_flops" in lc:
            rv = result["roofline_valu"]
            rv["executed_flops"] = lc["executed_flops"]
            rv["frac_executed"] = lc["executed_flops"] / (k_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TF
            rv["useful_over_executed"] = flops_launch / lc["executed_flops"]
            rv["fp64_instructions_per_launch"] = lc["flop_insts"]
        for k in ("traffic_error", "sq_error", "flops_error"):
            if k in lc:
                result["roofline"][k] = lc[k]
    if result["roofline"]["traffic"] is None:
        traffic_file = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(traffic_file):
            try:
                tr = json.load(open(traffic_file)).get(args.workload)
                if tr:
                    result["roofline"]["traffic"] = tr["bytes_per_launch"]
                    result["roofline"]["traffic_source"] = "NOT measured in this run; " + tr.get("source", "profiles/")
            except Exception:
                pass

    if rank == 0 and not args.no_cpu_baseline:
        x0h = x0.cpu().numpy()
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        ostop = oracle_lib.parity_stop() if args.stop != "default" else oracle_lib.default_stop()
        if args.stop == "variant_a":
            ostop.x_delta = 1e-9
        if own_host is not None:
            port, reference, (xs, fs, ps, sample) = cpu_legs(
                x0h[:own_host.shape[0]], n, m, objective="squared_error_ridge_own", stop=ostop,
                params=np.array([float(rows), wl["lam"]]), per_problem=own_host)
        elif ridge_host is not None:
            port, reference, (xs, fs, ps, sample) = cpu_legs(
                x0h, n, m, objective="squared_error_ridge", stop=ostop, second_mode=bool(wl.get("second")),
                params=np.concatenate([[float(rows), wl["lam"]], ridge_host[0].ravel()]), per_problem=ridge_host[1])
        elif dense_bfgs:
            port, reference, (xs, fs, ps, sample) = cpu_legs(x0h, n, m, linesearch=args.linesearch, stop=ostop, solver="bfgs")
        elif args.workload == "cfg5":
            port, reference, (xs, fs, ps, sample) = cpu_legs(x0h, n, m, box=(wl["lower"], wl["upper"]),
                                                             stop=lbfgsb_tight_stop(oracle_lib.default_stop()))
            port["sample"] = port["sample"].replace("lbfgs_oracle.hpp", "lbfgsb_oracle.hpp")
            lbfgsb_useful_flops(result, port.pop("lbfgsb_model_counts"), n, iters_sum, nfev_sum, k_ms)
        elif wl.get("objective") == "diag_quadratic":
            port, reference, (xs, fs, ps, sample) = cpu_legs(x0h, n, m, objective="diag_quadratic", params=dq_params,
                                                             stop=ostop)
        else:
            port, reference, (xs, fs, ps, sample) = cpu_legs(x0h, n, m, linesearch=args.linesearch, stop=ostop)
        result["cpu_baseline"] = port
        if reference is not None:
            result["cpu_reference"] = reference
        xh, fh = x.cpu().numpy()[:sample], f.cpu().numpy()[:sample]
        result["config"]["parity_vs_cpu_sample"] = {
            "problems": int(sample), "max_abs_dx": float(np.max(np.abs(xh - xs))),
            "max_abs_df": float(np.max(np.abs(fh - fs))), "tol": 1e-06,
            "against": "the strict oracle build in the reference's (sequential) summation order, bit-identical to the "
                       "reference binary on the CPU (tests/test_oracle.py)" + (
                           "; under the default preset two summation orders only agree to ~1e-3 (a property of the "
                           "reference, SURVEY section 7) — the exact comparison there is against the twin, in tests/"
                           if args.stop == "default" else "")}

    if args.workload == "f_bfgs":
        result["metric"] = "dense-BFGS solves/sec (batched Rosenbrock-N)"
        result["roofline"]["model"] = (
            "STATE-STREAMING MODEL, NOT A BANDWIDTH: 8 (4 n^2 + 6 n) bytes per iteration (the n x n approximation read for "
            "H g, for H y and read + written by the rank-two update); the kernel keeps H in LDS, so the physical HBM figure "
            "is hbm_frac_measured.  roofline_valu.useful flops here = (12 n^2 + 22 n) T + 19 n nfev (bfgs.h:81,123-134)")
    if args.workload == "f_second":
        result["metric"] = "L-BFGS solves/sec (batched SquaredError ridge, Second mode)"
    if args.workload == "cfg4big":
        result["metric"] = "L-BFGS solves/sec (batched SquaredError ridge, A 1000 x 200)"
    if args.workload == "cfg4own":
        result["metric"] = "L-BFGS solves/sec (batched SquaredError ridge, one 128 x 64 matrix per problem)"
        result["config"]["ridge_form"] = (
            "one matrix per problem (objective id 6): per-problem pre-pass on the matrix cores writes G_b = A_b^T A_b + lambda I, "
            "c_b, y_b.y_b (33 KB per problem in HBM), the solve streams its G_b on every evaluation; A_b is read once")
    if args.workload == "wide":
        result["metric"] = "L-BFGS solves/sec (batched DiagQuadratic-N, n > 256)"
        result["roofline"]["model"] = (
            "here the state DOES live in memory (x, g, trial point, direction and the 2m-vector correction ring in an HBM "
            "workspace per resident workgroup), so achieved / peak is a bandwidth fraction of the MODEL bytes 8n(6T + 2 sum_k); "
            "the kernel actually moves ~4x that (every two-loop step reads a history vector and reads + writes the "
            "direction): compare traffic with algorithmic_bytes_per_launch")
    if rank == 0 and world == 1 and not args.no_secondary and args.workload not in ("wide", "cfg4own"):
        # PCIe-inclusive rate through the host-pointer entry point (pinned staging, chunked overlap); informational
        x0h = x0.cpu().numpy()
        pph = ridge_host[1] if ridge_host is not None else None
        solver.minimize_host(obj, x0h, per_problem=pph)   # (the context's staging buffers grow to the batch here)
        dts = []
        for _ in range(3):
            t0 = time.perf_counter()
            solver.minimize_host(obj, x0h, per_problem=pph)
            dts.append(time.perf_counter() - t0)
        dth = float(np.median(dts))
        result["config"]["pcie_inclusive_host_entry"] = {
            "value": x0h.shape[0] / dth, "unit": "solves/s", "ms": dth * 1e3,
            "note": "pageable numpy arrays in, freshly allocated numpy arrays out (first-touch page faults included), "
                    "through mi355_lbfgs_minimize_batch_host: warm staging, median of 3"}

    if rank == 0 and world == 1 and not args.no_secondary and args.workload == "cfg2":
        result["config"].update(secondary_figures(args, amd, solver, torch))

    if dry_run:
        result["dry_run"] = True
        result["dry_run_note"] = ("SHARED-DEVICE DRY RUN (%s=%d): %d ranks on ONE GPU over gloo — a rehearsal of the N > 1 code "
                                  "paths (rank launch, sharding, the 3-word all-reduce per step, rank-0-only counter passes and "
                                  "CPU legs with the other ranks parked, the strong configs[2] row, the line assembly), NOT a "
                                  "multi-GPU measurement: value is what ONE GPU does when %d processes share it"
                                  % (DRY_RUN_ENV, world, world, world))
    result["multi_gpu"] = {
        "ranks_in_this_run": world,
        "collective_backend": rt.collective_backend,
        "rccl_ranks": rccl_ranks,
        "launch": plan["mode"] if "MI355_BENCH_SELF_LAUNCHED" not in os.environ else "self-launch",
        "devices": [r["device"] for r in records],
        "problems_per_rank": [r["solved"] for r in records],
        "kernel_ms_per_rank": [r["kernel_ms"] for r in records],
        "measured": world > 1 and not dry_run,
        "note": "shared-device dry run: see dry_run_note" if dry_run else
                ("this line was measured on %d GPUs (one process per GPU, RCCL process group)" % world) if world > 1 else
                "this line is a ONE-GPU measurement: nothing about G > 1 is measured or claimed here; under "
                "`--gpus N` (N > 1) the line carries secondary_cfg3full_strong = configs[2] sharded over the N ranks"}

    physical_roofline(result)
    lift_north_star(result)
    rt.host_barrier()      # ranks != 0 were parked here while rank 0 ran its counter passes and CPU legs
    rt.finish()
    return result if rank == 0 else None


def lbfgsb_useful_flops(result, counts, n, iters_sum, nfev_sum, k_ms):
    """configs[4]: replace the L-BFGS flop model in `roofline_valu` (meaningless for this kernel) by the operation count of
    the reference's L-BFGS-B algebra as lbfgsb.h writes it — p = W^T d, the 2m x 2m solves of the Cauchy loop, the WZ
    products and the N = I - M^-1 WZ WZ^T / theta system of the subspace step, S^T Y / S^T S / MM.lu() per accepted
    pair (oracle/lbfgsb_oracle.hpp ReferenceStepFlops, file:line per term) — counted step by step by the strict oracle
    on the parity sample and scaled by iterations to the launch, plus (4 + 15) n per objective evaluation."""
    rv = result["roofline_valu"]
    per_step = counts["flops"] / max(1.0, counts["steps"])
    useful = per_step * float(iters_sum) + 19.0 * n * float(nfev_sum)
    rv["useful_flops_per_launch"] = useful
    rv["achieved"] = useful / (k_ms * 1e-3) / 1e12
    rv["frac"] = rv["achieved"] / rv["peak"]
    rv["frac_of_fma_peak"] = rv["achieved"] / FP64_VALU_PEAK_TF
    if rv.get("executed_flops"):
        rv["useful_over_executed"] = useful / rv["executed_flops"]
    rv["lbfgsb_model"] = {
        "flops_per_step_reference_algebra": per_step,
        "breakpoints_per_step": counts["breakpoints"] / max(1.0, counts["steps"]),
        "free_variables_per_step": counts["free_variables"] / max(1.0, counts["steps"]),
        "sample_steps": counts["steps"],
        "note": "useful flops = the REFERENCE's operation count for the same iterates (lbfgsb.h:318-515 as written: three "
                "SolveM per breakpoint, one per column of N, S^T Y and S^T S recomputed per pair); the relaxed kernel "
                "reaches them with fewer operations (one solve per breakpoint by linearity, one elimination for the "
                "subspace step, incremental Gram updates), so useful_over_executed is work avoided plus lane utilisation"}
    rv["note"] = rv["note"].replace("useful flops = what the algorithm needs as the kernels execute it",
                                    "useful flops (Lbfgsb) = see lbfgsb_model; for Lbfgs: what the algorithm needs as "
                                    "the kernels execute it")


def physical_roofline(result):
    """The binding fraction, one hop from `roofline` (the state-streaming `frac` stays: it is the SURVEY 8d contract).
      frac_physical   max(measured HBM bytes / time / 8 TB/s, issued fp64 lane-flops / time / 78.6 TFLOP/s): how close the
                      kernel runs to the nearest physical ceiling of the chip
      bound_physical  which of the two that is ("hbm" / "valu-fp64")
      useful_frac     flops the ALGORITHM needs / time / 78.6 TFLOP/s (padding, redundant segment-uniform scalars and
                      tail-idle lanes not counted): the number a faster kernel would raise"""
    rf, rv = result["roofline"], result["roofline_valu"]
    rf["useful_frac"] = rv.get("frac_of_fma_peak")
    cands = [(v, k) for v, k in ((rf.get("hbm_frac_measured"), "hbm"), (rv.get("frac_executed"), "valu-fp64"))
             if v is not None]
    if cands:
        rf["frac_physical"], rf["bound_physical"] = max(cands)
        rf["physical_source"] = "counter passes of this run (hbm_frac_measured, roofline_valu.frac_executed)"
    else:
        rf["frac_physical"], rf["bound_physical"] = rf["useful_frac"], "valu-fp64"
        rf["physical_source"] = "no counter pass in this run: useful flops only (a lower bound of the issued fraction)"
    rf["valu_busy"] = rv.get("valu_busy")
    # what binds, as the FIRST keys of `roofline` (round-5 verdict, "Next" 5): no reader of the parsed line should take the
    # state-streaming `frac` (which exceeds 1 when the state never leaves the chip) for a bandwidth
    binding = {"bound": rf["bound_physical"], "frac": rf["useful_frac"], "issued": rv.get("frac_executed"),
               "valu_busy": rv.get("valu_busy"), "hbm_frac_measured": rf.get("hbm_frac_measured"),
               "frac_physical": rf["frac_physical"],
               "note": "frac = flops the algorithm needs / kernel time / 78.6 TFLOP/s (fp64 VALU peak with FMA); issued = "
                       "fp64 lane-flops the wavefronts issued / time / peak; valu_busy = share of VALU issue cycles in use; "
                       "hbm_frac_measured = counter-measured HBM bytes / time / 8 TB/s.  `roofline.frac` below is the "
                       "SURVEY 8(d) state-streaming MODEL over the kernel time, not a bandwidth"}
    exceeds = bool(rf["achieved"] > rf["peak"])
    result["roofline"] = dict([("binding", binding), ("model_exceeds_hbm_peak", exceeds)] + list(rf.items()))


NORTH_STAR_TARGET = 1.0e7   # BASELINE.json: >= 1e7 Rosenbrock-64 m=10 solves/s on 8 x MI355X at >= 0.30 of the roofline


def lift_north_star(result):
    """BASELINE.json's target workload (configs[2]: 1,048,576 x Rosenbrock-64, m = 10) as a TOP-LEVEL key next to `value`
    at every N: the strong-scaled row of this run's ranks (N > 1), or the whole batch on the one GPU (N = 1)."""
    row = result.get("secondary_cfg3full_strong") or result["config"].get("secondary_cfg3_full_batch_one_gpu")
    if not row:
        return
    ns = {"workload": row["workload"], "value": row["value"], "unit": row["unit"], "scaling": "strong",
          "n_gpus": row.get("n_gpus", 1), "problems_total": 1048576 if "problems_per_rank" not in row else
          int(sum(row["problems_per_rank"])),
          "target": NORTH_STAR_TARGET, "target_n_gpus": 8, "frac_of_target": row["value"] / NORTH_STAR_TARGET,
          "state_streaming_GBs": row.get("state_streaming_GBs"),
          "source_key": "secondary_cfg3full_strong" if "problems_per_rank" in row else
                        "config.secondary_cfg3_full_batch_one_gpu"}
    if ns["state_streaming_GBs"] is not None:
        ns["state_streaming_frac_of_n_gpus_x_8TBs"] = ns["state_streaming_GBs"] / (ns["n_gpus"] * HBM_PEAK_GBS)
    result["north_star"] = ns


def strong_cfg3full(args, amd, solver, torch, dist, sharded, rank, world, rt, steps=3):
    # (MI355_BENCH_STRONG_BATCH: a smaller total for tests)
    """BASELINE configs[2] / the north-star target row: 1,048,576 x Rosenbrock-64, m = 10, the WHOLE batch sharded over
    the ranks of this run (contiguous ranges, start points from the counter-based generator: no data moves), one
    3-word RCCL all-reduce per step.  Collective calls: every rank must enter."""
    w = WORKLOADS["cfg3full"]
    Bg, n, m = int(os.environ.get("MI355_BENCH_STRONG_BATCH", w["B"])), w["n"], w["m"]
    lo, hi = sharded.shard_range(Bg, rank, world)
    s3 = rt.lbfgs(like=solver, m=m, stopping_progress=amd.parity_stop(), arithmetic=args.arithmetic)
    x0 = s3.fill_x0(hi - lo, n, args.x0, SEED, first_problem=lo)
    rt.sync()

    def step():
        x, f, g, prog = s3.minimize(amd.Rosenbrock(), x0)
        status, iters, nfev, sum_k = sharded.progress_fields_device(prog)
        return prog, sharded.allreduce_flag(sharded.local_counts(status, iters))

    barrier = rt.barrier

    step()
    barrier()
    kms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        prog, flag = step()
        kms.append(s3.last_kernel_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    pn = amd.progress_to_numpy(prog)
    local_bytes = algorithmic_bytes(n, int(pn["num_iterations"].sum()), int(pn["sum_k"].sum()))
    t = torch.tensor([elapsed, float(np.mean(kms))], dtype=torch.float64, device=rt.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([local_bytes, 1.0, float(pn["num_iterations"].sum())], dtype=torch.float64, device=rt.device)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    per_rank = [torch.zeros(2, dtype=torch.float64, device=rt.device) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([float(np.mean(kms)), float(hi - lo)], dtype=torch.float64, device=rt.device))
    elapsed, k_max = float(t[0].item()), float(t[1].item())
    return {
        "workload": w["desc"] + "; parity stopping; strong scaling: total work fixed, %d ranks%s" % (
            world, " (shared-device DRY RUN: all on one GPU)" if getattr(rt, "dry_run", False) else ""),
        "value": Bg * steps / elapsed, "unit": "solves/s", "scaling": "strong",
        "n_gpus": 1 if getattr(rt, "dry_run", False) else world, "ranks": world, "steps": steps,
        "dry_run": bool(getattr(rt, "dry_run", False)),
        "ms_per_step": elapsed / steps * 1e3,
        "kernel_ms_per_rank": [float(v[0].item()) for v in per_rank],
        "problems_per_rank": [int(v[1].item()) for v in per_rank],
        "rccl_ranks": int(round(float(tot[1].item()))),
        "global_record": {"total": int(flag.total), "unconverged": int(flag.unconverged), "iterations": int(flag.iterations)},
        "mean_iterations": float(tot[2].item()) / Bg,
        "state_streaming_GBs": float(tot[0].item()) / (k_max * 1e-3) / 1e9,
        "state_streaming_frac_of_%d_x_8TBs" % (1 if getattr(rt, "dry_run", False) else world):
            float(tot[0].item()) / (k_max * 1e-3) / 1e9 / ((1 if getattr(rt, "dry_run", False) else world) * HBM_PEAK_GBS),
        "arithmetic": s3.last_arithmetic(),
        "north_star": ">= 1e7 solves/s on 8 GPUs at >= 0.30 of the state-streaming roofline (BASELINE.json)",
    }


def secondary_figures(args, amd, solver, torch):
    """Informational numbers printed with the default workload (none of them is `value`)."""
    out = {}
    dev = solver.device

    def rate(s, x0, reps):
        s.minimize(amd.Rosenbrock(), x0)
        torch.cuda.synchronize()
        ms = []
        t0 = time.perf_counter()
        for _ in range(reps):
            o = s.minimize(amd.Rosenbrock(), x0)
            ms.append(s.last_kernel_ms())
        torch.cuda.synchronize()
        return o, (time.perf_counter() - t0) / reps, float(np.mean(ms))

    # (1) the configs[2] per-GPU shard and the whole configs[2] batch on this one GPU (n = 64, m = 10)
    w3 = WORKLOADS["cfg3"]
    s3 = amd.BatchedLbfgs(m=w3["m"], stopping_progress=amd.parity_stop(), context=solver.ctx, arithmetic=args.arithmetic)
    for key, B3, reps in (("secondary_cfg3_shard", w3["B"], 2), ("secondary_cfg3_full_batch_one_gpu", 1048576, 1)):
        x03 = s3.fill_x0(B3, w3["n"], args.x0, SEED)
        o3, dt3, ms3 = rate(s3, x03, reps)
        p3 = amd.progress_to_numpy(o3[3])
        b3 = algorithmic_bytes(w3["n"], int(p3["num_iterations"].sum()), int(p3["sum_k"].sum()))
        out[key] = {"workload": "%d x Rosenbrock-64, L-BFGS m=10, fp64, parity stopping" % B3, "value": B3 / dt3,
                    "unit": "solves/s", "kernel_ms": ms3, "state_streaming_GBs": b3 / (ms3 * 1e-3) / 1e9,
                    "mean_iterations": float(p3["num_iterations"].mean()), "arithmetic": s3.last_arithmetic()}
        del x03, o3

    # (2) the other stopping rows of SURVEY 8d on the headline batch, and the exact-arithmetic build beside the fused one
    w2 = WORKLOADS["cfg2"]
    x02 = solver.fill_x0(w2["B"], w2["n"], args.x0, SEED)
    for key, stop, arith in (("secondary_variant_a_stop", "variant_a", args.arithmetic),
                             ("secondary_default_preset_stop", "default", args.arithmetic),
                             ("secondary_exact_arithmetic", "parity", "exact")):
        st = amd.parity_stop() if stop != "default" else amd.capi.default_stop()
        if stop == "variant_a":
            st.x_delta = 1e-9
        s2 = amd.BatchedLbfgs(m=w2["m"], stopping_progress=st, context=solver.ctx, arithmetic=arith)
        o2, dt2, ms2 = rate(s2, x02, 3)
        p2 = amd.progress_to_numpy(o2[3])
        out[key] = {"workload": "configs[1] batch, stopping '%s', arithmetic %s" % (stop, s2.last_arithmetic()),
                    "value": w2["B"] / dt2, "unit": "solves/s", "kernel_ms": ms2,
                    "mean_iterations": float(p2["num_iterations"].mean()),
                    "max_iterations": int(p2["num_iterations"].max())}

    # (3) a stream of batches: two contexts on two streams, so that the tail of one batch (a handful of long solves on
    # an otherwise empty chip) overlaps the bulk of the next — every launch still solves its whole batch
    ctx_b = amd.Context(solver.ctx.device)
    sa = amd.BatchedLbfgs(m=w2["m"], stopping_progress=amd.parity_stop(), context=solver.ctx, arithmetic=args.arithmetic)
    sb = amd.BatchedLbfgs(m=w2["m"], stopping_progress=amd.parity_stop(), context=ctx_b, arithmetic=args.arithmetic)
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    for warm in (True, False):
        reps = 2 if warm else 8
        t0 = time.perf_counter()
        for i in range(reps):
            with torch.cuda.stream(streams[i & 1]):
                (sa if (i & 1) == 0 else sb).minimize(amd.Rosenbrock(), x02)
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / reps
    out["secondary_two_streams_pipelined"] = {
        "workload": "configs[1] batches issued alternately on two streams / two contexts (tails overlap the next batch)",
        "value": w2["B"] / dtp, "unit": "solves/s", "ms_per_batch": dtp * 1e3}
    ctx_b.close()

    # (4) constrained solves through the augmented-Lagrangian outer loop (the full line with its CPU
    # baseline and parity check comes from scripts/auglag_bench.py): 16 384 problems of n = 64,
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
    out["secondary_augmented_lagrangian"] = {
        "workload": "16,384 constrained problems, n = 64: diagonal quadratic, one equality, one inequality; "
                    "Lbfgs<m=10> inner solver, whole outer loop in one kernel launch",
        "value": Bal / dta, "unit": "solves/s", "ms": dta * 1e3,
        "finished_fraction": float(np.mean(pa["status"] == 6)), "max_violation": float(viol.max().item()),
        "mean_outer_iterations": float(pa["num_iterations"].mean()),
        "mean_inner_iterations": float(pa["inner_iterations"].mean())}
    return out


if __name__ == "__main__":
    main()
