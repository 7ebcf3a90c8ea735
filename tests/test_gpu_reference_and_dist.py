"""GPU tests that close two chains on the GPU box itself:

* the device against the REFERENCE BINARY directly (oracle/_ref/libref.so: the unmodified reference headers over the
  Eigen shim, built in the authoring container and shipped as a binary) — one assertion instead of the three-link
  chain device == twin, twin ~ reference order, reference order == libref;
* the multi-GPU launch path: bench.py under torch.distributed.run with the "nccl" (RCCL) backend at world size 1, so
  that process-group initialisation, the device-side stop-flag all-reduce and the MAX-over-ranks timing are green
  before the driver's 8-GPU run.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-6


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    dst = capi.Stop()
    for name, _ in oracle_stop._fields_:
        setattr(dst, name, getattr(oracle_stop, name))
    return dst


@pytest.mark.parametrize("n,m,B,kind", [(32, 6, 192, "std"), (64, 10, 96, "std"), (32, 6, 64, "u2"), (2, 10, 16, "u2")])
@pytest.mark.parametrize("arithmetic", ["exact", "fma"])
def test_lbfgs_device_vs_reference_binary(gpu_solver_factory, oracle, reference, n, m, B, kind, arithmetic):
    """configs[1] / configs[2] shapes: x*, f* of the device within 1e-6 of what the reference's own
    Lbfgs<F, m>::Minimize returns for the same start points (parity stopping), for both arithmetic builds."""
    import torch
    import cppnumericalsolvers_amd as amd
    x0 = amd.synthetic_x0_host(B, n, kind)
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), arithmetic=arithmetic)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    xr, fr, gr, pr = reference.minimize_batch("rosenbrock", x0, m=m, stop=st)
    assert np.max(np.abs(x.cpu().numpy() - xr)) <= TOL
    assert np.max(np.abs(f.cpu().numpy() - fr)) <= TOL
    pg = amd.progress_to_numpy(p)
    assert np.all(pg["status"] != 1) and np.all(pr["status"] != 1)
    # iteration counts are informational (different rounding, same algorithm): they stay close on average
    assert abs(pg["num_iterations"].mean() - pr["num_iterations"].mean()) <= 0.05 * pr["num_iterations"].mean() + 2


@pytest.mark.parametrize("n,m", [(32, 6), (64, 10), (7, 5)])
def test_hager_zhang_lbfgs_device_vs_reference_binary(gpu_solver_factory, oracle, reference, n, m):
    """Lbfgs<F, m, HagerZhang> in ONE assertion against the reference's own class (oracle/_ref: linesearch/hager_zhang.h
    under solver/lbfgs.h): x*, f* within 1e-6 under parity stopping.  (The two-link chain — device == twin bit for bit,
    twin == reference binary bit for bit — is tests/test_gpu_parity.py and tests/test_oracle.py.)"""
    import torch
    import cppnumericalsolvers_amd as amd
    x0 = amd.synthetic_x0_host(96, n, "std" if n % 2 == 0 else "u2")
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), linesearch="hager_zhang")
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    xr, fr, gr, pr = reference.minimize_batch("rosenbrock", x0, m=m, stop=st, linesearch="hager_zhang")
    assert np.all(amd.progress_to_numpy(p)["status"] != 1) and np.all(pr["status"] != 1)
    assert max(np.max(np.abs(x.cpu().numpy() - xr)), np.max(np.abs(f.cpu().numpy() - fr))) <= TOL


@pytest.mark.parametrize("n,linesearch", [(16, "more_thuente"), (32, "more_thuente"), (64, "hager_zhang")])
def test_dense_bfgs_device_vs_reference_binary(gpu_solver_factory, oracle, reference, n, linesearch):
    """Bfgs<F, LineSearch> in ONE assertion against the reference's own solver/bfgs.h (oracle/_ref): x*, f* within 1e-6."""
    import torch
    import cppnumericalsolvers_amd as amd
    x0 = amd.synthetic_x0_host(48, n, "std")
    st = oracle.parity_stop()
    base = gpu_solver_factory()
    s = amd.BatchedBfgs(stopping_progress=_engine_stop(st), context=base.ctx, linesearch=linesearch)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    xr, fr, gr, pr = reference.bfgs_minimize_batch("rosenbrock", x0, stop=st, linesearch=linesearch)
    assert np.all(amd.progress_to_numpy(p)["status"] != 1) and np.all(pr["status"] != 1)
    assert max(np.max(np.abs(x.cpu().numpy() - xr)), np.max(np.abs(f.cpu().numpy() - fr))) <= TOL


def test_augmented_lagrangian_device_vs_reference_binary(reference):
    """AugmentedLagrangian<Problem, Lbfgs<FunctionExpr>> in ONE assertion against the reference's own solver
    (oracle/_ref: solver/augmented_lagrangian.h, function_penalty.h, progress.h's constrained branch): a diagonal quadratic on
    the simplex with one inequality, tight inner and outer stopping so that both orders of summation converge to the same
    constrained minimiser: x*, multipliers and violation within 1e-6 — and the same with a constraint family of forty rows."""
    import auglag_lib as al
    import oracle_lib
    from cppnumericalsolvers_amd import BatchedAugmentedLagrangian
    from test_gpu_auglag import _engine_problem
    inner = oracle_lib.make_stop(num_iterations=10000, x_delta=1e-12, x_delta_violations=1, f_delta=0.0,
                                 gradient_norm=1e-10, past=0)
    cfg = al.default_config(constraint_threshold=1e-8, kkt_stationarity_threshold=1e-6, outer_num_iterations=60)
    for p, B in ((al.quadratic_simplex_problem(24, seed=2), 24), (al.random_family_problem(20, 6, 34, seed=3, table=True), 12)):
        x0 = np.random.default_rng(p.n).uniform(-1, 1, (B, p.n))
        s = BatchedAugmentedLagrangian(inner_stopping_progress=_engine_stop(inner))
        c = s.default_config()
        for name, _ in cfg._fields_:
            setattr(c, name, getattr(cfg, name))
        s.config = c
        d = s.minimize_host(_engine_problem(p), x0)
        r = al.ref_minimize(p, x0, config=cfg, inner_stop=inner)
        assert np.all(d["progress"]["status"] == 6) and np.all(r["progress"]["status"] == 6)       # Finished, both
        worst = max(np.max(np.abs(d["x"] - r["x"])), np.max(np.abs(d["max_violation"] - r["max_violation"])))
        assert worst <= TOL, worst
        assert np.max(np.abs(d["lambda"] - r["lambda"])) <= 1e-5 and np.max(np.abs(d["mu"] - r["mu"])) <= 1e-5


@pytest.mark.parametrize("n,m", [(32, 5), (32, 6), (64, 10)])
def test_lbfgsb_device_vs_reference_binary(gpu_solver_factory, oracle, reference, n, m):
    """configs[4] shape: Lbfgsb<F, 5> in the box [-1.5, 0.8]^32 against the reference's own Lbfgsb; the same for the
    other two kernel layouts: m = 6 (eight columns, 16 rows of the compact representation) and m = 10 (32 lanes per
    problem), `Lbfgsb<F, 6>` / `Lbfgsb<F, 10>` of the reference."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    B = 96 if m == 5 else 48
    x0 = amd.synthetic_x0_host(B, n, "u2")
    st = oracle.lbfgsb_default_stop()
    st.x_delta, st.f_delta, st.gradient_norm, st.past = 1e-11, 0.0, 1e-8, 0
    base = gpu_solver_factory()
    s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=_engine_stop(st), context=base.ctx)
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    s.SetBounds(lo, hi)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    L = reference.lib()
    xr, gr, fr = np.empty_like(x0), np.empty_like(x0), np.empty(B)
    pr = np.zeros(B, dtype=oracle.PROGRESS_DTYPE)
    dp = oracle._dp
    pz = np.zeros(1)
    import ctypes as C
    rc = L.ref_lbfgsb_minimize_batch(0, dp(pz), n, m, B, C.byref(st), dp(lo), dp(hi), dp(x0), dp(xr), dp(fr), dp(gr),
                                     pr.ctypes.data)
    assert rc == 0
    assert np.max(np.abs(x.cpu().numpy() - xr)) <= TOL
    assert np.max(np.abs(f.cpu().numpy() - fr)) <= TOL


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_under_torchrun_with_the_rccl_backend():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the driver's multi-GPU launch line at
    the one world size a single-GPU box can run.  Covers init_process_group("nccl"), the device-side all-reduce of the
    3-word convergence record, the barrier and the MAX-over-ranks reduction of the elapsed time."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MI355_BENCH_STRONG_ROW="1", MI355_BENCH_STRONG_BATCH="16384")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--batch", "8192", "--no-secondary", "--no-cpu-baseline", "--no-counters"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "solves/s"
    assert d["config"]["all_converged"] is True and d["config"]["unconverged"] == 0
    assert d["config"]["problems_total"] == 8192 and d["value"] > 0
    assert d["roofline"]["kernel_ms"] > 0
    # the strong-scaled configs[2] row the line carries under --gpus N > 1 (forced here at world size 1, reduced batch)
    st = d["secondary_cfg3full_strong"]
    assert st["scaling"] == "strong" and st["rccl_ranks"] == 1 and st["n_gpus"] == 1 and st["value"] > 0
    assert st["problems_per_rank"] == [16384] and st["global_record"]["total"] == 16384
    assert st["global_record"]["unconverged"] == 0 and len(st["kernel_ms_per_rank"]) == 1
    assert d["multi_gpu"]["measured"] is False and d["multi_gpu"]["ranks_in_this_run"] == 1
    assert d["multi_gpu"]["launch"] == "rank-of-launcher" and d["multi_gpu"]["rccl_ranks"] == 1


def test_bench_launches_its_own_ranks_and_refuses_a_mislabelled_line():
    """Plain `python bench.py --gpus N` (no RANK / WORLD_SIZE in the environment — the shape of the driver's one-GPU
    command): with --launcher torchrun the script re-executes itself under torch.distributed.run (exercised here at
    N = 1, the only world size a one-GPU box has); with --gpus 2 on this box it must exit non-zero and print no JSON
    line instead of an `n_gpus: 1` one."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4096",
            "--no-secondary", "--no-cpu-baseline", "--no-counters"]
    out = subprocess.run(base + ["--gpus", "1", "--launcher", "torchrun"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["problems_total"] == 4096
    mg = d["multi_gpu"]
    assert mg["launch"] == "self-launch" and mg["rccl_ranks"] == 1 and mg["ranks_in_this_run"] == 1
    assert mg["problems_per_rank"] == [4096] and len(mg["kernel_ms_per_rank"]) == 1 and len(mg["devices"]) == 1
    want = torch.cuda.device_count() + 1
    out = subprocess.run(base + ["--gpus", str(want)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert "only %d GPU(s) visible" % (want - 1) in out.stderr
    out = subprocess.run(base + ["--gpus", str(want), "--launch-plan"], cwd=ROOT, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 3
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert plan["mode"] == "self-launch" and plan["visible_gpus"] == want - 1 and plan["error"]


def test_sharded_driver_on_the_gpu_with_a_process_group(gpu_solver_factory):
    """ShardedLbfgs.minimize_global on the real solver inside an initialised (world size 1, RCCL) process group."""
    import torch
    import torch.distributed as dist
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import sharded
    created = False
    if not dist.is_initialized():
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        created = True
    try:
        s = gpu_solver_factory(m=6, stopping_progress=amd.parity_stop())
        drv = sharded.ShardedLbfgs(s, rank=0, world_size=1)
        (lo, hi), (x, f, g, prog), flag = drv.minimize_global(
            amd.Rosenbrock(), 3000, lambda first, count: s.fill_x0(count, 32, "std", first_problem=first))
        assert (lo, hi) == (0, 3000) and flag.total == 3000 and flag.all_converged
        assert flag.iterations == int(amd.progress_to_numpy(prog)["num_iterations"].sum())
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("arithmetic,twin", [("exact", dict(reduction="butterfly")),
                                             ("fma", dict(reduction="butterfly_fma", fma_group=4))])
def test_config1_every_problem_of_the_batch(gpu_solver_factory, oracle, arithmetic, twin):
    """BASELINE configs[1] at its full size, ALL 65,536 problems: the device equals its twin bit for bit (x*, f*,
    status, iteration and evaluation counts) and is within 1e-6 of the reference-order solve — not a sample."""
    import torch
    import cppnumericalsolvers_amd as amd
    B, n, m = 65536, 32, 6
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), arithmetic=arithmetic)
    x0 = s.fill_x0(B, n, "std")
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    x0h, x, f, pg = x0.cpu().numpy(), x.cpu().numpy(), f.cpu().numpy(), amd.progress_to_numpy(p)
    xb, fb, _, pb = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=st, width=32, **twin)
    np.testing.assert_array_equal(x, xb)
    np.testing.assert_array_equal(f, fb)
    for k in ("status", "num_iterations", "nfev", "sum_k"):
        np.testing.assert_array_equal(pg[k], pb[k], err_msg=k)
    xs, fs, _, ps = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=st)
    assert np.max(np.abs(x - xs)) <= TOL and np.max(np.abs(f - fs)) <= TOL
    assert np.all(pg["status"] != 1) and np.all(ps["status"] != 1)


# The every-problem comparisons with the reference binary are host-core bound (1,048,576 reference solves: 189 s of a
# 760 s suite, profiles/r5_gpu_durations_before.txt).  By default they compare a STRIDED sample (every 8th problem of
# configs[2], every 4th of configs[3]: 131,072 / 65,536 problems spread over the whole batch) — the GPU still solves and
# checks the convergence of EVERY problem — and MI355_FULL_PARITY=1 compares every problem, as run once per round for the
# record (profiles/).  configs[1] and configs[4] stay every-problem: they take seconds.
# Since round 6 configs[2] — the 1,048,576-problem row the north-star target is quoted on — is compared on EVERY problem
# by default (round-5 verdict, "Next" 3: ~ +150 s of the suite on the driver's box); MI355_FULL_PARITY=0 brings the
# strided sample back for a quick local run.  configs[3] stays strided unless MI355_FULL_PARITY=1.
FULL_PARITY = os.environ.get("MI355_FULL_PARITY") == "1"
FULL_PARITY_CONFIG2 = os.environ.get("MI355_FULL_PARITY") != "0"


@pytest.fixture(scope="module")
def config2_solved(gpu_solver_factory, oracle):
    """configs[2] at its stated size — 1,048,576 x Rosenbrock-64, m = 10 — solved ONCE on the GPU for the tests below."""
    import torch
    import cppnumericalsolvers_amd as amd
    B, n, m = 1048576, 64, 10
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), arithmetic="fma")
    x0 = s.fill_x0(B, n, "std")
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0, want_gradient=False)
    torch.cuda.synchronize()
    return dict(B=B, n=n, m=m, st=st, x0=x0, x=x, f=f, p=p)


def test_config2_full_batch_on_one_gpu_every_shard_range(config2_solved, oracle):
    """BASELINE configs[2] at its stated size — 1,048,576 x Rosenbrock-64, m = 10 — solved on ONE GPU (the G = 1 row of
    BASELINE.md section 4); 8,192 problems from each of the eight per-GPU shard ranges (65,536 in all) are compared
    with the twin bit for bit and with the reference-order solve at 1e-6; every problem converged."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import sharded
    c = config2_solved
    B, n, m, st, x0, x, f, p = c["B"], c["n"], c["m"], c["st"], c["x0"], c["x"], c["f"], c["p"]
    G, K = 8, 8192
    status, iters, _, _ = sharded.progress_fields_device(p)
    assert int((status <= 1).sum().item()) == 0            # nobody hit the iteration limit
    assert 300 < float(iters.float().mean().item()) < 450
    pg = amd.progress_to_numpy(p)
    for r in range(G):
        lo, hi = sharded.shard_range(B, r, G)
        sel = slice(lo + (hi - lo) // 3, lo + (hi - lo) // 3 + K)   # a block inside shard r
        x0h = x0[sel].cpu().numpy()
        np.testing.assert_array_equal(x0h, amd.synthetic_x0_host(K, n, "std", first_problem=sel.start))
        xb, fb, _, pb = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=st, reduction="butterfly_fma", width=64,
                                              fma_group=4)
        np.testing.assert_array_equal(x[sel].cpu().numpy(), xb, err_msg="shard %d" % r)
        np.testing.assert_array_equal(f[sel].cpu().numpy(), fb)
        np.testing.assert_array_equal(pg["num_iterations"][sel], pb["num_iterations"])
        xs, fs, _, _ = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=st)
        assert np.max(np.abs(xb - xs)) <= TOL and np.max(np.abs(fb - fs)) <= TOL


def test_config2_every_problem_vs_reference_binary(config2_solved, reference):
    """BASELINE configs[2] at its full size, 1,048,576 x Rosenbrock-64 (m = 10) — the north star's target row — solved on
    one GPU in the production (fused) arithmetic (every problem converged) and compared with the REFERENCE BINARY (the
    reference's own Lbfgs<F, 10> over the Eigen stand-in, oracle/_ref, all host threads): x* and f* within 1e-6 on every
    EVERY problem of the batch by default (every 8th with MI355_FULL_PARITY=0).  The max|dx| line goes into pytest's
    terminal summary (tests/conftest.py), i.e. into the tail of the driver's GPU-test record."""
    import cppnumericalsolvers_amd as amd
    import conftest
    import time
    c = config2_solved
    step = 1 if FULL_PARITY_CONFIG2 else 8
    assert np.all(amd.progress_to_numpy(c["p"])["status"] != 1)
    # The reference leg is CPU work on the box's host cores (147-173 s for every problem on the boxes seen so far).  It runs
    # in eight blocks under a time budget (MI355_FULL_PARITY_BUDGET_S, default 480 s of the suite's 1200 s): a slower host
    # widens the stride of the blocks still to come (2, 4, 8) instead of running the suite into its limit, and the line
    # below always states how many problems were compared.
    budget = float(os.environ.get("MI355_FULL_PARITY_BUDGET_S", "480"))
    x0_all, x_all, f_all = c["x0"].cpu().numpy(), c["x"].cpu().numpy(), c["f"].cpu().numpy()
    B, blocks = c["B"], 8
    t0, compared, dx, df, widened = time.time(), 0, 0.0, 0.0, False
    for b in range(blocks):
        lo, hi = b * B // blocks, (b + 1) * B // blocks
        idx = np.arange(lo, hi, step)
        xr, fr, _, pr = reference.minimize_batch_threaded("rosenbrock", np.ascontiguousarray(x0_all[idx]), m=c["m"], stop=c["st"],
                                                          threads=os.cpu_count() or 8, chunk=256)
        assert np.all(pr["status"] != 1)
        dx = max(dx, float(np.max(np.abs(x_all[idx] - xr))))
        df = max(df, float(np.max(np.abs(f_all[idx] - fr))))
        compared += len(idx)
        elapsed = time.time() - t0
        left = (B - hi) / step
        while step < 8 and left > 0 and elapsed + left * (elapsed / compared) > budget:
            step, left, widened = step * 2, left / 2, True
    assert dx <= TOL and df <= TOL, (dx, df)
    line = ("configs[2] (1,048,576 x Rosenbrock-64, m = 10): %d of %d problems against the reference binary, "
            "max|dx| %.3g max|df| %.3g (tolerance %g)%s" % (compared, c["B"], dx, df, TOL,
            " [stride widened to keep the reference leg within %.0f s on this host]" % budget if widened else ""))
    print(line)
    conftest.record_summary_line(line)


def test_config3_every_problem_vs_reference_binary(gpu_solver_factory, oracle, reference):
    """BASELINE configs[3] at its full size, 262,144 ridge problems (A 128 x 64, lambda 0.1, one right-hand side
    each): both device forms -- the normal-equation form (objective id 5, what bench.py --workload cfg4 times) and the
    per-evaluation matrix-core kernel (id 3) -- solve the WHOLE batch (every problem converged) and are compared with the
    REFERENCE BINARY: the README functors SquaredError(A, y_b) + lambda * L2Reg composed by the reference's own expression
    templates and minimised by its Lbfgs<F, 10> (oracle/_ref, all host threads).  x* and f* within 1e-6 on every 4th
    problem by default, on EVERY problem with MI355_FULL_PARITY=1."""
    import torch
    import cppnumericalsolvers_amd as amd
    B, rows, n, m, lam = 262144, 128, 64, 10, 0.1
    step = 1 if FULL_PARITY else 4
    st = oracle.parity_stop()
    A, Y = amd.synthetic_ridge_host(B, rows, n, 20260923)
    x0 = np.zeros((B, n))
    # (the pinned build of the reference, oracle/_ref/libref.so -- not the -O3 timing build)
    xr, fr, _, pr = reference.ridge_minimize_batch_threaded(A, lam, np.ascontiguousarray(Y[::step]), x0[::step], stop=st,
                                                            threads=os.cpu_count() or 8, chunk=64)
    assert np.all(pr["status"] != 1)
    Yd, x0d = _to_dev(Y), _to_dev(x0)
    for form in (dict(gram=True), dict(matrix_cores=True)):
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), arithmetic="default")
        x, f, g, p = s.minimize(amd.SquaredErrorRidge(A, lam, **form), x0d, per_problem=Yd)
        torch.cuda.synchronize()
        x, f = x.cpu().numpy()[::step], f.cpu().numpy()[::step]
        assert np.all(amd.progress_to_numpy(p)["status"] != 1)
        assert np.max(np.abs(x - xr)) <= TOL, (form, float(np.max(np.abs(x - xr))))
        assert np.max(np.abs(f - fr)) <= TOL, (form, float(np.max(np.abs(f - fr))))     # absolute, as the north star says
