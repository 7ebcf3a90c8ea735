"""GPU tests of the round-2 boundary: the per-iteration trace, the host-pointer pipeline (single slot and chunked),
the device group with its RCCL all-reduce — through the Python binding of the C-ABI."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    dst = capi.Stop()
    for name, _ in oracle_stop._fields_:
        setattr(dst, name, getattr(oracle_stop, name))
    return dst


@pytest.mark.parametrize("arithmetic", ["exact", "fma"])
def test_trace_records_every_iteration(gpu_solver_factory, oracle, arithmetic):
    """Traced problems: one record per iteration (value, deltas, gradient norm, status) + the iterate and its
    gradient; the last record is the returned state; the records of a traced solve equal a prefix-limited solve's
    results (iteration k of the trace == the result of the same solve stopped after k iterations)."""
    import torch
    import cppnumericalsolvers_amd as amd
    n, m, B = 32, 6, 40
    x0 = amd.synthetic_x0_host(B, n, "std")
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), arithmetic=arithmetic)
    traced = [3, 17, 39, 0]
    tr = amd.Trace(traced, capacity=4096, n=n, device=s.device, with_x=True, with_g=True)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0), trace=tr)
    torch.cuda.synchronize()
    x, f, g = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
    pn = amd.progress_to_numpy(p)
    # the same batch without a trace: identical results
    x2, f2, _, p2 = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(x, x2.cpu().numpy())
    for i, b in enumerate(traced):
        rec, tx, tg = tr.history(i)
        T = int(pn["num_iterations"][b])
        assert len(rec) == T and int(tr.written[i].item()) == T
        np.testing.assert_array_equal(rec["num_iterations"], np.arange(1, T + 1))
        assert np.all(rec["status"][:-1] == 0) and rec["status"][-1] == pn["status"][b]
        assert rec["value"][-1] == f[b]
        np.testing.assert_array_equal(tx[-1], x[b])
        np.testing.assert_array_equal(tg[-1], g[b])
        assert rec["x_delta"][-1] == pn["x_delta"][b] and rec["gradient_norm"][-1] == pn["gradient_norm"][b]
        assert np.all(np.diff(rec["value"]) <= 0)            # sufficient decrease every iteration
        # iteration k of the trace is what a solve limited to k - 1 iterations returns (strict '>' of progress.h:212)
        for k in (2, 5, T // 2):
            lim = oracle.parity_stop()
            lim.num_iterations = k - 1
            sk = gpu_solver_factory(m=m, stopping_progress=_engine_stop(lim), arithmetic=arithmetic)
            xk, fk, _, pk = sk.minimize(amd.Rosenbrock(), _to_dev(x0[b:b + 1]))
            torch.cuda.synchronize()
            assert int(amd.progress_to_numpy(pk)["num_iterations"][0]) == k
            np.testing.assert_array_equal(xk.cpu().numpy()[0], tx[k - 1])
            assert fk.cpu().numpy()[0] == rec["value"][k - 1]


def test_trace_ring_keeps_the_tail_and_lbfgsb_traces_too(gpu_solver_factory, oracle):
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    n, B = 16, 8
    x0 = amd.synthetic_x0_host(B, n, "u2")
    s = gpu_solver_factory(m=5, stopping_progress=_engine_stop(oracle.parity_stop()))
    tr = amd.Trace([5], capacity=7, n=n, device=s.device)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0), trace=tr)
    torch.cuda.synchronize()
    T = int(amd.progress_to_numpy(p)["num_iterations"][5])
    rec, tx, _ = tr.history(0)
    assert T > 7 and len(rec) == 7
    np.testing.assert_array_equal(rec["num_iterations"], np.arange(T - 6, T + 1))
    np.testing.assert_array_equal(tx[-1], x.cpu().numpy()[5])
    # L-BFGS-B
    sb = amd.BatchedLbfgsb(arithmetic="exact", m=5, context=s.ctx)
    sb.SetBounds(np.full(n, -1.5), np.full(n, 0.8))
    trb = amd.Trace([0, 7], capacity=512, n=n, device=s.device)
    xb, fb, gb, pb = sb.minimize(amd.Rosenbrock(), _to_dev(x0), trace=trb)
    torch.cuda.synchronize()
    pbn = amd.progress_to_numpy(pb)
    for i, b in enumerate((0, 7)):
        rec, tx, _ = trb.history(i)
        assert len(rec) == int(pbn["num_iterations"][b]) and rec["value"][-1] == fb.cpu().numpy()[b]
        assert np.all(tx <= 0.8 + 1e-15) and np.all(tx >= -1.5 - 1e-15)   # every iterate inside the box
    # invalid traces are refused
    bad = amd.Trace([B], capacity=4, n=n, device=s.device)
    with pytest.raises(capi.EngineError):
        s.minimize(amd.Rosenbrock(), _to_dev(x0), trace=bad)


def test_host_entry_single_slot_and_chunked_equal_the_device_entry(gpu_solver_factory, oracle):
    """mi355_lbfgs_minimize_batch_host: pinned staging + persistent device buffers; with a small staging slot the
    batch goes through the chunked, double-buffered loop.  Results are the device entry point's, bit for bit."""
    import torch
    import cppnumericalsolvers_amd as amd
    n, m, B = 32, 6, 5000
    x0 = amd.synthetic_x0_host(B, n, "std")
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(oracle.parity_stop()))
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    for stage in (None, "200000", "70000"):
        if stage:
            os.environ["MI355_HOST_STAGE_BYTES"] = stage
        try:
            for _ in range(2):   # second call: warm buffers
                xh, fh, gh, ph = s.minimize_host(amd.Rosenbrock(), x0)
        finally:
            os.environ.pop("MI355_HOST_STAGE_BYTES", None)
        np.testing.assert_array_equal(xh, x.cpu().numpy())
        np.testing.assert_array_equal(fh, f.cpu().numpy())
        np.testing.assert_array_equal(gh, g.cpu().numpy())
        for k in ("status", "num_iterations", "nfev", "x_delta"):
            np.testing.assert_array_equal(ph[k], pn[k])
    # per-problem data (ridge right-hand sides) travel through the staging slots as well, chunked
    rows = 24
    A, Y = amd.synthetic_ridge_host(600, rows, n)
    obj = amd.SquaredErrorRidge(A, 0.1)
    z = np.zeros((600, n))
    xd, fd, _, _ = s.minimize(obj, _to_dev(z), per_problem=_to_dev(Y))
    torch.cuda.synchronize()
    os.environ["MI355_HOST_STAGE_BYTES"] = "80000"
    try:
        xh, fh, _, _ = s.minimize_host(obj, z, per_problem=Y)
    finally:
        os.environ.pop("MI355_HOST_STAGE_BYTES", None)
    np.testing.assert_array_equal(xh, xd.cpu().numpy())
    np.testing.assert_array_equal(fh, fd.cpu().numpy())


def test_device_group_shards_and_allreduces(gpu_solver_factory, oracle):
    """mi355_lbfgs_group: two contexts on device 0, contiguous shards solved by two host threads, the RCCL all-reduce
    of [problems, unconverged, iterations]; the sharded result is the unsharded one bit for bit."""
    import torch
    import cppnumericalsolvers_amd as amd
    n, m, B = 32, 6, 4001    # ragged split
    x0 = amd.synthetic_x0_host(B, n, "std")
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(oracle.parity_stop()))
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    for devices in ([0], [0, 0], [0, 0, 0]):
        grp = amd.DeviceGroup(devices)
        assert grp.size() == len(devices)
        xs, fs, gs, ps, flag = grp.minimize_host(s, amd.Rosenbrock(), x0)
        np.testing.assert_array_equal(xs, x.cpu().numpy())
        np.testing.assert_array_equal(fs, f.cpu().numpy())
        assert flag["total"] == B and flag["unconverged"] == 0 and flag["all_converged"]
        assert flag["iterations"] == int(pn["num_iterations"].sum())
        lim = oracle.parity_stop()
        lim.num_iterations = 7
        s2 = gpu_solver_factory(m=m, stopping_progress=_engine_stop(lim))
        _, _, _, ps2, flag2 = grp.minimize_host(s2, amd.Rosenbrock(), x0)
        assert flag2["unconverged"] == int((ps2["status"] <= 1).sum()) == B and not flag2["all_converged"]
        grp.close()


def test_device_group_lbfgsb_bfgs_and_device_resident_shards(gpu_solver_factory, oracle):
    """The rest of the group API (include/mi355_lbfgs.h, "more than one GPU"): Lbfgsb and Bfgs over host arrays, the
    device-resident sharded solves (per-member device pointers; solve and record count on each member's own stream),
    and the stand-alone collective after solves the caller enqueued itself.  Sharded == unsharded bit for bit."""
    import ctypes as C
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    base = gpu_solver_factory()
    n, B = 32, 1003
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    x0 = amd.synthetic_x0_host(B, n, "u2", seed=17)
    for arith in ("exact", "fma"):
        sb = amd.BatchedLbfgsb(m=5, context=base.ctx, arithmetic=arith)
        sb.SetBounds(lo, hi)
        x, f, g, p = sb.minimize(amd.Rosenbrock(), _to_dev(x0))
        torch.cuda.synchronize()
        pn = amd.progress_to_numpy(p)
        for devices in ([0], [0, 0, 0]):
            grp = amd.DeviceGroup(devices)
            xs, fs, gs, ps, flag = grp.minimize_host_lbfgsb(sb, amd.Rosenbrock(), x0, lo, hi)
            np.testing.assert_array_equal(xs, x.cpu().numpy())
            np.testing.assert_array_equal(fs, f.cpu().numpy())
            np.testing.assert_array_equal(ps["num_iterations"], pn["num_iterations"])
            assert flag["total"] == B and flag["iterations"] == int(pn["num_iterations"].sum())
            assert flag["unconverged"] == int((pn["status"] <= 1).sum())
            # device-resident shards of unequal sizes (one of them empty)
            G = len(devices)
            cuts = [0, B] if G == 1 else [0, 400, 400, B]
            shards = [_to_dev(x0[cuts[s]:cuts[s + 1]]) for s in range(G)]
            outs, flagd = grp.minimize_device(sb, amd.Rosenbrock(), shards, lower=lo, upper=hi)
            xd = np.concatenate([o[0].cpu().numpy() for o in outs])
            np.testing.assert_array_equal(xd, x.cpu().numpy())
            assert flagd == flag
            grp.close()
    # Lbfgs, device resident, + the stand-alone collective on progress arrays the caller's own solves produced
    s = gpu_solver_factory(m=6, stopping_progress=_engine_stop(oracle.parity_stop()))
    x0l = amd.synthetic_x0_host(B, n, "std")
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0l))
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    grp = amd.DeviceGroup([0, 0])
    outs, flagd = grp.minimize_device(s, amd.Rosenbrock(), [_to_dev(x0l[:500]), _to_dev(x0l[500:])])
    np.testing.assert_array_equal(np.concatenate([o[0].cpu().numpy() for o in outs]), x.cpu().numpy())
    assert flagd["total"] == B and flagd["iterations"] == int(pn["num_iterations"].sum()) and flagd["all_converged"]
    # allreduce_flags right behind asynchronous solves on torch's stream (no synchronisation by the caller)
    lib = capi.load()
    progs = []
    for part in (x0l[:500], x0l[500:]):
        progs.append(s.minimize(amd.Rosenbrock(), _to_dev(part))[3])
    arr = (C.c_void_p * 2)(*[t.data_ptr() for t in progs])
    counts = (C.c_int64 * 2)(500, B - 500)
    flag = np.zeros(3, dtype=np.uint64)
    capi.check(lib.mi355_lbfgs_group_allreduce_flags(grp._h, arr, counts, flag.ctypes.data))
    assert int(flag[0]) == B and int(flag[1]) == 0 and int(flag[2]) == int(pn["num_iterations"].sum())
    # dense Bfgs over host arrays
    sbf = amd.BatchedBfgs(context=base.ctx)
    xb, fb, gb, pb = sbf.minimize(amd.Rosenbrock(), _to_dev(x0l[:257]))
    xs, fs, gs, ps, flag = grp.minimize_host_bfgs(sbf, amd.Rosenbrock(), x0l[:257])
    np.testing.assert_array_equal(xs, xb.cpu().numpy())
    assert flag["total"] == 257
    grp.close()


def test_device_group_dry_run_of_more_than_one_rank(gpu_solver_factory, oracle, monkeypatch):
    """MI355_GROUP_DRY_RUN_RANKS=1: on a one-GPU box the members of a group are ranks of their own (three contexts on
    device 0 = three ranks) and the all-reduce among them is a host-side sum — the D > 1 code paths of the group (one record
    and one flag buffer per rank, per-rank counting kernels, the agreement check over the ranks, ragged and empty shards)
    run end to end, which a one-rank RCCL communicator never asks of them.  Host entry, device-resident entry and the
    stand-alone collective; sharded == unsharded bit for bit, the global record is the sum over the ranks."""
    import ctypes as C
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    monkeypatch.setenv("MI355_GROUP_DRY_RUN_RANKS", "1")
    n, m, B = 32, 6, 2003
    x0 = amd.synthetic_x0_host(B, n, "std")
    lim = oracle.parity_stop()
    lim.num_iterations = 230                                      # about half of the problems stop on the limit: 0 < unconverged < B
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(lim))
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    bad, its = int((pn["status"] <= 1).sum()), int(pn["num_iterations"].sum())
    assert 0 < bad < B
    grp = amd.DeviceGroup([0, 0, 0])
    assert grp.size() == 3
    xs, fs, gs, ps, flag = grp.minimize_host(s, amd.Rosenbrock(), x0)
    np.testing.assert_array_equal(xs, x.cpu().numpy())
    assert (flag["total"], flag["unconverged"], flag["iterations"]) == (B, bad, its) and not flag["all_converged"]
    cuts = [0, 700, 700, B]                                        # rank 1 holds an empty shard
    shards = [_to_dev(x0[cuts[r]:cuts[r + 1]]) for r in range(3)]
    outs, flagd = grp.minimize_device(s, amd.Rosenbrock(), shards)
    np.testing.assert_array_equal(np.concatenate([o[0].cpu().numpy() for o in outs]), x.cpu().numpy())
    assert flagd == flag
    progs = [s.minimize(amd.Rosenbrock(), sh)[3] if sh.shape[0] else None for sh in shards]
    arr = (C.c_void_p * 3)(*[t.data_ptr() if t is not None else None for t in progs])
    counts = (C.c_int64 * 3)(700, 0, B - 700)
    rec = np.zeros(3, dtype=np.uint64)
    capi.check(capi.load().mi355_lbfgs_group_allreduce_flags(grp._h, arr, counts, rec.ctypes.data))
    assert [int(v) for v in rec] == [B, bad, its]
    grp.close()


@pytest.mark.parametrize("matrix_cores", [False, True])
def test_hessian_condition_stopping(gpu_solver_factory, oracle, matrix_cores):
    """condition_hessian stopping test of Second-mode functions (progress.h:203-210, :318-325): off, on without
    firing, on and firing — device == twin (which equals the reference, tests/test_oracle.py)."""
    import torch
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(5)
    rows, n, B = 20, 6, 37
    A = rng.normal(size=(rows, n))
    Y = rng.normal(size=(B, rows))
    x0 = rng.normal(size=(B, n))
    params = oracle.ridge_params(A, 0.3)
    obj = amd.SquaredErrorRidge(A, 0.3, differentiability="second", matrix_cores=matrix_cores)
    twin = "squared_error_ridge_mfma" if matrix_cores else "squared_error_ridge"
    for threshold in (0.0, 1e9, 2.0):
        s = gpu_solver_factory(m=10, condition_hessian=threshold)
        x, f, g, p = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
        torch.cuda.synchronize()
        pg = amd.progress_to_numpy(p)
        oracle.lib().oracle_set_condition_hessian_stop(threshold)
        try:
            xo, fo, go, po = oracle.minimize_batch(twin, x0, m=10, params=params, per_problem=Y, second_mode=True,
                                                   reduction="butterfly", width=8 if not matrix_cores else 64)
            co = oracle.lib().oracle_last_hessian_condition()
        finally:
            oracle.lib().oracle_set_condition_hessian_stop(0.0)
        np.testing.assert_array_equal(x.cpu().numpy(), xo)
        np.testing.assert_array_equal(pg["status"], po["status"])
        np.testing.assert_array_equal(pg["num_iterations"], po["num_iterations"])
        assert abs(s.last_hessian_condition - co) <= 1e-10 * co
        if threshold == 2.0:
            assert np.all(pg["status"] == 5) and np.all(pg["num_iterations"] == 1)
    # First-mode functions have no such test (progress.h:203 / :318 are `if constexpr (Second)`): the field is ignored
    _, _, _, p1 = gpu_solver_factory(m=5, condition_hessian=10.0).minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    assert np.all(amd.progress_to_numpy(p1)["status"] != 5)


@pytest.mark.parametrize("n,m", [(2, 10), (12, 6), (32, 6), (64, 10), (100, 8), (256, 5)])
def test_second_mode_with_a_non_constant_hessian(gpu_solver_factory, oracle, n, m):
    """Round 3: a Second-mode function whose Hessian is not constant.  The reference rebuilds the
    diagonal preconditioner of the two-loop recursion from function(x, &g, &H) at every iterate (solver/lbfgs.h:116-139);
    the device takes diag H(x) from the functor's hess_diag (mi355_lbfgs_desc::hessian_from_functor).  Device == twin bit
    for bit under both arithmetic policies and both line searches; the twin equals the reference's Lbfgs on the chained
    Rosenbrock declared Second mode (tests/test_oracle.py::test_second_mode_with_a_non_constant_hessian_matches_reference);
    <= 1e-6 from that reference-order solve under tight stopping."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    B = 48
    x0 = amd.synthetic_x0_host(B, n, "std", seed=n + m)
    obj = amd.Rosenbrock(differentiability="second")
    tight = oracle.parity_stop()
    for arithmetic in ("exact", "default"):
        for ls in ("more_thuente", "hager_zhang"):
            for stop_o in (oracle.default_stop(), tight):
                stop = capi.Stop()
                for name, _ in stop_o._fields_:
                    setattr(stop, name, getattr(stop_o, name))
                s = gpu_solver_factory(m=m, stopping_progress=stop, arithmetic=arithmetic, linesearch=ls)
                x, f, g, p = s.minimize(obj, _to_dev(x0))
                torch.cuda.synchronize()
                ll = s.last_launch()
                assert ll["y_columns_in_registers"] == 0       # the history is kept in LDS in this mode
                W, E = ll["lanes_per_problem"], ll["elems_per_lane"]
                fused = arithmetic == "default" and ls == "more_thuente"
                xo, fo, go, po = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, second_mode="functor", linesearch=ls,
                                                       reduction="butterfly_fma" if fused else "butterfly", width=W * E,
                                                       fma_group=E if fused else 0)
                np.testing.assert_array_equal(x.cpu().numpy(), xo)
                np.testing.assert_array_equal(f.cpu().numpy(), fo)
                np.testing.assert_array_equal(g.cpu().numpy(), go)
                pg = amd.progress_to_numpy(p)
                for k in ("status", "num_iterations", "nfev", "sum_k"):
                    np.testing.assert_array_equal(pg[k], po[k], err_msg=k)
            xs, fs, _, ps = oracle.minimize_batch("rosenbrock", x0, m=m, stop=tight, second_mode="functor", linesearch=ls)
            assert np.max(np.abs(x.cpu().numpy() - xs)) <= 1e-6 and np.max(np.abs(f.cpu().numpy() - fs)) <= 1e-6
    # ... and it is a different iteration from the First-mode solve
    _, _, _, p1 = gpu_solver_factory(m=m, stopping_progress=stop).minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    assert not np.array_equal(amd.progress_to_numpy(p1)["num_iterations"], pg["num_iterations"])
    # objectives without a hess_diag, the other solvers and the condition-number test refuse loudly
    with pytest.raises(capi.EngineError) as e:
        dq = amd.DiagQuadratic(np.ones(n), 0.0)
        dq.hessian_from_functor = True
        gpu_solver_factory(m=m).minimize(dq, _to_dev(x0))
    assert e.value.code == capi.ERR_UNSUPPORTED
    if n > 64:   # (n <= 64: test_condition_hessian_stopping_with_a_non_constant_hessian)
        with pytest.raises(capi.EngineError) as e:
            gpu_solver_factory(m=m, condition_hessian=10.0).minimize(obj, _to_dev(x0))
        assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError):
        amd.BatchedLbfgsb(m=5, context=s.ctx).minimize(obj, _to_dev(x0))


@pytest.mark.parametrize("n,m", [(2, 10), (6, 10), (16, 6), (32, 5), (33, 6), (64, 10)])
def test_condition_hessian_stopping_with_a_non_constant_hessian(gpu_solver_factory, oracle, n, m):
    """Progress::Update of a Second-mode function evaluates function(current_x, nullptr, &H) and
    condition_hessian = ||H|| ||H^-1|| at EVERY iterate (progress.h:203-210) and tests it last (:318-325).  With
    hessian_from_functor the solve kernel builds H(x) in LDS from the functor's hess_full and factorises it itself
    (csrc/hessian_condition_device.hpp).  Device == twin bit for bit — x, f, g, status, iteration and evaluation counts —
    with the test on and never firing and on and firing after a few iterations, under both arithmetic policies and both
    line searches; the twin's condition numbers are the reference binary's bit for bit
    (tests/test_oracle.py::test_condition_hessian_stopping_with_a_non_constant_hessian_matches_reference).  The device's
    Frobenius sums are butterflies where the reference's are chains, so a decision could differ only for a condition number
    within rounding of the threshold: the test asserts none of its problems is that close."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    B = 40
    rng = np.random.default_rng(31 * n + m)
    x0 = np.vstack([np.tile([-1.2, 1.0], n)[:n], rng.uniform(-2, 2, (B - 1, n))])
    obj = amd.Rosenbrock(differentiability="second")
    fired_total = 0
    for arithmetic in ("exact", "default"):
        for ls in ("more_thuente", "hager_zhang"):
            for threshold in (1e13, 3e3):
                s = gpu_solver_factory(m=m, arithmetic=arithmetic, linesearch=ls, condition_hessian=threshold)
                x, f, g, p = s.minimize(obj, _to_dev(x0))
                torch.cuda.synchronize()
                ll = s.last_launch()
                W, E = ll["lanes_per_problem"], ll["elems_per_lane"]
                assert ll["y_columns_in_registers"] == 0 and E <= 2
                fused = arithmetic == "default" and ls == "more_thuente"
                oracle.lib().oracle_set_condition_hessian_stop(threshold)
                try:
                    xo, fo, go, po = oracle.minimize_batch("rosenbrock", x0, m=m, second_mode="functor", linesearch=ls,
                                                           reduction="butterfly_fma" if fused else "butterfly", width=W * E,
                                                           fma_group=E if fused else 0)
                    co = oracle.hessian_conditions(B)
                finally:
                    oracle.lib().oracle_set_condition_hessian_stop(0.0)
                assert np.all(np.abs(co - threshold) > 1e-9 * threshold)
                np.testing.assert_array_equal(x.cpu().numpy(), xo)
                np.testing.assert_array_equal(f.cpu().numpy(), fo)
                np.testing.assert_array_equal(g.cpu().numpy(), go)
                pg = amd.progress_to_numpy(p)
                for k in ("status", "num_iterations", "nfev", "sum_k"):
                    np.testing.assert_array_equal(pg[k], po[k], err_msg=k)
                if threshold == 3e3:
                    fired_total += int(np.sum(pg["status"] == 5))
                    assert np.all(co[pg["status"] == 5] > threshold)
                else:
                    assert np.all(pg["status"] != 5)
    assert fired_total >= (4 if n == 2 else 40)
    # four coordinates per lane has no such kernel; neither has dense BFGS
    with pytest.raises(capi.EngineError) as e:
        gpu_solver_factory(m=m, condition_hessian=3e3, lanes_per_problem=16, elems_per_lane=4).minimize(obj, _to_dev(x0))
    assert e.value.code == capi.ERR_UNSUPPORTED
