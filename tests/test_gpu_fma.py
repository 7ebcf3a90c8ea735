"""GPU parity tests of the fused-arithmetic kernels (mi355_lbfgs_desc.arithmetic = MI355_ARITH_FMA).

Two links, as everywhere: the device equals its CPU twin bit for bit — the oracle's `butterfly_fma` policy
(oracle/lbfgs_oracle.hpp, Reducer::fma_group = coordinates per lane of the kernel) — and that twin, like the device,
stays within the north star's 1e-6 of the reference-order (sequential, unfused) solve under parity stopping.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6
MAPPINGS = [(8, 1), (8, 2), (8, 4), (16, 1), (16, 2), (16, 4), (32, 2), (32, 4), (64, 1), (64, 4)]


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    dst = capi.Stop()
    for name, _ in oracle_stop._fields_:
        setattr(dst, name, getattr(oracle_stop, name))
    return dst


def _solve(s, objective, x0):
    import torch
    import cppnumericalsolvers_amd as amd
    x, f, g, p = s.minimize(objective, _to_dev(x0))
    torch.cuda.synchronize()
    return x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)


def _width(n):
    return 1 << max(3, int(np.ceil(np.log2(n))))


def _same_progress(pg, po):
    for k in ("status", "num_iterations", "nfev", "sum_k", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(pg[k], po[k], err_msg=k)


@pytest.mark.parametrize("n", [2, 5, 31, 32, 33, 64, 100, 256])
def test_fused_objectives_bitwise(gpu_solver_factory, oracle, n):
    """RosenbrockObjective::eval_fma / DiagQuadraticObjective::eval_fma against the twin on every mapping."""
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, size=(19, n))
    a = rng.uniform(0.5, 100, n)
    params = np.concatenate([a, [5.0]])
    for W, E in MAPPINGS:
        if W * E < n:
            continue
        s = gpu_solver_factory(lanes_per_problem=W, elems_per_lane=E, arithmetic="fma")
        f, g = s.evaluate(amd.Rosenbrock(), _to_dev(X))
        fq, gq = s.evaluate(amd.DiagQuadratic(a, 5.0), _to_dev(X))
        f, g, fq, gq = f.cpu().numpy(), g.cpu().numpy(), fq.cpu().numpy(), gq.cpu().numpy()
        width = max(_width(n), E)
        for b in range(X.shape[0]):
            fe, ge = oracle.evaluate("rosenbrock", X[b], reduction="butterfly_fma", width=width, fma_group=E)
            assert f[b] == fe, (W, E, b)
            np.testing.assert_array_equal(g[b], ge)
            fe, ge = oracle.evaluate("diag_quadratic", X[b], params=params, reduction="butterfly_fma", width=width,
                                     fma_group=E)
            assert fq[b] == fe, (W, E, b)
            np.testing.assert_array_equal(gq[b], ge)


@pytest.mark.parametrize("n,m,kind", [(32, 6, "std"), (32, 6, "u2"), (64, 10, "std"), (48, 10, "u2"), (100, 5, "std"),
                                      (20, 7, "u2"), (2, 10, "u2"), (64, 17, "std")])
def test_fused_solves_match_twin_and_reference_order(gpu_solver_factory, oracle, n, m, kind):
    """Full solves under parity stopping: bit-identical to the butterfly_fma twin (values, gradients, status, iteration
    and evaluation counts, deltas), within 1e-6 of the reference-order solve."""
    import cppnumericalsolvers_amd as amd
    B = 256
    x0 = amd.synthetic_x0_host(B, n, kind)
    stop_o = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="fma")
    x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
    assert s.last_arithmetic() == "fma"
    E = s.last_launch()["elems_per_lane"]
    xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly_fma",
                                           width=max(_width(n), E), fma_group=E)
    np.testing.assert_array_equal(x, xb)
    np.testing.assert_array_equal(f, fb)
    np.testing.assert_array_equal(g, gb)
    _same_progress(p, pb)
    xs, fs, _, _ = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o)
    assert np.max(np.abs(x - xs)) <= TOL
    assert np.max(np.abs(f - fs)) <= TOL
    assert np.all(p["status"] != 1)


def test_fused_solves_every_mapping_and_history_placement(gpu_solver_factory, oracle):
    """The fused tree depends on the coordinates per lane (and on nothing else): every mapping equals the twin built
    for its E; the history placement does not matter."""
    import cppnumericalsolvers_amd as amd
    n, m, B = 48, 6, 64
    x0 = amd.synthetic_x0_host(B, n, "u2")
    stop_o = oracle.parity_stop()
    for W, E in [(16, 4), (32, 2), (64, 1), (64, 2), (32, 4)]:
        twin = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly_fma", width=64,
                                     fma_group=E)
        for placement in (1, 2):
            s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="fma", lanes_per_problem=W,
                                   elems_per_lane=E, history_placement=placement)
            x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
            np.testing.assert_array_equal(x, twin[0], err_msg=str((W, E, placement)))
            np.testing.assert_array_equal(f, twin[1])
            _same_progress(p, twin[3])


def test_fused_default_presets_and_quadratic(gpu_solver_factory, oracle):
    """Reference default / conservative presets (plateau ring) and the README quick-start quadratic under the fused
    arithmetic: bit-identical to the twin; the quick-start expectations of the reference hold."""
    import cppnumericalsolvers_amd as amd
    n, m, B = 32, 6, 128
    x0 = amd.synthetic_x0_host(B, n, "u2")
    for preset in ("default", "conservative"):
        st = oracle.default_stop(preset)
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), arithmetic="fma")
        x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
        E = s.last_launch()["elems_per_lane"]
        xb, fb, _, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=st, reduction="butterfly_fma", width=32,
                                              fma_group=E)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        _same_progress(p, pb)
    s = gpu_solver_factory(m=10, arithmetic="fma")
    x, f, g, p = _solve(s, amd.DiagQuadratic([5.0, 100.0], 5.0), np.array([[-10.0, 2.0]]))
    assert np.all(np.abs(x) < 1e-4) and abs(f[0] - 5.0) < 1e-4   # Dockerfile.test:39-42
    E = s.last_launch()["elems_per_lane"]
    xb, fb, _, pb = oracle.minimize_batch("diag_quadratic", np.array([[-10.0, 2.0]]), m=10,
                                          params=np.array([5.0, 100.0, 5.0]), reduction="butterfly_fma", width=8,
                                          fma_group=E)
    np.testing.assert_array_equal(x, xb)
    _same_progress(p, pb)


def test_default_arithmetic_resolution(gpu_solver_factory, oracle):
    """MI355_ARITH_DEFAULT is the fused arithmetic where it is built (Lbfgs with either line search on Rosenbrock /
    DiagQuadratic — Hager-Zhang since round 6, First mode) and the exact one elsewhere; asking for the fused arithmetic where
    it is not built is refused, not ignored."""
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    x0 = amd.synthetic_x0_host(8, 16, "u2")
    s = gpu_solver_factory(m=5, arithmetic="default")
    s.minimize(amd.Rosenbrock(), _to_dev(x0))
    assert s.last_arithmetic() == "fma"
    s = gpu_solver_factory(m=5, arithmetic="default", linesearch="hager_zhang")
    s.minimize(amd.Rosenbrock(), _to_dev(x0))
    assert s.last_arithmetic() == "fma"
    s = gpu_solver_factory(m=5, arithmetic="default", linesearch="hager_zhang")       # Second mode: the exact kernel
    s.minimize(amd.Rosenbrock(differentiability="second"), _to_dev(x0))
    assert s.last_arithmetic() == "exact"
    s = gpu_solver_factory(m=5, arithmetic="exact")
    s.minimize(amd.Rosenbrock(), _to_dev(x0))
    assert s.last_arithmetic() == "exact"
    with pytest.raises(capi.EngineError) as e:    # Hager-Zhang + Second mode has no fused kernel
        gpu_solver_factory(m=5, arithmetic="fma", linesearch="hager_zhang").minimize(
            amd.Rosenbrock(differentiability="second"), _to_dev(x0))
    assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError) as e:    # ... and neither has dense BFGS
        b = amd.BatchedBfgs(context=s.ctx)
        b.arithmetic = capi.ARITH_FMA
        b.minimize(amd.Rosenbrock(), _to_dev(x0))
    assert e.value.code == capi.ERR_UNSUPPORTED
    A = np.random.default_rng(0).normal(size=(12, 16))
    with pytest.raises(capi.EngineError) as e:
        gpu_solver_factory(m=5, arithmetic="fma").minimize(amd.SquaredErrorRidge(A, 0.1), _to_dev(x0),
                                                            per_problem=_to_dev(np.zeros((8, 12))))
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_fused_solver_side_of_the_matrix_core_ridge_kernel(gpu_solver_factory, oracle):
    """The matrix-core ridge kernel under MI355_ARITH_FMA: the solver's inner products, two-loop axpys and trial point are
    fused (the objective's products are MFMA chains under both policies).  Device == the butterfly_fma twin on the same
    objective, for both mappings of the sixteen slots (each has its own chain grouping), <= 1e-6 from the reference-order
    solve and from the closed form; the default arithmetic resolves to the fused one."""
    import torch
    import cppnumericalsolvers_amd as amd
    rows, n, m, lam, B = 128, 64, 10, 0.1, 70
    A, Y = amd.synthetic_ridge_host(B, rows, n, seed=17)
    x0 = np.zeros((B, n))
    params = oracle.ridge_params(A, lam)
    obj = amd.SquaredErrorRidge(A, lam, matrix_cores=True)
    for W, E in ((32, 2), (16, 4)):
        for stop_o in (oracle.default_stop(), oracle.parity_stop()):
            s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="fma", lanes_per_problem=W,
                                   elems_per_lane=E)
            x, f, g, p = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
            torch.cuda.synchronize()
            assert s.last_arithmetic() == "fma"
            x, f, g, p = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)
            xb, fb, gb, pb = oracle.minimize_batch("squared_error_ridge_mfma", x0, m=m, stop=stop_o, params=params,
                                                   reduction="butterfly_fma", fma_group=E, width=64, per_problem=Y)
            np.testing.assert_array_equal(x, xb)
            np.testing.assert_array_equal(f, fb)
            np.testing.assert_array_equal(g, gb)
            _same_progress(p, pb)
        xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge", x0, m=m, stop=oracle.parity_stop(), params=params,
                                             per_problem=Y)
        assert np.max(np.abs(x - xs)) <= TOL and np.max(np.abs(f - fs)) <= TOL
        closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
        assert np.max(np.abs(x - closed)) <= TOL
    s = gpu_solver_factory(m=m, arithmetic="default")
    s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
    assert s.last_arithmetic() == "fma"
    s = gpu_solver_factory(m=m, arithmetic="exact")
    s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
    assert s.last_arithmetic() == "exact"


def test_fused_hostile_starts_match_twin(gpu_solver_factory, oracle):
    """NaN / inf / overflowing start points: the fused kernels take the same branches as their twin."""
    import cppnumericalsolvers_amd as amd
    for n in (8, 32):
        x0 = oracle.hostile_starts(n)
        for st in (oracle.default_stop(), oracle.parity_stop()):
            s = gpu_solver_factory(m=5, stopping_progress=_engine_stop(st), arithmetic="fma")
            x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
            E = s.last_launch()["elems_per_lane"]
            xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=5, stop=st, reduction="butterfly_fma",
                                                   width=_width(n), fma_group=E)
            np.testing.assert_array_equal(x, xb)
            np.testing.assert_array_equal(f, fb)
            np.testing.assert_array_equal(g, gb)
            _same_progress(p, pb)


@pytest.mark.parametrize("n,m,W", [(32, 6, 4), (64, 10, 8), (20, 5, 4), (48, 10, 8), (64, 3, 8)])
def test_eight_coordinates_per_lane(gpu_solver_factory, oracle, n, m, W):
    """The wide-lane mappings (4 x 8 for n <= 32, 8 x 8 for n <= 64: sixteen / eight problems per wavefront at one
    wavefront per SIMD).  Exact arithmetic: the canonical pairwise tree, bit-identical to every other mapping's twin.
    Fused arithmetic: a lane's eight coordinates are two chains of four added pairwise, i.e. the tree of the
    four-coordinate kernels — the same twin (fma_group = 4)."""
    import cppnumericalsolvers_amd as amd
    B = 200
    x0 = amd.synthetic_x0_host(B, n, "std" if n != 48 else "u2")
    stop_o = oracle.parity_stop()
    width = max(_width(n), 8)
    for arithmetic, twin_kw in (("exact", dict(reduction="butterfly", width=width)),
                                ("fma", dict(reduction="butterfly_fma", width=width, fma_group=4))):
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic=arithmetic, lanes_per_problem=W,
                               elems_per_lane=8)
        x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
        ll = s.last_launch()
        assert ll["lanes_per_problem"] == W and ll["elems_per_lane"] == 8 and ll["y_columns_in_registers"] >= m
        xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, **twin_kw)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        _same_progress(p, pb)
        # objective alone, and the diagonal quadratic
        fe, ge = s.evaluate(amd.Rosenbrock(), _to_dev(x0[:9]))
        for b in range(9):
            fo, go = oracle.evaluate("rosenbrock", x0[b], **twin_kw)
            assert fe.cpu().numpy()[b] == fo
            np.testing.assert_array_equal(ge.cpu().numpy()[b], go)
    a = np.linspace(0.5, 30.0, n)
    s = gpu_solver_factory(m=m, arithmetic="fma", lanes_per_problem=W, elems_per_lane=8)
    x, f, g, p = _solve(s, amd.DiagQuadratic(a, 2.0), x0[:32] * 2.0)
    xb, fb, _, pb = oracle.minimize_batch("diag_quadratic", x0[:32] * 2.0, m=m, params=np.concatenate([a, [2.0]]),
                                          reduction="butterfly_fma", width=width, fma_group=4)
    np.testing.assert_array_equal(x, xb)
    _same_progress(p, pb)


# ---- Hager-Zhang in the fused arithmetic (round 6) -------------------------------------------------------------------
@pytest.mark.parametrize("n,m,kind", [(32, 6, "std"), (32, 6, "u2"), (64, 10, "std"), (48, 10, "u2"), (100, 5, "std"),
                                      (20, 7, "u2"), (2, 10, "u2"), (64, 17, "std"), (200, 8, "u2")])
def test_fused_hager_zhang_solves_match_twin_and_reference_order(gpu_solver_factory, oracle, n, m, kind):
    """`Lbfgs<F, m, HagerZhang>` under MI355_ARITH_FMA: the objective and the directional derivative of every trial point
    are the fused forms, the trial point itself stays `x0 - alpha d` — exactly what the oracle's HagerZhang::Run::evaluate
    does under its butterfly_fma policy.  Bit-identical to that twin (values, gradients, status, iteration and evaluation
    counts, deltas) on the library's mapping with the y history in registers, within 1e-6 of the reference-order solve."""
    import cppnumericalsolvers_amd as amd
    B = 192
    x0 = amd.synthetic_x0_host(B, n, kind)
    stop_o = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="fma", linesearch="hager_zhang")
    x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
    assert s.last_arithmetic() == "fma"
    E = s.last_launch()["elems_per_lane"]
    xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly_fma",
                                           width=max(_width(n), E), fma_group=E, linesearch="hager_zhang")
    np.testing.assert_array_equal(x, xb)
    np.testing.assert_array_equal(f, fb)
    np.testing.assert_array_equal(g, gb)
    _same_progress(p, pb)
    xs, fs, _, _ = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, linesearch="hager_zhang")
    assert np.max(np.abs(x - xs)) <= TOL
    assert np.max(np.abs(f - fs)) <= TOL
    assert np.all(p["status"] != 1)


def test_fused_hager_zhang_every_mapping_placement_and_preset(gpu_solver_factory, oracle):
    """The fused Hager-Zhang kernels on every mapping (register history for two and four coordinates per lane, LDS ring
    otherwise and on request), under the default preset (plateau ring) and on the diagonal quadratic; the default
    arithmetic of `Lbfgs<F, m, HagerZhang>` on the built-in objectives IS the fused one now, `exact` keeps the round-2 kernel."""
    import cppnumericalsolvers_amd as amd
    n, m, B = 48, 6, 64
    x0 = amd.synthetic_x0_host(B, n, "u2")
    stop_o = oracle.parity_stop()
    for W, E in [(16, 4), (32, 2), (64, 1), (64, 2), (32, 4)]:
        twin = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly_fma", width=64,
                                     fma_group=E, linesearch="hager_zhang")
        for placement in (1, 2):
            s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="fma", lanes_per_problem=W,
                                   elems_per_lane=E, history_placement=placement, linesearch="hager_zhang")
            x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
            np.testing.assert_array_equal(x, twin[0], err_msg=str((W, E, placement)))
            np.testing.assert_array_equal(f, twin[1])
            _same_progress(p, twin[3])
    x0 = amd.synthetic_x0_host(128, 32, "u2")
    st = oracle.default_stop("default")
    s = gpu_solver_factory(m=6, stopping_progress=_engine_stop(st), arithmetic="default", linesearch="hager_zhang")
    x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
    assert s.last_arithmetic() == "fma"
    E = s.last_launch()["elems_per_lane"]
    xb, fb, _, pb = oracle.minimize_batch("rosenbrock", x0, m=6, stop=st, reduction="butterfly_fma", width=32, fma_group=E,
                                          linesearch="hager_zhang")
    np.testing.assert_array_equal(x, xb)
    _same_progress(p, pb)
    a = np.linspace(1.0, 40.0, 20)
    x0 = amd.synthetic_x0_host(32, 20, "u2")
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(stop_o), arithmetic="fma", linesearch="hager_zhang")
    x, f, g, p = _solve(s, amd.DiagQuadratic(a, 2.0), x0)
    E = s.last_launch()["elems_per_lane"]
    xb, fb, _, pb = oracle.minimize_batch("diag_quadratic", x0, m=10, stop=stop_o, params=np.concatenate([a, [2.0]]),
                                          reduction="butterfly_fma", width=32, fma_group=E, linesearch="hager_zhang")
    np.testing.assert_array_equal(x, xb)
    _same_progress(p, pb)
    s = gpu_solver_factory(m=6, stopping_progress=_engine_stop(stop_o), arithmetic="exact", linesearch="hager_zhang")
    _solve(s, amd.Rosenbrock(), amd.synthetic_x0_host(8, 32))
    assert s.last_arithmetic() == "exact"
