"""L-BFGS-B beyond the reference's default shape: history sizes 6..8 (the template argument of lbfgsb.h:44-49) and a
regression objective under bounds (src/examples/linear_regression.cc:58-74).  Device == oracle twin bit for bit,
<= 1e-6 against the reference-order solve under tight stopping."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
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


def _same_progress(pg, po):
    for k in ("status", "num_iterations", "nfev", "sum_k", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(pg[k], po[k], err_msg=k)


@pytest.mark.parametrize("n,m,boxed", [(32, 6, True), (32, 8, True), (20, 7, False), (64, 8, True), (8, 8, True),
                                       (32, 10, True), (64, 10, True), (20, 9, False), (64, 9, False), (6, 10, True)])
def test_lbfgsb_history_sizes_up_to_ten(gpu_solver_factory, oracle, n, m, boxed):
    """The history size is a template argument of the reference's Lbfgsb (lbfgsb.h:44-49).  m <= 8: 2m <= 16 rows of the
    compact representation, a row per lane of a 16-lane segment; m = 9, 10: 32 lanes per problem.  Device == twin bit
    for bit (default and tight stopping), <= 1e-6 from the reference-order solve under tight stopping."""
    import torch
    import cppnumericalsolvers_amd as amd
    B = 64
    x0 = amd.synthetic_x0_host(B, n, "u2", seed=5 * n + m)
    lo = np.full(n, -1.5) if boxed else None
    hi = np.full(n, 0.8) if boxed else None
    width = 1 << max(3, int(np.ceil(np.log2(n))))
    if m > 8:
        width = 32 if n <= 32 else 64    # 32 lanes x 1 or 2 coordinates
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8,
                             past=0)
    base = gpu_solver_factory()
    for stop_o, tol in ((oracle.lbfgsb_default_stop(), None), (tight, TOL)):
        s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=_engine_stop(stop_o), context=base.ctx)
        if boxed:
            s.SetBounds(lo, hi)
        x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
        torch.cuda.synchronize()
        assert s.last_launch()["lanes_per_problem"] == (32 if m > 8 else 16)
        x, f, g, p = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                       reduction="butterfly", width=width)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        _same_progress(p, pb)
        if tol is not None:
            xs, fs, _, _ = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                        std_sort_order=True)
            assert np.max(np.abs(x - xs)) <= tol and np.max(np.abs(f - fs)) <= tol
            assert np.all(p["status"] != 1)
    from cppnumericalsolvers_amd import capi
    with pytest.raises(capi.EngineError) as e:   # the row-per-lane algebra is built up to 2m = 20 rows
        amd.BatchedLbfgsb(arithmetic="exact", m=11, context=base.ctx).minimize(amd.Rosenbrock(), _to_dev(x0))
    assert e.value.code == capi.ERR_UNSUPPORTED


@pytest.mark.parametrize("n,kind,boxed", [(100, "std", True), (128, "u2", True), (65, "u2", False), (200, "std", True),
                                          (256, "u2", True), (129, "u2", False)])
def test_lbfgsb_up_to_256_coordinates(gpu_solver_factory, oracle, n, kind, boxed):
    """64 < n <= 128: eight coordinates per lane of the 16-lane segment; 128 < n <= 256: of a 32-lane segment.  Device ==
    twin bit for bit, <= 1e-6 from the reference-order solve under tight stopping; every point inside the box."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    B, m = 24, 5
    x0 = amd.synthetic_x0_host(B, n, kind, seed=n)
    lo = np.full(n, -1.5) if boxed else None
    hi = np.full(n, 0.8) if boxed else None
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8,
                             past=0)
    base = gpu_solver_factory()
    for stop_o in (oracle.lbfgsb_default_stop(), tight):
        s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=_engine_stop(stop_o), context=base.ctx)
        if boxed:
            s.SetBounds(lo, hi)
        x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
        torch.cuda.synchronize()
        assert s.last_launch()["elems_per_lane"] == 8 and s.last_launch()["lanes_per_problem"] == (16 if n <= 128 else 32)
        x, f, g, p = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                       reduction="butterfly", width=128 if n <= 128 else 256)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        _same_progress(p, pb)
    xs, fs, _, _ = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=tight, lower=lo, upper=hi,
                                                std_sort_order=True)
    assert np.max(np.abs(x - xs)) <= TOL and np.max(np.abs(f - fs)) <= TOL
    if boxed:
        assert np.all(x <= 0.8) and np.all(x >= -1.5)
    with pytest.raises(capi.EngineError):
        amd.BatchedLbfgsb(arithmetic="exact", m=5, context=base.ctx).minimize(amd.Rosenbrock(), _to_dev(np.zeros((2, 257))))


def test_lbfgsb_wide_layouts_edge_shapes(gpu_solver_factory, oracle):
    """The 32-lane layouts at the edges: one coordinate, one problem, a ragged batch (three problems for two slots per
    wavefront), the default (unbounded) box, an empty batch."""
    import torch
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    for n, m, B, width in ((1, 10, 1, 32), (2, 9, 3, 32), (129, 5, 1, 256), (200, 3, 3, 256), (33, 10, 5, 64)):
        x0 = amd.synthetic_x0_host(B, n, "u2", seed=7 * n + m)
        for boxed in (False, True):
            lo = np.full(n, -1.1) if boxed else None
            hi = np.full(n, 0.9) if boxed else None
            st = oracle.lbfgsb_default_stop()
            s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=_engine_stop(st), context=base.ctx)
            if boxed:
                s.SetBounds(lo, hi)
            x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(x0))
            torch.cuda.synchronize()
            assert s.last_launch()["lanes_per_problem"] == 32
            xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=st, lower=lo, upper=hi,
                                                           reduction="butterfly", width=width)
            np.testing.assert_array_equal(x.cpu().numpy(), xb)
            np.testing.assert_array_equal(f.cpu().numpy(), fb)
            np.testing.assert_array_equal(g.cpu().numpy(), gb)
            _same_progress(amd.progress_to_numpy(p), pb)
    s = amd.BatchedLbfgsb(arithmetic="exact", m=10, context=base.ctx)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(np.zeros((0, 8))))
    assert x.shape == (0, 8) and f.shape == (0,)


def test_lbfgsb_on_a_regression_objective(gpu_solver_factory, oracle):
    """The reference's linear_regression.cc: residuals (b1 + 2 b2 - 4, 3 b1 + b2 - 5), box [0, 1] x [1, 2], start
    (-1, 2) -> (1, 1.6); then random bounded least-squares problems against the twin."""
    import torch
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    A = np.array([[1.0, 2.0], [3.0, 1.0]])
    obj = amd.SquaredErrorRidge(A, 0.0)
    s = amd.BatchedLbfgsb(arithmetic="exact", m=5, context=base.ctx)
    s.SetBounds(np.array([0.0, 1.0]), np.array([1.0, 2.0]))
    x, f, g, p = s.minimize(obj, _to_dev(np.array([[-1.0, 2.0]])), per_problem=_to_dev(np.array([[4.0, 5.0]])))
    torch.cuda.synchronize()
    np.testing.assert_allclose(x.cpu().numpy()[0], [1.0, 1.6], atol=1e-5)   # "optimal solution is suppose to be [1, 1.6]"
    # random bounded regressions: n = 12 coefficients, 24 observations each, one right-hand side per problem
    rng = np.random.default_rng(11)
    rows, n, B = 24, 12, 80
    A = rng.normal(size=(rows, n))
    Y = rng.normal(size=(B, rows)) * 3.0
    x0 = rng.uniform(-1, 1, size=(B, n))
    lo, hi = np.full(n, -0.25), np.full(n, 0.4)
    obj = amd.SquaredErrorRidge(A, 0.05)
    params = oracle.ridge_params(A, 0.05)
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8,
                             past=0)
    for stop_o in (oracle.lbfgsb_default_stop(), tight):
        s = amd.BatchedLbfgsb(arithmetic="exact", m=5, stopping_progress=_engine_stop(stop_o), context=base.ctx)
        s.SetBounds(lo, hi)
        x, f, g, p = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
        torch.cuda.synchronize()
        x, f, g, p = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("squared_error_ridge", x0, m=5, stop=stop_o, lower=lo, upper=hi,
                                                       reduction="butterfly", width=16, params=params, per_problem=Y)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        _same_progress(p, pb)
    # KKT of the box-constrained least-squares problem at the returned points (tight stopping)
    assert np.all(x >= lo - 1e-15) and np.all(x <= hi + 1e-15)
    free = (x > lo + 1e-9) & (x < hi - 1e-9)
    assert np.max(np.abs(g[free])) < 1e-6
    assert np.all(g[x <= lo + 1e-9] > -1e-6) and np.all(g[x >= hi - 1e-9] < 1e-6)
    assert np.any(~free)   # some bounds are active


def _solve_and_compare(amd, oracle, base, objective, oracle_name, x0, m, lo, hi, width, lanes, linesearch="more_thuente",
                       per_problem=None, **oracle_kw):
    import torch
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8,
                             past=0)
    for stop_o in (oracle.lbfgsb_default_stop(), tight):
        s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=_engine_stop(stop_o), context=base.ctx,
                              linesearch=linesearch)
        if lo is not None:
            s.SetBounds(lo, hi)
        kw = {} if per_problem is None else {"per_problem": _to_dev(per_problem)}
        x, f, g, p = s.minimize(objective, _to_dev(x0), **kw)
        torch.cuda.synchronize()
        assert s.last_launch()["lanes_per_problem"] == lanes
        x, f, g, p = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)
        if per_problem is not None:
            oracle_kw["per_problem"] = per_problem
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch(oracle_name, x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                       reduction="butterfly", width=width, linesearch=linesearch, **oracle_kw)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        _same_progress(p, pb)
    xs, fs, _, _ = oracle.lbfgsb_minimize_batch(oracle_name, x0, m=m, stop=tight, lower=lo, upper=hi, std_sort_order=True,
                                                linesearch=linesearch, **oracle_kw)
    assert np.max(np.abs(x - xs)) <= TOL and np.max(np.abs(f - fs)) <= TOL   # vs the reference-order solve
    if lo is not None:
        assert np.all(x >= lo) and np.all(x <= hi)


@pytest.mark.parametrize("n,m,boxed", [(100, 8, True), (128, 10, True), (65, 6, False), (200, 6, True), (256, 10, True),
                                       (129, 9, False)])
def test_lbfgsb_history_sizes_up_to_ten_above_64_coordinates(gpu_solver_factory, oracle, n, m, boxed):
    """Round 3: Lbfgsb<F, m> for m = 6..10 at 64 < n <= 256 -- 32 lanes per problem, four (n <= 128) or eight coordinates
    per lane.  Device == twin bit for bit; <= 1e-6 from the reference-order solve under tight stopping."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    x0 = amd.synthetic_x0_host(20, n, "u2", seed=3 * n + m)
    lo, hi = (np.full(n, -1.5), np.full(n, 0.8)) if boxed else (None, None)
    _solve_and_compare(amd, oracle, base, amd.Rosenbrock(), "rosenbrock", x0, m, lo, hi, 128 if n <= 128 else 256, 32)


def test_lbfgsb_diag_quadratic_m10_at_200_coordinates(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    n, rng = 200, np.random.default_rng(4)
    a, c = rng.uniform(0.5, 30.0, n), 2.0
    x0 = rng.uniform(-2, 2, (12, n))
    _solve_and_compare(amd, oracle, base, amd.DiagQuadratic(a, c), "diag_quadratic", x0, 10, np.full(n, -0.5), np.full(n, 1.0),
                       256, 32, params=np.concatenate([a, [c]]))


@pytest.mark.parametrize("n,m", [(32, 7), (64, 8), (12, 6), (20, 10), (64, 9), (40, 10)])
def test_lbfgsb_hager_zhang_with_history_sizes_up_to_ten(gpu_solver_factory, oracle, n, m):
    """Round 3: Lbfgsb<F, m, HagerZhang> for m = 6..10 (n <= 64): sixteen lanes per problem up to m = 8, thirty-two above."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    x0 = amd.synthetic_x0_host(32, n, "u2", seed=11 * n + m)
    if m <= 8:
        lanes, width = 16, 16 * (1 if n <= 16 else (2 if n <= 32 else 4))
    else:
        lanes, width = 32, (32 if n <= 32 else 64)
    _solve_and_compare(amd, oracle, base, amd.Rosenbrock(), "rosenbrock", x0, m, np.full(n, -1.5), np.full(n, 0.8), width, lanes,
                       linesearch="hager_zhang")


@pytest.mark.parametrize("n,m", [(12, 8), (40, 7), (12, 10), (40, 9)])
def test_lbfgsb_on_a_regression_objective_with_history_sizes_up_to_ten(gpu_solver_factory, oracle, n, m):
    """Round 3: the bounded regression of src/examples/linear_regression.cc with m = 6..10."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    rng = np.random.default_rng(100 * n + m)
    rows, B = 2 * n, 40
    A = rng.normal(size=(rows, n))
    Y = rng.normal(size=(B, rows)) * 3.0
    x0 = rng.uniform(-1, 1, size=(B, n))
    if m <= 8:
        lanes, width = 16, 16 * (1 if n <= 16 else (2 if n <= 32 else 4))
    else:
        lanes, width = 32, (32 if n <= 32 else 64)
    _solve_and_compare(amd, oracle, base, amd.SquaredErrorRidge(A, 0.05), "squared_error_ridge", x0, m, np.full(n, -0.25),
                       np.full(n, 0.4), width, lanes, per_problem=Y, params=oracle.ridge_params(A, 0.05))


@pytest.mark.parametrize("n,m,boxed", [(100, 5, True), (65, 3, False), (128, 5, True), (200, 5, True), (256, 4, False),
                                       (100, 8, True), (128, 10, True), (70, 6, False)])
def test_lbfgsb_hager_zhang_above_64_coordinates(gpu_solver_factory, oracle, n, m, boxed):
    """Round 4: Lbfgsb<F, m, HagerZhang> for 64 < n <= 256 (the line-search template argument of lbfgsb.h:44-49 does not
    depend on the dimension): m <= 5 on sixteen lanes x eight coordinates (n <= 128) / thirty-two x eight (n <= 256),
    m = 6..10 on thirty-two x four (n <= 128).  Device == twin bit for bit; <= 1e-6 from the reference-order solve."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    x0 = amd.synthetic_x0_host(12, n, "u2", seed=5 * n + m)
    lo, hi = (np.full(n, -1.5), np.full(n, 0.8)) if boxed else (None, None)
    lanes = 16 if (m <= 5 and n <= 128) else 32
    _solve_and_compare(amd, oracle, base, amd.Rosenbrock(), "rosenbrock", x0, m, lo, hi, 128 if n <= 128 else 256, lanes,
                       linesearch="hager_zhang")


def test_lbfgsb_hager_zhang_above_64_on_a_quadratic_and_the_refused_corner(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    base = gpu_solver_factory()
    n, rng = 150, np.random.default_rng(6)
    a, c = rng.uniform(0.5, 30.0, n), 2.0
    x0 = rng.uniform(-2, 2, (10, n))
    _solve_and_compare(amd, oracle, base, amd.DiagQuadratic(a, c), "diag_quadratic", x0, 5, np.full(n, -0.5), np.full(n, 1.0),
                       256, 32, linesearch="hager_zhang", params=np.concatenate([a, [c]]))
    # m > 5 with eight coordinates per lane under Hager-Zhang would need > 1 KB of scratch per lane: refused, not shipped
    s = amd.BatchedLbfgsb(arithmetic="exact", m=8, context=base.ctx, linesearch="hager_zhang")
    with pytest.raises(capi.EngineError) as e:
        s.minimize(amd.Rosenbrock(), _to_dev(amd.synthetic_x0_host(4, 200, "u2", seed=1)))
    assert e.value.code == capi.ERR_UNSUPPORTED
