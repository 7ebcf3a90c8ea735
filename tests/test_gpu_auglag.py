"""GPU parity tests of the augmented-Lagrangian path.

mi355_auglag_* (through the C-ABI) against oracle/auglag_oracle.hpp under the butterfly reduction policy,
which performs the same IEEE operations in the same order: every comparison below is exact.  The oracle
itself is pinned bit for bit to the unmodified reference solver by tests/test_auglag_oracle.py.
"""
import numpy as np
import pytest

import auglag_lib as al

pytestmark = pytest.mark.gpu


def _engine_problem(p):
    from cppnumericalsolvers_amd import ConstrainedProblem
    mk = lambda t: ConstrainedProblem.term(t["prims"], t["form"], t["k"], product=t.get("product", False))
    return ConstrainedProblem(p.n, mk(p.terms[0]), [mk(t) for t in p.table_eq], [mk(t) for t in p.table_ineq],
                              family_equality=p.family_equality, family_inequality=p.family_inequality)


def _solver(**kw):
    from cppnumericalsolvers_amd import BatchedAugmentedLagrangian
    return BatchedAugmentedLagrangian(**kw)


LOOP = "auto"   # device execution mode used by _engine_config (the parametrised tests below switch it)


def _engine_config(solver, cfg):
    from cppnumericalsolvers_amd import capi
    c = solver.default_config()
    for name, _ in cfg._fields_:
        setattr(c, name, getattr(cfg, name))
    c.loop = capi.AL_LOOP[LOOP]
    return c


@pytest.fixture(params=["fused", "lockstep"])
def both_loops(request):
    """Runs a test once per device execution mode of the outer loop (results must be identical)."""
    global LOOP
    LOOP = request.param
    yield request.param
    LOOP = "auto"


def _padded(n):
    P = 8
    while P < n:
        P <<= 1
    return P


def _mixed_problem(n, seed):
    """Every term kind and form at once: Rosenbrock objective, two equalities, two inequalities."""
    rng = np.random.default_rng(seed)
    return al.Problem(
        n, al.term("rosenbrock"),
        [al.term("linear", "value_minus_k", 0.3, a=rng.uniform(-1, 1, n)),
         al.term("diag_quadratic", "k_minus_value", 2.0, a=rng.uniform(0.1, 1.0, n), c=0.25)],
        [al.term("squared_norm", "k_minus_value", 0.4 * n),
         al.term("linear", "plain", a=rng.uniform(0.0, 1.0, n))])


# n values cover every kernel mapping: (8,1) (8,2) (16,2) (32,2) (64,2) (64,4)
@pytest.mark.parametrize("n", [2, 7, 12, 30, 64, 100, 200])
def test_composite_matches_oracle_bitwise(n):
    p = _mixed_problem(n, seed=n)
    rng = np.random.default_rng(100 + n)
    B = 37
    x = rng.uniform(-1.5, 1.5, (B, n))
    lam = rng.uniform(-2, 2, (B, 2))
    lam[::5, 0] = 0.0                                   # MulExpression's c == 0 short circuit
    mu = rng.uniform(0, 3, (B, 2))
    pen = rng.uniform(0.1, 20.0, B)
    pen[::7] = 0.0                                      # rho = 0: inequality part skipped, penalty part zeroed
    s = _solver()
    f, g = s.evaluate_host(_engine_problem(p), x, lam, mu, pen)
    fo, go = al.oracle_eval(p, x, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)


def test_composite_golden_values_of_the_reference_tests():
    """src/test/augmented_lagrangian_test.cc:397-474: 22.5, -1.625 and 4.0."""
    half = al.term("diag_quadratic", a=[0.5, 0.5])
    x0m = lambda t: al.term("linear", "value_minus_k", t, a=[1.0, 0.0])
    s = _solver()
    f, _ = s.evaluate_host(_engine_problem(al.Problem(2, half, [x0m(1.0)])), [[3.0, 4.0]], [[2.0]], None, 3.0)
    assert abs(f[0] - 22.5) < 1e-12
    p = _engine_problem(al.Problem(2, half, [], [x0m(0.5)]))
    f, g = s.evaluate_host(p, [[3.0, 0.0]], None, [[7.0]], 4.0)
    assert abs(f[0] + 1.625) < 1e-12
    np.testing.assert_array_equal(g[0], [3.0, 0.0])
    f, g = s.evaluate_host(p, [[0.0, 0.0]], None, [[7.0]], 4.0)
    assert abs(f[0] - 4.0) < 1e-12
    np.testing.assert_allclose(g[0], [-9.0, 0.0], rtol=0, atol=1e-12)


def _assert_same(d, o):
    for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient"):
        np.testing.assert_array_equal(d[k], o[k], err_msg=k)
    for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm", "inner_iterations", "nfev", "sum_k"):
        np.testing.assert_array_equal(d["progress"][k], o["progress"][k], err_msg=k)


CASES = {
    "circle": lambda rng: (al.circle_problem(), np.vstack([[2.0, 10.0], rng.uniform(-3, 3, (63, 2))]), 1.0, {}),
    "simplex12": lambda rng: (al.quadratic_simplex_problem(12), rng.uniform(-1, 1, (64, 12)), 0.0, {}),
    "simplex40": lambda rng: (al.quadratic_simplex_problem(40, seed=3), rng.uniform(-1, 1, (48, 40)), 0.0, {}),
    "simplex100": lambda rng: (al.quadratic_simplex_problem(100, seed=4), rng.uniform(-1, 1, (24, 100)), 0.0, {}),
    "simplex200": lambda rng: (al.quadratic_simplex_problem(200, seed=5), rng.uniform(-1, 1, (12, 200)), 0.0, {}),
    "rosenbrock_ball": lambda rng: (al.rosenbrock_ball_problem(10), rng.uniform(-1, 1, (40, 10)), 0.0,
                                    {"outer_num_iterations": 12}),
    # 150 outer iterations: the penalty grows without bound (overflow, NaN deltas) and the device's ring of
    # per-iteration counters wraps twice
    "long_run": lambda rng: (al.rosenbrock_ball_problem(10), rng.uniform(-1, 1, (9, 10)), 0.0,
                             {"outer_num_iterations": 150}),
    "mixed30": lambda rng: (_mixed_problem(30, 7), rng.uniform(-1, 1, (32, 30)), 0.0, {"outer_num_iterations": 15}),
    "unconstrained": lambda rng: (al.Problem(5, al.term("rosenbrock")), rng.uniform(-1, 1, (16, 5)), 1.0, {}),
    "manual_penalty": lambda rng: (al.quadratic_simplex_problem(7, seed=5), rng.uniform(-2, 2, (33, 7)), 3.0,
                                   {"auto_scale_initial_penalty": 0, "penalty_growth_factor": 4.0,
                                    "warmup_max_inner_iterations": 0}),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_solve_matches_oracle_bitwise(case, both_loops):
    p, x0, pen0, cfg_kw = CASES[case](np.random.default_rng(11))
    cfg = al.default_config(**cfg_kw)
    s = _solver()
    s.config = _engine_config(s, cfg)
    d = s.minimize_host(_engine_problem(p), x0, penalty0=pen0)
    o = al.oracle_minimize(p, x0, penalty0=pen0, config=cfg, reduction="butterfly", width=_padded(p.n))
    _assert_same(d, o)
    if case in ("circle", "simplex12", "manual_penalty"):
        assert np.all(d["progress"]["status"] == 6) and np.all(d["max_violation"] <= 1e-5)


@pytest.mark.parametrize("n", [2, 12, 40, 200])
def test_hager_zhang_inner_solver_matches_oracle_bitwise(n, both_loops):
    """AugmentedLagrangian<Problem, Lbfgs<F, 10, HagerZhang>> and the composite objective under the same search."""
    from cppnumericalsolvers_amd import AugLagComposite, BatchedLbfgs
    rng = np.random.default_rng(300 + n)
    p = al.circle_problem() if n == 2 else al.quadratic_simplex_problem(n, seed=n)
    x0 = rng.uniform(-1, 1, (20, n))
    cfg = al.default_config(outer_num_iterations=25)
    s = _solver(linesearch="hager_zhang")
    s.config = _engine_config(s, cfg)
    d = s.minimize_host(_engine_problem(p), x0, penalty0=1.0)
    o = al.oracle_minimize(p, x0, penalty0=1.0, config=cfg, reduction="butterfly", width=_padded(n),
                           linesearch="hager_zhang")
    _assert_same(d, o)
    rows = np.hstack([rng.uniform(-1, 1, (20, 1)), rng.uniform(0, 2, (20, 1)), rng.uniform(0.5, 5.0, (20, 1))])
    lb = BatchedLbfgs(m=10, linesearch="hager_zhang")
    x, f, g, prog = lb.minimize_host(AugLagComposite(_engine_problem(p)), x0, per_problem=rows)
    xo, fo, go, po = al.oracle_composite_minimize(p, x0, rows[:, :1], rows[:, 1:2], rows[:, 2], reduction="butterfly",
                                                  width=_padded(n), linesearch="hager_zhang")
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(prog["nfev"], po["nfev"])


@pytest.mark.parametrize("n", [8, 30, 100])
def test_per_problem_term_constants(n, both_loops):
    """B different problems of one shape: each row of term_constants replaces the constants k of the terms."""
    from cppnumericalsolvers_amd import AugLagComposite, BatchedLbfgs
    p = _mixed_problem(n, seed=60 + n)
    ep = _engine_problem(p)
    rng = np.random.default_rng(n)
    B = 26
    x0 = rng.uniform(-1, 1, (B, n))
    tc = np.column_stack([np.zeros(B), rng.uniform(0.0, 0.6, B), rng.uniform(1.5, 2.5, B),
                          rng.uniform(0.3, 0.5, B) * n, np.zeros(B)])
    lam, mu, pen = rng.uniform(-1, 1, (B, 2)), rng.uniform(0, 2, (B, 2)), rng.uniform(0.5, 4.0, B)
    s = _solver()
    f, g = s.evaluate_host(ep, x0, lam, mu, pen, term_constants=tc)
    fo, go = al.oracle_eval(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n), term_constants=tc)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)
    cfg = al.default_config(outer_num_iterations=12)
    s.config = _engine_config(s, cfg)
    d = s.minimize_host(ep, x0, term_constants=tc)
    o = al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=_padded(n), term_constants=tc)
    _assert_same(d, o)
    assert not np.array_equal(d["x"], s.minimize_host(ep, x0)["x"])
    # the composite objective takes the constants appended to its per-problem rows
    rows = np.hstack([lam, mu, pen[:, None], tc])
    x, fv, _, prog = BatchedLbfgs(m=10).minimize_host(AugLagComposite(ep), x0, per_problem=rows)
    f2, _ = s.evaluate_host(ep, x, lam, mu, pen, term_constants=tc)
    np.testing.assert_array_equal(fv, f2)


@pytest.mark.parametrize("n", [6, 12, 24, 48])          # E = 1, 1, 2, 4 of the sixteen-lane L-BFGS-B kernel
@pytest.mark.parametrize("bounds", ["set", "never_set"])
def test_lbfgsb_inner_solver_matches_oracle_bitwise(n, bounds, both_loops):
    """mi355_auglag_box_minimize_batch: AugmentedLagrangian<Problem, Lbfgsb<F, m>>."""
    p, lower, upper = al.boxed_rosenbrock_problem(n)
    x0 = np.random.default_rng(n).uniform(-1, 1, (21, n))
    cfg = al.default_config(outer_num_iterations=20)
    kw = dict(lower=lower, upper=upper) if bounds == "set" else {}
    for m, ls in ((5, "more_thuente"), (3, "hager_zhang")):
        s = _solver(inner="lbfgsb", m=m, linesearch=ls, **kw)
        s.config = _engine_config(s, cfg)
        d = s.minimize_host(_engine_problem(p), x0)
        o = al.oracle_box_minimize(p, x0, config=cfg, m=m, linesearch=ls, reduction="butterfly", width=_padded(n),
                                   std_sort_order=False, **kw)
        _assert_same(d, o)
        if bounds == "set":
            assert np.all(d["x"] >= lower) and np.all(d["x"] <= upper)


@pytest.mark.parametrize("n", [7, 40, 130])
def test_summed_terms_match_oracle_bitwise(n, both_loops):
    """Terms that are sums of two and three primitives (AddExpression), every form."""
    from cppnumericalsolvers_amd import AugLagComposite, BatchedLbfgs
    p = al.three_part_problem(n)
    ep = _engine_problem(p)
    rng = np.random.default_rng(n)
    B = 19
    x0 = rng.uniform(-1, 1, (B, n))
    lam, mu, pen = rng.uniform(-1, 1, (B, 1)), rng.uniform(0, 2, (B, 2)), rng.uniform(0.5, 4.0, B)
    s = _solver()
    f, g = s.evaluate_host(ep, x0, lam, mu, pen)
    fo, go = al.oracle_eval(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)
    cfg = al.default_config(outer_num_iterations=15)
    s.config = _engine_config(s, cfg)
    _assert_same(s.minimize_host(ep, x0), al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=_padded(n)))
    rows = np.hstack([lam, mu, pen[:, None]])
    x, fv, _, _ = BatchedLbfgs(m=10).minimize_host(AugLagComposite(ep), x0, per_problem=rows)
    xo, fo2, _, _ = al.oracle_composite_minimize(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_array_equal(fv, fo2)


@pytest.mark.parametrize("n", [2, 9, 40, 130])
def test_product_terms_match_oracle_bitwise(n, both_loops):
    """ProdExpression (function_expressions.h:260-315) as a node of the term table: value a b, gradient b grad a + a grad b.
    Composite values / gradients and whole constrained solves equal the oracle bit for bit; at n = 2 the problem is Hs029
    written over the menu and reaches the reference test's optimum."""
    from cppnumericalsolvers_amd import AugLagComposite, BatchedLbfgs
    p = al.hs029_product_problem() if n == 2 else al.product_terms_problem(n)
    ep = _engine_problem(p)
    rng = np.random.default_rng(n)
    B = 13
    x0 = rng.uniform(0.1, 1.0, (B, n))
    lam, mu = rng.uniform(-1, 1, (B, p.n_eq)), rng.uniform(0, 2, (B, p.n_ineq))
    pen = rng.uniform(0.5, 4.0, B)
    s = _solver()
    f, g = s.evaluate_host(ep, x0, lam, mu, pen)
    fo, go = al.oracle_eval(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)
    cfg = al.default_config(outer_num_iterations=15 if n > 2 else 10000)
    s.config = _engine_config(s, cfg)
    d = s.minimize_host(ep, x0)
    _assert_same(d, al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=_padded(n)))
    if n == 2:
        d1 = s.minimize_host(ep, np.array([[1.0, 1.0]]))
        np.testing.assert_allclose(d1["x"][0], [2.0 * np.sqrt(6.0), 2.0 * np.sqrt(3.0)], atol=1e-3)
    rows = np.hstack([lam, mu, pen[:, None]])
    x, fv, _, _ = BatchedLbfgs(m=10).minimize_host(AugLagComposite(ep), x0, per_problem=rows)
    xo, fo2, _, _ = al.oracle_composite_minimize(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_array_equal(fv, fo2)


@pytest.mark.parametrize("n", [5, 11, 27, 50, 90, 170])   # one per kernel mapping
def test_random_term_tables_match_oracle_bitwise(n, both_loops):
    """Every kind in every position of one-, two- and three-part terms: composite values and gradients, and three
    outer iterations including the KKT norm, against the oracle."""
    rng = np.random.default_rng(7000 + n)
    s = _solver()
    cfg = al.default_config(outer_num_iterations=3)
    s.config = _engine_config(s, cfg)
    for trial in range(10):
        p = al.random_problem(n, rng)
        ep = _engine_problem(p)
        B = 9
        x0 = rng.uniform(-1, 1, (B, n))
        lam = rng.uniform(-1, 1, (B, max(p.n_eq, 1)))
        mu = rng.uniform(0, 2, (B, max(p.n_ineq, 1)))
        pen = rng.uniform(0.5, 4.0, B)
        f, g = s.evaluate_host(ep, x0, lam[:, :p.n_eq], mu[:, :p.n_ineq], pen)
        fo, go = al.oracle_eval(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n))
        np.testing.assert_array_equal(f, fo, err_msg="trial %d" % trial)
        np.testing.assert_array_equal(g, go, err_msg="trial %d" % trial)
        pen0 = 0.0 if trial % 2 else 1.0      # 0: the auto-scaled initial penalty (values of every term at x0)
        d = s.minimize_host(ep, x0, penalty0=pen0)
        o = al.oracle_minimize(p, x0, penalty0=pen0, config=cfg, reduction="butterfly", width=_padded(n))
        _assert_same(d, o)


@pytest.mark.parametrize("shape", [(2, 2), (9, 5), (33, 11), (64, 14), (100, 7)])
def test_squared_affine_terms_match_oracle_bitwise(shape, both_loops):
    """MI355_AL_TERM_SQUARED_AFFINE (ABI 9): a least-squares objective as a sum of (a.x - c)^2 primitives — composite
    values / gradients and whole solves against the oracle bit for bit, and against the reference binary within 1e-6."""
    n, rows = shape
    p = al.linear_regression_problem() if n == 2 else al.least_squares_problem(n, rows)
    ep = _engine_problem(p)
    rng = np.random.default_rng(400 + n)
    B = 11
    x0 = rng.uniform(-1, 1, (B, n))
    if n == 2:
        x0[0] = [-1.0, 2.0]                         # the start of src/examples/linear_regression.cc:65-66
    lam, mu, pen = rng.uniform(-1, 1, (B, max(p.n_eq, 1))), rng.uniform(0, 2, (B, max(p.n_ineq, 1))), rng.uniform(0.5, 4.0, B)
    s = _solver()
    f, g = s.evaluate_host(ep, x0, lam[:, :p.n_eq], mu[:, :p.n_ineq], pen)
    fo, go = al.oracle_eval(p, x0, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)
    cfg = al.default_config(outer_num_iterations=25)
    s.config = _engine_config(s, cfg)
    d = s.minimize_host(ep, x0, penalty0=1.0)
    _assert_same(d, al.oracle_minimize(p, x0, penalty0=1.0, config=cfg, reduction="butterfly", width=_padded(n)))
    if n == 2:
        np.testing.assert_allclose(d["x"][0], [1.0, 1.6], atol=1e-4)
    import ref_lib
    if ref_lib.available():
        r = al.ref_minimize(p, x0, penalty0=1.0, config=cfg)
        same = d["progress"]["num_iterations"] == r["progress"]["num_iterations"]   # (a different summation order may
        assert same.mean() >= 0.8                                                   #  move a stop by one outer step)
        np.testing.assert_allclose(d["x"][same], r["x"][same], atol=1e-6)


def test_reference_test_problems_on_the_device():
    """BothEqualityAndInequalityActive (:583-621) and BoxPinnedOptimumStopsOnKkt (:1198-1275) over the menu."""
    q = al.quadratic_at_12_problem()
    d = _solver().minimize_host(_engine_problem(q), [[1.0, 1.0]], penalty0=1.0)
    _assert_same(d, al.oracle_minimize(q, [[1.0, 1.0]], penalty0=1.0, reduction="butterfly", width=8))
    np.testing.assert_allclose(d["x"][0], [0.5, 1.5], atol=1e-3)
    p, lower, upper = al.hs016_problem()
    s = _solver(inner="lbfgsb", lower=lower, upper=upper)
    d = s.minimize_host(_engine_problem(p), [[-2.0, 1.0]])
    s.config.loop = 2                                   # ... and under the lock-step form of the loop
    _assert_same(s.minimize_host(_engine_problem(p), [[-2.0, 1.0]]), d)
    _assert_same(d, al.oracle_box_minimize(p, [[-2.0, 1.0]], lower=lower, upper=upper, reduction="butterfly", width=8,
                                           std_sort_order=False))
    assert d["progress"]["status"][0] == 6 and d["progress"]["num_iterations"][0] < 20
    np.testing.assert_allclose(d["x"][0], [0.5, 0.25], atol=1e-4)


def test_history_size_and_initial_multipliers(both_loops):
    p = al.quadratic_simplex_problem(20, seed=8)
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1, 1, (40, 20))
    lam0 = rng.uniform(-1, 1, (40, 1))
    mu0 = rng.uniform(0, 2, (40, 1))
    for m in (3, 5, 10):
        s = _solver(m=m)
        s.config = _engine_config(s, al.default_config())
        d = s.minimize_host(_engine_problem(p), x0, lam0, mu0, 2.0)
        o = al.oracle_minimize(p, x0, lam0, mu0, 2.0, m=m, reduction="butterfly", width=32)
        _assert_same(d, o)


@pytest.mark.parametrize("n", [6, 30, 200])
def test_composite_as_an_lbfgs_objective(n):
    """MI355_OBJ_AL_COMPOSITE through mi355_lbfgs_minimize_batch_host == oracle Lbfgs on the composite, bit for bit."""
    from cppnumericalsolvers_amd import AugLagComposite, BatchedLbfgs
    p = _mixed_problem(n, seed=40 + n)
    rng = np.random.default_rng(n)
    B = 24
    x0 = rng.uniform(-1, 1, (B, n))
    rows = np.hstack([rng.uniform(-1, 1, (B, 2)), rng.uniform(0, 2, (B, 2)), rng.uniform(0.5, 5.0, (B, 1))])
    s = BatchedLbfgs(m=10)
    x, f, g, prog = s.minimize_host(AugLagComposite(_engine_problem(p)), x0, per_problem=rows)
    xo, fo, go, po = al.oracle_composite_minimize(p, x0, rows[:, :2], rows[:, 2:4], rows[:, 4], reduction="butterfly",
                                                  width=_padded(n))
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)
    for k in ("status", "num_iterations", "nfev"):
        np.testing.assert_array_equal(prog[k], po[k], err_msg=k)


def test_verify_cc_circle_on_the_device():
    """src/test/verify.cc:290-312: expects (-1, -1) within 1e-3."""
    d = _solver().minimize_host(_engine_problem(al.circle_problem()), [[2.0, 10.0]], penalty0=1.0)
    np.testing.assert_allclose(d["x"][0], [-1.0, -1.0], atol=1e-3)


def test_device_tensor_entry_matches_host_entry():
    import torch
    p = al.quadratic_simplex_problem(12)
    ep = _engine_problem(p)
    x0 = np.random.default_rng(2).uniform(-1, 1, (500, 12))
    s = _solver()
    h = s.minimize_host(ep, x0)
    x = torch.from_numpy(x0).to("cuda:0")
    lam = torch.zeros(500, 1, dtype=torch.float64, device="cuda:0")
    mu = torch.zeros(500, 1, dtype=torch.float64, device="cuda:0")
    pen = torch.zeros(500, dtype=torch.float64, device="cuda:0")
    viol, kkt, prog = s.minimize(ep, x, lam, mu, pen)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(x.cpu().numpy(), h["x"])
    np.testing.assert_array_equal(lam.cpu().numpy(), h["lambda"])
    np.testing.assert_array_equal(mu.cpu().numpy(), h["mu"])
    np.testing.assert_array_equal(pen.cpu().numpy(), h["penalty"])
    np.testing.assert_array_equal(viol.cpu().numpy(), h["max_violation"])
    from cppnumericalsolvers_amd import capi
    pr = prog.cpu().numpy().view(capi.AL_PROGRESS_DTYPE)
    np.testing.assert_array_equal(pr["num_iterations"], h["progress"]["num_iterations"])


def test_device_entry_with_term_constants_and_sharded_driver():
    """Device tensors end to end (asynchronous fused loop), per-problem constants included, through the sharded
    driver with a single rank."""
    import torch
    from cppnumericalsolvers_amd import capi, sharded
    n, B = 40, 300
    p = al.quadratic_simplex_problem(n, seed=21)
    ep = _engine_problem(p)
    rng = np.random.default_rng(22)
    x0 = rng.uniform(-1, 1, (B, n))
    tc = np.column_stack([np.zeros(B), rng.uniform(0.5, 2.0, B), rng.uniform(0.05, 0.5, B)])
    cfg = al.default_config(outer_num_iterations=15)
    s = _solver()
    s.config = _engine_config(s, cfg)
    h = s.minimize_host(ep, x0, term_constants=tc)
    dev = torch.device("cuda:0")

    def make_state(first, count):
        sl = slice(first, first + count)
        return (torch.from_numpy(x0[sl]).to(dev), torch.zeros(count, 1, dtype=torch.float64, device=dev),
                torch.zeros(count, 1, dtype=torch.float64, device=dev), torch.zeros(count, dtype=torch.float64, device=dev))

    class WithConstants:
        """solver facade that adds this shard's constants (the sharded driver passes the state only)"""
        def minimize(self, problem, x, lam, mu, pen):
            return s.minimize(problem, x, lam, mu, pen, term_constants=torch.from_numpy(tc).to(dev))

    (lo, hi), (x, lam, mu, pen, viol, kkt, prog), flag = sharded.ShardedAugmentedLagrangian(WithConstants()).minimize_global(
        ep, B, make_state)
    torch.cuda.synchronize()
    assert (lo, hi) == (0, B) and flag.total == B
    np.testing.assert_array_equal(x.cpu().numpy(), h["x"])
    np.testing.assert_array_equal(lam.cpu().numpy(), h["lambda"])
    np.testing.assert_array_equal(kkt.cpu().numpy(), h["max_lagrangian_gradient"])
    pr = prog.cpu().numpy().view(capi.AL_PROGRESS_DTYPE)
    np.testing.assert_array_equal(pr["status"], h["progress"]["status"])
    assert flag.unconverged == int((pr["status"] <= 1).sum()) and flag.iterations == int(pr["num_iterations"].sum())
    o = al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=64, term_constants=tc)
    _assert_same(h, o)


def test_empty_batch_single_problem_and_maximum_table(both_loops):
    """B = 0, B = 1, and the largest problem the menu takes: n = 256 with four equalities, four inequalities and
    sixteen primitives."""
    from cppnumericalsolvers_amd import capi
    p = al.quadratic_simplex_problem(5)
    s = _solver()
    s.config = _engine_config(s, al.default_config(outer_num_iterations=10))
    empty = s.minimize_host(_engine_problem(p), np.zeros((0, 5)))
    assert empty["x"].shape == (0, 5) and empty["progress"].shape == (0,)
    one = s.minimize_host(_engine_problem(p), [[0.3, -0.2, 0.1, 0.5, -0.4]])
    _assert_same(one, al.oracle_minimize(p, [[0.3, -0.2, 0.1, 0.5, -0.4]], config=al.default_config(outer_num_iterations=10),
                                         reduction="butterfly", width=8))
    n = 256
    rng = np.random.default_rng(256)
    lin = lambda: ("linear", rng.uniform(-1, 1, n))
    quad = lambda: ("diag_quadratic", rng.uniform(0.01, 0.1, n), float(rng.uniform(-0.1, 0.1)))
    big = al.Problem(
        n, al.term([("rosenbrock",), quad(), lin()]),
        [al.term([lin(), ("squared_norm",)], "value_minus_k", 0.1 * n), al.term([lin()], "value_minus_k", 0.3),
         al.term([quad(), lin()], "k_minus_value", 1.0), al.term([lin(), lin()], "plain")],
        [al.term([("squared_norm",)], "k_minus_value", 0.5 * n), al.term([lin(), quad()], "plain"),
         al.term([lin(), ("squared_norm",)], "k_minus_value", 2.0 * n), al.term([quad()], "value_minus_k", -3.0)])
    assert len(big.kinds) == capi.AL_MAX_ROWS and big.n_eq == big.n_ineq == capi.AL_MAX_CONSTRAINTS
    x0 = rng.uniform(-0.5, 0.5, (5, n))
    cfg = al.default_config(outer_num_iterations=4)
    s.config = _engine_config(s, cfg)
    _assert_same(s.minimize_host(_engine_problem(big), x0),
                 al.oracle_minimize(big, x0, config=cfg, reduction="butterfly", width=256))
    with pytest.raises(ValueError):                      # one primitive too many for the table
        _engine_problem(al.Problem(n, al.term([("rosenbrock",), quad(), lin(), lin()]), big.terms[1:5], big.terms[5:]))


@pytest.mark.parametrize("case", ["nonfinite_start", "clamped_multipliers", "kkt_test_disabled", "loose_feasibility"])
def test_edge_configurations_match_oracle_bitwise(case, both_loops):
    """NaN / inf / overflowing start points, multipliers pinned at multiplier_max, the stationarity test switched off
    (the oracle's behaviour on these is pinned to the reference in tests/test_auglag_oracle.py)."""
    from test_auglag_oracle import _edge_cases
    p = al.quadratic_simplex_problem(6)
    x0, cfg = _edge_cases()[case]
    s = _solver()
    s.config = _engine_config(s, cfg)
    _assert_same(s.minimize_host(_engine_problem(p), x0), al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=8))


def test_weird_initial_states_match_oracle_bitwise(both_loops):
    """negative / tiny / huge / infinite / NaN penalties and multipliers handed in by the caller"""
    from test_auglag_oracle import _weird_initial_states
    p = al.quadratic_simplex_problem(6)
    x0, lam, mu, pen = _weird_initial_states()
    cfg = al.default_config(outer_num_iterations=6)
    s = _solver()
    s.config = _engine_config(s, cfg)
    _assert_same(s.minimize_host(_engine_problem(p), x0, lam, mu, pen),
                 al.oracle_minimize(p, x0, lam, mu, pen, config=cfg, reduction="butterfly", width=8))


@pytest.mark.parametrize("name", ["circle", "simplex12", "simplex40_hz", "quadratic_at_12", "hs016_box", "boxed_rosenbrock6"])
def test_device_against_the_reference_golden_vectors(name):
    """The committed outputs of the reference solver itself (tests/golden/auglag_reference_vectors.npz): same status
    and outer-iteration count, x and multipliers within the tolerances of the reference's own tests (1e-3 primal,
    1e-2 dual — the loop stops at a KKT norm of 1e-4; the device sums in another order)."""
    from test_auglag_oracle import _golden
    cases, gold = _golden()
    p, x0, pen0, cfg_kw, inner, bounds, ls = cases[name]
    kw = dict(inner="lbfgsb", lower=bounds[0], upper=bounds[1]) if inner == "lbfgsb" else {}
    s = _solver(linesearch=ls, **kw)
    s.config = _engine_config(s, al.default_config(**cfg_kw))
    d = s.minimize_host(_engine_problem(p), x0, penalty0=pen0)
    np.testing.assert_array_equal(d["progress"]["status"], gold[name + "/status"])
    np.testing.assert_allclose(d["x"], gold[name + "/x"], rtol=0, atol=1e-3)
    np.testing.assert_allclose(d["lambda"], gold[name + "/lambda"], rtol=0, atol=1e-2)
    np.testing.assert_allclose(d["mu"], gold[name + "/mu"], rtol=0, atol=1e-2)
    assert np.all(d["max_violation"] <= 1e-5)


def test_large_batch_properties():
    """65536 constrained problems (n = 12) and 16384 (n = 64): every finished problem is feasible to the threshold,
    every returned point (finished or not) is the best iterate seen — near-feasible here —, the multipliers have the
    right signs, and a strided sample equals the oracle bit for bit."""
    import torch
    from cppnumericalsolvers_amd import capi
    for n, B, stride in ((12, 65536, 512), (64, 16384, 256)):
        p = al.quadratic_simplex_problem(n, seed=3)
        ep = _engine_problem(p)
        rng = np.random.default_rng(20260923)
        x0 = rng.uniform(-1, 1, (B, n))
        cfg = al.default_config(outer_num_iterations=40)
        s = _solver()
        s.config = _engine_config(s, cfg)
        dev = torch.device("cuda:0")
        x = torch.from_numpy(x0).to(dev)
        lam = torch.zeros(B, 1, dtype=torch.float64, device=dev)
        mu = torch.zeros(B, 1, dtype=torch.float64, device=dev)
        pen = torch.zeros(B, dtype=torch.float64, device=dev)
        viol, kkt, prog = s.minimize(ep, x, lam, mu, pen)
        torch.cuda.synchronize()
        pr = prog.cpu().numpy().view(capi.AL_PROGRESS_DTYPE)
        xh, violh, kkth = x.cpu().numpy(), viol.cpu().numpy(), kkt.cpu().numpy()
        assert np.all(np.isfinite(xh)) and set(np.unique(pr["status"])) <= {1, 6}
        fin = pr["status"] == 6
        assert fin.mean() > 0.97
        # (the returned state is the best iterate of the filter — feasible, lowest objective — not necessarily the one
        #  that met the 1e-4 stationarity threshold and ended the loop, exactly as in the reference)
        assert np.all(violh[fin] <= 1e-5) and np.all(np.isfinite(kkth)) and np.median(kkth[fin]) <= 1e-4
        assert np.all(np.abs(xh[fin].sum(axis=1) - 1.0) <= 1e-5) and np.all(xh[fin, 0] <= 0.2 + 1e-5)
        assert np.all(violh <= 1e-4)                      # the best-iterate filter hands back a near-feasible point
        assert np.all(mu.cpu().numpy() >= 0.0) and np.all(pen.cpu().numpy() > 0.0)
        assert np.all(pr["num_iterations"] <= 41)
        idx = np.arange(0, B, stride)
        o = al.oracle_minimize(p, x0[idx], config=cfg, reduction="butterfly", width=_padded(n))
        np.testing.assert_array_equal(xh[idx], o["x"])
        np.testing.assert_array_equal(lam.cpu().numpy()[idx], o["lambda"])
        np.testing.assert_array_equal(kkth[idx], o["max_lagrangian_gradient"])
        np.testing.assert_array_equal(pr["status"][idx], o["progress"]["status"])
        np.testing.assert_array_equal(pr["inner_iterations"][idx], o["progress"]["inner_iterations"])


def test_invalid_arguments_fail_loudly():
    from cppnumericalsolvers_amd import capi
    s = _solver(m=11)
    with pytest.raises(capi.EngineError):
        s.minimize_host(_engine_problem(al.circle_problem()), [[1.0, 1.0]], penalty0=1.0)
    bad = _solver()
    bad.config.loop = 7
    with pytest.raises(capi.EngineError):
        bad.minimize_host(_engine_problem(al.circle_problem()), [[1.0, 1.0]], penalty0=1.0)
    with pytest.raises(capi.EngineError):            # the Lbfgsb inner solver is built for n <= 128
        _solver(inner="lbfgsb").minimize_host(_engine_problem(al.rosenbrock_ball_problem(130)), np.zeros((2, 130)),
                                              penalty0=1.0)


def test_restart_from_a_returned_state_matches_oracle(both_loops):
    """max_violation is in/out: a returned state fed back into the solver continues exactly like the twin's (and, on
    the CPU, the reference's: tests/test_auglag_oracle.py::test_restart_from_a_returned_state_matches_reference)."""
    rng = np.random.default_rng(21)
    p = al.rosenbrock_ball_problem(8)
    x0 = rng.uniform(-1, 1, (12, 8))
    s = _solver()
    cfg1, cfg2 = al.default_config(outer_num_iterations=3), al.default_config(outer_num_iterations=4)
    s.config = _engine_config(s, cfg1)
    d1 = s.minimize_host(_engine_problem(p), x0)
    o1 = al.oracle_minimize(p, x0, config=cfg1, reduction="butterfly", width=_padded(p.n))
    _assert_same(d1, o1)
    s.config = _engine_config(s, cfg2)
    d2 = s.minimize_host(_engine_problem(p), d1["x"], lambda0=d1["lambda"], mu0=d1["mu"], penalty0=d1["penalty"],
                         max_violation0=d1["max_violation"])
    o2 = al.oracle_minimize(p, o1["x"], lambda0=o1["lambda"], mu0=o1["mu"], penalty0=o1["penalty"], config=cfg2,
                            max_violation0=o1["max_violation"], reduction="butterfly", width=_padded(p.n))
    _assert_same(d2, o2)
