"""Augmented-Lagrangian oracle (oracle/auglag_oracle.hpp) against the reference.

CPU only.  The oracle is pinned (a) bit for bit against the unmodified reference solver built over
oracle/eigen_shim (oracle/_ref/libref.so, ref_auglag_capi.cpp) and (b) against the closed-form
values and KKT expectations of the reference's own tests (src/test/augmented_lagrangian_test.cc,
src/test/verify.cc:290-312), restated over the device engine's term menu.
"""
import numpy as np
import pytest

import auglag_lib as al
import ref_lib

needs_ref = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libref.so not available")

HALF_SQUARED_NORM = al.term("diag_quadratic", a=[0.5, 0.5])          # HalfSquaredNorm2D (:85-92)


def x0_minus(target, form="value_minus_k"):                           # X0MinusTarget (:118-130)
    return al.term("linear", form, target, a=[1.0, 0.0])


# Section B of the reference test: composite assembly ------------------------------------------------
def test_composite_equality_only_matches_closed_form():              # :397-414, expects 22.5
    p = al.Problem(2, HALF_SQUARED_NORM, [x0_minus(1.0)])
    f, _ = al.oracle_eval(p, [[3.0, 4.0]], [[2.0]], None, 3.0)
    assert abs(f[0] - 22.5) < 1e-12


def test_composite_phr_inactive_side():                               # :431-446, expects -1.625
    p = al.Problem(2, HALF_SQUARED_NORM, [], [x0_minus(0.5)])
    f, g = al.oracle_eval(p, [[3.0, 0.0]], None, [[7.0]], 4.0)
    assert abs(f[0] + 1.625) < 1e-12
    np.testing.assert_array_equal(g[0], [3.0, 0.0])                   # constant on the inactive side


def test_composite_phr_active_side():                                 # :459-474, expects 4.0
    p = al.Problem(2, HALF_SQUARED_NORM, [], [x0_minus(0.5)])
    f, g = al.oracle_eval(p, [[0.0, 0.0]], None, [[7.0]], 4.0)
    assert abs(f[0] - 4.0) < 1e-12
    # d/dx0 of (1/(2 rho)) max(0, mu - rho (x0 - 0.5))^2 = -(mu - rho (x0 - 0.5)) = -9
    np.testing.assert_allclose(g[0], [-9.0, 0.0], rtol=0, atol=1e-12)


def test_composite_zero_multiplier_short_circuit():
    """MulExpression with c == 0 returns exact zeros (function_expressions.h:205-213), even for a NaN term."""
    p = al.Problem(2, al.term("linear", a=[1.0, 1.0]), [al.term("squared_norm", "value_minus_k", 1.0)])
    f, g = al.oracle_eval(p, [[1e200, 1e200]], [[0.0]], None, 0.0)     # |x|^2 overflows to inf
    assert np.isfinite(f[0]) and np.all(np.isfinite(g[0]))


def test_composite_gradient_matches_finite_differences():
    p = al.rosenbrock_ball_problem(6)
    rng = np.random.default_rng(3)
    x = rng.uniform(-1.2, 1.2, (1, 6))
    lam, mu, rho = [[0.7]], [[1.3]], 2.5
    f, g = al.oracle_eval(p, x, lam, mu, rho)
    for i in range(6):
        h = 1e-6
        xp, xm = x.copy(), x.copy()
        xp[0, i] += h
        xm[0, i] -= h
        fd = (al.oracle_eval(p, xp, lam, mu, rho)[0][0] - al.oracle_eval(p, xm, lam, mu, rho)[0][0]) / (2 * h)
        assert abs(fd - g[0, i]) < 1e-5 * max(1.0, abs(g[0, i]))


# Section C: outer-loop KKT expectations -------------------------------------------------------------
def test_kkt_equality_only_quadratic():                                # :492-516
    p = al.Problem(2, HALF_SQUARED_NORM, [x0_minus(1.0)])
    r = al.oracle_minimize(p, [[5.0, 5.0]], penalty0=1.0)
    assert abs(r["x"][0, 0] - 1.0) <= 1e-5 and abs(r["x"][0, 1]) < 1e-3
    assert abs(r["lambda"][0, 0] + 1.0) < 1e-2
    assert r["progress"]["status"][0] == 6


def test_kkt_inequality_active_recovers_multiplier():                  # after :541-575 (x0 >= 1 active)
    p = al.Problem(2, HALF_SQUARED_NORM, [], [x0_minus(1.0)])
    r = al.oracle_minimize(p, [[5.0, 5.0]], penalty0=1.0)
    assert abs(r["x"][0, 0] - 1.0) < 1e-3 and abs(r["x"][0, 1]) < 1e-3
    assert r["x"][0, 0] - 1.0 >= -1e-5
    assert abs(r["mu"][0, 0] - 1.0) < 1e-2


def test_feasible_start_converges_immediately():                       # :627-650
    p = al.Problem(2, HALF_SQUARED_NORM, [al.term("linear", a=[0.0, 0.0])])
    r = al.oracle_minimize(p, [[0.0, 0.0]], penalty0=1.0)
    assert r["progress"]["status"][0] == 6 and r["progress"]["num_iterations"][0] <= 5
    np.testing.assert_allclose(r["x"][0], 0.0, atol=1e-3)


def test_verify_cc_circle_problem():                                   # src/test/verify.cc:290-312
    r = al.oracle_minimize(al.circle_problem(), [[2.0, 10.0]], penalty0=1.0)
    np.testing.assert_allclose(r["x"][0], [-1.0, -1.0], atol=1e-3)


def test_penalty_auto_scaling_and_growth():
    """penalty 0 -> auto-scaled on the first outer iteration (augmented_lagrangian.h, ComputeAutoScaledPenalty);
    it then only grows while the violation does not shrink by violation_shrink_ratio."""
    p = al.quadratic_simplex_problem(6)
    x0 = np.random.default_rng(0).uniform(-1, 1, (3, 6))
    r = al.oracle_minimize(p, x0, penalty0=0.0)
    assert np.all(r["penalty"] > 0) and np.all(r["max_violation"] <= 1e-5)
    np.testing.assert_allclose(r["x"].sum(axis=1), 1.0, atol=1e-5)
    assert np.all(r["x"][:, 0] <= 0.2 + 1e-5)
    fixed = al.oracle_minimize(p, x0, penalty0=5.0, config=al.default_config(penalty_growth_factor=1.0))
    np.testing.assert_array_equal(fixed["penalty"], 5.0)


# Pinning against the unmodified reference solver ---------------------------------------------------
def _assert_same(o, r):
    for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient"):
        np.testing.assert_array_equal(o[k], r[k], err_msg=k)
    for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(o["progress"][k], r["progress"][k], err_msg=k)


@needs_ref
@pytest.mark.parametrize("case", ["circle", "simplex", "rosenbrock_ball", "unconstrained", "manual_penalty"])
def test_oracle_is_bit_identical_to_reference(case):
    rng = np.random.default_rng(11)
    cfg = al.default_config()
    pen0 = 0.0
    if case == "circle":
        p, x0, pen0 = al.circle_problem(), np.vstack([[2.0, 10.0], rng.uniform(-3, 3, (7, 2))]), 1.0
    elif case == "simplex":
        p, x0 = al.quadratic_simplex_problem(12), rng.uniform(-1, 1, (8, 12))
    elif case == "rosenbrock_ball":
        # the default thresholds are never met on this problem (the penalty grows without bound): cap the loop
        p, x0, cfg = al.rosenbrock_ball_problem(10), rng.uniform(-1, 1, (6, 10)), al.default_config(outer_num_iterations=12)
    elif case == "unconstrained":                                       # NoConstraintsIsUnconstrained (:661-690)
        p, x0, pen0 = al.Problem(5, al.term("rosenbrock")), rng.uniform(-1, 1, (4, 5)), 1.0
    else:
        p, x0, pen0 = al.quadratic_simplex_problem(7, seed=5), rng.uniform(-2, 2, (6, 7)), 3.0
        cfg = al.default_config(auto_scale_initial_penalty=0, penalty_growth_factor=4.0, warmup_max_inner_iterations=0)
    o = al.oracle_minimize(p, x0, penalty0=pen0, config=cfg)
    r = al.ref_minimize(p, x0, penalty0=pen0, config=cfg)
    _assert_same(o, r)


@needs_ref
@pytest.mark.parametrize("case", ["circle", "simplex"])
def test_oracle_is_bit_identical_to_reference_with_hager_zhang_inner_solver(case):
    """AugmentedLagrangian<Problem, Lbfgs<FunctionExpr, 10, HagerZhang>>."""
    rng = np.random.default_rng(21)
    if case == "circle":
        p, x0, pen0 = al.circle_problem(), rng.uniform(-3, 3, (6, 2)), 1.0
    else:
        p, x0, pen0 = al.quadratic_simplex_problem(9, seed=6), rng.uniform(-1, 1, (6, 9)), 0.0
    cfg = al.default_config(outer_num_iterations=25)
    o = al.oracle_minimize(p, x0, penalty0=pen0, config=cfg, linesearch="hager_zhang")
    r = al.ref_minimize(p, x0, penalty0=pen0, config=cfg, linesearch="hager_zhang")
    _assert_same(o, r)
    mt = al.oracle_minimize(p, x0, penalty0=pen0, config=cfg)
    assert not np.array_equal(o["x"], mt["x"])   # the line search does change the trajectory


@needs_ref
def test_oracle_matches_reference_with_per_problem_constants():
    """Every problem of the batch has its own right-hand sides: sum x = s_b, x_0 <= u_b."""
    p = al.quadratic_simplex_problem(8, seed=12)
    rng = np.random.default_rng(13)
    B = 10
    x0 = rng.uniform(-1, 1, (B, 8))
    tc = np.column_stack([np.zeros(B), rng.uniform(0.5, 2.0, B), rng.uniform(0.05, 0.5, B)])
    o = al.oracle_minimize(p, x0, term_constants=tc)
    r = al.ref_minimize(p, x0, term_constants=tc)
    _assert_same(o, r)
    fin = o["progress"]["status"] == 6
    assert fin.sum() >= B // 2
    np.testing.assert_allclose(o["x"][fin].sum(axis=1), tc[fin, 1], atol=1e-5)
    assert np.all(o["x"][fin, 0] <= tc[fin, 2] + 1e-5)
    shared = al.oracle_minimize(p, x0)
    assert not np.array_equal(shared["x"], o["x"])


def _edge_cases():
    rng = np.random.default_rng(5)
    x0 = rng.uniform(-1, 1, (6, 6))
    bad = x0.copy()
    bad[1, 2], bad[3, 0], bad[4, :] = np.nan, np.inf, 1e200
    return {"nonfinite_start": (bad, al.default_config(outer_num_iterations=6)),
            "clamped_multipliers": (x0, al.default_config(outer_num_iterations=12, multiplier_max=0.05)),
            "kkt_test_disabled": (x0, al.default_config(outer_num_iterations=12, kkt_stationarity_threshold=0.0)),
            "loose_feasibility": (x0, al.default_config(outer_num_iterations=8, constraint_threshold=1e-3))}


@needs_ref
@pytest.mark.parametrize("case", ["nonfinite_start", "clamped_multipliers", "kkt_test_disabled", "loose_feasibility"])
def test_oracle_matches_reference_on_edge_configurations(case):
    """NaN / inf / overflowing start points (ClampMultiplier's isfinite branch, the non-finite exit of
    Progress::Update), multipliers pinned at multiplier_max, the stationarity test switched off."""
    p = al.quadratic_simplex_problem(6)
    x0, cfg = _edge_cases()[case]
    o = al.oracle_minimize(p, x0, config=cfg)
    _assert_same(o, al.ref_minimize(p, x0, config=cfg))
    if case == "nonfinite_start":
        assert list(o["progress"]["status"][[1, 3, 4]]) == [1, 1, 1] and o["progress"]["status"][0] == 6
    if case == "clamped_multipliers":
        np.testing.assert_array_equal(np.abs(o["lambda"]), 0.05)


def _weird_initial_states():
    """negative / tiny / huge / infinite / NaN penalties and multipliers, one problem each"""
    rng = np.random.default_rng(15)
    x0 = rng.uniform(-1, 1, (8, 6))
    pen = np.array([-1.0, -0.0, 1e-300, 1e300, np.inf, np.nan, 3.0, 0.0])
    lam = np.array([0.0, 1e30, -1e30, np.nan, np.inf, 0.5, -0.5, 1e-320])[:, None]
    mu = np.array([-1.0, 0.0, 1e25, np.nan, 2.0, np.inf, 0.5, -0.0])[:, None]
    return x0, lam, mu, pen


@needs_ref
def test_oracle_matches_reference_on_weird_initial_states():
    p = al.quadratic_simplex_problem(6)
    x0, lam, mu, pen = _weird_initial_states()
    cfg = al.default_config(outer_num_iterations=6)
    _assert_same(al.oracle_minimize(p, x0, lam, mu, pen, config=cfg), al.ref_minimize(p, x0, lam, mu, pen, config=cfg))


@needs_ref
def test_oracle_matches_reference_with_initial_multipliers():
    p = al.quadratic_simplex_problem(5, seed=2)
    x0 = np.random.default_rng(4).uniform(-1, 1, (5, 5))
    lam0 = np.linspace(-1.0, 1.0, 5)[:, None]
    mu0 = np.linspace(0.0, 2.0, 5)[:, None]
    o = al.oracle_minimize(p, x0, lambda0=lam0, mu0=mu0, penalty0=2.0)
    r = al.ref_minimize(p, x0, lambda0=lam0, mu0=mu0, penalty0=2.0)
    _assert_same(o, r)


@needs_ref
@pytest.mark.parametrize("bounds", ["set", "never_set"])
@pytest.mark.parametrize("linesearch", ["more_thuente", "hager_zhang"])
def test_oracle_is_bit_identical_to_reference_with_lbfgsb_inner_solver(bounds, linesearch):
    """AugmentedLagrangian<Problem, Lbfgsb<FunctionExpr>>: box by the inner solver, projected KKT norm when the
    bounds were set on the solver (HasProjectedGradientInfNorm, augmented_lagrangian.h:47-58)."""
    p, lower, upper = al.boxed_rosenbrock_problem(6)
    x0 = np.random.default_rng(3).uniform(-1, 1, (6, 6))
    cfg = al.default_config(outer_num_iterations=25)
    kw = dict(lower=lower, upper=upper) if bounds == "set" else {}
    o = al.oracle_box_minimize(p, x0, config=cfg, linesearch=linesearch, **kw)
    r = al.ref_box_minimize(p, x0, config=cfg, linesearch=linesearch, **kw)
    _assert_same(o, r)
    if bounds == "set":
        assert np.all(o["x"] >= lower - 1e-12) and np.all(o["x"] <= upper + 1e-12)
        assert np.any(np.abs(o["x"][:, 0] - upper[0]) < 1e-9)        # the box is active at the solution
        fin = o["progress"]["status"] == 6
        assert fin.any() and np.all(o["max_violation"][fin] <= 1e-5)


# Terms that are sums of primitives (the reference's AddExpression) -----------------------------------------------
@needs_ref
def test_summed_terms_are_bit_identical_to_reference():
    rng = np.random.default_rng(31)
    cfg = al.default_config(outer_num_iterations=20)
    p = al.three_part_problem(7)
    x0 = rng.uniform(-1, 1, (6, 7))
    _assert_same(al.oracle_minimize(p, x0, config=cfg), al.ref_minimize(p, x0, config=cfg))
    q = al.quadratic_at_12_problem()
    o = al.oracle_minimize(q, [[1.0, 1.0]], penalty0=1.0)
    _assert_same(o, al.ref_minimize(q, [[1.0, 1.0]], penalty0=1.0))
    np.testing.assert_allclose(o["x"][0], [0.5, 1.5], atol=1e-3)       # the reference test's expectations (:602-611)
    assert abs(o["x"][0, 0] - 0.5) <= 1e-5 and 2.0 - o["x"][0].sum() >= -1e-5 and o["mu"][0, 0] >= -1e-2


# One residual of a least-squares function as a primitive (MI355_AL_TERM_SQUARED_AFFINE, ABI 9) -------------------
@needs_ref
def test_squared_affine_terms_are_bit_identical_to_reference():
    """(a.x - c)^2 with gradient (2 (a.x - c)) a, summed left to right: the oracle against a reference functor written
    the way a user writes one (oracle/ref_auglag_capi.cpp SquaredAffineTerm) inside the reference's own solver; and the
    augmented-Lagrangian half of src/examples/linear_regression.cc reaches that program's optimum (1, 1.6)."""
    cfg = al.default_config(outer_num_iterations=25)
    for n, rows in ((3, 2), (9, 5), (33, 11)):
        p = al.least_squares_problem(n, rows)
        x0 = np.random.default_rng(n).uniform(-1, 1, (5, n))
        _assert_same(al.oracle_minimize(p, x0, config=cfg), al.ref_minimize(p, x0, config=cfg))
        lo, hi = np.full(n, -0.3), np.full(n, 0.4)
        _assert_same(al.oracle_box_minimize(p, x0, lower=lo, upper=hi, config=cfg),
                     al.ref_box_minimize(p, x0, lower=lo, upper=hi, config=cfg))
    q = al.linear_regression_problem()
    o = al.oracle_minimize(q, [[-1.0, 2.0]], penalty0=1.0)
    _assert_same(o, al.ref_minimize(q, [[-1.0, 2.0]], penalty0=1.0))
    np.testing.assert_allclose(o["x"][0], [1.0, 1.6], atol=1e-4)
    assert o["progress"]["status"][0] == 6                               # Finished


# Terms that are products of two primitives (the reference's ProdExpression) -------------------------------------
@needs_ref
def test_product_terms_are_bit_identical_to_reference():
    """function_expressions.h:260-315: value fx * gx, gradient gx * grad_f + fx * grad_g.  The oracle's product node
    against the reference's own operator* inside its augmented-Lagrangian solver; Hs029 written over the menu reaches the
    optimum of the reference's test (:1064-1150) and equals the user-functor formulation's."""
    cfg = al.default_config(outer_num_iterations=20)
    for n in (3, 9, 33):
        p = al.product_terms_problem(n)
        x0 = np.random.default_rng(n).uniform(0.1, 1.0, (5, n))
        _assert_same(al.oracle_minimize(p, x0, config=cfg), al.ref_minimize(p, x0, config=cfg))
    q = al.hs029_product_problem()
    o = al.oracle_minimize(q, [[1.0, 1.0]])
    _assert_same(o, al.ref_minimize(q, [[1.0, 1.0]]))
    np.testing.assert_allclose(o["x"][0], [2.0 * np.sqrt(6.0), 2.0 * np.sqrt(3.0)], atol=1e-3)
    assert o["progress"]["status"][0] == 6                               # Finished


@needs_ref
def test_hs016_box_pinned_optimum():
    """AugmentedLagrangianBoxInterface.BoxPinnedOptimumStopsOnKkt (:1198-1275) over the menu: Finished, fewer than
    20 outer iterations, x* = (0.5, 0.25) — and the oracle equals the reference solver bit for bit."""
    p, lower, upper = al.hs016_problem()
    o = al.oracle_box_minimize(p, [[-2.0, 1.0]], lower=lower, upper=upper)
    _assert_same(o, al.ref_box_minimize(p, [[-2.0, 1.0]], lower=lower, upper=upper))
    assert o["progress"]["status"][0] == 6 and o["progress"]["num_iterations"][0] < 20
    np.testing.assert_allclose(o["x"][0], [0.5, 0.25], atol=1e-4)


def test_butterfly_policy_agrees_with_sequential_to_rounding():
    p = al.quadratic_simplex_problem(12)
    x0 = np.random.default_rng(9).uniform(-1, 1, (6, 12))
    a = al.oracle_minimize(p, x0)
    b = al.oracle_minimize(p, x0, reduction="butterfly", width=16)
    np.testing.assert_allclose(a["x"], b["x"], atol=2e-4)
    assert np.all(b["max_violation"] <= 1e-5)


@needs_ref
@pytest.mark.parametrize("n", [3, 9, 20])
def test_random_term_tables_bitwise_vs_reference(n):
    """Every primitive kind in every position of one- to three-part terms, every form, 0-2 constraints of each kind:
    the composite, the inner solves and three outer steps against the reference solver."""
    rng = np.random.default_rng(9000 + n)
    cfg = al.default_config(outer_num_iterations=3)
    for trial in range(12):
        p = al.random_problem(n, rng)
        x0 = rng.uniform(-1, 1, (4, n))
        pen0 = 0.0 if trial % 2 else 1.0
        _assert_same(al.oracle_minimize(p, x0, penalty0=pen0, config=cfg), al.ref_minimize(p, x0, penalty0=pen0, config=cfg))
        if trial % 3 == 0:      # ... the other inner solvers of the reference's template argument
            _assert_same(al.oracle_minimize(p, x0, penalty0=pen0, config=cfg, linesearch="hager_zhang"),
                         al.ref_minimize(p, x0, penalty0=pen0, config=cfg, linesearch="hager_zhang"))
            lo, hi = np.full(n, -1.0), np.full(n, 0.6)
            _assert_same(al.oracle_box_minimize(p, x0, lower=lo, upper=hi, penalty0=pen0, config=cfg),
                         al.ref_box_minimize(p, x0, lower=lo, upper=hi, penalty0=pen0, config=cfg))


# Golden vectors produced by the reference itself (tests/golden/make_golden_auglag.py); they travel to the GPU box ------
def _golden():
    import os
    sys_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden_auglag", os.path.join(sys_path, "make_golden_auglag.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.cases(), np.load(os.path.join(sys_path, "auglag_reference_vectors.npz"))


def _oracle_run(case, **kw):
    p, x0, pen0, cfg_kw, inner, bounds, ls = case
    cfg = al.default_config(**cfg_kw)
    if inner == "lbfgsb":
        return al.oracle_box_minimize(p, x0, lower=bounds[0], upper=bounds[1], penalty0=pen0, config=cfg, linesearch=ls, **kw)
    return al.oracle_minimize(p, x0, penalty0=pen0, config=cfg, linesearch=ls, **kw)


@pytest.mark.parametrize("name", ["circle", "simplex12", "simplex40_hz", "quadratic_at_12", "three_part7", "hs016_box",
                                  "boxed_rosenbrock6", "hs024_user_box", "hs029_user"])
def test_oracle_reproduces_the_reference_golden_vectors(name):
    """No libref.so needed: the committed outputs of the reference solver, bit for bit."""
    cases, gold = _golden()
    np.testing.assert_array_equal(cases[name][1], gold[name + "/x0"])
    o = _oracle_run(cases[name])
    for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient"):
        np.testing.assert_array_equal(o[k], gold[name + "/" + k], err_msg=k)
    for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(o["progress"][k], gold[name + "/" + k], err_msg=k)


@needs_ref
def test_restart_from_a_returned_state_matches_reference():
    """AugmentedLagrangeState is in/out in full: a state returned by a solve that ran out of outer iterations and fed
    back into Minimize carries its max_violation into the first penalty-growth test (augmented_lagrangian.h:435).
    Oracle == reference bit for bit across the restart; dropping the carried violation changes the penalty path."""
    rng = np.random.default_rng(21)
    p = al.rosenbrock_ball_problem(8)
    x0 = rng.uniform(-1, 1, (12, 8))
    cfg = al.default_config(outer_num_iterations=3)
    o1 = al.oracle_minimize(p, x0, config=cfg)
    r1 = al.ref_minimize(p, x0, config=cfg)
    _assert_same(o1, r1)
    assert np.any(o1["max_violation"] > 0)
    kw = dict(lambda0=o1["lambda"], mu0=o1["mu"], penalty0=o1["penalty"], config=al.default_config(outer_num_iterations=4))
    o2 = al.oracle_minimize(p, o1["x"], max_violation0=o1["max_violation"], **kw)
    r2 = al.ref_minimize(p, r1["x"], max_violation0=r1["max_violation"], **kw)
    _assert_same(o2, r2)
    o3 = al.oracle_minimize(p, o1["x"], max_violation0=0.0, **kw)      # the carried violation matters
    assert not np.array_equal(o3["penalty"], o2["penalty"])


def test_user_term_problems_reach_the_optima_of_the_reference_tests():
    """src/test/augmented_lagrangian_test.cc:1018-1060 (HS024: (3, sqrt 3), f* = -1, tolerances 1e-1 / 0.5) and
    :1115-1150 (HS029: (2 sqrt 6, 2 sqrt 3), f* = -12 sqrt 2, 2e-1 / 0.5), from the tests' own starts, on the twins of
    the user term functors (kinds 100-102) -- sequential (reference order) and in the device's order."""
    p24, lo, hi = al.hs024_problem()
    for kw in ({}, {"reduction": "butterfly", "width": 16, "std_sort_order": False}):
        r = al.oracle_box_minimize(p24, [[1.0, 0.5]], lower=lo, upper=hi, **kw)
        x = r["x"][0]
        assert abs(x[0] - 3.0) <= 1e-1 and abs(x[1] - np.sqrt(3.0)) <= 1e-1
        f = ((x[0] - 3.0) ** 2 - 9.0) * x[1] ** 3 / (27.0 * np.sqrt(3.0))
        assert abs(f + 1.0) <= 0.5
    for kw in ({}, {"reduction": "butterfly", "width": 8}):
        r = al.oracle_minimize(al.hs029_problem(), [[1.0, 1.0]], **kw)
        x = r["x"][0]
        assert abs(x[0] - 2.0 * np.sqrt(6.0)) <= 2e-1 and abs(x[1] - 2.0 * np.sqrt(3.0)) <= 2e-1
        assert abs(-x[0] * x[1] + 12.0 * np.sqrt(2.0)) <= 0.5


@needs_ref
def test_user_term_twins_are_bit_identical_to_reference_functors():
    """Random starts, both inner solvers and line searches: the twins of the user terms against the reference solver on
    the reference-style functors (oracle/ref_auglag_capi.cpp kinds 100-102)."""
    rng = np.random.default_rng(77)
    p24, lo, hi = al.hs024_problem()
    x0 = rng.uniform(0.2, 4.0, (12, 2)) * [1.0, 0.4]
    for ls in ("more_thuente", "hager_zhang"):
        _assert_same(al.oracle_box_minimize(p24, x0, lower=lo, upper=hi, linesearch=ls),
                     al.ref_box_minimize(p24, x0, lower=lo, upper=hi, linesearch=ls))
        x1 = rng.uniform(0.3, 3.0, (12, 2))
        _assert_same(al.oracle_minimize(al.hs029_problem(), x1, linesearch=ls),
                     al.ref_minimize(al.hs029_problem(), x1, linesearch=ls))


# The reference's src/examples/svm_dual_al.cc: AL outside (the equality sum alpha_i y_i = 0), L-BFGS-B inside (the box
# 0 <= alpha <= C) on a dense user objective with a precomputed matrix ------------------------------------------------
def _svm_dual_al_starts(n, B):
    return np.vstack([np.zeros(n), np.random.default_rng(3).uniform(0.0, 1.0, size=(B - 1, n))])


@needs_ref
def test_svm_dual_al_twin_is_the_reference_binary_bit_for_bit():
    """`AugmentedLagrangian<Problem, Lbfgsb<FunctionExprD>>` exactly as the example sets it up (alpha0 = 0 — and a second,
    random start — penalty 1, default configuration and stopping) on a restated SvmDualObjective / linear equality:
    the sequential twin (user term kind 103 over the blob [n, Q]) reproduces every returned number, 150+ outer
    iterations deep."""
    p, y = al.svm_dual_al_problem()
    x0 = _svm_dual_al_starts(p.n, 2)
    o = al.oracle_box_minimize(p, x0, lower=0.0, upper=1.0, penalty0=1.0)
    r = al.ref_box_minimize(p, x0, lower=0.0, upper=1.0, penalty0=1.0)
    _assert_same(o, r)
    assert np.all(r["progress"]["num_iterations"] > 100)
    assert np.all(r["x"] >= 0.0) and np.all(r["x"] <= 1.0) and np.all(np.abs(r["x"] @ y) <= 1e-4)
    sv = (r["x"] > 1e-5).sum(axis=1)
    assert np.all(sv >= 10) and np.all(sv <= 60)            # a sparse dual solution: the support vectors


@needs_ref
def test_svm_dual_al_device_order_twin_is_within_1e6_of_the_reference_binary():
    """The butterfly (device-order) twin — what the GPU kernels equal bit for bit, tests/test_gpu_auglag_user_terms.py —
    against the reference binary with thresholds that let the outer loop converge (constraint 1e-8, stationarity 1e-7,
    tight inner stop; under the DEFAULT thresholds the loop stops at a violation of ~1e-5 after 155+ outer iterations and
    two summation orders land 2e-2 apart on this rank-deficient dual — the reference's own stop, not an arithmetic
    property)."""
    p, y = al.svm_dual_al_problem()
    x0 = _svm_dual_al_starts(p.n, 2)
    cfg = al.default_config(constraint_threshold=1e-8, kkt_stationarity_threshold=1e-7)
    import oracle_lib as O
    tight = O.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8, past=0)
    r = al.ref_box_minimize(p, x0, lower=0.0, upper=1.0, penalty0=1.0, config=cfg, inner_stop=tight)
    b = al.oracle_box_minimize(p, x0, lower=0.0, upper=1.0, penalty0=1.0, config=cfg, inner_stop=tight,
                               reduction="butterfly", width=128, std_sort_order=False)
    assert np.max(np.abs(b["x"] - r["x"])) <= 1e-6 and np.max(np.abs(b["lambda"] - r["lambda"])) <= 1e-6
    np.testing.assert_array_equal(b["progress"]["status"], r["progress"]["status"])
    assert np.all(r["max_violation"] <= 1e-8)


# The reference's outer-loop scenarios (src/test/augmented_lagrangian_test.cc, section C.6 and :1156-1183) over the menu:
# HalfSquaredNorm2D = the diagonal quadratic with a = (1/2, 1/2), X0MinusTarget(1) = the linear form (1, 0) minus 1,
# ZeroConstraint = the linear form (0, 0).
def _half_squared_norm_problem(equality):
    return al.Problem(2, al.term("diag_quadratic", a=[0.5, 0.5], c=0.0), [equality])


@needs_ref
def test_reference_penalty_schedule_scenarios():
    x0_minus_1 = al.term("linear", "value_minus_k", 1.0, a=[1.0, 0.0])
    # PenaltyHoldsFlatOnFeasibleProblem (:694-718): the conditional schedule never fires; rho stays EXACTLY 1
    p = _half_squared_norm_problem(al.term("linear", a=[0.0, 0.0]))
    o = al.oracle_minimize(p, [[0.0, 0.0]], penalty0=1.0)
    _assert_same(o, al.ref_minimize(p, [[0.0, 0.0]], penalty0=1.0))
    assert o["penalty"][0] == 1.0
    # PenaltyGrowthCanBeDisabled (:728-753): penalty_growth_factor = 1 from an infeasible start
    p = _half_squared_norm_problem(x0_minus_1)
    cfg = al.default_config(penalty_growth_factor=1.0)
    o = al.oracle_minimize(p, [[5.0, 5.0]], penalty0=1.0, config=cfg)
    _assert_same(o, al.ref_minimize(p, [[5.0, 5.0]], penalty0=1.0, config=cfg))
    assert o["penalty"][0] == 1.0
    # PenaltyGrowsOnlyWhileViolationLags (:766-790): some growth, bounded by 1e4
    o = al.oracle_minimize(p, [[5.0, 5.0]], penalty0=1.0)
    _assert_same(o, al.ref_minimize(p, [[5.0, 5.0]], penalty0=1.0))
    assert 1.0 <= o["penalty"][0] <= 1e4
    # KktStationarityReportedOnFinishedState (:1156-1183): Finished, with a small reported Lagrangian gradient
    assert o["progress"]["status"][0] == 6 and o["max_lagrangian_gradient"][0] <= 1e-2
    np.testing.assert_allclose(o["x"][0], [1.0, 0.0], atol=1e-4)
    np.testing.assert_allclose(o["lambda"][0], -1.0, atol=1e-3)     # stationarity: x0 + lambda = 0 at x0 = 1
