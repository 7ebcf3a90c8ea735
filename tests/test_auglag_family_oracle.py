"""Constraint FAMILIES of the augmented-Lagrangian path (mi355_al_problem.family_eq / family_ineq) on the CPU checkers.

The reference takes constraint vectors of any length (function_problem.h:57-84); its src/examples/svm_primal_al.cc:139-147
pushes 200 affine functors into one.  Here such constraints travel as matrices; to the checkers they are ordinary terms
`LinearTerm(a_i) - k_i` that follow the table's terms of their kind:
  * the oracle's sequential form == the REFERENCE binary bit for bit (oracle/_ref: the reference's own AugmentedLagrangian,
    ToAugmentedLagrangian and Progress::Update over those functors) — multipliers of all 200 constraints included;
  * the oracle's butterfly form (what the device computes: tests/test_gpu_auglag_family.py) differs from it only through
    the objective's and the solver's reductions: a family constraint is the reference's ascending chain under every policy;
  * the primal SVM of the reference's example reaches the classifier the dual formulations reach."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import auglag_lib as al  # noqa: E402
import ref_lib  # noqa: E402

needs_ref = pytest.mark.skipif(not ref_lib.available(), reason="oracle/_ref/libref.so not built")


def _same(a, b):
    for k in ("x", "lambda", "mu", "penalty", "max_violation", "max_lagrangian_gradient"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    for k in ("status", "num_iterations", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(a["progress"][k], b["progress"][k], err_msg=k)


@needs_ref
@pytest.mark.parametrize("n,f_eq,f_ineq,table", [(6, 3, 9, True), (12, 0, 40, False), (20, 7, 0, True), (33, 5, 60, True)])
def test_sequential_twin_equals_the_reference_binary_with_families(n, f_eq, f_ineq, table):
    p = al.random_family_problem(n, f_eq, f_ineq, seed=n, table=table)
    assert p.n_eq == f_eq + (1 if table else 0) and p.n_ineq == f_ineq + (1 if table else 0)
    x0 = np.random.default_rng(n).uniform(-1, 1, (5, n))
    cfg = al.default_config(outer_num_iterations=25)
    o = al.oracle_minimize(p, x0, config=cfg)
    r = al.ref_minimize(p, x0, config=cfg)
    _same(o, r)
    # (the reference's loop drives these to feasibility; its default inner stopping test leaves the KKT norm above the outer
    #  threshold, so the solves end on the outer iteration limit — in the reference binary exactly as here)
    assert np.all(o["max_violation"] <= 1e-4) and np.all(np.isfinite(o["max_lagrangian_gradient"]))
    assert np.all(o["progress"]["num_iterations"] == 26)
    # ... and the chain kind IS the linear kind under the sequential policy
    q = al.Problem(n, p.terms[0], equality=[dict(t, prims=[("linear",) + t["prims"][0][1:]]) for t in p.terms[1:1 + p.n_eq]],
                   inequality=[dict(t, prims=[("linear",) + t["prims"][0][1:]]) for t in p.terms[1 + p.n_eq:]])
    _same(al.oracle_minimize(q, x0, config=cfg), o)


@needs_ref
def test_primal_svm_with_two_hundred_constraints_equals_the_reference_binary():
    """The shape of src/examples/svm_primal_al.cc: 105 variables, 200 inequality constraints, start at the origin with
    penalty 1 (:158-167), the example's solver defaults."""
    p, X, y = al.svm_primal_al_problem()
    assert (p.n, p.n_eq, p.n_ineq) == (105, 0, 200)
    x0 = np.vstack([np.zeros(p.n), np.random.default_rng(5).uniform(-0.5, 0.5, (1, p.n))])
    cfg = al.default_config(outer_num_iterations=60)
    o = al.oracle_minimize(p, x0, penalty0=1.0, config=cfg)
    r = al.ref_minimize(p, x0, penalty0=1.0, config=cfg)
    _same(o, r)
    d = X.shape[1]
    w, b, xi = o["x"][0, :d], o["x"][0, d], o["x"][0, d + 1:]
    assert o["max_violation"][0] <= 1e-4
    assert np.mean(np.sign(X @ w + b) == y) >= 0.85
    # complementary slackness: a positive multiplier sits on an active constraint
    margins = y * (X @ w + b) - 1.0 + xi
    mu = o["mu"][0]
    assert np.all(np.abs(margins[mu[:100] > 1e-3]) <= 1e-3) and np.all(np.abs(xi[mu[100:] > 1e-3]) <= 1e-3)


def test_butterfly_twin_differs_only_through_the_reductions_outside_the_family():
    """One composite evaluation: with an objective whose value needs no reduction tree (a single coordinate) the
    butterfly form equals the sequential form bit for bit — the family part has no policy; with a dense objective the
    two differ by rounding only."""
    n = 24
    rng = np.random.default_rng(2)
    p = al.random_family_problem(n, 4, 30, seed=1, table=False)
    x = rng.uniform(-1, 1, (8, n))
    lam, mu, pen = rng.normal(size=(8, p.n_eq)), rng.uniform(0, 2, (8, p.n_ineq)), rng.uniform(0.5, 20, 8)
    e0 = np.zeros(n)
    e0[0] = 2.0
    single = al.Problem(n, al.term("linear", a=e0), family_equality=p.family_equality, family_inequality=p.family_inequality)
    fs, gs = al.oracle_eval(single, x, lam, mu, pen)
    fb, gb = al.oracle_eval(single, x, lam, mu, pen, reduction="butterfly", width=32)
    np.testing.assert_array_equal(fs, fb)
    np.testing.assert_array_equal(gs, gb)
    fs, gs = al.oracle_eval(p, x, lam, mu, pen)
    fb, gb = al.oracle_eval(p, x, lam, mu, pen, reduction="butterfly", width=32)
    np.testing.assert_allclose(fb, fs, rtol=1e-13)
    np.testing.assert_array_equal(gb, gs)      # the diagonal quadratic's gradient needs no reduction either


@needs_ref
def test_hostile_starts_and_degenerate_rows_equal_the_reference_binary():
    """A NaN and an infinite start, a row of zeros, a duplicated row, and an infeasible pair of rows whose multipliers run
    into multiplier_max: the sequential twin still equals the reference binary bit for bit (NaNs included)."""
    rng = np.random.default_rng(77)
    n = 16
    G = rng.normal(size=(32, n))
    G[5] = 0.0
    G[9] = G[8]
    k = G @ np.full(n, 0.2) - 0.1
    k[5] = -0.3
    p = al.Problem(n, al.term("diag_quadratic", a=rng.uniform(0.5, 2.0, n), c=0.0), family_inequality=(G, k))
    x0 = rng.uniform(-1, 1, (9, n))
    x0[3, 4] = np.nan
    x0[6, 0] = np.inf
    cfg = al.default_config(outer_num_iterations=12)
    o, r = al.oracle_minimize(p, x0, config=cfg), al.ref_minimize(p, x0, config=cfg)
    _same(o, r)
    assert np.isnan(o["x"][3]).any() and np.all(o["mu"][:, 5] == 0.0)
    A = np.zeros((2, 4))
    A[0, 0], A[1, 0] = 1.0, -1.0
    q = al.Problem(4, al.term("squared_norm"), family_inequality=(A, np.array([1.0, 1.0])))
    cfgc = al.default_config(outer_num_iterations=25, multiplier_max=50.0)
    x0 = rng.uniform(-1, 1, (5, 4))
    o, r = al.oracle_minimize(q, x0, config=cfgc), al.ref_minimize(q, x0, config=cfgc)
    _same(o, r)
    assert np.max(o["mu"]) == 50.0
