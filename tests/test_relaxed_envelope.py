"""Where the 1e-6 envelope of the two RELAXED policies ends (CPU; round-3 verdict, Weak 1).

The relaxed-algebra L-BFGS-B twin (oracle/lbfgsb_fast_oracle.hpp = csrc/lbfgsb_fast_kernel.hpp) and the normal-equation
ridge twin (objective id 5 = csrc/ridge_gram.hpp) against the REFERENCE binary (oracle/_ref/libref.so: the unmodified
solver/lbfgsb.h, solver/lbfgs.h) on ill-conditioned inputs, with the tolerance stated per regime:

  * the EXACT policy (sequential twin) equals the reference binary bit for bit in every regime — it is the direct libref
    link of Lbfgsb on the diagonal-quadratic and regression objectives (src/examples/linear_regression.cc:58-74 runs
    Lbfgsb on a regression objective);
  * a relaxed policy is within 1e-6 of the reference while the reference itself is within ~1e-6 of the true minimiser
    (diag spectrum spread <= 1e4 under Lbfgsb; cond(A^T A + lambda I) <= 3e2 for the ridge objective);
  * beyond, the reference's own stopping tests fire 1e-6 ... 0.6 away from the minimiser; the relaxed policy is then held
    to the reference's f* (relative) and to "no further from the TRUE minimiser than the reference" (L-BFGS-B, factor 3) /
    "a point the reference's own gradient test accepts, evaluated independently in the direct form" (ridge).
The device == twin half of the chain is tests/test_gpu_relaxed_envelope.py."""
import numpy as np
import pytest

import envelope_cases as E
import oracle_lib as O

TOL = 1e-6


def _ref():
    import ref_lib
    if not ref_lib.available() or not hasattr(ref_lib.lib(), "ref_lbfgsb_ridge_minimize_batch"):
        pytest.skip("oracle/_ref/libref.so (with the Lbfgsb regression entry) not built")
    return ref_lib


def _x0(B, n, seed):
    from cppnumericalsolvers_amd.engine import synthetic_x0_host
    return synthetic_x0_host(B, n, "u2", seed=seed)


@pytest.mark.parametrize("box", sorted(E.BOXES))
@pytest.mark.parametrize("spread", E.SPREADS)
def test_lbfgsb_policies_on_ill_conditioned_quadratics(spread, box):
    R = _ref()
    n, B, m = 32, 12, 5
    a, params = E.diag_spectrum(n, spread)
    lo, hi = E.box_arrays(n, E.BOXES[box])
    x0 = _x0(B, n, seed=5)
    st = E.tight_stop(O)
    xr, fr, gr, pr = R.lbfgsb_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo, upper=hi)
    # exact policy: the reference's operation order (std::sort's tie order included) -> bit for bit, every regime
    xe, fe, ge, pe = O.lbfgsb_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo, upper=hi,
                                             std_sort_order=True)
    np.testing.assert_array_equal(xe, xr)
    np.testing.assert_array_equal(fe, fr)
    np.testing.assert_array_equal(pe["status"], pr["status"])
    np.testing.assert_array_equal(pe["num_iterations"], pr["num_iterations"])
    # relaxed policy
    xf, ff, gf, pf = O.lbfgsb_fast_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo, upper=hi)
    assert np.all(np.isfinite(xf)) and np.all(np.isfinite(ff))
    if lo is not None:
        assert np.all(xf >= lo) and np.all(xf <= hi)
    true = E.diag_minimiser(n, E.BOXES[box])
    ref_err, fast_err = np.max(np.abs(xr - true)), np.max(np.abs(xf - true))
    dx, df = np.max(np.abs(xf - xr)), np.max(np.abs(ff - fr))
    rel_df = np.max(np.abs(ff - fr) / np.maximum(1.0, np.abs(fr)))
    if spread <= E.LBFGSB_SPREAD_1E6_BAR:
        assert dx <= TOL and df <= TOL, (dx, df)
        np.testing.assert_array_equal(pf["status"], pr["status"])
    elif spread <= 1e6:
        assert dx <= 1e-5 and rel_df <= 1e-9, (dx, rel_df)          # measured 1.8e-6 / 1.7e-12
        assert np.all(pf["status"] != 1) and np.all(pr["status"] != 1)
    else:
        # neither the reference nor the relaxed policy converges within 10 001 iterations on the unbounded / weakly
        # bounded problems: same status, f* to 1e-3 relative, and no further from the minimiser than the reference
        np.testing.assert_array_equal(pf["status"] == 1, pr["status"] == 1)
        assert rel_df <= 1e-3, rel_df
    assert fast_err <= 3.0 * ref_err + TOL, (fast_err, ref_err)
    # iteration counts: the relaxed algebra does not cost iterations
    assert pf["num_iterations"].mean() <= 1.05 * pr["num_iterations"].mean() + 1.0


@pytest.mark.parametrize("lam", E.RIDGE_LAMBDAS)
@pytest.mark.parametrize("cond", E.RIDGE_CONDITIONS)
def test_gram_form_on_ill_conditioned_regression(cond, lam):
    R = _ref()
    rows, n, B = 128, 64, 8
    A, Y = E.ridge_case(rows, n, cond, B)
    x0 = np.zeros((B, n))
    st = O.parity_stop()
    params = O.ridge_params(A, lam)
    xr, fr, gr, pr = R.ridge_minimize_batch(A, lam, Y, x0, stop=st)
    # reference-order policy (objective id 2, sequential): bit for bit in every regime
    xs, fs, _, ps = O.minimize_batch("squared_error_ridge", x0, m=10, stop=st, params=params, per_problem=Y)
    np.testing.assert_array_equal(xs, xr)
    np.testing.assert_array_equal(fs, fr)
    xg, fg, gg, pg = O.minimize_batch("squared_error_ridge_gram", x0, m=10, stop=st, params=params, per_problem=Y,
                                      reduction="butterfly_fma", width=64, fma_group=2)
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    ref_err, gram_err = np.max(np.abs(xr - closed)), np.max(np.abs(xg - closed))
    dx, df = np.max(np.abs(xg - xr)), np.max(np.abs(fg - fr))
    kH = E.ridge_hessian_condition(A, lam)
    assert np.all(pg["status"] != 1) and np.all(pr["status"] != 1)
    if kH <= E.RIDGE_COND_H_1E6_BAR:
        assert dx <= TOL and df <= TOL, (kH, dx, df)
        assert gram_err <= 3.0 * ref_err + TOL, (kH, gram_err, ref_err)
    else:
        # the reference's relative gradient-norm test fires up to 0.6 away from the closed form here (ref_err); f* still
        # agrees, and the normal-equation form stops where the reference's OWN test — evaluated independently, in the
        # direct form — is satisfied: ||grad f(x)||_inf < 1e-8 max(1, ||x||_inf)  (progress.h:299-317)
        assert df <= 1e-5 * max(1.0, float(np.max(np.abs(fr)))), (kH, df)
        assert ref_err > TOL                                       # (the regime: the reference is not at the minimiser)
        for x in (xr, xg):
            grad = 2.0 * (x @ A.T - Y) @ A + 2.0 * lam * x
            rel = np.abs(grad).max(axis=1) / np.maximum(1.0, np.abs(x).max(axis=1))
            assert np.all(rel <= 1.1e-8), (kH, rel.max())


@pytest.mark.parametrize("m,box", [(5, (-0.25, 0.4)), (5, None), (10, (-0.25, 0.4)), (8, (0.0, 1.0))])
def test_lbfgsb_on_the_regression_objective_equals_the_reference_binary(m, box):
    """src/examples/linear_regression.cc:58-74 (Lbfgsb on a regression objective) at the README's ridge functors: the
    exact twin == Lbfgsb<FunctionExpr, m> of the reference, bit for bit."""
    R = _ref()
    from cppnumericalsolvers_amd.engine import synthetic_ridge_host
    rows, n, B, lam = 40, 24, 24, 0.05
    A, Y = synthetic_ridge_host(B, rows, n, seed=77)
    x0 = np.random.default_rng(2).uniform(-0.5, 0.5, size=(B, n))
    lo, hi = E.box_arrays(n, box)
    for st in (O.lbfgsb_default_stop(), E.tight_stop(O)):
        xr, fr, gr, pr = R.lbfgsb_ridge_minimize_batch(A, lam, Y, x0, m=m, stop=st, lower=lo, upper=hi)
        xe, fe, ge, pe = O.lbfgsb_minimize_batch("squared_error_ridge", x0, m=m, stop=st, params=O.ridge_params(A, lam),
                                                 lower=lo, upper=hi, per_problem=Y, std_sort_order=True)
        np.testing.assert_array_equal(xe, xr)
        np.testing.assert_array_equal(fe, fr)
        np.testing.assert_array_equal(ge, gr)
        np.testing.assert_array_equal(pe["status"], pr["status"])
        np.testing.assert_array_equal(pe["num_iterations"], pr["num_iterations"])
    if box is not None:
        assert np.any(xr == lo) or np.any(xr == hi)      # the box is active


def test_lbfgsb_quadratic_reference_link_across_history_sizes():
    """Lbfgsb<DiagQuadraticN, m> for the history sizes libref instantiates, default preset and tight stop: exact twin ==
    reference binary bit for bit (the second direct link beside Rosenbrock)."""
    R = _ref()
    n, B = 20, 16
    a, params = E.diag_spectrum(n, 50.0)
    x0 = _x0(B, n, seed=9)
    for m in (3, 5, 6, 8, 10):
        for box in ((-1.5, 0.8), None):
            lo, hi = E.box_arrays(n, box)
            for st in (O.lbfgsb_default_stop(), E.tight_stop(O)):
                xr, fr, gr, pr = R.lbfgsb_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo, upper=hi)
                xe, fe, ge, pe = O.lbfgsb_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo,
                                                         upper=hi, std_sort_order=True)
                np.testing.assert_array_equal(xe, xr)
                np.testing.assert_array_equal(fe, fr)
                np.testing.assert_array_equal(pe["num_iterations"], pr["num_iterations"])


def test_default_policy_constants_and_the_automatic_ridge_form():
    """The envelope constants the product carries are the ones these tests pin: capi.LBFGSB_RELAXED_MAX_SPREAD ==
    MI355_LBFGSB_RELAXED_MAX_SPREAD of include/mi355_lbfgs.h == the last spread with 1e-6 above; gram="auto" takes the
    normal-equation form for the configs[3] matrix and refuses it when the (rigorous) condition bound leaves 3e2."""
    import os
    import re
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi, engine
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mi355_lbfgs.h")).read()
    assert float(re.search(r"#define MI355_LBFGSB_RELAXED_MAX_SPREAD\s+(\S+)", header).group(1)) == capi.LBFGSB_RELAXED_MAX_SPREAD
    assert capi.LBFGSB_RELAXED_MAX_SPREAD == E.LBFGSB_SPREAD_1E6_BAR
    assert engine.GRAM_AUTO_MAX_CONDITION == E.RIDGE_COND_H_1E6_BAR
    A, _ = amd.synthetic_ridge_host(1, 128, 64, 20260923)          # bench.py --workload cfg4
    assert np.linalg.cond(A.T @ A + 0.1 * np.eye(64)) <= engine.ridge_condition_bound(A, 0.1) <= engine.GRAM_AUTO_MAX_CONDITION
    assert amd.SquaredErrorRidge(A, 0.1, gram="auto").name == "squared_error_ridge_gram"
    assert amd.SquaredErrorRidge(A, 1e-6, gram="auto").name == "squared_error_ridge"
    assert amd.SquaredErrorRidge(A, 0.0, gram="auto").name == "squared_error_ridge"
    big = np.random.default_rng(0).normal(size=(300, 80)) / np.sqrt(300.0)
    assert amd.SquaredErrorRidge(big, 0.1, gram="auto").name == "squared_error_ridge_gram"       # rows <= 4096, n <= 256 (round 4)
    huge = np.random.default_rng(0).normal(size=(64, 300))
    assert amd.SquaredErrorRidge(huge, 100.0, gram="auto").name == "squared_error_ridge"        # outside the built shapes
    for cond in E.RIDGE_CONDITIONS:
        for lam in E.RIDGE_LAMBDAS:
            Ac = E.conditioned_matrix(128, 64, cond)
            assert E.ridge_hessian_condition(Ac, lam) <= engine.ridge_condition_bound(Ac, lam) * (1 + 1e-12)
