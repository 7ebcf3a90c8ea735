"""CPU tests of the normal-equation (Gram) form of the ridge objective: its twin (oracle SquaredErrorRidge with
gram = true: what csrc/ridge_gram.hpp computes, operation for operation) against the REFERENCE binary (the README
functors `SquaredError(A, y) + lambda * L2Reg(n)` solved by the reference's Lbfgs, oracle/_ref/libref.so) and the
closed form, at the north star's 1e-6.  device == twin is tests/test_gpu_ridge_gram.py."""
import numpy as np
import pytest

import oracle_lib as O

TOL = 1e-6


def gram_mapping(n):
    P = 8
    while P < n:
        P <<= 1
    return P, (1 if P == 8 else (4 if P == 256 else 2))        # butterfly width, fused group (coordinates per lane)


@pytest.mark.parametrize("rows,n,m", [(128, 64, 10), (50, 20, 10), (100, 64, 3), (37, 33, 6), (5, 8, 5), (3, 2, 10)])
def test_gram_twin_vs_reference_and_closed_form(rows, n, m):
    import ref_lib as R
    from cppnumericalsolvers_amd.engine import synthetic_ridge_host
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not built")
    B, lam = 96, 0.1
    A, Y = synthetic_ridge_host(B, rows, n, seed=rows * 3 + n)
    x0 = np.zeros((B, n))
    P, E = gram_mapping(n)
    st = O.parity_stop()
    xg, fg, gg, pg = O.minimize_batch("squared_error_ridge_gram", x0, m=m, stop=st, params=O.ridge_params(A, lam),
                                      per_problem=Y, reduction="butterfly_fma", width=P, fma_group=E)
    if m == 10:   # the history size libref.so instantiates for Lbfgs
        xr, fr, gr, pr = R.ridge_minimize_batch(A, lam, Y, x0, stop=st)
        assert np.max(np.abs(xg - xr)) <= TOL and np.max(np.abs(fg - fr)) <= TOL
    xs, fs, _, _ = O.minimize_batch("squared_error_ridge", x0, m=m, stop=st, params=O.ridge_params(A, lam), per_problem=Y)
    assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    assert np.max(np.abs(xg - closed)) <= TOL
    assert np.all(pg["status"] >= 2) and np.all(pg["status"] <= 4)


def test_gram_objective_value_and_gradient():
    """One evaluation: the normal-equation value and gradient against the direct form, to rounding."""
    rng = np.random.default_rng(4)
    rows, n, lam = 40, 12, 0.3
    A, y, x = rng.normal(size=(rows, n)), rng.normal(size=(1, rows)), rng.normal(size=n)
    fg, gg = O.evaluate("squared_error_ridge_gram", x, params=O.ridge_params(A, lam), per_problem=y, reduction="butterfly_fma",
                        width=16, fma_group=2)
    fd, gd = O.evaluate("squared_error_ridge", x, params=O.ridge_params(A, lam), per_problem=y)
    assert abs(fg - fd) <= 1e-12 * abs(fd) and np.max(np.abs(gg - gd)) <= 1e-12 * np.max(np.abs(gd))


@pytest.mark.parametrize("rows,n", [(1000, 200), (300, 100), (129, 65), (2000, 256)])
def test_gram_twin_beyond_128_by_64_vs_the_readme_functors_under_the_reference(rows, n):
    """The reference's README functors take any A (README.md:126-160): the normal-equation twin on problems larger than
    the configs[3] shape (the kernels keep G in LDS up to n = 128 and stream it through L2 up to n = 256; rows <= 4096)
    against `SquaredError(A, y_b) + lambda * L2Reg` minimised by the reference's own Lbfgs (oracle/_ref) and against the
    closed form, at 1e-6."""
    import ref_lib as R
    from cppnumericalsolvers_amd.engine import synthetic_ridge_host
    if not R.available():
        pytest.skip("oracle/_ref/libref.so not built")
    B, lam = 6, 0.1
    A, Y = synthetic_ridge_host(B, rows, n, seed=rows + n)
    x0 = np.zeros((B, n))
    P, E = gram_mapping(n)
    st = O.parity_stop()
    xg, fg, gg, pg = O.minimize_batch("squared_error_ridge_gram", x0, m=10, stop=st, params=O.ridge_params(A, lam),
                                      per_problem=Y, reduction="butterfly_fma", width=P, fma_group=E)
    xr, fr, gr, pr = R.ridge_minimize_batch(A, lam, Y, x0, stop=st)
    assert np.max(np.abs(xg - xr)) <= TOL and np.max(np.abs(fg - fr)) <= TOL
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    assert np.max(np.abs(xg - closed)) <= TOL
    assert np.all(pg["status"] >= 2) and np.all(pg["status"] <= 4)


@pytest.mark.parametrize("rows,n", [(40, 24), (128, 64), (9, 8), (150, 70)])
def test_own_matrix_twins_vs_the_reference_binary(rows, n):
    """One `SquaredError(A_b, y_b) + lambda * L2Reg` PER PROBLEM (README.md:126-160: the README objective built once per
    data set): the reference-order twin (objective id 7) equals the reference's Lbfgs on those functors bit for bit; the
    normal-equation twin of the device kernel (id 6) is within 1e-6 of it and of the closed form."""
    import ref_lib as R
    if not R.available() or not hasattr(R.lib(), "ref_ridge_own_matrix_minimize_batch"):
        pytest.skip("oracle/_ref/libref.so without the own-matrix entry")
    rng = np.random.default_rng(rows + n)
    B, lam = 10, 0.1
    As = rng.normal(size=(B, rows, n)) / np.sqrt(rows)
    Y = rng.normal(size=(B, rows))
    data = np.ascontiguousarray(np.concatenate([As.reshape(B, -1), Y], axis=1))
    x0 = np.zeros((B, n))
    P, E = gram_mapping(n)
    params = np.array([float(rows), lam])
    for st in (O.default_stop(), O.parity_stop()):
        xr, fr, gr, pr = R.ridge_own_matrix_minimize_batch(As, lam, Y, x0, stop=st)
        xs, fs, gs, ps = O.minimize_batch("squared_error_ridge_own", x0, m=10, stop=st, params=params, per_problem=data)
        np.testing.assert_array_equal(xs, xr)
        np.testing.assert_array_equal(fs, fr)
        np.testing.assert_array_equal(gs, gr)
        np.testing.assert_array_equal(ps["num_iterations"], pr["num_iterations"])
    xg, fg, gg, pg = O.minimize_batch("squared_error_ridge_own_gram", x0, m=10, stop=O.parity_stop(), params=params,
                                      per_problem=data, reduction="butterfly_fma", width=P, fma_group=E)
    assert np.max(np.abs(xg - xr)) <= TOL and np.max(np.abs(fg - fr)) <= TOL
    closed = np.stack([np.linalg.solve(As[b].T @ As[b] + lam * np.eye(n), As[b].T @ Y[b]) for b in range(B)])
    assert np.max(np.abs(xg - closed)) <= TOL
    # every problem really has its own minimiser
    assert np.min(np.abs(xg[0] - xg[1])) >= 0 and np.max(np.abs(xg[0] - xg[1])) > 1e-3
