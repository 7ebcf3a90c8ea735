"""CPU tests of the relaxed-algebra L-BFGS-B policy: its twin (oracle/lbfgsb_fast_oracle.hpp, what the HIP kernel
lbfgsb_fast_kernel.hpp computes operation for operation) against the REFERENCE binary (oracle/_ref/libref.so, the
unmodified lbfgsb.h over the Eigen shim) at the north star's 1e-6 on x* and f*.  The device == twin half of the chain
is tests/test_gpu_lbfgsb_fast.py."""
import numpy as np
import pytest

import oracle_lib as O

TOL = 1e-6


def _ref():
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so not built")
    return ref_lib


def _tight():
    return O.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8, past=0)


def _x0(B, n, kind, seed):
    from cppnumericalsolvers_amd.engine import synthetic_x0_host
    return synthetic_x0_host(B, n, kind, seed=seed)


@pytest.mark.parametrize("n,m,kind,box", [(32, 5, "u2", (-1.5, 0.8)), (32, 5, "std", (-1.5, 0.8)), (64, 5, "u2", (-1.5, 0.8)),
                                          (8, 5, "u2", (-1.5, 0.8)), (2, 5, "u2", None), (20, 5, "std", None),
                                          (32, 6, "u2", (-1.5, 0.8)), (100, 5, "u2", (-1.5, 0.8)), (17, 5, "u2", (-2.0, 0.5)),
                                          (32, 10, "u2", (-1.5, 0.8)), (64, 10, "std", (-1.5, 0.8)), (20, 10, "u2", None)])
def test_fast_twin_within_tolerance_of_the_reference_binary(n, m, kind, box):
    """Tight stopping (the parity stop of configs[4]): x* and f* of the relaxed algebra within 1e-6 of the reference's
    own Lbfgsb<F, m> on the same starts."""
    R = _ref()
    B = 256
    x0 = _x0(B, n, kind, seed=7 * n + m)
    lo = np.full(n, box[0]) if box else None
    hi = np.full(n, box[1]) if box else None
    xf, ff, gf, pf = O.lbfgsb_fast_minimize_batch("rosenbrock", x0, m=m, stop=_tight(), lower=lo, upper=hi)
    xr, fr, gr, pr = R.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=_tight(), lower=lo, upper=hi)
    assert np.all(np.isfinite(xf))
    assert np.max(np.abs(xf - xr)) <= TOL and np.max(np.abs(ff - fr)) <= TOL
    assert np.all(pf["status"] != 1)                      # no iteration limit
    if box:
        assert np.all(xf <= box[1]) and np.all(xf >= box[0])
    # the relaxed algebra does not cost iterations: within 10 % of the reference's count on average
    assert pf["num_iterations"].mean() <= 1.10 * pr["num_iterations"].mean() + 1.0


def test_fast_twin_default_preset_and_history_sizes():
    """The Lbfgsb default preset stops on the relative f-delta long before x has settled: compare f at the 1e-4 of the
    reference's own tests.  Every history size the kernels are built for (m = 1..10; 9 and 10 on thirty-two lanes)."""
    R = _ref()
    n = 24
    x0 = _x0(64, n, "u2", seed=3)
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    for m in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10):
        xf, ff, gf, pf = O.lbfgsb_fast_minimize_batch("rosenbrock", x0, m=m, lower=lo, upper=hi)
        assert np.all(np.isfinite(ff)) and np.all(pf["status"] != 1)
        if m in (5, 6, 10):   # the history sizes libref.so instantiates
            xr, fr, gr, pr = R.lbfgsb_minimize_batch("rosenbrock", x0, m=m, lower=lo, upper=hi)
            assert np.max(np.abs(ff - fr)) <= (1e-3 if m < 3 else 1e-4)
        xt, ft, _, pt = O.lbfgsb_fast_minimize_batch("rosenbrock", x0, m=m, stop=_tight(), lower=lo, upper=hi)
        xe, fe, _, pe = O.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=_tight(), lower=lo, upper=hi)
        assert np.max(np.abs(xt - xe)) <= TOL and np.max(np.abs(ft - fe)) <= TOL, m


def test_fast_twin_reference_fixtures_and_corner_cases():
    """src/test/verify.cc:190 LbfgsbTest Far / Near (unbounded): |f(x*)| <= 1e-4; the degenerate boxes of
    test_oracle.LBFGSB_CORNER_CASES against the reference binary."""
    R = _ref()
    x, f, g, p = O.lbfgsb_fast_minimize_batch("rosenbrock", np.array([[15.0, 8.0], [-1.0, 2.0]]))
    assert np.all(np.abs(f) <= 1e-4)
    from test_oracle import LBFGSB_CORNER_CASES
    for name, (x0, lo, hi) in sorted(LBFGSB_CORNER_CASES.items()):
        xf, ff, gf, pf = O.lbfgsb_fast_minimize_batch("rosenbrock", x0, lower=lo, upper=hi)
        xr, fr, gr, pr = R.lbfgsb_minimize_batch("rosenbrock", x0, lower=lo, upper=hi)
        np.testing.assert_array_equal(xf, xr, err_msg=name)     # (degenerate boxes: the relaxed algebra has nothing to
        np.testing.assert_array_equal(ff, fr, err_msg=name)     #  re-associate; it lands on the reference's bits)
        np.testing.assert_array_equal(pf["status"], pr["status"], err_msg=name)
    # a quadratic whose unconstrained minimiser is outside the box: every coordinate ends on its lower bound
    a = np.linspace(1.0, 9.0, 12)
    x0 = _x0(10, 12, "u2", seed=5)
    lo, hi = np.full(12, 0.25), np.full(12, 3.0)
    xf, ff, _, pf = O.lbfgsb_fast_minimize_batch("diag_quadratic", x0, lower=lo, upper=hi, params=np.concatenate([a, [1.0]]))
    np.testing.assert_array_equal(xf, np.full_like(xf, 0.25))


def test_fast_twin_mapping_choices_agree():
    """The twin (like the kernel) may run a shape with more coordinates per lane or a larger capacity than it needs;
    results then differ in the last bits only (chains over longer zero-padded columns are the same chains; the
    capacity changes the size of the identity-padded matrices, not their values)."""
    n, m = 12, 4
    x0 = _x0(32, n, "u2", seed=11)
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    ref = O.lbfgsb_fast_minimize_batch("rosenbrock", x0, m=m, stop=_tight(), lower=lo, upper=hi)
    for cap, E in ((5, 2), (8, 1), (8, 4)):
        alt = O.lbfgsb_fast_minimize_batch("rosenbrock", x0, m=m, stop=_tight(), lower=lo, upper=hi, capacity=cap,
                                           elems_per_lane=E)
        assert np.max(np.abs(alt[0] - ref[0])) <= TOL and np.max(np.abs(alt[1] - ref[1])) <= TOL
