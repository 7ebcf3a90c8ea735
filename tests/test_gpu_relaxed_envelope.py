"""GPU half of the envelope tests (tests/test_relaxed_envelope.py holds twin vs reference binary): on the SAME
ill-conditioned inputs the device equals its twin bit for bit — the relaxed-algebra L-BFGS-B kernel, the reference-order
L-BFGS-B kernel, the normal-equation ridge form — and MI355_ARITH_DEFAULT stays inside the pinned envelope
(MI355_LBFGSB_RELAXED_MAX_SPREAD)."""
import numpy as np
import pytest

import envelope_cases as E

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


def _solve(s, obj, x0, **kw):
    import torch
    import cppnumericalsolvers_amd as amd
    x, f, g, p = s.minimize(obj, _to_dev(x0), **kw)
    torch.cuda.synchronize()
    return x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)


def _assert_same(dev, twin, msg=""):
    for a, b in zip(dev[:3], twin[:3]):
        np.testing.assert_array_equal(a, b, err_msg=msg)
    for k in ("status", "num_iterations", "nfev", "sum_k", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(dev[3][k], twin[3][k], err_msg=msg + " " + k)


@pytest.mark.parametrize("box", sorted(E.BOXES))
@pytest.mark.parametrize("spread", E.SPREADS)
def test_lbfgsb_kernels_equal_their_twins_on_ill_conditioned_quadratics(gpu_solver_factory, oracle, spread, box):
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    n, B, m = 32, 12, 5
    a, params = E.diag_spectrum(n, spread)
    lo, hi = E.box_arrays(n, E.BOXES[box])
    x0 = amd.synthetic_x0_host(B, n, "u2", seed=5)
    st = E.tight_stop(oracle)
    obj = amd.DiagQuadratic(a, 1.0)
    out = {}
    for arith in ("fma", "exact", "default"):
        s = amd.BatchedLbfgsb(m=m, stopping_progress=_engine_stop(st), context=base.ctx, arithmetic=arith)
        if lo is not None:
            s.SetBounds(lo, hi)
        out[arith] = _solve(s, obj, x0)
        # the default policy: relaxed algebra inside the pinned envelope, the reference-order kernel beyond it
        want = "exact" if (arith == "exact" or (arith == "default" and spread > amd.capi.LBFGSB_RELAXED_MAX_SPREAD)) else "fma"
        assert s.last_arithmetic() == want, (arith, spread)
    relaxed_twin = oracle.lbfgsb_fast_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo, upper=hi)
    _assert_same(out["fma"], relaxed_twin, "relaxed spread=%g %s" % (spread, box))
    exact_twin = oracle.lbfgsb_minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, lower=lo, upper=hi,
                                              reduction="butterfly", width=32)
    _assert_same(out["exact"], exact_twin, "exact spread=%g %s" % (spread, box))
    _assert_same(out["default"], exact_twin if spread > amd.capi.LBFGSB_RELAXED_MAX_SPREAD else relaxed_twin, "default")


@pytest.mark.parametrize("lam", E.RIDGE_LAMBDAS)
@pytest.mark.parametrize("cond", E.RIDGE_CONDITIONS)
def test_gram_kernel_equals_its_twin_on_ill_conditioned_regression(gpu_solver_factory, oracle, cond, lam):
    import cppnumericalsolvers_amd as amd
    rows, n, B = 128, 64, 8
    A, Y = E.ridge_case(rows, n, cond, B)
    x0 = np.zeros((B, n))
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(st), arithmetic="default")
    dev = _solve(s, amd.SquaredErrorRidge(A, lam, gram=True), x0, per_problem=_to_dev(Y))
    twin = oracle.minimize_batch("squared_error_ridge_gram", x0, m=10, stop=st, params=oracle.ridge_params(A, lam),
                                 per_problem=Y, reduction="butterfly_fma", width=64, fma_group=2)
    _assert_same(dev, twin, "cond=%g lambda=%g" % (cond, lam))
    # gram="auto" takes the normal-equation form only inside its pinned envelope (a rigorous upper bound of cond(H))
    auto = amd.SquaredErrorRidge(A, lam, gram="auto")
    inside = amd.engine.ridge_condition_bound(A, lam) <= amd.engine.GRAM_AUTO_MAX_CONDITION
    assert auto.name == ("squared_error_ridge_gram" if inside else "squared_error_ridge")
    assert E.ridge_hessian_condition(A, lam) <= amd.engine.ridge_condition_bound(A, lam) * (1 + 1e-12)
    if not inside:   # ... and beyond it runs the reference-order kernel: device == the exact twin
        se = gpu_solver_factory(m=10, stopping_progress=_engine_stop(st), arithmetic="exact")
        deve = _solve(se, auto, x0, per_problem=_to_dev(Y))
        twine = oracle.minimize_batch("squared_error_ridge", x0, m=10, stop=st, params=oracle.ridge_params(A, lam),
                                      per_problem=Y, reduction="butterfly", width=64)
        _assert_same(deve, twine, "exact cond=%g lambda=%g" % (cond, lam))
