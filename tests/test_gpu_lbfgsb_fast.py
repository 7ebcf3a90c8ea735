"""GPU tests of the relaxed-algebra L-BFGS-B kernels (csrc/lbfgsb_fast_kernel.hpp, MI355_ARITH_FMA on the
mi355_lbfgsb_* entry points): bit for bit against their CPU twin (oracle/lbfgsb_fast_oracle.hpp), and within the north
star's 1e-6 of the reference binary itself (oracle/_ref/libref.so travels with the tree)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6


def _torch():
    import torch
    return torch


def _to_dev(a):
    return _torch().from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    dst = capi.Stop()
    for name, _ in oracle_stop._fields_:
        setattr(dst, name, getattr(oracle_stop, name))
    return dst


def _tight(oracle):
    return oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8,
                            past=0)


def _solve(s, objective, x0):
    import cppnumericalsolvers_amd as amd
    x, f, g, p = s.minimize(objective, _to_dev(x0))
    _torch().cuda.synchronize()
    return x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)


def _assert_same(dev, twin, msg=""):
    xg, fg, gg, pg = dev
    xb, fb, gb, pb = twin
    np.testing.assert_array_equal(xg, xb, err_msg=msg)
    np.testing.assert_array_equal(fg, fb, err_msg=msg)
    np.testing.assert_array_equal(gg, gb, err_msg=msg)
    for k in ("status", "num_iterations", "nfev", "sum_k", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(pg[k], pb[k], err_msg=msg + " " + k)


@pytest.mark.parametrize("n,kind,boxed,m", [(32, "u2", True, 10), (64, "u2", True, 9), (20, "std", False, 10), (8, "u2", True, 9),
                                            (33, "u2", True, 10), (64, "std", True, 10),
                                            (32, "u2", True, 5), (32, "std", True, 5), (64, "u2", True, 5),
                                            (8, "u2", True, 5), (2, "u2", False, 5), (20, "std", False, 5),
                                            (32, "u2", True, 3), (20, "std", False, 1), (48, "u2", True, 4),
                                            (32, "u2", True, 8), (16, "u2", True, 6), (64, "u2", True, 7),
                                            (100, "u2", True, 5), (128, "std", True, 2), (17, "u2", True, 5)])
def test_fast_kernel_equals_its_twin(gpu_solver_factory, oracle, n, kind, boxed, m):
    """Rosenbrock in the box [-1.5, 0.8] (configs[4] shape and around it): the device equals the twin bit for bit — x*,
    f*, g*, status, iteration / evaluation counts, deltas — under the Lbfgsb default preset and the tight stop."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    B = 96
    x0 = amd.synthetic_x0_host(B, n, kind, seed=n * 3 + 1)
    lo = np.full(n, -1.5) if boxed else None
    hi = np.full(n, 0.8) if boxed else None
    for stop_o in (oracle.lbfgsb_default_stop(), _tight(oracle)):
        s = amd.BatchedLbfgsb(m=m, stopping_progress=_engine_stop(stop_o), context=base.ctx, arithmetic="fma")
        if boxed:
            s.SetBounds(lo, hi)
        dev = _solve(s, amd.Rosenbrock(), x0)
        assert s.last_arithmetic() == "fma"
        twin = oracle.lbfgsb_fast_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi)
        _assert_same(dev, twin, "n=%d m=%d" % (n, m))
        assert np.all(dev[3]["status"] != 1)
        if boxed:
            assert np.all(dev[0] <= 0.8) and np.all(dev[0] >= -1.5)
    xh, fh, gh, ph = s.minimize_host(amd.Rosenbrock(), x0[:5])
    np.testing.assert_array_equal(xh, dev[0][:5])


def test_fast_kernel_is_the_default_and_exact_stays_selectable(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    base = gpu_solver_factory()
    n = 32
    x0 = amd.synthetic_x0_host(64, n, "u2", seed=9)
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    st = _tight(oracle)
    out = {}
    for arith in ("default", "fma", "exact"):
        s = amd.BatchedLbfgsb(m=5, stopping_progress=_engine_stop(st), context=base.ctx, arithmetic=arith)
        s.SetBounds(lo, hi)
        out[arith] = _solve(s, amd.Rosenbrock(), x0)
        assert s.last_arithmetic() == ("exact" if arith == "exact" else "fma")
    np.testing.assert_array_equal(out["default"][0], out["fma"][0])
    exact_twin = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=5, stop=st, lower=lo, upper=hi, reduction="butterfly",
                                              width=32)
    np.testing.assert_array_equal(out["exact"][0], exact_twin[0])
    assert np.max(np.abs(out["fma"][0] - out["exact"][0])) <= TOL
    # where the relaxed kernels are not built the request is refused, and the default falls back to the exact ones
    s = amd.BatchedLbfgsb(m=5, context=base.ctx, arithmetic="fma", linesearch="hager_zhang")
    with pytest.raises(capi.EngineError):
        s.minimize(amd.Rosenbrock(), _to_dev(x0))
    x0w = amd.synthetic_x0_host(8, 100, "u2", seed=9)      # m = 10 above n = 64: no relaxed kernel
    s = amd.BatchedLbfgsb(m=10, context=base.ctx, arithmetic="fma")
    with pytest.raises(capi.EngineError):
        s.minimize(amd.Rosenbrock(), _to_dev(x0w))
    s = amd.BatchedLbfgsb(m=10, context=base.ctx)
    _solve(s, amd.Rosenbrock(), x0w)
    assert s.last_arithmetic() == "exact"
    s = amd.BatchedLbfgsb(m=10, context=base.ctx)          # ... and up to n = 64 the 32-lane relaxed kernel is the default
    _solve(s, amd.Rosenbrock(), x0)
    assert s.last_arithmetic() == "fma" and s.last_launch()["lanes_per_problem"] == 32


def test_fast_kernel_quadratic_corner_cases_and_ragged_batches(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    from test_oracle import LBFGSB_CORNER_CASES
    base = gpu_solver_factory()
    for name, (x0, lo, hi) in sorted(LBFGSB_CORNER_CASES.items()):
        s = amd.BatchedLbfgsb(m=5, context=base.ctx, arithmetic="fma")
        s.SetBounds(lo, hi)
        _assert_same(_solve(s, amd.Rosenbrock(), x0), oracle.lbfgsb_fast_minimize_batch("rosenbrock", x0, lower=lo, upper=hi),
                     name)
    a = np.linspace(1.0, 9.0, 12)
    x0 = amd.synthetic_x0_host(10, 12, "u2", seed=5)
    lo, hi = np.full(12, 0.25), np.full(12, 3.0)
    s = amd.BatchedLbfgsb(m=5, context=base.ctx, arithmetic="fma")
    s.SetBounds(lo, hi)
    dev = _solve(s, amd.DiagQuadratic(a, 1.0), x0)
    twin = oracle.lbfgsb_fast_minimize_batch("diag_quadratic", x0, lower=lo, upper=hi, params=np.concatenate([a, [1.0]]))
    _assert_same(dev, twin, "quadratic")
    np.testing.assert_array_equal(dev[0], np.full_like(dev[0], 0.25))
    # batches that do not fill the last wavefront, a single problem, an empty batch
    n = 32
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    for B in (1, 3, 5, 67, 130):
        x0 = amd.synthetic_x0_host(B, n, "u2", seed=B)
        s = amd.BatchedLbfgsb(m=5, context=base.ctx, arithmetic="fma")
        s.SetBounds(lo, hi)
        _assert_same(_solve(s, amd.Rosenbrock(), x0), oracle.lbfgsb_fast_minimize_batch("rosenbrock", x0, lower=lo, upper=hi),
                     "B=%d" % B)
    s = amd.BatchedLbfgsb(m=5, context=base.ctx, arithmetic="fma")
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(np.zeros((0, 4))))
    assert x.shape == (0, 4)


def test_fast_kernel_vs_reference_binary(gpu_solver_factory, oracle):
    """The direct link: device (relaxed algebra) against the reference's own Lbfgsb<F, 5 / 6> on the configs[4] shape."""
    import cppnumericalsolvers_amd as amd
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so not in the tree")
    base = gpu_solver_factory()
    n = 32
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    st = _tight(oracle)
    for m in (5, 6):
        x0 = amd.synthetic_x0_host(2048, n, "u2", seed=20260923 + m)
        s = amd.BatchedLbfgsb(m=m, stopping_progress=_engine_stop(st), context=base.ctx, arithmetic="fma")
        s.SetBounds(lo, hi)
        xg, fg, gg, pg = _solve(s, amd.Rosenbrock(), x0)
        xr, fr, gr, pr = ref_lib.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=st, lower=lo, upper=hi)
        assert np.max(np.abs(xg - xr)) <= TOL and np.max(np.abs(fg - fr)) <= TOL


def test_fast_kernel_whole_config5_batch(gpu_solver_factory, oracle):
    """configs[4] at its full size (262,144 x Rosenbrock-32 in [-1.5, 0.8]^32, m = 5): size-independent properties of
    every solution (feasible, projected gradient below the tolerance or the step test fired, an active bound, descent)
    + exact parity with the twin on a strided sample + 1e-6 against the reference binary on another."""
    import cppnumericalsolvers_amd as amd
    import ref_lib
    base = gpu_solver_factory()
    n, B = 32, 262144
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    st = _tight(oracle)
    s = amd.BatchedLbfgsb(m=5, stopping_progress=_engine_stop(st), context=base.ctx)
    s.SetBounds(lo, hi)
    x0 = amd.synthetic_x0_host(B, n, "u2")
    x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
    assert s.last_arithmetic() == "fma"
    assert np.all(np.isfinite(x)) and np.all(x <= 0.8) and np.all(x >= -1.5)
    assert np.all((p["status"] == 2) | (p["status"] == 4))
    pgrad = np.where((x <= -1.5) & (g > 0), 0.0, np.where((x >= 0.8) & (g < 0), 0.0, g))
    assert np.max(np.abs(pgrad)) <= 1e-5
    assert np.all(np.any((x == 0.8) | (x == -1.5), axis=1))   # the unconstrained minimiser (1, .., 1) is outside the box
    for b in range(0, B, 4099):                               # descent from the clipped start
        assert f[b] <= oracle.evaluate("rosenbrock", np.clip(x0[b], -1.5, 0.8))[0]
    idx = np.arange(0, B, 1024)
    twin = oracle.lbfgsb_fast_minimize_batch("rosenbrock", x0[idx], m=5, stop=st, lower=lo, upper=hi)
    np.testing.assert_array_equal(x[idx], twin[0])
    np.testing.assert_array_equal(f[idx], twin[1])
    np.testing.assert_array_equal(p["num_iterations"][idx], twin[3]["num_iterations"])
    if ref_lib.available():
        idr = np.arange(7, B, 128)
        xr, fr, gr, pr = ref_lib.lbfgsb_minimize_batch("rosenbrock", x0[idr], m=5, stop=st, lower=lo, upper=hi)
        assert np.max(np.abs(x[idr] - xr)) <= TOL and np.max(np.abs(f[idr] - fr)) <= TOL


def test_fast_kernel_every_problem_of_config5_vs_reference_binary(gpu_solver_factory, oracle):
    """the relaxed-algebra policy accepted at 1e-6 against `libref` on ALL 262,144 problems of
    configs[4] -- the reference's own Lbfgsb<F, 5> (oracle/_ref/libref.so, the pinned build, every host thread) on the
    same starts; x* and f* of every problem, no sample."""
    import os
    import cppnumericalsolvers_amd as amd
    import ref_lib
    if not ref_lib.available():
        pytest.skip("oracle/_ref/libref.so not in the tree")
    base = gpu_solver_factory()
    n, B = 32, 262144
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)
    st = _tight(oracle)
    s = amd.BatchedLbfgsb(m=5, stopping_progress=_engine_stop(st), context=base.ctx)
    s.SetBounds(lo, hi)
    x0 = amd.synthetic_x0_host(B, n, "u2")
    x, f, g, p = _solve(s, amd.Rosenbrock(), x0)
    assert s.last_arithmetic() == "fma"
    xr, fr, gr, pr = ref_lib.minimize_batch_threaded("rosenbrock", x0, m=5, stop=st, threads=os.cpu_count() or 8, chunk=64,
                                                     lower=lo, upper=hi)
    dx, df = np.max(np.abs(x - xr), axis=1), np.abs(f - fr)
    assert float(dx.max()) <= TOL and float(df.max()) <= TOL, (float(dx.max()), float(df.max()), int(np.argmax(dx)))
    assert np.all(pr["status"] != 1) and np.all(p["status"] != 1)
