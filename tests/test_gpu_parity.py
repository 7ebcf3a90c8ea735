"""GPU parity tests: the HIP engine (through the C-ABI) against the CPU oracle.

Bar (BASELINE.json north_star): x* and f* within 1e-6 of the reference
algorithm on the same inputs.  Because engine and oracle(butterfly) perform the
same IEEE operations in the same order, most checks below are in fact exact.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TOL = 1e-6  # north_star tolerance on x* and f*

MAPPINGS = [(8, 1), (8, 2), (8, 4), (16, 1), (16, 2), (16, 4), (32, 1), (32, 2), (32, 4),
            (64, 1), (64, 2), (64, 4)]


def _torch():
    import torch
    return torch


def _to_dev(a):
    torch = _torch()
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _copy_stop(dst, src):
    for name, _ in src._fields_:
        setattr(dst, name, getattr(src, name))
    return dst


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    return _copy_stop(capi.Stop(), oracle_stop)


# --------------------------------------------------------------------------
# hardware primitives
# --------------------------------------------------------------------------
def test_selftest_lane_maps_and_ieee(gpu_solver_factory):
    torch = _torch()
    s = gpu_solver_factory()
    from cppnumericalsolvers_amd import capi
    maps = torch.zeros(14 * 64, dtype=torch.int32, device="cuda:0")
    rng = np.random.default_rng(1)
    probe = np.concatenate([rng.uniform(1e-300, 1e300, 16), rng.uniform(0.5, 2.0, 32),
                            10.0 ** rng.uniform(-30, 30, 16)])
    pin = _to_dev(probe)
    pout = torch.zeros(128, dtype=torch.float64, device="cuda:0")
    capi.check(s.ctx._lib.mi355_lbfgs_selftest(s.ctx.handle, maps.data_ptr(), pin.data_ptr(),
                                               pout.data_ptr(), None))
    torch.cuda.synchronize()
    m = maps.cpu().numpy().reshape(14, 64)
    lane = np.arange(64)
    np.testing.assert_array_equal(m[0], lane ^ 1)
    np.testing.assert_array_equal(m[1], lane ^ 2)
    np.testing.assert_array_equal(m[2], (lane & ~7) | (7 - (lane & 7)))
    np.testing.assert_array_equal(m[3], (lane & ~15) | (15 - (lane & 15)))
    np.testing.assert_array_equal(m[4], lane ^ 16)
    np.testing.assert_array_equal(m[5], lane ^ 32)
    np.testing.assert_array_equal(m[6][:63], lane[:63] + 1)
    np.testing.assert_array_equal(m[7][1:], lane[1:] - 1)
    np.testing.assert_array_equal(m[8], (lane & ~15) | 3)      # row_newbcast:3
    np.testing.assert_array_equal(m[9], (lane & ~15) | 11)     # row_newbcast:11
    np.testing.assert_array_equal(m[10], (lane & ~31) | 3)     # lane 3 / 19 of a 32-lane segment
    np.testing.assert_array_equal(m[11], (lane & ~31) | 19)
    np.testing.assert_array_equal(m[12], lane ^ 16)            # v_permlane16_swap partner
    np.testing.assert_array_equal(m[13], lane & ~31)           # minimum over a 32-lane segment
    out = pout.cpu().numpy()
    np.testing.assert_array_equal(out[:64], np.sqrt(probe))   # correctly rounded sqrt
    np.testing.assert_array_equal(out[64:], 1.0 / probe)      # correctly rounded division


# --------------------------------------------------------------------------
# cstep: the reference's golden vectors (src/test/cstep_test.cc) on the device
# --------------------------------------------------------------------------
def _device_cstep(s, recs):
    torch = _torch()
    from cppnumericalsolvers_amd import capi
    r = _to_dev(np.asarray(recs, dtype=np.float64).reshape(-1, 13))
    ret = torch.zeros(r.shape[0], dtype=torch.int32, device="cuda:0")
    capi.check(s.ctx._lib.mi355_lbfgs_cstep_batch(s.ctx.handle, r.shape[0], r.data_ptr(),
                                                  ret.data_ptr(), None))
    torch.cuda.synchronize()
    return r.cpu().numpy(), ret.cpu().numpy()


def test_cstep_golden_vectors_on_device(gpu_solver_factory):
    from golden_cstep import CASES
    s = gpu_solver_factory()
    recs = [[c["stx"], c["fx"], c["dx"], c["sty"], c["fy"], c["dy"], c["stp"], c["fp"], c["dp"],
             float(c["brackt"]), c["stpmin"], c["stpmax"], 0.0] for c in CASES]
    out, ret = _device_cstep(s, recs)
    for c, o, rc in zip(CASES, out, ret):
        exp = c["expect"]
        assert rc == exp["rc"], c["name"]
        if rc != 0:
            continue
        assert int(o[12]) == exp["info"], c["name"]
        assert bool(o[9]) == exp["brackt"], c["name"]
        for key, col in (("stx", 0), ("fx", 1), ("dx", 2), ("sty", 3), ("fy", 4), ("dy", 5)):
            if key in exp:
                assert o[col] == exp[key], (c["name"], key)
        if "stp" in exp:
            assert abs(o[6] - exp["stp"]) <= exp.get("stp_tol", 0.0), c["name"]
        if "stp_le" in exp:
            assert o[6] <= exp["stp_le"], c["name"]
        if "stp_ge" in exp:
            assert o[6] >= exp["stp_ge"], c["name"]
        if "stp_gt" in exp:
            assert o[6] > exp["stp_gt"], c["name"]


def test_cstep_random_matches_oracle_bitwise(gpu_solver_factory, oracle):
    s = gpu_solver_factory()
    rng = np.random.default_rng(7)
    recs = []
    for _ in range(4096):
        stx = rng.uniform(0, 2)
        stp = stx + rng.uniform(0.01, 3) * rng.choice([1.0, 1.0, -0.3])
        dx = -np.sign(stp - stx) * rng.uniform(0.01, 5)
        fx = rng.normal()
        fp = fx + rng.normal() * 0.5
        dp = rng.normal() * 3
        brackt = rng.random() < 0.4
        sty = stp + np.sign(stp - stx) * rng.uniform(0.01, 2) if brackt else 0.0
        fy = fx + abs(rng.normal())
        dy = rng.normal()
        lo, hi = min(stx, sty if brackt else stx), max(stx, sty if brackt else stp * 5)
        recs.append([stx, fx, dx, sty, fy, dy, stp, fp, dp, float(brackt), lo, hi + 1.0, 0.0])
    out, ret = _device_cstep(s, recs)
    for r, o, rc in zip(recs, out, ret):
        e = oracle.cstep(*r[:9], bool(r[9]), r[10], r[11])
        assert rc == e["rc"]
        got = dict(stx=o[0], fx=o[1], dx=o[2], sty=o[3], fy=o[4], dy=o[5], stp=o[6])
        for k, v in got.items():
            assert v == e[k] or (np.isnan(v) and np.isnan(e[k])), (k, r)
        assert int(o[12]) == e["info"] and bool(o[9]) == e["brackt"]


# --------------------------------------------------------------------------
# objective functors
# --------------------------------------------------------------------------
@pytest.mark.parametrize("n", [2, 5, 8, 31, 32, 33, 64, 100, 256])
def test_rosenbrock_eval_bitwise(gpu_solver_factory, oracle, n):
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(n)
    X = rng.uniform(-2, 2, size=(37, n))
    width = 1 << max(3, int(np.ceil(np.log2(n))))
    for W, E in MAPPINGS:
        if W * E < n:
            continue
        s = gpu_solver_factory(lanes_per_problem=W, elems_per_lane=E)
        f, g = s.evaluate(amd.Rosenbrock(), _to_dev(X))
        f, g = f.cpu().numpy(), g.cpu().numpy()
        for b in range(X.shape[0]):
            fe, ge = oracle.evaluate("rosenbrock", X[b], reduction="butterfly", width=width)
            assert f[b] == fe, (W, E, b)
            np.testing.assert_array_equal(g[b], ge)


def test_diag_quadratic_eval_bitwise(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(3)
    for n in (2, 7, 32, 64, 90):
        a = rng.uniform(0.5, 100, n)
        X = rng.uniform(-5, 5, size=(9, n))
        params = np.concatenate([a, [5.0]])
        width = 1 << max(3, int(np.ceil(np.log2(n))))
        for W, E in MAPPINGS:
            if W * E < n:
                continue
            s = gpu_solver_factory(lanes_per_problem=W, elems_per_lane=E)
            f, g = s.evaluate(amd.DiagQuadratic(a, 5.0), _to_dev(X))
            f, g = f.cpu().numpy(), g.cpu().numpy()
            for b in range(X.shape[0]):
                fe, ge = oracle.evaluate("diag_quadratic", X[b], params=params, reduction="butterfly",
                                         width=width)
                assert f[b] == fe
                np.testing.assert_array_equal(g[b], ge)


def test_ridge_eval_bitwise(gpu_solver_factory, oracle):
    """config 4 objective: ||A x - y_b||^2 + lambda ||x||^2, bit-exact against the oracle twin."""
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(5)
    for rows, n in ((128, 64), (100, 50), (3, 2), (128, 20)):
        A, Y = amd.synthetic_ridge_host(11, rows, n, seed=rows * 7 + n)
        X = rng.normal(size=(11, n))
        params = oracle.ridge_params(A, 0.1)
        width = 1 << max(3, int(np.ceil(np.log2(n))))
        for W, E in MAPPINGS:
            if W * E < n:
                continue
            s = gpu_solver_factory(lanes_per_problem=W, elems_per_lane=E)
            if W * E > 128:   # the LDS copy of A (W*E x 129 doubles) does not fit: loud refusal
                with pytest.raises(amd.capi.EngineError):
                    s.evaluate(amd.SquaredErrorRidge(A, 0.1), _to_dev(X), per_problem=_to_dev(Y))
                continue
            f, g = s.evaluate(amd.SquaredErrorRidge(A, 0.1), _to_dev(X), per_problem=_to_dev(Y))
            f, g = f.cpu().numpy(), g.cpu().numpy()
            for b in range(X.shape[0]):
                fe, ge = oracle.evaluate("squared_error_ridge", X[b], params=params, reduction="butterfly",
                                         width=width, per_problem=Y[b:b + 1])
                assert f[b] == fe, (rows, n, W, E, b)
                np.testing.assert_array_equal(g[b], ge)


# --------------------------------------------------------------------------
# end-to-end solves
# --------------------------------------------------------------------------
def _solve_gpu(s, objective, x0):
    import cppnumericalsolvers_amd as amd
    x, f, g, p = s.minimize(objective, _to_dev(x0))
    _torch().cuda.synchronize()
    return x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)


def _assert_same_progress(pg, po):
    for k in ("status", "num_iterations", "nfev", "sum_k"):
        np.testing.assert_array_equal(pg[k], po[k], err_msg=k)
    for k in ("x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(pg[k], po[k], err_msg=k)


def test_quickstart_quadratic(gpu_solver_factory, oracle):
    """configs[0]: README.md:21-35 / Dockerfile.test:31-42 (|x*|<1e-4, |f*-5|<1e-4)."""
    import cppnumericalsolvers_amd as amd
    s = gpu_solver_factory(m=10)  # Lbfgs<Quadratic> defaults
    x0 = np.array([[-10.0, 2.0]])
    x, f, g, p = _solve_gpu(s, amd.DiagQuadratic([5.0, 100.0], 5.0), x0)
    assert np.all(np.abs(x) < 1e-4) and abs(f[0] - 5.0) < 1e-4
    xo, fo, go, po = oracle.minimize_batch("diag_quadratic", x0, m=10, params=[5, 100, 5],
                                           reduction="butterfly", width=8)
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_array_equal(f, fo)
    _assert_same_progress(p, po)
    # sequential order (what the reference computes at n=2) is the same tree at n = 2
    xs, fs, _, ps = oracle.minimize_batch("diag_quadratic", x0, m=10, params=[5, 100, 5])
    np.testing.assert_array_equal(x, xs)
    assert p["num_iterations"][0] == 10 and p["nfev"][0] == 11 and p["status"][0] == 4
    # host-pointer entry point gives the same answer
    xh, fh, gh, ph = s.minimize_host(amd.DiagQuadratic([5.0, 100.0], 5.0), x0)
    np.testing.assert_array_equal(xh, x)
    np.testing.assert_array_equal(fh, f)


@pytest.mark.parametrize("start", [(15.0, 8.0), (-1.0, 2.0), (-1.2, 1.0)])
def test_rosenbrock_2d_reference_fixtures(gpu_solver_factory, oracle, start):
    """src/test/verify.cc:188 LbfgsTest RosenbrockGradientFar/Near: |f(x*)| <= 1e-4."""
    import cppnumericalsolvers_amd as amd
    x0 = np.array([start])
    s = gpu_solver_factory(m=10)
    x, f, g, p = _solve_gpu(s, amd.Rosenbrock(), x0)
    assert abs(f[0]) <= 1e-4
    xo, fo, go, po = oracle.minimize_batch("rosenbrock", x0, m=10)
    np.testing.assert_array_equal(x, xo)   # n = 2: butterfly == sequential
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)
    _assert_same_progress(p, po)


@pytest.mark.parametrize("n,m,kind", [(32, 6, "std"), (32, 6, "u2"), (64, 10, "std"), (48, 10, "u2"),
                                      (100, 5, "std"), (20, 7, "u2"), (64, 3, "std")])
def test_batched_rosenbrock_parity_stop(gpu_solver_factory, oracle, n, m, kind):
    """configs[1]/[2] shapes at oracle-sized batches, 'parity stopping (B)'."""
    import cppnumericalsolvers_amd as amd
    B = 256
    x0 = amd.synthetic_x0_host(B, n, kind)
    width = 1 << max(3, int(np.ceil(np.log2(n))))
    stop_o = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o))
    x, f, g, p = _solve_gpu(s, amd.Rosenbrock(), x0)
    # (1) exact twin: oracle with the engine's summation tree
    xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly",
                                           width=width)
    np.testing.assert_array_equal(x, xb)
    np.testing.assert_array_equal(f, fb)
    np.testing.assert_array_equal(g, gb)
    _assert_same_progress(p, pb)
    # (2) the bar: within 1e-6 of the reference-order (sequential) solve
    xs, fs, gs, ps = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o)
    assert np.max(np.abs(x - xs)) <= TOL
    assert np.max(np.abs(f - fs)) <= TOL
    assert np.all(p["status"] != 1)  # nobody ran into the iteration limit


def test_default_stopping_matches_twin(gpu_solver_factory, oracle):
    """Reference default preset (plateau test, past=3): exact vs the twin; vs the
    sequential order only to the 1e-3 the reference's own tests use."""
    import cppnumericalsolvers_amd as amd
    n, m, B = 32, 6, 256
    x0 = amd.synthetic_x0_host(B, n, "u2")
    s = gpu_solver_factory(m=m)
    x, f, g, p = _solve_gpu(s, amd.Rosenbrock(), x0)
    xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, reduction="butterfly", width=32)
    np.testing.assert_array_equal(x, xb)
    np.testing.assert_array_equal(f, fb)
    _assert_same_progress(p, pb)
    for preset in ("conservative",):
        st = oracle.default_stop(preset)
        s2 = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st))
        x2, f2, g2, p2 = _solve_gpu(s2, amd.Rosenbrock(), x0)
        xb2, fb2, _, pb2 = oracle.minimize_batch("rosenbrock", x0, m=m, stop=st, reduction="butterfly",
                                                 width=32)
        np.testing.assert_array_equal(x2, xb2)
        _assert_same_progress(p2, pb2)


def test_ridge_solves_config4_shape(gpu_solver_factory, oracle):
    """configs[3] shape (A 128x64 shared, y_b per problem, lambda 0.1, x0 = 0, m = 10) at an
    oracle-sized batch: exact vs the twin, <= 1e-6 vs the sequential order and vs the closed form
    x* = (A^T A + lambda I)^-1 A^T y_b."""
    import cppnumericalsolvers_amd as amd
    B, rows, n, m, lam = 96, 128, 64, 10, 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n)
    x0 = np.zeros((B, n))
    params = oracle.ridge_params(A, lam)
    obj = amd.SquaredErrorRidge(A, lam)
    for stop_o in (oracle.parity_stop(), oracle.default_stop()):
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o))
        xg, fg, gg, pg = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
        _torch().cuda.synchronize()
        xg, fg, gg, pg = xg.cpu().numpy(), fg.cpu().numpy(), gg.cpu().numpy(), amd.progress_to_numpy(pg)
        xb, fb, gb, pb = oracle.minimize_batch("squared_error_ridge", x0, m=m, stop=stop_o, params=params,
                                               reduction="butterfly", width=64, per_problem=Y)
        np.testing.assert_array_equal(xg, xb)
        np.testing.assert_array_equal(fg, fb)
        np.testing.assert_array_equal(gg, gb)
        _assert_same_progress(pg, pb)
    # parity stop (first loop iteration result is overwritten: recompute)
    stop_o = oracle.parity_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o))
    xg, fg, gg, pg = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
    xg, fg = xg.cpu().numpy(), fg.cpu().numpy()
    xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge", x0, m=m, stop=stop_o, params=params,
                                         per_problem=Y)
    assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    assert np.max(np.abs(xg - closed)) <= TOL
    # host-pointer entry point
    xh, fh, gh, ph = s.minimize_host(obj, x0, per_problem=Y)
    np.testing.assert_array_equal(xh, xg)
    # mapping / placement invariance for this objective too
    for W, E, H in [(64, 1, 1), (32, 2, 1), (16, 4, 1), (16, 4, 2)]:
        s2 = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), lanes_per_problem=W,
                                elems_per_lane=E, history_placement=H)
        x2, f2, g2, p2 = s2.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
        np.testing.assert_array_equal(x2.cpu().numpy(), xg)


# --------------------------------------------------------------------------
# config 5: box-constrained L-BFGS-B
# --------------------------------------------------------------------------
def _lbfgsb(gpu_solver_factory, stop=None, m=5):
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    return amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=stop, context=base.ctx)


@pytest.mark.parametrize("n,kind,boxed,m", [(32, "u2", True, 5), (32, "std", True, 5), (64, "u2", True, 5),
                                            (8, "u2", True, 5), (2, "u2", False, 5), (20, "std", False, 5),
                                            (32, "u2", True, 3), (20, "std", False, 1), (48, "u2", True, 4)])
def test_lbfgsb_parity(gpu_solver_factory, oracle, n, kind, boxed, m):
    """configs[4] shape (Rosenbrock in the box [-1.5, 0.8], Lbfgsb m = 5): exact vs the oracle twin
    (butterfly reductions, index-ordered breakpoints), <= 1e-6 vs the reference-order solve."""
    import cppnumericalsolvers_amd as amd
    B = 96
    x0 = amd.synthetic_x0_host(B, n, kind, seed=n * 3 + 1)
    lo = np.full(n, -1.5) if boxed else None
    hi = np.full(n, 0.8) if boxed else None
    width = 1 << max(3, int(np.ceil(np.log2(n))))
    # (stopping, x / f tolerance vs the reference-order solve): the Lbfgsb default preset stops on
    # the relative f-delta / plateau tests well before x has settled (compare f at the 1e-4 of the
    # reference's own tests, x loosely); the tight "parity" stopping carries the 1e-6 bar.
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0,
                             gradient_norm=1e-8, past=0)
    # (with a single stored pair the default preset stops a factor of ten further from the minimiser)
    for stop_o, tol, ftol in ((oracle.lbfgsb_default_stop(), 5e-3 if m >= 3 else 5e-2, 1e-4 if m >= 3 else 1e-3),
                              (tight, TOL, TOL)):
        s = _lbfgsb(gpu_solver_factory, stop=_engine_stop(stop_o), m=m)
        if boxed:
            s.SetBounds(lo, hi)
        xg, fg, gg, pg = s.minimize(amd.Rosenbrock(), _to_dev(x0))
        _torch().cuda.synchronize()
        xg, fg, gg, pg = xg.cpu().numpy(), fg.cpu().numpy(), gg.cpu().numpy(), amd.progress_to_numpy(pg)
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                       reduction="butterfly", width=width)
        np.testing.assert_array_equal(xg, xb)
        np.testing.assert_array_equal(fg, fb)
        np.testing.assert_array_equal(gg, gb)
        _assert_same_progress(pg, pb)
        xs, fs, _, ps = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                     std_sort_order=True)
        assert np.max(np.abs(xg - xs)) <= tol and np.max(np.abs(fg - fs)) <= ftol
        assert np.all(pg["status"] != 1)
        if boxed:
            assert np.all(xg <= 0.8) and np.all(xg >= -1.5)
    # host-pointer entry point
    xh, fh, gh, ph = s.minimize_host(amd.Rosenbrock(), x0[:5])
    np.testing.assert_array_equal(xh, xg[:5])


def test_lbfgsb_corner_cases_on_device(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    from test_oracle import LBFGSB_CORNER_CASES
    for name, (x0, lo, hi) in sorted(LBFGSB_CORNER_CASES.items()):
        n = x0.shape[1]
        s = _lbfgsb(gpu_solver_factory)
        s.SetBounds(lo, hi)
        xg, fg, gg, pg = s.minimize(amd.Rosenbrock(), _to_dev(x0))
        _torch().cuda.synchronize()
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, lower=lo, upper=hi, reduction="butterfly",
                                                       width=max(8, 1 << int(np.ceil(np.log2(max(n, 2))))))
        np.testing.assert_array_equal(xg.cpu().numpy(), xb, err_msg=name)
        np.testing.assert_array_equal(fg.cpu().numpy(), fb, err_msg=name)
        _assert_same_progress(amd.progress_to_numpy(pg), pb)
    # a quadratic with the unconstrained minimiser outside the box
    a = np.linspace(1.0, 9.0, 12)
    x0 = amd.synthetic_x0_host(10, 12, "u2", seed=5)
    lo, hi = np.full(12, 0.25), np.full(12, 3.0)
    s = _lbfgsb(gpu_solver_factory)
    s.SetBounds(lo, hi)
    xg, fg, gg, pg = s.minimize(amd.DiagQuadratic(a, 1.0), _to_dev(x0))
    xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("diag_quadratic", x0, lower=lo, upper=hi, reduction="butterfly",
                                                   width=16, params=np.concatenate([a, [1.0]]))
    np.testing.assert_array_equal(xg.cpu().numpy(), xb)
    np.testing.assert_array_equal(xb, np.full_like(xb, 0.25))   # every coordinate ends on its lower bound
    _assert_same_progress(amd.progress_to_numpy(pg), pb)


def test_lbfgsb_reference_fixtures_on_device(gpu_solver_factory):
    """src/test/verify.cc:190 LbfgsbTest Far/Near: EXPECT_NEAR(0, f(x*), 1e-4)."""
    import cppnumericalsolvers_amd as amd
    s = _lbfgsb(gpu_solver_factory)
    x, f, g, p = s.minimize(amd.Rosenbrock(), _to_dev(np.array([[15.0, 8.0], [-1.0, 2.0]])))
    f = f.cpu().numpy()
    assert np.all(np.abs(f) <= 1e-4)
    from cppnumericalsolvers_amd import capi
    with pytest.raises(capi.EngineError):
        amd.BatchedLbfgsb(arithmetic="exact", m=11, context=s.ctx).minimize(amd.Rosenbrock(), _to_dev(np.zeros((1, 4))))   # built for m <= 10


def test_mapping_invariance(gpu_solver_factory):
    """Results do not depend on the (lanes_per_problem, elems_per_lane) mapping."""
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    n, m, B = 32, 6, 130   # ragged: B is not a multiple of the problems per wavefront
    x0 = amd.synthetic_x0_host(B, n, "std")
    ref = None
    # (W, E, history placement): 1 = both ring halves in LDS, 2 = y half in registers
    for W, E, H in [(32, 1, 1), (16, 2, 1), (16, 2, 2), (8, 4, 1), (8, 4, 2), (64, 1, 1), (32, 2, 2),
                    (64, 4, 2), (0, 0, 0)]:
        s = gpu_solver_factory(m=m, stopping_progress=amd.parity_stop(), lanes_per_problem=W,
                               elems_per_lane=E, history_placement=H)
        out = _solve_gpu(s, amd.Rosenbrock(), x0)
        if H == 2:
            assert s.last_launch()["y_columns_in_registers"] == m
        if ref is None:
            ref = out
        else:
            np.testing.assert_array_equal(out[0], ref[0])
            np.testing.assert_array_equal(out[1], ref[1])
            np.testing.assert_array_equal(out[2], ref[2])
            _assert_same_progress(out[3], ref[3])


def test_edge_cases(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    torch = _torch()
    s = gpu_solver_factory(m=10)
    # empty batch
    x, f, g, p = s.minimize(amd.Rosenbrock(), torch.empty(0, 4, dtype=torch.float64, device="cuda:0"))
    assert x.shape == (0, 4)
    # start at the minimiser: gradient 0 -> dginit = 0 >= 0 -> cvsrch returns at once (quirk Q1)
    x0 = np.ones((3, 6))
    xg, fg, gg, pg = _solve_gpu(s, amd.Rosenbrock(), x0)
    xo, fo, go, po = oracle.minimize_batch("rosenbrock", x0, m=10, reduction="butterfly", width=8)
    np.testing.assert_array_equal(xg, xo)
    _assert_same_progress(pg, po)
    assert np.all(pg["status"] == 2) and np.all(pg["num_iterations"] == 1)
    # iteration limit: strict '>' -> limit+1 steps (progress.h:212-216)
    st = oracle.make_stop(num_iterations=5, past=0, gradient_norm=0.0, x_delta=0.0)
    s2 = gpu_solver_factory(m=4, stopping_progress=_engine_stop(st))
    x0 = amd.synthetic_x0_host(5, 10, "u2")
    xg, fg, gg, pg = _solve_gpu(s2, amd.Rosenbrock(), x0)
    xo, fo, go, po = oracle.minimize_batch("rosenbrock", x0, m=4, stop=st, reduction="butterfly", width=16)
    assert np.all(pg["status"] == 1) and np.all(pg["num_iterations"] == 6)
    np.testing.assert_array_equal(xg, xo)
    _assert_same_progress(pg, po)
    # non-finite start: NaN objective -> "return current" path; must terminate
    x0 = np.full((2, 8), 1e200)
    xg, fg, gg, pg = _solve_gpu(s, amd.Rosenbrock(), x0)
    assert np.all(pg["status"] != 0)
    # argument validation
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=0).minimize(amd.Rosenbrock(), _to_dev(np.zeros((1, 4))))
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=4, lanes_per_problem=8, elems_per_lane=1).minimize(
            amd.Rosenbrock(), _to_dev(np.zeros((1, 9))))
    # (n > 256 runs on the workgroup kernel since round 3 -- tests/test_gpu_wide.py -- for Lbfgs in the exact arithmetic)
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(arithmetic="fma").minimize(amd.Rosenbrock(), _to_dev(np.zeros((1, 300))))


def test_fill_x0_matches_host_generator(gpu_solver_factory):
    import cppnumericalsolvers_amd as amd
    s = gpu_solver_factory()
    for kind in ("std", "u2"):
        d = s.fill_x0(1000, 32, kind, first_problem=12345).cpu().numpy()
        h = amd.synthetic_x0_host(1000, 32, kind, first_problem=12345)
        np.testing.assert_array_equal(d, h)


@pytest.mark.parametrize("n,width", [(6, 8), (20, 32), (40, 64)])
def test_non_finite_and_overflowing_starts_match_oracle(gpu_solver_factory, oracle, n, width):
    """NaN / inf coordinates and magnitudes whose powers overflow, through Lbfgs (both line searches), Lbfgsb and
    Bfgs under both presets: values, gradients, status, counts and deltas equal the twin's, NaNs included (the
    twin takes the reference's branches on these inputs: test_oracle.py)."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    x0 = oracle.hostile_starts(n)
    lo, hi = np.full(n, -1.5), np.full(n, 0.8)

    def same(dev, ora):
        for a, b in zip(dev[:3], ora[:3]):
            np.testing.assert_array_equal(a, b)
        _assert_same_progress(dev[3], ora[3])

    for stop_o in (oracle.default_stop(), oracle.parity_stop()):
        st = _engine_stop(stop_o)
        for ls in ("more_thuente", "hager_zhang"):
            s = amd.BatchedLbfgs(m=5, stopping_progress=st, linesearch=ls, context=base.ctx, arithmetic="exact")
            same(s.minimize_host(amd.Rosenbrock(), x0),
                 oracle.minimize_batch("rosenbrock", x0, m=5, stop=stop_o, reduction="butterfly", width=width, linesearch=ls))
        sb = amd.BatchedLbfgsb(arithmetic="exact", m=5, stopping_progress=st, context=base.ctx)
        sb.SetBounds(lo, hi)
        same(sb.minimize_host(amd.Rosenbrock(), x0),
             oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=5, stop=stop_o, lower=lo, upper=hi, reduction="butterfly",
                                          width=max(width, 16)))
        same(amd.BatchedBfgs(stopping_progress=st, context=base.ctx).minimize_host(amd.Rosenbrock(), x0),
             oracle.bfgs_minimize_batch("rosenbrock", x0, stop=stop_o, reduction="butterfly", width=width))


@pytest.mark.parametrize("box", sorted(["pinned", "all_pinned", "crossed", "infinite", "half_infinite", "tiny_box",
                                        "huge"]))
def test_lbfgsb_degenerate_boxes_match_oracle(gpu_solver_factory, oracle, box):
    """Pinned coordinates, an empty interval, infinite / huge bounds (the twin takes the reference's branches on these:
    test_oracle.py); n = 12 and n = 40 (one and four coordinates per lane).  NaN bounds are refused: the order of NaN
    breakpoints is undefined in the reference too."""
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    base = gpu_solver_factory()
    bad = amd.BatchedLbfgsb(arithmetic="exact", m=5, context=base.ctx)
    nan_lo, nan_hi = oracle.degenerate_boxes(6)["nan_bound"]
    with pytest.raises(ValueError):
        bad.SetBounds(nan_lo, nan_hi)
    bad._lower, bad._upper = _to_dev(nan_lo), _to_dev(nan_hi)     # past the Python check: the C-ABI refuses as well
    with pytest.raises(capi.EngineError):
        bad.minimize_host(amd.Rosenbrock(), np.zeros((2, 6)))
    for n, width in ((12, 16), (40, 64)):
        x0 = np.random.default_rng(5).uniform(-2, 2, (8, n))
        lo, hi = oracle.degenerate_boxes(n)[box]
        for stop_o in (oracle.lbfgsb_default_stop(), oracle.parity_stop()):
            s = amd.BatchedLbfgsb(arithmetic="exact", m=5, stopping_progress=_engine_stop(stop_o), context=base.ctx)
            s.SetBounds(lo, hi)
            xg, fg, gg, pg = s.minimize_host(amd.Rosenbrock(), x0)
            xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=5, stop=stop_o, lower=lo, upper=hi,
                                                           reduction="butterfly", width=width)
            np.testing.assert_array_equal(xg, xb)
            np.testing.assert_array_equal(fg, fb)
            np.testing.assert_array_equal(gg, gb)
            _assert_same_progress(pg, pb)


@pytest.mark.parametrize("case", ["iteration_limit_off", "everything_off_but_limit", "x_delta_needs_3",
                                  "x_delta_violations_0", "f_delta_abs", "f_delta_rel", "past8", "past1", "past_delta0",
                                  "grad_abs", "limit1"])
def test_stopping_field_edge_values_match_oracle(gpu_solver_factory, oracle, case):
    """Edge values of every stopping field through Lbfgs (register-history and LDS-ring kernels), Bfgs and Lbfgsb."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    stop_o = oracle.make_stop(**oracle.stopping_edge_cases()[case])
    st = _engine_stop(stop_o)
    lo, hi = np.full(10, -1.5), np.full(10, 0.8)
    x0 = np.random.default_rng(9).uniform(-1.2, 1.2, (6, 10))

    def same(dev, ora):
        for a, b in zip(dev[:3], ora[:3]):
            np.testing.assert_array_equal(a, b)
        _assert_same_progress(dev[3], ora[3])

    for m, placement in ((10, 0), (4, 1)):
        s = amd.BatchedLbfgs(m=m, stopping_progress=st, context=base.ctx, history_placement=placement, arithmetic="exact")
        same(s.minimize_host(amd.Rosenbrock(), x0),
             oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly", width=16))
    same(amd.BatchedBfgs(stopping_progress=st, context=base.ctx).minimize_host(amd.Rosenbrock(), x0),
         oracle.bfgs_minimize_batch("rosenbrock", x0, stop=stop_o, reduction="butterfly", width=16))
    sb = amd.BatchedLbfgsb(arithmetic="exact", m=5, stopping_progress=st, context=base.ctx)
    sb.SetBounds(lo, hi)
    same(sb.minimize_host(amd.Rosenbrock(), x0),
         oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=5, stop=stop_o, lower=lo, upper=hi, reduction="butterfly", width=16))


@pytest.mark.parametrize("case", ["one_row", "underdetermined", "lambda0_singular", "zero_column", "huge_rhs",
                                  "nonfinite_rhs", "zero_matrix", "full_tile"])
def test_ridge_degenerate_data_match_oracle(gpu_solver_factory, oracle, case):
    """Degenerate ridge data through both ridge kernels (exact-order VALU, matrix cores), First and Second mode."""
    import cppnumericalsolvers_amd as amd
    A, lam, Y = oracle.degenerate_ridge_data()[case]
    n = A.shape[1]
    x0 = np.zeros((Y.shape[0], n))
    params = oracle.ridge_params(A, lam)
    for second in (False, True):
        mode = "second" if second else "first"
        for stop_o in (oracle.default_stop(), oracle.parity_stop()):
            s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(stop_o))
            for cores, twin in ((False, "squared_error_ridge"), (True, "squared_error_ridge_mfma")):
                obj = amd.SquaredErrorRidge(A, lam, differentiability=mode, matrix_cores=cores)
                xg, fg, gg, pg = s.minimize_host(obj, x0, per_problem=Y)
                xb, fb, gb, pb = oracle.minimize_batch(twin, x0, m=10, stop=stop_o, params=params, reduction="butterfly",
                                                       width=64, per_problem=Y, second_mode=second)
                np.testing.assert_array_equal(xg, xb, err_msg=twin)
                np.testing.assert_array_equal(fg, fb, err_msg=twin)
                np.testing.assert_array_equal(gg, gb, err_msg=twin)
                _assert_same_progress(pg, pb)


@pytest.mark.parametrize("n", [1, 3, 256])
def test_smallest_and_largest_dimensions_and_histories(gpu_solver_factory, oracle, n):
    """n = 1 (an objective without terms), n = 3 and the largest n; history sizes 1, 10 and the maximum 32; both line
    searches; Lbfgsb and Bfgs at n = 1 and 3.  (n = 255 and the combinations left out at n = 256 pass as well; they are
    trimmed here for the twin's run time.)"""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    width = 1 << max(3, int(np.ceil(np.log2(n))))
    big = n > 64
    x0 = np.random.default_rng(n).uniform(-1.2, 1.2, (2 if big else 5, n))
    a = np.linspace(0.5, 3.0, n)

    def same(dev, ora):
        for u, v in zip(dev[:3], ora[:3]):
            np.testing.assert_array_equal(u, v)
        _assert_same_progress(dev[3], ora[3])

    for stop_o in ((oracle.default_stop(),) if big else (oracle.default_stop(), oracle.parity_stop())):
        st = _engine_stop(stop_o)
        for m in ((1, 32) if big else (1, 10, 32)):
            for ls in ("more_thuente", "hager_zhang"):
                s = amd.BatchedLbfgs(m=m, stopping_progress=st, linesearch=ls, context=base.ctx, arithmetic="exact")
                same(s.minimize_host(amd.Rosenbrock(), x0),
                     oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly", width=width, linesearch=ls))
                if not big:
                    same(s.minimize_host(amd.DiagQuadratic(a, 0.25), x0),
                         oracle.minimize_batch("diag_quadratic", x0, m=m, stop=stop_o, params=np.concatenate([a, [0.25]]),
                                               reduction="butterfly", width=width, linesearch=ls))
        if n <= 64:
            lo, hi = np.full(n, -1.5), np.full(n, 0.8)
            for m in (1, 5):
                sb = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=st, context=base.ctx)
                sb.SetBounds(lo, hi)
                same(sb.minimize_host(amd.Rosenbrock(), x0),
                     oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                  reduction="butterfly", width=max(width, 16)))
            same(amd.BatchedBfgs(stopping_progress=st, context=base.ctx).minimize_host(amd.Rosenbrock(), x0),
                 oracle.bfgs_minimize_batch("rosenbrock", x0, stop=stop_o, reduction="butterfly", width=width))


def test_full_size_config1_properties(gpu_solver_factory, oracle):
    """configs[1] at full size (B=65536, n=32, m=6): size-independent properties
    + exact parity on a strided sample of 512 problems."""
    import cppnumericalsolvers_amd as amd
    torch = _torch()
    B, n, m = 65536, 32, 6
    s = gpu_solver_factory(m=m, stopping_progress=amd.parity_stop())
    x0 = s.fill_x0(B, n, "std")
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    xh, fh, gh = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
    assert np.all(pn["status"] >= 2) and np.all(pn["status"] <= 4)
    assert np.all(np.isfinite(xh)) and np.all(np.isfinite(fh))
    # stationarity: the returned gradient IS the gradient at the returned point, and it is small
    fe, ge = s.evaluate(amd.Rosenbrock(), x)
    np.testing.assert_array_equal(fe.cpu().numpy(), fh)
    np.testing.assert_array_equal(ge.cpu().numpy(), gh)
    assert np.max(np.abs(gh)) < 1e-5
    assert np.all(fh <= amd_f0(s, x0) + 0.0)  # monotone: never worse than the start
    # idempotence: restarting from x* stops within a couple of iterations at the same point
    x2, f2, g2, p2 = s.minimize(amd.Rosenbrock(), x)
    torch.cuda.synchronize()
    assert np.max(np.abs(x2.cpu().numpy() - xh)) < 1e-6
    p2n = amd.progress_to_numpy(p2)["num_iterations"]
    assert np.median(p2n) <= 3 and np.max(p2n) < np.max(pn["num_iterations"])
    # exact parity on a sample
    idx = np.arange(0, B, 128)
    x0h = x0.cpu().numpy()[idx]
    xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=oracle.parity_stop(),
                                           reduction="butterfly", width=32)
    np.testing.assert_array_equal(xh[idx], xb)
    np.testing.assert_array_equal(fh[idx], fb)
    _assert_same_progress(pn[idx], pb)
    xs, fs, _, _ = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=oracle.parity_stop())
    assert np.max(np.abs(xh[idx] - xs)) <= TOL and np.max(np.abs(fh[idx] - fs)) <= TOL


def test_full_size_config2_shard_properties(gpu_solver_factory, oracle):
    """configs[2] per-GPU shard at full size (B=131072 of the 1,048,576, n=64, m=10): the same size-independent
    properties + exact parity on a strided sample of 256 problems, taken from the END of the global batch
    (the shard of rank 7) so that the counter-based start points beyond the first shard are covered."""
    import cppnumericalsolvers_amd as amd
    torch = _torch()
    B, n, m, first = 131072, 64, 10, 7 * 131072
    s = gpu_solver_factory(m=m, stopping_progress=amd.parity_stop())
    x0 = s.fill_x0(B, n, "std", first_problem=first)
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    xh, fh, gh = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
    assert np.all(pn["status"] >= 2) and np.all(pn["status"] <= 4)
    assert np.all(np.isfinite(xh)) and np.all(np.isfinite(fh))
    fe, ge = s.evaluate(amd.Rosenbrock(), x)
    np.testing.assert_array_equal(fe.cpu().numpy(), fh)
    np.testing.assert_array_equal(ge.cpu().numpy(), gh)
    assert np.max(np.abs(gh)) < 1e-5
    assert np.all(fh <= amd_f0(s, x0))
    x2, f2, g2, p2 = s.minimize(amd.Rosenbrock(), x)
    torch.cuda.synchronize()
    assert np.max(np.abs(x2.cpu().numpy() - xh)) < 1e-6
    assert np.median(amd.progress_to_numpy(p2)["num_iterations"]) <= 3
    idx = np.arange(0, B, 512)
    x0h = amd.synthetic_x0_host(B, n, "std", first_problem=first)[idx]     # the host generator, not a copy
    np.testing.assert_array_equal(x0.cpu().numpy()[idx], x0h)
    xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=oracle.parity_stop(),
                                           reduction="butterfly", width=64)
    np.testing.assert_array_equal(xh[idx], xb)
    np.testing.assert_array_equal(fh[idx], fb)
    _assert_same_progress(pn[idx], pb)
    xs, fs, _, _ = oracle.minimize_batch("rosenbrock", x0h, m=m, stop=oracle.parity_stop())
    assert np.max(np.abs(xh[idx] - xs)) <= TOL and np.max(np.abs(fh[idx] - fs)) <= TOL


def test_full_size_config3_ridge_properties(gpu_solver_factory, oracle):
    """configs[3] at full size (262144 ridge problems, A 128x64, lambda 0.1, x0 = 0, m=10, matrix cores): every
    problem against the closed form (A^T A + lambda I)^-1 A^T y_b, the gradient identity 2 A^T (A x - y) + 2 lambda x,
    and exact parity with the twin on a strided sample."""
    import cppnumericalsolvers_amd as amd
    torch = _torch()
    B, rows, n, m, lam = 262144, 128, 64, 10, 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n)
    obj = amd.SquaredErrorRidge(A, lam, matrix_cores=True)
    s = gpu_solver_factory(m=m, stopping_progress=amd.parity_stop())
    x, f, g, p = s.minimize(obj, torch.zeros(B, n, dtype=torch.float64, device="cuda:0"), per_problem=_to_dev(Y))
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    xh, fh, gh = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
    assert np.all(pn["status"] >= 2) and np.all(pn["status"] <= 4)
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    assert np.max(np.abs(xh - closed)) <= TOL
    r = xh @ A.T - Y
    f_closed = np.einsum("ij,ij->i", r, r) + lam * np.einsum("ij,ij->i", xh, xh)
    assert np.max(np.abs(fh - f_closed)) <= 1e-9 * np.max(f_closed)
    assert np.max(np.abs(gh - (2.0 * r @ A + 2.0 * lam * xh))) < 1e-9
    assert np.max(np.abs(gh)) < 1e-6
    idx = np.arange(0, B, 1024)
    xb, fb, gb, pb = oracle.minimize_batch("squared_error_ridge_mfma", np.zeros((idx.size, n)), m=m,
                                           stop=oracle.parity_stop(), params=oracle.ridge_params(A, lam),
                                           reduction="butterfly", width=64, per_problem=Y[idx])
    np.testing.assert_array_equal(xh[idx], xb)
    np.testing.assert_array_equal(fh[idx], fb)
    _assert_same_progress(pn[idx], pb)


def test_full_size_config4_box_properties(gpu_solver_factory, oracle):
    """configs[4] at full size (262144 x Rosenbrock-32 in [-1.5, 0.8]^32, Lbfgsb m=5, start 'u2'): feasibility of every
    returned point, a vanishing projected gradient, an active bound, descent from the (clipped) start, idempotence,
    and exact parity with the twin on a strided sample."""
    import cppnumericalsolvers_amd as amd
    import bench
    torch = _torch()
    B, n, m, lo, hi = 262144, 32, 5, -1.5, 0.8
    base = gpu_solver_factory()
    stop = bench.lbfgsb_tight_stop(amd.capi.default_stop("lbfgsb"))
    s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=stop, context=base.ctx)
    s.SetBounds(np.full(n, lo), np.full(n, hi))
    x0 = s.fill_x0(B, n, "u2")
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    pn = amd.progress_to_numpy(p)
    xh, fh, gh = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
    assert np.all(pn["status"] >= 2) and np.all(pn["status"] <= 4)
    assert np.all(xh >= lo) and np.all(xh <= hi)
    pg = np.where((xh <= lo) & (gh > 0), 0.0, np.where((xh >= hi) & (gh < 0), 0.0, gh))
    assert np.max(np.abs(pg)) < 1e-5                                     # Lbfgsb::ProjectedGradientInfNorm
    assert np.all(np.any(xh == hi, axis=1))                              # the unconstrained minimiser x = 1 is infeasible
    f0, _ = s.evaluate(amd.Rosenbrock(), torch.clamp(x0, lo, hi))
    assert np.all(fh <= f0.cpu().numpy())
    x2, f2, g2, p2 = s.minimize(amd.Rosenbrock(), x)
    torch.cuda.synchronize()
    assert np.max(np.abs(x2.cpu().numpy() - xh)) < 1e-6
    assert np.median(amd.progress_to_numpy(p2)["num_iterations"]) <= 3
    idx = np.arange(0, B, 1024)
    stop_o = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0,
                              gradient_norm=1e-8, past=0)
    xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0.cpu().numpy()[idx], m=m, stop=stop_o,
                                                   lower=np.full(n, lo), upper=np.full(n, hi),
                                                   reduction="butterfly", width=32)
    np.testing.assert_array_equal(xh[idx], xb)
    np.testing.assert_array_equal(fh[idx], fb)
    _assert_same_progress(pn[idx], pb)


def amd_f0(s, x0):
    import cppnumericalsolvers_amd as amd
    f0, _ = s.evaluate(amd.Rosenbrock(), x0)
    return f0.cpu().numpy()


def test_ridge_second_mode_preconditioned_path(gpu_solver_factory, oracle):
    """Second-mode functions take the diagonal-preconditioner branch of
    Lbfgs (lbfgs.h:116-139, :177-179).  Device == twin bit for bit; <= 1e-6 vs the sequential
    order, which test_oracle pins to the reference's README functors; != the First-mode path."""
    import cppnumericalsolvers_amd as amd
    lam = 0.1
    cases = [(np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]), np.tile(np.array([7.0, 8.0, 9.0]), (5, 1)))]
    for rows, n, B in ((128, 64, 70), (50, 20, 33)):
        cases.append(amd.synthetic_ridge_host(B, rows, n, seed=rows))
    for A, Y in cases:
        B, n = Y.shape[0], A.shape[1]
        x0 = np.zeros((B, n))
        params = oracle.ridge_params(A, lam)
        obj2 = amd.SquaredErrorRidge(A, lam, differentiability="second")
        obj1 = amd.SquaredErrorRidge(A, lam)
        P = 8
        while P < n:
            P *= 2
        for stop_o in (oracle.default_stop(), oracle.parity_stop()):
            s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(stop_o))
            xg, fg, gg, pg = s.minimize(obj2, _to_dev(x0), per_problem=_to_dev(Y))
            xg, fg, gg, pg = xg.cpu().numpy(), fg.cpu().numpy(), gg.cpu().numpy(), amd.progress_to_numpy(pg)
            xb, fb, gb, pb = oracle.minimize_batch("squared_error_ridge", x0, m=10, stop=stop_o, params=params,
                                                   reduction="butterfly", width=P, per_problem=Y,
                                                   second_mode=True)
            np.testing.assert_array_equal(xg, xb)
            np.testing.assert_array_equal(fg, fb)
            np.testing.assert_array_equal(gg, gb)
            _assert_same_progress(pg, pb)
            x1, _, _, _ = s.minimize(obj1, _to_dev(x0), per_problem=_to_dev(Y))
            assert not np.array_equal(x1.cpu().numpy(), xg)
        xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge", x0, m=10, stop=oracle.parity_stop(),
                                             params=params, per_problem=Y, second_mode=True)
        assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
        closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
        assert np.max(np.abs(xg - closed)) <= TOL
        # host-pointer entry point and explicit mappings give the same bits
        xh, _, _, _ = s.minimize_host(obj2, x0, per_problem=Y)
        np.testing.assert_array_equal(xh, xg)
    # L-BFGS-B never uses second-order information (lbfgsb.h:48-49): explicit refusal
    sb = amd.BatchedLbfgsb(arithmetic="exact", m=5)
    with pytest.raises(amd.capi.EngineError):
        sb.minimize(obj2, _to_dev(x0), per_problem=_to_dev(Y))


def test_history_sizes_between_the_built_register_variants(gpu_solver_factory, oracle):
    """The register-history kernels are built for 5, 6 and 10 columns and serve every m up to their size
    (m = 7..9 on the 10-column kernel, m <= 4 on the 5-column one): bit-identical to the twin, and the
    library does pick them (y_columns_in_registers > 0)."""
    import cppnumericalsolvers_amd as amd
    for n, ms in ((32, (1, 3, 4, 7, 8, 9)), (64, (2, 8)), (100, (9,))):
        x0 = amd.synthetic_x0_host(41, n, seed=n)
        P = 8
        while P < n:
            P *= 2
        for m in ms:
            for stop_o in (oracle.default_stop(), oracle.parity_stop()):
                s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o))
                xg, fg, gg, pg = _solve_gpu(s, amd.Rosenbrock(), x0)
                assert s.last_launch()["y_columns_in_registers"] in (5, 10)
                xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly", width=P)
                np.testing.assert_array_equal(xg, xb)
                np.testing.assert_array_equal(fg, fb)
                np.testing.assert_array_equal(gg, gb)
                _assert_same_progress(pg, pb)


def test_register_scalar_kernels_with_plateau_ring_in_global_scratch(gpu_solver_factory, oracle):
    """Kernels that keep rho / alpha in registers hold the plateau ring of the stopping test (progress.h:139-140)
    in global scratch instead of LDS.  Exercised with past = 1..8 on ragged batches that refill segments in
    place (the ring must be re-initialised per problem): bit-identical to the twin."""
    import cppnumericalsolvers_amd as amd
    for n, m, B in ((32, 6, 203), (64, 10, 77), (100, 5, 37), (256, 10, 9), (200, 6, 5)):
        x0 = amd.synthetic_x0_host(B, n, seed=n + m)
        P = 8
        while P < n:
            P *= 2
        for past in (1, 3, 8):
            stop_o = oracle.default_stop()
            stop_o.past = past
            stop_o.past_delta = 1e-6
            s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o))
            xg, fg, gg, pg = _solve_gpu(s, amd.Rosenbrock(), x0)
            assert s.last_launch()["elems_per_lane"] == 4
            xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly", width=P)
            np.testing.assert_array_equal(xg, xb)
            np.testing.assert_array_equal(fg, fb)
            _assert_same_progress(pg, pb)


# ---- Hager-Zhang line search -----------------------------------------
def test_hz_search_device_vs_twin_bitwise(gpu_solver_factory, oracle):
    """Stand-alone HagerZhang::Search on the device (one state machine per wavefront segment) == the
    oracle in the device's reduction order, bit for bit, on the inputs of the committed reference
    vectors (every stage of hzls: 1 to > 150 evaluations, failures, non-finite trial points) — and
    <= 1e-6 on the accepted step against the reference's own outputs wherever both accept."""
    import cppnumericalsolvers_amd as amd
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hager_zhang_reference_vectors.npz"))
    s = gpu_solver_factory(m=6)
    for n in (2, 8, 32):
        k = "search_n%d" % n
        x, d, a0 = G[k + ".x"], G[k + ".s"], G[k + ".alpha_init"]
        P = 8
        while P < n:
            P *= 2
        xb, fb, gb, ab, nb = oracle.hz_search("rosenbrock", x, d, a0, reduction="butterfly", width=P)
        for W, E in ((0, 0), (64, 1), (8, 4) if n <= 32 else (16, 4)):
            sv = gpu_solver_factory(m=6, lanes_per_problem=W, elems_per_lane=E)
            xg, fg, gg, ag, ng = sv.hz_search(amd.Rosenbrock(), _to_dev(x), _to_dev(d), _to_dev(a0))
            _torch().cuda.synchronize()
            np.testing.assert_array_equal(ag.cpu().numpy(), ab)
            np.testing.assert_array_equal(xg.cpu().numpy(), xb)
            np.testing.assert_array_equal(fg.cpu().numpy(), fb)
            np.testing.assert_array_equal(gg.cpu().numpy(), gb)
            np.testing.assert_array_equal(ng.cpu().numpy().astype(np.uint64), nb)
        # against the reference's outputs (sequential summation order).  A line-search step is not a
        # continuous function of rounding (a secant through two nearly equal slopes), so single
        # records may differ visibly; the bulk must agree to rounding level.
        ar = G[k + ".alpha_out"]
        _, _, _, _, ns = oracle.hz_search("rosenbrock", x, d, a0)
        same = (nb == ns) & (ab > 0) & (ar > 0)
        assert same.mean() > 0.5
        rel = np.abs(ab[same] - ar[same]) / np.maximum(1e-300, np.abs(ar[same]))
        assert np.median(rel) <= 1e-9 and np.mean(rel <= TOL) > 0.8


def test_lbfgs_with_hager_zhang_solves(gpu_solver_factory, oracle):
    """Lbfgs<F, m, HagerZhang> (lbfgs.h:41): device == twin bit for bit (x*, f*, g*, status, iterations,
    evaluations) under the default and parity presets; <= 1e-6 against the reference-order solve under
    parity stopping; and == the committed reference vectors within that tolerance."""
    import cppnumericalsolvers_amd as amd
    for n, m, B in ((2, 10, 40), (32, 6, 96), (64, 10, 48), (100, 5, 17)):
        x0 = amd.synthetic_x0_host(B, n, seed=3 * n + m)
        P = 8
        while P < n:
            P *= 2
        for stop_o in (oracle.default_stop(), oracle.parity_stop()):
            s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), linesearch="hager_zhang")
            xg, fg, gg, pg = _solve_gpu(s, amd.Rosenbrock(), x0)
            xb, fb, gb, pb = oracle.minimize_batch("rosenbrock", x0, m=m, stop=stop_o, reduction="butterfly",
                                                   width=P, linesearch="hager_zhang")
            np.testing.assert_array_equal(xg, xb)
            np.testing.assert_array_equal(fg, fb)
            np.testing.assert_array_equal(gg, gb)
            _assert_same_progress(pg, pb)
        xs, fs, _, ps = oracle.minimize_batch("rosenbrock", x0, m=m, stop=oracle.parity_stop(),
                                              linesearch="hager_zhang")
        assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
        assert np.all(pg["status"] >= 2)
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hager_zhang_reference_vectors.npz"))
    name = "hz_rosen32_m6_parity"
    s = gpu_solver_factory(m=6, stopping_progress=_engine_stop(oracle.parity_stop()), linesearch="hager_zhang")
    xg, fg, _, _ = _solve_gpu(s, amd.Rosenbrock(), G[name + ".x0"])
    assert np.max(np.abs(xg - G[name + ".x"])) <= TOL and np.max(np.abs(fg - G[name + ".f"])) <= TOL
    # ridge objective with this line search, and the explicit refusal of L-BFGS-B
    A, Y = amd.synthetic_ridge_host(24, 50, 20, seed=5)
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(oracle.parity_stop()), linesearch="hager_zhang")
    xr, fr, gr, pr = s.minimize(amd.SquaredErrorRidge(A, 0.1), _to_dev(np.zeros((24, 20))), per_problem=_to_dev(Y))
    xo, fo, _, po = oracle.minimize_batch("squared_error_ridge", np.zeros((24, 20)), m=10, stop=oracle.parity_stop(),
                                          params=oracle.ridge_params(A, 0.1), per_problem=Y, reduction="butterfly",
                                          width=32, linesearch="hager_zhang")
    np.testing.assert_array_equal(xr.cpu().numpy(), xo)
    closed = np.linalg.solve(A.T @ A + 0.1 * np.eye(20), A.T @ Y.T).T
    assert np.max(np.abs(xo - closed)) <= TOL


def test_lbfgsb_with_hager_zhang(gpu_solver_factory, oracle):
    """`Lbfgsb<F, m, HagerZhang>` — the drop-in use hager_zhang.h:39-42 advertises: device == twin bit for bit,
    <= 1e-6 against the reference-order solve under tight stopping (the oracle is pinned to the reference's
    Lbfgsb<F, m, HagerZhang> in test_oracle)."""
    import cppnumericalsolvers_amd as amd
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0,
                             gradient_norm=1e-8, past=0)
    base = gpu_solver_factory()
    for n, m, boxed in ((32, 5, True), (8, 3, True), (16, 5, False), (64, 5, True)):
        x0 = amd.synthetic_x0_host(48, n, "u2", seed=n)
        lo = np.full(n, -1.5) if boxed else None
        hi = np.full(n, 0.8) if boxed else None
        width = 1 << max(3, int(np.ceil(np.log2(n))))
        for stop_o in (oracle.lbfgsb_default_stop(), tight):
            s = amd.BatchedLbfgsb(arithmetic="exact", m=m, stopping_progress=_engine_stop(stop_o), context=base.ctx, linesearch="hager_zhang")
            if boxed:
                s.SetBounds(lo, hi)
            xg, fg, gg, pg = s.minimize(amd.Rosenbrock(), _to_dev(x0))
            _torch().cuda.synchronize()
            xg, fg, gg, pg = xg.cpu().numpy(), fg.cpu().numpy(), gg.cpu().numpy(), amd.progress_to_numpy(pg)
            xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=stop_o, lower=lo, upper=hi,
                                                           reduction="butterfly", width=width, linesearch="hager_zhang")
            np.testing.assert_array_equal(xg, xb)
            np.testing.assert_array_equal(fg, fb)
            np.testing.assert_array_equal(gg, gb)
            _assert_same_progress(pg, pb)
        xs, fs, _, _ = oracle.lbfgsb_minimize_batch("rosenbrock", x0, m=m, stop=tight, lower=lo, upper=hi,
                                                     std_sort_order=True, linesearch="hager_zhang")
        assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL


# ---- ridge objective on the matrix cores (config 4: "objective GEMV on MFMA") ---------------------
def test_ridge_matrix_core_kernel(gpu_solver_factory, oracle):
    """Objective id 3: workgroups of sixteen problem slots, one joint objective evaluation per pass on
    v_mfma_f64_16x16x4_f64, Moré–Thuente as a state machine around it.  Bit for bit the oracle twin
    (FMA-chain matrix-vector products, butterfly reductions) — x*, f*, g*, status, iterations,
    evaluations — for full and ragged batches, rows/n below the tile sizes, both stopping presets,
    history sizes up to 10 and the Second-mode preconditioner; within 1e-6 of the reference-order solve
    (README functors, multiply-then-add sums) and of the closed form."""
    import cppnumericalsolvers_amd as amd
    lam = 0.1
    for rows, n, B, m in ((128, 64, 70, 10), (128, 64, 16, 10), (50, 20, 37, 10), (3, 2, 5, 10), (128, 64, 33, 6),
                          (100, 64, 19, 3)):
        if rows == 3:
            A = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
            Y = np.tile(np.array([7.0, 8.0, 9.0]), (B, 1))
        else:
            A, Y = amd.synthetic_ridge_host(B, rows, n, seed=rows + n + B)
        x0 = np.zeros((B, n))
        params = oracle.ridge_params(A, lam)
        for second in (False, True):
            obj = amd.SquaredErrorRidge(A, lam, differentiability="second" if second else "first", matrix_cores=True)
            for stop_o in (oracle.default_stop(), oracle.parity_stop()):
                s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o))
                xg, fg, gg, pg = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
                _torch().cuda.synchronize()
                xg, fg, gg, pg = xg.cpu().numpy(), fg.cpu().numpy(), gg.cpu().numpy(), amd.progress_to_numpy(pg)
                xb, fb, gb, pb = oracle.minimize_batch("squared_error_ridge_mfma", x0, m=m, stop=stop_o, params=params,
                                                       reduction="butterfly", width=64, per_problem=Y,
                                                       second_mode=second)
                np.testing.assert_array_equal(xg, xb)
                np.testing.assert_array_equal(fg, fb)
                np.testing.assert_array_equal(gg, gb)
                _assert_same_progress(pg, pb)
            ll = s.last_launch()
            assert ll["threads"] == 512 and ll["lanes_per_problem"] == 32 and ll["elems_per_lane"] == 2
            # the other mapping of the sixteen slots (four wavefronts x four problems): the same bits
            s4 = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), lanes_per_problem=16, elems_per_lane=4)
            x4, f4, g4, p4 = s4.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
            _torch().cuda.synchronize()
            np.testing.assert_array_equal(x4.cpu().numpy(), xg)
            np.testing.assert_array_equal(f4.cpu().numpy(), fg)
            _assert_same_progress(amd.progress_to_numpy(p4), pg)
            l4 = s4.last_launch()
            assert l4["threads"] == 256 and l4["lanes_per_problem"] == 16 and l4["elems_per_lane"] == 4
            # parity stopping: against the reference's arithmetic (sequential, multiply-then-add) and the closed form
            xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge", x0, m=m, stop=oracle.parity_stop(),
                                                 params=params, per_problem=Y, second_mode=second)
            assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
            closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
            assert np.max(np.abs(xg - closed)) <= TOL
        # host-pointer entry point
        xh, fh, gh, ph = s.minimize_host(obj, x0, per_problem=Y)
        np.testing.assert_array_equal(xh, xg)
    # what the kernel is not built for is refused, not approximated
    s = gpu_solver_factory(m=12)
    with pytest.raises(amd.capi.EngineError):
        s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
    A, Y = amd.synthetic_ridge_host(4, 100, 100, seed=1)
    with pytest.raises(amd.capi.EngineError):
        gpu_solver_factory(m=10).minimize(amd.SquaredErrorRidge(A, lam, matrix_cores=True), _to_dev(np.zeros((4, 100))),
                                          per_problem=_to_dev(Y))


# ---- dense BFGS --------------------------------------------------------
def test_dense_bfgs_solves(oracle, gpu_solver_factory):
    """Bfgs<F, LineSearch> (solver/bfgs.h): device == twin bit for bit (x*, f*, g*, status, iterations,
    evaluations) for every mapping the library picks (n <= 8, 16, 32, 64), both line searches and both
    presets, on ragged batches; <= 1e-6 against the reference-order solve under parity stopping (the oracle
    is pinned to the reference's Bfgs in test_oracle)."""
    import cppnumericalsolvers_amd as amd
    base = gpu_solver_factory()
    for n, B in ((2, 40), (7, 33), (16, 50), (32, 67), (50, 21), (64, 9)):
        x0 = amd.synthetic_x0_host(B, n, "std" if n % 2 == 0 else "u2", seed=5 * n + 1)
        P = 8
        while P < n:
            P *= 2
        for ls in ("more_thuente", "hager_zhang"):
            for stop_o in (oracle.default_stop(), oracle.parity_stop()):
                s = amd.BatchedBfgs(stopping_progress=_engine_stop(stop_o), context=base.ctx, linesearch=ls)
                xg, fg, gg, pg = _solve_gpu(s, amd.Rosenbrock(), x0)
                xb, fb, gb, pb = oracle.bfgs_minimize_batch("rosenbrock", x0, stop=stop_o, reduction="butterfly",
                                                            width=P, linesearch=ls)
                np.testing.assert_array_equal(xg, xb)
                np.testing.assert_array_equal(fg, fb)
                np.testing.assert_array_equal(gg, gb)
                for k in ("status", "num_iterations", "nfev"):
                    np.testing.assert_array_equal(pg[k], pb[k], err_msg=k)
            xs, fs, _, ps = oracle.bfgs_minimize_batch("rosenbrock", x0, stop=oracle.parity_stop(), linesearch=ls)
            assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
            assert np.all(pg["status"] >= 2)
    # every built split of the padded width returns the bits of the library's choice (round 6: one column of H per lane at
    # 32 and 64), and a split that does not cover exactly the padded width is refused
    for n, B, splits in ((29, 37, ((8, 4), (16, 2), (32, 1))), (64, 11, ((16, 4), (32, 2), (64, 1))), (16, 19, ((8, 2),))):
        x0 = amd.synthetic_x0_host(B, n, "std", seed=3 * n + 2)
        for ls in ("more_thuente", "hager_zhang"):
            st = _engine_stop(oracle.parity_stop())
            ref = _solve_gpu(amd.BatchedBfgs(stopping_progress=st, context=base.ctx, linesearch=ls), amd.Rosenbrock(), x0)
            assert ref[3]["status"].min() >= 2
            for W, E in splits:
                s = amd.BatchedBfgs(stopping_progress=st, context=base.ctx, linesearch=ls, lanes_per_problem=W, elems_per_lane=E)
                got = _solve_gpu(s, amd.Rosenbrock(), x0)
                assert s.last_launch()["lanes_per_problem"] == W and s.last_launch()["elems_per_lane"] == E
                for a, b in zip(got[:3], ref[:3]):
                    np.testing.assert_array_equal(a, b, err_msg="n=%d %dx%d %s" % (n, W, E, ls))
                np.testing.assert_array_equal(got[3]["num_iterations"], ref[3]["num_iterations"])
    launch = amd.BatchedBfgs(context=base.ctx)
    _solve_gpu(launch, amd.Rosenbrock(), amd.synthetic_x0_host(8, 32))
    assert launch.last_launch()["lanes_per_problem"] == 32 and launch.last_launch()["elems_per_lane"] == 1
    for W, E in ((16, 4), (8, 2), (64, 1), (32, 4), (16, 1)):     # 32 coordinates: P = 32
        with pytest.raises(amd.capi.EngineError):
            amd.BatchedBfgs(context=base.ctx, lanes_per_problem=W, elems_per_lane=E).minimize(
                amd.Rosenbrock(), _to_dev(amd.synthetic_x0_host(2, 32)))
    # diagonal quadratic, host-pointer entry point, and what the kernel is not built for
    p = np.concatenate([np.linspace(1.0, 50.0, 20), [5.0]])
    x0 = amd.synthetic_x0_host(16, 20, "u2")
    s = amd.BatchedBfgs(context=base.ctx)
    xh, fh, gh, ph = s.minimize_host(amd.DiagQuadratic(p[:-1], p[-1]), x0)
    xb, fb, _, pb = oracle.bfgs_minimize_batch("diag_quadratic", x0, params=p, reduction="butterfly", width=32)
    np.testing.assert_array_equal(xh, xb)
    np.testing.assert_array_equal(ph["num_iterations"], pb["num_iterations"])
    with pytest.raises(amd.capi.EngineError):
        s.minimize(amd.Rosenbrock(), _to_dev(amd.synthetic_x0_host(2, 100)))
