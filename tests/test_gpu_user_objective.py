"""GPU: a user objective compiled into a build of the library (the SVM functor of the reference's
src/examples/svm_primal_lbfgs.cc): evaluation and full solves bit-identical to the oracle twin, within 1e-6 of the
reference's own Lbfgs on the example's functor; the C++ host class drives it through the ordinary headers."""
import os
import subprocess

import numpy as np
import pytest

import svm_data

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    dst = capi.Stop()
    for name, _ in oracle_stop._fields_:
        setattr(dst, name, getattr(oracle_stop, name))
    return dst


@pytest.fixture(scope="module")
def svm_context():
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import _build
    path = os.path.join(_build.PKG_DIR, "libmi355_lbfgs_svm.so")
    assert os.path.exists(path), "the example library is built by __graft_entry__.build() and travels with the tree"
    ctx = amd.Context(0, library=path)
    yield ctx
    ctx.close()


def test_svm_user_objective_matches_twin_and_reference(svm_context, oracle):
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    import ref_lib
    X, y = svm_data.two_blobs()
    p = svm_data.params(X, y, C=1.0)
    n = X.shape[1] + 1
    obj = amd.Objective(capi.OBJ_USER_FIRST, p, "svm_squared_hinge")
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(33, n))
    # evaluation: every mapping that covers n
    for W, E in [(8, 1), (8, 2), (16, 1), (16, 4), (32, 2), (64, 1)]:
        s = amd.BatchedLbfgs(m=10, context=svm_context, lanes_per_problem=W, elems_per_lane=E, arithmetic="exact")
        f, g = s.evaluate(obj, _to_dev(pts))
        f, g = f.cpu().numpy(), g.cpu().numpy()
        for b in range(pts.shape[0]):
            fe, ge = oracle.evaluate("svm_squared_hinge", pts[b], params=p, reduction="butterfly", width=8)
            assert f[b] == fe, (W, E, b)
            np.testing.assert_array_equal(g[b], ge)
    # solves: from the origin (the example's start) and random starts, both presets
    x0 = np.vstack([np.zeros(n), rng.normal(size=(200, n))])
    for st in (oracle.default_stop(), oracle.parity_stop()):
        s = amd.BatchedLbfgs(m=10, stopping_progress=_engine_stop(st), context=svm_context, arithmetic="exact")
        x, f, g, pr = s.minimize(obj, _to_dev(x0))
        torch.cuda.synchronize()
        x, f, g = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
        xb, fb, gb, pb = oracle.minimize_batch("svm_squared_hinge", x0, m=10, stop=st, params=p, reduction="butterfly",
                                               width=8)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        pg = amd.progress_to_numpy(pr)
        for k in ("status", "num_iterations", "nfev"):
            np.testing.assert_array_equal(pg[k], pb[k])
    if ref_lib.available() and hasattr(ref_lib.lib(), "ref_svm_minimize_batch"):
        xr, fr, _, _ = ref_lib.svm_minimize_batch(p, x0, m=10, stop=oracle.parity_stop())
        assert np.max(np.abs(x - xr)) <= 1e-6 and np.max(np.abs(f - fr)) <= 1e-6
    w, b = x[0, :-1], x[0, -1]
    assert np.mean(np.sign(X @ w + b) == y) > 0.9
    # what is not built is refused, not ignored
    with pytest.raises(capi.EngineError) as e:
        amd.BatchedLbfgs(m=10, context=svm_context, arithmetic="fma").minimize(obj, _to_dev(x0))   # no eval_fma
    assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError):   # the default library has no objective 100
        amd.BatchedLbfgs(m=10, arithmetic="exact").minimize(obj, _to_dev(x0))


def test_svm_user_objective_other_solvers(svm_context, oracle):
    """A user functor gets every solver of the path, not only Lbfgs + More-Thuente: Lbfgs<F, m, HagerZhang> and dense
    Bfgs<F> with either line search, device == twin bit for bit (the reference's line searches and Bfgs are templates
    over the function like Lbfgs)."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    X, y = svm_data.two_blobs()
    p = svm_data.params(X, y, C=1.0)
    n = X.shape[1] + 1
    obj = amd.Objective(capi.OBJ_USER_FIRST, p, "svm_squared_hinge")
    x0 = np.vstack([np.zeros(n), np.random.default_rng(3).normal(size=(90, n))])

    def same(dev, tw):
        x, f, g, pr = dev
        torch.cuda.synchronize()
        np.testing.assert_array_equal(x.cpu().numpy(), tw[0])
        np.testing.assert_array_equal(f.cpu().numpy(), tw[1])
        np.testing.assert_array_equal(g.cpu().numpy(), tw[2])
        pg = amd.progress_to_numpy(pr)
        for k in ("status", "num_iterations", "nfev"):
            np.testing.assert_array_equal(pg[k], tw[3][k])

    for st in (oracle.default_stop(), oracle.parity_stop()):
        s = amd.BatchedLbfgs(m=10, stopping_progress=_engine_stop(st), context=svm_context, linesearch="hager_zhang")
        same(s.minimize(obj, _to_dev(x0)),
             oracle.minimize_batch("svm_squared_hinge", x0, m=10, stop=st, params=p, reduction="butterfly", width=8,
                                   linesearch="hager_zhang"))
        for ls in ("more_thuente", "hager_zhang"):
            b = amd.BatchedBfgs(stopping_progress=_engine_stop(st), context=svm_context, linesearch=ls)
            same(b.minimize(obj, _to_dev(x0)),
                 oracle.bfgs_minimize_batch("svm_squared_hinge", x0, stop=st, params=p, reduction="butterfly", width=8,
                                            linesearch=ls))
    xs, fs, _, _ = oracle.minimize_batch("svm_squared_hinge", x0, m=10, stop=oracle.parity_stop(), params=p)
    xb, fb, _, _ = [t.cpu().numpy() if hasattr(t, "cpu") else t
                    for t in amd.BatchedBfgs(stopping_progress=_engine_stop(oracle.parity_stop()),
                                             context=svm_context).minimize(obj, _to_dev(x0))]
    assert np.max(np.abs(xb - xs)) <= 1e-5   # another algorithm, the same minimiser


def test_svm_user_objective_under_lbfgsb(svm_context, oracle):
    """The user functor under the box-constrained solver (weights in [-0.25, 0.25]): device == twin bit for bit,
    <= 1e-6 from the reference's Lbfgsb<F, 5> on the example's functor."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    import ref_lib
    X, y = svm_data.two_blobs()
    p = svm_data.params(X, y, C=1.0)
    n = X.shape[1] + 1
    lo = np.concatenate([np.full(n - 1, -0.25), [-1e3]])
    hi = np.concatenate([np.full(n - 1, 0.25), [1e3]])
    obj = amd.Objective(capi.OBJ_USER_FIRST, p, "svm_squared_hinge")
    x0 = np.vstack([np.zeros(n), np.random.default_rng(2).normal(size=(150, n))])
    for st in (oracle.lbfgsb_default_stop(), oracle.parity_stop()):
        s = amd.BatchedLbfgsb(arithmetic="exact", m=5, stopping_progress=_engine_stop(st), context=svm_context)
        s.SetBounds(lo, hi)
        x, f, g, pr = s.minimize(obj, _to_dev(x0))
        torch.cuda.synchronize()
        x, f, g = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
        xb, fb, gb, pb = oracle.lbfgsb_minimize_batch("svm_squared_hinge", x0, m=5, stop=st, params=p, lower=lo, upper=hi,
                                                       reduction="butterfly", width=16)
        np.testing.assert_array_equal(x, xb)
        np.testing.assert_array_equal(f, fb)
        np.testing.assert_array_equal(g, gb)
        pg = amd.progress_to_numpy(pr)
        for k in ("status", "num_iterations", "nfev"):
            np.testing.assert_array_equal(pg[k], pb[k])
        assert np.all(x <= hi) and np.all(x >= lo) and np.any(np.abs(x[:, :-1]) == 0.25)
    if ref_lib.available():
        xr, fr, _, _ = ref_lib.lbfgsb_minimize_batch("svm_squared_hinge", x0, m=5, stop=oracle.parity_stop(), lower=lo,
                                                     upper=hi, params=p)
        assert np.max(np.abs(x - xr)) <= 1e-6 and np.max(np.abs(f - fr)) <= 1e-6
    with pytest.raises(capi.EngineError) as e:   # built for m <= 5
        amd.BatchedLbfgsb(arithmetic="exact", m=6, context=svm_context).minimize(obj, _to_dev(x0))
    assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError):        # the default library has no objective 100
        amd.BatchedLbfgsb(arithmetic="exact", m=5).minimize(obj, _to_dev(x0))


def test_dual_svm_user_objective_under_lbfgsb(svm_context, oracle):
    """The second worked example (examples/user_objective_svm_dual: the dual SVM of the reference's
    src/examples/svm_dual_lbfgsb.cc, dense, n = 100, box [0, C], `Lbfgsb<F>` = m 5): sixteen lanes x eight coordinates.
    Reference-order kernel == the butterfly twin and relaxed kernel == its twin, bit for bit; both within 1e-6 of the
    reference binary's Lbfgsb on the example's functor (tight stop); shapes the build does not hold are refused."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    import ref_lib
    X, y = svm_data.standardised_blobs(100, 4, seed=7)
    p, Q = svm_data.dual_params(X, y)
    n, C = 100, 1.0
    lo, hi = np.zeros(n), np.full(n, C)
    obj = amd.Objective(capi.OBJ_USER_FIRST + 1, p, "svm_dual")
    x0 = np.vstack([np.zeros(n), np.random.default_rng(3).uniform(0.0, C, size=(40, n))])
    tight = oracle.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8, past=0)
    results = {}
    for arith in ("exact", "fma"):
        for st in (oracle.lbfgsb_default_stop(), tight):
            s = amd.BatchedLbfgsb(arithmetic=arith, m=5, stopping_progress=_engine_stop(st), context=svm_context)
            s.SetBounds(lo, hi)
            x, f, g, pr = s.minimize(obj, _to_dev(x0))
            torch.cuda.synchronize()
            x, f, g = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy()
            assert s.last_arithmetic() == arith and s.last_launch()["elems_per_lane"] == 8
            if arith == "exact":
                twin = oracle.lbfgsb_minimize_batch("svm_dual", x0, m=5, stop=st, params=p, lower=lo, upper=hi,
                                                    reduction="butterfly", width=128)
            else:
                twin = oracle.lbfgsb_fast_minimize_batch("svm_dual", x0, m=5, stop=st, params=p, lower=lo, upper=hi)
            np.testing.assert_array_equal(x, twin[0], err_msg=arith)
            np.testing.assert_array_equal(f, twin[1], err_msg=arith)
            np.testing.assert_array_equal(g, twin[2], err_msg=arith)
            pg = amd.progress_to_numpy(pr)
            for k in ("status", "num_iterations", "nfev", "sum_k"):
                np.testing.assert_array_equal(pg[k], twin[3][k], err_msg=arith + " " + k)
        results[arith] = (x, f)
        assert np.all(x >= 0) and np.all(x <= C) and np.any(x == C) and np.any(x == 0)
    if ref_lib.available():
        xr, fr, _, pr_ = ref_lib.lbfgsb_minimize_batch("svm_dual", x0, m=5, stop=tight, lower=lo, upper=hi, params=p)
        for arith, (x, f) in results.items():
            assert np.max(np.abs(x - xr)) <= 1e-6 and np.max(np.abs(f - fr)) <= 1e-6, arith
    # the default arithmetic of a user objective is the reference-order kernel
    s = amd.BatchedLbfgsb(m=5, stopping_progress=_engine_stop(tight), context=svm_context)
    s.SetBounds(lo, hi)
    s.minimize(obj, _to_dev(x0[:3]))
    assert s.last_arithmetic() == "exact"
    for kw in (dict(m=6), dict(m=5, linesearch="hager_zhang")):      # this build holds m <= 5, More-Thuente
        with pytest.raises(capi.EngineError) as e:
            sb = amd.BatchedLbfgsb(arithmetic="exact", context=svm_context, **kw)
            sb.SetBounds(lo, hi)
            sb.minimize(obj, _to_dev(x0[:3]))
        assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError) as e:                       # ... and no Lbfgs kernels for this objective
        amd.BatchedLbfgs(m=5, context=svm_context).minimize(obj, _to_dev(x0[:3]))
    assert e.value.code == capi.ERR_UNSUPPORTED


def test_dual_svm_example_through_the_cpp_headers():
    """examples/user_objective_svm_dual/svm_dual_lbfgsb.cc — the reference example's main() over the drop-in headers."""
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "svm_dual_lbfgsb")
    lib = os.path.join(ROOT, "cppnumericalsolvers_amd")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "user_objective_svm_dual", "svm_dual_lbfgsb.cc"),
                        "-L" + lib, "-l:libmi355_lbfgs_svm.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "support vecs" in r.stdout and "PASS" in r.stdout


def test_svm_example_through_the_cpp_headers():
    """examples/user_objective_svm/svm_primal_lbfgs.cc — the reference example's main() over the drop-in headers,
    linked against the build that holds the device functor."""
    build = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "svm_primal_lbfgs")
    lib = os.path.join(ROOT, "cppnumericalsolvers_amd")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "user_objective_svm", "svm_primal_lbfgs.cc"),
                        "-L" + lib, "-l:libmi355_lbfgs_svm.so", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib",
                        "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "accuracy" in r.stdout and "PASS" in r.stdout


@pytest.mark.parametrize("d", [400, 700])
def test_svm_user_objective_with_hundreds_of_features(svm_context, oracle, d):
    """n = d + 1 > 256: the user objective on the workgroup kernel (wide_type functor,
    examples/user_objective_svm/svm_squared_hinge_wide.hpp; registers at n = 401, workspace + LDS direction at n = 701).
    Device == twin (strided policy) bit for bit; <= 1e-6 from the reference-order solve, which equals the reference's own
    Lbfgs on the example's functor (tests/test_user_objective.py)."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    X, y = svm_data.two_blobs(N=120, d=d, seed=d, separation=0.15)
    p = svm_data.params(X, y, C=0.5)
    n = d + 1
    obj = amd.Objective(capi.OBJ_USER_FIRST, p, "svm_squared_hinge")
    x0 = np.vstack([np.zeros(n), 0.05 * np.random.default_rng(d).normal(size=(9, n))])
    for stop_o in (oracle.default_stop(), oracle.parity_stop()):
        s = amd.BatchedLbfgs(m=10, stopping_progress=_engine_stop(stop_o), context=svm_context)
        x, f, g, pr = s.minimize(obj, _to_dev(x0))
        torch.cuda.synchronize()
        ll = s.last_launch()
        assert ll["threads"] == 256 and ll["elems_per_lane"] == (2 if n <= 512 else 0)
        xo, fo, go, po = oracle.minimize_batch("svm_squared_hinge", x0, m=10, stop=stop_o, params=p, reduction="strided", width=256)
        np.testing.assert_array_equal(x.cpu().numpy(), xo)
        np.testing.assert_array_equal(f.cpu().numpy(), fo)
        np.testing.assert_array_equal(g.cpu().numpy(), go)
        pg = amd.progress_to_numpy(pr)
        for k in ("status", "num_iterations", "nfev", "sum_k"):
            np.testing.assert_array_equal(pg[k], po[k], err_msg=k)
    xs, fs, _, ps = oracle.minimize_batch("svm_squared_hinge", x0, m=10, stop=oracle.parity_stop(), params=p)
    assert np.max(np.abs(x.cpu().numpy() - xs)) <= 1e-6 and np.max(np.abs(f.cpu().numpy() - fs)) <= 1e-6
    w, b = x.cpu().numpy()[0, :-1], x.cpu().numpy()[0, -1]
    assert np.mean(np.sign(X @ w + b) == y) > 0.9
    # the default library has no functor for this id
    with pytest.raises(capi.EngineError):
        amd.BatchedLbfgs(m=10).minimize(obj, _to_dev(x0))
