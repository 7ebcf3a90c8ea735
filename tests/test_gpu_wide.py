"""GPU: Lbfgs for problems larger than a wavefront holds (n > 256): one problem per workgroup, vectors and correction ring
in an HBM workspace (csrc/lbfgs_wide_kernel.hpp).  The reference is dynamic in n; this lifts the n <= 256 cap of the
wavefront-resident kernels for Lbfgs<F, m, MoreThuente> on the Rosenbrock and DiagQuadratic objectives.

Device == the oracle's `strided` twin (256 lanes, lane t owns j = t, t + 256, ...) bit for bit: x*, f*, g*, status,
iteration / evaluation counts and the three deltas; <= 1e-6 from the reference-order solve under parity stopping (the
reference-order oracle is itself bit-identical to the reference binary at these sizes: tests/test_oracle.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _to_dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _engine_stop(oracle_stop):
    from cppnumericalsolvers_amd import capi
    dst = capi.Stop()
    for name, _ in oracle_stop._fields_:
        setattr(dst, name, getattr(oracle_stop, name))
    return dst


def _problem(objective, n, B, seed):
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(seed)
    if objective == "rosenbrock":
        return amd.Rosenbrock(), None, np.tile([-1.2, 1.0], n)[:n] + 0.1 * rng.uniform(-1, 1, (B, n))
    a, c = rng.uniform(0.5, 20.0, n), 1.5
    return amd.DiagQuadratic(a, c), np.concatenate([a, [c]]), rng.uniform(-2, 2, (B, n))


def _compare(x, f, g, p, twin):
    import cppnumericalsolvers_amd as amd
    xo, fo, go, po = twin
    np.testing.assert_array_equal(x.cpu().numpy(), xo)
    np.testing.assert_array_equal(f.cpu().numpy(), fo)
    np.testing.assert_array_equal(g.cpu().numpy(), go)
    pg = amd.progress_to_numpy(p)
    for k in ("status", "num_iterations", "nfev", "sum_k", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(pg[k], po[k], err_msg=k)
    return pg


@pytest.mark.parametrize("objective,n,m,B", [("rosenbrock", 257, 6, 9), ("rosenbrock", 300, 10, 40), ("rosenbrock", 1000, 5, 12),
                                             ("rosenbrock", 4096, 10, 6), ("rosenbrock", 1500, 7, 5), ("rosenbrock", 4097, 6, 3),
                                             ("diag_quadratic", 2048, 10, 7), ("diag_quadratic", 513, 17, 8),
                                             ("diag_quadratic", 700, 6, 33), ("diag_quadratic", 5000, 32, 3)])
def test_wide_kernel_equals_its_twin(gpu_solver_factory, oracle, objective, n, m, B):
    import torch
    obj, params, x0 = _problem(objective, n, B, seed=7 * n + m)
    stops = [oracle.default_stop(), oracle.parity_stop(), oracle.default_stop("conservative")]
    if objective == "rosenbrock" and n >= 1000:
        # (the long chain needs tens of thousands of iterations under the tight tests: cut them at 400 -- the iteration
        #  limit is one more stopping test to agree on)
        for st in stops[1:]:
            st.num_iterations = 400
    for stop_o in stops:
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="default")
        x, f, g, p = s.minimize(obj, _to_dev(x0))
        torch.cuda.synchronize()
        ll = s.last_launch()
        assert ll["threads"] == 256 and s.last_arithmetic() == "exact"
        # up to n = 512 the vectors stay in registers (2 coordinates per thread); above they live in the workspace, the
        # direction in LDS up to n = 4096
        assert ll["elems_per_lane"] == (2 if n <= 512 else 0)
        assert ll["lds_bytes"] == (8 * ((n + 1) & ~1) if 512 < n <= 4096 else 0)
        twin = oracle.minimize_batch(objective, x0, m=m, stop=stop_o, params=params, reduction="strided", width=256)
        _compare(x, f, g, p, twin)
    if objective == "rosenbrock" and n >= 1000:
        return   # not converged at the cut: two summation orders are only comparable at a minimiser
    xs, fs, _, ps = oracle.minimize_batch(objective, x0, m=m, stop=oracle.parity_stop(), params=params)   # reference order
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(oracle.parity_stop()))
    x, f, g, p = s.minimize(obj, _to_dev(x0))
    torch.cuda.synchronize()
    assert np.all(ps["status"] != 1)
    assert np.max(np.abs(x.cpu().numpy() - xs)) <= TOL
    assert np.max(np.abs(f.cpu().numpy() - fs) / np.maximum(1.0, np.abs(fs))) <= TOL


def test_wide_kernel_more_problems_than_resident_workgroups_and_host_entry(gpu_solver_factory, oracle):
    """1 500 problems of n = 260 (more than the grid holds: the work queue refills workgroups), through the device and
    the host-pointer entry points; a single problem; an empty batch."""
    import torch
    import cppnumericalsolvers_amd as amd
    n, m, B = 260, 6, 1500
    obj, params, x0 = _problem("rosenbrock", n, B, seed=1)
    st = oracle.default_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st))
    x, f, g, p = s.minimize(obj, _to_dev(x0))
    torch.cuda.synchronize()
    assert s.last_launch()["blocks"] < B
    twin = oracle.minimize_batch("rosenbrock", x0, m=m, stop=st, reduction="strided", width=256)
    _compare(x, f, g, p, twin)
    xh, fh, gh, ph = s.minimize_host(obj, x0)
    np.testing.assert_array_equal(xh, twin[0])
    np.testing.assert_array_equal(fh, twin[1])
    np.testing.assert_array_equal(ph["num_iterations"], twin[3]["num_iterations"])
    x1, f1, g1, p1 = s.minimize(obj, _to_dev(x0[:1]))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(x1.cpu().numpy(), twin[0][:1])
    x0e = _to_dev(np.zeros((0, n)))
    xe, fe, ge, pe = s.minimize(obj, x0e)
    assert xe.shape == (0, n)


def test_wide_kernel_large_dimension_and_edge_values(gpu_solver_factory, oracle):
    """n = 100 000 (two problems) and n = 1 000 003 (one problem, not a multiple of anything); NaN / inf start
    coordinates end the way the twin does."""
    import torch
    for n, B, m in ((100000, 2, 8), (1000003, 1, 4)):
        obj, params, x0 = _problem("diag_quadratic", n, B, seed=n)
        st = oracle.default_stop()
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st))
        x, f, g, p = s.minimize(obj, _to_dev(x0))
        torch.cuda.synchronize()
        assert s.last_launch()["threads"] == 1024        # n >= 32768: sixteen wavefronts per problem, 1024 lanes in the sums
        _compare(x, f, g, p, oracle.minimize_batch("diag_quadratic", x0, m=m, stop=st, params=params, reduction="strided", width=1024))
    n, m = 400, 5
    obj, params, x0 = _problem("rosenbrock", n, 6, seed=3)
    x0[1, 7] = np.nan
    x0[2, 300] = np.inf
    x0[3, 0] = -np.inf
    x0[4, :] = 1.0        # the minimiser: zero gradient at the start (quirk Q1)
    x0[5, 399] = 1e200
    st = oracle.default_stop()
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st))
    x, f, g, p = s.minimize(obj, _to_dev(x0))
    torch.cuda.synchronize()
    _compare(x, f, g, p, oracle.minimize_batch("rosenbrock", x0, m=m, stop=st, reduction="strided", width=256))


def test_wide_kernel_refuses_what_it_is_not_built_for(gpu_solver_factory):
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    x0 = _to_dev(np.zeros((2, 300)))
    for kw in (dict(arithmetic="fma"),):
        with pytest.raises(capi.EngineError) as e:
            gpu_solver_factory(m=5, **kw).minimize(amd.Rosenbrock(), x0)
        assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError):
        amd.BatchedLbfgsb(m=5).minimize(amd.Rosenbrock(), x0)
    with pytest.raises(capi.EngineError):
        amd.BatchedBfgs().minimize(amd.Rosenbrock(), x0)
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=5).minimize(amd.SquaredErrorRidge(np.ones((4, 300)), 0.1), x0, per_problem=_to_dev(np.ones((2, 4))))


def test_wide_kernel_register_and_memory_forms_agree(gpu_solver_factory, oracle, monkeypatch):
    """The two storage forms (vectors in registers up to n = 512, in the workspace above) execute the same operations in
    the same order: forcing the memory form at a small n (MI355_WIDE_IN_MEMORY=1) changes no bit."""
    import torch
    obj, params, x0 = _problem("rosenbrock", 500, 10, seed=5)
    st = oracle.default_stop()
    s = gpu_solver_factory(m=8, stopping_progress=_engine_stop(st))
    a = s.minimize(obj, _to_dev(x0))
    torch.cuda.synchronize()
    assert s.last_launch()["elems_per_lane"] == 2
    monkeypatch.setenv("MI355_WIDE_IN_MEMORY", "1")
    b = s.minimize(obj, _to_dev(x0))
    torch.cuda.synchronize()
    assert s.last_launch()["elems_per_lane"] == 0
    for u, v in zip(a[:3], b[:3]):
        np.testing.assert_array_equal(u.cpu().numpy(), v.cpu().numpy())
    monkeypatch.setenv("MI355_WIDE_LDS_MAX_N", "0")          # ... and with the direction in memory instead of LDS
    c = s.minimize(obj, _to_dev(x0))
    torch.cuda.synchronize()
    assert s.last_launch()["lds_bytes"] == 0
    for u, v in zip(a[:3], c[:3]):
        np.testing.assert_array_equal(u.cpu().numpy(), v.cpu().numpy())


@pytest.mark.parametrize("objective,n,m,B", [("rosenbrock", 300, 6, 12), ("rosenbrock", 2000, 10, 5), ("diag_quadratic", 700, 5, 20),
                                             ("diag_quadratic", 6000, 10, 3), ("diag_quadratic", 40000, 6, 2)])
def test_wide_kernel_with_the_hager_zhang_line_search(gpu_solver_factory, oracle, objective, n, m, B):
    """Lbfgs<F, m, HagerZhang> above n = 256: the scalar state machine of the wavefront kernels (hager_zhang_device.hpp,
    hz_search_core) over the workgroup kernel's evaluation.  Device == strided twin bit for bit."""
    import torch
    obj, params, x0 = _problem(objective, n, B, seed=11 * n + m)
    stops = [oracle.default_stop(), oracle.parity_stop()]
    if objective == "rosenbrock" and n >= 1000:
        stops[1].num_iterations = 300
    for stop_o in stops:
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), linesearch="hager_zhang")
        x, f, g, p = s.minimize(obj, _to_dev(x0))
        torch.cuda.synchronize()
        T = 256      # (Hager-Zhang: the four-wavefront kernel at every n — the 1024-thread form spilled and was removed)
        assert s.last_launch()["threads"] == T
        twin = oracle.minimize_batch(objective, x0, m=m, stop=stop_o, params=params, reduction="strided", width=T,
                                     linesearch="hager_zhang")
        _compare(x, f, g, p, twin)


@pytest.mark.parametrize("n,m,linesearch", [(300, 6, "more_thuente"), (1500, 10, "more_thuente"), (700, 5, "hager_zhang")])
def test_wide_kernel_second_mode_with_a_non_constant_hessian(gpu_solver_factory, oracle, n, m, linesearch):
    """hessian_from_functor above n = 256: the preconditioner 1 / (|H_jj(x)| + eps) from the workgroup functor's hess_diag
    at every iterate (solver/lbfgs.h:129-134), in the register form (n = 300) and the workspace forms."""
    import torch
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    rng = np.random.default_rng(n)
    x0 = np.tile([-1.2, 1.0], n)[:n] + 0.1 * rng.uniform(-1, 1, (7, n))
    st = oracle.default_stop()
    st.num_iterations = 250
    s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(st), linesearch=linesearch)
    x, f, g, p = s.minimize(amd.Rosenbrock(differentiability="second"), _to_dev(x0))
    torch.cuda.synchronize()
    twin = oracle.minimize_batch("rosenbrock", x0, m=m, stop=st, second_mode="functor", linesearch=linesearch,
                                 reduction="strided", width=256)
    pg = _compare(x, f, g, p, twin)
    x1, f1, g1, p1 = s.minimize(amd.Rosenbrock(), _to_dev(x0))
    torch.cuda.synchronize()
    assert not np.array_equal(x1.cpu().numpy(), x.cpu().numpy())   # a different iteration from the First-mode one
    dq = amd.DiagQuadratic(np.ones(n), 0.0)
    dq.hessian_from_functor = True
    with pytest.raises(capi.EngineError):
        s.minimize(dq, _to_dev(x0))
