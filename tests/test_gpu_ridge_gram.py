"""GPU tests of the ridge objective in normal-equation form (objective id 5, csrc/ridge_gram.hpp): Gram matrix and
c_b = A^T y_b by batched GEMMs on the matrix cores, then the ordinary Lbfgs kernel on the n x n quadratic.
device == twin bit for bit; device within 1e-6 of the reference binary and of the closed form."""
import os

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


def _mapping(n):
    P = 8
    while P < n:
        P <<= 1
    return P, (1 if P == 8 else (4 if P == 256 else 2))


def _same(dev, twin, msg=""):
    import cppnumericalsolvers_amd as amd
    x, f, g, p = dev
    _torch().cuda.synchronize()
    xg, fg, gg, pg = x.cpu().numpy(), f.cpu().numpy(), g.cpu().numpy(), amd.progress_to_numpy(p)
    np.testing.assert_array_equal(xg, twin[0], err_msg=msg)
    np.testing.assert_array_equal(fg, twin[1], err_msg=msg)
    np.testing.assert_array_equal(gg, twin[2], err_msg=msg)
    for k in ("status", "num_iterations", "nfev", "sum_k", "x_delta", "f_delta", "gradient_norm"):
        np.testing.assert_array_equal(pg[k], twin[3][k], err_msg=msg + " " + k)
    return xg, fg, gg, pg


def test_gram_objective_evaluation_equals_its_twin(gpu_solver_factory, oracle):
    """One evaluation per problem (mi355_lbfgs_eval_batch): the pre-pass (c_b = A^T y_b on the matrix cores, y_b . y_b)
    and the n x n product, bit for bit.  At x = 0 the value IS y_b . y_b and the gradient IS -2 c_b."""
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(8)
    for rows, n in ((128, 64), (37, 33), (50, 20), (9, 8), (3, 2), (128, 17)):
        B = 21
        A, Y = amd.synthetic_ridge_host(B, rows, n, seed=rows + n)
        params = oracle.ridge_params(A, 0.1)
        P, E = _mapping(n)
        s = gpu_solver_factory(m=10, arithmetic="default")
        obj = amd.SquaredErrorRidge(A, 0.1, gram=True)
        for X in (np.zeros((B, n)), rng.normal(size=(B, n))):
            f, g = s.evaluate(obj, _to_dev(X), per_problem=_to_dev(Y))
            _torch().cuda.synchronize()
            f, g = f.cpu().numpy(), g.cpu().numpy()
            for b in range(B):
                fe, ge = oracle.evaluate("squared_error_ridge_gram", X[b], params=params, reduction="butterfly_fma", width=P,
                                         per_problem=Y[b:b + 1], fma_group=E)
                assert f[b] == fe, ("value", rows, n, b, f[b], fe)
                np.testing.assert_array_equal(g[b], ge, err_msg="gradient rows=%d n=%d b=%d" % (rows, n, b))


def test_gram_kernel_equals_its_twin(gpu_solver_factory, oracle):
    """Full and ragged batches, rows / n below the tile sizes, history sizes 3..10 (+ the LDS-ring kernel, m = 12), both
    presets, First and Second mode: x*, f*, g*, status, iteration and evaluation counts equal the twin's; under parity
    stopping they are within 1e-6 of the reference-order solve and of the closed form."""
    import cppnumericalsolvers_amd as amd
    lam = 0.1
    for rows, n, B, m in ((128, 64, 70, 10), (128, 64, 16, 10), (50, 20, 37, 10), (3, 2, 5, 10), (128, 64, 33, 6),
                          (100, 64, 19, 3), (37, 33, 21, 5), (9, 8, 12, 10), (128, 64, 11, 12)):
        if rows == 3:
            A = np.array([[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]])
            Y = np.tile(np.array([7.0, 8.0, 9.0]), (B, 1))
        else:
            A, Y = amd.synthetic_ridge_host(B, rows, n, seed=rows + n + B)
        x0 = np.zeros((B, n))
        params = oracle.ridge_params(A, lam)
        P, E = _mapping(n)
        for second in (False, True):
            obj = amd.SquaredErrorRidge(A, lam, differentiability="second" if second else "first", gram=True)
            for stop_o in (oracle.default_stop(), oracle.parity_stop()):
                s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="default")
                dev = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
                twin = oracle.minimize_batch("squared_error_ridge_gram", x0, m=m, stop=stop_o, params=params,
                                             reduction="butterfly_fma", width=P, fma_group=E, per_problem=Y,
                                             second_mode=second)
                xg, fg, gg, pg = _same(dev, twin, "rows=%d n=%d m=%d second=%s" % (rows, n, m, second))
            assert s.last_arithmetic() == "fma"
            xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge", x0, m=m, stop=oracle.parity_stop(),
                                                 params=params, per_problem=Y, second_mode=second)
            assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
            closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
            assert np.max(np.abs(xg - closed)) <= TOL
        xh, fh, gh, ph = s.minimize_host(obj, x0, per_problem=Y)   # host-pointer entry point
        np.testing.assert_array_equal(xh, xg)
    # a second matrix on the same context (the cached Gram matrix is rebuilt), non-zero start points
    A2, Y2 = amd.synthetic_ridge_host(40, 60, 24, seed=99)
    x02 = np.random.default_rng(1).normal(size=(40, 24))
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(oracle.parity_stop()), arithmetic="default")
    for lam2 in (0.1, 0.7):
        dev = s.minimize(amd.SquaredErrorRidge(A2, lam2, gram=True), _to_dev(x02), per_problem=_to_dev(Y2))
        twin = oracle.minimize_batch("squared_error_ridge_gram", x02, m=10, stop=oracle.parity_stop(),
                                     params=oracle.ridge_params(A2, lam2), reduction="butterfly_fma", width=32, fma_group=2,
                                     per_problem=Y2)
        _same(dev, twin, "lambda %g" % lam2)
    # refused, not approximated: the exact arithmetic, n > 64, the Hager-Zhang line search
    from cppnumericalsolvers_amd import capi
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=10, arithmetic="exact").minimize(amd.SquaredErrorRidge(A2, 0.1, gram=True), _to_dev(x02),
                                                             per_problem=_to_dev(Y2))
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=10, arithmetic="default", linesearch="hager_zhang").minimize(
            amd.SquaredErrorRidge(A2, 0.1, gram=True), _to_dev(x02), per_problem=_to_dev(Y2))
    A3, Y3 = amd.synthetic_ridge_host(4, 4100, 16, seed=1)          # rows > MI355_LBFGS_GRAM_MAX_ROWS
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=10, arithmetic="default").minimize(amd.SquaredErrorRidge(A3, 0.1, gram=True),
                                                               _to_dev(np.zeros((4, 16))), per_problem=_to_dev(Y3))
    A4, Y4 = amd.synthetic_ridge_host(2, 64, 300, seed=1)           # n > 256
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=10, arithmetic="default").minimize(amd.SquaredErrorRidge(A4, 0.1, gram=True),
                                                               _to_dev(np.zeros((2, 300))), per_problem=_to_dev(Y4))


def test_gram_kernel_vs_reference_binary_and_degenerate_data(gpu_solver_factory, oracle):
    import cppnumericalsolvers_amd as amd
    import ref_lib
    B, rows, n, lam = 1024, 128, 64, 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n, seed=5)
    x0 = np.zeros((B, n))
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(st), arithmetic="default")
    x, f, g, p = s.minimize(amd.SquaredErrorRidge(A, lam, gram=True), _to_dev(x0), per_problem=_to_dev(Y))
    _torch().cuda.synchronize()
    if ref_lib.available():
        xr, fr, gr, pr = ref_lib.ridge_minimize_batch(A, lam, Y, x0, stop=st)
        assert np.max(np.abs(x.cpu().numpy() - xr)) <= TOL and np.max(np.abs(f.cpu().numpy() - fr)) <= TOL
    # degenerate data: the twin's bits, NaN / inf right-hand sides included
    for case, (Ad, lamd, Yd) in sorted(oracle.degenerate_ridge_data().items()):
        nd = Ad.shape[1]
        P, E = _mapping(nd)
        x0d = np.zeros((Yd.shape[0], nd))
        sd = gpu_solver_factory(m=10, arithmetic="default")
        dev = sd.minimize(amd.SquaredErrorRidge(Ad, lamd, gram=True), _to_dev(x0d), per_problem=_to_dev(Yd))
        twin = oracle.minimize_batch("squared_error_ridge_gram", x0d, m=10, params=oracle.ridge_params(Ad, lamd),
                                     reduction="butterfly_fma", width=P, fma_group=E, per_problem=Yd)
        _same(dev, twin, case)


def test_gram_kernel_whole_config3_batch(gpu_solver_factory, oracle):
    """configs[3] at its full size (262,144 ridge problems, A 128 x 64, lambda 0.1, x0 = 0, m = 10): every solution
    against the closed form and the gradient identity; exact parity with the twin on a strided sample."""
    import cppnumericalsolvers_amd as amd
    torch = _torch()
    B, rows, n, m, lam = 262144, 128, 64, 10, 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n)
    s = gpu_solver_factory(m=m, stopping_progress=amd.parity_stop(), arithmetic="default")
    x, f, g, p = s.minimize(amd.SquaredErrorRidge(A, lam, gram=True), torch.zeros(B, n, dtype=torch.float64, device="cuda:0"),
                            per_problem=_to_dev(Y))
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
    idx = np.arange(0, B, 1024)
    twin = oracle.minimize_batch("squared_error_ridge_gram", np.zeros((idx.size, n)), m=m, stop=oracle.parity_stop(),
                                 params=oracle.ridge_params(A, lam), reduction="butterfly_fma", width=64, fma_group=2,
                                 per_problem=Y[idx])
    np.testing.assert_array_equal(xh[idx], twin[0])
    np.testing.assert_array_equal(fh[idx], twin[1])
    np.testing.assert_array_equal(pn["num_iterations"][idx], twin[3]["num_iterations"])


def test_hager_zhang_search_entry_refuses_the_solve_only_ridge_forms(gpu_solver_factory):
    """mi355_lbfgs_hz_search_batch has no normal-equation / matrix-core form: both ids are refused with
    MI355_ERR_UNSUPPORTED and the outputs stay untouched (round-3 advisor finding: the Gram id used to return
    MI355_OK after one evaluation, without a search and without writing x_out / alpha_out / nfev_out)."""
    import cppnumericalsolvers_amd as amd
    from cppnumericalsolvers_amd import capi
    B, rows, n = 5, 24, 16
    A, Y = amd.synthetic_ridge_host(B, rows, n, seed=3)
    rng = np.random.default_rng(4)
    x, d = rng.normal(size=(B, n)), rng.normal(size=(B, n))
    for kw in (dict(gram=True), dict(matrix_cores=True)):
        s = gpu_solver_factory(m=5, arithmetic="exact", linesearch="hager_zhang")
        with pytest.raises(capi.EngineError) as e:
            s.hz_search(amd.SquaredErrorRidge(A, 0.1, **kw), _to_dev(x), _to_dev(d), _to_dev(np.ones(B)),
                        per_problem=_to_dev(Y))
        assert e.value.code == capi.ERR_UNSUPPORTED


@pytest.mark.parametrize("rows,n,B,m", [(1000, 200, 24, 10), (300, 100, 40, 10), (129, 65, 33, 10), (4096, 256, 9, 10),
                                        (200, 128, 17, 6), (700, 129, 5, 12), (513, 90, 21, 3)])
def test_gram_kernel_beyond_128_by_64_equals_its_twin(gpu_solver_factory, oracle, rows, n, B, m):
    """64 < n <= 256, rows <= 4096 (README.md:126-160 takes any A): Gram matrix and c_b = A^T y_b on the matrix cores,
    then a whole wavefront per problem — G in LDS (n <= 128) or streamed through L2 (n <= 256).  One evaluation and the
    full solve equal the twin bit for bit; x*, f* within 1e-6 of the closed form and of the reference-order solve."""
    import cppnumericalsolvers_amd as amd
    lam = 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n, seed=rows + n + B)
    params = oracle.ridge_params(A, lam)
    P, E = _mapping(n)
    obj = amd.SquaredErrorRidge(A, lam, gram=True)
    X = np.random.default_rng(rows).normal(size=(B, n))
    s = gpu_solver_factory(m=m, arithmetic="default")
    f, g = s.evaluate(obj, _to_dev(X), per_problem=_to_dev(Y))
    _torch().cuda.synchronize()
    f, g = f.cpu().numpy(), g.cpu().numpy()
    for b in range(min(B, 6)):
        fe, ge = oracle.evaluate("squared_error_ridge_gram", X[b], params=params, reduction="butterfly_fma", width=P,
                                 per_problem=Y[b:b + 1], fma_group=E)
        assert f[b] == fe, ("value", b, f[b], fe)
        np.testing.assert_array_equal(g[b], ge)
    x0 = np.zeros((B, n))
    for stop_o in (oracle.default_stop(), oracle.parity_stop()):
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="default")
        dev = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(Y))
        twin = oracle.minimize_batch("squared_error_ridge_gram", x0, m=m, stop=stop_o, params=params,
                                     reduction="butterfly_fma", width=P, fma_group=E, per_problem=Y)
        xg, fg, gg, pg = _same(dev, twin, "rows=%d n=%d m=%d" % (rows, n, m))
    assert s.last_arithmetic() == "fma" and s.last_launch()["lanes_per_problem"] == 64
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    assert np.max(np.abs(xg - closed)) <= TOL
    if rows <= 1024:    # (the exact twin stages the residual on its stack: rows <= 1024)
        xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge", x0, m=m, stop=oracle.parity_stop(), params=params,
                                             per_problem=Y)
        assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
    xh, fh, gh, ph = s.minimize_host(obj, x0, per_problem=Y)   # host-pointer entry point
    np.testing.assert_array_equal(xh, xg)


def test_gram_kernel_vs_reference_binary_on_a_1000_by_200_problem(gpu_solver_factory, oracle, reference):
    """The README ridge example at 1000 x 200 (README.md:126-160: `SquaredError(A, y) + lambda * L2Reg(n)` under the
    reference's Lbfgs): the device's normal-equation form within 1e-6 of the reference binary on x* and f*."""
    import cppnumericalsolvers_amd as amd
    B, rows, n, lam = 96, 1000, 200, 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n, seed=4)
    x0 = np.zeros((B, n))
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(st), arithmetic="default")
    x, f, g, p = s.minimize(amd.SquaredErrorRidge(A, lam, gram=True), _to_dev(x0), per_problem=_to_dev(Y))
    _torch().cuda.synchronize()
    x, f = x.cpu().numpy(), f.cpu().numpy()
    xr, fr, _, pr = reference.ridge_minimize_batch_threaded(A, lam, Y, x0, stop=st, threads=os.cpu_count() or 8, chunk=4)
    assert np.all(pr["status"] != 1) and np.all(amd.progress_to_numpy(p)["status"] != 1)
    assert np.max(np.abs(x - xr)) <= TOL and np.max(np.abs(f - fr)) <= TOL


@pytest.mark.parametrize("rows,n,B,m", [(128, 64, 70, 10), (40, 24, 33, 10), (9, 8, 12, 5), (150, 70, 9, 10), (300, 200, 5, 10),
                                        (64, 32, 40, 12), (3, 2, 6, 10)])
def test_own_matrix_kernel_equals_its_twin(gpu_solver_factory, oracle, rows, n, B, m):
    """Objective id 6: one matrix per problem (README.md:126-160 built once per data set).  Per-problem pre-pass on the matrix
    cores (G_b, c_b, y_b . y_b), then the Lbfgs kernel streaming each problem's own G_b: evaluation and full solve equal
    the twin bit for bit; x*, f* within 1e-6 of the closed form and of the reference-order twin."""
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(rows * 7 + n)
    lam = 0.1
    As = rng.normal(size=(B, rows, n)) / np.sqrt(rows)
    Y = rng.normal(size=(B, rows))
    data = amd.ridge_per_problem_rows(As, Y)
    params = np.array([float(rows), lam])
    P, E = _mapping(n)
    obj = amd.SquaredErrorRidgePerProblem(rows, lam)
    X = rng.normal(size=(B, n))
    s = gpu_solver_factory(m=m, arithmetic="default")
    f, g = s.evaluate(obj, _to_dev(X), per_problem=_to_dev(data))
    _torch().cuda.synchronize()
    f, g = f.cpu().numpy(), g.cpu().numpy()
    for b in range(min(B, 6)):
        fe, ge = oracle.evaluate("squared_error_ridge_own_gram", X[b], params=params, reduction="butterfly_fma", width=P,
                                 per_problem=data[b:b + 1], fma_group=E)
        assert f[b] == fe, ("value", b, f[b], fe)
        np.testing.assert_array_equal(g[b], ge)
    x0 = np.zeros((B, n))
    for stop_o in (oracle.default_stop(), oracle.parity_stop()):
        s = gpu_solver_factory(m=m, stopping_progress=_engine_stop(stop_o), arithmetic="default")
        dev = s.minimize(obj, _to_dev(x0), per_problem=_to_dev(data))
        twin = oracle.minimize_batch("squared_error_ridge_own_gram", x0, m=m, stop=stop_o, params=params,
                                     reduction="butterfly_fma", width=P, fma_group=E, per_problem=data)
        xg, fg, gg, pg = _same(dev, twin, "rows=%d n=%d m=%d" % (rows, n, m))
    assert s.last_arithmetic() == "fma"
    closed = np.stack([np.linalg.solve(As[b].T @ As[b] + lam * np.eye(n), As[b].T @ Y[b]) for b in range(B)])
    assert np.max(np.abs(xg - closed)) <= TOL
    xs, fs, _, _ = oracle.minimize_batch("squared_error_ridge_own", x0, m=m, stop=oracle.parity_stop(), params=params,
                                         per_problem=data)
    assert np.max(np.abs(xg - xs)) <= TOL and np.max(np.abs(fg - fs)) <= TOL
    xh, fh, gh, ph = s.minimize_host(obj, x0, per_problem=data)   # host-pointer entry point
    np.testing.assert_array_equal(xh, xg)
    from cppnumericalsolvers_amd import capi
    with pytest.raises(capi.EngineError):      # fused form only; the stride must hold a matrix and a right-hand side
        gpu_solver_factory(m=m, arithmetic="exact").minimize(obj, _to_dev(x0), per_problem=_to_dev(data))
    with pytest.raises(capi.EngineError):
        gpu_solver_factory(m=m, arithmetic="default").minimize(obj, _to_dev(x0), per_problem=_to_dev(data[:, :-1]))


def test_own_matrix_kernel_vs_reference_binary(gpu_solver_factory, oracle, reference):
    """The device's own-matrix form against the reference binary: one README objective per problem under the reference's
    Lbfgs (oracle/_ref), 256 problems of 128 x 64, x* and f* within 1e-6."""
    import cppnumericalsolvers_amd as amd
    rng = np.random.default_rng(8)
    B, rows, n, lam = 256, 128, 64, 0.1
    As = rng.normal(size=(B, rows, n)) / np.sqrt(rows)
    Y = rng.normal(size=(B, rows))
    st = oracle.parity_stop()
    s = gpu_solver_factory(m=10, stopping_progress=_engine_stop(st), arithmetic="default")
    x, f, g, p = s.minimize(amd.SquaredErrorRidgePerProblem(rows, lam), _to_dev(np.zeros((B, n))),
                            per_problem=_to_dev(amd.ridge_per_problem_rows(As, Y)))
    _torch().cuda.synchronize()
    xr, fr, _, pr = reference.ridge_own_matrix_minimize_batch(As, lam, Y, np.zeros((B, n)), stop=st)
    assert np.max(np.abs(x.cpu().numpy() - xr)) <= TOL and np.max(np.abs(f.cpu().numpy() - fr)) <= TOL


def test_whole_bench_batches_beyond_configs_against_the_closed_form(gpu_solver_factory, oracle):
    """The two bench rows beyond BASELINE.json at their full sizes, EVERY problem against its closed form
    (A^T A + lambda I)^-1 A^T y at 1e-6: 32 768 problems of the shared 1000 x 200 matrix (`--workload cfg4big`) and 65 536
    problems with their own 128 x 64 matrix each (`--workload cfg4own`, matrices generated on the device as bench.py does)."""
    import cppnumericalsolvers_amd as amd
    torch = _torch()
    st = _engine_stop(oracle.parity_stop())
    # shared 1000 x 200 matrix
    B, rows, n, lam = 32768, 1000, 200, 0.1
    A, Y = amd.synthetic_ridge_host(B, rows, n, 20260923)
    s = gpu_solver_factory(m=10, stopping_progress=st, arithmetic="default")
    x, f, g, p = s.minimize(amd.SquaredErrorRidge(A, lam, gram=True), _to_dev(np.zeros((B, n))), per_problem=_to_dev(Y))
    torch.cuda.synchronize()
    closed = np.linalg.solve(A.T @ A + lam * np.eye(n), A.T @ Y.T).T
    assert np.max(np.abs(x.cpu().numpy() - closed)) <= TOL
    pg = amd.progress_to_numpy(p)
    assert np.all(pg["status"] >= 2) and np.all(pg["status"] <= 4)
    del x, f, g, p
    # one matrix per problem
    B, rows, n = 65536, 128, 64
    gen = torch.Generator(device="cuda:0")
    gen.manual_seed(20260923)
    data = torch.randn(B, rows * n + rows, dtype=torch.float64, device="cuda:0", generator=gen)
    data[:, :rows * n] *= 1.0 / np.sqrt(float(rows))
    s = gpu_solver_factory(m=10, stopping_progress=st, arithmetic="default")
    x, f, g, p = s.minimize(amd.SquaredErrorRidgePerProblem(rows, lam), torch.zeros(B, n, dtype=torch.float64, device="cuda:0"),
                            per_problem=data)
    torch.cuda.synchronize()
    As = data[:, :rows * n].reshape(B, rows, n)
    Ys = data[:, rows * n:]
    G = torch.matmul(As.transpose(1, 2), As) + lam * torch.eye(n, dtype=torch.float64, device="cuda:0")
    c = torch.matmul(As.transpose(1, 2), Ys.unsqueeze(2))
    closed_t = torch.linalg.solve(G, c).squeeze(2)          # (independent of the engine: torch's batched LU)
    assert float((x - closed_t).abs().max().item()) <= TOL
    fx = ((torch.matmul(As, x.unsqueeze(2)).squeeze(2) - Ys) ** 2).sum(dim=1) + lam * (x ** 2).sum(dim=1)
    assert float((f - fx).abs().max().item()) <= 1e-9
    pg = amd.progress_to_numpy(p)
    assert np.all(pg["status"] >= 2) and np.all(pg["status"] <= 4)
