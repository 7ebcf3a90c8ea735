"""Constraint FAMILIES on the device (mi355_al_problem.family_eq / family_ineq, csrc/auglag_family.hip): hundreds of affine
constraints per problem, as the reference's src/examples/svm_primal_al.cc:139-147 builds them.  Through the C-ABI:
  * one composite evaluation == the oracle's butterfly twin bit for bit, for every kernel mapping, with table terms ahead of
    the family rows, zero multipliers (MulExpression's short circuit), clamped and unclamped inequalities, rho = 0;
  * full solves (fused outer loop) == the twin bit for bit: x, all multipliers, penalty, violation, KKT norm, progress;
  * the primal SVM with its 200 constraints: device == twin bit for bit, and the twin's sequential form IS the reference
    binary (tests/test_auglag_family_oracle.py); the classifier separates the data;
  * what the family kernels are not built for is refused, not approximated."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import auglag_lib as al  # noqa: E402
from test_gpu_auglag import _assert_same, _engine_problem, _padded  # noqa: E402

pytestmark = pytest.mark.gpu


def _solver(**kw):
    from cppnumericalsolvers_amd import BatchedAugmentedLagrangian
    return BatchedAugmentedLagrangian(**kw)


def _configure(s, cfg):
    c = s.default_config()
    for name, _ in cfg._fields_:
        setattr(c, name, getattr(cfg, name))
    s.config = c


# (n, family equalities, family inequalities, table terms too): every mapping (8,1) (8,2) (16,2) (32,2) (64,2) (64,4), the
# capacity of each (4 x lanes), one lane owning 1..4 constraints, families of one kind only
SHAPES = [(5, 3, 9, True), (8, 0, 32, False), (12, 5, 20, True), (16, 32, 0, False), (30, 7, 50, True), (32, 0, 64, False),
          (40, 20, 100, True), (64, 64, 64, False), (100, 3, 200, True), (105, 0, 200, False), (128, 128, 128, False),
          (200, 10, 240, True), (256, 0, 256, False)]


@pytest.mark.parametrize("n,f_eq,f_ineq,table", SHAPES)
def test_composite_with_families_matches_oracle_bitwise(n, f_eq, f_ineq, table):
    p = al.random_family_problem(n, f_eq, f_ineq, seed=n, table=table)
    rng = np.random.default_rng(300 + n)
    B = 21
    x = rng.uniform(-1.5, 1.5, (B, n))
    lam = rng.uniform(-2, 2, (B, p.n_eq))
    mu = rng.uniform(0, 3, (B, p.n_ineq))
    if p.n_eq:
        lam[::4, ::3] = 0.0                                  # MulExpression's c == 0 short circuit, family rows included
    if p.n_ineq:
        mu[::3, ::2] = 0.0                                   # mu = 0: clamped wherever the constraint holds
        mu[1::5] = 50.0                                      # large mu: nothing clamped
    pen = rng.uniform(0.1, 20.0, B)
    pen[::7] = 0.0                                           # rho = 0: inequality part skipped, penalty part zeroed
    f, g = _solver().evaluate_host(_engine_problem(p), x, lam, mu, pen)
    fo, go = al.oracle_eval(p, x, lam, mu, pen, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(f, fo)
    np.testing.assert_array_equal(g, go)


@pytest.mark.parametrize("n,f_eq,f_ineq,table,B", [(5, 3, 9, True, 40), (12, 0, 32, False, 33), (16, 8, 24, True, 24),
                                                   (30, 7, 50, True, 20), (64, 10, 100, False, 12), (100, 3, 60, True, 6),
                                                   (200, 4, 40, True, 4)])
def test_solves_with_families_match_oracle_bitwise(n, f_eq, f_ineq, table, B):
    p = al.random_family_problem(n, f_eq, f_ineq, seed=n, table=table)
    x0 = np.random.default_rng(n).uniform(-1, 1, (B, n))
    cfg = al.default_config(outer_num_iterations=15)
    s = _solver()
    _configure(s, cfg)
    d = s.minimize_host(_engine_problem(p), x0)
    o = al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=_padded(n))
    _assert_same(d, o)
    assert np.all(d["max_violation"] <= 1e-3)
    # the sequential form (== the reference binary, tests/test_auglag_family_oracle.py) is the same iteration in another
    # summation order: after fifteen outer steps of the reference's default (loose) inner stopping test the two agree to
    # the accuracy that test leaves
    q = al.oracle_minimize(p, x0, config=cfg)
    assert np.max(np.abs(d["x"] - q["x"])) <= 2e-2


def test_states_fed_back_and_nonzero_start_multipliers():
    """Multipliers of the family rows enter as part of the state (warm start) and come back updated."""
    n = 20
    p = al.random_family_problem(n, 6, 30, seed=9, table=True)
    rng = np.random.default_rng(4)
    x0 = rng.uniform(-1, 1, (10, n))
    lam0, mu0 = rng.normal(size=(10, p.n_eq)), rng.uniform(0, 1, (10, p.n_ineq))
    cfg = al.default_config(outer_num_iterations=4)
    s = _solver()
    _configure(s, cfg)
    d = s.minimize_host(_engine_problem(p), x0, lambda0=lam0, mu0=mu0, penalty0=2.0, max_violation0=0.5)
    o = al.oracle_minimize(p, x0, lambda0=lam0, mu0=mu0, penalty0=2.0, max_violation0=0.5, config=cfg,
                           reduction="butterfly", width=_padded(n))
    _assert_same(d, o)
    d2 = s.minimize_host(_engine_problem(p), d["x"], lambda0=d["lambda"], mu0=d["mu"], penalty0=d["penalty"],
                         max_violation0=d["max_violation"])
    o2 = al.oracle_minimize(p, o["x"], lambda0=o["lambda"], mu0=o["mu"], penalty0=o["penalty"],
                            max_violation0=o["max_violation"], config=cfg, reduction="butterfly", width=_padded(n))
    _assert_same(d2, o2)


def test_primal_svm_with_two_hundred_constraints_matches_twin_bitwise():
    """src/examples/svm_primal_al.cc: 105 variables, 200 inequality constraints (one family), start at the origin with
    penalty 1; a batch of perturbed starts beside it."""
    p, X, y = al.svm_primal_al_problem()
    x0 = np.vstack([np.zeros(p.n), np.random.default_rng(5).uniform(-0.5, 0.5, (7, p.n))])
    cfg = al.default_config(outer_num_iterations=60)
    s = _solver()
    _configure(s, cfg)
    d = s.minimize_host(_engine_problem(p), x0, penalty0=1.0)
    o = al.oracle_minimize(p, x0, penalty0=1.0, config=cfg, reduction="butterfly", width=128)
    _assert_same(d, o)
    dd = X.shape[1]
    w, b = d["x"][0, :dd], d["x"][0, dd]
    assert d["max_violation"][0] <= 1e-4 and np.mean(np.sign(X @ w + b) == y) >= 0.85
    # against the reference order (the reference binary bit for bit on the CPU): the same classifier
    q = al.oracle_minimize(p, x0[:1], penalty0=1.0, config=cfg)
    obj = lambda x: 0.5 * float(x[:dd] @ x[:dd]) + float(np.sum(x[dd + 1:]))
    assert abs(obj(d["x"][0]) - obj(q["x"][0])) <= 1e-3 * max(1.0, abs(obj(q["x"][0])))
    assert np.max(np.abs(d["x"][0, :dd + 1] - q["x"][0, :dd + 1])) <= 1e-2


def test_device_tensor_entry_with_families():
    import torch
    import cppnumericalsolvers_amd as amd
    n = 12
    p = al.random_family_problem(n, 5, 20, seed=12, table=True)
    x0 = np.random.default_rng(1).uniform(-1, 1, (16, n))
    cfg = al.default_config(outer_num_iterations=6)
    s = _solver()
    _configure(s, cfg)
    dev = "cuda:0"
    x = torch.from_numpy(x0.copy()).to(dev)
    lam = torch.zeros(16, p.n_eq, dtype=torch.float64, device=dev)
    mu = torch.zeros(16, p.n_ineq, dtype=torch.float64, device=dev)
    pen = torch.zeros(16, dtype=torch.float64, device=dev)
    viol, kkt, prog = s.minimize(_engine_problem(p), x, lam, mu, pen)
    torch.cuda.synchronize()
    o = al.oracle_minimize(p, x0, config=cfg, reduction="butterfly", width=_padded(n))
    np.testing.assert_array_equal(x.cpu().numpy(), o["x"])
    np.testing.assert_array_equal(lam.cpu().numpy(), o["lambda"])
    np.testing.assert_array_equal(mu.cpu().numpy(), o["mu"])
    np.testing.assert_array_equal(pen.cpu().numpy(), o["penalty"])
    np.testing.assert_array_equal(viol.cpu().numpy(), o["max_violation"])
    np.testing.assert_array_equal(amd.al_progress_to_numpy(prog)["num_iterations"], o["progress"]["num_iterations"])


def test_what_the_family_kernels_are_not_built_for_is_refused():
    from cppnumericalsolvers_amd import capi
    lib = capi.load()
    lib.mi355_auglag_family_capacity.restype = capi.C.c_int32
    assert [lib.mi355_auglag_family_capacity(n) for n in (1, 8, 16, 17, 32, 33, 64, 65, 128, 256, 257, 0)] == \
        [32, 32, 32, 64, 64, 128, 128, 256, 256, 256, 0, 0]
    x0 = np.zeros((2, 10))

    def refused(problem, **kw):
        with pytest.raises(capi.EngineError) as e:
            _solver(**kw).minimize_host(_engine_problem(problem), x0)
        return e.value.code

    over = al.random_family_problem(10, 10, 23, seed=1, table=False)            # 33 > capacity 32 at n = 10
    assert refused(over) == capi.ERR_UNSUPPORTED
    p = al.random_family_problem(10, 2, 6, seed=1, table=False)
    assert refused(p, linesearch="hager_zhang") == capi.ERR_UNSUPPORTED
    assert refused(p, inner="lbfgsb") == capi.ERR_UNSUPPORTED
    s = _solver()
    c = s.default_config()
    c.loop = capi.AL_LOOP["lockstep"]
    s.config = c
    with pytest.raises(capi.EngineError) as e:
        s.minimize_host(_engine_problem(p), x0)
    assert e.value.code == capi.ERR_UNSUPPORTED
    with pytest.raises(capi.EngineError):                                       # per-problem constants with families
        _solver().minimize_host(_engine_problem(p), x0, term_constants=np.zeros((2, 1)))
    ep = _engine_problem(p)                                                     # counts whose sum would wrap an int32
    ps = ep.c_struct()
    ps.n_family_eq = ps.n_family_ineq = 2 ** 31 - 1
    s2 = _solver()
    rc = lib.mi355_auglag_eval_batch_host(s2.ctx.handle, capi.C.byref(ps), 0, None, None, None, None, None, None, None)
    assert rc == capi.ERR_UNSUPPORTED
    ok = _solver()
    ok.config = ok.default_config(outer_num_iterations=2)
    ok.minimize_host(_engine_problem(p), x0)                                    # (the same problem is accepted as it is)


def test_primal_svm_example_through_the_cpp_headers():
    """examples/svm_primal_al/svm_primal_al.cc — the reference example's main() (200 constraint functors pushed into the
    inequality vector of a ConstrainedOptimizationProblem, AugmentedLagrangian over Lbfgs) over the drop-in headers: the
    constraint vector travels as a family."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    build = os.path.join(root, "tests", "cpp", "_build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "svm_primal_al")
    lib = os.path.join(root, "cppnumericalsolvers_amd")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "include"),
                        os.path.join(root, "examples", "svm_primal_al", "svm_primal_al.cc"),
                        "-L" + lib, "-lmi355_lbfgs", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib,
                        "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "active margins" in r.stdout and "PASS" in r.stdout


def test_family_edge_cases_match_twin_bitwise():
    """Shapes and values at the edges: one variable / one constraint, a family at exactly the capacity of its mapping, rows of
    zeros (a constraint that cannot be moved), duplicated rows, a NaN and an infinite start, an empty batch, and multipliers
    that run into multiplier_max."""
    rng = np.random.default_rng(77)
    s = _solver()
    # n = 1, one inequality x >= 0.25 on min 2 x^2
    p1 = al.Problem(1, al.term("diag_quadratic", a=[2.0], c=0.0), family_inequality=(np.array([[1.0]]), np.array([0.25])))
    x0 = np.array([[-1.0], [0.0], [3.0]])
    cfg = al.default_config(outer_num_iterations=12)
    _configure(s, cfg)
    d = s.minimize_host(_engine_problem(p1), x0)
    _assert_same(d, al.oracle_minimize(p1, x0, config=cfg, reduction="butterfly", width=8))
    assert np.all(np.abs(d["x"] - 0.25) <= 1e-3)
    # zero rows (value -k whatever x is), duplicated rows, capacity exactly reached (n = 16: 32 rows)
    n = 16
    G = rng.normal(size=(32, n))
    G[5] = 0.0
    G[9] = G[8]
    k = G @ np.full(n, 0.2) - 0.1
    k[5] = -0.3                                      # 0 . x - (-0.3) = 0.3 >= 0: satisfied, multiplier stays 0
    p2 = al.Problem(n, al.term("diag_quadratic", a=rng.uniform(0.5, 2.0, n), c=0.0), family_inequality=(G, k))
    x0 = rng.uniform(-1, 1, (9, n))
    x0[3, 4] = np.nan                                # hostile starts: the solve must come back, bit for bit like the twin
    x0[6, 0] = np.inf
    d = s.minimize_host(_engine_problem(p2), x0)
    o = al.oracle_minimize(p2, x0, config=cfg, reduction="butterfly", width=16)
    _assert_same(d, o)
    assert np.all(d["mu"][[0, 1, 2, 4, 5, 7, 8], 5] == 0.0)
    # an infeasible pair of rows drives their multipliers into the clamp
    A = np.zeros((2, 4))
    A[0, 0], A[1, 0] = 1.0, -1.0
    p3 = al.Problem(4, al.term("squared_norm"), family_inequality=(A, np.array([1.0, 1.0])))      # x0 >= 1 and -x0 >= 1
    cfgc = al.default_config(outer_num_iterations=25, multiplier_max=50.0)
    _configure(s, cfgc)
    x0 = rng.uniform(-1, 1, (5, 4))
    d = s.minimize_host(_engine_problem(p3), x0)
    _assert_same(d, al.oracle_minimize(p3, x0, config=cfgc, reduction="butterfly", width=8))
    assert np.max(d["mu"]) == 50.0
    # an empty batch
    e = s.minimize_host(_engine_problem(p3), np.zeros((0, 4)))
    assert e["x"].shape == (0, 4) and e["mu"].shape == (0, 2)
