"""Problem description and ctypes bindings for the augmented-Lagrangian checkers.

TEST INFRASTRUCTURE: `oracle_*` call oracle/_build/liboracle.so (auglag_oracle.hpp),
`ref_*` call oracle/_ref/libref.so (the unmodified reference solver, ref_auglag_capi.cpp).
The problem layout is the one the product C-ABI takes (include/mi355_lbfgs.h): term 0 is the
objective, then the equalities, then the inequalities (g >= 0); term t has kinds[t], forms[t],
ks[t]; a term is the sum of parts[t] primitives, table row r being kinds[r] over coef[r*(n+1) : (r+1)*(n+1)].
"""
import ctypes as C

import numpy as np

import oracle_lib
import ref_lib

KIND = {"rosenbrock": 0, "diag_quadratic": 1, "linear": 2, "squared_norm": 3,
        # a row of a constraint FAMILY (mi355_al_problem.family_*): a.dot(x) as the ascending chain under every reduction
        # policy (the device gives the constraint to one lane); a plain LinearTerm to the reference
        "linear_chain": 50,
        # one residual of a least-squares function, (a.x - c)^2 (MI355_AL_TERM_SQUARED_AFFINE, ABI 9)
        "squared_affine": 4,
        # USER term functors (MI355_AL_TERM_USER): examples/user_al_terms/hs_terms.hpp, compiled into the build of the
        # library that __graft_entry__.build() calls libmi355_lbfgs_hs.so; the oracle and oracle/_ref carry their twins
        "hs024_objective": 100, "product_objective": 101, "hs029_ellipse": 102,
        # examples/user_objective_svm_dual/svm_dual.hpp as a term (libmi355_lbfgs_svm.so, id 103): the objective of the
        # reference's src/examples/svm_dual_al.cc; parameters [n, Q] in Problem.user_params
        "svm_dual": 103}
FORM = {"plain": 0, "value_minus_k": 1, "k_minus_value": 2}


def term(kind, form="plain", k=0.0, a=None, c=0.0, product=False):
    """One primitive as a term; `kind` may also be a list of (kind, a, c) primitives that are summed (F1 + F2 + ...) or,
    product=True with two of them, multiplied (F1 * F2: the reference's ProdExpression)."""
    prims = kind if isinstance(kind, (list, tuple)) else [(kind, a, c)]
    assert not product or len(prims) == 2
    return {"prims": [(p[0], p[1] if len(p) > 1 else None, float(p[2]) if len(p) > 2 else 0.0) for p in prims],
            "form": form, "k": float(k), "product": bool(product)}


def family_terms(pair):
    """(A [F, n], k [F]) -> the F terms `LinearTerm(A[i]) - k[i]` a reference program pushes into its constraint vector."""
    if pair is None:
        return []
    A, k = np.asarray(pair[0], dtype=np.float64), np.asarray(pair[1], dtype=np.float64).ravel()
    assert A.ndim == 2 and A.shape[0] == k.size
    return [term("linear_chain", "value_minus_k", float(k[i]), a=A[i]) for i in range(k.size)]


class Problem:
    def __init__(self, n, objective, equality=(), inequality=(), user_params=None, family_equality=None,
                 family_inequality=None):
        self.n = n
        # the blob of the terms that take their parameters from the problem (mi355_al_problem.user_params)
        self.user_params = None if user_params is None else np.ascontiguousarray(user_params, dtype=np.float64).ravel()
        # constraint families: to the checkers (oracle, reference) they are ordinary terms that FOLLOW the table's terms
        # of their kind; the engine takes them as matrices (family_equality / family_inequality are kept for it)
        self.table_eq, self.table_ineq = list(equality), list(inequality)
        self.family_equality, self.family_inequality = family_equality, family_inequality
        equality = self.table_eq + family_terms(family_equality)
        inequality = self.table_ineq + family_terms(family_inequality)
        self.terms = [objective] + list(equality) + list(inequality)
        self.n_eq, self.n_ineq = len(equality), len(inequality)
        self.parts = np.array([-2 if t.get("product") else len(t["prims"]) for t in self.terms], dtype=np.int32)   # -2: MI355_AL_PARTS_PRODUCT
        prims = [p for t in self.terms for p in t["prims"]]
        self.kinds = np.array([KIND[p[0]] for p in prims], dtype=np.int32)
        self.forms = np.array([FORM[t["form"]] for t in self.terms], dtype=np.int32)
        self.ks = np.array([t["k"] for t in self.terms], dtype=np.float64)
        self.coef = np.zeros((len(prims), n + 1))
        for i, p in enumerate(prims):
            if p[1] is not None:
                self.coef[i, :n] = np.asarray(p[1], dtype=np.float64)
            self.coef[i, n] = p[2]


class Config(C.Structure):
    _fields_ = [("penalty_growth_factor", C.c_double), ("violation_shrink_ratio", C.c_double),
                ("auto_scale_initial_penalty", C.c_int32), ("penalty_auto_objective_scale", C.c_double),
                ("penalty_auto_min", C.c_double), ("penalty_auto_max", C.c_double),
                ("warmup_max_inner_iterations", C.c_int32), ("warmup_inner_gradient_tolerance", C.c_double),
                ("multiplier_max", C.c_double), ("outer_num_iterations", C.c_uint64),
                ("constraint_threshold", C.c_double), ("kkt_stationarity_threshold", C.c_double),
                ("loop", C.c_int32)]


def default_config(**kw):
    """AugmentedLagrangianConfig defaults + the constrained stopping defaults (progress.h:112-126, :353)."""
    c = Config(10.0, 0.25, 1, 10.0, 1e-8, 1e8, 10, 1e-2, 1e20, 10000, 1e-5, 1e-4, 0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


class Progress(C.Structure):
    _fields_ = [("status", C.c_int32), ("num_iterations", C.c_uint32), ("x_delta", C.c_double),
                ("f_delta", C.c_double), ("gradient_norm", C.c_double), ("inner_iterations", C.c_uint64),
                ("nfev", C.c_uint64), ("sum_k", C.c_uint64)]


PROGRESS_DTYPE = np.dtype([("status", np.int32), ("num_iterations", np.uint32), ("x_delta", np.float64),
                           ("f_delta", np.float64), ("gradient_norm", np.float64), ("inner_iterations", np.uint64),
                           ("nfev", np.uint64), ("sum_k", np.uint64)], align=True)
assert PROGRESS_DTYPE.itemsize == C.sizeof(Progress)

_dp = oracle_lib._dp
_ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))


def _state(problem, x0, lambda0, mu0, penalty0):
    x = np.array(x0, dtype=np.float64, order="C", ndmin=2)
    B = x.shape[0]
    lam = np.zeros((B, max(problem.n_eq, 1))) if lambda0 is None else np.array(lambda0, dtype=np.float64, ndmin=2)
    mu = np.zeros((B, max(problem.n_ineq, 1))) if mu0 is None else np.array(mu0, dtype=np.float64, ndmin=2)
    lam = np.ascontiguousarray(np.broadcast_to(lam, (B, lam.shape[1]))[:, :max(problem.n_eq, 0)].copy())
    mu = np.ascontiguousarray(np.broadcast_to(mu, (B, mu.shape[1]))[:, :max(problem.n_ineq, 0)].copy())
    pen = np.ascontiguousarray(np.broadcast_to(np.asarray(penalty0, dtype=np.float64), (B,)).copy())
    return x, lam, mu, pen


_keep = []


def _constants(problem, term_constants, B):
    """null, or the [B][1 + n_eq + n_ineq] array that gives every problem its own constants k."""
    if term_constants is None:
        return None
    tc = np.ascontiguousarray(term_constants, dtype=np.float64)
    assert tc.shape == (B, len(problem.terms))
    _keep.append(tc)
    del _keep[:-8]
    return _dp(tc)


def _result(x, lam, mu, pen, viol, kkt, prog):
    return {"x": x, "lambda": lam, "mu": mu, "penalty": pen, "max_violation": viol,
            "max_lagrangian_gradient": kkt, "progress": prog}


LS = {"more_thuente": 0, "hager_zhang": 1}


def _set_user_params(L, setter, problem):
    up = getattr(problem, "user_params", None)
    fn = getattr(L, setter)
    fn.restype = C.c_int
    if up is None:
        rc = fn(None, C.c_int64(0))
    else:
        rc = fn(_dp(up), C.c_int64(up.size))
    if rc != 0:
        raise ValueError("%s rc=%d" % (setter, rc))


def oracle_minimize(problem, x0, lambda0=None, mu0=None, penalty0=0.0, config=None, inner_stop=None, m=10,
                    reduction="sequential", width=0, nthreads=0, linesearch="more_thuente", term_constants=None, max_violation0=0.0):
    L = oracle_lib.lib()
    _set_user_params(L, "oracle_auglag_set_user_params", problem)
    x, lam, mu, pen = _state(problem, x0, lambda0, mu0, penalty0)
    B, n = x.shape
    cfg = config or default_config()
    st = inner_stop or oracle_lib.default_stop()
    viol, kkt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_violation0, dtype=np.float64), (B,)).copy()), np.empty(B)
    prog = np.zeros(B, dtype=PROGRESS_DTYPE)
    red = 1 if reduction == "butterfly" else 0
    if red and not width:
        width = 1 << max(0, (n - 1).bit_length())
    L.oracle_auglag_minimize_batch.restype = C.c_int
    rc = L.oracle_auglag_minimize_batch(
        C.c_int(n), C.c_int64(B), C.c_int(problem.n_eq), C.c_int(problem.n_ineq), _ip(problem.kinds),
        _ip(problem.forms), _dp(problem.ks), _dp(problem.coef), C.byref(cfg), C.byref(st), C.c_int(m), C.c_int(red),
        C.c_int(width), _dp(x), _dp(lam), _dp(mu), _dp(pen), _dp(viol), _dp(kkt), C.c_void_p(prog.ctypes.data),
        C.c_int(nthreads), C.c_int(LS[linesearch]), _constants(problem, term_constants, B), _ip(problem.parts))
    if rc != 0:
        raise ValueError("oracle_auglag_minimize_batch rc=%d" % rc)
    return _result(x, lam, mu, pen, viol, kkt, prog)


def ref_minimize(problem, x0, lambda0=None, mu0=None, penalty0=0.0, config=None, inner_stop=None,
                 linesearch="more_thuente", term_constants=None, max_violation0=0.0):
    L = ref_lib.lib()
    _set_user_params(L, "ref_auglag_set_user_params", problem)
    x, lam, mu, pen = _state(problem, x0, lambda0, mu0, penalty0)
    B, n = x.shape
    cfg = config or default_config()
    st = inner_stop or oracle_lib.default_stop()
    viol, kkt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_violation0, dtype=np.float64), (B,)).copy()), np.empty(B)
    prog = np.zeros(B, dtype=PROGRESS_DTYPE)
    L.ref_auglag_minimize_batch.restype = C.c_int
    rc = L.ref_auglag_minimize_batch(
        C.c_int(n), C.c_int64(B), C.c_int(problem.n_eq), C.c_int(problem.n_ineq), _ip(problem.kinds),
        _ip(problem.forms), _dp(problem.ks), _dp(problem.coef), C.byref(cfg), C.byref(st), _dp(x), _dp(lam), _dp(mu),
        _dp(pen), _dp(viol), _dp(kkt), C.c_void_p(prog.ctypes.data), C.c_int(LS[linesearch]),
        _constants(problem, term_constants, B), _ip(problem.parts))
    if rc != 0:
        raise ValueError("ref_auglag_minimize_batch rc=%d" % rc)
    return _result(x, lam, mu, pen, viol, kkt, prog)


def _bounds(n, lower, upper):
    if lower is None:
        return None, None
    lo = np.ascontiguousarray(np.broadcast_to(np.asarray(lower, dtype=np.float64), (n,)).copy())
    up = np.ascontiguousarray(np.broadcast_to(np.asarray(upper, dtype=np.float64), (n,)).copy())
    _keep.extend([lo, up])
    return _dp(lo), _dp(up)


def oracle_box_minimize(problem, x0, lower=None, upper=None, lambda0=None, mu0=None, penalty0=0.0, config=None,
                        inner_stop=None, m=5, reduction="sequential", width=0, nthreads=0, linesearch="more_thuente",
                        term_constants=None, std_sort_order=True, max_violation0=0.0):
    """AugmentedLagrangian<Problem, Lbfgsb<F, m>> (inner_stop defaults to the Lbfgsb constructor's stopping test)."""
    L = oracle_lib.lib()
    _set_user_params(L, "oracle_auglag_set_user_params", problem)
    x, lam, mu, pen = _state(problem, x0, lambda0, mu0, penalty0)
    B, n = x.shape
    cfg = config or default_config()
    st = inner_stop or oracle_lib.lbfgsb_default_stop()
    viol, kkt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_violation0, dtype=np.float64), (B,)).copy()), np.empty(B)
    prog = np.zeros(B, dtype=PROGRESS_DTYPE)
    red = 1 if reduction == "butterfly" else 0
    if red and not width:
        width = 1 << max(0, (n - 1).bit_length())
    lo, up = _bounds(n, lower, upper)
    L.oracle_auglag_box_minimize_batch.restype = C.c_int
    rc = L.oracle_auglag_box_minimize_batch(
        C.c_int(n), C.c_int64(B), C.c_int(problem.n_eq), C.c_int(problem.n_ineq), _ip(problem.kinds),
        _ip(problem.forms), _dp(problem.ks), _dp(problem.coef), C.byref(cfg), C.byref(st), C.c_int(m), C.c_int(red),
        C.c_int(width), _dp(x), _dp(lam), _dp(mu), _dp(pen), _dp(viol), _dp(kkt), C.c_void_p(prog.ctypes.data),
        C.c_int(nthreads), C.c_int(LS[linesearch]), _constants(problem, term_constants, B), lo, up,
        C.c_int(1 if std_sort_order else 0), _ip(problem.parts))
    if rc != 0:
        raise ValueError("oracle_auglag_box_minimize_batch rc=%d" % rc)
    return _result(x, lam, mu, pen, viol, kkt, prog)


def ref_box_minimize(problem, x0, lower=None, upper=None, lambda0=None, mu0=None, penalty0=0.0, config=None,
                     inner_stop=None, linesearch="more_thuente", term_constants=None, max_violation0=0.0):
    L = ref_lib.lib()
    _set_user_params(L, "ref_auglag_set_user_params", problem)
    x, lam, mu, pen = _state(problem, x0, lambda0, mu0, penalty0)
    B, n = x.shape
    cfg = config or default_config()
    st = inner_stop or oracle_lib.lbfgsb_default_stop()
    viol, kkt = np.ascontiguousarray(np.broadcast_to(np.asarray(max_violation0, dtype=np.float64), (B,)).copy()), np.empty(B)
    prog = np.zeros(B, dtype=PROGRESS_DTYPE)
    lo, up = _bounds(n, lower, upper)
    L.ref_auglag_box_minimize_batch.restype = C.c_int
    rc = L.ref_auglag_box_minimize_batch(
        C.c_int(n), C.c_int64(B), C.c_int(problem.n_eq), C.c_int(problem.n_ineq), _ip(problem.kinds),
        _ip(problem.forms), _dp(problem.ks), _dp(problem.coef), C.byref(cfg), C.byref(st), _dp(x), _dp(lam), _dp(mu),
        _dp(pen), _dp(viol), _dp(kkt), C.c_void_p(prog.ctypes.data), C.c_int(LS[linesearch]),
        _constants(problem, term_constants, B), lo, up, _ip(problem.parts))
    if rc != 0:
        raise ValueError("ref_auglag_box_minimize_batch rc=%d" % rc)
    return _result(x, lam, mu, pen, viol, kkt, prog)


def oracle_eval(problem, x, lam, mu, penalty, reduction="sequential", width=0, term_constants=None):
    L = oracle_lib.lib()
    _set_user_params(L, "oracle_auglag_set_user_params", problem)
    x, lam, mu, pen = _state(problem, x, lam, mu, penalty)
    B, n = x.shape
    f, g = np.empty(B), np.empty_like(x)
    red = 1 if reduction == "butterfly" else 0
    if red and not width:
        width = 1 << max(0, (n - 1).bit_length())
    L.oracle_auglag_eval.restype = C.c_int
    rc = L.oracle_auglag_eval(C.c_int(n), C.c_int64(B), C.c_int(problem.n_eq), C.c_int(problem.n_ineq),
                              _ip(problem.kinds), _ip(problem.forms), _dp(problem.ks), _dp(problem.coef), C.c_int(red),
                              C.c_int(width), _dp(x), _dp(lam), _dp(mu), _dp(pen), _dp(f), _dp(g),
                              _constants(problem, term_constants, B), _ip(problem.parts))
    if rc != 0:
        raise ValueError("oracle_auglag_eval rc=%d" % rc)
    return f, g


def oracle_composite_minimize(problem, x0, lam, mu, penalty, stop=None, m=10, reduction="sequential", width=0,
                              linesearch="more_thuente"):
    """Lbfgs::Minimize on ToAugmentedLagrangian(problem, (lam, mu), penalty), one row each."""
    L = oracle_lib.lib()
    _set_user_params(L, "oracle_auglag_set_user_params", problem)
    x, lam, mu, pen = _state(problem, x0, lam, mu, penalty)
    B, n = x.shape
    st = stop or oracle_lib.default_stop()
    xo, go, fo = np.empty_like(x), np.empty_like(x), np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    red = 1 if reduction == "butterfly" else 0
    if red and not width:
        width = 1 << max(0, (n - 1).bit_length())
    L.oracle_auglag_composite_minimize.restype = C.c_int
    rc = L.oracle_auglag_composite_minimize(
        C.c_int(n), C.c_int64(B), C.c_int(problem.n_eq), C.c_int(problem.n_ineq), _ip(problem.kinds),
        _ip(problem.forms), _dp(problem.ks), _dp(problem.coef), C.byref(st), C.c_int(m), C.c_int(red), C.c_int(width),
        _dp(x), _dp(lam), _dp(mu), _dp(pen), _dp(xo), _dp(fo), _dp(go), C.c_void_p(prog.ctypes.data),
        C.c_int(LS[linesearch]), _ip(problem.parts))
    if rc != 0:
        raise ValueError("oracle_auglag_composite_minimize rc=%d" % rc)
    return xo, fo, go, prog


# Problems used by the CPU and GPU suites -----------------------------------------------------
def hs016_problem():
    """src/test/augmented_lagrangian_test.cc:1198-1275 (BoxPinnedOptimumStopsOnKkt): 2-D Rosenbrock, x0^2 + x1 >= 0,
    x0 + x1^2 >= 0 — each a sum of two menu primitives — inside the box [-0.5, 0.5] x [-1e20, 1]; optimum (0.5, 0.25)."""
    p = Problem(2, term("rosenbrock"), [],
                [term([("diag_quadratic", [1.0, 0.0]), ("linear", [0.0, 1.0])]),
                 term([("linear", [1.0, 0.0]), ("diag_quadratic", [0.0, 1.0])])])
    return p, np.array([-0.5, -1e20]), np.array([0.5, 1.0])


def quadratic_at_12_problem():
    """:583-621 (BothEqualityAndInequalityActive): min (x0-1)^2 + (x1-2)^2  s.t.  x0 = 0.5,  2 - (x0 + x1) >= 0;
    the objective as diag(1, 1; c = 5) + linear(-2, -4); optimum (0.5, 1.5)."""
    return Problem(2, term([("diag_quadratic", [1.0, 1.0], 5.0), ("linear", [-2.0, -4.0])]),
                   [term("linear", "value_minus_k", 0.5, a=[1.0, 0.0])],
                   [term("linear", "k_minus_value", 2.0, a=[1.0, 1.0])])


def three_part_problem(n, seed=8):
    """Terms of one, two and three primitives with every form."""
    rng = np.random.default_rng(seed)
    return Problem(
        n, term([("rosenbrock",), ("diag_quadratic", rng.uniform(0.1, 0.5, n), 0.25), ("linear", rng.uniform(-1, 1, n))]),
        [term([("linear", rng.uniform(-1, 1, n)), ("squared_norm",)], "value_minus_k", 0.8)],
        [term([("squared_norm",), ("linear", rng.uniform(0, 1, n))], "k_minus_value", 0.5 * n),
         term("linear", a=rng.uniform(0.0, 1.0, n))])


def boxed_rosenbrock_problem(n, seed=2):
    """Chained Rosenbrock on a hyperplane inside a ball, with a box that pins the leading coordinates (after the
    structure of src/test/augmented_lagrangian_test.cc:1198-1275: box by the inner Lbfgsb, the rest by the outer loop)."""
    rng = np.random.default_rng(seed)
    p = Problem(n, term("rosenbrock"), [term("linear", "value_minus_k", 0.5, a=np.ones(n))],
                [term("squared_norm", "k_minus_value", 2.0)])
    lower = np.full(n, -0.5)
    upper = np.where(np.arange(n) < 2, 0.25, 0.6) + 0.0 * rng.uniform(size=n)
    return p, lower, upper


def hs024_problem():
    """src/test/augmented_lagrangian_test.cc:945-1060 (Hs024TriangleEscapesSpuriousOrigin): the user objective
    ((x0-3)^2 - 9) x1^3 / (27 sqrt 3) over the triangle x0/sqrt3 - x1 >= 0, x0 + sqrt3 x1 >= 0, 6 - x0 - sqrt3 x1 >= 0
    (linear menu terms), x >= 0 by the inner Lbfgsb; start (1, 0.5); optimum (3, sqrt 3), f* = -1."""
    r3 = np.sqrt(3.0)
    p = Problem(2, term("hs024_objective"), [],
                [term("linear", a=[1.0 / r3, -1.0]), term("linear", a=[1.0, r3]),
                 term("linear", "k_minus_value", 6.0, a=[1.0, r3])])
    return p, np.array([0.0, 0.0]), np.array([1e20, 1e20])


def hs029_product_problem():
    """Hs029 (:1064-1150) written over the MENU with the reference's ProdExpression (function_expressions.h:260-315)
    instead of user functors: objective `(-x0) * x1` = linear(-1, 0) * linear(0, 1), constraint
    48 - (x0^2 + 2 x1^2) >= 0 as a diagonal quadratic; inner Lbfgs, start (1, 1); optimum (2 sqrt 6, 2 sqrt 3)."""
    return Problem(2, term([("linear", [-1.0, 0.0]), ("linear", [0.0, 1.0])], product=True), [],
                   [term("diag_quadratic", "k_minus_value", 48.0, a=[1.0, 2.0])])


def product_terms_problem(n, seed=5):
    """Products in every position and form: objective Rosenbrock + nothing else, an equality `(a.x) * (b.x) - k = 0`, an
    inequality `k - |x|^2 * (c.x)` >= 0 and a plain product inequality `(d.x) * sum(w_i x_i^2 + 1)` >= 0."""
    rng = np.random.default_rng(seed)
    return Problem(
        n, term("rosenbrock"),
        [term([("linear", rng.uniform(0.2, 1.0, n)), ("linear", rng.uniform(0.2, 1.0, n))], "value_minus_k", 0.3, product=True)],
        [term([("squared_norm",), ("linear", rng.uniform(0.0, 0.5, n))], "k_minus_value", 1.0 * n, product=True),
         term([("linear", rng.uniform(0.1, 1.0, n)), ("diag_quadratic", rng.uniform(0.1, 1.0, n), 1.0)], product=True)])


def hs029_problem():
    """:1064-1150 (Hs029EllipseEscapesOrigin): user objective -x0 x1, user constraint 48 - x0^2 - 2 x1^2 >= 0, inner
    Lbfgs, start (1, 1); optimum (2 sqrt 6, 2 sqrt 3), f* = -12 sqrt 2."""
    return Problem(2, term("product_objective"), [], [term("hs029_ellipse")])


def linear_regression_problem():
    """The augmented-Lagrangian half of src/examples/linear_regression.cc:78-104: the least-squares objective
    (x0 + 2 x1 - 4)^2 + (3 x0 + x1 - 5)^2 as the sum of two squared-affine primitives, the box [0,1] x [1,2] as four
    inequality constraints x0 - 0, x1 - 1, -1 * (x0 - 1) = 1 - x0, -1 * (x1 - 2) = 2 - x1; optimum (1, 1.6)."""
    return Problem(2, term([("squared_affine", [1.0, 2.0], 4.0), ("squared_affine", [3.0, 1.0], 5.0)]), [],
                   [term("linear", "value_minus_k", 0.0, a=[1.0, 0.0]), term("linear", "value_minus_k", 1.0, a=[0.0, 1.0]),
                    term("linear", "k_minus_value", 1.0, a=[1.0, 0.0]), term("linear", "k_minus_value", 2.0, a=[0.0, 1.0])])


def least_squares_problem(n, rows, seed=3):
    """||A x - y||^2 (rows squared-affine primitives) on a hyperplane inside a ball."""
    rng = np.random.default_rng(seed)
    A, y = rng.normal(size=(rows, n)) / np.sqrt(rows), rng.normal(size=rows)
    return Problem(n, term([("squared_affine", A[i], float(y[i])) for i in range(rows)]),
                   [term("linear", "value_minus_k", 0.25, a=np.ones(n) / n)],
                   [term("squared_norm", "k_minus_value", 0.5)])


# --------------------------------------------------------- -----------------------------------------------------
def circle_problem():
    """src/test/verify.cc:290-312 / src/examples/constrained_simple2.cc: min x0 + x1 s.t. |x|^2 = 2, 2 - |x|^2 >= 0."""
    return Problem(2, term("linear", a=[1.0, 1.0]), [term("squared_norm", "value_minus_k", 2.0)],
                   [term("squared_norm", "k_minus_value", 2.0)])


def quadratic_simplex_problem(n, seed=0):
    """min sum a_i x_i^2  s.t.  sum x = 1  and  x_0 <= 0.2  (written 0.2 - x_0 >= 0)."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.5, 4.0, n)
    e0 = np.zeros(n)
    e0[0] = 1.0
    return Problem(n, term("diag_quadratic", a=a, c=0.5), [term("linear", "value_minus_k", 1.0, a=np.ones(n))],
                   [term("linear", "k_minus_value", 0.2, a=e0)])


def rosenbrock_ball_problem(n, radius2=1.5, seed=1):
    """Chained Rosenbrock inside a ball, on a hyperplane: |x|^2 <= radius2, w.x = 0.5."""
    rng = np.random.default_rng(seed)
    w = rng.uniform(-1.0, 1.0, n)
    return Problem(n, term("rosenbrock"), [term("linear", "value_minus_k", 0.5, a=w)],
                   [term("squared_norm", "k_minus_value", radius2)])


def random_problem(n, rng):
    """Random term table: 0-2 equalities, 0-2 inequalities, 1-3 primitives per term, any kind in any position."""
    def prim():
        kind = ["rosenbrock", "diag_quadratic", "linear", "squared_norm", "squared_affine"][rng.integers(0, 5)]
        if kind == "diag_quadratic":
            return (kind, rng.uniform(0.05, 0.6, n), float(rng.uniform(-0.5, 0.5)))
        if kind == "squared_affine":   # (a.x - c)^2, one residual of a least-squares function
            return (kind, rng.uniform(-0.7, 0.7, n), float(rng.uniform(-1.0, 1.0)))
        if kind == "linear":
            return (kind, rng.uniform(-1, 1, n))
        return (kind,)

    def make(scale=1.0):
        form = ["plain", "value_minus_k", "k_minus_value"][rng.integers(0, 3)]
        return term([prim() for _ in range(rng.integers(1, 4))], form, float(rng.uniform(-1, 1) * scale))

    eq = [make() for _ in range(rng.integers(0, 3))]
    ineq = [make(n) for _ in range(rng.integers(0, 3))]
    while sum(len(t["prims"]) for t in eq + ineq) > 12:
        (eq or ineq).pop()
    objective = term([("rosenbrock",)] + [prim() for _ in range(rng.integers(0, 3))])
    return Problem(n, objective, eq, ineq)


def svm_dual_al_problem(N=100, d=4, seed=7, separation=1.2):
    """src/examples/svm_dual_al.cc: min 0.5 a^T Q a - 1^T a  s.t.  sum_i a_i y_i = 0 (the equality, handled by the outer
    loop), 0 <= a <= C (the box, handled by the Lbfgsb inner solver).  Returns (problem, y): the objective is the user
    term `svm_dual` over the blob [n, Q] (tests/svm_data.dual_params), the equality the menu's linear term y . a."""
    import svm_data
    X, y = svm_data.standardised_blobs(N, d, seed, separation)
    blob, _ = svm_data.dual_params(X, y)
    return Problem(N, term("svm_dual"), equality=[term("linear", a=y)], user_params=blob), y


def svm_primal_al_problem(N=100, d=4, seed=7, separation=1.2, C=1.0):
    """src/examples/svm_primal_al.cc:33-147 on the synthetic two-class data of tests/svm_data.py: variables (w [d], b, xi [N]),
        min 0.5 ||w||^2 + C sum(xi)   s.t.   y_i (w . x_i + b) - 1 + xi_i >= 0,   xi_i >= 0       (2 N inequalities),
    the objective as the menu's diagonal quadratic + linear form, the 2 N affine constraints as ONE inequality family
    (row i of the margin block: (y_i x_i, y_i, e_i) with k = 1; of the slack block: e_{d+1+i} with k = 0) — the reference
    example pushes the same 2 N functors into its inequality vector (:139-147).  Returns (problem, X, y)."""
    import svm_data
    X, y = svm_data.standardised_blobs(N, d, seed, separation)
    n = d + 1 + N
    half = np.zeros(n)
    half[:d] = 0.5
    slack = np.zeros(n)
    slack[d + 1:] = C
    A = np.zeros((2 * N, n))
    A[:N, :d] = y[:, None] * X
    A[:N, d] = y
    A[np.arange(N), d + 1 + np.arange(N)] = 1.0
    A[N + np.arange(N), d + 1 + np.arange(N)] = 1.0
    k = np.concatenate([np.ones(N), np.zeros(N)])
    objective = term([("diag_quadratic", half, 0.0), ("linear", slack)])
    return Problem(n, objective, family_inequality=(A, k)), X, y


def random_family_problem(n, f_eq, f_ineq, seed=0, table=True):
    """A convex quadratic with constraint families: min sum a_i x_i^2 + c  s.t.  E x = e (f_eq rows), G x >= h (f_ineq rows),
    and — `table` — one table equality and one table inequality AHEAD of the family rows of their kind (the order the C-ABI
    defines).  The family rows are dense; the point x = 0.3 satisfies every constraint strictly or exactly."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.5, 3.0, n)
    xs = np.full(n, 0.3)
    E = rng.normal(size=(f_eq, n))
    G = rng.normal(size=(f_ineq, n))
    fam_eq = (E, E @ xs) if f_eq else None
    fam_ineq = (G, G @ xs - rng.uniform(0.0, 0.5, f_ineq)) if f_ineq else None
    eq, ineq = [], []
    if table:
        eq = [term("linear", "value_minus_k", float(np.sum(xs)), a=np.ones(n))]
        e0 = np.zeros(n)
        e0[0] = 1.0
        ineq = [term("linear", "k_minus_value", 0.5, a=e0)]
    return Problem(n, term("diag_quadratic", a=a, c=0.25), equality=eq, inequality=ineq, family_equality=fam_eq,
                   family_inequality=fam_ineq)
