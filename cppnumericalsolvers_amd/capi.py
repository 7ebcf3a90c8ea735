"""ctypes binding of the C-ABI declared in include/mi355_lbfgs.h.

This module only loads libmi355_lbfgs.so (the HIP engine) and declares its
entry points.  There is no fallback of any kind: if the library is missing,
`load()` raises, and if no MI355X is visible `mi355_lbfgs_create` fails with
MI355_ERR_NO_DEVICE.
"""
import ctypes as C
import os

import numpy as np

from . import _build

MI355_OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_HIP = -2
ERR_NO_DEVICE = -3
ERR_UNSUPPORTED = -4

OBJ_ROSENBROCK = 0
OBJ_DIAG_QUADRATIC = 1
OBJ_SQUARED_ERROR_RIDGE = 2
OBJ_SQUARED_ERROR_RIDGE_MFMA = 3
OBJ_AL_COMPOSITE = 4
OBJ_SQUARED_ERROR_RIDGE_GRAM = 5
OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM = 6   # one matrix per problem: params = rows, lambda; per-problem rows = A_b, y_b
OBJ_USER_FIRST = 100
MAX_ROWS = 128
LS_MORE_THUENTE = 0
LS_HAGER_ZHANG = 1
HISTORY_AUTO, HISTORY_LDS, HISTORY_Y_IN_REGISTERS = 0, 1, 2
ARITH_DEFAULT, ARITH_EXACT, ARITH_FMA = 0, 1, 2

MAX_PAST = 8
MAX_N = 256
MAX_M = 32

# Every symbol include/mi355_lbfgs.h declares (tests check that all are exported).
EXPORTED_SYMBOLS = [
    "mi355_lbfgs_abi_version", "mi355_auglag_family_capacity", "mi355_lbfgs_create", "mi355_lbfgs_destroy", "mi355_lbfgs_last_error",
    "mi355_lbfgs_default_stop", "mi355_lbfgs_minimize_batch", "mi355_lbfgs_minimize_batch_host",
    "mi355_lbfgsb_minimize_batch", "mi355_lbfgsb_minimize_batch_host",
    "mi355_bfgs_minimize_batch", "mi355_bfgs_minimize_batch_host",
    "mi355_lbfgs_last_kernel_ms", "mi355_lbfgs_last_launch", "mi355_lbfgs_last_arithmetic", "mi355_lbfgs_hessian_condition",
    "mi355_lbfgs_fill_x0",
    "mi355_lbfgs_eval_batch", "mi355_lbfgs_hz_search_batch", "mi355_lbfgs_hz_search_host", "mi355_lbfgs_cstep_batch", "mi355_lbfgs_cstep_host", "mi355_lbfgs_selftest",
    "mi355_auglag_default_config", "mi355_auglag_minimize_batch", "mi355_auglag_minimize_batch_host",
    "mi355_auglag_eval_batch_host", "mi355_auglag_box_minimize_batch", "mi355_auglag_box_minimize_batch_host",
    "mi355_lbfgs_group_create", "mi355_lbfgs_group_destroy", "mi355_lbfgs_group_size", "mi355_lbfgs_group_context",
    "mi355_lbfgs_group_minimize_batch_host", "mi355_lbfgs_group_allreduce_flags",
    "mi355_lbfgsb_group_minimize_batch_host", "mi355_bfgs_group_minimize_batch_host",
    "mi355_lbfgs_group_minimize_batch", "mi355_lbfgsb_group_minimize_batch",
]


class Stop(C.Structure):
    """mi355_lbfgs_stop — stopping fields of cppoptlib::solver::Progress (progress.h:87-136)."""
    _fields_ = [
        ("num_iterations", C.c_uint64),
        ("x_delta", C.c_double),
        ("x_delta_violations", C.c_int32),
        ("f_delta", C.c_double),
        ("f_delta_violations", C.c_int32),
        ("f_delta_relative", C.c_int32),
        ("gradient_norm", C.c_double),
        ("gradient_norm_relative", C.c_int32),
        ("past", C.c_int32),
        ("past_delta", C.c_double),
    ]


class Desc(C.Structure):
    """mi355_lbfgs_desc."""
    _fields_ = [
        ("objective", C.c_int32),
        ("linesearch", C.c_int32),
        ("n", C.c_int32),
        ("m", C.c_int32),
        ("objective_params", C.POINTER(C.c_double)),
        ("n_params", C.c_int32),
        ("per_problem_data", C.c_void_p),
        ("per_problem_stride", C.c_int32),
        ("lanes_per_problem", C.c_int32),
        ("elems_per_lane", C.c_int32),
        ("history_placement", C.c_int32),
        ("arithmetic", C.c_int32),
        ("hessian_from_functor", C.c_int32),
        ("hessian_diagonal", C.POINTER(C.c_double)),
        ("trace", C.c_void_p),
        ("hessian_condition", C.c_double),
        ("hessian_condition_stop", C.c_double),
        ("stop", Stop),
    ]


class Trace(C.Structure):
    """mi355_lbfgs_trace."""
    _fields_ = [("count", C.c_int32), ("capacity", C.c_int32), ("problems", C.POINTER(C.c_int64)),
                ("records", C.c_void_p), ("x", C.c_void_p), ("g", C.c_void_p), ("written", C.c_void_p)]


TRACE_RECORD_DTYPE = np.dtype([("num_iterations", "<u4"), ("status", "<i4"), ("value", "<f8"), ("x_delta", "<f8"),
                               ("f_delta", "<f8"), ("gradient_norm", "<f8")], align=True)
assert TRACE_RECORD_DTYPE.itemsize == 40
MAX_TRACED = 64
LBFGSB_RELAXED_MAX_SPREAD = 1.0e4   # MI355_LBFGSB_RELAXED_MAX_SPREAD (include/mi355_lbfgs.h)


AL_MAX_CONSTRAINTS = 4
AL_MAX_ROWS = 16
AL_PARTS_PRODUCT = -2   # MI355_AL_PARTS_PRODUCT: the term is the product of its two primitives
AL_TERM = {"rosenbrock": 0, "diag_quadratic": 1, "linear": 2, "squared_norm": 3, "squared_affine": 4}
AL_TERM_USER = 100   # kinds >= this: objective id of a user functor compiled in as a term (MI355_AL_TERM_USER)
AL_FORM = {"plain": 0, "value_minus_k": 1, "k_minus_value": 2}


class AlProblem(C.Structure):
    """mi355_al_problem — ConstrainedOptimizationProblem over the device term menu (host pointers)."""
    _fields_ = [("n", C.c_int32), ("n_eq", C.c_int32), ("n_ineq", C.c_int32),
                ("kinds", C.POINTER(C.c_int32)), ("forms", C.POINTER(C.c_int32)),
                ("ks", C.POINTER(C.c_double)), ("coef", C.POINTER(C.c_double)), ("parts", C.POINTER(C.c_int32)),
                ("user_params", C.POINTER(C.c_double)), ("user_params_count", C.c_int64),
                # constraint families (ABI 8): rows (a_i, k_i) of affine constraints a_i . x - k_i beyond the table's
                ("n_family_eq", C.c_int32), ("n_family_ineq", C.c_int32),
                ("family_eq", C.POINTER(C.c_double)), ("family_ineq", C.POINTER(C.c_double))]


class AlConfig(C.Structure):
    """mi355_al_config — AugmentedLagrangianConfig + the constrained stopping thresholds."""
    _fields_ = [("penalty_growth_factor", C.c_double), ("violation_shrink_ratio", C.c_double),
                ("auto_scale_initial_penalty", C.c_int32), ("penalty_auto_objective_scale", C.c_double),
                ("penalty_auto_min", C.c_double), ("penalty_auto_max", C.c_double),
                ("warmup_max_inner_iterations", C.c_int32), ("warmup_inner_gradient_tolerance", C.c_double),
                ("multiplier_max", C.c_double), ("outer_num_iterations", C.c_uint64),
                ("constraint_threshold", C.c_double), ("kkt_stationarity_threshold", C.c_double),
                ("loop", C.c_int32)]


AL_LOOP = {"auto": 0, "fused": 1, "lockstep": 2}


AL_PROGRESS_DTYPE = np.dtype(
    [("status", "<i4"), ("num_iterations", "<u4"), ("x_delta", "<f8"), ("f_delta", "<f8"),
     ("gradient_norm", "<f8"), ("inner_iterations", "<u8"), ("nfev", "<u8"), ("sum_k", "<u8")], align=True)
assert AL_PROGRESS_DTYPE.itemsize == 56

# mi355_lbfgs_progress as a numpy record (40 bytes, natural alignment).
PROGRESS_DTYPE = np.dtype(
    [("status", "<i4"), ("num_iterations", "<u4"), ("nfev", "<u4"), ("sum_k", "<u4"),
     ("x_delta", "<f8"), ("f_delta", "<f8"), ("gradient_norm", "<f8")], align=True)
assert PROGRESS_DTYPE.itemsize == 40

_lib = None


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("mi355_lbfgs error %d: %s" % (code, message))
        self.code = code


def lib_path():
    """The in-tree build, or another build of the same sources named by MI355_LBFGS_LIBRARY
    (A/B runs of kernel variants, scripts/ab_variants.sh)."""
    return os.environ.get("MI355_LBFGS_LIBRARY") or _build.LIB_PATH


_loaded = {}   # path -> bound library (builds with user objectives compiled in are loaded next to the default one)


def load(path=None):
    """dlopen the HIP engine (the in-tree build, or the build at `path`); raises if it has not been built."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    default = path is None
    path = os.path.abspath(path or lib_path())
    if path in _loaded:
        return _loaded[path]
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950). There is no CPU fallback." % path)
    # torch first: it ships its own HIP runtime, and a process must end up with ONE — whichever runtime is
    # initialised second finds no device.  With torch's already mapped, the engine's libamdhip64 dependency
    # resolves to it.
    import torch  # noqa: F401
    L = _bind(C.CDLL(path))
    _loaded[path] = L
    if default:
        _lib = L
    return L


def _bind(L):
    vp = C.c_void_p
    L.mi355_lbfgs_abi_version.restype = C.c_int
    L.mi355_lbfgs_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mi355_lbfgs_destroy.argtypes = [vp]
    L.mi355_lbfgs_destroy.restype = None
    L.mi355_lbfgs_last_error.restype = C.c_char_p
    L.mi355_lbfgs_default_stop.argtypes = [C.c_int, C.POINTER(Stop)]
    L.mi355_lbfgs_minimize_batch.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mi355_lbfgs_minimize_batch_host.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp, vp]
    L.mi355_bfgs_minimize_batch.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mi355_bfgs_minimize_batch_host.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp, vp]
    L.mi355_lbfgsb_minimize_batch.argtypes = [vp, C.POINTER(Desc), vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mi355_lbfgsb_minimize_batch_host.argtypes = [vp, C.POINTER(Desc), vp, vp, C.c_int64, vp, vp, vp, vp, vp]
    L.mi355_lbfgs_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.mi355_lbfgs_last_launch.argtypes = [vp] + [C.POINTER(C.c_int32)] * 6
    L.mi355_lbfgs_last_arithmetic.argtypes = [vp, C.POINTER(C.c_int32)]
    L.mi355_lbfgs_hessian_condition.argtypes = [vp, C.c_int32, C.POINTER(C.c_double)]
    L.mi355_lbfgs_hz_search_batch.argtypes = [vp, C.POINTER(Desc), C.c_int64] + [vp] * 9
    L.mi355_lbfgs_hz_search_host.argtypes = [vp, C.POINTER(Desc), C.c_int64] + [vp] * 8
    L.mi355_lbfgs_fill_x0.argtypes = [vp, C.c_int32, C.c_uint64, C.c_int64, C.c_int64, C.c_int32, vp, vp]
    L.mi355_lbfgs_eval_batch.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp]
    L.mi355_lbfgs_cstep_batch.argtypes = [vp, C.c_int64, vp, vp, vp]
    L.mi355_lbfgs_cstep_host.argtypes = [vp, C.c_int64, vp, vp]
    L.mi355_lbfgs_selftest.argtypes = [vp, vp, vp, vp, vp]
    L.mi355_auglag_default_config.argtypes = [C.POINTER(AlConfig)]
    L.mi355_auglag_minimize_batch.argtypes = [vp, C.POINTER(AlProblem), C.POINTER(AlConfig), C.POINTER(Stop),
                                              C.c_int32, C.c_int32, C.c_int64] + [vp] * 9
    L.mi355_auglag_minimize_batch_host.argtypes = [vp, C.POINTER(AlProblem), C.POINTER(AlConfig), C.POINTER(Stop),
                                                   C.c_int32, C.c_int32, C.c_int64] + [vp] * 8
    L.mi355_auglag_box_minimize_batch.argtypes = [vp, C.POINTER(AlProblem), C.POINTER(AlConfig), C.POINTER(Stop),
                                                  C.c_int32, C.c_int32, vp, vp, C.c_int64] + [vp] * 9
    L.mi355_auglag_box_minimize_batch_host.argtypes = [vp, C.POINTER(AlProblem), C.POINTER(AlConfig),
                                                       C.POINTER(Stop), C.c_int32, C.c_int32, vp, vp,
                                                       C.c_int64] + [vp] * 8
    L.mi355_auglag_eval_batch_host.argtypes = [vp, C.POINTER(AlProblem), C.c_int64] + [vp] * 7
    L.mi355_lbfgs_group_create.argtypes = [C.POINTER(C.c_int), C.c_int, C.POINTER(vp)]
    L.mi355_lbfgs_group_destroy.argtypes = [vp]
    L.mi355_lbfgs_group_destroy.restype = None
    L.mi355_lbfgs_group_size.argtypes = [vp]
    L.mi355_lbfgs_group_context.argtypes = [vp, C.c_int]
    L.mi355_lbfgs_group_context.restype = vp
    L.mi355_lbfgs_group_minimize_batch_host.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mi355_lbfgs_group_allreduce_flags.argtypes = [vp, vp, vp, vp]
    L.mi355_lbfgsb_group_minimize_batch_host.argtypes = [vp, C.POINTER(Desc), vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mi355_bfgs_group_minimize_batch_host.argtypes = [vp, C.POINTER(Desc), C.c_int64, vp, vp, vp, vp, vp, vp]
    L.mi355_lbfgs_group_minimize_batch.argtypes = [vp, C.POINTER(Desc)] + [vp] * 8
    L.mi355_lbfgsb_group_minimize_batch.argtypes = [vp, C.POINTER(Desc)] + [vp] * 10
    for name in EXPORTED_SYMBOLS:
        if name not in ("mi355_lbfgs_destroy", "mi355_lbfgs_last_error", "mi355_lbfgs_abi_version",
                        "mi355_lbfgs_group_destroy", "mi355_lbfgs_group_context"):
            getattr(L, name).restype = C.c_int
    return L


def check(rc):
    if rc != MI355_OK:
        msg = b""
        for L in list(_loaded.values()) or [load()]:   # the failing call's library holds the text
            msg = L.mi355_lbfgs_last_error() or msg
            if msg:
                break
        raise EngineError(rc, msg.decode() if msg else "")


def default_stop(preset="default"):
    """'default' / 'conservative' (solver/progress.h:353-464) or 'lbfgsb' (lbfgsb.h:84-87)."""
    s = Stop()
    check(load().mi355_lbfgs_default_stop({"default": 0, "conservative": 1, "lbfgsb": 2}[preset], C.byref(s)))
    return s
