"""ctypes binding of oracle/_ref/libref.so: the UNMODIFIED reference headers
(/root/reference/include/cppoptlib) compiled over oracle/eigen_shim.

TEST INFRASTRUCTURE.  The .so is built by `make -C oracle ref` where the
reference tree exists (this container); on the GPU box only the prebuilt binary
is present.  `available()` is False when there is neither.
"""
import ctypes as C
import os
import subprocess

import numpy as np

import oracle_lib

LIB_PATH = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "libref.so")
_lib = None


def available():
    if os.path.exists(LIB_PATH):
        return True
    if os.path.isdir("/root/reference/include/cppoptlib"):
        subprocess.call(["make", "-s", "-C", oracle_lib.ORACLE_DIR, "ref"])
    return os.path.exists(LIB_PATH)


FAST_LIB_PATH = os.path.join(oracle_lib.ORACLE_DIR, "_ref", "libref_o3.so")
_fast = None


def fast_lib():
    """oracle/_ref/libref_o3.so: the reference headers built -O3 -march=x86-64-v3 (timing only, see oracle/Makefile);
    None when it is absent or this CPU lacks AVX2."""
    global _fast
    if _fast is None and os.path.exists(FAST_LIB_PATH):
        try:
            flags = open("/proc/cpuinfo").read()
        except OSError:
            flags = ""
        if " avx2" in flags and " fma" in flags:
            _fast = _bind(C.CDLL(FAST_LIB_PATH), full=False)
    return _fast


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError("oracle/_ref/libref.so is not available")
        _lib = _bind(C.CDLL(LIB_PATH))
    return _lib


def _bind(L, full=True):
    dp = C.POINTER(C.c_double)
    L.ref_lbfgs_minimize_batch.argtypes = [C.c_int, dp, C.c_int, C.c_int, C.c_int64,
                                           C.POINTER(oracle_lib.Stop), dp, dp, dp, dp, C.c_void_p]
    L.ref_lbfgs_minimize_batch.restype = C.c_int
    L.ref_lbfgsb_minimize_batch.argtypes = [C.c_int, dp, C.c_int, C.c_int, C.c_int64,
                                            C.POINTER(oracle_lib.Stop), dp, dp, dp, dp, dp, dp, C.c_void_p]
    L.ref_lbfgsb_minimize_batch.restype = C.c_int
    L.ref_ridge_minimize_batch.argtypes = [dp, C.c_int, C.c_int64, C.POINTER(oracle_lib.Stop), C.c_int,
                                           dp, dp, dp, dp, dp, C.c_void_p]
    L.ref_ridge_minimize_batch.restype = C.c_int
    L.ref_lbfgs_hz_minimize_batch.argtypes = L.ref_lbfgs_minimize_batch.argtypes
    L.ref_lbfgs_hz_minimize_batch.restype = C.c_int
    L.ref_hz_search.argtypes = [C.c_int, dp, C.c_int, C.c_int64, dp, dp, dp, dp, dp, dp, dp]
    L.ref_hz_search.restype = C.c_int
    L.ref_lbfgsb_minimize_batch_ls.argtypes = L.ref_lbfgsb_minimize_batch.argtypes + [C.c_int]
    L.ref_lbfgsb_minimize_batch_ls.restype = C.c_int
    L.ref_bfgs_minimize_batch.argtypes = [C.c_int, dp, C.c_int, C.c_int64, C.POINTER(oracle_lib.Stop), dp, dp, dp, dp,
                                          C.c_void_p, C.c_int]
    L.ref_bfgs_minimize_batch.restype = C.c_int
    L.ref_cstep.argtypes = [dp, C.c_double, C.c_double, C.POINTER(C.c_int), C.c_double, C.c_double,
                            C.POINTER(C.c_int)]
    L.ref_cstep.restype = C.c_int
    L.ref_default_stop.argtypes = [C.POINTER(oracle_lib.Stop), C.c_int]
    if hasattr(L, "ref_ridge_minimize_batch_cond"):
        L.ref_ridge_minimize_batch_cond.argtypes = [dp, C.c_int, C.c_int64, C.POINTER(oracle_lib.Stop), C.c_int, C.c_double,
                                                    dp, dp, dp, dp, dp, C.c_void_p, dp]
        L.ref_ridge_minimize_batch_cond.restype = C.c_int
    if hasattr(L, "ref_lbfgsb_ridge_minimize_batch"):
        L.ref_lbfgsb_ridge_minimize_batch.argtypes = [dp, C.c_int, C.c_int, C.c_int64, C.POINTER(oracle_lib.Stop), dp, dp,
                                                      dp, dp, dp, dp, dp, C.c_void_p]
        L.ref_lbfgsb_ridge_minimize_batch.restype = C.c_int
    if hasattr(L, "ref_ridge_own_matrix_minimize_batch"):
        L.ref_ridge_own_matrix_minimize_batch.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int64, C.POINTER(oracle_lib.Stop),
                                                          dp, dp, dp, dp, dp, C.c_void_p]
        L.ref_ridge_own_matrix_minimize_batch.restype = C.c_int
    if hasattr(L, "ref_svm_minimize_batch"):
        L.ref_svm_minimize_batch.argtypes = [dp, C.c_int, C.c_int, C.c_int64, C.POINTER(oracle_lib.Stop), dp, dp, dp, dp,
                                             C.c_void_p]
        L.ref_svm_minimize_batch.restype = C.c_int
    return L


def hz_search(objective, x, s, alpha_init, params=None):
    """HagerZhang::Search (State overload) of the reference, one call per row."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    s = np.ascontiguousarray(s, dtype=np.float64)
    B, n = x.shape
    a0 = np.ascontiguousarray(np.broadcast_to(np.asarray(alpha_init, dtype=np.float64), (B,)))
    p = np.ascontiguousarray(params if params is not None else np.zeros(1), dtype=np.float64)
    xo, go = np.empty_like(x), np.empty_like(x)
    fo, ao = np.empty(B), np.empty(B)
    rc = lib().ref_hz_search(oracle_lib.OBJ[objective], oracle_lib._dp(p), n, B, oracle_lib._dp(x),
                             oracle_lib._dp(s), oracle_lib._dp(a0), oracle_lib._dp(xo), oracle_lib._dp(fo),
                             oracle_lib._dp(go), oracle_lib._dp(ao))
    if rc != 0:
        raise ValueError("ref_hz_search rc=%d" % rc)
    return xo, fo, go, ao


def minimize_batch(objective, x0, m=10, stop=None, params=None, linesearch="more_thuente"):
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = np.ascontiguousarray(params if params is not None else np.zeros(1), dtype=np.float64)
    x = np.empty_like(x0)
    g = np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    entry = lib().ref_lbfgs_hz_minimize_batch if linesearch == "hager_zhang" else lib().ref_lbfgs_minimize_batch
    rc = entry(oracle_lib.OBJ[objective], oracle_lib._dp(p), n, m, B,
                                        C.byref(stop), oracle_lib._dp(x0), oracle_lib._dp(x),
                                        oracle_lib._dp(f), oracle_lib._dp(g), prog.ctypes.data)
    if rc != 0:
        raise ValueError("ref_lbfgs_minimize_batch rc=%d (m=%d not instantiated?)" % (rc, m))
    return x, f, g, prog


def svm_minimize_batch(params, x0, m=10, stop=None):
    """The reference's Lbfgs<F, m> on the SVM functor of src/examples/svm_primal_lbfgs.cc (params = N, d, C, X, y)."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = np.ascontiguousarray(params, dtype=np.float64)
    x, g = np.empty_like(x0), np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    rc = lib().ref_svm_minimize_batch(oracle_lib._dp(p), n, m, B, C.byref(stop), oracle_lib._dp(x0), oracle_lib._dp(x),
                                      oracle_lib._dp(f), oracle_lib._dp(g), prog.ctypes.data)
    if rc != 0:
        raise ValueError("ref_svm_minimize_batch rc=%d" % rc)
    return x, f, g, prog


def minimize_batch_threaded(objective, x0, m=10, stop=None, params=None, threads=1, chunk=32, library=None,
                            lower=None, upper=None, solver="lbfgs", linesearch="more_thuente"):
    """The reference's Lbfgs (or, with bounds, Lbfgsb; solver="bfgs": its dense Bfgs; linesearch="hager_zhang": its
    Lbfgs<F, m, HagerZhang>) over the rows of x0 on `threads` host threads: the batch is cut
    into chunks of `chunk` problems that a thread pool pulls dynamically (the library's own loop is serial; ctypes
    releases the GIL for the duration of a call).  bench.py's "cpu_reference" leg."""
    from concurrent.futures import ThreadPoolExecutor
    L = library or lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = np.ascontiguousarray(params if params is not None else np.zeros(1), dtype=np.float64)
    x, g = np.empty_like(x0), np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    lo = np.ascontiguousarray(lower, dtype=np.float64) if lower is not None else None
    hi = np.ascontiguousarray(upper, dtype=np.float64) if upper is not None else None
    dp = oracle_lib._dp

    def run(b0):
        b1 = min(B, b0 + chunk)
        if lo is not None:
            return L.ref_lbfgsb_minimize_batch(oracle_lib.OBJ[objective], dp(p), n, m, b1 - b0, C.byref(stop), dp(lo),
                                               dp(hi), dp(x0[b0:b1]), dp(x[b0:b1]), dp(f[b0:b1]), dp(g[b0:b1]),
                                               prog[b0:b1].ctypes.data)
        if solver == "bfgs":
            return L.ref_bfgs_minimize_batch(oracle_lib.OBJ[objective], dp(p), n, b1 - b0, C.byref(stop), dp(x0[b0:b1]),
                                             dp(x[b0:b1]), dp(f[b0:b1]), dp(g[b0:b1]), prog[b0:b1].ctypes.data,
                                             oracle_lib.LINESEARCH[linesearch])
        entry = L.ref_lbfgs_hz_minimize_batch if linesearch == "hager_zhang" else L.ref_lbfgs_minimize_batch
        return entry(oracle_lib.OBJ[objective], dp(p), n, m, b1 - b0, C.byref(stop),
                     dp(x0[b0:b1]), dp(x[b0:b1]), dp(f[b0:b1]), dp(g[b0:b1]), prog[b0:b1].ctypes.data)

    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        rcs = list(pool.map(run, range(0, B, chunk)))
    if any(rcs):
        raise ValueError("reference solve failed (m=%d not instantiated?)" % m)
    return x, f, g, prog


def bfgs_minimize_batch(objective, x0, stop=None, params=None, linesearch="more_thuente"):
    """Bfgs<F, LineSearch>::Minimize of the reference (solver/bfgs.h), one call per row of x0."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = np.ascontiguousarray(params if params is not None else np.zeros(1), dtype=np.float64)
    x, g = np.empty_like(x0), np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    rc = lib().ref_bfgs_minimize_batch(oracle_lib.OBJ[objective], oracle_lib._dp(p), n, B, C.byref(stop),
                                       oracle_lib._dp(x0), oracle_lib._dp(x), oracle_lib._dp(f), oracle_lib._dp(g),
                                       prog.ctypes.data, oracle_lib.LINESEARCH[linesearch])
    if rc != 0:
        raise ValueError("ref_bfgs_minimize_batch rc=%d" % rc)
    return x, f, g, prog


def ridge_minimize_batch(A, lam, Y, x0, stop=None, second_mode=False):
    """The README ridge example on the reference (Lbfgs, m = 10), one row of Y per problem."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = oracle_lib.ridge_params(A, lam)
    x = np.empty_like(x0)
    g = np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    rc = lib().ref_ridge_minimize_batch(oracle_lib._dp(p), n, B, C.byref(stop), 1 if second_mode else 0,
                                        oracle_lib._dp(Y), oracle_lib._dp(x0), oracle_lib._dp(x),
                                        oracle_lib._dp(f), oracle_lib._dp(g), prog.ctypes.data)
    if rc != 0:
        raise ValueError("ref_ridge_minimize_batch rc=%d" % rc)
    return x, f, g, prog


def ridge_minimize_batch_threaded(A, lam, Y, x0, stop=None, threads=1, chunk=32, library=None, second_mode=False):
    """The README ridge example on the reference's Lbfgs (m = 10), one row of Y per problem, on `threads` host threads
    pulling chunks of `chunk` problems (bench.py's "cpu_reference" leg of configs[3])."""
    from concurrent.futures import ThreadPoolExecutor
    L = library or lib()
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = oracle_lib.ridge_params(A, lam)
    x, g = np.empty_like(x0), np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    dp = oracle_lib._dp

    def run(b0):
        b1 = min(B, b0 + chunk)
        return L.ref_ridge_minimize_batch(dp(p), n, b1 - b0, C.byref(stop), 1 if second_mode else 0, dp(Y[b0:b1]), dp(x0[b0:b1]), dp(x[b0:b1]),
                                          dp(f[b0:b1]), dp(g[b0:b1]), prog[b0:b1].ctypes.data)

    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        rcs = list(pool.map(run, range(0, B, chunk)))
    if any(rcs):
        raise ValueError("reference ridge solve failed")
    return x, f, g, prog


def rosenbrock_second_minimize_batch_cond(x0, m=10, stop=None, condition_hessian=0.0):
    """The reference's Lbfgs<RosenbrockNSecond, m> (a Second-mode function with a NON-constant Hessian) with
    stopping_progress.condition_hessian; also returns Progress::condition_hessian after the last Update."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    x, g = np.empty_like(x0), np.empty_like(x0)
    f, cond = np.empty(B), np.zeros(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    L = lib()
    L.ref_rosenbrock_second_minimize_batch_cond.restype = C.c_int
    rc = L.ref_rosenbrock_second_minimize_batch_cond(C.c_int(n), C.c_int(m), C.c_int64(B), C.byref(stop),
                                                     C.c_double(float(condition_hessian)), oracle_lib._dp(x0), oracle_lib._dp(x),
                                                     oracle_lib._dp(f), oracle_lib._dp(g), C.c_void_p(prog.ctypes.data),
                                                     oracle_lib._dp(cond))
    if rc != 0:
        raise ValueError("ref_rosenbrock_second_minimize_batch_cond rc=%d" % rc)
    return x, f, g, prog, cond


def ridge_minimize_batch_cond(A, lam, Y, x0, stop=None, second_mode=True, condition_hessian=0.0):
    """ridge_minimize_batch with stopping_progress.condition_hessian; also returns Progress::condition_hessian."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.default_stop()
    p = oracle_lib.ridge_params(A, lam)
    x, g = np.empty_like(x0), np.empty_like(x0)
    f, cond = np.empty(B), np.zeros(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    rc = lib().ref_ridge_minimize_batch_cond(oracle_lib._dp(p), n, B, C.byref(stop), 1 if second_mode else 0,
                                             float(condition_hessian), oracle_lib._dp(Y), oracle_lib._dp(x0),
                                             oracle_lib._dp(x), oracle_lib._dp(f), oracle_lib._dp(g), prog.ctypes.data,
                                             oracle_lib._dp(cond))
    if rc != 0:
        raise ValueError("ref_ridge_minimize_batch_cond rc=%d" % rc)
    return x, f, g, prog, cond


def lbfgsb_minimize_batch(objective, x0, m=5, stop=None, lower=None, upper=None, linesearch="more_thuente", params=None):
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.lbfgsb_default_stop()
    lo = np.ascontiguousarray(lower, dtype=np.float64) if lower is not None else None
    hi = np.ascontiguousarray(upper, dtype=np.float64) if upper is not None else None
    x = np.empty_like(x0)
    g = np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    p = np.ascontiguousarray(params if params is not None else np.zeros(1), dtype=np.float64)
    rc = lib().ref_lbfgsb_minimize_batch_ls(oracle_lib.OBJ[objective], oracle_lib._dp(p), n, m, B, C.byref(stop),
                                            oracle_lib._dp(lo) if lo is not None else None,
                                            oracle_lib._dp(hi) if hi is not None else None,
                                            oracle_lib._dp(x0), oracle_lib._dp(x), oracle_lib._dp(f),
                                            oracle_lib._dp(g), prog.ctypes.data, oracle_lib.LINESEARCH[linesearch])
    if rc != 0:
        raise ValueError("ref_lbfgsb_minimize_batch rc=%d" % rc)
    return x, f, g, prog


def ridge_own_matrix_minimize_batch(As, lam, Y, x0, stop=None):
    """One `SquaredError(A_b, y_b) + lam * L2Reg` per problem under the reference's Lbfgs<F> (m = 10).  As: [B, rows, n]."""
    As = np.ascontiguousarray(As, dtype=np.float64)
    B, rows, n = As.shape
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    stop = stop or oracle_lib.default_stop()
    data = np.ascontiguousarray(np.concatenate([As.reshape(B, rows * n), np.asarray(Y, dtype=np.float64)], axis=1))
    x, g = np.empty_like(x0), np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    dp = oracle_lib._dp
    rc = lib().ref_ridge_own_matrix_minimize_batch(rows, float(lam), n, B, C.byref(stop), dp(data), dp(x0), dp(x), dp(f),
                                                   dp(g), prog.ctypes.data)
    if rc != 0:
        raise ValueError("ref_ridge_own_matrix_minimize_batch rc=%d" % rc)
    return x, f, g, prog


def lbfgsb_ridge_minimize_batch(A, lam, Y, x0, m=5, stop=None, lower=None, upper=None):
    """The reference's Lbfgsb<F, m> on `SquaredError(A, y_b) + lam * L2Reg` (README.md:126-160 functors in First mode,
    wrapped in a FunctionExpr as src/examples/linear_regression.cc:58-74 does), one row of Y per problem."""
    x0 = np.ascontiguousarray(x0, dtype=np.float64)
    B, n = x0.shape
    stop = stop or oracle_lib.lbfgsb_default_stop()
    params = np.ascontiguousarray(oracle_lib.ridge_params(A, lam), dtype=np.float64)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    lo = np.ascontiguousarray(lower, dtype=np.float64) if lower is not None else None
    hi = np.ascontiguousarray(upper, dtype=np.float64) if upper is not None else None
    x, g = np.empty_like(x0), np.empty_like(x0)
    f = np.empty(B)
    prog = np.zeros(B, dtype=oracle_lib.PROGRESS_DTYPE)
    dp = oracle_lib._dp
    rc = lib().ref_lbfgsb_ridge_minimize_batch(dp(params), n, m, B, C.byref(stop), dp(lo) if lo is not None else None,
                                               dp(hi) if hi is not None else None, dp(Y), dp(x0), dp(x), dp(f), dp(g),
                                               prog.ctypes.data)
    if rc != 0:
        raise ValueError("ref_lbfgsb_ridge_minimize_batch rc=%d" % rc)
    return x, f, g, prog


def cstep(stx, fx, dx, sty, fy, dy, stp, fp, dp, brackt, stpmin, stpmax):
    v = np.array([stx, fx, dx, sty, fy, dy, stp], dtype=np.float64)
    b = C.c_int(1 if brackt else 0)
    info = C.c_int(0)
    rc = lib().ref_cstep(oracle_lib._dp(v), fp, dp, C.byref(b), stpmin, stpmax, C.byref(info))
    return dict(rc=rc, info=info.value, brackt=bool(b.value), stx=v[0], fx=v[1], dx=v[2], sty=v[3],
                fy=v[4], dy=v[5], stp=v[6])


def default_stop(preset="default"):
    s = oracle_lib.Stop()
    lib().ref_default_stop(C.byref(s), 1 if preset == "conservative" else 0)
    return s
