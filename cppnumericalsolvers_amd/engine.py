"""Host-side driver of the batched L-BFGS engine (Python flavour).

`BatchedLbfgs` mirrors `cppoptlib::solver::Lbfgs<FunctionType, m, MoreThuente>`
(reference solver/lbfgs.h:40-45 + solver/solver.h:156-231) for a *batch* of
independent problems: a public `stopping_progress`, a history size `m`, and
`minimize(objective, x0)` returning the final states and the per-problem
progress records.  PyTorch is used only for device memory and the stream; all
arithmetic runs in the HIP kernels behind the C-ABI (cppnumericalsolvers_amd/capi.py).
"""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import capi


@dataclass
class Objective:
    """A device objective: id + shared parameter blob (host doubles)."""
    objective_id: int
    params: np.ndarray = field(default_factory=lambda: np.zeros(0))
    name: str = ""


def Rosenbrock(differentiability="first"):
    """Chained Rosenbrock-N (== reference src/test/verify.cc:58-69 at N = 2).  differentiability="second": the function
    declared Second mode -- Lbfgs then rebuilds its diagonal preconditioner from diag H(x) at every iterate
    (solver/lbfgs.h:116-139); on the device the diagonal comes from the functor's hess_diag (hessian_from_functor)."""
    obj = Objective(capi.OBJ_ROSENBROCK, np.zeros(0), "rosenbrock")
    if differentiability == "second":
        obj.hessian_from_functor = True
    elif differentiability != "first":
        raise ValueError("differentiability: 'first' or 'second'")
    return obj


def DiagQuadratic(a, c=0.0):
    """f(x) = sum_i a_i x_i^2 + c (README.md:21-28 quick start: a=(5,100), c=5)."""
    a = np.asarray(a, dtype=np.float64).ravel()
    return Objective(capi.OBJ_DIAG_QUADRATIC, np.concatenate([a, [float(c)]]), "diag_quadratic")


GRAM_MAX_N, GRAM_MAX_ROWS = 256, 4096   # shapes the normal-equation kernels are built for
GRAM_AUTO_MAX_CONDITION = 3.0e2         # cond(A^T A + lam I) up to which the form is pinned to 1e-6 of the reference


def ridge_condition_bound(A, lam):
    """Upper bound of cond_2(A^T A + lam I) = lambda_max(G) / lambda_min(G), lambda_min(G) >= lam: the smaller of two
    rigorous bounds of lambda_max(G) — the largest Gershgorin row sum of |G|, and trace(G^q)^(1/q) with q = 32 / 16 / 8
    (within n^(1/q) of lambda_max; G scaled by its Gershgorin bound so the powers stay in (0, 1]) — over lam.  The same
    bound as the drop-in headers' NormalEquationConditionBound (include/cppoptlib/mi355/objectives.h)."""
    A = np.asarray(A, dtype=np.float64)
    if not lam > 0.0:
        return float("inf")
    n = A.shape[1]
    G = A.T @ A + float(lam) * np.eye(n)
    gershgorin = float(np.abs(G).sum(axis=1).max())
    if not (gershgorin > 0.0 and np.isfinite(gershgorin)):
        return float("inf")
    M, q = G / gershgorin, 1
    for _ in range(5 if n <= 64 else (4 if n <= 128 else 3)):
        M, q = M @ M, q * 2
    trace = float(np.trace(M))
    by_trace = gershgorin * trace ** (1.0 / q) * (1.0 + 1e-9) if (trace > 0.0 and np.isfinite(trace)) else gershgorin
    return min(gershgorin, by_trace) / float(lam)


def SquaredErrorRidge(A, lam, differentiability="first", matrix_cores=False, gram=False):
    """f(x) = ||A x - y_b||^2 + lam ||x||^2 (README.md:122-167 ridge example); the right-hand
    sides y_b are passed per problem (`per_problem=` of minimize / evaluate).

    differentiability="second" declares the functor Second-mode as the README prints it: Lbfgs
    then centres the two-loop recursion on the diagonal preconditioner 1/(|H_jj| + eps)
    (lbfgs.h:116-139) built from the constant Hessian diagonal
    H_jj = sum_i (2 A_ij) A_ij + lam * 2   (README `hess` of SquaredError and L2Reg)."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    if isinstance(gram, str):
        if gram != "auto":
            raise ValueError("gram: True, False or 'auto'")
        # the normal-equation form inside its pinned envelope only (tests/test_relaxed_envelope.py): within 1e-6 of the
        # reference binary while cond(A^T A + lam I) <= 3e2; bounded here from above without an eigen-solve by
        # Gershgorin rows of G over lam (A^T A is positive semi-definite, so lambda_min(G) >= lam)
        gram = bool(A.shape[1] <= GRAM_MAX_N and A.shape[0] <= GRAM_MAX_ROWS and
                    ridge_condition_bound(A, lam) <= GRAM_AUTO_MAX_CONDITION)
    # matrix_cores=True: the two matrix-vector products of every evaluation run on v_mfma_f64_16x16x4_f64,
    # sixteen problems at a time (objective id 3: FMA chains instead of multiply-then-add sums; same
    # function, results within the 1e-6 tolerance; n <= 64, m <= 10)
    # gram=True: the normal-equation form (objective id 5): one Gram matrix G = A^T A + lam I for the batch, c_b = A^T y_b
    # once per problem on the matrix cores, then n^2 multiply-adds per evaluation in the ordinary Lbfgs kernel
    # (n <= 256, rows <= 4096; "auto": only inside the envelope where it is pinned to 1e-6 of the reference)
    if gram and matrix_cores:
        raise ValueError("gram=True and matrix_cores=True are two different kernels: pick one")
    oid, oname = ((capi.OBJ_SQUARED_ERROR_RIDGE_GRAM, "squared_error_ridge_gram") if gram else
                  (capi.OBJ_SQUARED_ERROR_RIDGE_MFMA, "squared_error_ridge_mfma") if matrix_cores else
                  (capi.OBJ_SQUARED_ERROR_RIDGE, "squared_error_ridge"))
    obj = Objective(oid, np.concatenate([[float(A.shape[0]), float(lam)], A.ravel()]), oname)
    if differentiability == "second":
        acc = (2.0 * A[0]) * A[0]
        for i in range(1, A.shape[0]):   # ascending rows: the order of the reference's product
            acc = acc + (2.0 * A[i]) * A[i]
        obj.hessian_diagonal = np.ascontiguousarray(acc + float(lam) * 2.0)
        # the whole (constant) Hessian 2 A^T A + 2 lam I, for the condition_hessian stopping test (progress.h:203-210)
        H = np.zeros((A.shape[1], A.shape[1]))
        for i in range(A.shape[0]):
            H = H + np.outer(2.0 * A[i], A[i])
        obj.hessian = np.ascontiguousarray(H + np.eye(A.shape[1]) * (float(lam) * 2.0))
    elif differentiability != "first":
        raise ValueError("differentiability must be 'first' or 'second'")
    return obj


def SquaredErrorRidgePerProblem(rows, lam):
    """f_b(x) = ||A_b x - y_b||^2 + lam ||x||^2 with ONE MATRIX PER PROBLEM (objective id 6): what a program computes that
    builds the README objective `SquaredError(A_b, y_b) + lam * L2Reg(n)` once per data set (README.md:126-160).  Pass
    `per_problem=ridge_per_problem_rows(As, Y)` ([B, rows * n + rows]: A_b row major, then y_b) to minimize / evaluate.
    Normal-equation form per problem (G_b, c_b on the matrix cores once, then n^2 multiply-adds per evaluation streamed from
    the problem's own G_b); fused arithmetic, First mode, More-Thuente, n <= 256, rows <= 4096."""
    return Objective(capi.OBJ_SQUARED_ERROR_RIDGE_OWN_GRAM, np.array([float(int(rows)), float(lam)]),
                     "squared_error_ridge_own_gram")


def ridge_per_problem_rows(As, Y):
    """[B, rows, n] matrices and [B, rows] right-hand sides -> the [B, rows * n + rows] per-problem rows of
    SquaredErrorRidgePerProblem."""
    As = np.asarray(As, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    B, rows, n = As.shape
    if Y.shape != (B, rows):
        raise ValueError("Y must be [B, rows]")
    return np.ascontiguousarray(np.concatenate([As.reshape(B, rows * n), Y], axis=1))


def parity_stop():
    """'parity stopping (B)' of SURVEY.md section 7: tight enough for 1e-6 parity on x*."""
    s = capi.default_stop()
    s.num_iterations = 10000
    s.x_delta = 1e-11
    s.x_delta_violations = 1
    s.f_delta = 0.0
    s.gradient_norm = 1e-8
    s.gradient_norm_relative = 1
    s.past = 0
    return s


class Context:
    """One engine context per device (owns scratch + timing events)."""

    def __init__(self, device=0, library=None):
        # `library`: path of another build of the engine, e.g. one with user objectives compiled in
        # (_build.build(output=..., user_objectives=[...])); default: the in-tree library
        self._lib = capi.load(library)   # (imports torch before mapping the engine: one HIP runtime per process)
        h = C.c_void_p()
        capi.check(self._lib.mi355_lbfgs_create(int(device), C.byref(h)))
        self._h = h
        self.device = int(device)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355_lbfgs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h


class Trace:
    """Opt-in per-iteration trace of chosen problems (mi355_lbfgs_trace): what `Solver::step_callback_` of the
    reference observes (solver/solver.h:197, :222), recorded on the device.  `problems`: batch indices (at most 64);
    `capacity`: iterations kept per problem (a ring: the last `capacity` are kept); with_x / with_g: also keep the
    iterate / its gradient after every traced iteration.  Pass it to `minimize(..., trace=t)`; after synchronising,
    `t.history(i)` returns the records of traced problem i in chronological order (+ x, g arrays)."""

    def __init__(self, problems, capacity, n, device, with_x=True, with_g=False):
        import torch
        self.problems = np.ascontiguousarray(problems, dtype=np.int64)
        if not (1 <= self.problems.size <= capi.MAX_TRACED):
            raise ValueError("1..%d traced problems" % capi.MAX_TRACED)
        self.capacity, self.n = int(capacity), int(n)
        K = self.problems.size
        self.records = torch.zeros(K * self.capacity * capi.TRACE_RECORD_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.x = torch.zeros(K, self.capacity, n, dtype=torch.float64, device=device) if with_x else None
        self.g = torch.zeros(K, self.capacity, n, dtype=torch.float64, device=device) if with_g else None
        self.written = torch.zeros(K, dtype=torch.int32, device=device)
        self._c = capi.Trace()
        self._c.count, self._c.capacity = K, self.capacity
        self._c.problems = self.problems.ctypes.data_as(C.POINTER(C.c_int64))
        self._c.records = self.records.data_ptr()
        self._c.x = self.x.data_ptr() if with_x else None
        self._c.g = self.g.data_ptr() if with_g else None
        self._c.written = self.written.data_ptr()

    def c_pointer(self):
        return C.addressof(self._c)

    def history(self, i):
        """(records, x, g) of traced problem i, oldest kept iteration first."""
        w = int(self.written[i].item())
        rec = self.records.cpu().numpy().view(capi.TRACE_RECORD_DTYPE).reshape(-1, self.capacity)[i]
        kept = min(w, self.capacity)
        idx = [(t - 1) % self.capacity for t in range(w - kept + 1, w + 1)]
        x = self.x[i].cpu().numpy()[idx] if self.x is not None else None
        g = self.g[i].cpu().numpy()[idx] if self.g is not None else None
        return rec[idx], x, g


class BatchedLbfgs:
    """Batched `Lbfgs<F, m>` — one problem per wavefront segment on the GPU.

    stopping_progress: capi.Stop (defaults to DefaultStoppingSolverProgress).
    m: history size (reference default 10).
    lanes_per_problem / elems_per_lane: optional explicit wavefront mapping.
    """

    _entry = "mi355_lbfgs_minimize_batch"  # C-ABI entry point (subclasses: other solvers of the same shape)

    def __init__(self, m=10, stopping_progress=None, device=0, lanes_per_problem=0, elems_per_lane=0,
                 context=None, history_placement=0, linesearch="more_thuente", arithmetic="default",
                 condition_hessian=0.0):
        import torch
        self._torch = torch
        self.m = int(m)
        self.stopping_progress = stopping_progress or capi.default_stop()
        self.lanes_per_problem = int(lanes_per_problem)
        self.elems_per_lane = int(elems_per_lane)
        self.history_placement = int(history_placement)
        # the LineSearch template argument of the reference's Lbfgs (lbfgs.h:41)
        self.linesearch = {"more_thuente": capi.LS_MORE_THUENTE, "hager_zhang": capi.LS_HAGER_ZHANG}[linesearch]
        # mi355_arithmetic: "exact" (no FMA, bit-identical to the oracle's butterfly policy), "fma" (fused
        # multiply-adds in the inner products, axpys, trial point and objective; bit-identical to the oracle's
        # butterfly_fma policy, within 1e-6 of the reference-order solve), "default" = fma where it is built
        self.arithmetic = {"default": capi.ARITH_DEFAULT, "exact": capi.ARITH_EXACT, "fma": capi.ARITH_FMA}[arithmetic]
        # stopping_progress.condition_hessian of the reference (progress.h:110): Second-mode objectives only, 0 = off
        self.condition_hessian = float(condition_hessian)
        self.ctx = context or Context(device)
        self.device = torch.device("cuda", self.ctx.device)

    # -- helpers -----------------------------------------------------------
    def _desc(self, objective, n, per_problem=None, per_problem_stride=0):
        d = capi.Desc()
        d.objective = objective.objective_id
        d.linesearch = self.linesearch
        d.n = int(n)
        d.m = self.m
        p = np.ascontiguousarray(objective.params, dtype=np.float64)
        self._params_keepalive = p
        d.objective_params = p.ctypes.data_as(C.POINTER(C.c_double)) if p.size else None
        d.n_params = int(p.size)
        d.per_problem_data = per_problem
        d.per_problem_stride = int(per_problem_stride)
        d.lanes_per_problem = self.lanes_per_problem
        d.elems_per_lane = self.elems_per_lane
        d.history_placement = self.history_placement
        d.arithmetic = self.arithmetic
        if getattr(objective, "hessian_from_functor", False):
            d.hessian_from_functor = 1
            d.hessian_condition_stop = self.condition_hessian   # (n <= 64: the kernel evaluates cond H(x) of every iterate)
        h = getattr(objective, "hessian_diagonal", None)
        if h is not None:
            if h.shape != (int(n),):
                raise ValueError("hessian_diagonal must hold n entries")
            self._hess_keepalive = h
            d.hessian_diagonal = h.ctypes.data_as(C.POINTER(C.c_double))
            H = getattr(objective, "hessian", None)
            if H is not None:
                cond = C.c_double()
                capi.check(self.ctx._lib.mi355_lbfgs_hessian_condition(H.ctypes.data, int(n), C.byref(cond)))
                d.hessian_condition = cond.value
                self.last_hessian_condition = cond.value
            d.hessian_condition_stop = self.condition_hessian
        d.stop = self.stopping_progress
        return d

    def _stream(self):
        return C.c_void_p(self._torch.cuda.current_stream(self.device).cuda_stream)

    # -- API ---------------------------------------------------------------
    def _on_device(self, t, what):
        """The C entry points dereference raw pointers on the context's device."""
        if t.device != self.device:
            raise ValueError("%s lives on %s, the solver's context on %s" % (what, t.device, self.device))

    def _pp_device(self, per_problem, B):
        if per_problem is None:
            return None, 0
        torch = self._torch
        if per_problem.dtype != torch.float64 or per_problem.dim() != 2 or per_problem.shape[0] != B \
                or not per_problem.is_cuda:
            raise ValueError("per_problem must be a [B, stride] float64 CUDA tensor")
        self._on_device(per_problem, "per_problem")
        pp = per_problem.contiguous()
        self._pp_keepalive = pp
        return pp.data_ptr(), pp.shape[1]

    def minimize(self, objective, x0, want_gradient=True, want_progress=True, per_problem=None, trace=None):
        """Batched Solver::Minimize.  x0: [B, n] float64 tensor on this device.

        Returns (x, f, g, progress) — device tensors; progress is a uint8 tensor
        of B*40 bytes viewable with `progress_to_numpy`.  Asynchronous on the
        current stream.
        """
        torch = self._torch
        if x0.dtype != torch.float64 or x0.dim() != 2 or not x0.is_cuda:
            raise ValueError("x0 must be a [B, n] float64 CUDA tensor")
        self._on_device(x0, "x0")
        x0 = x0.contiguous()
        B, n = x0.shape
        x = torch.empty_like(x0)
        f = torch.empty(B, dtype=torch.float64, device=x0.device)
        g = torch.empty_like(x0) if want_gradient else None
        prog = torch.empty(B * capi.PROGRESS_DTYPE.itemsize, dtype=torch.uint8, device=x0.device) \
            if want_progress else None
        d = self._desc(objective, n, *self._pp_device(per_problem, B))
        if trace is not None:
            self._trace_keepalive = trace
            d.trace = trace.c_pointer()
        capi.check(getattr(self.ctx._lib, self._entry)(
            self.ctx.handle, C.byref(d), B, x0.data_ptr(), x.data_ptr(), f.data_ptr(),
            g.data_ptr() if g is not None else None, prog.data_ptr() if prog is not None else None,
            self._stream()))
        return x, f, g, prog

    def minimize_host(self, objective, x0, per_problem=None):
        """Same through the host-pointer entry point (numpy in, numpy out, synchronous)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B, n = x0.shape
        pp_ptr, pp_stride = None, 0
        if per_problem is not None:
            pp = np.ascontiguousarray(per_problem, dtype=np.float64)
            self._pp_keepalive = pp
            pp_ptr, pp_stride = pp.ctypes.data, pp.shape[1]
        x = np.empty_like(x0)
        g = np.empty_like(x0)
        f = np.empty(B)
        prog = np.zeros(B, dtype=capi.PROGRESS_DTYPE)
        d = self._desc(objective, n, pp_ptr, pp_stride)
        capi.check(getattr(self.ctx._lib, self._entry + "_host")(
            self.ctx.handle, C.byref(d), B, x0.ctypes.data, x.ctypes.data, f.ctypes.data, g.ctypes.data,
            prog.ctypes.data))
        return x, f, g, prog

    def evaluate(self, objective, x, per_problem=None):
        """One objective evaluation per row of x (device functor parity tests)."""
        torch = self._torch
        self._on_device(x, "x")
        x = x.contiguous()
        B, n = x.shape
        f = torch.empty(B, dtype=torch.float64, device=x.device)
        g = torch.empty_like(x)
        d = self._desc(objective, n, *self._pp_device(per_problem, B))
        capi.check(self.ctx._lib.mi355_lbfgs_eval_batch(
            self.ctx.handle, C.byref(d), B, x.data_ptr(), f.data_ptr(), g.data_ptr(), self._stream()))
        return f, g

    def hz_search(self, objective, x, direction, alpha_init, per_problem=None):
        """One HagerZhang::Search (hager_zhang.h:100-116) per row of x along the rows of `direction`;
        returns the accepted x, f, g, the step widths and the evaluation counts."""
        torch = self._torch
        for t, what in ((x, "x"), (direction, "direction"), (alpha_init, "alpha_init")):
            self._on_device(t, what)
        x = x.contiguous()
        direction = direction.contiguous()
        B, n = x.shape
        a0 = alpha_init.contiguous()
        xo, go = torch.empty_like(x), torch.empty_like(x)
        fo = torch.empty(B, dtype=torch.float64, device=x.device)
        ao = torch.empty(B, dtype=torch.float64, device=x.device)
        nf = torch.empty(B, dtype=torch.int32, device=x.device)
        d = self._desc(objective, n, *self._pp_device(per_problem, B))
        capi.check(self.ctx._lib.mi355_lbfgs_hz_search_batch(
            self.ctx.handle, C.byref(d), B, x.data_ptr(), direction.data_ptr(), a0.data_ptr(), xo.data_ptr(),
            fo.data_ptr(), go.data_ptr(), ao.data_ptr(), nf.data_ptr(), self._stream()))
        return xo, fo, go, ao, nf

    def fill_x0(self, B, n, kind="std", seed=20260923, first_problem=0):
        """Seeded synthetic start points generated on the device."""
        torch = self._torch
        x0 = torch.empty(B, n, dtype=torch.float64, device=self.device)
        capi.check(self.ctx._lib.mi355_lbfgs_fill_x0(
            self.ctx.handle, 0 if kind == "std" else 1, seed, first_problem, B, n, x0.data_ptr(),
            self._stream()))
        return x0

    def last_kernel_ms(self):
        ms = C.c_float()
        capi.check(self.ctx._lib.mi355_lbfgs_last_kernel_ms(self.ctx.handle, C.byref(ms)))
        return float(ms.value)

    def last_arithmetic(self):
        """'exact' or 'fma': what the most recent solve on this solver's context ran with."""
        v = C.c_int32()
        capi.check(self.ctx._lib.mi355_lbfgs_last_arithmetic(self.ctx.handle, C.byref(v)))
        return {capi.ARITH_EXACT: "exact", capi.ARITH_FMA: "fma"}[v.value]

    def last_launch(self):
        v = [C.c_int32() for _ in range(6)]
        capi.check(self.ctx._lib.mi355_lbfgs_last_launch(self.ctx.handle, *[C.byref(t) for t in v]))
        out = dict(zip(("lanes_per_problem", "elems_per_lane", "blocks", "threads", "lds_bytes",
                        "y_columns_in_registers"),
                       [t.value for t in v]))
        return out


class BatchedBfgs(BatchedLbfgs):
    """Batched dense `Bfgs<F, LineSearch>` (reference solver/bfgs.h): the same driver, line searches and
    stopping tests with an explicit inverse-Hessian approximation per problem; n <= 64."""
    _entry = "mi355_bfgs_minimize_batch"

    def __init__(self, stopping_progress=None, device=0, context=None, linesearch="more_thuente",
                 lanes_per_problem=0, elems_per_lane=0):
        # mapping 0 x 0: the library's choice (one column of H per lane at the padded widths 32 and 64); an explicit split
        # must cover exactly the padded width with a built shape (include/mi355_lbfgs.h) — same bits either way
        super().__init__(m=1, stopping_progress=stopping_progress, device=device, context=context,
                         linesearch=linesearch, arithmetic="exact", lanes_per_problem=lanes_per_problem,
                         elems_per_lane=elems_per_lane)


class BatchedLbfgsb(BatchedLbfgs):
    """Batched `Lbfgsb<F, m>` (reference solver/lbfgsb.h, default m = 5; built for m <= 10): box-constrained L-BFGS-B.

    `SetBounds(lower, upper)` mirrors the reference (lbfgsb.h:89-93); without it the box is
    unbounded.  stopping_progress defaults to what a default-constructed reference Lbfgsb uses
    (f_delta = 2.22e-9 relative on top of the default preset); its gradient_norm is the
    projected-gradient tolerance."""

    def __init__(self, m=5, stopping_progress=None, device=0, context=None, linesearch="more_thuente",
                 arithmetic="default"):
        # arithmetic: "exact" = the reference-order kernels (lbfgsb_kernel.hpp, bit-identical to the oracle's butterfly
        # policy), "fma" = the relaxed-algebra kernels (lbfgsb_fast_kernel.hpp, bit-identical to their own CPU twin,
        # within 1e-6 of the reference), "default" = the library's choice (relaxed where it is built)
        super().__init__(m=m, linesearch=linesearch, arithmetic=arithmetic,
                         stopping_progress=stopping_progress or capi.default_stop("lbfgsb"),
                         device=device, context=context)
        self._lower = None
        self._upper = None

    def SetBounds(self, lower, upper):
        if np.isnan(np.asarray(lower, dtype=np.float64)).any() or np.isnan(np.asarray(upper, dtype=np.float64)).any():
            raise ValueError("NaN bound (the breakpoint order of the Cauchy search would be undefined, as in the reference)")
        torch = self._torch
        self._lower = torch.as_tensor(np.ascontiguousarray(lower, dtype=np.float64)).to(self.device)
        self._upper = torch.as_tensor(np.ascontiguousarray(upper, dtype=np.float64)).to(self.device)

    def minimize(self, objective, x0, want_gradient=True, want_progress=True, per_problem=None, trace=None):
        torch = self._torch
        if x0.dtype != torch.float64 or x0.dim() != 2 or not x0.is_cuda:
            raise ValueError("x0 must be a [B, n] float64 CUDA tensor")
        self._on_device(x0, "x0")
        x0 = x0.contiguous()
        B, n = x0.shape
        if self._lower is not None and self._lower.numel() != n:
            raise ValueError("bounds have %d entries, problems have %d" % (self._lower.numel(), n))
        x = torch.empty_like(x0)
        f = torch.empty(B, dtype=torch.float64, device=x0.device)
        g = torch.empty_like(x0) if want_gradient else None
        prog = torch.empty(B * capi.PROGRESS_DTYPE.itemsize, dtype=torch.uint8, device=x0.device) \
            if want_progress else None
        d = self._desc(objective, n, *self._pp_device(per_problem, B))
        if trace is not None:
            self._trace_keepalive = trace
            d.trace = trace.c_pointer()
        capi.check(self.ctx._lib.mi355_lbfgsb_minimize_batch(
            self.ctx.handle, C.byref(d),
            self._lower.data_ptr() if self._lower is not None else None,
            self._upper.data_ptr() if self._upper is not None else None,
            B, x0.data_ptr(), x.data_ptr(), f.data_ptr(),
            g.data_ptr() if g is not None else None, prog.data_ptr() if prog is not None else None,
            self._stream()))
        return x, f, g, prog

    def minimize_host(self, objective, x0, per_problem=None):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B, n = x0.shape
        x = np.empty_like(x0)
        g = np.empty_like(x0)
        f = np.empty(B)
        prog = np.zeros(B, dtype=capi.PROGRESS_DTYPE)
        d = self._desc(objective, n)
        lo = self._lower.cpu().numpy() if self._lower is not None else None
        hi = self._upper.cpu().numpy() if self._upper is not None else None
        capi.check(self.ctx._lib.mi355_lbfgsb_minimize_batch_host(
            self.ctx.handle, C.byref(d), lo.ctypes.data if lo is not None else None,
            hi.ctypes.data if hi is not None else None, B, x0.ctypes.data, x.ctypes.data, f.ctypes.data,
            g.ctypes.data, prog.ctypes.data))
        return x, f, g, prog


class DeviceGroup:
    """mi355_lbfgs_group: one engine context per entry of `devices` plus an RCCL communicator over the distinct devices.
    `minimize_host` shards a host batch into contiguous ranges, one host thread per member, and returns the
    all-reduced convergence record (problems, unconverged, iterations) — the path's only collective."""

    def __init__(self, devices):
        self._lib = capi.load()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        capi.check(self._lib.mi355_lbfgs_group_create(devs, len(devices), C.byref(h)))
        self._h = h
        self.devices = [int(d) for d in devices]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355_lbfgs_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def size(self):
        return int(self._lib.mi355_lbfgs_group_size(self._h))

    def minimize_host(self, solver, objective, x0, per_problem=None):
        """`solver`: a BatchedLbfgs carrying m / stopping / line search / arithmetic (its own context is not used)."""
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B, n = x0.shape
        pp_ptr, pp_stride = None, 0
        if per_problem is not None:
            pp = np.ascontiguousarray(per_problem, dtype=np.float64)
            pp_ptr, pp_stride = pp.ctypes.data, pp.shape[1]
        x, g, f = np.empty_like(x0), np.empty_like(x0), np.empty(B)
        prog = np.zeros(B, dtype=capi.PROGRESS_DTYPE)
        flag = np.zeros(3, dtype=np.uint64)
        d = solver._desc(objective, n, pp_ptr, pp_stride)
        capi.check(self._lib.mi355_lbfgs_group_minimize_batch_host(
            self._h, C.byref(d), B, x0.ctypes.data, x.ctypes.data, f.ctypes.data, g.ctypes.data, prog.ctypes.data,
            flag.ctypes.data))
        return x, f, g, prog, {"total": int(flag[0]), "unconverged": int(flag[1]), "iterations": int(flag[2]),
                               "all_converged": int(flag[1]) == 0}


    def _host_call(self, entry, solver, objective, x0, per_problem, extra=()):
        x0 = np.ascontiguousarray(x0, dtype=np.float64)
        B, n = x0.shape
        pp_ptr, pp_stride = None, 0
        if per_problem is not None:
            pp = np.ascontiguousarray(per_problem, dtype=np.float64)
            pp_ptr, pp_stride = pp.ctypes.data, pp.shape[1]
        x, g, f = np.empty_like(x0), np.empty_like(x0), np.empty(B)
        prog = np.zeros(B, dtype=capi.PROGRESS_DTYPE)
        flag = np.zeros(3, dtype=np.uint64)
        d = solver._desc(objective, n, pp_ptr, pp_stride)
        capi.check(getattr(self._lib, entry)(self._h, C.byref(d), *extra, B, x0.ctypes.data, x.ctypes.data, f.ctypes.data,
                                             g.ctypes.data, prog.ctypes.data, flag.ctypes.data))
        return x, f, g, prog, self._flag(flag)

    @staticmethod
    def _flag(flag):
        return {"total": int(flag[0]), "unconverged": int(flag[1]), "iterations": int(flag[2]),
                "all_converged": int(flag[1]) == 0}

    def minimize_host_lbfgsb(self, solver, objective, x0, lower=None, upper=None, per_problem=None):
        """mi355_lbfgsb_group_minimize_batch_host: `solver` a BatchedLbfgsb (m / stopping / arithmetic)."""
        lo = np.ascontiguousarray(lower, dtype=np.float64) if lower is not None else None
        hi = np.ascontiguousarray(upper, dtype=np.float64) if upper is not None else None
        return self._host_call("mi355_lbfgsb_group_minimize_batch_host", solver, objective, x0, per_problem,
                               (lo.ctypes.data if lo is not None else None, hi.ctypes.data if hi is not None else None))

    def minimize_host_bfgs(self, solver, objective, x0, per_problem=None):
        return self._host_call("mi355_bfgs_group_minimize_batch_host", solver, objective, x0, per_problem)

    def minimize_device(self, solver, objective, x0_shards, lower=None, upper=None, per_problem_shards=None):
        """Device-resident sharded solve (mi355_lbfgs_group_minimize_batch / mi355_lbfgsb_group_minimize_batch):
        x0_shards[s] is a [B_s, n] float64 CUDA tensor on member s's device; lower / upper (host arrays) select Lbfgsb.
        Returns per-member (x, f, g, progress) tensors and the all-reduced convergence record."""
        import torch
        G = self.size()
        if len(x0_shards) != G:
            raise ValueError("one shard per group member")
        n = int(x0_shards[0].shape[1])
        outs, keep = [], []
        ptr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() if t is not None and t.numel() else None for t in ts])
        for s, x0 in enumerate(x0_shards):
            if x0.dtype != torch.float64 or x0.dim() != 2 or not x0.is_cuda or int(x0.shape[1]) != n:
                raise ValueError("shards must be [B_s, %d] float64 CUDA tensors" % n)
            x0_shards[s] = x0.contiguous()
            Bs = int(x0.shape[0])
            outs.append((torch.empty_like(x0), torch.empty(Bs, dtype=torch.float64, device=x0.device), torch.empty_like(x0),
                         torch.zeros(max(Bs, 1) * capi.PROGRESS_DTYPE.itemsize, dtype=torch.uint8, device=x0.device)))
        counts = (C.c_int64 * G)(*[int(t.shape[0]) for t in x0_shards])
        pps = None
        d = solver._desc(objective, n)
        if per_problem_shards is not None:
            pps = [p.contiguous() for p in per_problem_shards]
            d.per_problem_stride = int(pps[0].shape[1])
        flag = np.zeros(3, dtype=np.uint64)
        args = [counts, ptr(x0_shards), ptr([o[0] for o in outs]), ptr([o[1] for o in outs]), ptr([o[2] for o in outs]),
                ptr([o[3] for o in outs]), ptr(pps) if pps is not None else None, flag.ctypes.data]
        for t in x0_shards:                      # inputs produced on torch's streams must be complete
            torch.cuda.synchronize(t.device)
        if lower is not None:
            for t in x0_shards:
                keep.append((torch.as_tensor(np.ascontiguousarray(lower, dtype=np.float64)).to(t.device),
                             torch.as_tensor(np.ascontiguousarray(upper, dtype=np.float64)).to(t.device)))
                torch.cuda.synchronize(t.device)
            capi.check(self._lib.mi355_lbfgsb_group_minimize_batch(self._h, C.byref(d), ptr([k[0] for k in keep]),
                                                                   ptr([k[1] for k in keep]), *args))
        elif getattr(solver, "_entry", "") == "mi355_lbfgsb_minimize_batch" or isinstance(solver, BatchedLbfgsb):
            capi.check(self._lib.mi355_lbfgsb_group_minimize_batch(self._h, C.byref(d), None, None, *args))
        else:
            capi.check(self._lib.mi355_lbfgs_group_minimize_batch(self._h, C.byref(d), *args))
        return outs, self._flag(flag)


class ConstrainedProblem:
    """`ConstrainedOptimizationProblem` (function_problem.h:44-74) over the device term menu.

    Terms are built with `ConstrainedProblem.term`: a primitive (kind in capi.AL_TERM, or the id of a user term functor
    compiled into the library: MI355_AL_TERM_USER; `a` the coefficient vector of
    the linear / diagonal-quadratic kinds, `c` the constant of the diagonal quadratic) or a sum of primitives, entering
    as form in capi.AL_FORM ("plain" F, "value_minus_k" F - k, "k_minus_value" k - F).  `inequality` constraints
    mean g(x) >= 0, as in the reference.
    """

    @staticmethod
    def term(kind, form="plain", k=0.0, a=None, c=0.0, product=False):
        """One primitive (kind, a, c) as a term, or — `kind` a list of (kind, a, c) tuples — the sum F1 + F2 + ... of
        several (the reference's AddExpression, left to right); product=True with two tuples: their product F1 * F2
        (the reference's ProdExpression, function_expressions.h:260-315)."""
        prims = kind if isinstance(kind, (list, tuple)) else [(kind, a, c)]
        if product and len(prims) != 2:
            raise ValueError("a product term has exactly two primitives")
        return ([(p[0], p[1] if len(p) > 1 else None, float(p[2]) if len(p) > 2 else 0.0) for p in prims], form, float(k),
                bool(product))

    def __init__(self, n, objective, equality=(), inequality=(), user_params=None, family_equality=None,
                 family_inequality=None):
        # user_params: the blob handed to the user term functors that declare kTermParamsFromProblem (the parameters
        # the same functor takes as an objective; mi355_al_problem.user_params)
        self.user_params = None if user_params is None else np.ascontiguousarray(user_params, dtype=np.float64).ravel()
        # constraint FAMILIES (mi355_al_problem.family_eq / family_ineq): (A [F, n], k [F]) stands for the F affine
        # constraints A[i] . x - k[i] (= 0 / >= 0) — what a reference program pushes one `LinearFunctor(a_i) - k_i` at a
        # time into its constraint vectors (src/examples/svm_primal_al.cc:139-147); they follow the table's terms of
        # their kind in the problem's constraint order (and in lambda / mu)

        def family(pair):
            if pair is None:
                return np.zeros((0, int(n) + 1))
            A, k = np.asarray(pair[0], dtype=np.float64), np.asarray(pair[1], dtype=np.float64).ravel()
            if A.ndim != 2 or A.shape != (k.size, int(n)):
                raise ValueError("a constraint family is (A [F, n], k [F])")
            return np.ascontiguousarray(np.concatenate([A, k[:, None]], axis=1))

        self.family_eq, self.family_ineq = family(family_equality), family(family_inequality)
        terms = [objective] + list(equality) + list(inequality)
        if len(equality) > capi.AL_MAX_CONSTRAINTS or len(inequality) > capi.AL_MAX_CONSTRAINTS:
            raise ValueError("at most %d constraints of each kind" % capi.AL_MAX_CONSTRAINTS)
        prims = [p for t in terms for p in t[0]]
        if len(prims) > capi.AL_MAX_ROWS:
            raise ValueError("at most %d primitives" % capi.AL_MAX_ROWS)
        self.n, self.n_eq, self.n_ineq = int(n), len(equality), len(inequality)
        self.parts = np.array([capi.AL_PARTS_PRODUCT if (len(t) > 3 and t[3]) else len(t[0]) for t in terms], dtype=np.int32)
        # a primitive's kind: a name of the menu, or the objective id (>= capi.AL_TERM_USER) of a user term functor
        self.kinds = np.array([p[0] if isinstance(p[0], (int, np.integer)) else capi.AL_TERM[p[0]] for p in prims],
                              dtype=np.int32)
        self.forms = np.array([capi.AL_FORM[t[1]] for t in terms], dtype=np.int32)
        self.ks = np.array([t[2] for t in terms], dtype=np.float64)
        self.coef = np.zeros((len(prims), self.n + 1))
        for i, p in enumerate(prims):
            if p[1] is not None:
                self.coef[i, :self.n] = np.asarray(p[1], dtype=np.float64)
            self.coef[i, self.n] = p[2]

    def c_struct(self):
        p = capi.AlProblem()
        p.n, p.n_eq, p.n_ineq = self.n, self.n_eq, self.n_ineq
        p.kinds = self.kinds.ctypes.data_as(C.POINTER(C.c_int32))
        p.forms = self.forms.ctypes.data_as(C.POINTER(C.c_int32))
        p.ks = self.ks.ctypes.data_as(C.POINTER(C.c_double))
        p.coef = self.coef.ctypes.data_as(C.POINTER(C.c_double))
        p.parts = self.parts.ctypes.data_as(C.POINTER(C.c_int32))
        if self.user_params is not None and self.user_params.size:
            p.user_params = self.user_params.ctypes.data_as(C.POINTER(C.c_double))
            p.user_params_count = int(self.user_params.size)
        p.n_family_eq, p.n_family_ineq = self.family_eq.shape[0], self.family_ineq.shape[0]
        if p.n_family_eq:
            p.family_eq = self.family_eq.ctypes.data_as(C.POINTER(C.c_double))
        if p.n_family_ineq:
            p.family_ineq = self.family_ineq.ctypes.data_as(C.POINTER(C.c_double))
        return p

    @property
    def n_eq_all(self):     # width of lambda: the table's equalities, then the family's
        return self.n_eq + self.family_eq.shape[0]

    @property
    def n_ineq_all(self):   # width of mu
        return self.n_ineq + self.family_ineq.shape[0]


def AugLagComposite(problem):
    """`ToAugmentedLagrangian(problem, multipliers, penalty)` (function_penalty.h:239-246) as an objective for
    BatchedLbfgs: pass the rows (lambda, mu, penalty) as `per_problem=` of minimize / minimize_host.

    The composite objective (MI355_OBJ_AL_COMPOSITE) is described by the term TABLE only; a problem that also carries
    constraint families (family_equality / family_inequality) is refused — silently dropping them would hand the caller
    the value and gradient of a different function (round-5 advisor finding)."""
    if problem.family_eq.shape[0] or problem.family_ineq.shape[0]:
        raise ValueError("AugLagComposite: the problem carries constraint families (%d equalities, %d inequalities); the "
                         "composite objective holds table terms only — solve it with BatchedAugmentedLagrangian"
                         % (problem.family_eq.shape[0], problem.family_ineq.shape[0]))
    terms = np.column_stack([problem.parts.astype(np.float64), problem.forms.astype(np.float64), problem.ks])
    rows = np.concatenate([problem.kinds[:, None].astype(np.float64), problem.coef], axis=1)
    head = [float(problem.n_eq), float(problem.n_ineq), float(len(problem.kinds))]
    return Objective(capi.OBJ_AL_COMPOSITE, np.concatenate([head, terms.ravel(), rows.ravel()]), "al_composite")


class BatchedAugmentedLagrangian:
    """Batched `AugmentedLagrangian<Problem, Lbfgs<FunctionExpr, m>>` (solver/augmented_lagrangian.h).

    config: capi.AlConfig (default: the reference's AugmentedLagrangianConfig and constrained stopping defaults);
    inner_stopping_progress: the inner solver's stopping_progress (default DefaultStoppingSolverProgress).
    """

    def __init__(self, m=None, config=None, inner_stopping_progress=None, device=0, context=None,
                 linesearch="more_thuente", inner="lbfgs", lower=None, upper=None):
        # inner="lbfgsb": Lbfgsb<F, m> (default m = 5) as the inner solver, `lower` / `upper` its SetBounds box (n
        # doubles each, or None = never set); its default stopping test is the Lbfgsb constructor's
        if inner not in ("lbfgs", "lbfgsb"):
            raise ValueError("inner must be 'lbfgs' or 'lbfgsb'")
        self.box = inner == "lbfgsb"
        if (lower is None) != (upper is None) or (lower is not None and not self.box):
            raise ValueError("lower / upper come together and need inner='lbfgsb'")
        self.lower = None if lower is None else np.ascontiguousarray(lower, dtype=np.float64)
        self.upper = None if upper is None else np.ascontiguousarray(upper, dtype=np.float64)
        if m is None:
            m = 5 if self.box else 10
        self.m = int(m)
        # the LineSearch template argument of the inner Lbfgs (lbfgs.h:41)
        self.linesearch = {"more_thuente": capi.LS_MORE_THUENTE, "hager_zhang": capi.LS_HAGER_ZHANG}[linesearch]
        self.ctx = context or Context(device)
        self.config = config or self.default_config()
        self.inner_stopping_progress = inner_stopping_progress or capi.default_stop()
        if self.box and inner_stopping_progress is None:  # Lbfgsb(): f_delta = 2.22e-9, relative (lbfgsb.h:84-87)
            self.inner_stopping_progress.f_delta = 2.22e-9
            self.inner_stopping_progress.f_delta_relative = 1

    def _bounds(self, n):
        if self.lower is None:
            return None, None
        if self.lower.shape != (n,) or self.upper.shape != (n,):
            raise ValueError("lower / upper must hold n entries")
        return self.lower.ctypes.data, self.upper.ctypes.data

    def default_config(self, **overrides):
        c = capi.AlConfig()
        capi.check(self.ctx._lib.mi355_auglag_default_config(C.byref(c)))
        for k, v in overrides.items():
            setattr(c, k, v)
        return c

    @staticmethod
    def _state(problem, x0, lambda0, mu0, penalty0):
        x = np.array(x0, dtype=np.float64, order="C", ndmin=2)
        B = x.shape[0]

        def rows(v, width):
            if width == 0:
                return np.zeros((B, 0))
            v = np.zeros((B, width)) if v is None else np.asarray(v, dtype=np.float64)
            return np.ascontiguousarray(np.broadcast_to(v.reshape(-1, width) if v.ndim else v, (B, width)).copy())

        pen = np.ascontiguousarray(np.broadcast_to(np.asarray(penalty0, dtype=np.float64), (B,)).copy())
        return x, rows(lambda0, problem.n_eq_all), rows(mu0, problem.n_ineq_all), pen

    @staticmethod
    def _constants(problem, term_constants, B):
        if term_constants is None:
            return None
        tc = np.ascontiguousarray(term_constants, dtype=np.float64)
        if tc.shape != (B, 1 + problem.n_eq + problem.n_ineq):
            raise ValueError("term_constants must be [B, 1 + n_eq + n_ineq]")
        return tc

    def minimize_host(self, problem, x0, lambda0=None, mu0=None, penalty0=0.0, term_constants=None, max_violation0=0.0):
        """numpy in, dict of numpy out (x, lambda, mu, penalty, max_violation, max_lagrangian_gradient, progress).
        term_constants [B, 1 + n_eq + n_ineq]: row b replaces the constants k of the problem's terms, so the batch
        is B different problems of one shape rather than B starts of one problem."""
        x, lam, mu, pen = self._state(problem, x0, lambda0, mu0, penalty0)
        B = x.shape[0]
        tc = self._constants(problem, term_constants, B)
        # max_violation is in/out: the incoming state's value (0 for a fresh state) feeds the first penalty-growth test
        viol = np.ascontiguousarray(np.broadcast_to(np.asarray(max_violation0, dtype=np.float64), (B,)).copy())
        kkt = np.empty(B)
        prog = np.zeros(B, dtype=capi.AL_PROGRESS_DTYPE)
        ps = problem.c_struct()
        head = (self.ctx.handle, C.byref(ps), C.byref(self.config), C.byref(self.inner_stopping_progress), self.m,
                self.linesearch)
        tail = (B, tc.ctypes.data if tc is not None else None, x.ctypes.data, lam.ctypes.data if lam.size else None,
                mu.ctypes.data if mu.size else None, pen.ctypes.data, viol.ctypes.data, kkt.ctypes.data,
                prog.ctypes.data)
        if self.box:
            capi.check(self.ctx._lib.mi355_auglag_box_minimize_batch_host(*head, *self._bounds(problem.n), *tail))
        else:
            capi.check(self.ctx._lib.mi355_auglag_minimize_batch_host(*head, *tail))
        return {"x": x, "lambda": lam, "mu": mu, "penalty": pen, "max_violation": viol,
                "max_lagrangian_gradient": kkt, "progress": prog}

    def minimize(self, problem, x, lam, mu, penalty, term_constants=None, max_violation=None):
        """Device tensors, updated in place: x [B, n], lam [B, n_eq (+ family)], mu [B, n_ineq (+ family)], penalty [B] (float64, CUDA);
        term_constants: optional [B, 1 + n_eq + n_ineq] device tensor (see minimize_host).
        Returns (max_violation, max_lagrangian_gradient, progress bytes) as device tensors."""
        import torch
        B = x.shape[0]
        if term_constants is not None and tuple(term_constants.shape) != (B, 1 + problem.n_eq + problem.n_ineq):
            raise ValueError("term_constants must be [B, 1 + n_eq + n_ineq]")
        for t in (x, lam, mu, penalty, term_constants):
            if t is not None and (t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous()):
                raise ValueError("state tensors must be contiguous float64 CUDA tensors")
        # in/out: the incoming states' max_violation (a tensor returned by an earlier call), zeros for fresh states
        viol = max_violation if max_violation is not None else torch.zeros(B, dtype=torch.float64, device=x.device)
        kkt = torch.empty_like(viol)
        prog = torch.empty(B * capi.AL_PROGRESS_DTYPE.itemsize, dtype=torch.uint8, device=x.device)
        ps = problem.c_struct()
        stream = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        head = (self.ctx.handle, C.byref(ps), C.byref(self.config), C.byref(self.inner_stopping_progress), self.m,
                self.linesearch)
        tail = (B, term_constants.data_ptr() if term_constants is not None else None, x.data_ptr(),
                lam.data_ptr() if problem.n_eq_all else None, mu.data_ptr() if problem.n_ineq_all else None,
                penalty.data_ptr(), viol.data_ptr(), kkt.data_ptr(), prog.data_ptr(), stream)
        if self.box:
            capi.check(self.ctx._lib.mi355_auglag_box_minimize_batch(*head, *self._bounds(problem.n), *tail))
        else:
            capi.check(self.ctx._lib.mi355_auglag_minimize_batch(*head, *tail))
        return viol, kkt, prog

    def evaluate_host(self, problem, x, lam=None, mu=None, penalty=0.0, term_constants=None):
        """Value and gradient of ToAugmentedLagrangian(problem, (lam, mu), penalty) at every row of x."""
        x, lam, mu, pen = self._state(problem, x, lam, mu, penalty)
        B = x.shape[0]
        tc = self._constants(problem, term_constants, B)
        f, g = np.empty(B), np.empty_like(x)
        ps = problem.c_struct()
        capi.check(self.ctx._lib.mi355_auglag_eval_batch_host(
            self.ctx.handle, C.byref(ps), B, tc.ctypes.data if tc is not None else None, x.ctypes.data,
            lam.ctypes.data if lam.size else None,
            mu.ctypes.data if mu.size else None, pen.ctypes.data, f.ctypes.data, g.ctypes.data))
        return f, g


def al_progress_to_numpy(prog):
    """Device progress bytes of BatchedAugmentedLagrangian.minimize -> numpy records (capi.AL_PROGRESS_DTYPE)."""
    return prog.cpu().numpy().view(capi.AL_PROGRESS_DTYPE)


def progress_to_numpy(prog):
    """Device uint8 progress buffer -> numpy record array (copies to host)."""
    return prog.cpu().numpy().view(capi.PROGRESS_DTYPE)


def _splitmix64(z):
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _normal_from_counter(seed, idx):
    """N(0,1) from a counter: two splitmix64 uniforms -> Box-Muller (host only)."""
    a = _splitmix64(np.uint64(seed) ^ (idx * np.uint64(2)))
    b = _splitmix64(np.uint64(seed) ^ (idx * np.uint64(2) + np.uint64(1)))
    u1 = ((a >> np.uint64(11)).astype(np.float64) + 1.0) * (1.0 / 9007199254740992.0)  # (0, 1]
    u2 = (b >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def synthetic_ridge_host(B, rows=128, n=64, seed=20260923, first_problem=0):
    """Config-4 inputs: one shared A (rows x n) with N(0,1)/sqrt(rows)
    entries and one right-hand side y_b ~ N(0,1)^rows per problem; counter based, so shards of
    a global batch line up.  Returns (A, Y[B, rows])."""
    ia = np.arange(rows * n, dtype=np.uint64)
    A = (_normal_from_counter(seed ^ 0xA11CE, ia) / np.sqrt(float(rows))).reshape(rows, n)
    iy = (np.arange(first_problem, first_problem + B, dtype=np.uint64)[:, None] * np.uint64(rows)
          + np.arange(rows, dtype=np.uint64)[None, :])
    Y = _normal_from_counter(seed ^ 0xB0B, iy)
    return A, Y


def synthetic_x0_host(B, n, kind="std", seed=20260923, first_problem=0):
    """Host twin of mi355_lbfgs_fill_x0 (bit-identical), for CPU-side checks."""
    idx = (np.arange(first_problem, first_problem + B, dtype=np.uint64)[:, None] * np.uint64(n)
           + np.arange(n, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        z = np.uint64(seed) ^ idx
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    if kind == "std":
        base = np.where(np.arange(n) % 2 == 1, 1.0, -1.2)[None, :]
        return base + 0.1 * (2.0 * u - 1.0)
    return -2.0 + 4.0 * u
