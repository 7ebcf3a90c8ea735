"""Ill-conditioned inputs for the envelope tests of the two relaxed policies (the relaxed-algebra L-BFGS-B kernel and the
normal-equation ridge form).  Shared by tests/test_relaxed_envelope.py (twins vs the reference binary, CPU) and
tests/test_gpu_relaxed_envelope.py (device == twin, GPU) so that both halves of the chain see the same problems."""
import numpy as np

SPREADS = (1e2, 1e4, 1e6, 1e8)                       # max a_i / min a_i of the diagonal quadratic
BOXES = {"rosenbrock-box": (-1.5, 0.8), "lower-bound-active": (0.25, 3.0), "unbounded": None}
RIDGE_CONDITIONS = (1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7)   # cond(A)
RIDGE_LAMBDAS = (0.1, 1e-6)
# where the north star's 1e-6 on x* holds for a re-associated policy (measured): below these the
# reference's own distance to the true minimiser is < 1e-6; above, every summation order (including the reference built
# with another compiler flag) moves x* by about cond(H) x the gradient tolerance
LBFGSB_SPREAD_1E6_BAR = 1e4
RIDGE_COND_H_1E6_BAR = 3e2


def diag_spectrum(n, spread):
    """a_i log-spaced over [1, spread]; params blob = a, c."""
    a = np.logspace(0.0, np.log10(spread), n)
    return a, np.concatenate([a, [1.0]])


def box_arrays(n, box):
    if box is None:
        return None, None
    return np.full(n, box[0]), np.full(n, box[1])


def diag_minimiser(n, box):
    """argmin of sum a_i x_i^2 + c over the box (a_i > 0): the projection of 0."""
    if box is None:
        return np.zeros(n)
    return np.clip(np.zeros(n), box[0], box[1])


def conditioned_matrix(rows, n, cond, seed=11):
    """A = U diag(s) V^T with singular values log-spaced over [1/cond, 1]."""
    rng = np.random.default_rng(seed)
    U, _ = np.linalg.qr(rng.normal(size=(rows, n)))
    V, _ = np.linalg.qr(rng.normal(size=(n, n)))
    s = np.logspace(0.0, -np.log10(cond), n)
    return np.ascontiguousarray((U * s) @ V.T)


def ridge_case(rows, n, cond, B, seed=12):
    A = conditioned_matrix(rows, n, cond)
    Y = np.random.default_rng(seed).normal(size=(B, rows))
    return A, Y


def ridge_hessian_condition(A, lam):
    return float(np.linalg.cond(A.T @ A + lam * np.eye(A.shape[1])))


def tight_stop(O):
    return O.make_stop(num_iterations=10000, x_delta=1e-11, x_delta_violations=1, f_delta=0.0, gradient_norm=1e-8, past=0)
