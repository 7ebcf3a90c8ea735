"""cppnumericalsolvers_amd — MI355X-native batched L-BFGS engine.

The hot path of PatWie/CppNumericalSolvers (Solver::Minimize -> Lbfgs ->
MoreThuente -> objective) rebuilt as hand-written HIP for gfx950 behind a C-ABI
(include/mi355_lbfgs.h).  Importing this package does not load the HIP library;
`capi.load()` / `engine.Context()` do, and fail loudly when it is missing.
"""
from . import _build, capi  # noqa: F401
from .engine import (AugLagComposite, BatchedAugmentedLagrangian, BatchedBfgs, BatchedLbfgs, BatchedLbfgsb, ConstrainedProblem, Context, DeviceGroup, DiagQuadratic, Objective, Rosenbrock, Trace,  # noqa: F401
                     SquaredErrorRidge, SquaredErrorRidgePerProblem, ridge_per_problem_rows, al_progress_to_numpy, parity_stop, progress_to_numpy, synthetic_ridge_host,
                     synthetic_x0_host)

__all__ = ["AugLagComposite", "BatchedAugmentedLagrangian", "ConstrainedProblem", "BatchedBfgs", "BatchedLbfgs", "BatchedLbfgsb", "Context", "DeviceGroup", "Trace", "DiagQuadratic", "Objective", "Rosenbrock", "SquaredErrorRidge", "SquaredErrorRidgePerProblem", "ridge_per_problem_rows", "parity_stop",
           "al_progress_to_numpy", "progress_to_numpy", "synthetic_ridge_host", "synthetic_x0_host", "capi"]
