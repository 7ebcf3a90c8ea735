"""Cycle breakdown of the L-BFGS-B iteration (config 5 shape) from a profiling build of the library:

  python -c "from cppnumericalsolvers_amd import _build as b; b.build(extra_flags=['-DMI355_LBFGSB_PHASE_TIMING'], output=b.PKG_DIR + '/variants/lib_phases.so')"
  MI355_LBFGS_LIBRARY=$PWD/cppnumericalsolvers_amd/variants/lib_phases.so python scripts/lbfgsb_phases.py [B] [exact|fma]

Every wavefront sums s_memtime deltas per phase; the table shows each phase's share of the
wavefront-resident time (all wavefronts, whole launch)."""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

PHASES = ["fetch / prologue / exit", "clip + projected gradient", "Cauchy: breakpoints, p = W^T d, M^-1 p",
          "Cauchy: breakpoint loop", "subspace: M^-1 c, r, WZ r, M^-1 WZ r", "subspace: WZ WZ^T",
          "subspace: N = I - M^-1 N, LU(N), v", "subspace: du, alpha*", "line search",
          "history: shift, S^T Y / S^T S", "MM assembly + LU", "Progress::Update + results"]
PHASES_RELAXED = ["fetch / prologue / exit", "clip + projected gradient", "Cauchy: breakpoints, p = W^T d, M^-1 p",
                  "Cauchy: breakpoint loop", "subspace: r, WZ r", "subspace: K = K0 + active rank-one terms",
                  "subspace: v = K^-1 WZ r", "subspace: du, alpha*", "line search",
                  "history: ring slot, S^T Y / S^T S / Y^T Y", "MM, K0 assembly + LU", "Progress::Update + results"]
import bench
wl = bench.WORKLOADS["cfg5"]
B, n = (int(sys.argv[1]) if len(sys.argv) > 1 else wl["B"]), wl["n"]
ARITH = sys.argv[2] if len(sys.argv) > 2 else "exact"
if ARITH != "exact":
    PHASES = PHASES_RELAXED
s = amd.BatchedLbfgsb(arithmetic=ARITH, m=wl["m"], stopping_progress=bench.lbfgsb_tight_stop(amd.capi.default_stop("lbfgsb")))
s.SetBounds(np.full(n, wl["lower"]), np.full(n, wl["upper"]))
x0 = s.fill_x0(B, n, wl["x0"], bench.SEED)
x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
lib = s.ctx._lib
lib.mi355_lbfgsb_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
amd.capi.check(lib.mi355_lbfgsb_phase_cycles(s.ctx.handle, out))
cyc = np.array(list(out), dtype=np.float64)
it = amd.progress_to_numpy(p)["num_iterations"]
print("kernel %.2f ms, %d problems, mean iterations %.1f" % (s.last_kernel_ms(), B, it.mean()))
for name, c in zip(PHASES, cyc):
    print("%-40s %6.2f %%" % (name, 100.0 * c / cyc.sum()))
