"""Cycle breakdown of the L-BFGS iteration from a profiling build of the library:

  python -c "from cppnumericalsolvers_amd import _build as b; b.build(extra_flags=['-DMI355_LBFGS_PHASE_TIMING'], output=b.PKG_DIR + '/variants/lib_lphases.so')"
  MI355_LBFGS_LIBRARY=$PWD/cppnumericalsolvers_amd/variants/lib_lphases.so python scripts/lbfgs_phases.py [B]

Every wavefront sums s_memtime deltas per phase; shares are of the wavefront-resident time.  B = 1 shows
the latency profile of a lone problem, the default B the steady state of the headline batch."""
import ctypes as C
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

PHASES = ["loop top / exit", "two-loop recursion", "descent test, initial step", "line search",
          "s, y, curvature test, push, scaling", "Progress::Update", "results / refill", "work-queue atomic",
          "kernel prologue", "start point from HBM", "first evaluation + solver reset"]
if len(sys.argv) > 1 and sys.argv[1] == "ridge":   # the matrix-core ridge kernel (config 4 shape)
    B, rows, n, m = 65536, 128, 64, 10
    A, Y = amd.synthetic_ridge_host(B, rows, n)
    s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop())
    x, f, g, p = s.minimize(amd.SquaredErrorRidge(A, 0.1, matrix_cores=True),
                            torch.zeros(B, n, dtype=torch.float64, device="cuda"), per_problem=torch.from_numpy(Y).cuda())
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib = s.ctx._lib
    lib.mi355_lbfgsb_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    amd.capi.check(lib.mi355_lbfgsb_phase_cycles(s.ctx.handle, out))
    cyc = np.array(list(out)[:8], dtype=np.float64)
    print("ridge on the matrix cores, B = %d: kernel %.3f ms" % (B, s.last_kernel_ms()))
    for name, c in zip(["refill from the work queue", "r = A X - Y (8 wavefronts)", "barrier B", "G = A^T R (8 wavefronts) + ||r||^2, ||x||^2",
                        "barrier C", "f, g pick-up + line-search logic", "end of iteration + two-loop + search set-up", "publish + barrier A"], cyc):
        print("   %-48s %6.2f %%" % (name, 100.0 * c / cyc.sum()))
    sys.exit(0)
n, m = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (32, 6)   # usage: lbfgs_phases.py [B [n m]]
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [1, 8, 65536]):
    x0 = torch.from_numpy(amd.synthetic_x0_host(B, n, first_problem=37097 if B == 1 else 0)).cuda()
    # MI355_PHASES_LINESEARCH=hager_zhang: the same table for Lbfgs<F, m, HagerZhang> (round 6)
    s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop(),
                         linesearch=os.environ.get("MI355_PHASES_LINESEARCH", "more_thuente"))
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 16)()
    lib = s.ctx._lib
    lib.mi355_lbfgsb_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
    amd.capi.check(lib.mi355_lbfgsb_phase_cycles(s.ctx.handle, out))
    cyc = np.array(list(out)[:11], dtype=np.float64)
    it = amd.progress_to_numpy(p)["num_iterations"]
    print("n = %d m = %d B = %d: kernel %.3f ms, mean iterations %.1f, max %d, wave-cycles per problem-iteration %.0f" % (
        n, m, B, s.last_kernel_ms(), it.mean(), it.max(), cyc.sum() / it.sum()))
    for name, c in zip(PHASES, cyc):
        print("   %-40s %6.2f %%" % (name, 100.0 * c / cyc.sum()))
