"""Latency of ONE wavefront: solves B = 1, 2, 4, 8 problems (one wavefront of the 8-lanes-per-problem
mapping) and prints kernel time per pass.  Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU
SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES` to get instructions per pass."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

n, m = 32, 6
for B in (1, 2, 4, 8, 16):
    x0 = torch.from_numpy(amd.synthetic_x0_host(B, n, first_problem=37097 if B == 1 else 0)).cuda()
    s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop())
    for _ in range(3):
        x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    ms = s.last_kernel_ms()
    pn = amd.progress_to_numpy(p)
    it, nf = pn["num_iterations"].astype(np.int64), pn["nfev"].astype(np.int64)
    print("B %2d  kernel %.3f ms  iterations max %d sum %d  nfev max %d  -> %.3f us per pass of the longest problem"
          % (B, ms, it.max(), it.sum(), nf.max(), 1e3 * ms / it.max()))
