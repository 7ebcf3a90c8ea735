"""How much of a batch's wall time is its longest solves?  Runs the cfg2 workload with the iteration
cap of the stopping criteria lowered step by step (the cap cuts the stragglers, nothing else) and
prints kernel time next to the iteration-count distribution."""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

B, n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 65536, 32, 6
x0 = torch.from_numpy(amd.synthetic_x0_host(B, n)).cuda()
for cap in (10000, 1000, 600, 400, 300, 260, 230):
    st = amd.parity_stop()
    st.num_iterations = cap
    s = amd.BatchedLbfgs(m=m, stopping_progress=st)
    for _ in range(3):
        x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    ms = s.last_kernel_ms()
    it = amd.progress_to_numpy(p)["num_iterations"].astype(np.int64)
    print("cap %5d  kernel %.3f ms  iterations: mean %.1f p50 %d p99 %d max %d  sum %.3e  -> %.2f M problem-iterations/ms"
          % (cap, ms, it.mean(), np.percentile(it, 50), np.percentile(it, 99), it.max(), it.sum(), it.sum() / ms / 1e6))
