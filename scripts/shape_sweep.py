"""Throughput of the L-BFGS kernel across problem shapes (Rosenbrock-n, parity stopping), one line per
(n, m): mapping chosen by the library, kernel time, solves/s and problem-iterations/ms."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

for n, m, B in ((8, 6, 262144), (16, 6, 262144), (32, 6, 262144), (64, 10, 131072), (100, 10, 65536), (128, 10, 65536),
                (256, 10, 32768), (256, 5, 32768), (64, 17, 65536)):
    x0 = torch.from_numpy(amd.synthetic_x0_host(B, n)).cuda()
    s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop())
    for _ in range(2):
        x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    ms = s.last_kernel_ms()
    pn = amd.progress_to_numpy(p)
    ll = s.last_launch()
    print("n %3d m %2d B %6d: %3d lanes x %d elems, y in registers: %2d, %4d workgroups, %6d B LDS; %8.2f ms, "
          "%6.2f M solves/s, %5.2f M problem-iterations/ms (mean %5.1f iterations, max %d), converged %d/%d" % (
              n, m, B, ll["lanes_per_problem"], ll["elems_per_lane"], ll["y_columns_in_registers"], ll["blocks"],
              ll["lds_bytes"], ms, B / ms / 1e3, pn["num_iterations"].sum() / ms / 1e6, pn["num_iterations"].mean(),
              pn["num_iterations"].max(), int((pn["status"] >= 2).sum()), B))
