"""Throughput of the dense-BFGS kernel on the config-2 shape (65,536 x Rosenbrock-32, parity stopping), next to
L-BFGS m = 6 on the same batch.  Not a BASELINE configuration: a data point for DESIGN.md section 3.9."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

B, n = 65536, 32
x0 = torch.from_numpy(amd.synthetic_x0_host(B, n)).cuda()
for name, s in (("Bfgs", amd.BatchedBfgs(stopping_progress=amd.parity_stop())),
                ("Lbfgs m=6", amd.BatchedLbfgs(m=6, stopping_progress=amd.parity_stop()))):
    for _ in range(3):
        x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
    torch.cuda.synchronize()
    ms = s.last_kernel_ms()
    pn = amd.progress_to_numpy(p)
    ll = s.last_launch()
    print("%-10s kernel %.2f ms -> %.2f M solves/s; iterations mean %.1f max %d, evaluations mean %.1f; "
          "%d x %d lanes x elems, %d workgroups, %d B LDS" % (
              name, ms, B / ms / 1e3, pn["num_iterations"].mean(), pn["num_iterations"].max(), pn["nfev"].mean(),
              ll["lanes_per_problem"], ll["elems_per_lane"], ll["blocks"], ll["lds_bytes"]))
