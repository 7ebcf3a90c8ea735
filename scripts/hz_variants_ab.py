#!/usr/bin/env python3
"""A/B of the Hager-Zhang Lbfgs kernel variants on configs[1]'s shape (65,536 x Rosenbrock-32, m = 6, parity stopping):
exact arithmetic + LDS ring (rounds 2-5) against the fused arithmetic with the LDS ring / the y history in registers on
the mappings that cover 32 coordinates.   python scripts/hz_variants_ab.py > gpurun_out/r6_ab_hz.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cppnumericalsolvers_amd as amd  # noqa: E402

ctx = amd.Context(0)
print("Lbfgs<F, m, HagerZhang>, Rosenbrock-n, parity stopping; %s" % torch.cuda.get_device_name(0))
for n, m, B in ((32, 6, 65536), (64, 10, 131072)):
    x0 = torch.from_numpy(amd.synthetic_x0_host(B, n)).cuda()
    P = 8
    while P < n:
        P *= 2
    variants = [("exact, LDS ring (rounds 2-5)", dict(arithmetic="exact"))]
    for W, E in ((P // 4, 4), (P // 2, 2)):
        for placement, pname in ((1, "LDS ring"), (2, "y in registers")):
            variants.append(("fused, %2d lanes x %d, %s" % (W, E, pname),
                             dict(arithmetic="fma", lanes_per_problem=W, elems_per_lane=E, history_placement=placement)))
    variants.append(("fused, library default", dict(arithmetic="default")))
    for name, kw in variants:
        s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop(), context=ctx, linesearch="hager_zhang", **kw)
        ms = []
        for _ in range(4):
            x, f, g, p = s.minimize(amd.Rosenbrock(), x0, want_gradient=False)
            torch.cuda.synchronize()
            ms.append(s.last_kernel_ms())
        ms = float(np.median(ms[1:]))
        pn = amd.progress_to_numpy(p)
        ll = s.last_launch()
        print("n %3d m %2d B %7d  %-40s kernel %8.2f ms -> %6.3f M solves/s; %2d x %d, %4d workgroups, %6d B LDS, y columns in "
              "registers %2d; iterations mean %.1f max %d, evaluations mean %.1f" % (
                  n, m, B, name, ms, B / ms / 1e3, ll["lanes_per_problem"], ll["elems_per_lane"], ll["blocks"], ll["lds_bytes"],
                  ll["y_columns_in_registers"], pn["num_iterations"].mean(), pn["num_iterations"].max(), pn["nfev"].mean()))
