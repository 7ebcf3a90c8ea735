#!/usr/bin/env python3
"""A/B of the dense-BFGS kernel's wavefront mapping (round 6).  H is (W*E)^2 doubles per problem in LDS, so LDS capacity —
not registers — caps the problems in flight per CU; the packed mappings of the Lbfgs kernels (8 lanes x 4 coordinates at
n = 32: eight problems = 70 KB per wavefront) leave half a wavefront per SIMD.  Spreading a problem over as many lanes as
it has columns keeps the LDS per problem and multiplies the wavefronts per CU.  The exact arithmetic is a butterfly over
the padded width whatever the split, so every mapping returns the same bits (checked).

    python scripts/bfgs_mapping_ab.py > gpurun_out/r6_ab_bfgs_mapping.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import cppnumericalsolvers_amd as amd  # noqa: E402

ctx = amd.Context(0)
print("dense BFGS (Bfgs<F, MoreThuente>), Rosenbrock-n, parity stopping, exact arithmetic; %s" % torch.cuda.get_device_name(0))
for n, B, mappings in ((32, 65536, ((8, 4), (16, 2), (32, 1))), (64, 32768, ((16, 4), (32, 2), (64, 1))),
                       (16, 65536, ((8, 2),)), (8, 65536, ((8, 1),))):
    x0 = torch.from_numpy(amd.synthetic_x0_host(B, n)).cuda()
    ref = None
    for W, E in mappings:
        for ls in (("more_thuente", "hager_zhang") if n >= 32 else ("more_thuente",)):
            s = amd.BatchedBfgs(stopping_progress=amd.parity_stop(), context=ctx, linesearch=ls, lanes_per_problem=W,
                                elems_per_lane=E)
            ms = []
            for _ in range(4):
                x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
                torch.cuda.synchronize()
                ms.append(s.last_kernel_ms())
            ms = float(np.median(ms[1:]))
            pn = amd.progress_to_numpy(p)
            ll = s.last_launch()
            same = ""
            if ref is None:
                ref = {}
            if ls not in ref:
                ref[ls] = (x.clone(), f.clone())
            else:
                same = "; bits == first mapping: %s" % bool(torch.equal(x, ref[ls][0]) and torch.equal(f, ref[ls][1]))
            print("n %3d B %6d  %-12s %2d lanes x %d: kernel %8.2f ms -> %6.3f M solves/s; %4d workgroups x %3d threads, %6d B LDS; "
                  "iterations mean %.1f max %d, evaluations mean %.1f%s" % (
                      n, B, ls, W, E, ms, B / ms / 1e3, ll["blocks"], ll["threads"], ll["lds_bytes"], pn["num_iterations"].mean(),
                      pn["num_iterations"].max(), pn["nfev"].mean(), same))
