"""Latency of the longest config-2 solve (problem 37097: 1626 iterations) alone on the chip under each mapping:
how much faster would the tail of the headline batch be if a straggler ran on a wider mapping?"""
import sys
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cppnumericalsolvers_amd as amd

n, m = 32, 6
x0 = torch.from_numpy(amd.synthetic_x0_host(1, n, first_problem=37097)).cuda()
ref = None
for (W, E) in ((8, 4), (16, 2), (32, 1), (16, 4), (32, 2), (64, 1)):
    for placement in (0, 1):
        s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop(), lanes_per_problem=W, elems_per_lane=E,
                             history_placement=placement)
        for _ in range(3):
            x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
        torch.cuda.synchronize()
        ms = s.last_kernel_ms()
        pn = amd.progress_to_numpy(p)
        it = int(pn["num_iterations"][0])
        xs = x.cpu().numpy()
        if ref is None:
            ref = xs
        print("W %2d E %d placement %d (mr %s): kernel %.3f ms, %d iterations -> %.3f us per iteration; same bits as (8,4): %s"
              % (W, E, placement, s.last_launch(), ms, it, 1e3 * ms / it, np.array_equal(xs, ref)))
