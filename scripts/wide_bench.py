"""Throughput of the workgroup kernel for n > 256 (csrc/lbfgs_wide_kernel.hpp) at fixed work: Rosenbrock-n, m = 10, exactly
100 iterations per problem (every stopping test but the iteration limit switched off).  Prints problem-iterations/s and
the state-streaming bytes of SURVEY section 8d, 8n(6T + 2 sum_k), over the kernel time -- in THIS regime the vectors and
the correction ring really live in HBM / L2, so the figure is a bandwidth.
  python scripts/wide_bench.py [n:B ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import cppnumericalsolvers_amd as amd  # noqa: E402


def main():
    cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(512, 4096), (1024, 4096), (4096, 2048),
                                                                           (16384, 1024), (65536, 512), (1048576, 32)]
    m, T = 10, 100
    stop = amd.capi.default_stop()
    stop.num_iterations, stop.x_delta, stop.f_delta, stop.gradient_norm, stop.past = T, 0.0, 0.0, 0.0, 0
    s = amd.BatchedLbfgs(m=m, stopping_progress=stop)
    print("%9s %6s %10s %12s %14s %12s %8s" % ("n", "B", "kernel ms", "prob-it/s", "model GB/s", "frac 8TB/s", "blocks"))
    for n, B in cases:
        rng = np.random.default_rng(n)
        x0 = torch.from_numpy(np.tile([-1.2, 1.0], n)[:n] + 0.1 * rng.uniform(-1, 1, (B, n))).to("cuda:0")
        best = None
        for rep in range(3):
            x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
            torch.cuda.synchronize()
            ms = s.last_kernel_ms()
            best = ms if best is None else min(best, ms)
        pg = amd.progress_to_numpy(p)
        its, sum_k = pg["num_iterations"].astype(np.float64), pg["sum_k"].astype(np.float64)
        model_bytes = float(np.sum(8.0 * n * (6.0 * its + 2.0 * sum_k)))
        print("%9d %6d %10.3f %12.4g %14.1f %12.3f %8d" % (n, B, best, its.sum() / (best * 1e-3), model_bytes / (best * 1e-3) / 1e9,
                                                          model_bytes / (best * 1e-3) / 8e12, s.last_launch()["blocks"]))


if __name__ == "__main__":
    main()
