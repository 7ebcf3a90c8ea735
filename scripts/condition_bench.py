#!/usr/bin/env python3
"""Cost of the condition_hessian stopping test of a Second-mode function with a non-constant Hessian (csrc/hessian_condition_device.hpp):
the same batch of chained-Rosenbrock problems solved with the test off and on (threshold never reached), kernel time from the
library's HIP events.  One JSON line per shape."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import cppnumericalsolvers_amd as amd
    ctx = amd.Context(0)
    obj = amd.Rosenbrock(differentiability="second")
    for n, m, B in ((8, 10, 65536), (16, 10, 65536), (32, 10, 32768), (64, 10, 16384)):
        x0 = torch.from_numpy(amd.synthetic_x0_host(B, n, "std")).to("cuda:0")
        row = {"n": n, "m": m, "B": B}
        for name, threshold in (("off", 0.0), ("on", 1e300)):
            s = amd.BatchedLbfgs(m=m, context=ctx, condition_hessian=threshold, arithmetic="default")
            ms = []
            for _ in range(4):
                x, f, g, p = s.minimize(obj, x0)
                torch.cuda.synchronize()
                ms.append(s.last_kernel_ms())
            prog = amd.progress_to_numpy(p)
            ll = s.last_launch()
            it = float(prog["num_iterations"].sum())
            row[name] = {"kernel_ms": round(float(np.median(ms[1:])), 3), "solves_per_s": round(B / (np.median(ms[1:]) * 1e-3)),
                         "iterations_mean": round(it / B, 1), "lanes": ll["lanes_per_problem"], "elems": ll["elems_per_lane"],
                         "lds_bytes": ll["lds_bytes"], "workgroups": ll["blocks"]}
        extra_ms = row["on"]["kernel_ms"] - row["off"]["kernel_ms"]
        its = row["on"]["iterations_mean"] * B
        flops = its * (2.0 * n ** 3 / 3 + 2.0 * n ** 3)      # LU + n column solves of 2 n^2 each
        row["condition_test"] = {"extra_ms": round(extra_ms, 3), 
                                 "model_gflops": round(flops / 1e9, 2), "achieved_tflops_in_extra_time": round(flops / (extra_ms * 1e-3) / 1e12, 3) if extra_ms > 0 else None}
        print(json.dumps(row), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
