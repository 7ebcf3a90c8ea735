"""Iteration counts of the config-2 batch (production arithmetic) -> gpurun_out/r2/iters_cfg2.npy, for the
off-line scheduling model in scripts/tail_model.py."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cppnumericalsolvers_amd as amd
import bench

wl = bench.WORKLOADS["cfg2"]
B, n, m = wl["B"], wl["n"], wl["m"]
s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop())
x0 = s.fill_x0(B, n, wl.get("x0", "std"), bench.SEED)
for _ in range(3):
    x, f, g, p = s.minimize(amd.Rosenbrock(), x0)
torch.cuda.synchronize()
pr = amd.progress_to_numpy(p)
it = pr["num_iterations"].astype(np.int32)
os.makedirs("gpurun_out/r2", exist_ok=True)
np.save("gpurun_out/r2/iters_cfg2.npy", it)
print("kernel %.3f ms arithmetic %s; iterations mean %.1f max %d at index %d" % (s.last_kernel_ms(), s.last_arithmetic(), it.mean(), it.max(), it.argmax()))
for q in (50, 90, 99, 99.9, 99.99):
    print("p%s = %d" % (q, np.percentile(it, q)))
