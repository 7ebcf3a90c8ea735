"""PCIe-inclusive rate of the host-pointer entry point (mi355_lbfgs_minimize_batch_host) on the configs[1] batch:
repeated calls with reused and with freshly allocated output arrays, for several piece counts of the DMA overlap.
  python scripts/host_path_probe.py"""
import os
import sys
import time
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cppnumericalsolvers_amd as amd
from cppnumericalsolvers_amd import capi

B, n, m = 65536, 32, 6
x0 = amd.synthetic_x0_host(B, n, "std")
s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop())
xd = torch.from_numpy(x0).cuda()
s.minimize(amd.Rosenbrock(), xd)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    s.minimize(amd.Rosenbrock(), xd)
torch.cuda.synchronize()
print("device-resident: %.2f ms per batch (kernel %.2f ms)" % ((time.perf_counter() - t0) / 5 * 1e3, s.last_kernel_ms()))
x, g, f = np.empty_like(x0), np.empty_like(x0), np.empty(B)
prog = np.zeros(B, dtype=capi.PROGRESS_DTYPE)
d = s._desc(amd.Rosenbrock(), n)
for pieces in ("1", "2", "4", "8", ""):
    if pieces:
        os.environ["MI355_HOST_PIECES"] = pieces
    else:
        os.environ.pop("MI355_HOST_PIECES", None)
    ts = []
    for rep in range(7):
        t0 = time.perf_counter()
        capi.check(s.ctx._lib.mi355_lbfgs_minimize_batch_host(s.ctx.handle, C.byref(d), B, x0.ctypes.data, x.ctypes.data,
                                                              f.ctypes.data, g.ctypes.data, prog.ctypes.data))
        ts.append((time.perf_counter() - t0) * 1e3)
    tf = []
    for rep in range(4):
        t0 = time.perf_counter()
        s.minimize_host(amd.Rosenbrock(), x0)     # fresh numpy outputs: first-touch page faults included
        tf.append((time.perf_counter() - t0) * 1e3)
    print("pieces %-4s reused outputs: median %.2f ms (min %.2f) = %.2f M solves/s;  fresh outputs: median %.2f ms" % (
        pieces or "auto", np.median(ts[1:]), min(ts), B / np.median(ts[1:]) / 1e3, np.median(tf)))
