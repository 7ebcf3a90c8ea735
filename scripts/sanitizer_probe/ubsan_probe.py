import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import cppnumericalsolvers_amd as amd
n, m, B = 8, 6, 3
x0 = amd.synthetic_x0_host(B, n, "std")
s = amd.BatchedLbfgs(m=m, stopping_progress=amd.parity_stop(), device=0)
x0d = torch.from_numpy(x0).to("cuda:0")
x, f, g, p = s.minimize(amd.Rosenbrock(), x0d)
torch.cuda.synchronize()
print("x0", x0[0])
print("x ", x.cpu().numpy()[0])
print("g ", g.cpu().numpy()[0])
print("f ", f.cpu().numpy())
print("p ", amd.progress_to_numpy(p))
print("x0 after", x0d.cpu().numpy()[0])
ll = s.last_launch(); print(ll)
