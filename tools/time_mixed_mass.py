"""Mass / weighted mass / stiffness on a mixed-degree plan (p = 3 ... 9 over the level-4 brick): one launch for the deg_quad = deg <= 7 buckets
(tuning key 6 = 0 switches the even-odd kernels -- and with them the one-launch form -- off)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
degs = 3 + (np.arange(8 ** 4) * 5) % 7
m = M.BrickMesh(4, degs)
J, rst = m.geometry(None)
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0)
x = torch.from_numpy(m.field()).to(dev); y = torch.empty_like(x)
c = torch.rand(m.local_nodes_quad, dtype=torch.float64, device=dev)
def t(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for eo in (-1, 0):
    plan.set_tuning(6, eo)
    print("tuning EO=%d: mass %.1f us | weighted mass %.1f us | stiffness %.1f us" % (eo, t(lambda: plan.apply_mass_matrix(x, y)), t(lambda: plan.apply_weighted_mass_matrix(x, c, y)), t(lambda: plan.apply_stiffness_matrix(x, y))))
