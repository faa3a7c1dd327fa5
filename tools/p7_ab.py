"""Config-2 headline kernel timing (general path), p = 7 / 5 / 3 at level 4, and p = 7 at level 5 on the wave kernel: tools/p7_ab.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
for deg, level, tw in ((7, 4, -1), (7, 4, -1), (5, 4, -1), (3, 4, -1), (6, 4, -1), (7, 5, 11)):
    m = M.BrickMesh(level, deg)
    J, rst = m.geometry(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, 0)
    if tw >= 0: plan.set_tuning(0, tw)
    du = torch.from_numpy(u).to(dev); out = torch.empty_like(du)
    for _ in range(20): plan.apply_stiffness_matrix(du, out)
    torch.cuda.synchronize()
    best = 1e9; ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100): plan.apply_stiffness_matrix(du, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 100 * 1e-3)
    t = float(np.median(ts))
    print("p=%d level %d: %7.2f us (min %.2f)  %6.1f GDoF/s  frac %.3f  sum=%.12e  %s" % (deg, level, t * 1e6, min(ts) * 1e6, m.local_nodes / t / 1e9, 64 * m.local_nodes / t / 8e12, float(out.double().abs().sum()), plan.last_kernel()), flush=True)
    plan.destroy(); del du, out
