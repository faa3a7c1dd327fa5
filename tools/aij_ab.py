"""apply_aij timing (general volume metric), several (level, p): tools/aij_ab.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
for level, deg in ((4, 7), (4, 7), (5, 7), (4, 5), (4, 6), (4, 3)):
    m = M.BrickMesh(level, deg)
    J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, 0); plan.set_faces(sides)
    du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
    for _ in range(20): plan.apply_aij(du, Au)
    torch.cuda.synchronize()
    ts = []
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): plan.apply_aij(du, Au)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e-3)
    t = float(np.median(ts))
    print("level %d p=%d: apply_aij %7.2f us (min %.2f)  %6.1f GDoF/s  sum=%.12e  %s" % (level, deg, t * 1e6, min(ts) * 1e6, m.local_nodes / t / 1e9, float(Au.abs().sum()), plan.last_kernel()), flush=True)
    plan.destroy(); del du, Au
