"""apply_lhs with the zeroth-order term (plan_set_lhs_coefficient) against apply_aij: tools/time_lhs.py <level> <deg>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level, deg = int(sys.argv[1]), int(sys.argv[2])
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0); plan.set_faces(sides)
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
c = 1.0 + torch.rand(m.local_nodes_quad, dtype=torch.float64, device=dev)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
t_aij = t(lambda: plan.apply_aij(du, Au))
plan.set_lhs_coefficient(c)
t_lhs = t(lambda: plan.apply_lhs(du, Au))
print("level %d p %d [%s]: apply_aij %.1f us | apply_lhs with the zeroth-order term %.1f us (%+.1f %%)" % (level, deg, plan.face_path(), t_aij, t_lhs, 100 * (t_lhs / t_aij - 1)))
