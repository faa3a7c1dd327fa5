"""apply_aij of one face path, timed alone: tools/time_whole.py <level> <deg> [tuning key 11 = 2] [reps]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level, deg = int(sys.argv[1]), int(sys.argv[2])
key11 = int(sys.argv[3]) if len(sys.argv) > 3 else 2
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0); plan.set_tuning(11, key11)
if os.environ.get('D4EST_STREAM'): plan.set_tuning(12, int(os.environ['D4EST_STREAM']))   # stream mode forced off / on
plan.set_faces(sides)
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
for _ in range(5): plan.apply_aij(du, Au)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): plan.apply_aij(du, Au)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e3
print("%s level %d p %d key11=%d: apply_aij %.1f us = %.1f GDoF/s  [%s]" % (os.environ.get("D4EST_HIP_LIBRARY", "default")[-24:], level, deg, key11, t, m.local_nodes / t / 1e3, plan.face_path()), flush=True)
