"""Timing of the full operator on a locally refined (hanging-face) brick.  Usage: time_hp.py [base_level] [deg] [refine_every]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
every = int(sys.argv[3]) if len(sys.argv) > 3 else 3
refine = np.zeros(8 ** level, dtype=bool)
refine[::every] = True
m = M.HangingBrickMesh(level, refine, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0)
if os.environ.get('D4EST_SPLIT'): plan.set_tuning(13, int(os.environ['D4EST_SPLIT']))
plan.set_faces(sides)
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
tr = torch.empty(plan.trace_size, dtype=torch.float64, device=dev)
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
hang = int((sides["side_hang"] == 1).sum())
he = int((np.asarray(sides["side_hang"]).reshape(-1, 6) != 0).any(axis=1).sum())
print("hanging mesh (split %s): base level %d p %d: %d elements (%d with a hanging side), %d DoF, %d hanging faces: stiffness %.1f us | traces %.1f us | flux %.1f us | apply_aij %.1f us" % (
    os.environ.get('D4EST_SPLIT', 'auto'), level, deg, m.n_elements, he, m.local_nodes, hang, t(lambda: plan.apply_stiffness_matrix(du, Au)),
    t(lambda: plan.compute_face_traces(du, tr)), t(lambda: plan.apply_flux(tr, None, Au)), t(lambda: plan.apply_aij(du, Au))))
