"""Full operator on a mixed-p mesh (p = 3 ... 9, the hp-adaptive case of BASELINE config 4): per-kernel timing."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (3, 9)
n = 8 ** level
deg = lo + (np.arange(n) * 5 % (hi - lo + 1))
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0); plan.set_faces(sides)
if os.environ.get('D4EST_FORK'): plan.set_tuning(5, int(os.environ['D4EST_FORK']))
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
tr = torch.empty(plan.trace_size, dtype=torch.float64, device=dev)
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("level %d p %d..%d (%d elements, %.2f MDoF): stiffness %.1f us | traces %.1f us | flux %.1f us | apply_aij %.1f us" % (
    level, lo, hi, n, m.local_nodes / 1e6, t(lambda: plan.apply_stiffness_matrix(du, Au)), t(lambda: plan.compute_face_traces(du, tr)),
    t(lambda: plan.apply_flux(tr, None, Au)), t(lambda: plan.apply_aij(du, Au))))
