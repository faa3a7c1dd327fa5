"""Per-kernel timing of the full operator (stiffness / traces / flux) with HIP events, one process."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 7
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0); plan.set_faces(sides)  # general (streamed-metric) path
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
tr = torch.empty(plan.trace_size, dtype=torch.float64, device=dev)
d1 = torch.empty_like(du); d2 = torch.empty_like(du)   # (allocated once: a clone inside the timed call would be timed with it)
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("level %d p %d: stiffness %.1f us | traces %.1f us | flux %.1f us | apply_aij %.1f us | mass %.1f us | dudr %.1f us" % (
    level, deg, t(lambda: plan.apply_stiffness_matrix(du, Au)), t(lambda: plan.compute_face_traces(du, tr)),
    t(lambda: plan.apply_flux(tr, None, Au)), t(lambda: plan.apply_aij(du, Au)), t(lambda: plan.apply_mass_matrix(du, Au)),
    t(lambda: plan.compute_dudr(du, Au, d1, d2))))
