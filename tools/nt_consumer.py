"""Stiffness apply alone and followed by a reader of A u (dot product, as in CG): does a non-temporal store of A u cost the consumer
what it saves the producer?  usage: tools/nt_consumer.py [level deg]; variants via D4EST_HIP_LIBRARY."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 7
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0)
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
def t(fn, reps=50):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
def both():
    plan.apply_stiffness_matrix(du, Au); torch.dot(du, Au)
a = [t(lambda: plan.apply_stiffness_matrix(du, Au)) for _ in range(3)]
b = [t(both) for _ in range(3)]
c = t(lambda: torch.dot(du, Au))
print("%s level %d p %d: stiffness %s us | stiffness + dot(u, Au) %s us | dot alone %.1f us  [%s]" % (
    os.path.basename(os.environ.get("D4EST_HIP_LIBRARY", "default")), level, deg, " ".join("%.2f" % x for x in a), " ".join("%.2f" % x for x in b), c, plan.last_kernel()))
