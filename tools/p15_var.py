"""Run-to-run spread of the p = 15 matrix-core stiffness kernel: every apply timed on its own. argv: n_elements [level]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
n_el = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
m = M.BrickMesh(5, 15, count=n_el)
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry_brick(np.ones(m.n_elements, dtype=np.int32), 32.0, [0, 1, 0, 1, 0, 1.0])
plan.set_tuning(7, 0)
x = torch.rand(m.local_nodes, dtype=torch.float64, device=dev); y = torch.empty_like(x)
for _ in range(5): plan.apply_stiffness_matrix(x, y)
torch.cuda.synchronize()
ts = []
for _ in range(40):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); plan.apply_stiffness_matrix(x, y); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
ts = np.array(ts)
print("%s: %d elements: min %.1f median %.1f max %.1f us -> %.1f / %.1f / %.1f GDoF/s" % (
    plan.last_kernel(), n_el, ts.min(), np.median(ts), ts.max(), m.local_nodes / ts.min() / 1e3, m.local_nodes / np.median(ts) / 1e3, m.local_nodes / ts.max() / 1e3))
print(" ".join("%.0f" % t for t in ts))
