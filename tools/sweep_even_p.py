"""Stiffness apply at the degrees with an ODD number of nodes per direction (p = 2, 4, ..., 14), level 4 or a 2048-element slice:
the even-odd contractions now take odd sizes too (argv[1] = "6=0" switches them off for the multi-wave kernels; "1=3" selects the plain
single-wave kernel at p <= 6)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
for deg, level, count in ((2, 4, None), (4, 4, None), (6, 4, None), (8, 4, None), (10, 4, None), (12, 4, 2048), (14, 4, 2048)):
    m = M.BrickMesh(level, deg, count=count)
    J, rst = m.geometry(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, 0)
    du = torch.from_numpy(u).to(dev); out = torch.empty_like(du)
    if len(sys.argv) > 1:
        for kv in sys.argv[1].split(","):
            k_, v_ = kv.split("="); plan.set_tuning(int(k_), int(v_))
    for _ in range(10): plan.apply_stiffness_matrix(du, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps): plan.apply_stiffness_matrix(du, out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print("p=%2d elements %6d DoF %9d: %8.1f us  %6.1f GDoF/s  %s" % (deg, m.n_elements, m.local_nodes, t * 1e6, m.local_nodes / t / 1e9, plan.last_kernel()))
    plan.destroy()
