"""Stiffness apply rate for several p at roughly constant DoF count (secondary numbers for DESIGN.md)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
for deg, level, count in ((1, 6, None), (2, 5, None), (3, 5, None), (5, 5, 16384), (7, 5, 16384), (9, 4, None), (11, 4, None), (13, 4, 2048), (15, 4, 2048), (17, 3, None), (19, 3, None)):
    m = M.BrickMesh(level, deg, count=count)
    J, rst = m.geometry(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, 0)  # general path unless overridden
    du = torch.from_numpy(u).to(dev); out = torch.empty_like(du)
    if len(sys.argv) > 1:  # "key=value,key=value" tuning overrides
        for kv in sys.argv[1].split(","):
            k_, v_ = kv.split("="); plan.set_tuning(int(k_), int(v_))
    for _ in range(5): plan.apply_stiffness_matrix(du, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps): plan.apply_stiffness_matrix(du, out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    n = deg + 1
    print("p=%2d elements %6d DoF %9d: %8.1f us  %6.1f GDoF/s  %6.0f GB/s alg  %5.1f TFLOP/s alg  %s" % (
        deg, m.n_elements, m.local_nodes, t * 1e6, m.local_nodes / t / 1e9, 64 * m.local_nodes / t / 1e9,
        (32 * n + 15) * m.local_nodes / t / 1e12, plan.last_kernel()))
    plan.destroy()
