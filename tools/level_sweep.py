"""p = 7 stiffness apply (general path: per-node metric streamed) at levels 3 ... 6 -- SURVEY.md section 8d asks for the level sweep that
leaves the launch-latency floor.  The geometric factors are generated on the device (plan_set_geometry_brick), so level 6 (134 MDoF,
6.4 GB of metric) needs no host arrays."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
ROOT = 1 << 30
for level in (3, 4, 5, 6):
    m = M.BrickMesh(level, 7)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry_brick(np.full(m.n_elements, ROOT >> level, dtype=np.int32), float(ROOT), [0.0, 1.0, 0.0, 1.0, 0.0, 1.0])
    u = torch.rand(m.local_nodes, dtype=torch.float64, device=dev); Au = torch.empty_like(u)
    out = []
    for affine in (0, -1):
        plan.set_tuning(7, affine)
        for _ in range(3): plan.apply_stiffness_matrix(u, Au)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50 if level < 6 else 10
        e0.record()
        for _ in range(reps): plan.apply_stiffness_matrix(u, Au)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        out.append((us, m.local_nodes / us / 1e3, plan.last_kernel()))
    print("level %d p 7: %9d elements %6.1f MDoF | general path %8.1f us %6.1f GDoF/s %5.2f of 8 TB/s (%s) | affine path %8.1f us %6.1f GDoF/s (%s)" % (
        level, m.n_elements, m.local_nodes / 1e6, out[0][0], out[0][1], out[0][1] * 64 / 8000, out[0][2], out[1][0], out[1][1], out[1][2]), flush=True)
    plan.destroy(); del u, Au
