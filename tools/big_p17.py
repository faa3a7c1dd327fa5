import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
for deg, level, count in ((17, 3, None), (17, 4, 2048), (19, 3, None), (19, 4, 2048), (16, 4, 2048), (18, 4, 2048)):
    m = M.BrickMesh(level, deg, count=count)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry_brick(np.ones(m.n_elements, dtype=np.int32), float(1 << level), [0.0, 1.0, 0.0, 1.0, 0.0, 1.0])
    plan.set_tuning(7, 0)
    du = torch.rand(m.local_nodes, dtype=torch.float64, device=dev); out = torch.empty_like(du)
    for _ in range(10): plan.apply_stiffness_matrix(du, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps): plan.apply_stiffness_matrix(du, out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print("p=%2d elements %6d: %8.1f us  %6.1f GDoF/s = %.2f of the roof  stream=%d  %s" % (deg, m.n_elements, t * 1e6, m.local_nodes / t / 1e9, m.local_nodes / t / 1e9 / 125.0, plan.stream_mode(), plan.last_kernel()), flush=True)
    plan.destroy(); del du, out
