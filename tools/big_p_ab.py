"""General-path stiffness rate of the multi-wave volume kernels (p >= 8) at the sizes the bench quotes; with D4EST_HIP_LIBRARY=<variant>
from tools/build_variant.sh this is the A/B harness of the metric-pipelining depths: tools/big_p_ab.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
modes = [0]
for deg, level, count in ((11, 4, None), (11, 5, None), (9, 4, None), (13, 4, 2048), (8, 4, None), (12, 4, 2048), (17, 3, None)):
    m = M.BrickMesh(level, deg, count=count)
    J, rst = m.geometry(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, 0)
    du = torch.from_numpy(u).to(dev); out = torch.empty_like(du); ref = None
    for mode in modes:
        for _ in range(5): plan.apply_stiffness_matrix(du, out)
        torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        ok = torch.equal(ref, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 30
        e0.record()
        for _ in range(reps): plan.apply_stiffness_matrix(du, out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        print("p=%2d elements %6d mode %d: %8.1f us  %6.1f GDoF/s  same=%s  %s" % (deg, m.n_elements, mode, t * 1e6, m.local_nodes / t / 1e9, ok, plan.last_kernel()), flush=True)
    plan.destroy(); del du, out
