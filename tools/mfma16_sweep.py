"""p = 12 ... 15: matrix-core kernel (tuning 4 = 2) vs vector-ALU kernel (1): parity on a curved brick and time on ~8 MDoF."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
from tests import oracle_lib
dev = torch.device("cuda:0"); oracle = oracle_lib.load()
for deg in (12, 13, 14, 15):
    m = M.BrickMesh(1, deg, count=5); mp = M.SineMap(0.05)
    J, rst = m.geometry(mp); u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=8)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0); plan.set_geometry(J, rst)
    du = torch.from_numpy(u).to(dev)
    errs = []
    for bigp in (1, 2):
        plan.set_tuning(4, bigp); out = torch.full_like(du, float("nan")); plan.apply_stiffness_matrix(du, out)
        errs.append(np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max())
    plan.destroy()
    n_el = int(8.4e6 / (deg + 1) ** 3)
    m = M.BrickMesh(5, deg, count=n_el)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry_brick(np.ones(m.n_elements, dtype=np.int32), 32.0, [0, 1, 0, 1, 0, 1.0]); plan.set_tuning(7, 0)
    x = torch.rand(m.local_nodes, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    res = []
    for bigp in (1, 2):
        plan.set_tuning(4, bigp)
        for _ in range(3): plan.apply_stiffness_matrix(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.apply_stiffness_matrix(x, y)
        e1.record(); torch.cuda.synchronize()
        res.append(m.local_nodes / (e0.elapsed_time(e1) / 20) / 1e6)
    print("p=%d: parity vector-ALU %.1e  matrix-core %.1e | %d elements: vector-ALU %.1f GDoF/s  matrix-core %.1f GDoF/s" % (deg, errs[0], errs[1], n_el, res[0], res[1]), flush=True)
    plan.destroy()
