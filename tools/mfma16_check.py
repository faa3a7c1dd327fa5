"""p = 15 stiffness: parity of the matrix-core kernel (tuning 4 = 2) against the oracle on a curved brick with more elements than
CUs (persistent loop), and its time against the vector-ALU kernel (tuning 4 = 1) at the bench's size."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
from tests import oracle_lib
dev = torch.device("cuda:0")
oracle = oracle_lib.load()
m = M.BrickMesh(3, 15, count=300)
mp = M.SineMap(0.05)
J, rst = m.geometry(mp); u = m.field(mp)
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
plan.set_geometry(J, rst)
du = torch.from_numpy(u).to(dev)
ref = None
for bigp in (1, 2):
    plan.set_tuning(4, bigp)
    out = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, out); torch.cuda.synchronize()
    got = out.cpu().numpy()
    if ref is None:
        sel = [0, 1, 2, 150, 255, 256, 257, 299]
        ref = {}
        for e in sel:
            sub = M.BrickMesh(3, 15, first=e, count=1)
            Je, rste = sub.geometry(mp)
            ref[e] = oracle.apply_stiffness(sub, Je, rste, np.ascontiguousarray(u[e * 4096:(e + 1) * 4096]))
    worst = max(np.abs(got[e * 4096:(e + 1) * 4096] - r).max() / np.abs(r).max() for e, r in ref.items())
    print("bigp=%d  %s  finite=%s  rel-inf vs oracle on %d elements: %.3e" % (bigp, plan.last_kernel(), np.isfinite(got).all(), len(ref), worst), flush=True)
    if bigp == 1: base = got
    else: print("   vs vector-ALU kernel over all elements: %.3e" % (np.abs(got - base).max() / np.abs(base).max()))
plan.destroy()
for n_el in (2048, 256, 8192):
    m = M.BrickMesh(5, 15, count=n_el)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry_brick(np.ones(m.n_elements, dtype=np.int32), 32.0, [0, 1, 0, 1, 0, 1.0]); plan.set_tuning(7, 0)
    x = torch.rand(m.local_nodes, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    for bigp in (1, 2):
        plan.set_tuning(4, bigp)
        for _ in range(3): plan.apply_stiffness_matrix(x, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): plan.apply_stiffness_matrix(x, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print("%5d elements bigp=%d: %.1f us  %.1f GDoF/s  (%.2f of the 125 GDoF/s HBM roof)" % (n_el, bigp, ms * 1e3, m.local_nodes / ms / 1e6, m.local_nodes / ms / 1e6 / 125), flush=True)
    plan.destroy()
