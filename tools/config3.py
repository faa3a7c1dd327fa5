"""BASELINE config 3: level 5 brick (32768 elements), p = 11 (56.6 MDoF): stiffness apply on the general path, parity on a sample."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level, deg = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5, 11)
t0 = time.time()
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None)
u = M.splitmix64_uniform(102321, m.local_nodes)
print("mesh: %d elements, %d DoF (%.1f s host)" % (m.n_elements, m.local_nodes, time.time() - t0), flush=True)
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst)
du = torch.from_numpy(u).to(dev); out = torch.empty_like(du)
for label, tune in (("general path", 0), ("affine path", -1)):
    plan.set_tuning(7, tune)
    for _ in range(3): plan.apply_stiffness_matrix(du, out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps): plan.apply_stiffness_matrix(du, out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print("%s: %.1f us  %.1f GDoF/s  %.0f GB/s algorithmic (64 B/DoF)  %s" % (label, t * 1e6, m.local_nodes / t / 1e9, 64 * m.local_nodes / t / 1e9, plan.last_kernel()), flush=True)
# parity on a few elements against the oracle
from tests import oracle_lib
oracle = oracle_lib.load()
plan.set_tuning(7, 0); plan.apply_stiffness_matrix(du, out)
got = out.cpu().numpy()
worst = 0.0
for e in (0, m.n_elements // 3, m.n_elements - 1):
    sub = M.BrickMesh(level, deg, first=e, count=1)
    Je, rste = sub.geometry(None)
    s, n3 = int(m.nodal_stride[e]), (deg + 1) ** 3
    ref = oracle.apply_stiffness(sub, Je, rste, np.ascontiguousarray(u[s:s + n3]))
    worst = max(worst, np.abs(got[s:s + n3] - ref).max() / np.abs(ref).max())
print("parity on sampled elements: rel-inf %.3e" % worst)
