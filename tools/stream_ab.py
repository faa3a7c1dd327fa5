"""Stream mode (tuning key 12: non-temporal metric / factor loads and A u stores) off / on / automatic choice, same process, same plan:
volume apply and whole operator.  usage: tools/stream_ab.py [quick]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
def t(fn, reps):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cases = [(7, 4, None, True), (7, 5, None, True), (11, 4, None, True), (11, 5, None, True), (9, 4, None, False), (8, 4, None, False), (12, 4, 2048, False),
         (13, 4, 2048, False), (15, 4, 2048, True), (15, 5, 8192, False), (5, 5, None, False), (3, 6, 65536, False)]
if len(sys.argv) > 1: cases = cases[:4]
for deg, level, count, faces in cases:
    m = M.BrickMesh(level, deg, count=count)
    J, rst = m.geometry(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, 0)
    if faces and count is None: plan.set_faces(m.build_sides(None))
    du = torch.from_numpy(u).to(dev); out = torch.empty_like(du)
    mb = (48 * m.local_nodes_quad + 16 * m.local_nodes) / 1e6
    reps = 30 if m.local_nodes < 3e7 else 10
    res = {0: [[], []], 1: [[], []]}; ref = None
    for rnd in range(3):                     # alternate, so that clock / cache warm-up cannot favour one setting
        for key in (0, 1):
            plan.set_tuning(12, key)
            res[key][0].append(t(lambda: plan.apply_stiffness_matrix(du, out), reps))
            if ref is None: ref = out.clone()
            same = torch.equal(ref, out)
            if faces and count is None:
                res[key][1].append(t(lambda: plan.apply_aij(du, out), reps))
    plan.set_tuning(12, -1)
    plan.apply_stiffness_matrix(du, out)
    f = lambda v: "/".join("%.1f" % x for x in v)
    line = "p=%2d elements %6d (%6.0f MB per apply): stiffness off %s on %s us" % (deg, m.n_elements, mb, f(res[0][0]), f(res[1][0]))
    if res[0][1]: line += " | apply_aij off %s on %s us" % (f(res[0][1]), f(res[1][1]))
    print(line + "  same=%s  auto: %s" % (same, plan.last_kernel()), flush=True)
    plan.destroy(); del du, out
