"""Apply time vs number of elements (p=7) for the default kernel selection: separates fixed latency from throughput."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
deg = 7
for count in (256, 512, 1024, 2048, 3072, 4096, 6144, 8192, 16384):
    m = M.BrickMesh(5, deg, count=count)
    J, rst = m.geometry(None); u = m.field()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst)
    for k, v in [a.split("=") for a in sys.argv[1:]]:
        plan.set_tuning(int(k), int(v))
    du = torch.from_numpy(u).to(dev); out = torch.empty_like(du)
    for _ in range(10): plan.apply_stiffness_matrix(du, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): plan.apply_stiffness_matrix(du, out)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 50 * 1e3)
    t = sorted(ts)[len(ts) // 2]
    print("elements %6d (%.1f per CU): %7.2f us  %.2f ns/element  %s" % (count, count / 256, t, t * 1e3 / count, plan.last_kernel()))
    plan.destroy()
