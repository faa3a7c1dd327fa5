"""Launch-bound regime: Chebyshev iterations on small (multigrid coarse level sized) meshes, kernel launches vs hipGraph replay
(tuning key 9).  Usage: time_small.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
dev = torch.device("cuda:0")
stream = torch.cuda.Stream()
for level, deg in ((1, 2), (2, 3), (2, 7), (3, 3), (3, 7), (4, 7)):
    m = M.BrickMesh(level, deg)
    J, rst = m.geometry(None); sides = m.build_sides(None)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=stream)
    plan.set_geometry(J, rst); plan.set_faces(sides)
    u = torch.zeros(m.local_nodes, dtype=torch.float64, device=dev); rhs = torch.ones_like(u); Au = torch.empty_like(u); r = torch.empty_like(u)
    def run(): plan.cheby_iterate(u, rhs, Au, r, 10, 1.0, 30.0, 0)
    res = []
    with torch.cuda.stream(stream):
        for graph in (0, 1):
            plan.set_tuning(9, graph)
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record(stream)
            for _ in range(reps): run()
            e1.record(stream); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / reps * 1e3)
    print("level %d p %d (%d elements, %d DoF): 10 Chebyshev iterations: launches %.1f us (%.1f / iteration), hipGraph %.1f us (%.1f / iteration)"
          % (level, deg, m.n_elements, m.local_nodes, res[0], res[0] / 10, res[1], res[1] / 10))
