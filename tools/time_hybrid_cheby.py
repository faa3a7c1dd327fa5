"""Chebyshev iteration on config 4's mesh class, update fused into the operator's kernels against the separate update kernel:
tools/time_hybrid_cheby.py <graded|hanging|dominant> [level]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
kind = sys.argv[1]; L = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
refine = np.zeros(8 ** L, dtype=bool); refine[::64] = True
ijk = M.morton_order(L)
if kind == "graded": m = M.BrickMesh(L, bench.graded_degrees(L))
elif kind == "hanging": m = M.HangingBrickMesh(L, refine, 7)
else:
    dd = np.where(ijk[:, 0] < (1 << L) // 8, 5, 7).astype(np.int32)
    m = M.HangingBrickMesh(L, refine, np.concatenate([np.full(8 if refine[b] else 1, dd[b]) for b in range(8 ** L)]).astype(np.int32))
J, rst = m.geometry(None); sides = m.build_sides(None)
x0 = torch.from_numpy(m.field()).to(dev)
rhs = torch.zeros_like(x0); Au = torch.empty_like(x0); r = torch.empty_like(x0)
res = {}
for fuse in (-1, 0):
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=st)
    p.set_geometry(J, rst); p.set_tuning(7, 0); p.set_faces(sides); p.set_tuning(10, fuse)
    x = x0.clone()
    ms_a = bench.time_region(lambda: p.apply_aij(x0, Au), 10, st, torch, warm=3)
    ms = bench.time_region(lambda: p.cheby_iterate(x, rhs, Au, r, 5, 1.0, 40.0, 0), 6, st, torch, warm=2)
    x = x0.clone(); p.cheby_iterate(x, rhs, Au, r, 5, 1.0, 40.0, 0)
    res[fuse] = x.clone()
    print("%-8s level %d %6d elements %.2f MDoF [%s]: apply %.1f us, Chebyshev iteration %s %.1f us = %.1f GDoF/s" % (
        kind, L, m.n_elements, m.local_nodes * 1e-6, p.face_path()[:40], ms_a * 1e3, "fused update" if fuse else "separate update kernel", ms * 1e3 / 5, 5 * m.local_nodes / ms * 1e-6))
    p.destroy()
print("  fused vs separate rel-inf %.2e" % float((res[-1] - res[0]).abs().max() / res[0].abs().max()))
