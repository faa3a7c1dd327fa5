"""The hybrid operator against the all-two-phase form on config 4's mesh class: tools/time_hybrid.py <mesh>
  graded   : level-4 brick, p = 3 ... 9 graded smoothly (bench.graded_degrees)
  plateau  : level-4 brick, p = 3, 5, 7, 9 in slabs four elements thick
  hanging  : level-4 brick, every 64th octant refined, p = 7
  hanging1 : the same brick with ONE refined octant
  dominant : hanging, p = 7 but for a slab of p = 5 (12.5 % of the mesh)
  combined : hanging + plateau degrees"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
import importlib.util
spec = importlib.util.spec_from_file_location("bench", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
kind = sys.argv[1]
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4   # refinement level of the base brick
dev = torch.device("cuda:0"); st = torch.cuda.current_stream()
ijk = M.morton_order(L)
plateau = np.asarray([3, 5, 7, 9], np.int32)[(ijk[:, 0] * 4) // (1 << L)]
refine = np.zeros(8 ** L, dtype=bool); refine[::64] = True
if kind == "graded": m = M.BrickMesh(L, bench.graded_degrees(L))
elif kind == "plateau": m = M.BrickMesh(L, plateau)
elif kind == "hanging": m = M.HangingBrickMesh(L, refine, 7)
elif kind == "dominant":   # hanging + one dominant degree: p = 7 but for a slab of p = 5 two elements thick (12.5 % of the mesh)
    dd = np.where(ijk[:, 0] < (1 << L) // 8, 5, 7).astype(np.int32)
    m = M.HangingBrickMesh(L, refine, np.concatenate([np.full(8 if refine[b] else 1, dd[b]) for b in range(8 ** L)]).astype(np.int32))
elif kind == "hanging1":   # ONE refined octant: a hanging plan that is uniform but for eight elements (what the hybrid machinery itself costs)
    refine[:] = False; refine[8 ** L // 2 + 5] = True
    m = M.HangingBrickMesh(L, refine, 7)
else: m = M.HangingBrickMesh(L, refine, np.concatenate([np.full(8 if refine[b] else 1, plateau[b]) for b in range(8 ** L)]).astype(np.int32))
J, rst = m.geometry(None); sides = m.build_sides(None)
x = torch.from_numpy(m.field()).to(dev); y = torch.empty_like(x)
by = bench.mixed_operator_bytes(m, sides)
out = {}
modes = (0, -1) if not os.environ.get('HYB_ONLY') else (int(os.environ['HYB_ONLY']),)
for hyb in modes:
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=st)
    p.set_tuning(14, hyb); p.set_geometry(J, rst); p.set_tuning(7, 0); p.set_faces(sides)
    ms = bench.time_region(lambda: p.apply_aij(x, y), 50, st, torch, warm=10)
    out[hyb] = (ms, y.clone(), p.face_path())
    p.destroy()
err = float((out[modes[0]][1] - out[modes[-1]][1]).abs().max() / out[modes[0]][1].abs().max())
for hyb in modes:
    ms = out[hyb][0]
    print("%-9s %4d elements %.2f MDoF  [%s]: apply_aij %.1f us = %.1f GDoF/s, %.2f of the HBM roof (%.0f B/DoF)" %
          (kind, m.n_elements, m.local_nodes * 1e-6, out[hyb][2], ms * 1e3, m.local_nodes / ms * 1e-6, by / (ms * 1e-3) / 8e12, by / m.local_nodes))
print("  hybrid vs two-phase rel-inf %.2e" % err)
