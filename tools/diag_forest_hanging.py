"""Diagnostic: apply_aij error vs the oracle per (f_m, f_p, orientation) triple for a hanging face on a tree boundary."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import forest as F, mesh as M, Plan
from tests import oracle_lib
from tests.test_forest import TRIPLES

oracle = oracle_lib.load()
dev = torch.device("cuda:0")
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mixed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = F.capi.load_library()
for trip, rots in sorted(TRIPLES.items()):
    conn = F.Connectivity.rotated_pair(*rots)
    row = []
    for refine in ([1, 0], [0, 1]):
        m0 = F.ForestMesh(conn, 0, deg, F.TrilinearMap(conn), refine=refine)
        d = deg + ((np.arange(m0.global_elements) * 7) % 3 if mixed else 0)
        m = F.ForestMesh(conn, 0, d, F.TrilinearMap(conn, M.SineMap(0.03)), refine=refine)
        J, rst = m.geometry(); s = m.build_sides(); u = m.field()
        ref = oracle.apply_aij(m, J, rst, s, u, penalty_prefactor=7.5, nthreads=8)
        for generic in (0, 1):
            p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
            if generic: p.set_tuning(3, 0)
            p.set_geometry(J, rst); p.set_faces(s, 7.5, 0)
            du = torch.from_numpy(u).to(dev); dAu = torch.full_like(du, float("nan"))
            p.apply_aij(du, dAu)
            got = dAu.cpu().numpy()
            row.append(np.abs(got - ref).max() / np.abs(ref).max())
            p.destroy()
    code = lib.d4est_hip_face_reorder_code(*trip)
    perm = [lib.d4est_hip_reorient_face_order(trip[0], trip[1], trip[2], i) for i in range(4)]
    print(trip, "code", code, "perm", perm, " ".join("%.1e" % e for e in row), "BAD" if max(row) > 1e-12 else "")
