"""Create / use / destroy plans, transfer objects and Schwarz smoothers in a loop and watch the device's free memory."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, Transfer, mesh as M
from disco4est_amd.schwarz import Schwarz
dev = torch.device("cuda:0")
m = M.BrickMesh(2, 4); mp = M.SineMap(0.03)
J, rst = m.geometry(mp); sides = m.build_sides(mp)
refine = np.zeros(8, dtype=bool); refine[[1, 6]] = True
mh = M.HangingBrickMesh(1, refine, 3)
Jh, rsth = mh.geometry(mp); sh = mh.build_sides(mp)
degm = np.where(np.arange(mh.global_elements) % 3 == 0, 8, 3).astype(np.int32)
mm = M.HangingBrickMesh(1, refine, degm)
Jm, rstm = mm.geometry(mp); sm = mm.build_sides(mp)
um = torch.from_numpy(mm.field(mp)).to(dev); Aum = torch.empty_like(um)
u = torch.from_numpy(m.field(mp)).to(dev); Au = torch.empty_like(u); r = torch.empty_like(u)
uh = torch.from_numpy(mh.field(mp)).to(dev); Auh = torch.empty_like(uh)
def cycle():
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    p.set_geometry(J, rst); p.set_faces(sides)
    p.apply_aij(u, Au); p.cheby_iterate(u.clone(), Au, torch.empty_like(u), r, 3, 1.0, 30.0, 1); p.cg_eigs(torch.zeros_like(u), Au, r, 4)
    coeff = torch.ones(p.local_nodes_quad, dtype=torch.float64, device=dev); p.set_lhs_coefficient(coeff); p.apply_lhs(u, Au)
    sz = Schwarz(m, sides, J, rst, 2, 3, 1e-15, 1e-15)
    sz.iterate(torch.zeros_like(u), Au); sz.smooth(p, torch.zeros_like(u), Au, r, 1)
    sz.destroy(); p.destroy()
    ph = Plan(mh.deg, mh.deg_quad, mh.nodal_stride, mh.quad_stride, 0)
    ph.set_geometry(Jh, rsth); ph.set_faces(sh); ph.apply_aij(uh, Auh)
    # (hanging plan: hybrid operator in its hanging-aware form, unit record kernels, the fused Chebyshev update with its second vector)
    ph.cheby_iterate(uh.clone(), Auh.clone(), torch.empty_like(uh), torch.empty_like(uh), 3, 1.0, 30.0, 0)
    ph.destroy()
    for k14 in (1, 0):   # the hybrid operator forced / off on a mixed-degree hanging plan (lists, side streams, family lists)
        pm = Plan(mm.deg, mm.deg_quad, mm.nodal_stride, mm.quad_stride, 0)
        pm.set_tuning(14, k14); pm.set_geometry(Jm, rstm); pm.set_faces(sm); pm.apply_aij(um, Aum); pm.destroy()
    n = 8
    t = Transfer(np.ones(n, dtype=np.int32), np.full(n, 2, dtype=np.int32), np.full(8 * n, 3, dtype=np.int32))
    xc = torch.zeros(t.coarse_nodes, dtype=torch.float64, device=dev); xf = torch.zeros(t.fine_nodes, dtype=torch.float64, device=dev)
    t.prolong(xc, xf); t.restrict(xf, xc); t.project(xf, xc); t.destroy()
for _ in range(3): cycle()
torch.cuda.synchronize(); torch.cuda.empty_cache()
free0 = torch.cuda.mem_get_info()[0]
for i in range(40): cycle()
torch.cuda.synchronize(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print("free device memory before / after 40 create-use-destroy cycles: %.1f MB / %.1f MB (delta %.2f MB)" % (free0 / 1e6, free1 / 1e6, (free0 - free1) / 1e6))
sys.exit(0 if free0 - free1 < 8e6 else 1)
