"""Random locally refined, mixed-degree, curved meshes through every form of the operator on config 4's mesh class -- default path, hybrid
operator forced (tuning key 14 = 1: hanging-aware where every degree is <= 7), hybrid off (two-phase: hp split incl. degrees above 7, unit
record kernels), hp split off (record kernels throughout) -- each against the oracle.  tools/stress_hybrid.py [trials]"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracle_lib
from disco4est_amd import Plan, mesh as M
o = oracle_lib.load()
dev = torch.device("cuda:0")
rng = np.random.RandomState(11)
worst = 0.0
paths = {}
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 18):
    level = 1 if trial % 3 else 2
    nb = 8 ** level
    refine = rng.rand(nb) < (0.15 if level == 2 else 0.3)
    if not refine.any(): refine[0] = True
    if refine.all(): refine[0] = False
    base = int(rng.randint(1, 10)); span = int(rng.randint(1, 4))
    m0 = M.HangingBrickMesh(level, refine, base)
    kind = trial % 3      # 0: one degree, 1: scattered degrees, 2: one dominant degree
    if kind == 0: deg = np.full(m0.global_elements, base)
    elif kind == 1: deg = base + rng.randint(0, span, size=m0.global_elements)
    else: deg = np.where(rng.rand(m0.global_elements) < 0.15, base + 1, base)
    m = M.HangingBrickMesh(level, refine, deg.astype(np.int32))
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp); u = m.field(mp)
    g = np.cos(sides["bndry_xyz"][0]) * sides["bndry_xyz"][2]
    fcn = trial % 4
    ref = o.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=3.0 + trial, penalty_fcn=fcn, nthreads=8)
    for name, k13, k14 in (("default", -1, -1), ("hybrid forced", -1, 1), ("hybrid off", -1, 0), ("split off", 0, 0)):
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_tuning(13, k13); plan.set_tuning(14, k14)
        plan.set_geometry(J, rst); plan.set_faces(sides, 3.0 + trial, fcn); plan.set_dirichlet_values(g)
        du = torch.from_numpy(u).to(dev); out = torch.full_like(du, float("nan"))
        plan.apply_aij(du, out)
        out2 = torch.full_like(du, float("nan")); plan.apply_aij(du, out2)
        assert torch.equal(out, out2), "not deterministic"
        err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        path = plan.face_path().split(":")[0]
        paths[path] = paths.get(path, 0) + 1
        print("trial %2d level %d elems %4d p %d..%d kind %d hang %3d  %-13s [%s]: rel err %.2e" % (trial, level, m.n_elements, deg.min(), deg.max(), kind, int((sides["side_hang"] == 1).sum()), name, plan.face_path()[:60], err), flush=True)
        plan.destroy()
print("paths taken:", paths)
print("worst", worst)
assert worst < 1e-12
