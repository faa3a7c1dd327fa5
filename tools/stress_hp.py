import sys, numpy as np, torch, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib
from disco4est_amd import Plan, mesh as M
o = oracle_lib.load()
dev = torch.device("cuda:0")
rng = np.random.RandomState(5)
worst = 0
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 14):
    level = 1 if trial % 3 else 2
    nb = 8 ** level
    refine = rng.rand(nb) < 0.3
    if not refine.any(): refine[0] = True
    if refine.all(): refine[0] = False
    base = int(rng.randint(1, 6)); span = int(rng.randint(1, 4)); inc = int(rng.randint(0, 3)); qt = int(rng.randint(0, 2))
    m0 = M.HangingBrickMesh(level, refine, base)
    deg = base + rng.randint(0, span, size=m0.global_elements)
    m = M.HangingBrickMesh(level, refine, deg, deg_quad_inc=inc, quad_type=qt)
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp); u = m.field(mp)
    g = np.cos(sides["bndry_xyz"][0]) * sides["bndry_xyz"][2]
    fcn = trial % 4
    ref = o.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=3.0 + trial, penalty_fcn=fcn, nthreads=8)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, qt)
    plan.set_geometry(J, rst); plan.set_faces(sides, 3.0 + trial, fcn); plan.set_dirichlet_values(g)
    for tune in (-1, 0):   # MFMA record kernels / generic record kernels
        plan.set_tuning(3, tune)
        du = torch.from_numpy(u).to(dev); out = torch.full_like(du, float("nan"))
        plan.apply_aij(du, out)
        err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
        worst = max(worst, err)
        print("trial %d level %d elems %d p %d..%d inc %d qt %d fcn %d hang %d tune %d: rel err %.2e" % (trial, level, m.n_elements, deg.min(), deg.max(), inc, qt, fcn, int((sides["side_hang"] == 1).sum()), tune, err), flush=True)
    plan.destroy()
    # the conforming / hanging split of the face kernels forced (tuning key 13 = 1; degrees <= 7 only: it is ignored otherwise)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, qt)
    plan.set_geometry(J, rst); plan.set_tuning(13, 1); plan.set_faces(sides, 3.0 + trial, fcn); plan.set_dirichlet_values(g)
    du = torch.from_numpy(u).to(dev); out = torch.full_like(du, float("nan"))
    plan.apply_aij(du, out)
    err = np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max()
    worst = max(worst, err)
    print("trial %d ... split forced: rel err %.2e" % (trial, err), flush=True)
    plan.destroy()
print("worst", worst)
assert worst < 1e-12
