"""A/B of the full operator and the Chebyshev loop: two-phase face kernels (tuning key 11 = 0) against the direct face kernel."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 7
affine = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # 0: general path (the metric is streamed, as the headline does); -1: affine buckets detected
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
def t(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
out = {}
for direct in (0, 1, 2):
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry(J, rst); plan.set_tuning(7, affine); plan.set_tuning(11, direct); plan.set_faces(sides)
    du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
    torch.manual_seed(5); rhs = torch.rand_like(du); r = torch.empty_like(du)
    plan.apply_aij(du, Au); torch.cuda.synchronize()
    out[direct] = Au.cpu().numpy().copy()
    t_aij = t(lambda: plan.apply_aij(du, Au))
    t_st = t(lambda: plan.apply_stiffness_matrix(du, Au))
    # a real eigenvalue window (power iteration), so that the 5 iterations contract instead of amplifying rounding differences
    torch.manual_seed(6); v = torch.rand_like(du)
    for _ in range(30):
        plan.apply_aij(v, Au); lam = float(torch.linalg.norm(Au) / torch.linalg.norm(v)); v = Au / torch.linalg.norm(Au)
    lmax = 1.1 * lam
    uc = du.clone()
    t_ch = t(lambda: plan.cheby_iterate(uc, rhs, Au, r, 5, lmax / 30, lmax, 0), reps=10) / 5
    uc = du.clone(); plan.cheby_iterate(uc, rhs, Au, r, 5, lmax / 30, lmax, 0); torch.cuda.synchronize()
    out[("c", direct)] = uc.cpu().numpy().copy()
    print(plan.face_path(), end=": ")
    print("level %d p %d direct=%d: apply_aij %.1f us (stiffness alone %.1f) | cheby %.1f us / iteration" % (level, deg, direct, t_aij, t_st, t_ch), flush=True)
    plan.destroy()
for k in (1, 2):
    d = np.abs(out[0] - out[k]).max() / np.abs(out[0]).max()
    dc = np.abs(out[("c", 0)] - out[("c", k)]).max() / np.abs(out[("c", 0)]).max()
    print("rel-inf difference, path %d against the two-phase kernels: apply_aij %.2e, 5 Chebyshev iterations %.2e" % (k, d, dc))
