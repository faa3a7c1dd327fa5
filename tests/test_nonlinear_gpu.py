"""The pieces compose into the reference's Newton-Krylov flow for a nonlinear problem of the ConstantDensityStar kind,
-Laplace(u) + u^3 = g with Dirichlet data (src/Problems/ConstantDensityStar/constant_density_star_fcns.h: build_residual = apply_aij +
apply_fofufofvlj, apply_jac = apply_aij + apply_fofufofvlilj): residual and Jacobian are evaluated on the device through the C-ABI
(interpolate, galerkin integral, apply_aij with boundary data, apply_lhs with the zeroth-order coefficient); the residual is checked
against the oracle, Newton converges quadratically and the discrete solution approximates the smooth exact one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_newton_on_cubic_reaction_diffusion(gpu, hiplib, oracle):
    import torch
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(1, 4, deg_quad_inc=1)
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    exact = lambda x, y, z: np.sin(1.3 * x + 0.4) * np.cos(0.9 * y) * np.exp(0.5 * z)
    lap = lambda x, y, z: (-(1.3 ** 2) - 0.9 ** 2 + 0.25) * exact(x, y, z)
    x, y, z = m.nodal_coords(mp)
    u_star = exact(x, y, z)
    # coordinates of the quadrature nodes: interpolate the nodal coordinates (isoparametric, like the reference's xyz_quad)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    q = [torch.empty(m.local_nodes_quad, dtype=torch.float64, device=gpu) for _ in range(3)]
    for c, t in zip((x, y, z), q):
        plan.interpolate(T(c), t)
    xq, yq, zq = (t.cpu().numpy() for t in q)
    g_q = T(-lap(xq, yq, zq) + exact(xq, yq, zq) ** 3)
    bx = sides["bndry_xyz"]
    g_bnd = exact(bx[0], bx[1], bx[2])
    Jd = T(J)

    def residual(u):
        plan.set_dirichlet_values(g_bnd)
        Au = torch.empty_like(u)
        plan.apply_aij(u, Au)
        uq = torch.empty(m.local_nodes_quad, dtype=torch.float64, device=gpu)
        plan.interpolate(u, uq)
        out = torch.empty_like(u)
        plan.apply_galerkin_integral(uq ** 3 - g_q, out)
        return Au + out, uq

    def jacobian(uq):
        plan.set_dirichlet_values(None)
        coeff = 3.0 * uq ** 2
        plan.set_lhs_coefficient(coeff)
        def apply(v):
            w = torch.empty_like(v)
            plan.apply_lhs(v, w)
            return w
        return apply

    def cg(apply, b, iters=400, tol=1e-13):
        xk = torch.zeros_like(b); r = b.clone(); d = r.clone(); rr = float(r @ r); r0 = rr
        for _ in range(iters):
            Ad = apply(d)
            a = rr / float(d @ Ad)
            xk += a * d; r -= a * Ad
            rn = float(r @ r)
            if rn <= tol * tol * r0:
                break
            d = r + (rn / rr) * d; rr = rn
        return xk

    u = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
    R, uq = residual(u)
    # the device residual is the reference's build_residual: A u (with boundary data) + V^T W J (u^3 - g)
    R_ref = oracle.apply_aij(m, J, rst, sides, u.cpu().numpy(), bndry_lobatto=g_bnd) + oracle.apply_galerkin(m, J, (uq ** 3 - g_q).cpu().numpy())
    assert np.abs(R.cpu().numpy() - R_ref).max() <= 1e-12 * np.abs(R_ref).max()
    hist = [float(R.norm())]
    for _ in range(6):
        du = cg(jacobian(uq), -R)
        u = u + du
        R, uq = residual(u)
        hist.append(float(R.norm()))
        if hist[-1] <= 1e-11 * hist[0]:
            break
    plan.set_lhs_coefficient(None)
    assert hist[-1] <= 1e-10 * hist[0], hist
    assert len(hist) <= 7 and hist[2] < 0.05 * hist[1]                    # Newton, not a crawl
    err = float((u - T(u_star)).abs().max())
    assert err < 2e-3, err                                                 # p = 4 on 8 elements


@pytest.mark.parametrize("deg,level,direct", [(7, 2, 2), (3, 1, 0), (9, 1, 2)])
def test_lhs_term_follows_a_later_change_of_the_jacobian(gpu, hiplib, deg, level, direct):
    """The fused operator kernels read w J c pre-combined at plan_set_lhs_coefficient; d4est_hip_plan_set_jacobian overwrites J afterwards
    (advisor, round 3): the cached stream must be rebuilt, so that apply_lhs = apply_aij + the weighted mass term with the NEW J on the
    whole-operator path (direct = 2) exactly as on the separate-kernel path (direct = 0)"""
    import ctypes
    import torch
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(level, deg)
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_tuning(11, direct)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    c = T(0.5 + M.splitmix64_uniform(1, m.local_nodes_quad))
    u = T(m.field(mp))
    plan.set_lhs_coefficient(c)
    a0 = torch.empty_like(u)
    plan.apply_lhs(u, a0)                      # builds w J c with the first J
    J2 = T(J * (1.0 + 0.5 * M.splitmix64_uniform(2, m.local_nodes_quad)))
    plan.lib.d4est_hip_plan_set_jacobian(plan.handle, ctypes.c_void_p(J2.data_ptr()), 1)
    a1, lap, wm = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    plan.apply_lhs(u, a1)
    plan.apply_aij(u, lap)
    plan.apply_weighted_mass_matrix(u, c, wm)   # reads the plan's (new) J
    ref = lap + wm
    assert float((a1 - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    assert float((a1 - a0).abs().max()) > 1e-6 * float(ref.abs().max())     # the change of J is visible in the term
    plan.destroy()
