"""The device pieces compose into a multigrid cycle without leaving the GPU: Chebyshev smoother (d4est_hip_cheby_iterate, window from
d4est_hip_cg_eigs as d4est_solver_multigrid_smoother_cheby.c:208-217 does), residual restriction with the transposed prolongation and
coarse-grid correction with the prolongation (d4est_solver_multigrid_callbacks.h:100-330), coarse operator by re-discretisation.
Not a restatement of the reference's V-cycle driver (out of scope) -- an integration test of the hot-path components it calls."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _level(m, mp, gpu, prefactor):
    from disco4est_amd import Plan
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, prefactor, 0)
    return plan


@pytest.mark.parametrize("hanging", [False, True])
def test_two_grid_cycle_converges(gpu, hiplib, hanging):
    import torch
    from disco4est_amd import Transfer, mesh as M
    mp = M.SineMap(0.03)
    if hanging:
        refine = np.zeros(8, dtype=bool)
        refine[[1, 6]] = True
        n_el = M.HangingBrickMesh(1, refine, 1).n_elements
        deg_f = 3 + (np.arange(n_el) % 2)          # fine grid: p = 3 / 4 on a locally refined mesh
        deg_c = np.full(n_el, 1)                    # coarse grid: p = 1 on the same elements (p-multigrid)
        mf = M.HangingBrickMesh(1, refine, deg_f)
        mc = M.HangingBrickMesh(1, refine, deg_c)
    else:
        n_el = 64
        deg_f, deg_c = np.full(n_el, 4), np.full(n_el, 2)
        mf, mc = M.BrickMesh(2, deg_f), M.BrickMesh(2, deg_c)
    # the coarse operator is a re-discretisation; its SIPG penalty (prefactor p^2 / h) is raised to the fine level's value
    # ("inherited" penalty), otherwise the coarse problem is too soft on the jump modes and the correction overshoots
    ratio = float(np.max(deg_f)) / float(np.max(deg_c))
    pf, pc = _level(mf, mp, gpu, 10.0), _level(mc, mp, gpu, 10.0 * ratio * ratio)
    degh = np.zeros(8 * n_el, dtype=np.int32)
    degh[0::8] = deg_f
    T = Transfer(np.zeros(n_el, dtype=np.int32), deg_c.astype(np.int32), degh)
    assert T.fine_nodes == mf.local_nodes and T.coarse_nodes == mc.local_nodes

    def vec(n, fill=0.0):
        return torch.full((n,), fill, dtype=torch.float64, device=gpu)

    x, y, z = mf.nodal_coords(mp)
    u_exact = torch.from_numpy(np.sin(2.0 * x) * np.cos(1.5 * y) + z * z + 0.02 * (M.splitmix64_uniform(3, mf.local_nodes) - 0.5)).to(gpu)
    rhs = vec(mf.local_nodes)
    pf.apply_aij(u_exact, rhs)
    # smoother windows from the CG-Lanczos estimate (10 iterations from a zero start), as the reference does
    lmax_f, _ = pf.cg_eigs(vec(mf.local_nodes), rhs, vec(mf.local_nodes), 20, 1)
    lmax_f *= 1.2                                   # cheby_eigs_max_multiplier of the reference's input files
    lmin_f = lmax_f / 10.0
    u = vec(mf.local_nodes)
    Au, r = vec(mf.local_nodes), vec(mf.local_nodes)
    rc, ec, Ac = vec(mc.local_nodes), vec(mc.local_nodes), vec(mc.local_nodes)
    ef = vec(mf.local_nodes)

    def err():
        return (u - u_exact).norm().item() / u_exact.norm().item()

    e0 = err()
    history = [e0]
    for cycle in range(4):
        pf.cheby_iterate(u, rhs, Au, r, 4, lmin_f, lmax_f, 1)          # pre-smoothing, r = rhs - A u on exit
        T.restrict(r, rc)                                              # residual to the coarse grid
        ec.zero_()
        pc.cg_eigs(ec, rc, Ac, 60, 1)                                   # coarse solve: 60 CG iterations
        T.prolong(ec, ef)                                              # correction
        u += ef
        pf.cheby_iterate(u, rhs, Au, r, 4, lmin_f, lmax_f, 1)          # post-smoothing
        history.append(err())
    # smoothing alone with the same number of fine applies, for comparison
    us = vec(mf.local_nodes)
    pf.cheby_iterate(us, rhs, Au, r, 32, lmin_f, lmax_f, 1)
    e_smooth = (us - u_exact).norm().item() / u_exact.norm().item()
    assert all(b < a for a, b in zip(history[:-1], history[1:])), history
    assert history[-1] < 0.4 * e0, (history, e_smooth)
    assert history[-1] < 0.5 * e_smooth, (history, e_smooth)   # the coarse-grid correction does the work
    for p in (pf, pc):
        p.destroy()
    T.destroy()


def test_two_grid_cycle_with_schwarz_smoother(gpu, hiplib):
    """the same cycle with the additive Schwarz smoother (d4est_solver_multigrid_smoother_schwarz = d4est_hip_schwarz_smooth) instead of
    Chebyshev: it is contractive, beats smoothing alone, and leaves r = rhs - A u on exit like the Chebyshev smoother"""
    import torch
    from disco4est_amd import Transfer, mesh as M
    from disco4est_amd.schwarz import Schwarz
    mp = M.SineMap(0.03)
    n_el = 64
    deg_f, deg_c = np.full(n_el, 4), np.full(n_el, 2)
    mf, mc = M.BrickMesh(2, deg_f), M.BrickMesh(2, deg_c)
    pf, pc = _level(mf, mp, gpu, 10.0), _level(mc, mp, gpu, 10.0 * 4.0)
    Jf, rstf = mf.geometry(mp); sf = mf.build_sides(mp)
    sz = Schwarz(mf, sf, Jf, rstf, 2, 12, 1e-15, 1e-4, 10.0, 0)        # overlap 2, loose subdomain solves (a smoother, not a solver)
    degh = np.zeros(8 * n_el, dtype=np.int32)
    degh[0::8] = deg_f
    T = Transfer(np.zeros(n_el, dtype=np.int32), deg_c.astype(np.int32), degh)
    vec = lambda n: torch.zeros(n, dtype=torch.float64, device=gpu)
    x, y, z = mf.nodal_coords(mp)
    u_exact = torch.from_numpy(np.sin(2.0 * x) * np.cos(1.5 * y) + z * z + 0.02 * (M.splitmix64_uniform(3, mf.local_nodes) - 0.5)).to(gpu)
    rhs = vec(mf.local_nodes)
    pf.apply_aij(u_exact, rhs)
    u, r, Au = vec(mf.local_nodes), vec(mf.local_nodes), vec(mf.local_nodes)
    rc, ec, Ac, ef = vec(mc.local_nodes), vec(mc.local_nodes), vec(mc.local_nodes), vec(mf.local_nodes)
    err = lambda v: (v - u_exact).norm().item() / u_exact.norm().item()
    history = [err(u)]
    for cycle in range(3):
        sz.smooth(pf, u, rhs, r, 1)                                    # pre-smoothing; r = rhs - A u on exit
        pf.apply_aij(u, Au)
        assert float((r - (rhs - Au)).abs().max()) <= 1e-11 * float(rhs.abs().max())
        T.restrict(r, rc)
        ec.zero_()
        pc.cg_eigs(ec, rc, Ac, 60, 1)
        T.prolong(ec, ef)
        u += ef
        sz.smooth(pf, u, rhs, r, 1)                                    # post-smoothing
        history.append(err(u))
    us = vec(mf.local_nodes)
    sz.smooth(pf, us, rhs, r, 6)                                       # smoothing alone, same number of Schwarz iterations
    assert all(b < 0.6 * a for a, b in zip(history[:-1], history[1:])), history
    assert history[-1] < 0.5 * err(us), (history, err(us))
    sz.destroy()
    for p in (pf, pc):
        p.destroy()
    T.destroy()
