"""GPU parity tests of the device-resident smoother loops against the oracle's restatement of
d4est_solver_multigrid_smoother_cheby_iterate_aux and cg_eigs."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _setup(level, deg, mapping, gpu, oracle):
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(level, deg)
    J, rst = m.geometry(mapping); sides = m.build_sides(mapping)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    oracle.set_operator(m, J, rst, sides, 10.0, 0)
    return m, plan


@pytest.mark.parametrize("level,deg,curved", [(1, 3, True), (1, 7, False), (2, 2, True), (1, 8, True), (1, 11, True)])
def test_cheby_iterate_parity(gpu, hiplib, oracle, level, deg, curved):
    import torch
    from disco4est_amd import mesh as M
    m, plan = _setup(level, deg, M.SineMap(0.05) if curved else None, gpu, oracle)
    u0 = M.splitmix64_uniform(11, m.local_nodes)
    rhs = M.splitmix64_uniform(12, m.local_nodes) - 0.5
    lmax, _ = oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 10)
    lmin = lmax / 30.0
    for at_end in (1, 0):
        u_ref, r_ref = oracle.cheby_iterate(u0, rhs, 6, lmin, lmax, at_end)
        du = _t(u0, gpu); drhs = _t(rhs, gpu)
        dAu = torch.empty_like(du); dr = torch.full_like(du, float("nan"))
        plan.cheby_iterate(du, drhs, dAu, dr, 6, lmin, lmax, at_end)
        assert _rel(du.cpu().numpy(), u_ref) <= 1e-11
        assert _rel(dr.cpu().numpy(), r_ref) <= 1e-10
    # the smoother reduces the residual of a smooth-free random start (sanity of the window)
    r0 = rhs - oracle.apply_aij(m, *oracle._op_keep[1:], u0)
    assert np.linalg.norm(r_ref) < np.linalg.norm(r0)


def test_smoother_on_hanging_mesh(gpu, hiplib, oracle):
    """Chebyshev iteration and cg_eigs on a locally refined, mixed-p mesh (hanging 1 <-> 4 mortars in the operator)."""
    import torch
    from disco4est_amd import Plan, mesh as M
    refine = np.zeros(8, dtype=bool)
    refine[[1, 6]] = True
    m0 = M.HangingBrickMesh(1, refine, 2)
    m = M.HangingBrickMesh(1, refine, 2 + (np.arange(m0.n_elements) % 2))
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    oracle.set_operator(m, J, rst, sides, 10.0, 0)
    oracle.set_hanging(sides)
    try:
        u0 = M.splitmix64_uniform(21, m.local_nodes)
        rhs = M.splitmix64_uniform(22, m.local_nodes) - 0.5
        b_ref, _ = oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 10, 1)
        du = _t(np.zeros(m.local_nodes), gpu); drhs = _t(rhs, gpu); dAu = torch.empty_like(du)
        b, _ = plan.cg_eigs(du, drhs, dAu, 10, 1)
        assert abs(b - b_ref) <= 1e-9 * abs(b_ref)
        lmax, lmin = b_ref, b_ref / 30.0
        u_ref, r_ref = oracle.cheby_iterate(u0, rhs, 5, lmin, lmax, 1)
        du = _t(u0, gpu); dr = torch.full_like(du, float("nan"))
        plan.cheby_iterate(du, drhs, dAu, dr, 5, lmin, lmax, 1)
        assert _rel(du.cpu().numpy(), u_ref) <= 1e-11
        assert _rel(dr.cpu().numpy(), r_ref) <= 1e-10
    finally:
        oracle.set_hanging(None)
    plan.destroy()


@pytest.mark.parametrize("use_new", [1, 0])
def test_cg_eigs_parity(gpu, hiplib, oracle, use_new):
    import torch
    from disco4est_amd import mesh as M
    m, plan = _setup(1, 4, M.SineMap(0.05), gpu, oracle)
    u0 = np.zeros(m.local_nodes)
    rhs = M.splitmix64_uniform(5, m.local_nodes) - 0.5
    imax = 12
    b_ref, u_ref = oracle.cg_eigs(u0, rhs, imax, use_new)
    du = _t(u0, gpu); drhs = _t(rhs, gpu); dAu = torch.empty_like(du)
    b, hist = plan.cg_eigs(du, drhs, dAu, imax, use_new)
    assert abs(b - b_ref) <= 1e-9 * abs(b_ref)
    assert _rel(du.cpu().numpy(), u_ref) <= 1e-9          # cg_eigs advances u exactly like the reference
    # the recorded (alpha_i, beta_i) pushed through the reference's Gershgorin formula give the same bound
    import ctypes
    f = oracle.lib.oracle_gershgorin_bound
    f.restype = ctypes.c_double
    dp = ctypes.POINTER(ctypes.c_double)
    f.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    a = np.ascontiguousarray(hist[:imax]); bb = np.ascontiguousarray(hist[imax:])
    assert abs(f(a.ctypes.data_as(dp), bb.ctypes.data_as(dp), imax, m.local_nodes, use_new) - b) <= 1e-14 * abs(b)
    # spectral sanity: after 12 Lanczos steps the bound is within a factor ~2 of the true largest eigenvalue
    # (power iteration on the GPU operator); the reference scales it by cheby_eigs_max_multiplier afterwards.
    v = _t(M.splitmix64_uniform(9, m.local_nodes), gpu); w = torch.empty_like(v)
    lam = 0.0
    for _ in range(60):
        plan.apply_lhs(v, w)
        lam = torch.dot(v, w).item() / torch.dot(v, v).item()
        v = w / w.norm()
    assert 0.5 * lam <= b <= 2.5 * lam


def test_dot_deterministic(gpu, hiplib):
    import torch
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(1, 3)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    x = _t(M.splitmix64_uniform(1, 100003), gpu); y = _t(M.splitmix64_uniform(2, 100003), gpu)
    o1 = torch.zeros(1, dtype=torch.float64, device=gpu); o2 = torch.zeros_like(o1)
    plan.vec_dot(x, y, o1); plan.vec_dot(x, y, o2)
    assert o1.item() == o2.item()
    ref = float(np.dot(x.cpu().numpy(), y.cpu().numpy()))
    assert abs(o1.item() - ref) <= 1e-12 * abs(ref)


def test_cheby_iterate_hipgraph_replay(gpu, hiplib, oracle):
    """tuning key 9: the captured loop gives bit-identical results, is replayed while the arguments stay the same, re-captured when they
    change, and dropped when the plan changes"""
    import torch
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(1, 3)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    stream = torch.cuda.Stream()
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=stream)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    u0 = M.splitmix64_uniform(11, m.local_nodes)
    rhs = _t(M.splitmix64_uniform(12, m.local_nodes) - 0.5, gpu)

    def run(iters, lmax):
        with torch.cuda.stream(stream):
            u = _t(u0, gpu); Au = torch.empty_like(u); r = torch.full_like(u, float("nan"))
            for _ in range(2):          # second call: replay (graph) on top of the first call's result
                plan.cheby_iterate(u, rhs, Au, r, iters, 1.0, lmax, 1)
            stream.synchronize()
        return u.cpu().numpy(), r.cpu().numpy()

    plan.set_tuning(9, 0)
    ref = {k: run(*k) for k in ((4, 30.0), (6, 30.0), (4, 25.0))}
    plan.set_tuning(9, 1)
    for k in ((4, 30.0), (6, 30.0), (4, 25.0), (4, 30.0)):
        got = run(*k)
        assert np.array_equal(got[0], ref[k][0]) and np.array_equal(got[1], ref[k][1]), k
    # a change of the plan's data drops the captured graph: new boundary values must show up in the result
    g = np.ones(int(sides["total_bndry_nodes"]))
    plan.set_dirichlet_values(g)
    got = run(4, 30.0)
    assert not np.array_equal(got[0], ref[(4, 30.0)][0])
    plan.set_tuning(9, 0)
    plain = run(4, 30.0)
    assert np.array_equal(got[0], plain[0])


@pytest.mark.parametrize("level,deg", [(1, 3), (2, 7), (2, "mixed"), (1, 9), (1, 15)])
def test_cheby_update_fused_into_flux_is_bit_identical(gpu, hiplib, oracle, level, deg):
    """tuning key 10: the update carried by the flux kernel's epilogue (default on conforming meshes up to p = 15) gives
    exactly the u, r, Au of the separate update kernel"""
    import torch
    from disco4est_amd import Plan, mesh as M
    if deg == "mixed":
        deg = 2 + (np.arange(8 ** level) % 5)
    m = M.BrickMesh(level, deg)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    u0 = M.splitmix64_uniform(11, m.local_nodes)
    rhs = _t(M.splitmix64_uniform(12, m.local_nodes) - 0.5, gpu)
    out = {}
    for fuse in (0, 1):
        plan.set_tuning(10, fuse)
        for at_end in (0, 1):
            u = _t(u0, gpu); Au = torch.full_like(u, float("nan")); r = torch.full_like(u, float("nan"))
            plan.cheby_iterate(u, rhs, Au, r, 5, 1.0, 40.0, at_end)
            out[(fuse, at_end)] = (u.cpu().numpy(), r.cpu().numpy(), Au.cpu().numpy())
    for at_end in (0, 1):
        for a, b in zip(out[(0, at_end)], out[(1, at_end)]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("level,deg,inc", [(1, 3, 0), (2, 2, 1), (1, 7, 0), (1, 6, 0), (1, 8, 0), (1, 11, 0)])
def test_lhs_with_zeroth_order_term(gpu, hiplib, oracle, level, deg, inc):
    """apply_lhs = Laplacian + V^T W J c V u (the Jacobian of the reference's nonlinear problems, e.g. constant_density_star_apply_jac):
    apply_lhs, the Chebyshev iteration, cg_eigs and the Schwarz smoother with the coefficient set, against the oracle"""
    import torch
    from disco4est_amd import Plan, mesh as M
    from disco4est_amd.schwarz import Schwarz
    m = M.BrickMesh(level, deg, deg_quad_inc=inc)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=1)
    coeff = 1.0 + 5.0 * M.splitmix64_uniform(3, m.local_nodes_quad)          # positive: the operator stays SPD
    dcoeff = _t(coeff, gpu)
    try:
        oracle.set_lhs_coefficient(coeff)
        plan.set_lhs_coefficient(dcoeff)
        u0 = M.splitmix64_uniform(11, m.local_nodes)
        rhs = M.splitmix64_uniform(12, m.local_nodes) - 0.5
        du = _t(u0, gpu); dAu = torch.empty_like(du)
        plan.apply_lhs(du, dAu)
        ref = oracle.apply_lhs(u0)
        assert _rel(dAu.cpu().numpy(), ref) <= 1e-12
        # really a different operator than the Laplacian
        plan.apply_aij(du, dAu)
        assert _rel(dAu.cpu().numpy(), ref) > 1e-6
        lmax, _ = oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 8)
        b, _ = plan.cg_eigs(torch.zeros_like(du), _t(rhs, gpu), dAu, 8)
        assert abs(b - lmax) <= 1e-10 * lmax
        u_ref, r_ref = oracle.cheby_iterate(u0, rhs, 5, lmax / 30.0, lmax, 1)
        du = _t(u0, gpu); dr = torch.empty_like(du)
        plan.cheby_iterate(du, _t(rhs, gpu), dAu, dr, 5, lmax / 30.0, lmax, 1)
        assert _rel(du.cpu().numpy(), u_ref) <= 1e-11 and _rel(dr.cpu().numpy(), r_ref) <= 1e-10
        if deg <= 3:
            sz = Schwarz(m, sides, J, rst, 2, 5, 1e-15, 1e-15)
            sz.plan.set_lhs_coefficient(dcoeff)                    # the subdomain operator carries the term too
            u_ref, it_ref, _ = oracle.schwarz_iterate(sz.metadata, u0, rhs, 5, 1e-15, 1e-15)
            du = _t(u0, gpu)
            sz.iterate(du, _t(rhs, gpu))
            assert _rel(du.cpu().numpy() - u0, u_ref - u0) <= 1e-9
            sz.destroy()
    finally:
        oracle.set_lhs_coefficient(None)
        plan.set_lhs_coefficient(None)
