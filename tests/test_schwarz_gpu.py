"""GPU parity tests of the batched additive Schwarz smoother (csrc/d4est_hip_schwarz.hip, disco4est_amd/schwarz.py) against the
oracle's serial restatement of d4est_solver_schwarz_iterate (oracle/d4est_oracle_schwarz.c).  Tolerances (fp64): the restriction is
a copy and must be bit-exact; the weighted correction follows the reference's order of additions and is compared at 2e-15 absolute
on O(1) data (the hat weights come from two independent node tables); the subdomain operator is compared at 1e-12 relative to the largest entry; a whole Schwarz iteration runs several CG iterations per
subdomain on differently rounded operators, so it is compared at 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _setup(level, deg, curved, rs, oracle, iters=200, atol=1e-15, rtol=1e-15, deg_quad_inc=0):
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import Schwarz
    m = M.BrickMesh(level, deg, deg_quad_inc=deg_quad_inc)
    mp = M.SineMap(0.04) if curved else None
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=1)
    sz = Schwarz(m, sides, J, rst, rs, iters, atol, rtol, 10.0, 0)
    return m, J, rst, sides, sz


def _over_subdomains_to_restricted(oracle, m, md, x, s):
    """restricted field of subdomain s (reference layout) out of a field over the subdomains"""
    a, b = int(md.sub_first[s]), int(md.sub_first[s + 1])
    off = np.concatenate([[0], np.cumsum(md.elem_nodal_size)])
    return np.concatenate([oracle.schwarz_apply_restrictor(x[off[k]:off[k + 1]], md.sub_faces[k], int(m.deg[md.sub_elem[k]]),
                                                           md.num_nodes_overlap) for k in range(a, b)])


def _restricted_to_over_subdomains(oracle, m, md, xr, s, out):
    a, b = int(md.sub_first[s]), int(md.sub_first[s + 1])
    off = np.concatenate([[0], np.cumsum(md.elem_nodal_size)])
    ro = 0
    for k in range(a, b):
        n = int(md.elem_restricted_nodal_size[k])
        out[off[k]:off[k + 1]] = oracle.schwarz_apply_restrictor(xr[ro:ro + n], md.sub_faces[k], int(m.deg[md.sub_elem[k]]),
                                                                  md.num_nodes_overlap, transpose=True)
        ro += n


def _mixed(level, lo, hi):
    n = 8 ** level
    return lo + (np.arange(n) * 7 % (hi - lo + 1))


@pytest.mark.parametrize("level,deg,curved,rs", [(1, 2, False, 2), (2, 3, True, 2), (2, "mixed", True, 3), (2, 4, False, 5), (2, "low", True, 2), (1, 8, True, 3)])
def test_restriction_and_operator_parity(gpu, hiplib, oracle, level, deg, curved, rs):
    import torch
    from disco4est_amd import mesh as M
    if deg == "mixed":
        deg = _mixed(level, 2, 4)
    elif deg == "low":
        deg = _mixed(level, 1, 3)        # p = 1 elements: the overlap covers the whole element
    m, J, rst, sides, sz = _setup(level, deg, curved, rs, oracle)
    md = sz.metadata
    field = M.splitmix64_uniform(21, m.local_nodes) - 0.5
    x = torch.full((sz.nodal_size,), float("nan"), dtype=torch.float64, device=gpu)
    sz.restrict_field(_t(field, gpu), x)
    xh = x.cpu().numpy()
    want = np.zeros(sz.nodal_size)
    for s in range(md.num_subdomains):
        elem, faces, _ = md.subdomain(s)
        xr = np.concatenate([oracle.schwarz_apply_restrictor(field[m.nodal_stride[e]:m.nodal_stride[e] + (m.deg[e] + 1) ** 3], f,
                                                             int(m.deg[e]), rs) for e, f in zip(elem, faces)])
        _restricted_to_over_subdomains(oracle, m, md, xr, s, want)
    np.testing.assert_array_equal(xh, want)                      # a copy: bit-exact
    assert int(np.count_nonzero(want)) == md.restricted_nodal_size
    Ax = torch.full_like(x, float("nan"))
    sz.apply_over_subdomains(x, Ax)
    Axh = Ax.cpu().numpy()
    scale = np.abs(Axh).max()
    check = range(md.num_subdomains) if md.num_subdomains <= 8 else (0, 7, 21, 22, 42, 63)
    for s in check:
        elem, faces, _ = md.subdomain(s)
        ref = oracle.schwarz_apply_over_subdomain(elem, faces, rs, _over_subdomains_to_restricted(oracle, m, md, xh, s))
        got = _over_subdomains_to_restricted(oracle, m, md, Axh, s)
        assert np.abs(got - ref).max() <= 1e-12 * scale, (s, np.abs(got - ref).max() / scale)
    # outside the overlap the result is exactly zero (it is a restricted field)
    chk = np.zeros(sz.nodal_size)
    for s in range(md.num_subdomains):
        _restricted_to_over_subdomains(oracle, m, md, _over_subdomains_to_restricted(oracle, m, md, Axh, s), s, chk)
    np.testing.assert_array_equal(chk, Axh)
    sz.destroy()


@pytest.mark.parametrize("level,deg,rs", [(1, 3, 2), (2, "mixed", 3)])
def test_correction_parity(gpu, hiplib, oracle, level, deg, rs):
    import torch
    from disco4est_amd import mesh as M
    if deg == "mixed":
        deg = _mixed(level, 2, 4)
    m, J, rst, sides, sz = _setup(level, deg, False, rs, oracle)
    md = sz.metadata
    du = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
    sz.restrict_field(_t(M.splitmix64_uniform(31, m.local_nodes) - 0.5, gpu), du)
    du *= _t(M.splitmix64_uniform(32, sz.nodal_size) + 0.5, gpu)          # different values in every subdomain
    u0 = M.splitmix64_uniform(33, m.local_nodes)
    u = _t(u0, gpu)
    sz.add_correction(du, u)
    duh = du.cpu().numpy()
    want = u0.copy()
    off = np.concatenate([[0], np.cumsum(md.elem_nodal_size)])
    for k in range(md.num_elements):                 # ascending subdomain, then element: the reference's order of additions
        e = int(md.sub_elem[k]); p = int(m.deg[e])
        xr = oracle.schwarz_apply_restrictor(duh[off[k]:off[k + 1]], md.sub_faces[k], p, rs)
        w = oracle.schwarz_apply_weights(xr, md.sub_core_faces[k], p, rs)
        want[m.nodal_stride[e]:m.nodal_stride[e] + (p + 1) ** 3] += oracle.schwarz_apply_restrictor(w, md.sub_faces[k], p, rs, transpose=True)
    np.testing.assert_allclose(u.cpu().numpy(), want, rtol=0, atol=2e-15)      # same order of additions; weights from two table builders
    # partition of unity: the same du = 1 in every subdomain adds exactly one to u away from the domain boundary
    if level == 2:
        ones = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
        sz.restrict_field(torch.ones(m.local_nodes, dtype=torch.float64, device=gpu), ones)
        u1 = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
        sz.add_correction(ones, u1)
        interior = np.nonzero(np.all((m.ijk >= 1) & (m.ijk <= 2), axis=1))[0]
        for e in interior:
            blk = u1.cpu().numpy()[m.nodal_stride[e]:m.nodal_stride[e] + (m.deg[e] + 1) ** 3]
            np.testing.assert_allclose(blk, 1.0, rtol=0, atol=1e-14)
    sz.destroy()


@pytest.mark.parametrize("level,deg,curved,rs,iters,rtol", [(1, 3, True, 2, 6, 1e-15), (2, 2, False, 2, 60, 3e-2), (2, "mixed", True, 2, 5, 1e-15), (1, 9, True, 3, 5, 1e-15)])
def test_iterate_parity(gpu, hiplib, oracle, level, deg, curved, rs, iters, rtol):
    """one d4est_solver_schwarz_iterate: same correction, same per-subdomain iteration counts and residuals as the serial oracle"""
    from disco4est_amd import mesh as M
    if deg == "mixed":
        deg = _mixed(level, 2, 3)
    m, J, rst, sides, sz = _setup(level, deg, curved, rs, oracle, iters, 1e-15, rtol)
    u0 = M.splitmix64_uniform(41, m.local_nodes) - 0.5
    rhs = M.splitmix64_uniform(42, m.local_nodes) - 0.5
    r = rhs - oracle.apply_aij(m, J, rst, sides, u0)
    u_ref, it_ref, res_ref = oracle.schwarz_iterate(sz.metadata, u0, r, iters, 1e-15, rtol)
    u = _t(u0, gpu)
    sweeps = sz.iterate(u, _t(r, gpu))
    it, res = sz.info()
    assert _rel(u.cpu().numpy() - u0, u_ref - u0) <= 1e-9
    np.testing.assert_array_equal(it, it_ref)
    np.testing.assert_allclose(res, res_ref, rtol=1e-6, atol=1e-13 * np.abs(r).max())
    assert sweeps == min(iters, int(it_ref.max()) + 1)
    if rtol > 1e-10:
        assert it_ref.min() < iters                 # the loose tolerance really made subdomains leave their loop early
    sz.destroy()


def test_schwarz_as_a_solver(gpu, hiplib, oracle):
    """u <- u + Schwarz(rhs - A u) entirely on the device (the loop of d4est_test_schwarz_cubic_new.c:363-470): the residual
    drops at every iteration; p = 4 on 64 elements, overlap 3"""
    import torch
    from disco4est_amd import Plan, mesh as M
    m, J, rst, sides, sz = _setup(2, 4, True, 3, oracle, 60, 1e-15, 1e-8)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    u_exact = _t(M.splitmix64_uniform(51, m.local_nodes) - 0.5, gpu)
    rhs = torch.empty_like(u_exact); Au = torch.empty_like(u_exact)
    plan.apply_aij(u_exact, rhs)
    u = torch.zeros_like(u_exact)
    hist = [float(rhs.norm())]
    for _ in range(5):
        plan.apply_aij(u, Au)
        r = rhs - Au
        sz.iterate(u, r)
        plan.apply_aij(u, Au)
        hist.append(float((rhs - Au).norm()))
    assert all(b < 0.7 * a for a, b in zip(hist[:-1], hist[1:])), hist
    assert float((u - u_exact).norm()) < 0.1 * float(u_exact.norm())
    it, res = sz.info()
    assert it.min() >= 1 and it.max() <= 60 and np.all(np.isfinite(res))
    sz.destroy()


def test_schwarz_p7_subdomains(gpu, hiplib, oracle):
    """p = 7 (the wave flux kernel + stiffness_wave_eo on the subdomain plan): operator parity on sampled subdomains, one iterate"""
    import torch
    from disco4est_amd import mesh as M
    m, J, rst, sides, sz = _setup(2, 7, False, 4, oracle, 3, 1e-15, 1e-15)
    md = sz.metadata
    assert sz.nodal_size == (8 * 8 + 24 * 12 + 24 * 18 + 8 * 27) * 512
    x = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
    sz.restrict_field(_t(M.splitmix64_uniform(61, m.local_nodes) - 0.5, gpu), x)
    Ax = torch.empty_like(x)
    sz.apply_over_subdomains(x, Ax)
    xh, Axh = x.cpu().numpy(), Ax.cpu().numpy()
    scale = np.abs(Axh).max()
    for s in (0, 21):
        elem, faces, _ = md.subdomain(s)
        ref = oracle.schwarz_apply_over_subdomain(elem, faces, 4, _over_subdomains_to_restricted(oracle, m, md, xh, s))
        assert np.abs(_over_subdomains_to_restricted(oracle, m, md, Axh, s) - ref).max() <= 1e-12 * scale
    u = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
    assert sz.iterate(u, _t(M.splitmix64_uniform(62, m.local_nodes) - 0.5, gpu)) == 3
    assert bool(torch.isfinite(u).all()) and float(u.abs().max()) > 0
    sz.destroy()


@pytest.mark.parametrize("pattern,degf,rs,curved", [([0], lambda n: np.full(n, 2), 2, False), ([1, 6], lambda n: 2 + (np.arange(n) % 3), 3, True)])
def test_schwarz_on_hanging_mesh(gpu, hiplib, oracle, pattern, degf, rs, curved):
    """subdomains across hanging 1 <-> 4 faces: the subdomain plan carries the hanging-face arrays of the copies (members outside the
    subdomain are zero ghosts); operator and a whole iterate against the oracle"""
    import torch
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import Schwarz
    refine = np.zeros(8, dtype=bool)
    refine[pattern] = True
    n = M.HangingBrickMesh(1, refine, 2).n_elements
    m = M.HangingBrickMesh(1, refine, degf(n).astype(np.int32))
    mp = M.SineMap(0.04) if curved else None
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    assert np.any(sides["side_hang"] != 0)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=1)
    oracle.set_hanging(sides)
    try:
        sz = Schwarz(m, sides, J, rst, rs, 6, 1e-15, 1e-15, 10.0, 0)
        md = sz.metadata
        x = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
        sz.restrict_field(_t(M.splitmix64_uniform(71, m.local_nodes) - 0.5, gpu), x)
        Ax = torch.empty_like(x)
        sz.apply_over_subdomains(x, Ax)
        xh, Axh = x.cpu().numpy(), Ax.cpu().numpy()
        scale = np.abs(Axh).max()
        for s in range(md.num_subdomains):
            elem, faces, _ = md.subdomain(s)
            ref = oracle.schwarz_apply_over_subdomain(elem, faces, rs, _over_subdomains_to_restricted(oracle, m, md, xh, s))
            got = _over_subdomains_to_restricted(oracle, m, md, Axh, s)
            assert np.abs(got - ref).max() <= 1e-12 * scale, (s, np.abs(got - ref).max() / scale)
        u0 = M.splitmix64_uniform(72, m.local_nodes) - 0.5
        r = M.splitmix64_uniform(73, m.local_nodes) - 0.5
        u_ref, it_ref, res_ref = oracle.schwarz_iterate(md, u0, r, 6, 1e-15, 1e-15)
        u = _t(u0, gpu)
        sz.iterate(u, _t(r, gpu))
        it, res = sz.info()
        assert _rel(u.cpu().numpy() - u0, u_ref - u0) <= 1e-9
        np.testing.assert_array_equal(it, it_ref)
        np.testing.assert_allclose(res, res_ref, rtol=1e-6)
        sz.destroy()
    finally:
        oracle.set_hanging(None)


def test_multigrid_schwarz_smoother_parity(gpu, hiplib, oracle):
    """d4est_hip_schwarz_smooth = d4est_solver_multigrid_smoother_schwarz: 3 x { r = rhs - A u; Schwarz iterate }, final residual"""
    import torch
    from disco4est_amd import Plan, mesh as M
    m, J, rst, sides, sz = _setup(2, 2, True, 2, oracle, 6, 1e-15, 1e-15)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    u0 = M.splitmix64_uniform(91, m.local_nodes) - 0.5
    rhs = M.splitmix64_uniform(92, m.local_nodes) - 0.5
    u_ref, r_ref = oracle.schwarz_smoother(sz.metadata, u0, rhs, 3, 6, 1e-15, 1e-15)
    u = _t(u0, gpu); r = torch.full_like(u, float("nan"))
    sz.smooth(plan, u, _t(rhs, gpu), r, 3)
    assert _rel(u.cpu().numpy(), u_ref) <= 1e-9
    assert _rel(r.cpu().numpy(), r_ref) <= 1e-8
    assert np.linalg.norm(r_ref) < np.linalg.norm(rhs - oracle.apply_aij(m, J, rst, sides, u0))
    sz.destroy()


def test_schwarz_with_overintegration(gpu, hiplib, oracle):
    """deg_quad = deg + 1 (quadrature grid finer than the Lobatto grid): operator over subdomains and one iterate"""
    import torch
    from disco4est_amd import mesh as M
    m, J, rst, sides, sz = _setup(2, 3, True, 2, oracle, 5, 1e-15, 1e-15, deg_quad_inc=1)
    md = sz.metadata
    x = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
    sz.restrict_field(_t(M.splitmix64_uniform(95, m.local_nodes) - 0.5, gpu), x)
    Ax = torch.empty_like(x)
    sz.apply_over_subdomains(x, Ax)
    xh, Axh = x.cpu().numpy(), Ax.cpu().numpy()
    scale = np.abs(Axh).max()
    for s in (0, 21, 63):
        elem, faces, _ = md.subdomain(s)
        ref = oracle.schwarz_apply_over_subdomain(elem, faces, 2, _over_subdomains_to_restricted(oracle, m, md, xh, s))
        assert np.abs(_over_subdomains_to_restricted(oracle, m, md, Axh, s) - ref).max() <= 1e-12 * scale
    u0 = np.zeros(m.local_nodes)
    r = M.splitmix64_uniform(96, m.local_nodes) - 0.5
    u_ref, it_ref, _ = oracle.schwarz_iterate(md, u0, r, 5, 1e-15, 1e-15)
    u = _t(u0, gpu)
    sz.iterate(u, _t(r, gpu))
    assert _rel(u.cpu().numpy(), u_ref) <= 1e-9
    sz.destroy()


@pytest.mark.parametrize("level,deg,curved,rs", [(2, 3, True, 2), (1, 7, True, 2), (2, 5, False, 2)])
def test_condensed_corner_copies(gpu, hiplib, oracle, monkeypatch, level, deg, curved, rs):
    """Corner copies of conforming one-degree subdomains keep their rows of the subdomain operator as dense blocks probed from the
    matrix-free operator (csrc/d4est_hip_schwarz.hip: ensure_condensed).  With and without them: the same operator (1e-13), the
    oracle's operator (1e-12), the same CG iteration counts and the same iterate (1e-10)."""
    import torch
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import Schwarz
    m, J, rst, sides, sz = _setup(level, deg, curved, rs, oracle, 6, 1e-15, 1e-15)
    md = sz.metadata
    direct = sz.plan.face_path() != "two-phase"
    n_cond = sz.condensed_copies()
    if direct:
        # every subdomain has one corner copy per corner of its patch: 8 at level 1 (one per subdomain), 27 000-style counts above
        assert n_cond > 0 and n_cond < md.num_elements
    else:
        assert n_cond == 0          # the list-driven operator kernel belongs to the direct face path
    monkeypatch.setenv("D4EST_HIP_SCHWARZ_CONDENSE", "0")
    sz0 = Schwarz(m, sides, J, rst, rs, 6, 1e-15, 1e-15, 10.0, 0)
    assert sz0.condensed_copies() == 0
    monkeypatch.delenv("D4EST_HIP_SCHWARZ_CONDENSE")
    x = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
    sz.restrict_field(_t(M.splitmix64_uniform(71, m.local_nodes) - 0.5, gpu), x)
    Ax, Ax0 = torch.full_like(x, float("nan")), torch.full_like(x, float("nan"))
    sz.apply_over_subdomains(x, Ax)
    sz0.apply_over_subdomains(x, Ax0)
    xh, Axh, Ax0h = x.cpu().numpy(), Ax.cpu().numpy(), Ax0.cpu().numpy()
    scale = np.abs(Ax0h).max()
    assert np.isfinite(Axh).all() and np.abs(Axh - Ax0h).max() <= 1e-13 * scale
    for s in ((0, md.num_subdomains - 1) if md.num_subdomains <= 8 else (0, 21, 42, 63)):
        elem, faces, _ = md.subdomain(s)
        ref = oracle.schwarz_apply_over_subdomain(elem, faces, rs, _over_subdomains_to_restricted(oracle, m, md, xh, s))
        assert np.abs(_over_subdomains_to_restricted(oracle, m, md, Axh, s) - ref).max() <= 1e-12 * scale
    r = _t(M.splitmix64_uniform(72, m.local_nodes) - 0.5, gpu)
    u, u0 = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu), torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
    assert sz.iterate(u, r) == sz0.iterate(u0, r)
    assert _rel(u.cpu().numpy(), u0.cpu().numpy()) <= 1e-10
    it, _ = sz.info(); it0, _ = sz0.info()
    np.testing.assert_array_equal(it, it0)
    sz.destroy(); sz0.destroy()


def test_condensed_blocks_follow_the_subdomain_operator(gpu, hiplib, oracle, monkeypatch):
    """The condensed corner copies hold rows of the subdomain operator probed from the matrix-free kernels.  When that operator changes
    after the first use (here: the SIPG penalty prefactor on the subdomain plan) the blocks are probed again -- the smoother must never
    mix the new matrix-free rows with stale dense ones; with inhomogeneous boundary data on the subdomain plan (an affine operator, which
    unit-vector probing cannot represent) nothing is condensed."""
    import torch
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import Schwarz
    monkeypatch.setenv("D4EST_HIP_FACE_DIRECT", "2")
    m = M.BrickMesh(2, 3)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    sz = Schwarz(m, sides, J, rst, 2, 6, 1e-15, 1e-15, 10.0, 0)
    assert sz.condensed_copies() > 0                       # probed with prefactor 10
    x = torch.empty(sz.nodal_size, dtype=torch.float64, device=gpu)
    sz.restrict_field(_t(M.splitmix64_uniform(81, m.local_nodes) - 0.5, gpu), x)
    Ax10 = torch.full_like(x, float("nan"))
    sz.apply_over_subdomains(x, Ax10)
    # the operator of the subdomain plan changes: new mortar factors (h halved: the SIPG penalty doubles)
    from disco4est_amd.capi import _vp
    sub = dict(sz._sub_sides)
    sub["hm"] = 0.5 * np.asarray(sub["hm"]); sub["hp"] = 0.5 * np.asarray(sub["hp"])
    arrs = [np.ascontiguousarray(sub[k], dtype=np.float64) for k in ("sj", "n", "drst_m", "drst_p", "hm", "hp")]
    sz.plan.lib.d4est_hip_plan_set_mortar_geometry(sz.plan.handle, *[a.ctypes.data_as(_vp) for a in arrs], 0)
    Ax25 = torch.full_like(x, float("nan"))
    sz.apply_over_subdomains(x, Ax25)
    sides2 = dict(sides)
    sides2["hm"] = 0.5 * np.asarray(sides["hm"]); sides2["hp"] = 0.5 * np.asarray(sides["hp"])
    sz_new = Schwarz(m, sides2, J, rst, 2, 6, 1e-15, 1e-15, 10.0, 0)
    ref = torch.full_like(x, float("nan"))
    sz_new.apply_over_subdomains(x, ref)
    scale = float(ref.abs().max())
    assert float((Ax25 - ref).abs().max()) <= 1e-13 * scale
    assert float((Ax10 - ref).abs().max()) > 1e-3 * scale  # the two operators really differ
    assert sz.condensed_copies() == sz_new.condensed_copies() > 0
    # inhomogeneous Robin data on the subdomain plan (an affine operator): nothing is condensed any more
    tm = int(sz._sub_sides["total_mortar_nodes"]) if hasattr(sz, "_sub_sides") else None
    if tm is not None:
        sz.plan.set_robin_values(np.ones(tm), np.ones(tm))
        assert sz.condensed_copies() == 0
    sz.destroy(); sz_new.destroy()
