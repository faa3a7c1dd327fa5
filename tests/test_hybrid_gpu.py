"""The hybrid operator on BASELINE config 4's mesh class (mixed degrees, hanging faces): clean elements -- all six sides conforming against
a local element of the same degree, or the boundary -- through the trace-free one-kernel path of their degree bucket
(faces_direct_kernel / operator_mw_kernel over element lists), the rest through the two-phase kernels on lists
(csrc/d4est_hip_direct.hip "hybrid", tuning key 14).  Reference: d4est_laplacian_apply_aij is one path for every mesh
(src/dGMath/d4est_laplacian.c:318-417); so every check here is against the oracle of that one path, and against the all-two-phase form.
Tolerance: fp64 re-association only, rel-inf <= 1e-12 per apply."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _plateau_degrees(level, degs):
    """degree plateaus: slabs in x of equal width, one degree each (clean elements inside a slab, mixed-degree sides between slabs)"""
    from disco4est_amd import mesh as M
    ijk = M.morton_order(level)
    n = 1 << level
    return np.asarray(degs, dtype=np.int32)[(ijk[:, 0] * len(degs)) // n]


def _mesh(kind):
    from disco4est_amd import mesh as M
    if kind == "mixed_p_2_5":            # p <= 7 everywhere: single-wave direct kernels on the clean elements, flux_wave on the dirty ones
        return M.BrickMesh(2, _plateau_degrees(2, [2, 5]))
    if kind == "mixed_p_3_8_9":          # degrees above 7: operator_mw_kernel on the clean p = 8 / 9 elements, tiled MFMA kernels on the dirty ones
        return M.BrickMesh(2, _plateau_degrees(2, [3, 8, 9, 9]))
    if kind == "hanging_p4":             # one degree, one refined octant: every element is clean (hanging-aware form: big sides skipped and
                                         # left to the record kernels, small sides read the big element's sub-mortar block and export their own)
        refine = np.zeros(64, dtype=bool); refine[[21]] = True
        return M.HangingBrickMesh(2, refine, 4)
    if kind == "hanging_mixed":          # both: refined octants and two degree plateaus
        refine = np.zeros(64, dtype=bool); refine[[5, 42]] = True
        base = _plateau_degrees(2, [3, 5])
        deg = np.concatenate([np.full(8 if refine[b] else 1, base[b]) for b in range(64)]).astype(np.int32)
        return M.HangingBrickMesh(2, refine, deg)
    raise ValueError(kind)


def _plan(m, J, rst, sides, hybrid):
    from disco4est_amd import Plan
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    p.set_tuning(14, hybrid)
    p.set_geometry(J, rst)
    p.set_faces(sides, 10.0, 0)
    return p


@pytest.mark.parametrize("kind", ["mixed_p_2_5", "mixed_p_3_8_9", "hanging_p4", "hanging_mixed"])
@pytest.mark.parametrize("curved", [False, True])
def test_hybrid_operator_parity(gpu, hiplib, oracle, kind, curved):
    import torch
    from disco4est_amd import mesh as M
    m = _mesh(kind)
    mp = M.SineMap(0.04) if curved else None
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    bx = sides["bndry_xyz"]
    g = np.sin(bx[0]) + bx[1] * bx[2]
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, nthreads=8)
    res = {}
    for hybrid in (1, 0):
        plan = _plan(m, J, rst, sides, hybrid)
        plan.set_dirichlet_values(g)
        path = plan.face_path()
        assert path.startswith("hybrid") == bool(hybrid), path
        du = _t(u, gpu)
        Au = torch.full_like(du, float("nan"))
        plan.apply_aij(du, Au)
        got = Au.cpu().numpy()
        assert np.isfinite(got).all()
        assert _rel(got, ref) <= RTOL, (kind, hybrid, _rel(got, ref))
        Au2 = torch.full_like(du, float("nan"))
        plan.apply_aij(du, Au2)
        assert torch.equal(Au, Au2)       # deterministic
        res[hybrid] = got
        if hybrid:
            n_clean = int(path.split(" on ")[1].split()[0])
            if kind == "hanging_p4":
                assert n_clean == m.n_elements and "hanging-aware" in path, path
            else:
                assert 0 < n_clean < m.n_elements, path
        plan.destroy()
    assert _rel(res[1], res[0]) <= RTOL


@pytest.mark.parametrize("kind", ["hanging_p4", "hanging_mixed"])
def test_hybrid_without_the_hanging_aware_form(gpu, hiplib, oracle, kind, monkeypatch):
    """D4EST_HIP_HYBRID_NO_HANGING=1: every element with a hanging side stays with the two-phase kernels (round 4's first form)"""
    import torch
    from disco4est_amd import mesh as M
    monkeypatch.setenv("D4EST_HIP_HYBRID_NO_HANGING", "1")
    m = _mesh(kind)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    u = m.field(mp)
    ref = oracle.apply_aij(m, J, rst, sides, u, nthreads=8)
    plan = _plan(m, J, rst, sides, 1)
    path = plan.face_path()
    assert path.startswith("hybrid") and "hanging-aware" not in path, path
    assert int(path.split(" on ")[1].split()[0]) < m.n_elements
    du = _t(u, gpu); Au = torch.full_like(du, float("nan"))
    plan.apply_aij(du, Au)
    assert _rel(Au.cpu().numpy(), ref) <= RTOL
    plan.destroy()


@pytest.mark.parametrize("kind", ["mixed_p_2_5", "hanging_mixed"])
def test_hybrid_smoothers_and_lhs_term(gpu, hiplib, oracle, kind):
    """apply_lhs with the zeroth-order coefficient, 5 Chebyshev iterations and cg_eigs on the hybrid operator against the oracle"""
    import torch
    from disco4est_amd import mesh as M
    m = _mesh(kind)
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    plan = _plan(m, J, rst, sides, 1)
    assert plan.face_path().startswith("hybrid")
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=8)
    oracle.set_hanging(sides)
    coeff = 0.5 + M.splitmix64_uniform(3, m.local_nodes_quad)
    oracle.set_lhs_coefficient(coeff); oracle.set_lhs_element_blocks(None)
    plan.set_lhs_coefficient(_t(coeff, gpu))
    u = m.field(mp)
    du = _t(u, gpu); Au = torch.empty_like(du)
    plan.apply_lhs(du, Au)
    ref = oracle.apply_lhs(u)
    assert _rel(Au.cpu().numpy(), ref) <= RTOL
    rhs = M.splitmix64_uniform(5, m.local_nodes) - 0.5
    lmax = 1.1 * oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 8)[0]
    lmin = lmax / 30.0
    ref_u, ref_r = oracle.cheby_iterate(np.zeros(m.local_nodes), rhs, 5, lmin, lmax, 1)
    x = torch.zeros_like(du); r = torch.empty_like(du)
    plan.cheby_iterate(x, _t(rhs, gpu), Au, r, 5, lmin, lmax, 1)
    assert _rel(x.cpu().numpy(), ref_u) <= 1e-11 and np.abs(r.cpu().numpy() - ref_r).max() <= 1e-11 * np.abs(rhs).max()
    ref_b, ref_ucg = oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 6)
    x.zero_()
    b, _ = plan.cg_eigs(x, _t(rhs, gpu), Au, 6, 1)
    assert abs(b - ref_b) <= 1e-9 * abs(ref_b) and _rel(x.cpu().numpy(), ref_ucg) <= 1e-9
    oracle.set_lhs_coefficient(None); oracle.set_hanging(None)
    plan.destroy()


@pytest.mark.parametrize("kind", ["mixed_p_2_5", "hanging_p4"])
def test_hybrid_robin_boundary(gpu, hiplib, oracle, kind):
    import torch
    from disco4est_amd import mesh as M
    m = _mesh(kind)
    J, rst = m.geometry(None); sides = m.build_sides(None)
    rc = 0.5 + M.splitmix64_uniform(7, int(sides["total_mortar_nodes"]))
    rr = M.splitmix64_uniform(8, int(sides["total_mortar_nodes"])) - 0.5
    u = m.field(None)
    ref = oracle.apply_aij(m, J, rst, sides, u, nthreads=8, robin=(rc, rr))
    plan = _plan(m, J, rst, sides, 1)
    plan.set_robin_values(rc, rr)
    du = _t(u, gpu); Au = torch.empty_like(du)
    plan.apply_aij(du, Au)
    assert _rel(Au.cpu().numpy(), ref) <= RTOL
    plan.destroy()


@pytest.mark.parametrize("hybrid", [1, -1])
def test_config4_mesh_class_at_size(gpu, hiplib, oracle, hybrid):
    """At size: the level-4 brick with every 64th octant refined (hanging faces) AND degrees p = 3, 5, 7, 9 in plateaus four elements thick (mixed-degree sides
    between them), 4544 elements, on the hybrid operator (forced: tuning 1) and on the default path (two-phase here: several clean
    degree buckets): the oracle on 71-element shards with ghost
    elements, A(x^2 + y^2 + z^2) = M(-6) with exact Dirichlet data, symmetry, positivity, determinism"""
    import torch
    from disco4est_amd import Plan, mesh as M
    refine = np.zeros(4096, dtype=bool); refine[::64] = True
    base = _plateau_degrees(4, [3, 5, 7, 9])
    deg = np.concatenate([np.full(8 if refine[b] else 1, base[b]) for b in range(4096)]).astype(np.int32)
    mk = lambda **kw: M.HangingBrickMesh(4, refine, deg, **kw)
    m = mk()
    J, rst = m.geometry(None); sides = m.build_sides(None)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_tuning(14, hybrid)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    path = plan.face_path()
    u = m.field(None)
    du = _t(u, gpu); Au = torch.full_like(du, float("nan")); Au2 = torch.full_like(du, float("nan"))
    plan.apply_aij(du, Au); plan.apply_aij(du, Au2)
    assert torch.equal(Au, Au2)
    got = Au.cpu().numpy()
    assert np.isfinite(got).all()
    n = m.n_elements
    for first in (0, (n // 2 // 71) * 71, n - 71):
        sub = mk(first=first, count=71)
        Js, rsts = sub.geometry(None); ss = sub.build_sides(None)
        s0 = sub.global_nodal_offset
        ref = oracle.apply_aij(sub, Js, rsts, ss, np.ascontiguousarray(u[s0:s0 + sub.local_nodes]), u_ghost=sub.gather_ghost(ss, u), nthreads=8)
        assert _rel(got[s0:s0 + sub.local_nodes], ref) <= RTOL
    x, y, z = m.nodal_coords()
    bx = sides["bndry_xyz"]
    plan.set_dirichlet_values(bx[0] ** 2 + bx[1] ** 2 + bx[2] ** 2)
    dq = _t(x * x + y * y + z * z, gpu); Mrhs = torch.empty_like(dq)
    plan.apply_aij(dq, Au)
    plan.apply_mass_matrix(torch.full_like(dq, -6.0), Mrhs)
    assert (Au - Mrhs).abs().max().item() <= 5e-9 * Mrhs.abs().max().item()
    plan.set_dirichlet_values(None)
    a = _t(M.splitmix64_uniform(1, m.local_nodes), gpu); b = _t(M.splitmix64_uniform(2, m.local_nodes), gpu)
    Aa = torch.empty_like(a); Ab = torch.empty_like(a)
    plan.apply_aij(a, Aa); plan.apply_aij(b, Ab)
    s1, s2 = torch.dot(b, Aa).item(), torch.dot(a, Ab).item()
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    assert torch.dot(a, Aa).item() > 0
    assert path.startswith("hybrid") == (hybrid == 1), path
    plan.destroy()


def test_default_path_on_a_one_degree_refined_mesh_is_hybrid(gpu, hiplib, oracle):
    """the bench's hanging_level4_p7 mesh class: one degree, local refinement -- the default is the hybrid operator (one clean bucket)"""
    import torch
    from disco4est_amd import Plan, mesh as M
    refine = np.zeros(512, dtype=bool); refine[::32] = True
    m = M.HangingBrickMesh(3, refine, 7)
    J, rst = m.geometry(None); sides = m.build_sides(None)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    assert plan.face_path().startswith("hybrid"), plan.face_path()
    u = m.field(None)
    du = _t(u, gpu); Au = torch.empty_like(du)
    plan.apply_aij(du, Au)
    got = Au.cpu().numpy()
    n = m.n_elements
    for first in (0, n - 64):
        sub = M.HangingBrickMesh(3, refine, 7, first=first, count=64)
        Js, rsts = sub.geometry(None); ss = sub.build_sides(None)
        s0 = sub.global_nodal_offset
        ref = oracle.apply_aij(sub, Js, rsts, ss, np.ascontiguousarray(u[s0:s0 + sub.local_nodes]), u_ghost=sub.gather_ghost(ss, u), nthreads=8)
        assert _rel(got[s0:s0 + sub.local_nodes], ref) <= RTOL
    plan.destroy()


@pytest.mark.parametrize("iters", [4, 5])
@pytest.mark.parametrize("curved", [False, True])
def test_hanging_aware_chebyshev_fused_update(gpu, hiplib, oracle, iters, curved):
    """cheby_iterate on the hanging-aware hybrid operator: the update rides in the whole-operator kernel (elements it finishes) and in the
    record flux kernel (elements with a record side), iterates alternating between two vectors -- against the oracle's recurrence
    (Solver/d4est_solver_multigrid_smoother_cheby.c:81-176) and against the separate update kernel (tuning key 10 = 0: same roundings)"""
    import torch
    from disco4est_amd import mesh as M
    refine = np.zeros(64, dtype=bool); refine[[21, 40]] = True
    m = M.HangingBrickMesh(2, refine, 4)
    mp = M.SineMap(0.03) if curved else None
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=8)
    oracle.set_hanging(sides)
    oracle.set_lhs_coefficient(None); oracle.set_lhs_element_blocks(None)
    rhs = M.splitmix64_uniform(5, m.local_nodes) - 0.5
    u0 = M.splitmix64_uniform(6, m.local_nodes) - 0.5
    lmax = 1.1 * oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 8)[0]
    lmin = lmax / 30.0
    out = {}
    for fuse in (-1, 0):
        plan = _plan(m, J, rst, sides, -1)
        assert "hanging-aware" in plan.face_path(), plan.face_path()
        plan.set_tuning(10, fuse)
        for flag in (0, 1):
            ref_u, ref_r = oracle.cheby_iterate(u0.copy(), rhs, iters, lmin, lmax, flag)
            x = _t(u0, gpu); Au = torch.full_like(x, float("nan")); r = torch.full_like(x, float("nan"))
            plan.cheby_iterate(x, _t(rhs, gpu), Au, r, iters, lmin, lmax, flag)
            assert _rel(x.cpu().numpy(), ref_u) <= 1e-11
            assert np.abs(r.cpu().numpy() - ref_r).max() <= 1e-11 * np.abs(rhs).max()
            assert np.isfinite(Au.cpu().numpy()).all()
            out[(fuse, flag)] = (x.cpu().numpy(), r.cpu().numpy(), Au.cpu().numpy())
        plan.destroy()
    oracle.set_hanging(None)
    for flag in (0, 1):
        for a, b in zip(out[(-1, flag)], out[(0, flag)]):
            assert _rel(a, b) <= 1e-13


def test_default_path_with_one_dominant_degree(gpu, hiplib, oracle):
    """a locally refined mesh with ONE dominant degree (p = 4 but for a column of p = 3 octants): the default keeps the largest clean bucket
    (hanging-aware one-kernel path) and leaves the other degrees' elements to the two-phase lists; D4EST_HIP_HYBRID_ONE_BUCKET_ONLY=1 is
    round 4's first rule (several clean buckets: no hybrid operator)"""
    import os
    import torch
    from disco4est_amd import mesh as M
    ijk = M.morton_order(3)
    refine = np.zeros(512, dtype=bool); refine[[77, 300]] = True
    base = np.where(ijk[:, 0] < 2, 3, 4)      # a slab of p = 3 two octants thick (its outer layer is clean too: two clean buckets)
    deg = np.concatenate([np.full(8 if refine[b] else 1, base[b]) for b in range(512)]).astype(np.int32)
    m = M.HangingBrickMesh(3, refine, deg)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    u = m.field(mp)
    ref = oracle.apply_aij(m, J, rst, sides, u, nthreads=8)
    plan = _plan(m, J, rst, sides, -1)
    path = plan.face_path()
    assert path.startswith("hybrid") and "hanging-aware" in path, path
    n_clean = int(path.split(" on ")[1].split()[0]); n_dirty = int(path.split(" on ")[2].split()[0])
    assert 2 * n_clean >= m.n_elements and n_dirty > 0 and n_clean + n_dirty == m.n_elements, path
    du = _t(u, gpu); Au = torch.full_like(du, float("nan"))
    plan.apply_aij(du, Au)
    assert _rel(Au.cpu().numpy(), ref) <= RTOL
    plan.destroy()
    os.environ["D4EST_HIP_HYBRID_ONE_BUCKET_ONLY"] = "1"
    try:
        plan = _plan(m, J, rst, sides, -1)
        assert plan.face_path() == "two-phase", plan.face_path()
        Au2 = torch.full_like(du, float("nan"))
        plan.apply_aij(du, Au2)
        assert _rel(Au2.cpu().numpy(), ref) <= RTOL
        plan.destroy()
    finally:
        del os.environ["D4EST_HIP_HYBRID_ONE_BUCKET_ONLY"]


@pytest.mark.parametrize("kind", ["hanging_mixed", "hanging_p4"])
def test_hybrid_chebyshev_graph_replay(gpu, hiplib, kind):
    """tuning key 9 = 1 (cheby_iterate captured into a hipGraph and replayed) on the hybrid operator: the clean buckets' side streams join
    the capture through their fork / join events, the hanging-aware form carries the fused update -- bit-identical to the plain stream"""
    import torch
    from disco4est_amd import Plan, mesh as M
    m = _mesh(kind)
    J, rst = m.geometry(None); sides = m.build_sides(None)
    st = torch.cuda.current_stream()
    rhs = _t(M.splitmix64_uniform(5, m.local_nodes) - 0.5, gpu)
    out = {}
    for graph in (0, 1):
        p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=st)
        p.set_tuning(14, 1); p.set_tuning(9, graph)
        p.set_geometry(J, rst); p.set_faces(sides, 10.0, 0)
        assert p.face_path().startswith("hybrid")
        x = torch.zeros_like(rhs); Au = torch.empty_like(rhs); r = torch.empty_like(rhs)
        for _ in range(3):      # (the second and third call replay the captured graph)
            x.zero_(); p.cheby_iterate(x, rhs, Au, r, 5, 1.0, 40.0, 0)
        torch.cuda.synchronize()
        out[graph] = (x.clone(), r.clone())
        p.destroy()
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


@pytest.mark.parametrize("kind", ["mixed_p_2_5", "mixed_p_3_8_9"])
@pytest.mark.parametrize("iters", [4, 5])
def test_hybrid_chebyshev_fused_update_on_mixed_degrees(gpu, hiplib, oracle, kind, iters):
    """cheby_iterate on the hybrid operator of a conforming mixed-degree plan: the update rides in every clean bucket's whole-operator
    kernel (single-wave and multi-wave instances, side streams) and in the flux kernels of the dirty list, iterates alternating between
    two vectors -- against the oracle's recurrence and against the separate update kernel (tuning key 10 = 0)"""
    import torch
    from disco4est_amd import mesh as M
    m = _mesh(kind)
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=8)
    oracle.set_lhs_coefficient(None); oracle.set_lhs_element_blocks(None)
    rhs = M.splitmix64_uniform(5, m.local_nodes) - 0.5
    u0 = M.splitmix64_uniform(6, m.local_nodes) - 0.5
    lmax = 1.1 * oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 8)[0]
    lmin = lmax / 30.0
    out = {}
    for fuse in (-1, 0):
        plan = _plan(m, J, rst, sides, 1)
        assert plan.face_path().startswith("hybrid"), plan.face_path()
        plan.set_tuning(10, fuse)
        for flag in (0, 1):
            ref_u, ref_r = oracle.cheby_iterate(u0.copy(), rhs, iters, lmin, lmax, flag)
            x = _t(u0, gpu); Au = torch.full_like(x, float("nan")); r = torch.full_like(x, float("nan"))
            plan.cheby_iterate(x, _t(rhs, gpu), Au, r, iters, lmin, lmax, flag)
            assert _rel(x.cpu().numpy(), ref_u) <= 1e-11
            assert np.abs(r.cpu().numpy() - ref_r).max() <= 1e-11 * np.abs(rhs).max()
            assert np.isfinite(Au.cpu().numpy()).all()
            out[(fuse, flag)] = (x.cpu().numpy(), r.cpu().numpy(), Au.cpu().numpy())
        plan.destroy()
    for flag in (0, 1):
        for a, b in zip(out[(-1, flag)], out[(0, flag)]):
            assert _rel(a, b) <= 1e-13
