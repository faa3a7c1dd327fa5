"""GPU parity tests of the face (SIPG mortar) path and the full operator apply_aij, through the C-ABI,
against the oracle's restatement of d4est_laplacian_apply_aij, plus the reference's own identities at size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12
# (deg + 1, deg_quad + 1) pairs the direct face kernels are instantiated for (csrc/d4est_hip_direct.hip: D4EST_HIP_DIRECT_PAIRS, one
# wavefront per element; csrc/d4est_hip_direct_mw.hip: D4EST_HIP_DIRECT_MW_SIZES, one multi-wave workgroup per element, p = 8 ... 15)
DIRECT_PAIRS = {(n, n) for n in range(2, 17)} | {(2, 3), (3, 4), (4, 5), (3, 6), (4, 6)}


def _face_path_values(plan):
    """Values of tuning key 11 to run a test with: every face path the plan supports (2 whole operator in the direct kernel, 1 direct
    face kernel + volume kernel, 0 two-phase kernels).  The default picks by size (two-phase below 768 elements), so the tests force."""
    plan.set_tuning(11, 2)
    best = plan.face_path()
    plan.set_tuning(11, -1)
    return {"direct+volume": (2, 1, 0), "direct": (1, 0), "two-phase": (0,)}[best]


def _best_face_path(deg, inc):
    """What tuning value 2 selects: the direct kernel where it is instantiated, with the volume term in it at deg_quad = deg."""
    if (deg + 1, deg + inc + 1) not in DIRECT_PAIRS:
        return "two-phase"
    return "direct+volume" if inc == 0 else "direct"


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _plan(m, J, rst, sides, prefactor=10.0, fcn=0):
    from disco4est_amd import Plan
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
    p.set_geometry(J, rst)
    p.set_faces(sides, prefactor, fcn)
    return p


@pytest.mark.parametrize("level,deg,inc,curved,fcn", [
    (1, 1, 0, True, 0), (1, 2, 0, False, 0), (1, 2, 1, True, 1), (1, 3, 0, True, 0), (2, 3, 0, True, 2),
    (1, 4, 2, True, 3), (1, 5, 0, True, 0), (1, 7, 0, False, 0), (1, 7, 0, True, 0), (1, 7, 1, True, 0),
    (1, 8, 0, True, 0), (1, 11, 0, True, 0), (0, 3, 0, True, 0), (0, 15, 0, True, 0),
    # over-integrated pairs the direct face kernel is instantiated for: (deg + 1, deg_quad + 1) = (2, 3), (4, 5), (3, 6), (4, 6)
    (1, 1, 1, True, 0), (1, 3, 1, True, 0), (1, 2, 3, True, 0), (1, 3, 2, True, 0), (1, 6, 0, True, 0),
])
def test_apply_aij_parity(gpu, hiplib, oracle, level, deg, inc, curved, fcn):
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc)
    mp = M.SineMap(0.05) if curved else None
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    bx = sides["bndry_xyz"]
    g = np.sin(bx[0]) + bx[1] * bx[2]
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=7.5, penalty_fcn=fcn, nthreads=8)
    plan = _plan(m, J, rst, sides, 7.5, fcn)
    plan.set_dirichlet_values(g)
    du = _t(u, gpu)
    dAu = torch.full_like(du, float("nan"))
    # uniform conforming plans up to deg_quad = 7 run the direct face kernel (traces formed from u in place) by default;
    # tuning key 11 = 0 selects the two-phase kernels: both are held to the oracle
    # a small mesh: up to p = 7 the default is the two-phase kernels (see d4est_hip.h, key 11); the multi-wave whole-operator kernel of
    # p = 8 ... 15 is the default at every size
    assert plan.face_path() == ("direct+volume" if (deg >= 8 and inc == 0) else "two-phase")
    vals = _face_path_values(plan)
    assert {2: "direct+volume", 1: "direct", 0: "two-phase"}[vals[0]] == _best_face_path(deg, inc)
    for direct in vals:
        plan.set_tuning(11, direct)
        assert plan.face_path() == {2: "direct+volume", 1: "direct", 0: "two-phase"}[direct]
        dAu.fill_(float("nan"))
        plan.apply_aij(du, dAu)
        got = dAu.cpu().numpy()
        assert np.isfinite(got).all()
        assert _rel(got, ref) <= RTOL
    plan.set_tuning(11, -1)
    # homogeneous operator (what apply_lhs uses) after resetting the Dirichlet data
    plan.set_dirichlet_values(None)
    plan.apply_aij(du, dAu)
    ref0 = oracle.apply_aij(m, J, rst, sides, u, penalty_prefactor=7.5, penalty_fcn=fcn, nthreads=8)
    assert _rel(dAu.cpu().numpy(), ref0) <= RTOL
    plan.destroy()


@pytest.mark.parametrize("level,deg,inc", [(1, 2, 0), (1, 3, 1), (1, 7, 0), (1, 9, 0), (0, 4, 2)])
def test_apply_aij_robin_parity(gpu, hiplib, oracle, level, deg, inc):
    """BC_ROBIN boundary sides (d4est_laplacian_flux_sipg_robin), then back to Dirichlet."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    tm = int(sides["total_mortar_nodes"])
    coeff = 0.5 + M.splitmix64_uniform(11, tm)
    rhs = M.splitmix64_uniform(12, tm) - 0.5
    coeff[::7] = 0.0  # Neumann nodes: coeff = 0 must not divide
    ref = oracle.apply_aij(m, J, rst, sides, u, penalty_prefactor=7.5, nthreads=8, robin=(coeff, rhs))
    plan = _plan(m, J, rst, sides, 7.5, 0)
    plan.set_robin_values(coeff, rhs)
    du = _t(u, gpu)
    dAu = torch.full_like(du, float("nan"))
    for direct in _face_path_values(plan):   # every face path that applies to the plan
        plan.set_tuning(11, direct)
        dAu.fill_(float("nan"))
        plan.apply_aij(du, dAu)
        assert _rel(dAu.cpu().numpy(), ref) <= RTOL
    plan.set_tuning(11, -1)
    plan.set_robin_values(None, None)
    plan.apply_aij(du, dAu)
    ref0 = oracle.apply_aij(m, J, rst, sides, u, penalty_prefactor=7.5, nthreads=8)
    assert _rel(dAu.cpu().numpy(), ref0) <= RTOL
    assert _rel(ref, ref0) > 1e-6  # the two boundary conditions really differ
    plan.destroy()


def _hanging_mesh(level, pattern, deg, inc, mixed):
    from disco4est_amd import mesh as M
    nb = 8 ** level
    refine = np.zeros(nb, dtype=bool)
    refine[np.asarray(pattern) % nb] = True
    m0 = M.HangingBrickMesh(level, refine, deg, deg_quad_inc=inc)
    if mixed:
        d = deg + (np.arange(m0.n_elements) * 7 % 3)  # degrees deg..deg+2 scattered over big and small elements
        return M.HangingBrickMesh(level, refine, d, deg_quad_inc=inc)
    return m0


@pytest.mark.parametrize("level,pattern,deg,inc,mixed,curved", [
    (1, [0], 2, 0, False, False), (1, [0, 5, 6], 2, 0, False, True), (1, [3], 3, 1, False, True), (1, [1, 2, 4, 7], 2, 0, True, True),
    (1, [0, 7], 4, 0, True, True), (2, [0, 9, 21, 42, 63], 2, 1, True, True), (1, [6], 7, 0, False, True), (1, [2, 5], 8, 0, False, True),
])
def test_apply_aij_hanging_parity(gpu, hiplib, oracle, level, pattern, deg, inc, mixed, curved):
    """Non-conforming (1 <-> 4) mortars, uniform and mixed p: full apply_aij against the oracle's restatement of the general
    d4est_laplacian_flux_interface (faces_m, faces_p in {1, 4}), Dirichlet data on, then off."""
    import torch
    from disco4est_amd import mesh as M
    m = _hanging_mesh(level, pattern, deg, inc, mixed)
    mp = M.SineMap(0.04) if curved else None
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    assert (sides["side_hang"] == 1).sum() > 0 and (sides["side_hang"] == 2).sum() == 4 * (sides["side_hang"] == 1).sum()
    u = m.field(mp)
    bx = sides["bndry_xyz"]
    g = np.sin(bx[0]) + bx[1] * bx[2]
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=7.5, nthreads=8)
    plan = _plan(m, J, rst, sides, 7.5, 0)
    plan.set_dirichlet_values(g)
    du = _t(u, gpu)
    dAu = torch.full_like(du, float("nan"))
    plan.apply_aij(du, dAu)
    got = dAu.cpu().numpy()
    assert np.isfinite(got).all()
    assert _rel(got, ref) <= RTOL
    plan.set_dirichlet_values(None)
    plan.apply_aij(du, dAu)
    ref0 = oracle.apply_aij(m, J, rst, sides, u, penalty_prefactor=7.5, nthreads=8)
    assert _rel(dAu.cpu().numpy(), ref0) <= RTOL
    # symmetry of the device operator on the hanging mesh (the reference's d4est_test_laplacian_symmetry idea)
    v = _t(M.splitmix64_uniform(9, m.local_nodes), gpu)
    Av = torch.empty_like(v); plan.apply_aij(v, Av)
    s1, s2 = torch.dot(v, dAu).item(), torch.dot(du, Av).item()
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    plan.destroy()


@pytest.mark.parametrize("base,span,inc,level", [(2, 5, 0, 2), (6, 4, 0, 1), (5, 8, 1, 1), (12, 5, 0, 1)])
def test_apply_aij_mixed_p(gpu, hiplib, oracle, base, span, inc, level):
    """p-nonconforming mortars (different degree on the two sides of a face), config-4 style: degrees within the wave-per-face
    kernels (p <= 7), straddling them (tiled MFMA kernels with per-side N, NQ) and reaching past p = 15 (generic kernels)."""
    import torch
    from disco4est_amd import mesh as M
    deg = base + (np.arange(8 ** level) * 5) % span
    m = M.BrickMesh(level, deg, deg_quad_inc=inc)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp); sides = m.build_sides(mp); u = m.field(mp)
    ref = oracle.apply_aij(m, J, rst, sides, u, nthreads=8)
    plan = _plan(m, J, rst, sides)
    du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
    plan.apply_aij(du, dAu)
    got = dAu.cpu().numpy()
    assert _rel(got, ref) <= RTOL
    for e in range(m.n_elements):
        s = m.nodal_stride[e]; n3 = (deg[e] + 1) ** 3
        assert _rel(got[s:s + n3], ref[s:s + n3]) <= 20 * RTOL


@pytest.mark.parametrize("deg,inc", [(4, 0), (3, 2), (9, 0)])
def test_traces_match_reference_chain(gpu, hiplib, oracle, deg, inc):
    """trace kernel = slicer -> project onto the mortar space -> interpolate to the mortar quadrature nodes, for u and
    the three dudr fields (d4est_laplacian_flux.c:575-815), evaluated once per side."""
    import ctypes
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(1, deg, deg_quad_inc=inc)
    J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
    plan = _plan(m, J, rst, sides)
    tr = torch.empty(plan.trace_size, dtype=torch.float64, device=gpu)
    plan.compute_face_traces(_t(u, gpu), tr)
    tr = tr.cpu().numpy()
    d = oracle.compute_dudr(m, u)
    n3 = (deg + 1) ** 3
    pq = deg + inc
    T = (pq + 1) ** 2
    I = oracle.lobatto_to_gauss(pq, pq)
    for e in range(m.n_elements):
        s0 = m.nodal_stride[e]
        for f in range(6):
            off = plan.lib.d4est_hip_plan_trace_offset(plan.handle, 6 * e + f)
            assert plan.lib.d4est_hip_plan_trace_block_len(plan.handle, 6 * e + f) == 4 * T
            fields = [u] + d
            for c in range(4):
                nodal = oracle.apply_slicer(np.ascontiguousarray(fields[c][s0:s0 + n3]), f, deg)
                mort = np.zeros(T)
                oracle.lib.oracle_apply_p_prolong(nodal.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), deg, 2, pq,
                                                  mort.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
                ref = oracle.kron_A1A2x(I, I, mort)
                got = tr[off + c * T: off + (c + 1) * T]
                assert np.abs(got - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1)


def test_sharded_equals_global(gpu, hiplib, oracle):
    """Rank-count invariance (d4est_test_mpi.sh): shards with ghost traces reproduce the single-rank operator."""
    import torch
    from disco4est_amd import mesh as M
    mp = M.SineMap(0.04)
    mg = M.BrickMesh(2, 3)
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp); ug = mg.field(mp)
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug, nthreads=8)
    got = np.zeros_like(ref)
    for first, count in [(0, 20), (20, 24), (44, 20)]:
        m = M.BrickMesh(2, 3, first=first, count=count)
        J, rst = m.geometry(mp); s = m.build_sides(mp); u = m.field(mp)
        plan = _plan(m, J, rst, s)
        assert plan.ghost_trace_size > 0
        ughost = _t(m.gather_ghost(s, ug), gpu)
        gt = torch.full((plan.ghost_trace_size,), float("nan"), dtype=torch.float64, device=gpu)
        plan.compute_ghost_traces(ughost, gt)
        du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
        plan.apply_aij(du, dAu, gt)
        got[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes] = dAu.cpu().numpy()
        # the shard also matches the oracle run on the shard with whole-element ghost data
        refs = oracle.apply_aij(m, J, rst, s, u, u_ghost=m.gather_ghost(s, ug))
        assert _rel(dAu.cpu().numpy(), refs) <= RTOL
    assert _rel(got, ref) <= RTOL


def test_consistency_and_symmetry_at_size(gpu, hiplib, oracle):
    """d4est_test_laplacian_consistency.c:418-426 and d4est_test_laplacian_symmetry.c:299-312 at level 3, p = 7
    (262 144 DoF) without an oracle: A(x^2+y^2+z^2) = M(-6) with exact Dirichlet data; v.Au = u.Av."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(3, 7)
    J, rst = m.geometry(None); sides = m.build_sides(None)
    plan = _plan(m, J, rst, sides)
    x, y, z = m.nodal_coords()
    u = x * x + y * y + z * z
    bx = sides["bndry_xyz"]
    plan.set_dirichlet_values(bx[0] ** 2 + bx[1] ** 2 + bx[2] ** 2)
    du = _t(u, gpu); Au = torch.empty_like(du); Mrhs = torch.empty_like(du)
    plan.apply_aij(du, Au)
    plan.apply_mass_matrix(torch.full_like(du, -6.0), Mrhs)
    assert (Au - Mrhs).abs().max().item() <= 1e-10 * Mrhs.abs().max().item()
    plan.set_dirichlet_values(None)
    a = _t(M.splitmix64_uniform(1, m.local_nodes), gpu); b = _t(M.splitmix64_uniform(2, m.local_nodes), gpu)
    Aa = torch.empty_like(a); Ab = torch.empty_like(a)
    plan.apply_aij(a, Aa); plan.apply_aij(b, Ab)
    s1, s2 = torch.dot(b, Aa).item(), torch.dot(a, Ab).item()
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    assert torch.dot(a, Aa).item() > 0


@pytest.mark.parametrize("kind,level,deg", [("uniform", 2, 3), ("uniform", 1, 7), ("uniform", 1, 9), ("hanging", 1, 2), ("hanging", 2, 4)])
def test_brick_geometry_on_device(gpu, hiplib, oracle, kind, level, deg):
    """d4est_hip_plan_set_geometry_brick / _set_mortar_geometry_brick (factors of the brick map generated on the device) give
    the operator of the array path, i.e. of the oracle fed with host-computed factors."""
    import torch
    from disco4est_amd import Plan, mesh as M
    ROOT = 1 << 30
    if kind == "uniform":
        m = M.BrickMesh(level, deg)
        dq = np.full(m.n_elements, ROOT >> level, dtype=np.int32)
    else:
        refine = np.zeros(8 ** level, dtype=bool)
        refine[[0, 5 % (8 ** level), (8 ** level) - 2]] = True
        m = M.HangingBrickMesh(level, refine, deg)
        dq = (m.size * (ROOT >> (level + 1))).astype(np.int32)
    J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
    g = np.sin(sides["bndry_xyz"][0]) + sides["bndry_xyz"][1]
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=6.0, nthreads=8)
    ext = [0.0, 1.0, 0.0, 1.0, 0.0, 1.0]
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry_brick(dq, float(ROOT), ext)
    plan.set_faces(sides, 6.0, 0, brick=(dq, float(ROOT), ext))
    plan.set_dirichlet_values(g)
    du = _t(u, gpu); out = torch.full_like(du, float("nan"))
    plan.apply_aij(du, out)
    assert _rel(out.cpu().numpy(), ref) <= RTOL
    # mass matrix uses the device-generated J as well
    Mu = torch.full_like(du, float("nan"))
    plan.apply_mass_matrix(du, Mu)
    assert _rel(Mu.cpu().numpy(), oracle.apply_mass(m, J, u)) <= RTOL
    plan.destroy()


def test_default_face_path_follows_the_size(gpu, hiplib, oracle):
    """Tuning key 11 left alone: the two-phase kernels below 768 elements (several wavefronts per element on a mostly empty chip), the
    one-wavefront direct kernel from there on; parity of the default on the larger mesh."""
    import torch
    from disco4est_amd import mesh as M
    paths = {}
    for level in (3, 4):
        m = M.BrickMesh(level, 1)
        J, rst = m.geometry(None)
        sides = m.build_sides(None)
        plan = _plan(m, J, rst, sides)
        paths[level] = plan.face_path()
        if level == 4:
            u = m.field(None)
            du = _t(u, gpu)
            dAu = torch.full_like(du, float("nan"))
            plan.apply_aij(du, dAu)
            ref = oracle.apply_aij(m, J, rst, sides, u, nthreads=8)
            assert _rel(dAu.cpu().numpy(), ref) <= RTOL
        plan.destroy()
    assert paths == {3: "two-phase", 4: "direct+volume"}      # 512 and 4096 elements


@pytest.mark.parametrize("level,deg,inc", [(1, 3, 0), (1, 7, 0), (1, 4, 1), (1, 5, 0), (1, 9, 0)])
def test_apply_aij_lobatto_quadrature(gpu, hiplib, oracle, level, deg, inc):
    """The reference's second quadrature type (Gauss-Lobatto points, Quadrature/d4est_quadrature_lobatto.c) through every face path:
    the interpolation to the quadrature nodes is the identity at deg_quad = deg, the weights differ."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc, quad_type=1)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    ref = oracle.apply_aij(m, J, rst, sides, u, penalty_prefactor=9.0, nthreads=8)
    plan = _plan(m, J, rst, sides, 9.0, 0)
    du = _t(u, gpu)
    dAu = torch.full_like(du, float("nan"))
    for direct in _face_path_values(plan):
        plan.set_tuning(11, direct)
        dAu.fill_(float("nan"))
        plan.apply_aij(du, dAu)
        assert _rel(dAu.cpu().numpy(), ref) <= RTOL, (direct, plan.face_path())
    plan.destroy()


def test_face_paths_agree_randomized(gpu, hiplib):
    """Every face path of a plan gives the same operator (1e-13) and the same Chebyshev iterate (1e-12) on randomly drawn cases:
    degree, over-integration, quadrature type, map amplitude, penalty function and prefactor, Dirichlet data -- level 2 (64 elements),
    no oracle in the loop (the paths are held to it one by one above)."""
    import torch
    from disco4est_amd import Plan, mesh as M
    rng = np.random.RandomState(20240607)
    pairs = sorted(DIRECT_PAIRS)
    for case in range(14):
        n, nq = pairs[rng.randint(len(pairs))]
        deg, inc = n - 1, nq - n
        qt = int(rng.randint(0, 2))
        m = M.BrickMesh(2, deg, deg_quad_inc=inc, quad_type=qt)
        mp = M.SineMap(float(rng.uniform(0.0, 0.06)))
        J, rst = m.geometry(mp)
        sides = m.build_sides(mp)
        fcn, pre = int(rng.randint(0, 4)), float(rng.uniform(2.0, 20.0))
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, qt)
        plan.set_geometry(J, rst)
        plan.set_faces(sides, pre, fcn)
        bx = sides["bndry_xyz"]
        plan.set_dirichlet_values(np.cos(bx[0] + 2 * bx[1]) * bx[2] if rng.randint(0, 2) else None)
        u = _t(M.splitmix64_uniform(100 + case, m.local_nodes) - 0.5, gpu)
        rhs = _t(M.splitmix64_uniform(200 + case, m.local_nodes) - 0.5, gpu)
        res = {}
        for key11 in _face_path_values(plan):
            plan.set_tuning(11, key11)
            Au = torch.full_like(u, float("nan"))
            plan.apply_aij(u, Au)
            uc, r = u.clone(), torch.empty_like(u)
            lam = float(torch.linalg.norm(Au) / torch.linalg.norm(u)) * 4.0
            if key11 != _face_path_values(plan)[0]:
                lam = res["lam"]
            res["lam"] = lam
            plan.cheby_iterate(uc, rhs, torch.empty_like(u), r, 3 + case % 3, lam / 30.0, lam, case % 2)
            res[key11] = (Au.cpu().numpy(), uc.cpu().numpy(), r.cpu().numpy())
        keys = [k for k in res if k != "lam"]
        assert len(keys) >= 2, (n, nq)
        ref = res[keys[-1]]          # the two-phase kernels
        for k in keys[:-1]:
            assert _rel(res[k][0], ref[0]) <= 1e-13, (case, n, nq, qt, fcn, k)
            assert _rel(res[k][1], ref[1]) <= 1e-12 and _rel(res[k][2], ref[2]) <= 1e-11, (case, n, nq, k)
        plan.destroy()


@pytest.mark.parametrize("deg", [5, 7, 8, 11, 15])
def test_stream_mode_whole_operator(gpu, hiplib, oracle, deg):
    """The stream-mode instantiations of the whole-operator kernels (faces_direct_kernel / operator_mw_kernel with VOL | 8; tuning key
    12 = 1: non-temporal metric / factor loads and A u stores; the automatic choice takes them from 320 MB per apply): same bits as
    the plain ones, the oracle's numbers, also as a Chebyshev loop."""
    import torch
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(1, deg)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp); sides = m.build_sides(mp); u = m.field(mp)
    ref = oracle.apply_aij(m, J, rst, sides, u, nthreads=8)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
    plan.set_geometry(J, rst)
    plan.set_tuning(7, 0)    # (the streamed-metric form: a brick would otherwise take the affine one, which has no metric stream)
    plan.set_tuning(11, 2)   # the whole-operator kernel whatever the size
    plan.set_faces(sides, 10.0, 0)
    assert plan.face_path() == "direct+volume"
    du = torch.from_numpy(u).to(gpu)
    outs, its = [], []
    rhs = torch.from_numpy(M.splitmix64_uniform(7, m.local_nodes) - 0.5).to(gpu)
    for key in (0, 1):
        plan.set_tuning(12, key)
        out = torch.full_like(du, float("nan"))
        plan.apply_aij(du, out)
        outs.append(out)
        uc = du.clone(); r = torch.empty_like(du); Auc = torch.empty_like(du)
        plan.cheby_iterate(uc, rhs, Auc, r, 3, 1.0, 40.0, 0)
        its.append((uc, r, Auc))
    assert torch.equal(outs[0], outs[1])
    assert all(torch.equal(x, y) for x, y in zip(its[0], its[1]))
    got = outs[1].cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    plan.destroy()


@pytest.mark.parametrize("level,pattern,deg,inc,mixed,curved,robin", [
    (1, [0], 2, 0, False, False, False), (1, [0, 5, 6], 2, 0, False, True, False), (1, [3], 3, 1, False, True, True),
    (1, [1, 2, 4, 7], 2, 0, True, True, False), (1, [0, 7], 4, 0, True, True, False), (2, [0, 9, 21, 42, 63], 2, 1, True, True, True),
    (1, [6], 7, 0, False, True, False), (2, [5], 5, 0, True, False, False), (2, [0, 63], 6, 1, False, True, False),
])
def test_hp_split_parity(gpu, hiplib, oracle, level, pattern, deg, inc, mixed, curved, robin):
    """Hanging meshes with every degree <= 7 (tuning key 13): the conforming sides through the fast conforming trace / flux kernels, the
    hanging sides through the mortar-record kernels over the elements that have one -- forced here (the automatic choice takes the
    split when at most half of the elements have a hanging side) and held to the oracle and to the all-records path (key 13 = 0);
    the traces the two kernel families leave in the trace array are the same as the record kernels' alone."""
    import torch
    from disco4est_amd import Plan, mesh as M
    m = _hanging_mesh(level, pattern, deg, inc, mixed)
    mp = M.SineMap(0.04) if curved else None
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    bx = sides["bndry_xyz"]
    g = np.sin(bx[0]) + bx[1] * bx[2]
    tm = int(sides["total_mortar_nodes"])
    rb = (0.5 + M.splitmix64_uniform(11, tm), M.splitmix64_uniform(12, tm) - 0.5) if robin else None
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=None if robin else g, penalty_prefactor=7.5, nthreads=8, robin=rb)
    du = _t(u, gpu)
    outs, traces = [], []
    for key in (0, 1):
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
        plan.set_geometry(J, rst)
        plan.set_tuning(13, key)
        plan.set_faces(sides, 7.5, 0)
        if robin:
            plan.set_robin_values(*rb)
        else:
            plan.set_dirichlet_values(g)
        out = torch.full_like(du, float("nan"))
        plan.apply_aij(du, out)
        outs.append(out.cpu().numpy())
        tr = torch.full((plan.trace_size,), float("nan"), dtype=torch.float64, device=gpu)
        plan.compute_face_traces(du, tr)
        traces.append(tr.cpu().numpy())
        plan.destroy()
    assert np.isfinite(outs[1]).all()
    assert _rel(outs[1], ref) <= RTOL and _rel(outs[0], ref) <= RTOL
    assert _rel(outs[1], outs[0]) <= 1e-13
    assert np.isfinite(traces[1]).all() and np.abs(traces[1] - traces[0]).max() <= 1e-13 * max(np.abs(traces[0]).max(), 1.0)
