"""CPU tests of the oracle's face terms (restatement of d4est_laplacian_flux*.c) through the
reference's own oracle-free pins: consistency A(x^2+y^2+z^2) = M(-6),
symmetry A = A^T, rank-count invariance (sharded == global)."""
import numpy as np
import pytest


def _setup(level, deg, inc=0, mapping=None, first=0, count=None):
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc, first=first, count=count)
    J, rst = m.geometry(mapping)
    sides = m.build_sides(mapping)
    return m, J, rst, sides


@pytest.mark.parametrize("level,deg,inc", [(1, 2, 0), (1, 3, 1), (2, 2, 0), (1, 4, 0)])
def test_consistency_quadratic(oracle, level, deg, inc):
    """d4est_test_laplacian_consistency.c:418-426: with exact Dirichlet data, A u = M(-Laplace u) = M(-6)
    node-wise for u = x^2+y^2+z^2 (deg >= 2), here on the affine brick."""
    m, J, rst, sides = _setup(level, deg, inc)
    x, y, z = m.nodal_coords()
    u = x * x + y * y + z * z
    bx = sides["bndry_xyz"]
    g = bx[0] ** 2 + bx[1] ** 2 + bx[2] ** 2
    Au = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=10.0)
    rhs = oracle.apply_mass(m, J, np.full(m.local_nodes, -6.0))
    assert np.abs(Au - rhs).max() <= 1e-11 * max(np.abs(rhs).max(), np.abs(Au).max())


def test_mixed_p_consistency(oracle):
    from disco4est_amd import mesh as M
    deg = 2 + (np.arange(8) * 3) % 3
    m = M.BrickMesh(1, deg)
    J, rst = m.geometry(None)
    sides = m.build_sides(None)
    x, y, z = m.nodal_coords()
    u = x * x + 2 * y * y - z * z + x * y
    bx = sides["bndry_xyz"]
    g = bx[0] ** 2 + 2 * bx[1] ** 2 - bx[2] ** 2 + bx[0] * bx[1]
    Au = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g)
    rhs = oracle.apply_mass(m, J, np.full(m.local_nodes, -4.0))
    assert np.abs(Au - rhs).max() <= 1e-11 * np.abs(rhs).max()


def test_symmetry_and_definiteness(oracle):
    """d4est_test_laplacian_symmetry.c:299-312: A (homogeneous Dirichlet data) is symmetric; SIPG with a
    sufficient penalty is positive definite."""
    from disco4est_amd import mesh as M
    m, J, rst, sides = _setup(1, 2, 1, M.SineMap(0.05))
    n = m.local_nodes
    A = np.zeros((n, n))
    for c in range(n):
        e = np.zeros(n); e[c] = 1.0
        A[:, c] = oracle.apply_aij(m, J, rst, sides, e, penalty_prefactor=20.0)
    assert np.abs(A - A.T).max() <= 1e-12 * np.abs(A).max()
    ev = np.linalg.eigvalsh(0.5 * (A + A.T))
    assert ev.min() > 0


def test_rank_count_invariance(oracle):
    """d4est_test_mpi.sh: the same answer from 1 rank and from several ranks.  Shards get their off-rank
    neighbours as ghost elements (whole-element data, like d4est_ghost_data_exchange)."""
    from disco4est_amd import mesh as M
    mp = M.SineMap(0.04)
    mg, Jg, rstg, sg = _setup(1, 3, 0, mp)
    ug = mg.field(mp)
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug)
    parts = [(0, 3), (3, 3), (6, 2)]
    got = np.zeros_like(ref)
    for first, count in parts:
        m, J, rst, s = _setup(1, 3, 0, mp, first=first, count=count)
        assert len(s["ghost_global_ids"]) > 0
        u = m.field(mp)
        np.testing.assert_array_equal(u, ug[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes])
        Au = oracle.apply_aij(m, J, rst, s, u, u_ghost=m.gather_ghost(s, ug))
        got[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes] = Au
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


def test_robin_boundary(oracle):
    """BC_ROBIN (d4est_laplacian_flux_sipg.c:339-489): with coeff = rhs = 0 the boundary sides add nothing, so the operator is
    the pure-Neumann one: constants are in its null space; and the Robin term is linear in (coeff, rhs)."""
    from disco4est_amd import mesh as M
    m = M.BrickMesh(1, 3)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    tm = int(sides["total_mortar_nodes"])
    z = np.zeros(tm)
    ones = np.ones(m.local_nodes)
    A1 = oracle.apply_aij(m, J, rst, sides, ones, robin=(z, z))
    u = m.field(mp)
    scale = np.abs(oracle.apply_aij(m, J, rst, sides, u, robin=(z, z))).max()
    assert np.abs(A1).max() <= 1e-11 * scale
    c = 0.5 + M.splitmix64_uniform(3, tm)
    r = M.splitmix64_uniform(4, tm)
    a0 = oracle.apply_aij(m, J, rst, sides, u, robin=(z, z))
    a1 = oracle.apply_aij(m, J, rst, sides, u, robin=(c, r))
    a2 = oracle.apply_aij(m, J, rst, sides, u, robin=(2 * c, 2 * r))
    assert np.abs((a2 - a0) - 2 * (a1 - a0)).max() <= 1e-12 * scale
    # symmetry of the Robin operator (rhs = 0): v.A u == u.A v
    v = M.splitmix64_uniform(5, m.local_nodes)
    Au = oracle.apply_aij(m, J, rst, sides, u, robin=(c, z))
    Av = oracle.apply_aij(m, J, rst, sides, v, robin=(c, z))
    assert abs(v @ Au - u @ Av) <= 1e-11 * abs(v @ Au)


@pytest.mark.parametrize("deg,inc,mixed", [(2, 0, False), (3, 1, False), (2, 0, True)])
def test_hanging_faces_consistency_and_symmetry(oracle, deg, inc, mixed):
    """Hanging (1 <-> 4) mortars: the reference's own identities on an adapted mesh -- A(x^2+y^2+z^2) with exact Dirichlet
    data equals M(-6) on the affine brick (d4est_test_laplacian_consistency.c:418-426 runs exactly this on a randomly
    hp-refined mesh) and A = A^T (d4est_test_laplacian_symmetry.c:299-312), also on a curved map."""
    from disco4est_amd import mesh as M
    refine = np.zeros(8, dtype=bool)
    refine[[0, 5, 6]] = True
    m = M.HangingBrickMesh(1, refine, deg, deg_quad_inc=inc)
    if mixed:
        m = M.HangingBrickMesh(1, refine, deg + (np.arange(m.n_elements) * 5 % 3), deg_quad_inc=inc)
    sides = m.build_sides(None)
    assert (sides["side_hang"] == 1).sum() > 0
    J, rst = m.geometry(None)
    x, y, z = m.nodal_coords(None)
    u = x * x + y * y + z * z
    bx = sides["bndry_xyz"]
    g = bx[0] ** 2 + bx[1] ** 2 + bx[2] ** 2
    Au = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g)
    Mf = oracle.apply_mass(m, J, np.full(m.local_nodes, -6.0))
    assert np.abs(Au - Mf).max() <= 1e-12 * max(1.0, np.abs(Mf).max())
    mp = M.SineMap(0.04)
    sides = m.build_sides(mp)
    J, rst = m.geometry(mp)
    v = M.splitmix64_uniform(1, m.local_nodes)
    w = M.splitmix64_uniform(2, m.local_nodes)
    Av = oracle.apply_aij(m, J, rst, sides, v)
    Aw = oracle.apply_aij(m, J, rst, sides, w)
    assert abs(w @ Av - v @ Aw) <= 1e-12 * abs(w @ Av)
    assert v @ Av > 0
