"""Pins of the oracle that do not share its code: every term of the face path, the Chebyshev recurrence
and the Schwarz hat weights against independent numpy restatements built from the reference's formulas and from the reference's
own tabulated nodes (tests/golden/reference_nodes_weights.json).  tests/dense_sipg.py holds the dense operator."""
import json
import os

import numpy as np
import pytest

from disco4est_amd import forest as F, mesh as M
from tests import dense_sipg as DS

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


def test_nodes_and_weights_against_the_reference_tables(hiplib, oracle):
    """d4est_hip_table and the oracle reproduce the reference's tabulated Gauss / Gauss-Lobatto abscissas and weights
    (dGMath/GL_and_GLL_nodes_and_weights.h, numbers extracted by tests/golden/make_reference_tables.py) to 1e-15, n = 2..20 --
    except ONE entry: the reference's Lobatto table for n = 12 (p = 11) carries a digit slip, +-0.6328761530318697 where the root of
    P_11' is 0.63287615303186067766..., a 9e-15 error that both the engine and the oracle decline to copy."""
    from disco4est_amd import capi
    d = json.load(open(os.path.join(GOLD, "reference_nodes_weights.json")))
    for kind, tx, tw, of in (("gauss", "gauss_nodes", "gauss_weights", oracle.gauss), ("lobatto", "lobatto_nodes", "lobatto_weights", oracle.lobatto)):
        for n in range(2, 21):
            x, w = np.array(d[kind][str(n)]["x"]), np.array(d[kind][str(n)]["w"])
            ox, ow = of(n - 1)
            ex = np.abs(capi.table(tx, n - 1) - x)
            assert np.abs(capi.table(tw, n - 1) - w).max() <= 2e-15 and np.abs(ow - w).max() <= 2e-15
            if kind == "lobatto" and n == 12:
                assert sorted(np.nonzero(ex > 2e-15)[0].tolist()) == [3, 8] and ex.max() < 1e-14
                assert abs(abs(capi.table(tx, 11)[3]) - 0.6328761530318606776624) < 1e-15
                continue
            assert ex.max() <= 2e-15 and np.abs(ox - x).max() <= 2e-15, (kind, n)


def _dense(m, J, rst, sides, fcn=0, pref=10.0):
    from disco4est_amd import capi
    lib = capi.load_library()
    return DS.DenseLaplacian(m, J, rst, sides, lambda a, b, c, d: int(lib.d4est_hip_reorient_face_order(a, b, c, d)), pref, fcn)


@pytest.mark.parametrize("fcn", [0, 1, 2, 3])
@pytest.mark.parametrize("quad_type", [0, 1])
def test_face_terms_conforming_mixed_p(oracle, fcn, quad_type):
    """conforming mortars between different degrees, curved map, over-integration, Dirichlet data, all four penalty functions"""
    deg = 2 + (np.arange(8) * 3) % 3
    m = M.BrickMesh(1, deg, deg_quad_inc=1, quad_type=quad_type)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    bx = sides["bndry_xyz"]
    g = np.sin(bx[0]) + bx[1] * bx[2]
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=7.5, penalty_fcn=fcn)
    got = _dense(m, J, rst, sides, fcn, 7.5).apply(u, g=g)
    assert _rel(got, ref) <= 1e-12


def test_face_terms_robin(oracle):
    m = M.BrickMesh(1, 3)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    tm = int(sides["total_mortar_nodes"])
    coeff = 0.5 + M.splitmix64_uniform(11, tm)
    rhs = M.splitmix64_uniform(12, tm) - 0.5
    ref = oracle.apply_aij(m, J, rst, sides, u, robin=(coeff, rhs))
    got = _dense(m, J, rst, sides).apply(u, robin=(coeff, rhs))
    assert _rel(got, ref) <= 1e-12


@pytest.mark.parametrize("inc", [0, 1])
def test_face_terms_hanging_mixed_p(oracle, inc):
    """1 <-> 4 mortars: hp-prolongation onto the children, the 1/2 on the big side's gradient and term 2"""
    refine = np.zeros(8, dtype=bool); refine[[0, 5]] = True
    m0 = M.HangingBrickMesh(1, refine, 2)
    m = M.HangingBrickMesh(1, refine, 2 + (np.arange(m0.n_elements) * 5) % 3, deg_quad_inc=inc)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    u = m.field(mp)
    ref = oracle.apply_aij(m, J, rst, sides, u)
    got = _dense(m, J, rst, sides).apply(u)
    assert _rel(got, ref) <= 1e-12


def test_face_terms_ghost_shard(oracle):
    m = M.BrickMesh(1, 3, first=2, count=4)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    ug = M.BrickMesh(1, 3).field(mp)
    u = m.field(mp)
    gh = m.gather_ghost(sides, ug)
    ref = oracle.apply_aij(m, J, rst, sides, u, u_ghost=gh)
    got = _dense(m, J, rst, sides).apply(u, u_ghost=gh)
    assert _rel(got, ref) <= 1e-12


def test_face_terms_between_trees(oracle):
    """re-oriented (+) traces: a sample of (f_m, f_p, orientation) triples covering all eight codes, geometric and not, conforming
    and with the hanging face on either side of the tree boundary -- the dense operator applies the reference's flip / transpose
    matrix and its sub-face permutation literally"""
    from tests.test_forest import TRIPLES
    seen = set()
    for trip, rots in sorted(TRIPLES.items()):
        code = F.face_reorder_code(*trip)
        key = (code, F.reference_reorientation_is_consistent(*trip))
        if key in seen:
            continue
        seen.add(key)
        conn = F.Connectivity.rotated_pair(*rots)
        for refine in (None, [1, 0], [0, 1]):
            m = F.ForestMesh(conn, 0, [2, 3] if refine is None else 2, F.TrilinearMap(conn, M.SineMap(0.03)), refine=refine)
            J, rst = m.geometry()
            sides = m.build_sides()
            u = m.field()
            ref = oracle.apply_aij(m, J, rst, sides, u)
            got = _dense(m, J, rst, sides).apply(u)
            assert _rel(got, ref) <= 1e-12, (trip, refine)
    assert {c for c, _ in seen} == set(range(8))


def test_chebyshev_recurrence(oracle):
    """d4est_solver_multigrid_smoother_cheby_iterate_aux (src/Solver/d4est_solver_multigrid_smoother_cheby.c:104-154) restated in
    numpy around the operator: d = (lmax+lmin)/2, c = (lmax-lmin)/2; alpha_0 = 1/d, alpha_1 = 2d/(2d^2-c^2),
    alpha_i = 1/(d - alpha_{i-1} c^2/4); beta = alpha d - 1; r = alpha (rhs - A u); p = r + beta p; u += p."""
    m = M.BrickMesh(1, 3)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=4)
    rhs = M.splitmix64_uniform(5, m.local_nodes) - 0.5
    u0 = M.splitmix64_uniform(6, m.local_nodes)
    lmin, lmax = 7.0, 210.0
    for at_end in (0, 1):
        u_ref, r_ref = oracle.cheby_iterate(u0, rhs, 6, lmin, lmax, residual_at_end=at_end)
        d, c = 0.5 * (lmax + lmin), 0.5 * (lmax - lmin)
        u, p, alpha = u0.copy(), np.zeros_like(u0), 0.0
        for i in range(6):
            r = rhs - oracle.apply_lhs(u)
            alpha = 1.0 / d if i == 0 else (2.0 * d / (2 * d * d - c * c) if i == 1 else 1.0 / (d - alpha * c * c / 4.0))
            beta = alpha * d - 1.0
            r = alpha * r
            p = r + beta * p
            u = u + p
        if at_end:
            r = rhs - oracle.apply_lhs(u)
        assert _rel(u, u_ref) <= 1e-13 and _rel(r, r_ref) <= 1e-12


def test_schwarz_hat_weights(oracle):
    """d4est_solver_schwarz_operators_build_schwarz_weights_1d (src/Solver/d4est_solver_schwarz_operators.c:7-40, :78-105):
    w(r) = (phi((r+1)/d0) - phi((r-1)/d0))/2, phi the quintic (15 r - 10 r^3 + 3 r^5)/8 clamped to sign(r) outside [-1,1],
    d0 = 1 - r[deg + 1 - restricted_size]; left / right neighbour weights at r -+ 2, then the core's."""
    def phi(r):
        return np.where(np.abs(r) > 1, np.sign(r), (15 * r - 10 * r ** 3 + 3 * r ** 5) / 8.0)

    for deg in (2, 3, 5, 7):
        x = DS.lobatto(deg)[0]
        for rs in range(2, deg + 2):
            d0 = 1.0 - x[deg + 1 - rs]
            w = lambda r: 0.5 * (phi((r + 1) / d0) - phi((r - 1) / d0))
            want = np.concatenate([w(x[deg + 1 - rs:] - 2), w(x[:rs] + 2), w(x)])
            got = oracle.schwarz_weights_1d(deg, rs)
            assert np.abs(got - want).max() <= 1e-14
