"""CPU tests of the multi-tree mesh builder (disco4est_amd/forest.py) and of the oracle's face terms on faces BETWEEN trees:
p4est orientation != 0, every (f_m, f_p, orientation) triple, reorder codes 0..7, hanging faces across tree boundaries, the
reference's 7-tree cubed sphere (config 5: src/Problems/TwoPunctures/multi_options.input:62-68).  Oracle-free pins as in
test_oracle_flux.py: consistency A(x^2+y^2+z^2) = M(-6), symmetry, rank-count invariance."""
import numpy as np
import pytest

from disco4est_amd import forest as F, mesh as M


def all_triples():
    """one (rot_a, rot_b) per (f_m, f_p, orientation) triple of tree 0's glued face"""
    seen = {}
    for ra in range(24):
        for rb in range(24):
            conn = F.Connectivity.rotated_pair(ra, rb)
            fa = [f for f in range(6) if conn.tree_to_tree[0, f] == 1][0]
            code = int(conn.tree_to_face[0, fa])
            seen.setdefault((fa, code % 6, code // 6), (ra, rb))
    return seen


TRIPLES = all_triples()


def test_all_triples_reachable():
    assert len(TRIPLES) == 144


def test_reorder_code_library_vs_oracle(hiplib, oracle):
    """d4est_hip_face_reorder_code (product, host) and the oracle's verbatim restatement of dGMath/d4est_operators.c:2031-2050 +
    p4est_expand_face_transform agree on all 144 triples; all eight codes occur; (f, f^1, 0) -- faces inside a tree -- is the identity."""
    codes = set()
    for f_m in range(6):
        for f_p in range(6):
            for o in range(4):
                a = hiplib.d4est_hip_face_reorder_code(f_m, f_p, o)
                b = oracle.lib.oracle_face_reorder_code(f_m, f_p, o)
                assert a == b == F.face_reorder_code(f_m, f_p, o), (f_m, f_p, o, a, b)
                codes.add(a)
                for i in range(4):
                    assert hiplib.d4est_hip_reorient_face_order(f_m, f_p, o, i) == oracle.lib.oracle_reorient_face_order(f_m, f_p, o, i)
    assert codes == set(range(8))
    for f in range(6):
        assert hiplib.d4est_hip_face_reorder_code(f, f ^ 1, 0) == 0


def test_face_transform_is_an_involution(oracle):
    """p4est's transform across a face and the transform back compose to the identity on quadrant coordinates"""
    import ctypes
    ft = (ctypes.c_int * 9)()
    for (f_m, f_p, o) in TRIPLES:
        fwd = F.expand_face_transform(f_m, f_p, o)
        oracle.lib.oracle_expand_face_transform(f_m, f_p + 6 * o, ft)
        assert list(ft) == fwd
        back = F.expand_face_transform(f_p, f_m, o)
        for q in ([-2, 0, 4], [8, 6, 2], [0, -2, 6], [4, 8, 0], [2, 4, -2], [6, 2, 8]):
            if not (q[f_m // 2] == (-2 if f_m % 2 == 0 else 8)):
                continue
            r = F.transform_quadrant(q, 2, 8, fwd)
            assert all(0 <= v < 8 for v in r)
            # step back out of the neighbour through f_p and return
            r_out = list(r)
            r_out[f_p // 2] += 2 if f_p % 2 else -2
            q_in = list(q)
            q_in[f_m // 2] += -2 if f_m % 2 else 2
            assert F.transform_quadrant(r_out, 2, 8, back) == q_in


@pytest.mark.parametrize("refine", [None, [1, 0], [0, 1]])
def test_topology_against_geometry(refine):
    """On every triple the (+) element(s) found through p4est's transform are the geometric neighbours: mortar nodes of the two
    sides coincide after the reference's re-orientation exactly where that re-orientation is geometric (forest.
    reference_reorientation_is_consistent: all but transpose-with-one-flip seen from the lower face), and the hanging sub-face
    permutation d4est_reference_reorient_face_order is geometric on ALL triples."""
    for trip, (ra, rb) in TRIPLES.items():
        conn = F.Connectivity.rotated_pair(ra, rb)
        m = F.ForestMesh(conn, 0, 2, F.TrilinearMap(conn, M.SineMap(0.03)), refine=refine)
        s = m.build_sides()
        ok = F.reference_reorientation_is_consistent(*trip) and F.reference_reorientation_is_consistent(trip[1], trip[0], trip[2])
        assert (s["mortar_xyz_mismatch"] <= 1e-12) == ok, (trip, s["mortar_xyz_mismatch"])
        assert s["hanging_order_mismatch"] == 0
        if refine is not None:
            assert (s["side_hang"] == 1).sum() == 1 and (s["side_hang"] == 2).sum() == 4


def test_cubed_sphere_connectivity_and_map():
    """The 7-tree connectivity (numbers from the reference, tests/golden) is symmetric, uses the codes {0,1,2,3,7}, and the restated
    d4est_geometry_cubed_sphere_7tree_X is continuous across every tree face under it (level 0 and a refined level)."""
    conn = F.cubed_sphere_7tree_connectivity()
    codes = set()
    for t in range(7):
        for f in range(6):
            tp, c = int(conn.tree_to_tree[t, f]), int(conn.tree_to_face[t, f])
            if tp == t and c == f:
                assert f == 5 and t < 6            # the outer boundary is face 5 of the six wedges
                continue
            fp, o = c % 6, c // 6
            assert int(conn.tree_to_tree[tp, fp]) == t and int(conn.tree_to_face[tp, fp]) == f + 6 * o
            codes.add(F.face_reorder_code(f, fp, o))
            assert F.reference_reorientation_is_consistent(f, fp, o)
    assert codes == {0, 1, 2, 3, 7}
    mp = F.CubedSphere7Map(1.0, 3.0)
    for level, refine in ((0, None), (0, [1, 0, 0, 1, 0, 0, 1]), (1, None)):
        m = F.ForestMesh(conn, level, 3, mp, refine=refine)
        s = m.build_sides()
        assert s["mortar_xyz_mismatch"] <= 1e-12 and s["hanging_order_mismatch"] == 0
        J, _ = m.geometry()
        assert J.min() > 0
    # the chain-rule Jacobian against central differences
    xi = np.array([[0.3, 0.6, 0.2], [0.9, 0.1, 0.7]])
    for comp in (False, True):
        mp = F.CubedSphere7Map(1.0, 3.0, compactify=comp)
        for t in range(7):
            D = mp.jacobian(t, xi)
            for k in range(3):
                h = np.zeros(3); h[k] = 1e-6
                fd = (mp.x(t, xi + h) - mp.x(t, xi - h)) / 2e-6
                assert np.abs(D[:, :, k] - fd).max() <= 1e-8


def _consistency(oracle, m, tol):
    J, rst = m.geometry()
    sides = m.build_sides()
    x, y, z = m.nodal_coords()
    u = x * x + y * y + z * z
    bx = sides["bndry_xyz"]
    g = bx[0] ** 2 + bx[1] ** 2 + bx[2] ** 2
    Au = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g)
    rhs = oracle.apply_mass(m, J, np.full(m.local_nodes, -6.0))
    err = np.abs(Au - rhs).max() / max(np.abs(rhs).max(), np.abs(Au).max())
    assert err <= tol, err
    return sides


def _symmetry(oracle, m, tol=1e-12):
    J, rst = m.geometry()
    sides = m.build_sides()
    v = M.splitmix64_uniform(1, m.local_nodes)
    w = M.splitmix64_uniform(2, m.local_nodes)
    Av = oracle.apply_aij(m, J, rst, sides, v)
    Aw = oracle.apply_aij(m, J, rst, sides, w)
    assert abs(w @ Av - v @ Aw) <= tol * abs(w @ Av), abs(w @ Av - v @ Aw) / abs(w @ Av)
    assert v @ Av > 0


def _pick(codes_wanted):
    """triples (with geometric re-orientation on both sides) covering the wanted reorder codes from both sides"""
    out = []
    have = set()
    for trip, rots in sorted(TRIPLES.items()):
        rev = (trip[1], trip[0], trip[2])
        if not (F.reference_reorientation_is_consistent(*trip) and F.reference_reorientation_is_consistent(*rev)):
            continue
        c = (F.face_reorder_code(*trip), F.face_reorder_code(*rev))     # no library load at collection time
        if c[0] in codes_wanted and c not in have:
            have.add(c)
            out.append((trip, rots))
    return out


@pytest.mark.parametrize("trip,rots", _pick(set(range(8))))
def test_oracle_identities_across_oriented_faces(oracle, trip, rots):
    """The reference's two identities (d4est_test_laplacian_consistency.c:418-426, d4est_test_laplacian_symmetry.c:299-312) on two
    trees glued with orientation != 0: affine trees give A(x^2+y^2+z^2) = M(-6) exactly -- a flipped or transposed (+) trace breaks
    it -- conforming, mixed p, and with a hanging face on either side of the tree boundary; symmetry on the warped (curved) pair."""
    conn = F.Connectivity.rotated_pair(*rots)
    _consistency(oracle, F.ForestMesh(conn, 0, 2, F.TrilinearMap(conn)), 1e-11)
    _consistency(oracle, F.ForestMesh(conn, 0, [2, 4], F.TrilinearMap(conn), deg_quad_inc=1), 1e-11)
    for refine in ([1, 0], [0, 1]):
        m = F.ForestMesh(conn, 0, 2, F.TrilinearMap(conn), refine=refine)
        m = F.ForestMesh(conn, 0, 2 + (np.arange(m.global_elements) * 5) % 3, F.TrilinearMap(conn), refine=refine)
        s = _consistency(oracle, m, 1e-11)
        assert (s["side_hang"] == 1).sum() == 1 and s["side_orientation"].max() == trip[2]
    warped = F.TrilinearMap(conn, M.SineMap(0.04))
    _symmetry(oracle, F.ForestMesh(conn, 0, 3, warped))
    _symmetry(oracle, F.ForestMesh(conn, 0, 2, warped, refine=[0, 1], deg_quad_inc=1))


def test_oracle_on_inconsistent_triples_is_still_linear(oracle):
    """Where the reference's re-orientation is not geometric (forest.reference_reorientation_is_consistent) the operator is what
    the reference computes, not a consistent discretisation: it stays linear and the oracle runs, but consistency fails there --
    pinned so that the engine's parity with the reference on those faces is a deliberate statement (tests/test_forest_gpu.py)."""
    trip = next(t for t in sorted(TRIPLES) if not F.reference_reorientation_is_consistent(*t))
    conn = F.Connectivity.rotated_pair(*TRIPLES[trip])
    m = F.ForestMesh(conn, 0, 2, F.TrilinearMap(conn))
    with pytest.raises(AssertionError):
        _consistency(oracle, m, 1e-11)


def test_cubed_sphere_oracle_symmetry_and_convergence(oracle):
    """7-tree cubed sphere, curved wedges: A = A^T, positive; the consistency error of u = x^2+y^2+z^2 falls spectrally with p
    (exact only on affine elements)."""
    conn = F.cubed_sphere_7tree_connectivity()
    mp = F.CubedSphere7Map(1.0, 2.0)
    _symmetry(oracle, F.ForestMesh(conn, 0, 3, mp), 1e-11)
    _symmetry(oracle, F.ForestMesh(conn, 0, 2, mp, refine=[1, 0, 0, 0, 0, 0, 1]), 1e-11)
    errs = []
    for p in (3, 6, 9):
        m = F.ForestMesh(conn, 0, p, mp, deg_quad_inc=2)
        J, rst = m.geometry()
        sides = m.build_sides()
        x, y, z = m.nodal_coords()
        bx = sides["bndry_xyz"]
        Au = oracle.apply_aij(m, J, rst, sides, x * x + y * y + z * z, bndry_lobatto=bx[0] ** 2 + bx[1] ** 2 + bx[2] ** 2)
        rhs = oracle.apply_mass(m, J, np.full(m.local_nodes, -6.0))
        # weak form: compare integrated against the (coarse) test space through the mass-weighted residual norm
        errs.append(np.abs(Au - rhs).sum() / np.abs(rhs).sum())
    assert errs[1] < 0.05 * errs[0] and errs[2] < 0.05 * errs[1], errs


def test_rank_count_invariance_through_oriented_faces(oracle):
    """d4est_test_mpi.sh on a multi-tree mesh: partition boundaries through faces with orientation != 0 (the cubed sphere's inner
    cube against its wedges, codes 1, 2, 3, 7), conforming and with hanging faces across ranks."""
    conn = F.cubed_sphere_7tree_connectivity()
    mp = F.CubedSphere7Map(1.0, 2.0)
    for refine, parts in ((None, [(0, 3), (3, 2), (5, 2)]), ([0, 0, 1, 0, 0, 0, 1], [(0, 6), (6, 9), (15, 6)])):
        mg = F.ForestMesh(conn, 0, 3, mp, refine=refine)
        assert mg.global_elements == sum(c for _, c in parts)
        Jg, rstg = mg.geometry()
        sg = mg.build_sides()
        ug = mg.field()
        ref = oracle.apply_aij(mg, Jg, rstg, sg, ug)
        got = np.zeros_like(ref)
        n_oriented_ghost = 0
        for first, count in parts:
            m = F.ForestMesh(conn, 0, 3, mp, refine=refine, first=first, count=count)
            J, rst = m.geometry()
            s = m.build_sides()
            n_oriented_ghost += int(((s["side_nbr"] <= -2) & (s["side_reorder"] != 0)).sum())
            u = m.field()
            np.testing.assert_array_equal(u, ug[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes])
            Au = oracle.apply_aij(m, J, rst, s, u, u_ghost=m.gather_ghost(s, ug))
            got[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes] = Au
        assert n_oriented_ghost > 0
        assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


def test_cubed_sphere_jacobian_chain_rule_vs_complex_step(oracle):
    """The chain-rule Jacobian of the 7-tree cubed-sphere map (forest.CubedSphere7Map.jacobian here, d4est_hip_maps.h on the device)
    against the oracle's complex-step derivative of the reference's X (src/Geometry/d4est_geometry_cubed_sphere.c:498-580), and X itself"""
    import ctypes
    dp = ctypes.POINTER(ctypes.c_double)
    xi = np.array([[0.3, 0.6, 0.2], [0.9, 0.1, 0.7], [0.5, 0.5, 1.0], [0.0, 1.0, 0.0]])
    for comp in (0, 1):
        mp = F.CubedSphere7Map(1.0, 3.0, compactify=bool(comp))
        for t in range(7):
            D = mp.jacobian(t, xi)
            X = mp.x(t, xi)
            for k in range(xi.shape[0]):
                tc = np.ascontiguousarray(xi[k])
                out, x = np.zeros(9), np.zeros(3)
                oracle.lib.oracle_cubed_sphere_7tree_DX(t, ctypes.c_double(1.0), ctypes.c_double(3.0), comp, tc.ctypes.data_as(dp), out.ctypes.data_as(dp))
                oracle.lib.oracle_cubed_sphere_7tree_X(t, ctypes.c_double(1.0), ctypes.c_double(3.0), comp, tc.ctypes.data_as(dp), x.ctypes.data_as(dp))
                assert np.abs(D[k] - out.reshape(3, 3)).max() <= 2e-14 * max(1.0, np.abs(out).max())
                assert np.abs(X[k] - x).max() <= 1e-15 * max(1.0, np.abs(x).max())


def _sides_from_c(hiplib, m):
    """d4est_hip_build_sides on the quadrant list of a ForestMesh shard (ghost quadrants = the shard's ghost elements)"""
    import ctypes
    s_py = m.build_sides()
    tree, q, dq = m.cells()
    gt, gq, gd = m.cells(s_py["ghost_global_ids"])
    ne = m.n_elements
    I = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    ins = [I(m.conn.tree_to_tree), I(m.conn.tree_to_face), I(tree), I(q), I(dq), I(m.deg), I(m.deg_quad), I(gt), I(gq), I(gd), I(s_py["ghost_deg_quad"])]
    out = {k: np.zeros(6 * ne, dtype=np.int32) for k in ("side_nbr", "side_nbr_face", "side_reorder", "side_orientation", "side_hang", "side_sub",
                                                         "side_mortar_stride", "side_bndry_stride")}
    out["side_nbr4"] = np.zeros(24 * ne, dtype=np.int32)
    tm, tb = ctypes.c_int(0), ctypes.c_int(0)
    hang = hiplib.d4est_hip_build_sides(m.conn.num_trees, vp(ins[0]), vp(ins[1]), m.nf, ne, vp(ins[2]), vp(ins[3]), vp(ins[4]), vp(ins[5]), vp(ins[6]),
                                        len(gd), vp(ins[7]), vp(ins[8]), vp(ins[9]), vp(ins[10]),
                                        vp(out["side_nbr"]), vp(out["side_nbr_face"]), vp(out["side_reorder"]), vp(out["side_orientation"]),
                                        vp(out["side_hang"]), vp(out["side_sub"]), vp(out["side_nbr4"]), vp(out["side_mortar_stride"]),
                                        vp(out["side_bndry_stride"]), ctypes.byref(tm), ctypes.byref(tb))
    return s_py, out, hang, tm.value, tb.value


def test_c_side_list_builder_matches_the_python_one(hiplib):
    """d4est_hip_build_sides (the host-side replacement of the p4est_iterate face walk for hosts without p4est) reproduces
    forest.ForestMesh.build_sides array for array: cubed sphere conforming / hanging / mixed p / shards with ghosts, rotated pairs"""
    conn = F.cubed_sphere_7tree_connectivity()
    mp = F.CubedSphere7Map(1.0, 2.0)
    cases = [F.ForestMesh(conn, 1, 2, mp), F.ForestMesh(conn, 0, 2 + np.arange(7) % 3, mp, deg_quad_inc=1),
             F.ForestMesh(conn, 0, 3, mp, refine=[1, 0, 0, 1, 0, 0, 1]), F.ForestMesh(conn, 1, 2, mp, first=10, count=30),
             F.ForestMesh(conn, 0, 2, mp, refine=[0, 0, 1, 0, 0, 0, 1], first=4, count=9)]
    for trip in [(0, 0, 1), (1, 4, 2), (2, 5, 3), (3, 3, 0), (5, 0, 1)]:
        c2 = F.Connectivity.rotated_pair(*TRIPLES[trip])
        cases += [F.ForestMesh(c2, 1, 2, F.TrilinearMap(c2)), F.ForestMesh(c2, 0, 2, F.TrilinearMap(c2), refine=[1, 0]),
                  F.ForestMesh(c2, 0, 2, F.TrilinearMap(c2), refine=[0, 1])]
    for m in cases:
        s_py, s_c, hang, tm, tb = _sides_from_c(hiplib, m)
        assert hang == int(m.has_hanging())
        assert tm == s_py["total_mortar_nodes"] and tb == s_py["total_bndry_nodes"]
        for k in ("side_nbr", "side_nbr_face", "side_reorder", "side_mortar_stride", "side_bndry_stride"):
            np.testing.assert_array_equal(s_c[k], s_py[k], err_msg=k)
        if hang:
            for k in ("side_hang", "side_sub", "side_nbr4", "side_orientation"):
                np.testing.assert_array_equal(s_c[k], s_py[k], err_msg=k)
        else:
            assert not s_c["side_hang"].any()
