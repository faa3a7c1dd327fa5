"""CPU tests of the additive-Schwarz restatement (oracle/d4est_oracle_schwarz.c) and of the host metadata builder
(disco4est_amd/schwarz.py).  Pins used, all taken from the reference's own Schwarz tests:
  * the restrictor picks nodes: restricting f(r) gives f at the restricted nodes, and its transpose pads with zeros
    (src/Tests/Unit/d4est_test_schwarz_operators.c:84-95, :238-243);
  * the hat weights tabulated by d4est_test_schwarz_operators.c:129-175 form a partition of unity across neighbouring subdomains;
  * repeated Schwarz iterations drive u to the solution of the Poisson problem (src/Tests/Unit/d4est_test_schwarz_cubic_new.c:363-470).
"""
import numpy as np
import pytest


def test_restrictor_picks_nodes(oracle):
    deg, rs = 3, 3
    x, _ = oracle.lobatto(deg)
    R = oracle.schwarz_restrictor_1d(deg, rs)
    np.testing.assert_array_equal(R[0] @ x, x[:rs])
    np.testing.assert_array_equal(R[1] @ x, x[deg + 1 - rs:])
    # 3-D: faces {0, 3, 4}: first rs nodes in x, last rs in y, first rs in z (d4est_test_schwarz_operators.c:42-52)
    N = deg + 1
    r3 = [np.tile(x, N * N), np.tile(np.repeat(x, N), N), np.repeat(x, N * N)]
    f = np.exp(r3[0]) * np.exp(r3[1]) * np.exp(r3[2])
    faces = [0, 3, 4]
    res_f = oracle.schwarz_apply_restrictor(f, faces, deg, rs)
    res_r = [oracle.schwarz_apply_restrictor(c, faces, deg, rs) for c in r3]
    assert res_f.size == rs ** 3
    np.testing.assert_allclose(res_f, np.exp(res_r[0]) * np.exp(res_r[1]) * np.exp(res_r[2]), rtol=0, atol=1e-12)
    assert res_r[0].max() == x[rs - 1] and res_r[1].min() == x[N - rs] and res_r[2].max() == x[rs - 1]
    back = oracle.schwarz_apply_restrictor(res_f, faces, deg, rs, transpose=True)
    keep = (r3[0] <= x[rs - 1]) & (r3[1] >= x[N - rs]) & (r3[2] <= x[rs - 1])
    np.testing.assert_array_equal(back[keep], f[keep])
    assert np.all(back[~keep] == 0.0)
    # a face element keeps whole lines in the other directions
    assert oracle.schwarz_apply_restrictor(f, [1, -1, -1], deg, 2).size == 2 * N * N


@pytest.mark.parametrize("deg,rs", [(2, 2), (3, 2), (4, 2), (5, 2), (6, 2), (6, 3), (6, 4), (7, 8)])
def test_weights_partition_of_unity(oracle, deg, rs):
    """A node of an element is weighted as core of its own subdomain, as the left element of the subdomain to its right and as the
    right element of the subdomain to its left; the three hats sum to one."""
    w = oracle.schwarz_weights_1d(deg, rs)
    N = deg + 1
    left, right, core = w[:rs], w[rs:2 * rs], w[2 * rs:]
    total = core.copy()
    total[N - rs:] += left        # this element seen from the subdomain centred on its right neighbour
    total[:rs] += right           # ... and from the one centred on its left neighbour
    np.testing.assert_allclose(total, 1.0, rtol=0, atol=1e-14)
    assert core.max() <= 1.0 + 1e-15 and w.min() >= -1e-15
    np.testing.assert_allclose(core, core[::-1], atol=1e-15)
    np.testing.assert_allclose(left, right[::-1], atol=1e-15)
    # 3-D weights are the tensor product; core_faces {0,-1,-1}: the element sits to the left of the core in x
    x = np.ones(rs * N * N)
    out = oracle.schwarz_apply_weights(x, [0, -1, -1], deg, rs)
    np.testing.assert_allclose(out.reshape(N, N, rs), core[:, None, None] * core[None, :, None] * left[None, None, :], atol=1e-15)


def _brick(level, deg):
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg)
    J, rst = m.geometry(None)
    sides = m.build_sides(None)
    return m, J, rst, sides


def test_metadata_of_uniform_brick():
    from disco4est_amd.schwarz import SchwarzMetadata
    m, _, _, sides = _brick(2, 3)
    md = SchwarzMetadata(m, sides, 2)
    assert md.num_subdomains == 64
    counts = np.diff(md.sub_first)
    n_bnd = ((m.ijk == 0) | (m.ijk == 3)).sum(axis=1)            # how many domain faces the core touches
    np.testing.assert_array_equal(counts, np.array([27, 18, 12, 8])[n_bnd])
    for s in (0, 21, 42, 63):
        elem, faces, core_faces = md.subdomain(s)
        assert np.all(np.diff(elem) > 0)                          # sorted by Morton id, no duplicates
        for e, f, cf in zip(elem, faces, core_faces):
            off = m.ijk[e] - m.ijk[s]
            assert np.abs(off).max() <= 1
            want = sorted(2 * d + (0 if off[d] > 0 else 1) for d in range(3) if off[d] != 0)
            assert sorted(int(v) for v in f if v >= 0) == want
            assert [int(v) for v in cf] == [(int(v) ^ 1) if v >= 0 else -1 for v in f]
            assert (e == s) == bool(np.all(f == -1))
    # sizes as d4est_solver_schwarz_metadata.c:497-520 computes them
    k = np.arange(md.num_elements)
    nrestr = (md.sub_faces >= 0).sum(axis=1)
    np.testing.assert_array_equal(md.elem_restricted_nodal_size, 2 ** nrestr * 4 ** (3 - nrestr))
    assert md.nodal_size == md.num_elements * 64 and k.size == counts.sum()
    with pytest.raises(ValueError):
        SchwarzMetadata(m, sides, 5)
    with pytest.raises(ValueError):
        SchwarzMetadata(m, sides, 0)
    with pytest.raises(ValueError):
        SchwarzMetadata(m, sides, 1)      # zero-width ramp: the reference's weights are 0/0 at the element ends


def test_subdomain_operator_is_symmetric_positive(oracle):
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import SchwarzMetadata
    m, J, rst, sides = _brick(1, 2)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=1)
    md = SchwarzMetadata(m, sides, 2)
    elem, faces, _ = md.subdomain(3)
    n = int(md.elem_restricted_nodal_size[md.sub_first[3]:md.sub_first[4]].sum())
    x = M.splitmix64_uniform(5, n) - 0.5
    y = M.splitmix64_uniform(6, n) - 0.5
    Ax = oracle.schwarz_apply_over_subdomain(elem, faces, 2, x)
    Ay = oracle.schwarz_apply_over_subdomain(elem, faces, 2, y)
    assert abs(y @ Ax - x @ Ay) <= 1e-12 * abs(y @ Ax)
    assert x @ Ax > 0 and y @ Ay > 0


@pytest.mark.parametrize("level,deg,rs,outer", [(1, 2, 2, 4), (2, 2, 2, 2)])
def test_schwarz_iterations_converge(oracle, level, deg, rs, outer):
    """u <- u + Schwarz(rhs - A u) as in d4est_test_schwarz_cubic_new.c: the residual falls at every iteration"""
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import SchwarzMetadata
    m, J, rst, sides = _brick(level, deg)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=1)
    md = SchwarzMetadata(m, sides, rs)
    u_exact = M.splitmix64_uniform(3, m.local_nodes) - 0.5
    rhs = oracle.apply_aij(m, J, rst, sides, u_exact)
    u = np.zeros(m.local_nodes)
    hist = [np.linalg.norm(rhs)]
    for _ in range(outer):
        r = rhs - oracle.apply_aij(m, J, rst, sides, u)
        u, it, res = oracle.schwarz_iterate(md, u, r, 200, 1e-15, 1e-10)
        assert it.max() < 200 and res.max() < 1e-8 * hist[0]       # every subdomain CG met its tolerance
        hist.append(np.linalg.norm(rhs - oracle.apply_aij(m, J, rst, sides, u)))
    assert all(b < 0.8 * a for a, b in zip(hist[:-1], hist[1:])), hist
    assert np.linalg.norm(u - u_exact) < (0.3 if outer >= 4 else 0.6) * np.linalg.norm(u_exact)


def test_metadata_geometric_builder_agrees_with_walk():
    """the corner-based builder used for hanging meshes reproduces the face-walk builder on a conforming mesh"""
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import SchwarzMetadata
    mh = M.HangingBrickMesh(2, np.zeros(64, dtype=bool), 2)          # nothing refined: conforming, but with origin / size arrays
    sides = mh.build_sides(None)
    md = SchwarzMetadata(mh, sides, 2)                                # walk path (no hanging side)
    c, e, f = SchwarzMetadata._corner_neighbours(mh)
    order = np.lexsort((e[0], c[0]))
    np.testing.assert_array_equal(c[0][order], md.sub_core)
    np.testing.assert_array_equal(e[0][order], md.sub_elem)
    np.testing.assert_array_equal(f[0][order], md.sub_faces)


def _hanging(level, pattern, deg):
    from disco4est_amd import mesh as M
    refine = np.zeros(8 ** level, dtype=bool)
    refine[np.asarray(pattern)] = True
    n = M.HangingBrickMesh(level, refine, 2).n_elements
    m = M.HangingBrickMesh(level, refine, deg(n) if callable(deg) else deg)
    J, rst = m.geometry(None)
    sides = m.build_sides(None)
    return m, J, rst, sides


def test_metadata_of_hanging_mesh():
    from disco4est_amd.schwarz import SchwarzMetadata
    m, _, _, sides = _hanging(1, [0], 2)                              # 8 small elements in one corner + 7 big ones
    assert m.n_elements == 15
    md = SchwarzMetadata(m, sides, 2)
    members = [set(md.subdomain(s)[0].tolist()) for s in range(15)]
    for s in range(15):
        assert s in members[s]
        for e in members[s]:
            assert s in members[e]                                    # sharing a conformal corner is symmetric
    # the small element at the domain corner only sees its 7 siblings; the small element at the centre of the brick touches all
    # seven big elements at the brick's centre point (a corner of all of them)
    small = np.nonzero(m.size == 1)[0]
    corner = [e for e in small if np.all(m.org[e] == 0)][0]
    centre = [e for e in small if np.all(m.org[e] == 1)][0]
    assert members[corner] == set(small.tolist())
    assert members[centre] == set(range(15))
    # a big element never meets the corner small element (only hanging corners in between), but meets the centre one
    big = np.nonzero(m.size == 2)[0]
    assert all(corner not in members[b] for b in big) and all(centre in members[b] for b in big)
    # faces: the big +x neighbour of the refined cell touches a small element of the x = 1 layer with its "-x" face only
    bx = [b for b in big if tuple(m.org[b]) == (2, 0, 0)][0]
    e_list, faces, core_faces = md.subdomain(bx)
    for e, f, cf in zip(e_list, faces, core_faces):
        if m.size[e] == 1:
            assert m.org[e][0] == 1 and int(f[0]) == 1 and int(cf[0]) == 0      # small element left of the core: its +x face


def test_schwarz_converges_on_hanging_mesh(oracle):
    """On a hanging face the big element is the side element of FOUR small-core subdomains, each weighted with the full ramp on the
    big element's slab (the weights are per subdomain element, d4est_solver_schwarz_helpers.c:344-370), so the hats sum to more than
    one there and the plain stationary iteration over-corrects; as a damped iteration (or a preconditioner) it contracts."""
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import SchwarzMetadata
    m, J, rst, sides = _hanging(1, [0, 7], lambda n: 2 + (np.arange(n) % 2))
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=1)
    try:
        md = SchwarzMetadata(m, sides, 2)
        u_exact = M.splitmix64_uniform(3, m.local_nodes) - 0.5
        rhs = oracle.apply_aij(m, J, rst, sides, u_exact)
        u = np.zeros(m.local_nodes)
        hist = [np.linalg.norm(rhs)]
        for _ in range(3):
            r = rhs - oracle.apply_aij(m, J, rst, sides, u)
            oracle.set_hanging(sides)                                  # apply_aij resets the hanging arrays on exit
            un, it, res = oracle.schwarz_iterate(md, u, r, 300, 1e-15, 1e-10)
            assert it.max() < 300
            u = u + 0.5 * (un - u)
            hist.append(np.linalg.norm(rhs - oracle.apply_aij(m, J, rst, sides, u)))
        assert all(b < 0.8 * a for a, b in zip(hist[:-1], hist[1:])), hist
    finally:
        oracle.set_hanging(None)


@pytest.mark.parametrize("world,pattern", [(2, [0, 7]), (3, [1, 2, 4])])
def test_sharded_metadata_matches_global(world, pattern):
    """subdomains built on a rank's extended mesh (own elements + ghost layer) of a hanging brick list the same global elements and
    faces as the single-rank builder"""
    from disco4est_amd import mesh as M, parallel as P
    from disco4est_amd.schwarz import SchwarzMetadata, ghost_layer_hanging
    refine = np.zeros(8, dtype=bool)
    refine[pattern] = True
    n = M.HangingBrickMesh(1, refine, 2).n_elements
    deg = 2 + (np.arange(n) % 2)
    mg = M.HangingBrickMesh(1, refine, deg)
    mdg = SchwarzMetadata(mg, mg.build_sides(None), 2)
    parts = P.partition_by_dofs(deg, world)
    seen = 0
    for rank, (first, count) in enumerate(parts):
        own, ghosts, needed_by = ghost_layer_hanging(1, refine, parts, rank)
        assert own.size == count and not set(own.tolist()) & set(ghosts.tolist())
        m = M.HangingBrickMesh(1, refine, deg, elements=np.concatenate([own, ghosts]))
        md = SchwarzMetadata(m, m.build_sides(None), 2, cores=np.arange(count), sort_key=m.elements)
        assert md.num_subdomains == count
        for s in range(count):
            e, f, cf = md.subdomain(s)
            eg, fg, cfg = mdg.subdomain(first + s)
            np.testing.assert_array_equal(m.elements[e], eg)
            np.testing.assert_array_equal(f, fg)
            np.testing.assert_array_equal(cf, cfg)
            seen += 1
        # every own element a peer needs is really in that peer's ghost layer
        for peer, mine in needed_by.items():
            _, pg, _ = ghost_layer_hanging(1, refine, parts, peer)
            assert set(mine) <= set(pg.tolist())
    assert seen == n
