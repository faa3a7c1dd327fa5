"""GPU parity on multi-tree meshes (config 5): faces between trees with p4est orientation != 0, every (f_m, f_p, orientation)
triple / reorder code 0..7, hanging faces across tree boundaries, the reference's 7-tree cubed sphere at p up to 15 (curved), a
partition boundary through oriented faces -- traces + flux (apply_aij), Chebyshev, cg_eigs and additive Schwarz through the C-ABI
against the oracle, on all three face-kernel families (p <= 7 vector-ALU, 8..15 tiled MFMA, generic).  Tolerance: fp64,
rel-inf <= 1e-12 for one operator apply (re-association only), 1e-11 / 1e-10 for the iterations."""
import numpy as np
import pytest

from disco4est_amd import forest as F, mesh as M

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _plan(m, J, rst, sides, prefactor=10.0, fcn=0, generic=False):
    from disco4est_amd import Plan
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
    if generic == "split":
        p.set_tuning(13, 1)    # D4EST_HIP_TUNE_HP_SPLIT = 1: conforming sides through the fast conforming kernels, hanging sides through the record kernels
    elif generic:
        p.set_tuning(3, 0)     # D4EST_HIP_TUNE_FLUX_FAST = 0: the generic trace / flux kernels
    p.set_geometry(J, rst)
    p.set_faces(sides, prefactor, fcn)
    return p


def _triples():
    from tests.test_forest import TRIPLES
    return TRIPLES


def _check_aij(gpu, oracle, m, generic=False, fcn=0, tol=RTOL):
    import torch
    J, rst = m.geometry()
    sides = m.build_sides()
    u = m.field()
    bx = sides["bndry_xyz"]
    g = np.sin(bx[0]) + bx[1] * bx[2]
    ref = oracle.apply_aij(m, J, rst, sides, u, bndry_lobatto=g, penalty_prefactor=7.5, penalty_fcn=fcn, nthreads=8)
    plan = _plan(m, J, rst, sides, 7.5, fcn, generic)
    plan.set_dirichlet_values(g)
    du = _t(u, gpu)
    dAu = torch.full_like(du, float("nan"))
    plan.apply_aij(du, dAu)
    got = dAu.cpu().numpy()
    assert np.isfinite(got).all()
    err = _rel(got, ref)
    plan.destroy()
    assert err <= tol, err
    return sides


@pytest.mark.parametrize("deg,generic", [(2, False), (3, True), (7, False), (9, False)])
def test_apply_aij_all_orientations(gpu, hiplib, oracle, deg, generic):
    """Two warped trees glued through each of the 144 (f_m, f_p, orientation) triples, both (-) views: apply_aij == oracle.  The
    oracle restates d4est_operators_reorient_face_data verbatim, so this is parity with the reference on ALL triples -- including
    the 36 whose re-orientation is not geometric in the reference (forest.reference_reorientation_is_consistent)."""
    codes = set()
    triples = _triples()
    step = 1 if deg <= 3 else 5            # the larger degrees sample the triples (every code still occurs)
    for k, (trip, rots) in enumerate(sorted(triples.items())):
        if k % step:
            continue
        conn = F.Connectivity.rotated_pair(*rots)
        m = F.ForestMesh(conn, 0, [deg, max(deg - 1, 1)], F.TrilinearMap(conn, M.SineMap(0.03)), deg_quad_inc=(1 if deg == 2 else 0))
        s = _check_aij(gpu, oracle, m, generic)
        codes.update(int(c) for c in s["side_reorder"])
    assert codes == set(range(8))


@pytest.mark.parametrize("deg,generic", [(2, False), (2, True), (4, False), (8, False), (2, "split"), (5, "split")])
def test_apply_aij_hanging_across_trees(gpu, hiplib, oracle, deg, generic):
    """A hanging (1 <-> 4) face ON the tree boundary, orientation 0..3 (d4est_reference_reorient_face_order), the refined tree on
    either side, mixed p: record kernels (generic and tiled MFMA) against the oracle."""
    triples = _triples()
    seen_o = set()
    n_nongeometric = 0
    for k, (trip, rots) in enumerate(sorted(triples.items())):
        if k % (1 if deg <= 2 else 5):
            continue
        conn = F.Connectivity.rotated_pair(*rots)
        for refine in ([1, 0], [0, 1]):
            m0 = F.ForestMesh(conn, 0, deg, F.TrilinearMap(conn), refine=refine)
            d = deg + (np.arange(m0.global_elements) * 7) % 3
            # the small side's view of the face; where the reference's re-orientation is not geometric the engine follows it for
            # sub-mortars of one quadrature degree (and aborts otherwise, include/d4est_hip.h): uniform p there
            small_view = trip if refine == [1, 0] else (trip[1], trip[0], trip[2])
            if not F.reference_reorientation_is_consistent(*small_view):
                d = deg
                n_nongeometric += 1
            m = F.ForestMesh(conn, 0, d, F.TrilinearMap(conn, M.SineMap(0.03)), refine=refine)
            s = _check_aij(gpu, oracle, m, generic)
            assert (s["side_hang"] == 1).sum() == 1
            seen_o.add(int(s["side_orientation"].max()))
    assert seen_o == {0, 1, 2, 3}
    assert n_nongeometric > 0


@pytest.mark.parametrize("deg,inc,level,refine,generic,compactify", [
    (2, 0, 1, None, False, False), (3, 1, 0, None, True, False), (5, 0, 0, [1, 0, 0, 0, 0, 0, 1], False, True),
    (7, 0, 0, None, False, False), (11, 0, 0, None, False, False), (15, 0, 0, None, False, False),
    (15, 0, 0, [0, 0, 0, 1, 0, 0, 0], False, True), (17, 0, 0, None, False, False),
    (5, 0, 0, [1, 0, 0, 0, 0, 0, 1], "split", True), (3, 1, 0, [0, 0, 1, 0, 0, 0, 0], "split", False),
])
def test_cubed_sphere_apply_aij(gpu, hiplib, oracle, deg, inc, level, refine, generic, compactify):
    """Config 5's mesh class: the reference's 7-tree cubed sphere (curved wedges around a cube; inter-tree codes 1, 2, 3, 7), up to
    p = 15 (tiled MFMA face kernels) and p = 17 (generic), conforming and with hanging faces between trees."""
    conn = F.cubed_sphere_7tree_connectivity()
    m = F.ForestMesh(conn, level, deg, F.CubedSphere7Map(1.0, 2.5, compactify), refine=refine, deg_quad_inc=inc)
    s = _check_aij(gpu, oracle, m, generic)
    assert set(int(c) for c in s["side_reorder"]) >= {0, 1, 2, 3, 7}


def test_cubed_sphere_consistency_at_size(gpu, hiplib):
    """Oracle-free at a size the oracle would not finish quickly: level 2 (448 elements), p = 7, 229 376 DoF, curved; the operator is
    symmetric (v.Aw = w.Av) and annihilates constants given matching Dirichlet data."""
    import torch
    conn = F.cubed_sphere_7tree_connectivity()
    m = F.ForestMesh(conn, 2, 7, F.CubedSphere7Map(1.0, 2.0))
    J, rst = m.geometry()
    sides = m.build_sides()
    assert sides["mortar_xyz_mismatch"] <= 1e-12
    plan = _plan(m, J, rst, sides, 10.0, 0)
    v = _t(M.splitmix64_uniform(1, m.local_nodes), gpu)
    w = _t(M.splitmix64_uniform(2, m.local_nodes), gpu)
    Av, Aw = torch.empty_like(v), torch.empty_like(w)
    plan.apply_aij(v, Av)
    plan.apply_aij(w, Aw)
    a, b = float(w @ Av), float(v @ Aw)
    assert abs(a - b) <= 1e-11 * abs(a)
    assert float(v @ Av) > 0
    plan.set_dirichlet_values(np.full(int(sides["total_bndry_nodes"]), 3.0))
    c = torch.full_like(v, 3.0)
    plan.apply_aij(c, Av)
    assert float(Av.abs().max()) <= 1e-9 * float(Aw.abs().max())
    plan.destroy()


def test_cubed_sphere_p15_at_size(gpu, hiplib, oracle, monkeypatch):
    """Config 5 at size: the 7-tree cubed sphere at level 2 (448 curved elements), p = 15, 1.84 MDoF, every geometric factor generated on the
    device from the analytic map, on the DEFAULT kernel path (the whole operator in operator_mw_kernel<16>: asserted).  The oracle runs
    on shards cut from the mesh (whole-element ghost data gathered from the global vector, host-computed factors of the shard) --
    through faces between trees with orientation != 0 --; around it the size-independent identities: symmetry, positivity, constants
    annihilated with matching Dirichlet data, determinism."""
    import torch
    from disco4est_amd import Plan
    monkeypatch.delenv("D4EST_HIP_FACE_DIRECT", raising=False)
    conn = F.cubed_sphere_7tree_connectivity()
    R0, R1 = 1.0, 2.0
    mp = F.CubedSphere7Map(R0, R1)
    m = F.ForestMesh(conn, 2, 15, mp)
    assert m.n_elements == 448
    tree, q, dq = m.cells()
    params = (R0, R1, 0.0)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
    plan.set_geometry_analytic(1, params, tree, q, dq, m.nf)
    s = m.build_sides_c()      # topology only (d4est_hip_build_sides): no host-side geometric factor anywhere in this plan
    plan.set_faces(s, 10.0, 0, analytic=(1, params, tree, q, dq, m.nf, None))
    assert plan.face_path() == "direct+volume"
    u = m.field()
    du = _t(u, gpu)
    Au = torch.full_like(du, float("nan"))
    plan.apply_aij(du, Au)
    assert "operator_mw_kernel<16" in plan.last_kernel(), plan.last_kernel()
    got = Au.cpu().numpy()
    assert np.isfinite(got).all()
    Au2 = torch.full_like(du, float("nan"))
    plan.apply_aij(du, Au2)
    assert torch.equal(Au, Au2)
    # oracle on three shards of 4 elements: inside the centre cube's neighbourhood (tree 6 against the wedges: orientation != 0), in a
    # wedge, and the last elements (outer boundary)
    n_oriented = 0
    for first in (6 * 64, 2 * 64 + 20, m.n_elements - 4):
        sub = F.ForestMesh(conn, 2, 15, mp, first=first, count=4)
        Js, rsts = sub.geometry(); ss = sub.build_sides()
        n_oriented += int(((ss["side_nbr"] <= -2) & (ss["side_reorder"] != 0)).sum())
        s0 = sub.global_nodal_offset
        ref = oracle.apply_aij(sub, Js, rsts, ss, np.ascontiguousarray(u[s0:s0 + sub.local_nodes]), u_ghost=sub.gather_ghost(ss, u), nthreads=8)
        assert _rel(got[s0:s0 + sub.local_nodes], ref) <= RTOL
    assert n_oriented > 0
    v = _t(M.splitmix64_uniform(1, m.local_nodes), gpu)
    w = _t(M.splitmix64_uniform(2, m.local_nodes), gpu)
    Av, Aw = torch.empty_like(v), torch.empty_like(w)
    plan.apply_aij(v, Av)
    plan.apply_aij(w, Aw)
    a, b = float(w @ Av), float(v @ Aw)
    assert abs(a - b) <= 1e-11 * abs(a)
    assert float(v @ Av) > 0
    plan.set_dirichlet_values(np.full(int(s["total_bndry_nodes"]), 3.0))
    c = torch.full_like(v, 3.0)
    plan.apply_aij(c, Av)
    assert float(Av.abs().max()) <= 1e-8 * float(Aw.abs().max())
    plan.destroy()


@pytest.mark.parametrize("deg,refine", [(3, None), (9, None), (4, [0, 0, 1, 0, 0, 0, 1])])
def test_cubed_sphere_smoothers(gpu, hiplib, oracle, deg, refine):
    """Chebyshev iteration and cg_eigs on the multi-tree operator (d4est_solver_multigrid_smoother_cheby_iterate_aux, cg_eigs)."""
    import torch
    conn = F.cubed_sphere_7tree_connectivity()
    m = F.ForestMesh(conn, 0, deg, F.CubedSphere7Map(1.0, 2.0), refine=refine)
    J, rst = m.geometry()
    sides = m.build_sides()
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=8)
    oracle.set_hanging(sides)
    try:
        plan = _plan(m, J, rst, sides, 10.0, 0)
        rhs = M.splitmix64_uniform(5, m.local_nodes) - 0.5
        u0 = M.splitmix64_uniform(6, m.local_nodes)
        bound_ref, u_after_ref = oracle.cg_eigs(u0, rhs, 8)
        du, drhs = _t(u0, gpu), _t(rhs, gpu)
        dAu, dr = torch.empty_like(du), torch.empty_like(du)
        bound, _ = plan.cg_eigs(du, drhs, dAu, 8)
        assert abs(bound - bound_ref) <= 1e-10 * abs(bound_ref)
        lmax = bound_ref
        u_ref, r_ref = oracle.cheby_iterate(u0, rhs, 6, lmax / 30.0, lmax)
        du = _t(u0, gpu)
        plan.cheby_iterate(du, drhs, dAu, dr, 6, lmax / 30.0, lmax)
        assert _rel(du.cpu().numpy(), u_ref) <= 1e-11
        assert _rel(dr.cpu().numpy(), r_ref) <= 1e-10
        plan.destroy()
    finally:
        oracle.set_hanging(None)


def test_partition_through_oriented_faces(gpu, hiplib, oracle):
    """Shards whose boundary runs through faces with orientation != 0 (the centre cube against its wedges) and through a hanging
    face: every shard's apply_aij, fed with ghost traces computed from ghost element data, equals its slice of the one-rank oracle."""
    import torch
    conn = F.cubed_sphere_7tree_connectivity()
    mp = F.CubedSphere7Map(1.0, 2.0)
    for deg, refine, parts in ((3, None, [(0, 3), (3, 2), (5, 2)]), (8, None, [(0, 4), (4, 3)])):
        mg = F.ForestMesh(conn, 0, deg, mp, refine=refine)
        Jg, rstg = mg.geometry()
        sg = mg.build_sides()
        ug = mg.field()
        ref = oracle.apply_aij(mg, Jg, rstg, sg, ug, nthreads=8)
        n_oriented = 0
        for first, count in parts:
            m = F.ForestMesh(conn, 0, deg, mp, refine=refine, first=first, count=count)
            J, rst = m.geometry()
            s = m.build_sides()
            n_oriented += int(((s["side_nbr"] <= -2) & (s["side_reorder"] != 0)).sum())
            plan = _plan(m, J, rst, s, 10.0, 0)
            du = _t(m.field(), gpu)
            gt = torch.zeros(int(plan.ghost_trace_size), dtype=torch.float64, device=gpu)
            plan.compute_ghost_traces(_t(m.gather_ghost(s, ug), gpu), gt)
            dAu = torch.full_like(du, float("nan"))
            plan.apply_aij(du, dAu, gt)
            sl = slice(m.global_nodal_offset, m.global_nodal_offset + m.local_nodes)
            assert _rel(dAu.cpu().numpy(), ref[sl]) <= RTOL
            plan.destroy()
        assert n_oriented > 0


@pytest.mark.parametrize("case", ["cubed_sphere_p3", "cubed_sphere_p9", "cubed_sphere_hanging", "rotated_pair_level1"])
def test_schwarz_on_multi_tree_meshes(gpu, hiplib, oracle, case):
    """Additive Schwarz (d4est_solver_schwarz_iterate) where subdomains reach across tree boundaries with orientation != 0: the
    subdomain metadata comes from the corner iteration restated on points (schwarz.SchwarzMetadata._corner_neighbours_forest), the
    subdomain operator re-orients the traces of the copies, every face-kernel family: restriction bit-exact, one iterate against the
    oracle's serial subdomain-by-subdomain CG at 1e-9 with identical iteration counts."""
    import torch
    from disco4est_amd.schwarz import Schwarz
    if case.startswith("cubed_sphere"):
        conn = F.cubed_sphere_7tree_connectivity()
        deg = 9 if case.endswith("p9") else 3
        refine = [0, 0, 1, 0, 0, 0, 1] if case.endswith("hanging") else None
        m = F.ForestMesh(conn, 0, deg, F.CubedSphere7Map(1.0, 2.0), refine=refine)
    else:
        rots = _triples()[(0, 3, 3)]                      # reorder code 7 (both flips + transpose), geometric from both sides
        conn = F.Connectivity.rotated_pair(*rots)
        m = F.ForestMesh(conn, 1, 2 + (np.arange(16) * 5) % 3, F.TrilinearMap(conn, M.SineMap(0.03)))
    J, rst = m.geometry()
    sides = m.build_sides()
    assert sides["mortar_xyz_mismatch"] <= 1e-12
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=8)
    oracle.set_hanging(sides)
    try:
        rs, iters = 2, 6
        sz = Schwarz(m, sides, J, rst, rs, iters, 1e-15, 1e-15, 10.0, 0)
        md = sz.metadata
        assert md.num_subdomains == m.n_elements
        r = M.splitmix64_uniform(7, m.local_nodes) - 0.5
        u_ref, it_ref, _ = oracle.schwarz_iterate(md, np.zeros(m.local_nodes), r, iters, 1e-15, 1e-15)
        u = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
        sz.iterate(u, _t(r, gpu))
        it, _ = sz.info()
        assert np.array_equal(it, it_ref)
        assert _rel(u.cpu().numpy(), u_ref) <= 1e-9
        sz.destroy()
    finally:
        oracle.set_hanging(None)


class _Mailbox:
    def __init__(self):
        self.box = {}


class _LocalTransport:
    """in-process stand-in for the point-to-point transport (as in tests/test_parallel_gpu.py)"""

    def __init__(self, rank, mailbox):
        self.rank, self.mb = rank, mailbox

    def start(self, send_buf, recv_buf):
        for p, t in send_buf.items():
            self.mb.box[(self.rank, p)] = t.clone()
        return recv_buf

    def finish(self, recv_buf):
        for p, t in recv_buf.items():
            t.copy_(self.mb.box[(p, self.rank)])


@pytest.mark.parametrize("deg,refine,world", [(3, None, 3), (8, None, 2), (2, [0, 0, 1, 0, 0, 0, 1], 4), (4, [1, 0, 0, 0, 0, 1, 0], 3)])
def test_trace_exchange_through_oriented_and_hanging_tree_faces(gpu, hiplib, oracle, deg, refine, world):
    """The FACE-TRACE exchange (what travels over RCCL: mortar-node blocks, not ghost elements) between virtual ranks whose boundary
    runs through tree faces with orientation != 0 and through hanging faces between trees: the receiver re-orients the sender's
    block, a small side receives the big element's sub-block d4est_reference_reorient_face_order names.  Assembled A u == the
    one-rank oracle."""
    import torch
    from disco4est_amd import Plan, parallel as P
    conn = F.cubed_sphere_7tree_connectivity()
    mp = F.CubedSphere7Map(1.0, 2.0)
    mg = F.ForestMesh(conn, 0, deg, mp, refine=refine)
    Jg, rstg = mg.geometry(); sg = mg.build_sides(); ug = mg.field()
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug, nthreads=8)
    parts = P.partition_by_dofs(mg.deg_global, world)
    mb = _Mailbox()
    ranks, oriented, hanging_cut = [], 0, 0
    for r, (first, count) in enumerate(parts):
        m = F.ForestMesh(conn, 0, deg, mp, refine=refine, first=first, count=count)
        J, rst = m.geometry(); s = m.build_sides()
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_geometry(J, rst); plan.set_faces(s)
        ex = P.TraceExchange(P.plan_schedule(plan, m, s, parts), _LocalTransport(r, mb), plan.copy_blocks, gpu)
        oriented += int(((s["side_nbr"] <= -2) & (s["side_reorder"] != 0)).sum())
        if "side_hang" in s:
            hanging_cut += int(((s["side_hang"] == 2) & (s["side_nbr"] <= -2)).sum())
        u = torch.from_numpy(m.field()).to(gpu)
        tr = torch.empty(plan.trace_size, dtype=torch.float64, device=gpu)
        gt = torch.full((max(plan.ghost_trace_size, 1),), float("nan"), dtype=torch.float64, device=gpu)
        ranks.append((m, plan, ex, u, tr, gt))
    assert oriented > 0 and (refine is None or hanging_cut > 0)
    for m, plan, ex, u, tr, gt in ranks:
        plan.compute_face_traces(u, tr)
        ex.begin(tr)
    got = np.zeros_like(ref)
    for m, plan, ex, u, tr, gt in ranks:
        ex.end(gt)
        Au = torch.full_like(u, float("nan"))
        plan.apply_stiffness_matrix(u, Au)
        plan.apply_flux(tr, gt, Au)
        got[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes] = Au.cpu().numpy()
        plan.destroy()
    assert np.isfinite(got).all()
    assert _rel(got, ref) <= RTOL


@pytest.mark.parametrize("deg,inc,level,refine,compactify,world", [
    (3, 0, 1, None, False, 1), (2, 1, 0, [1, 0, 0, 0, 0, 0, 1], True, 1), (7, 0, 0, None, False, 1), (9, 0, 0, [0, 0, 0, 1, 0, 0, 0], False, 1),
    (15, 0, 0, None, True, 1), (4, 0, 0, [0, 0, 1, 0, 0, 0, 1], False, 3),
])
def test_cubed_sphere_factors_generated_on_the_device(gpu, hiplib, oracle, deg, inc, level, refine, compactify, world):
    """SURVEY section 8f rank 4 for config 5's geometry: volume AND mortar factors of the cubed sphere from the analytic map on the
    device (d4est_hip_plan_set_geometry_analytic / _set_mortar_geometry_analytic: the host passes tree, q, dq per element) -- through
    oriented tree faces, hanging faces and ghost elements -- against the oracle fed with the host-computed arrays."""
    import torch
    from disco4est_amd import Plan, parallel as P
    conn = F.cubed_sphere_7tree_connectivity()
    R0, R1 = 1.0, 2.5
    mp = F.CubedSphere7Map(R0, R1, compactify)
    mg = F.ForestMesh(conn, level, deg, mp, refine=refine, deg_quad_inc=inc)
    parts = P.partition_by_dofs(mg.deg_global, world)
    ug = mg.field()
    Jg, rstg = mg.geometry(); sg = mg.build_sides()
    ref_full = oracle.apply_aij(mg, Jg, rstg, sg, ug, penalty_prefactor=7.5, nthreads=8)
    mb = _Mailbox()
    ranks = []
    for r, (first, count) in enumerate(parts):
        m = F.ForestMesh(conn, level, deg, mp, refine=refine, deg_quad_inc=inc, first=first, count=count)
        s = m.build_sides()
        tree, q, dq = m.cells()
        params = (R0, R1, float(compactify))
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
        plan.set_geometry_analytic(1, params, tree, q, dq, m.nf)
        plan.set_faces(s, 7.5, 0, analytic=(1, params, tree, q, dq, m.nf, m.cells(s["ghost_global_ids"])))
        ex = P.TraceExchange(P.plan_schedule(plan, m, s, parts), _LocalTransport(r, mb), plan.copy_blocks, gpu)
        du = _t(m.field(), gpu)
        tr = torch.empty(plan.trace_size, dtype=torch.float64, device=gpu)
        gt = torch.full((max(plan.ghost_trace_size, 1),), float("nan"), dtype=torch.float64, device=gpu)
        ranks.append((m, plan, ex, du, tr, gt))
    for m, plan, ex, du, tr, gt in ranks:
        plan.compute_face_traces(du, tr)
        ex.begin(tr)
    for m, plan, ex, du, tr, gt in ranks:
        ex.end(gt)
        dAu = torch.full_like(du, float("nan"))
        plan.apply_stiffness_matrix(du, dAu)
        # the volume factors alone: stiffness against the oracle with host arrays
        J, rst = m.geometry()
        assert _rel(dAu.cpu().numpy(), oracle.apply_stiffness(m, J, rst, m.field(), nthreads=8)) <= RTOL
        plan.apply_flux(tr, gt, dAu)
        sl = slice(m.global_nodal_offset, m.global_nodal_offset + m.local_nodes)
        assert _rel(dAu.cpu().numpy(), ref_full[sl]) <= RTOL
        plan.destroy()


@pytest.mark.parametrize("deg,refine", [(3, 1), (7, 0)])
def test_plain_c_multi_tree_host(gpu, hiplib, oracle, tmp_path, deg, refine):
    """tests/c/forest_probe.c: the whole config-5 path from plain C99 -- side list built by d4est_hip_build_sides (no p4est), factors
    generated on the device, host-pointer apply -- and its v.Aw against the oracle on the same mesh built by forest.py."""
    import os
    import re
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "disco4est_amd")
    exe = str(tmp_path / "forest_probe")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "forest_probe.c"), "-L" + lib_dir, "-ld4est_hip", "-lm", "-Wl,-rpath," + lib_dir, "-o", exe])
    out = subprocess.run([exe, str(deg), str(refine)], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok")
    vAw = float(re.search(r"v\.Aw = (\S+)", out.stdout).group(1))
    # the same mesh through forest.py + the oracle (host-computed factors)
    conn = F.cubed_sphere_7tree_connectivity()
    ref_mask = np.zeros(56, dtype=bool)
    if refine:
        ref_mask[[6 * 8 + 7, 2 * 8 + 0]] = True
    m0 = F.ForestMesh(conn, 1, deg, F.CubedSphere7Map(1.0, 2.0), refine=ref_mask)
    d = deg + (np.arange(m0.global_elements) % 3 == 1)
    m = F.ForestMesh(conn, 1, d, F.CubedSphere7Map(1.0, 2.0), refine=ref_mask)
    J, rst = m.geometry(); s = m.build_sides()
    v = M.splitmix64_uniform(1, m.local_nodes); w = M.splitmix64_uniform(2, m.local_nodes)
    Aw = oracle.apply_aij(m, J, rst, s, w, nthreads=8)
    assert abs(v @ Aw - vAw) <= 1e-11 * abs(vAw)
