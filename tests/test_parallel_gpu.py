"""GPU test of the sharded full operator with the trace exchange (HIP pack/unpack kernels + schedule), emulating
several ranks inside ONE process on one GPU (multi-GPU boxes are not available to the tests): every virtual rank has
its own plan; an in-process transport moves the packed buffers; the assembled result must equal the single-rank
operator (rank-count invariance, src/Tests/Regression/d4est_test_mpi.sh)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _Mailbox:
    def __init__(self):
        self.box = {}


class _LocalTransport:
    def __init__(self, rank, mailbox):
        self.rank, self.mb = rank, mailbox

    def start(self, send_buf, recv_buf):
        for p, t in send_buf.items():
            self.mb.box[(self.rank, p)] = t.clone()
        return recv_buf

    def finish(self, recv_buf):
        for p, t in recv_buf.items():
            t.copy_(self.mb.box[(p, self.rank)])


@pytest.mark.parametrize("world,level,deg_spec", [(2, 2, [3]), (3, 2, [2, 3, 4]), (4, 2, [7]), (2, 1, [9]), (3, 1, [12])])
def test_virtual_ranks_match_single_rank(gpu, hiplib, oracle, world, level, deg_spec):
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    n_global = 8 ** level
    deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n_global)])
    mp = M.SineMap(0.04)
    mg = M.BrickMesh(level, deg_global)
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp); ug = mg.field(mp)
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug, nthreads=8)
    parts = P.partition_by_dofs(deg_global, world)
    mb = _Mailbox()
    ranks = []
    for r, (first, count) in enumerate(parts):
        m = M.BrickMesh(level, deg_global, first=first, count=count)
        J, rst = m.geometry(mp); s = m.build_sides(mp)
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_geometry(J, rst); plan.set_faces(s)
        sched = P.plan_schedule(plan, m, s, parts)
        # the host-side layout formula used by the CPU-only tests is the library's layout
        toff, goff, blen = P.side_block_layout(s)
        for sd in range(6 * m.n_elements):
            assert plan.lib.d4est_hip_plan_trace_offset(plan.handle, sd) == toff[sd]
            assert plan.lib.d4est_hip_plan_ghost_trace_offset(plan.handle, sd) == goff[sd]
            assert plan.lib.d4est_hip_plan_trace_block_len(plan.handle, sd) == blen[sd]
        ex = P.TraceExchange(sched, _LocalTransport(r, mb), plan.copy_blocks, gpu)
        u = torch.from_numpy(m.field(mp)).to(gpu)
        tr = torch.empty(plan.trace_size, dtype=torch.float64, device=gpu)
        gt = torch.full((max(plan.ghost_trace_size, 1),), float("nan"), dtype=torch.float64, device=gpu)
        ranks.append((m, plan, ex, u, tr, gt))
    for m, plan, ex, u, tr, gt in ranks:       # phase 0 on every rank: traces + pack + post
        plan.compute_face_traces(u, tr)
        ex.begin(tr)
    got = np.zeros_like(ref)
    for m, plan, ex, u, tr, gt in ranks:       # phase 1: receive + unpack, then volume + flux
        ex.end(gt)
        Au = torch.full_like(u, float("nan"))
        plan.apply_stiffness_matrix(u, Au)
        plan.apply_flux(tr, gt, Au)
        got[m.global_nodal_offset:m.global_nodal_offset + m.local_nodes] = Au.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("split", [-1, 0, 1])
@pytest.mark.parametrize("world,level,pattern,deg_spec", [(2, 1, [0, 5, 6], [2]), (3, 1, [1, 2, 4, 7], [2, 3]), (4, 2, [0, 9, 21, 42, 63], [3]),
                                                          (3, 1, [3, 6], [7]), (2, 1, [2], [9])])
def test_virtual_ranks_hanging_mesh(gpu, hiplib, oracle, world, level, pattern, deg_spec, split):
    """Hanging (1 <-> 4) faces ACROSS rank boundaries: a big side sends one block per sub-mortar to the owners of its four small
    neighbours, a small side receives the big element's sub-block that faces it; the assembled operator equals the single-rank one.
    split: tuning key 13 -- automatic, every side through the record kernels, or (degrees <= 7) the conforming and small sides through
    the fast conforming kernels, ghost (+) sides included."""
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    refine = np.zeros(8 ** level, dtype=bool)
    refine[np.asarray(pattern)] = True
    m0 = M.HangingBrickMesh(level, refine, deg_spec[0])
    deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(m0.global_elements)])
    mp = M.SineMap(0.04)
    mg = M.HangingBrickMesh(level, refine, deg_global)
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp); ug = mg.field(mp)
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug, nthreads=8)
    parts = P.partition_by_dofs(deg_global, world)
    mb = _Mailbox()
    ranks = []
    crossing = 0
    for r, (first, count) in enumerate(parts):
        m = M.HangingBrickMesh(level, refine, deg_global, first=first, count=count)
        J, rst = m.geometry(mp); s = m.build_sides(mp)
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_geometry(J, rst); plan.set_tuning(13, split); plan.set_faces(s)
        sched = P.plan_schedule(plan, m, s, parts)
        # the host-side layout formula used by the CPU-only tests is the library's layout
        nblk, off, goff, ln, n_trace, n_ghost = P.side_block_layout_hp(m, s)
        assert plan.trace_size == n_trace and plan.ghost_trace_size == n_ghost
        for sd in range(6 * m.n_elements):
            assert plan.lib.d4est_hip_plan_side_blocks(plan.handle, sd) == nblk[sd]
            for sub in range(nblk[sd]):
                assert plan.lib.d4est_hip_plan_trace_offset_sub(plan.handle, sd, sub) == off[(sd, sub)]
                assert plan.lib.d4est_hip_plan_ghost_trace_offset_sub(plan.handle, sd, sub) == goff[(sd, sub)]
                assert plan.lib.d4est_hip_plan_trace_block_len_sub(plan.handle, sd, sub) == ln[(sd, sub)]
        hang, nbr, n4 = s["side_hang"], s["side_nbr"], s["side_nbr4"]
        crossing += int(((hang == 2) & (nbr <= -2)).sum()) + int(sum((n4[4 * i:4 * i + 4] <= -2).sum() for i in np.nonzero(hang == 1)[0]))
        ex = P.TraceExchange(sched, _LocalTransport(r, mb), plan.copy_blocks, gpu)
        u = torch.from_numpy(m.field(mp)).to(gpu)
        tr = torch.empty(plan.trace_size, dtype=torch.float64, device=gpu)
        gt = torch.full((max(plan.ghost_trace_size, 1),), float("nan"), dtype=torch.float64, device=gpu)
        ranks.append((m, plan, ex, u, tr, gt))
    assert crossing > 0, "the partition must cut at least one hanging face"
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
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()


def test_apply_lhs_hooks_single_rank_with_self_exchange(gpu, hiplib, oracle):
    """apply_lhs / cheby through the C callback hooks: a 2-rank split where the 'remote' rank is served in-process."""
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    level, deg = 1, 3
    deg_global = np.full(8, deg)
    mg = M.BrickMesh(level, deg_global)
    Jg, rstg = mg.geometry(None); sg = mg.build_sides(None); ug = mg.field()
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug)
    parts = [(0, 4), (4, 4)]
    mb = _Mailbox()
    objs = []
    for r, (first, count) in enumerate(parts):
        m = M.BrickMesh(level, deg_global, first=first, count=count)
        J, rst = m.geometry(None); s = m.build_sides(None)
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_geometry(J, rst); plan.set_faces(s)
        objs.append((m, s, plan))
    # rank 1's traces are produced up front and parked in the mailbox; rank 0 then runs apply_lhs with live hooks
    m1, s1, p1 = objs[1]
    ex1 = P.attach(p1, m1, s1, parts, _LocalTransport(1, mb), gpu)
    u1 = torch.from_numpy(m1.field()).to(gpu)
    tr1 = torch.empty(p1.trace_size, dtype=torch.float64, device=gpu)
    p1.compute_face_traces(u1, tr1)
    ex1.begin(tr1)
    m0, s0, p0 = objs[0]
    ex0 = P.attach(p0, m0, s0, parts, _LocalTransport(0, mb), gpu)
    u0 = torch.from_numpy(m0.field()).to(gpu)
    Au0 = torch.full_like(u0, float("nan"))
    p0.apply_lhs(u0, Au0)                     # C -> python exchange callback (phase 0, phase 1) -> C
    got = Au0.cpu().numpy()
    assert np.abs(got - ref[:m0.local_nodes]).max() <= 1e-12 * np.abs(ref).max()
    assert ex0 is not None


@pytest.mark.parametrize("world,level,deg_spec,rs,curved", [(2, 2, [2], 2, False), (3, 2, [2, 3], 2, True), (4, 2, [3], 3, True)])
def test_schwarz_virtual_ranks_match_single_rank(gpu, hiplib, oracle, world, level, deg_spec, rs, curved):
    """The Schwarz smoother on a sharded brick: every rank solves the subdomains of its own elements on its extended mesh (own elements
    + ghost layer), residuals of ghost-layer elements arrive by a whole-element exchange, their corrections travel back and are added
    by the owners.  The assembled u equals the single-rank smoother's (which the oracle pins) to 1e-10."""
    import torch
    from disco4est_amd import mesh as M, parallel as P
    from disco4est_amd.schwarz import Schwarz, SchwarzShard
    n_global = 8 ** level
    deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n_global)])
    mp = M.SineMap(0.04) if curved else None
    mg = M.BrickMesh(level, deg_global)
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp)
    iters = 5
    single = Schwarz(mg, sg, Jg, rstg, rs, iters, 1e-15, 1e-15)
    u0 = M.splitmix64_uniform(81, mg.local_nodes) - 0.5
    r = M.splitmix64_uniform(82, mg.local_nodes) - 0.5
    u_ref = torch.from_numpy(u0).to(gpu)
    single.iterate(u_ref, torch.from_numpy(r).to(gpu))
    u_ref = u_ref.cpu().numpy()
    oracle.set_operator(mg, Jg, rstg, sg, 10.0, 0, threads=1)
    u_orc, _, _ = oracle.schwarz_iterate(single.metadata, u0, r, iters, 1e-15, 1e-15)
    assert np.abs(u_ref - u_orc).max() <= 1e-9 * np.abs(u_orc - u0).max()
    parts = P.partition_by_dofs(deg_global, world)
    mb, mb_back = _Mailbox(), _Mailbox()       # in-process stand-in for two message streams (the phases of the virtual ranks interleave)
    shards = []
    for rank, (first, count) in enumerate(parts):
        sh = SchwarzShard(level, deg_global, parts, rank, mp, rs, iters, 1e-15, 1e-15, _LocalTransport(rank, mb), gpu,
                          transport_back=_LocalTransport(rank, mb_back))
        lo = int(mg.global_nodal_stride[first]); hi = lo + sh.own_nodes
        shards.append((sh, torch.from_numpy(u0[lo:hi].copy()).to(gpu), torch.from_numpy(r[lo:hi].copy()).to(gpu), lo, hi))
        assert sh.mesh.n_elements > count                      # the extended mesh really has a ghost layer
    for sh, u, rr, lo, hi in shards:
        sh.begin_residual_exchange(rr)
    for sh, u, rr, lo, hi in shards:
        sh.solve_and_begin_correction_exchange()
    got = np.empty_like(u_ref)
    for sh, u, rr, lo, hi in shards:
        sh.end_correction_exchange(u)
        got[lo:hi] = u.cpu().numpy()
    assert np.abs(got - u_ref).max() <= 1e-10 * np.abs(u_ref - u0).max()


@pytest.mark.parametrize("world,pattern,deg_spec,rs", [(2, [0, 7], [2], 2), (3, [1, 2, 4], [2, 3], 2)])
def test_schwarz_virtual_ranks_hanging_mesh(gpu, hiplib, oracle, world, pattern, deg_spec, rs):
    """Schwarz smoother on a sharded brick WITH hanging faces: subdomains reach across rank boundaries and across hanging faces; the
    assembled result equals the single-rank smoother's (itself checked against the oracle in test_schwarz_gpu.py)."""
    import torch
    from disco4est_amd import mesh as M, parallel as P
    from disco4est_amd.schwarz import Schwarz, SchwarzShard
    refine = np.zeros(8, dtype=bool)
    refine[pattern] = True
    n = M.HangingBrickMesh(1, refine, 2).n_elements
    deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n)], dtype=np.int32)
    mp = M.SineMap(0.04)
    mg = M.HangingBrickMesh(1, refine, deg_global)
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp)
    iters = 5
    single = Schwarz(mg, sg, Jg, rstg, rs, iters, 1e-15, 1e-15)
    u0 = M.splitmix64_uniform(83, mg.local_nodes) - 0.5
    r = M.splitmix64_uniform(84, mg.local_nodes) - 0.5
    u_ref = torch.from_numpy(u0).to(gpu)
    single.iterate(u_ref, torch.from_numpy(r).to(gpu))
    u_ref = u_ref.cpu().numpy()
    parts = P.partition_by_dofs(deg_global, world)
    mb, mb_back = _Mailbox(), _Mailbox()
    shards = []
    crossing = 0
    for rank, (first, count) in enumerate(parts):
        sh = SchwarzShard(1, deg_global, parts, rank, mp, rs, iters, 1e-15, 1e-15, _LocalTransport(rank, mb), gpu,
                          transport_back=_LocalTransport(rank, mb_back), refine=refine)
        lo = int(mg.global_nodal_stride[first]); hi = lo + sh.own_nodes
        shards.append((sh, torch.from_numpy(u0[lo:hi].copy()).to(gpu), torch.from_numpy(r[lo:hi].copy()).to(gpu), lo, hi))
        md = sh.schwarz.metadata
        crossing += int((md.sub_elem >= count).sum())           # subdomain members that live on other ranks
    assert crossing > 0
    for sh, u, rr, lo, hi in shards:
        sh.begin_residual_exchange(rr)
    for sh, u, rr, lo, hi in shards:
        sh.solve_and_begin_correction_exchange()
    got = np.empty_like(u_ref)
    for sh, u, rr, lo, hi in shards:
        sh.end_correction_exchange(u)
        got[lo:hi] = u.cpu().numpy()
    assert np.abs(got - u_ref).max() <= 1e-10 * np.abs(u_ref - u0).max()
