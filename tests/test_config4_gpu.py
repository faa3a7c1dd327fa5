"""Config 4 in miniature (BASELINE.json configs[3]: hp-AMR Newton-Krylov-multigrid -- locally refined mesh, mixed p = 3 ... 9, the
linearised operator with its zeroth-order term, the Chebyshev smoother, N ranks), every ingredient AT ONCE on one small mesh:

* one plan: apply_lhs and the Chebyshev iteration against the oracle (hanging 1 <-> 4 mortars between elements of different degree,
  curved map, over-integration, the term V^T W J c V u);
* 2 and 3 virtual ranks whose shard boundaries cut hanging faces: the smoother run in lockstep over the ranks (trace exchange per
  iteration, the library's own pieces: traces, volume term, zeroth-order term, flux, fused update) reproduces the one-plan iterate.
The pieces are tested one by one elsewhere (tests/test_faces_gpu.py, test_solver_gpu.py, test_parallel_gpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


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


def _mesh(first=0, count=None, deg_global=None):
    from disco4est_amd import mesh as M
    refine = np.zeros(8, dtype=bool)
    refine[[2, 5]] = True                      # two of the eight level-1 elements refined once: 22 elements, hanging faces around both
    if deg_global is None:
        m0 = M.HangingBrickMesh(1, refine, 3)
        # degrees 3 ... 9 scattered over big and small elements (config 4: min_degree 3, max_degree 9)
        deg_global = 3 + (np.arange(m0.global_elements) * 5 % 7)
    return M.HangingBrickMesh(1, refine, deg_global, deg_quad_inc=1, first=first, count=count), deg_global


def _coeff(m, offset_quad):
    from disco4est_amd import mesh as M
    return 1.0 + 4.0 * M.splitmix64_uniform(7, m.local_nodes_quad, offset=offset_quad)     # positive: the operator stays SPD


def test_config4_miniature_one_plan_against_the_oracle(gpu, hiplib, oracle):
    import torch
    from disco4est_amd import Plan, mesh as M
    m, deg_global = _mesh()
    assert deg_global.min() == 3 and deg_global.max() == 9 and "side_hang" in m.build_sides(None)
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    assert int((sides["side_hang"] != 0).sum()) > 0
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    assert plan.face_path() == "two-phase"       # mixed p + hanging faces: the mortar-record kernels
    coeff = _coeff(m, 0)
    dcoeff = _t(coeff, gpu)
    oracle.set_operator(m, J, rst, sides, 10.0, 0)
    oracle.set_hanging(sides)
    try:
        oracle.set_lhs_coefficient(coeff)
        plan.set_lhs_coefficient(dcoeff)
        u0 = M.splitmix64_uniform(31, m.local_nodes)
        rhs = M.splitmix64_uniform(32, m.local_nodes) - 0.5
        du = _t(u0, gpu); dAu = torch.full_like(du, float("nan"))
        plan.apply_lhs(du, dAu)
        assert _rel(dAu.cpu().numpy(), oracle.apply_lhs(u0)) <= 1e-12
        lmax, _ = oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 10)
        b, _ = plan.cg_eigs(torch.zeros_like(du), _t(rhs, gpu), dAu, 10)
        assert abs(b - lmax) <= 1e-9 * lmax
        u_ref, r_ref = oracle.cheby_iterate(u0, rhs, 5, lmax / 30.0, lmax, 1)
        du = _t(u0, gpu); dr = torch.full_like(du, float("nan"))
        plan.cheby_iterate(du, _t(rhs, gpu), dAu, dr, 5, lmax / 30.0, lmax, 1)
        assert _rel(du.cpu().numpy(), u_ref) <= 1e-11 and _rel(dr.cpu().numpy(), r_ref) <= 1e-10
    finally:
        oracle.set_lhs_coefficient(None)
        oracle.set_hanging(None)
        plan.set_lhs_coefficient(None)
    plan.destroy()


@pytest.mark.parametrize("world", [2, 3])
def test_config4_miniature_ranks_in_lockstep(gpu, hiplib, oracle, world):
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    mg, deg_global = _mesh()
    mp = M.SineMap(0.04)
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp)
    u0 = M.splitmix64_uniform(31, mg.local_nodes)
    rhs = M.splitmix64_uniform(32, mg.local_nodes) - 0.5
    coeff_g = _coeff(mg, 0)
    iters, lmax = 4, 60.0
    lmin = lmax / 30.0
    # ---- the one-plan iterate (held to the oracle by the test above)
    pg = Plan(mg.deg, mg.deg_quad, mg.nodal_stride, mg.quad_stride, 0)
    pg.set_geometry(Jg, rstg); pg.set_faces(sg, 10.0, 0)
    dcg = _t(coeff_g, gpu)
    pg.set_lhs_coefficient(dcg)
    du = _t(u0, gpu); dAu = torch.empty_like(du); dr = torch.empty_like(du)
    pg.cheby_iterate(du, _t(rhs, gpu), dAu, dr, iters, lmin, lmax, 0)
    u_one = du.cpu().numpy()
    pg.set_lhs_coefficient(None)
    # ---- the same over `world` shards, balanced by DoF count (what p4est_partition does with weights)
    parts = P.partition_by_dofs(deg_global, world)
    mb = _Mailbox()
    ranks, crossing = [], 0
    for r, (first, count) in enumerate(parts):
        m, _ = _mesh(first, count, deg_global)
        J, rst = m.geometry(mp); s = m.build_sides(mp)
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_geometry(J, rst); plan.set_faces(s, 10.0, 0)
        hang, nbr, n4 = s["side_hang"], s["side_nbr"], s["side_nbr4"]
        crossing += int(((hang == 2) & (nbr <= -2)).sum()) + int(sum((n4[4 * i:4 * i + 4] <= -2).sum() for i in np.nonzero(hang == 1)[0]))
        ex = P.TraceExchange(P.plan_schedule(plan, m, s, parts), _LocalTransport(r, mb), plan.copy_blocks, gpu)
        lo, n = m.global_nodal_offset, m.local_nodes
        qlo = int(np.concatenate([[0], np.cumsum((mg.deg_quad.astype(np.int64) + 1) ** 3)])[first])
        st = {"m": m, "plan": plan, "ex": ex, "lo": lo,
              "u": _t(u0[lo:lo + n], gpu), "rhs": _t(rhs[lo:lo + n], gpu), "coeff": _t(coeff_g[qlo:qlo + m.local_nodes_quad], gpu),
              "p": torch.zeros(n, dtype=torch.float64, device=gpu), "r": torch.empty(n, dtype=torch.float64, device=gpu),
              "Au": torch.empty(n, dtype=torch.float64, device=gpu), "w": torch.empty(n, dtype=torch.float64, device=gpu),
              "tr": torch.empty(plan.trace_size, dtype=torch.float64, device=gpu),
              "gt": torch.full((max(plan.ghost_trace_size, 1),), float("nan"), dtype=torch.float64, device=gpu)}
        ranks.append(st)
    assert crossing > 0, "the partition must cut at least one hanging face"
    d, c = (lmax + lmin) * .5, (lmax - lmin) * .5            # Solver/d4est_solver_multigrid_smoother_cheby.c:119-154
    alpha = 0.0
    for i in range(iters):
        alpha = 1. / d if i == 0 else (2. * d / (2 * d * d - c * c) if i == 1 else 1. / (d - (alpha * c * c / 4.)))
        beta = alpha * d - 1.
        for st in ranks:                                         # traces of the current iterate, packed and posted
            st["plan"].compute_face_traces(st["u"], st["tr"])
            st["ex"].begin(st["tr"])
        for st in ranks:                                         # volume + zeroth-order term, then the faces with the received traces
            st["ex"].end(st["gt"])
            pl = st["plan"]
            pl.apply_stiffness_matrix(st["u"], st["Au"])
            pl.apply_weighted_mass_matrix(st["u"], st["coeff"], st["w"])
            st["Au"] += st["w"]
            pl.apply_flux(st["tr"], st["gt"], st["Au"])
            pl.lib.d4est_hip_cheby_update(pl.handle, st["u"].numel(), st["rhs"].data_ptr(), st["Au"].data_ptr(), alpha, beta,
                                          st["r"].data_ptr(), st["p"].data_ptr(), st["u"].data_ptr())
    got = np.zeros_like(u_one)
    for st in ranks:
        got[st["lo"]:st["lo"] + st["m"].local_nodes] = st["u"].cpu().numpy()
        st["plan"].destroy()
    pg.destroy()
    assert np.isfinite(got).all()
    assert _rel(got, u_one) <= 1e-11
