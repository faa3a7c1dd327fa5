"""The DEFAULT kernel choices at sizes where they differ from what the small parity meshes get (tuning key 11 left alone):
a shard of 2048 elements with ghost sides runs the one-wavefront direct face kernel fed by the trace exchange."""
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


@pytest.mark.parametrize("level,deg", [(4, 1), (1, 9), (2, 11)])
def test_large_shard_default_is_the_direct_kernel_with_exchanged_ghost_sides(gpu, hiplib, oracle, monkeypatch, level, deg):
    """level 4, p = 1: 4096 elements in two shards of 2048; apply_lhs on rank 0 through the C exchange hooks (rank 1's traces served
    in-process) against the oracle's single-rank operator.  p = 9, 11: the multi-wave direct kernel (the default at every size) with
    ghost (+) sides that read the exchanged mortar-node blocks."""
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    monkeypatch.delenv("D4EST_HIP_FACE_DIRECT", raising=False)
    deg_global = np.full(8 ** level, deg)
    mg = M.BrickMesh(level, deg_global)
    Jg, rstg = mg.geometry(None); sg = mg.build_sides(None); ug = mg.field()
    ref = oracle.apply_aij(mg, Jg, rstg, sg, ug, nthreads=8)
    half = 8 ** level // 2
    parts = [(0, half), (half, half)]
    mb = _Mailbox()
    objs = []
    for first, count in parts:
        m = M.BrickMesh(level, deg_global, first=first, count=count)
        J, rst = m.geometry(None); s = m.build_sides(None)
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
        plan.set_geometry(J, rst); plan.set_faces(s)
        # (the name says what the plan CAN do; with ghost sides the volume kernel stays separate so that it overlaps the exchange)
        assert plan.face_path() == "direct+volume" and plan.ghost_trace_size > 0
        objs.append((m, s, plan))
    m1, s1, p1 = objs[1]
    ex1 = P.attach(p1, m1, s1, parts, _LocalTransport(1, mb), gpu)
    u1 = torch.from_numpy(m1.field()).to(gpu)
    tr1 = torch.empty(p1.trace_size, dtype=torch.float64, device=gpu)
    p1.compute_face_traces(u1, tr1)
    ex1.begin(tr1)                                # rank 1's traces of u wait in the mailbox
    m0, s0, p0 = objs[0]
    ex0 = P.attach(p0, m0, s0, parts, _LocalTransport(0, mb), gpu)
    u0 = torch.from_numpy(m0.field()).to(gpu)
    Au0 = torch.full_like(u0, float("nan"))
    p0.apply_lhs(u0, Au0)                         # traces -> exchange -> volume kernel -> direct face kernel with the ghost blocks
    got = Au0.cpu().numpy()
    assert np.isfinite(got).all()
    assert np.abs(got - ref[:m0.local_nodes]).max() <= 1e-12 * np.abs(ref).max()
    assert ex0 is not None
    for _, _, p in objs:
        p.destroy()
