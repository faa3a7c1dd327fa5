"""The C RCCL transport (csrc/d4est_hip_comm.hip) on real hardware.  A one-GPU box can only hold a one-rank communicator (RCCL refuses
two ranks on one device): the library's own ncclCommInitRank, and the whole per-apply path -- pack kernel, grouped ncclSend / ncclRecv
on the communicator's stream, event hand-over, unpack kernel -- with the rank as its own peer.  Wherever TWO OR MORE GPUs are visible
test_two_ranks starts two fresh processes (one per GPU) and holds the sharded operator, Chebyshev loop and cg_eigs over real
ncclSend / ncclRecv / ncclAllReduce to the one-rank results.  The multi-rank schedules are also covered over gloo (tests/test_parallel.py)
and by bench.py --gpus N."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


@pytest.mark.parametrize("deg,hanging", [(3, False), (7, False), (9, False), (2, True)])
def test_rccl_self_exchange_through_apply_lhs(gpu, hiplib, deg, hanging):
    """Shard A of a two-part mesh whose other part is ALSO 'owned' by rank 0: every send block comes back as the matching receive
    block (equal degrees: block k sent == block k expected), so apply_lhs over RCCL must equal apply_aij fed with a ghost buffer
    assembled on the host from the shard's own traces -- bit for bit."""
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    if hanging:
        refine = np.zeros(8, dtype=bool); refine[[0, 7]] = True
        mk = lambda **kw: M.HangingBrickMesh(1, refine, deg, **kw)
        total = mk().global_elements
        parts = [(0, total // 2), (total // 2, total - total // 2)]
    else:
        mk = lambda **kw: M.BrickMesh(1, deg, **kw)
        parts = [(0, 4), (4, 4)]
    m = mk(first=parts[0][0], count=parts[0][1])
    mp = M.SineMap(0.04)
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, 10.0, 0)
    assert plan.ghost_trace_size > 0
    comm = P.RcclComm(0, 1)
    assert hiplib.d4est_hip_comm_size(comm.handle) == 1
    x = P.attach_rccl(plan, m, sides, [(0, m.global_elements)], comm)      # everything owned by rank 0: the only peer is rank 0
    assert list(x.schedule.peers) == [0] and x.send_doubles == x.recv_doubles > 0
    du = _t(m.field(mp), gpu)
    dAu = torch.full_like(du, float("nan"))
    plan.apply_lhs(du, dAu)
    plan.apply_lhs(du, dAu)              # buffers are reused: a second round must not race with the first
    torch.cuda.synchronize()
    assert x.count() == 2
    # expected ghost buffer: block k of the send list lands in block k of the receive list
    trace = torch.empty(int(plan.trace_size), dtype=torch.float64, device=gpu)
    plan.compute_face_traces(du, trace)
    tr = trace.cpu().numpy()
    gt = np.zeros(int(plan.ghost_trace_size))
    snd, rcv = x.schedule.send[0], x.schedule.recv[0]
    assert len(snd) == len(rcv)
    for (so, sl), (ro, rl) in zip(snd, rcv):
        assert sl == rl
        gt[ro:ro + rl] = tr[so:so + sl]
    ref = torch.full_like(du, float("nan"))
    plan.apply_aij(du, ref, _t(gt, gpu))
    np.testing.assert_array_equal(dAu.cpu().numpy(), ref.cpu().numpy())
    # Chebyshev and cg_eigs run through the same hooks (allreduce is the identity on one rank)
    rhs = torch.zeros_like(du); r = torch.empty_like(du); u1 = du.clone()
    plan.cheby_iterate(u1, rhs, dAu, r, 3, 1.0, 30.0, 1)
    b, _ = plan.cg_eigs(du.clone(), rhs, dAu, 4)
    torch.cuda.synchronize()
    assert np.isfinite(u1.cpu().numpy()).all() and np.isfinite(b) and x.count() > 2
    x.destroy()
    plan.destroy()
    comm.destroy()


def test_rccl_sendrecv_and_allreduce_single_rank(gpu, hiplib):
    import ctypes
    import torch
    from disco4est_amd import Plan, mesh as M, parallel as P
    m = M.BrickMesh(0, 2)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    comm = P.RcclComm(0, 1)
    a = torch.arange(1000, dtype=torch.float64, device=gpu)
    b = torch.zeros_like(a)
    peer = np.zeros(1, dtype=np.int32)
    first = np.array([0, 1000], dtype=np.int64)
    vp = lambda arr: arr.ctypes.data_as(ctypes.c_void_p)
    hiplib.d4est_hip_comm_sendrecv(comm.handle, plan.handle, 1, vp(peer), ctypes.c_void_p(a.data_ptr()), vp(first),
                                   ctypes.c_void_p(b.data_ptr()), vp(first))
    s = torch.tensor([1.5, 2.5], dtype=torch.float64, device=gpu)
    hiplib.d4est_hip_comm_allreduce_sum(comm.handle, plan.handle, ctypes.c_void_p(s.data_ptr()), 2)
    torch.cuda.synchronize()
    assert torch.equal(a, b) and s.tolist() == [1.5, 2.5]
    plan.destroy()
    comm.destroy()


def test_two_ranks(hiplib):
    """two ranks on two GPUs through the real RCCL hooks (skipped on one-GPU boxes): tests/helpers/rccl_two_rank_child.py, started as
    fresh child processes -- the children never inherit this process's GPU state, and nothing here replaces a running program"""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:          # (counting devices does not initialise the GPU)
        pytest.skip("needs two GPUs; this box has %d" % torch.cuda.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(root, "tests", "helpers", "rccl_two_rank_child.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    print(out.stdout[-4000:], out.stderr[-3000:])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "two-rank RCCL check: ok" in out.stdout
