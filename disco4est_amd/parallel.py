"""Multi-GPU sharding of the element hot path (host side).

One process per GPU.  Elements are split into Morton-contiguous chunks balanced by DoF count -- what
``p4est_partition`` does for the reference (src/driver.c:136-151) -- and the only data exchanged per operator
apply are FACE TRACES of the elements adjacent to a partition boundary (u and du/dr_{0,1,2} on the shared face,
4 N^2 doubles) instead of the reference's whole mirror elements ((p+1)^3 doubles per field,
src/Mesh/d4est_ghost_data.c:143-256).  Neighbour traffic is point-to-point (``batch_isend_irecv`` = grouped
ncclSend/ncclRecv over xGMI on the nccl backend; gloo on CPU for the tests); scalar reductions of the
Lanczos/CG estimate use ``all_reduce``.

The schedule needs no metadata exchange: both sides of a partition boundary enumerate the shared faces in the same
canonical order (global element id of the SENDER, then face).
"""
import numpy as np


def partition_by_dofs(deg_global, world):
    """Morton-contiguous [first, count) ranges with (nearly) equal DoF counts; returns list of (first, count)."""
    w = (np.asarray(deg_global, dtype=np.int64) + 1) ** 3
    cum = np.concatenate([[0], np.cumsum(w)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        target = total * r / world
        k = int(np.searchsorted(cum, target, side="left"))
        k = min(max(k, cuts[-1]), len(w))
        cuts.append(k)
    cuts.append(len(w))
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def owner_of(parts, n_global):
    owner = np.empty(n_global, dtype=np.int32)
    for r, (f, c) in enumerate(parts):
        owner[f:f + c] = r
    return owner


class TraceSchedule:
    """Per-peer send / receive block lists for one rank.

    send[peer] / recv[peer]: arrays (offset, length) in canonical order, built from the rank's own side list only
    (mesh.build_sides): every side s = (e, f) whose (+) neighbour is a ghost owned by ``peer`` contributes
      * one SEND block: my mortar-node trace of side s        (trace_offset(s), block_len(s)) -- the peer's ghost data
      * one RECV block: the ghost's trace on its face f_p      (ghost_trace_offset(s), block_len(s)) -- my ghost data
    Both ends sort by (global element id of the SENDER, its face): no metadata is exchanged.
    """

    def __init__(self, mesh, sides, parts, trace_offset, ghost_trace_offset, block_len, side_blocks=None, reorient=None):
        """trace_offset / ghost_trace_offset / block_len take (side, sub); side_blocks(side) = number of own blocks (4 on the
        big side of a hanging face).  On meshes with hanging faces the counterpart of a big side's block i is small element
        i's block; the counterpart of a small side's block is the big element's block reorient(f_m, f_p, orientation, c)."""
        owner = owner_of(parts, mesh.global_elements)
        send, recv = {}, {}
        nbr = sides["side_nbr"]
        hang = sides.get("side_hang")
        for s in range(6 * mesh.n_elements):
            e, f = divmod(s, 6)
            nb = 1 if side_blocks is None else int(side_blocks(s))
            h = 0 if hang is None else int(hang[s])
            for sub in range(nb):
                goff = int(ghost_trace_offset(s, sub))
                if goff < 0:
                    continue
                ref = int(sides["side_nbr4"][4 * s + sub]) if h == 1 else int(nbr[s])
                gid = int(sides["ghost_global_ids"][-(ref + 2)])
                peer = int(owner[gid])
                f_p = int(sides["side_nbr_face"][s])
                sender_sub = 0
                if h == 2:   # the (+) element is the big one: it sends the sub-block that faces me
                    c = int(sides["side_sub"][s])
                    o = int(sides["side_orientation"][s])
                    sender_sub = c if reorient is None else int(reorient(f, f_p, o, c))
                my_gid = int(mesh.elements[e]) if hasattr(mesh, "elements") else mesh.first + e
                ln = int(block_len(s, sub))
                send.setdefault(peer, []).append((my_gid, f, sub, int(trace_offset(s, sub)), ln))
                recv.setdefault(peer, []).append((gid, f_p, sender_sub, goff, ln))
        self.peers = sorted(set(send) | set(recv))
        self.send = {p: np.array([(o, l) for _, _, _, o, l in sorted(send[p])], dtype=np.int64).reshape(-1, 2) for p in self.peers}
        self.recv = {p: np.array([(o, l) for _, _, _, o, l in sorted(recv[p])], dtype=np.int64).reshape(-1, 2) for p in self.peers}
        self.send_len = {p: int(self.send[p][:, 1].sum()) for p in self.peers}
        self.recv_len = {p: int(self.recv[p][:, 1].sum()) for p in self.peers}

    def pack_lists(self, which, peer):
        """(buffer_offsets, packed_offsets, lengths) of one peer's blocks"""
        blocks = (self.send if which == "send" else self.recv)[peer]
        packed = np.concatenate([[0], np.cumsum(blocks[:, 1])[:-1]]).astype(np.int64) if len(blocks) else np.zeros(0, np.int64)
        return blocks[:, 0].copy(), packed, blocks[:, 1].astype(np.int32)


class ElementSchedule:
    """Whole-element exchange lists of one rank for an extended mesh = own elements followed by a ghost layer (global ids
    ``ghost_gids``): what d4est_ghost_data_exchange moves for the Schwarz smoother's residual (src/Solver/d4est_solver_schwarz.c:197-203)
    and, run backwards, what d4est_solver_schwarz_transfer_ghost_data_and_add_corrections returns (…_transfer_ghost_data.c:60-176).

    ``needed_by[peer]``: global ids of OWN elements that lie in ``peer``'s ghost layer.  send / recv blocks are (offset, length) into the
    element-ordered vector of the extended mesh, both ends sorted by global id: no metadata is exchanged.  Same interface as
    TraceSchedule, so TraceExchange runs it; ``reversed()`` swaps the roles (ghost copies travel back to their owners)."""

    def __init__(self, ext_mesh, n_own, parts, needed_by):
        owner = owner_of(parts, ext_mesh.global_elements)
        n3 = (ext_mesh.deg.astype(np.int64) + 1) ** 3
        send, recv = {}, {}
        for peer, gids in needed_by.items():
            for g in sorted(int(v) for v in gids):
                l = int(ext_mesh._g2l[g])
                assert 0 <= l < n_own
                send.setdefault(int(peer), []).append((int(ext_mesh.nodal_stride[l]), int(n3[l])))
        for l in range(n_own, ext_mesh.n_elements):          # ghost layer, already sorted by global id
            g = int(ext_mesh.elements[l])
            recv.setdefault(int(owner[g]), []).append((int(ext_mesh.nodal_stride[l]), int(n3[l])))
        self._finish(send, recv)

    def _finish(self, send, recv):
        self.peers = sorted(set(send) | set(recv))
        arr = lambda d, p: np.array(d.get(p, []), dtype=np.int64).reshape(-1, 2)
        self.send = {p: arr(send, p) for p in self.peers}
        self.recv = {p: arr(recv, p) for p in self.peers}
        self.send_len = {p: int(self.send[p][:, 1].sum()) for p in self.peers}
        self.recv_len = {p: int(self.recv[p][:, 1].sum()) for p in self.peers}

    def reversed(self):
        r = ElementSchedule.__new__(ElementSchedule)
        r._finish({p: [tuple(b) for b in self.recv[p]] for p in self.peers}, {p: [tuple(b) for b in self.send[p]] for p in self.peers})
        return r

    pack_lists = TraceSchedule.pack_lists


class TraceExchange:
    """Runs the schedule: pack (HIP copy_blocks on the plan's stream) -> point-to-point -> unpack.

    ``transport`` moves the packed per-peer buffers: DistTransport (torch.distributed) or an in-process
    transport used by the single-GPU tests.  ``copy_blocks(n, src, src_off, dst, dst_off, len)`` is the
    device block copy (Plan.copy_blocks)."""

    def __init__(self, schedule, transport, copy_blocks, device):
        import torch
        self.s = schedule
        self.transport = transport
        self.copy_blocks = copy_blocks
        self.dev = device
        self.send_buf, self.recv_buf, self.idx = {}, {}, {}
        for p in schedule.peers:
            self.send_buf[p] = torch.empty(schedule.send_len[p], dtype=torch.float64, device=device)
            self.recv_buf[p] = torch.empty(schedule.recv_len[p], dtype=torch.float64, device=device)
            so, sp, sl = schedule.pack_lists("send", p)
            ro, rp, rl = schedule.pack_lists("recv", p)
            t = lambda a: torch.from_numpy(a).to(device)
            self.idx[p] = (t(so), t(sp), t(sl), t(ro), t(rp), t(rl))
        self._pending = None

    def begin(self, trace):
        for p in self.s.peers:
            so, sp, sl, _, _, _ = self.idx[p]
            self.copy_blocks(len(sl), trace, so, self.send_buf[p], sp, sl)
        self._pending = self.transport.start(self.send_buf, self.recv_buf)

    def end(self, ghost_trace):
        self.transport.finish(self._pending)
        self._pending = None
        for p in self.s.peers:
            _, _, _, ro, rp, rl = self.idx[p]
            self.copy_blocks(len(rl), self.recv_buf[p], rp, ghost_trace, ro, rl)


class DistTransport:
    """torch.distributed point-to-point: one grouped isend/irecv per neighbouring rank
    (ncclGroupStart/ncclSend/ncclRecv/ncclGroupEnd on the nccl = RCCL backend)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        # nccl / RCCL enqueues its transfers behind the current stream; a host-side backend (gloo: CPU tests, single-GPU rehearsals
        # with device tensors) touches the buffers from the host, so the pack kernels must have finished before the send is posted
        self.host_side = dist.is_initialized() and dist.get_backend(group) != "nccl"

    def _device_sync(self, bufs):
        if self.host_side:
            import torch
            if any(t.is_cuda for t in bufs.values()):
                torch.cuda.synchronize()

    def start(self, send_buf, recv_buf):
        dist = self.dist
        self._device_sync(send_buf)
        ops = []
        for p in sorted(recv_buf):
            if recv_buf[p].numel():
                ops.append(dist.P2POp(dist.irecv, recv_buf[p], p, group=self.group))
        for p in sorted(send_buf):
            if send_buf[p].numel():
                ops.append(dist.P2POp(dist.isend, send_buf[p], p, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def finish(self, reqs):
        for r in reqs or []:
            r.wait()

    def allreduce_sum(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)


def plan_schedule(plan, mesh, sides, parts):
    """TraceSchedule with the block offsets of a Plan whose faces are set"""
    lib, h = plan.lib, plan.handle
    return TraceSchedule(mesh, sides, parts, lambda s, b: lib.d4est_hip_plan_trace_offset_sub(h, s, b),
                         lambda s, b: lib.d4est_hip_plan_ghost_trace_offset_sub(h, s, b),
                         lambda s, b: lib.d4est_hip_plan_trace_block_len_sub(h, s, b),
                         side_blocks=lambda s: lib.d4est_hip_plan_side_blocks(h, s), reorient=lib.d4est_hip_reorient_face_order)


def side_block_layout(sides):
    """(trace_offset, ghost_trace_offset, block_len) per side computed on the host from a side list -- the same layout
    the C library uses (4 T doubles per side in side order; ghost blocks in side order) -- for CPU-only tests"""
    stride = np.asarray(sides["side_mortar_stride"], dtype=np.int64)
    total = int(sides["total_mortar_nodes"])
    T = np.diff(np.concatenate([stride, [total]]))
    off = 4 * stride
    is_ghost = np.asarray(sides["side_nbr"]) <= -2
    goff = np.full(len(stride), -1, dtype=np.int64)
    goff[is_ghost] = np.concatenate([[0], np.cumsum(4 * T[is_ghost])[:-1]]) if is_ghost.any() else []
    return off, goff, 4 * T


def side_block_layout_hp(mesh, sides):
    """Per (side, sub) block layout of a plan WITH hanging faces, computed on the host exactly as the C library does
    (mortar records in side order; a big side owns 4 blocks; T from the larger quadrature degree of the two elements
    of a mortar; ghost blocks in record order).  Returns (n_blocks[side], offset(side, sub), ghost_offset(side, sub), length(side, sub))."""
    ne = mesh.n_elements
    degq_l = np.asarray(mesh.deg_quad)
    degq_g = np.asarray(sides["ghost_deg_quad"])
    dq = lambda ref: int(degq_l[ref]) if ref >= 0 else int(degq_g[-(ref + 2)])
    nblk = np.ones(6 * ne, dtype=np.int32)
    off, goff, ln = {}, {}, {}
    q = g = 0
    for s in range(6 * ne):
        e = s // 6
        h = int(sides["side_hang"][s])
        nb = 4 if h == 1 else 1
        nblk[s] = nb
        for sub in range(nb):
            ref = int(sides["side_nbr4"][4 * s + sub]) if h == 1 else int(sides["side_nbr"][s])
            pq = int(degq_l[e]) if ref == -1 else max(int(degq_l[e]), dq(ref))
            T = (pq + 1) ** 2
            off[(s, sub)] = q
            ln[(s, sub)] = 4 * T
            q += 4 * T
            if ref <= -2:
                goff[(s, sub)] = g
                g += 4 * T
            else:
                goff[(s, sub)] = -1
    return nblk, off, goff, ln, q, g


def attach(plan, mesh, sides, parts, transport, device):
    """Wire a Plan (faces already set) to a transport: installs the exchange / allreduce hooks used by
    apply_lhs, cheby_iterate and cg_eigs.  Returns the TraceExchange (keep it alive)."""
    import ctypes
    import torch
    sched = plan_schedule(plan, mesh, sides, parts)
    ex = TraceExchange(sched, transport, plan.copy_blocks, device)
    n_trace, n_ghost = int(plan.trace_size), int(plan.ghost_trace_size)

    def view(ptr, n):
        # wrap a raw device pointer handed over by the C library as a tensor (no copy)
        class _Holder:
            pass
        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}
        return torch.as_tensor(h, device=device)

    import contextlib

    def on_plan_stream():
        # the collectives order themselves behind torch's CURRENT stream; the library's kernels run on the plan's stream
        ts = getattr(plan, "torch_stream", None)
        return torch.cuda.stream(ts) if ts is not None else contextlib.nullcontext()

    def exchange(phase, trace_ptr, ghost_ptr):
        with on_plan_stream():
            if phase == 0:
                ex.begin(view(trace_ptr, n_trace))
            else:
                ex.end(view(ghost_ptr, n_ghost))

    def allreduce(ptr, n):
        with on_plan_stream():
            transport.allreduce_sum(view(ptr, n))

    plan.set_comm(exchange if n_ghost > 0 else None, allreduce if hasattr(transport, "allreduce_sum") else None)
    return ex


# ---- RCCL transport in C (csrc/d4est_hip_comm.hip): no Python in the per-apply path --------------------------------------------
class RcclComm:
    """One RCCL communicator of the library (d4est_hip_comm_create = ncclCommInitRank), one rank per GPU.  The unique id is created on
    rank 0 and handed to the other ranks by ``broadcast(bytes or None) -> bytes`` (torch.distributed here; MPI_Bcast in a d4est build)."""

    def __init__(self, rank, world, broadcast=None):
        import ctypes
        from . import capi
        self.lib = capi.load_library()
        n = self.lib.d4est_hip_comm_unique_id_bytes()
        buf = ctypes.create_string_buffer(n)
        if rank == 0:
            self.lib.d4est_hip_comm_get_unique_id(buf)
        uid = bytes(buf.raw)
        if world > 1:
            if broadcast is None:
                broadcast = torch_broadcast_bytes
            uid = broadcast(uid if rank == 0 else None)
        self._uid = ctypes.create_string_buffer(uid, n)
        self.rank, self.world = rank, world
        self.handle = self.lib.d4est_hip_comm_try_create(self._uid, rank, world)
        if not self.handle:
            raise RuntimeError("RCCL communicator could not be created (rank %d of %d)" % (rank, world))

    def destroy(self):
        if self.handle:
            self.lib.d4est_hip_comm_destroy(self.handle)
            self.handle = None


def torch_broadcast_bytes(payload):
    """broadcast a bytes object from rank 0 over the default torch.distributed group"""
    import torch.distributed as dist
    box = [payload]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def flatten_schedule(sched):
    """per-peer block lists of a TraceSchedule / ElementSchedule as the flat arrays d4est_hip_plan_set_rccl_exchange takes"""
    peers = np.asarray(sched.peers, dtype=np.int32)
    sf, rf = [0], [0]
    so, sl, ro, rl = [], [], [], []
    for p in sched.peers:
        so.append(sched.send[p][:, 0]); sl.append(sched.send[p][:, 1])
        ro.append(sched.recv[p][:, 0]); rl.append(sched.recv[p][:, 1])
        sf.append(sf[-1] + len(sched.send[p])); rf.append(rf[-1] + len(sched.recv[p]))
    cat = lambda a, dt: np.ascontiguousarray(np.concatenate(a) if a else np.zeros(0), dtype=dt)
    return (peers, np.asarray(sf, dtype=np.int32), cat(so, np.int64), cat(sl, np.int32),
            np.asarray(rf, dtype=np.int32), cat(ro, np.int64), cat(rl, np.int32))


def schedule_summary(sched):
    """{peer: (doubles sent, doubles received)}: what the two ends of every pair must agree on before the first exchange"""
    return {int(p): (int(sched.send_len[p]), int(sched.recv_len[p])) for p in sched.peers}


def check_schedules_match(summaries):
    """summaries[r] = schedule_summary of rank r (all-gathered): rank a sends to b exactly what b expects from a.  A mismatch would
    hang the grouped ncclSend / ncclRecv round, so it is checked on the host before the transport is wired."""
    for a, sa in enumerate(summaries):
        for b, (sent, recvd) in sa.items():
            back = summaries[b].get(a)
            if back is None or back[1] != sent or back[0] != recvd:
                return False, "rank %d <-> %d: %r vs %r" % (a, b, (sent, recvd), back)
    return True, ""


def attach_rccl(plan, mesh, sides, parts, comm):
    """Wire a Plan (faces set) to the RCCL transport: d4est_hip_plan_set_rccl_exchange.  Returns the exchange handle wrapper."""
    import ctypes
    sched = plan_schedule(plan, mesh, sides, parts)
    peers, sf, so, sl, rf, ro, rl = flatten_schedule(sched)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    h = plan.lib.d4est_hip_plan_set_rccl_exchange(plan.handle, comm.handle, len(peers), vp(peers), vp(sf), vp(so), vp(sl), vp(rf), vp(ro), vp(rl))

    class _Exchange:
        pass
    x = _Exchange()
    x.handle, x.schedule, x.lib = h, sched, plan.lib
    x.count = lambda: plan.lib.d4est_hip_rccl_exchange_count(h)
    x.send_doubles = plan.lib.d4est_hip_rccl_exchange_send_doubles(h)
    x.recv_doubles = plan.lib.d4est_hip_rccl_exchange_recv_doubles(h)
    x.destroy = lambda: plan.lib.d4est_hip_rccl_exchange_destroy(h)
    return x
