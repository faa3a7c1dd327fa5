"""CPU tests of the multi-rank host logic: DoF-balanced Morton partition, the metadata-free trace schedule and the
point-to-point exchange itself over torch.distributed/gloo with world_size 2 and 3 (the N > 1 path of bench/apply_lhs
uses exactly this code with the nccl backend and HIP pack/unpack kernels)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np_copy_blocks(n, src, src_off, dst, dst_off, length):
    s, d = src.numpy(), dst.numpy()
    for b in range(n):
        so, do, ln = int(src_off[b]), int(dst_off[b]), int(length[b])
        d[do:do + ln] = s[so:so + ln]


def _encode(gid, f, length):
    """trace block content that identifies (global element, face, entry)"""
    return gid * 1000.0 + f * 100.0 + np.arange(length) / float(length)


def test_partition_by_dofs():
    from disco4est_amd import parallel as P
    deg = np.array([3] * 32 + [7] * 32)
    parts = P.partition_by_dofs(deg, 4)
    assert parts[0][0] == 0 and sum(c for _, c in parts) == 64
    for (f0, c0), (f1, _) in zip(parts[:-1], parts[1:]):
        assert f0 + c0 == f1
    w = (deg + 1) ** 3
    loads = [w[f:f + c].sum() for f, c in parts]
    assert max(loads) <= 1.3 * (w.sum() / 4)          # balanced by DoFs, not by element count
    assert parts[0][1] > parts[-1][1]                  # low-p elements are cheaper -> more of them per rank
    assert P.partition_by_dofs(np.full(8, 2), 1) == [(0, 8)]
    assert (P.owner_of(parts, 64)[:parts[0][1]] == 0).all()


def _worker(rank, world, port, level, deg_spec, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from disco4est_amd import mesh as M, parallel as P
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_global = 8 ** level
        deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n_global)])
        parts = P.partition_by_dofs(deg_global, world)
        first, count = parts[rank]
        m = M.BrickMesh(level, deg_global, first=first, count=count)
        sides = m.build_sides(None)
        toff, goff, blen = P.side_block_layout(sides)
        sched = P.TraceSchedule(m, sides, parts, lambda s, b: toff[s], lambda s, b: goff[s], lambda s, b: blen[s])
        # local trace buffer with identifying content
        trace = np.zeros(int((toff + blen).max()))
        for e in range(m.n_elements):
            for f in range(6):
                s = 6 * e + f
                trace[toff[s]:toff[s] + blen[s]] = _encode(first + e, f, blen[s])
        n_ghost_doubles = int(blen[goff >= 0].sum())
        ghost = np.full(n_ghost_doubles, np.nan)
        ex = P.TraceExchange(sched, P.DistTransport(), _np_copy_blocks, torch.device("cpu"))
        tt, gt = torch.from_numpy(trace), torch.from_numpy(ghost)
        ex.begin(tt)
        ex.end(gt)
        # every ghost side must now hold the owner's block of (ghost element, its face)
        nbr = sides["side_nbr"]
        checked = 0
        for s in np.nonzero(nbr <= -2)[0]:
            g = -(int(nbr[s]) + 2)
            f_p = int(sides["side_nbr_face"][s])
            np.testing.assert_array_equal(ghost[goff[s]:goff[s] + blen[s]], _encode(int(sides["ghost_global_ids"][g]), f_p, blen[s]))
            checked += 1
        assert not np.isnan(ghost).any()
        # scalar reduction used by cg_eigs
        t = torch.tensor([float(rank + 1), 2.0], dtype=torch.float64)
        P.DistTransport().allreduce_sum(t)
        assert t[0].item() == world * (world + 1) / 2 and t[1].item() == 2.0 * world
        q.put((rank, "ok", checked, len(sched.peers)))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), str(exc)))
    finally:
        dist.destroy_process_group()


def _worker_hp(rank, world, port, level, pattern, deg_spec, q):
    """hanging faces across ranks: every block that arrives must be the SENDER's block (global element, face, sub)"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from disco4est_amd import mesh as M, parallel as P
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        refine = np.zeros(8 ** level, dtype=bool)
        refine[np.asarray(pattern)] = True
        n_global = M.HangingBrickMesh(level, refine, 2).global_elements
        deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n_global)])
        parts = P.partition_by_dofs(deg_global, world)
        first, count = parts[rank]
        m = M.HangingBrickMesh(level, refine, deg_global, first=first, count=count)
        sides = m.build_sides(None)
        nblk, off, goff, ln, n_trace, n_ghost = P.side_block_layout_hp(m, sides)
        sched = P.TraceSchedule(m, sides, parts, lambda s, b: off[(s, b)], lambda s, b: goff[(s, b)], lambda s, b: ln[(s, b)],
                                side_blocks=lambda s: nblk[s])
        trace = np.zeros(n_trace)
        for (s, sub), o in off.items():
            e, f = divmod(s, 6)
            trace[o:o + ln[(s, sub)]] = _encode(first + e, 10 * f + sub, ln[(s, sub)])   # identity (element, face, sub)
        ghost = np.full(max(n_ghost, 1), np.nan)
        ex = P.TraceExchange(sched, P.DistTransport(), _np_copy_blocks, torch.device("cpu"))
        tt, gt = torch.from_numpy(trace), torch.from_numpy(ghost)
        ex.begin(tt)
        ex.end(gt)
        checked = crossing = 0
        for (s, sub), go in goff.items():
            if go < 0:
                continue
            h = int(sides["side_hang"][s])
            ref = int(sides["side_nbr4"][4 * s + sub]) if h == 1 else int(sides["side_nbr"][s])
            gid = int(sides["ghost_global_ids"][-(ref + 2)])
            f_p = int(sides["side_nbr_face"][s])
            sender_sub = int(sides["side_sub"][s]) if h == 2 else 0     # orientation 0 inside one tree
            np.testing.assert_array_equal(ghost[go:go + ln[(s, sub)]], _encode(gid, 10 * f_p + sender_sub, ln[(s, sub)]))
            checked += 1
            crossing += int(h != 0)
        if n_ghost:
            assert not np.isnan(ghost).any()
        q.put((rank, "ok", checked, crossing))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), str(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,level,pattern,deg_spec", [(2, 1, [0, 5, 6], [2]), (3, 1, [1, 2, 4, 7], [2, 3])])
def test_trace_exchange_gloo_hanging(world, level, pattern, deg_spec):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_hp, args=(r, world, port, level, pattern, deg_spec, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r
    assert sum(r[2] for r in res) > 0
    assert sum(r[3] for r in res) > 0, "the partition must cut at least one hanging face"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,level,deg_spec", [(2, 1, [3]), (2, 2, [2, 3, 4]), (3, 2, [3])])
def test_trace_exchange_gloo(world, level, deg_spec):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, level, deg_spec, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r
    assert sum(r[2] for r in res) > 0          # faces really crossed the partition boundary
    assert all(r[3] >= 1 for r in res)


def _worker_elements(rank, world, port, level, deg_spec, q):
    """whole-element exchange of the Schwarz smoother (forward: residual of the ghost layer; backward: corrections to the owners)"""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from disco4est_amd import mesh as M, parallel as P
    from disco4est_amd.schwarz import ghost_layer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_global = 8 ** level
        deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n_global)])
        parts = P.partition_by_dofs(deg_global, world)
        own, ghosts, needed_by = ghost_layer(level, parts, rank)
        m = M.BrickMesh(level, deg_global, elements=np.concatenate([own, ghosts]))
        sched = P.ElementSchedule(m, own.size, parts, needed_by)
        n3 = (m.deg.astype(np.int64) + 1) ** 3
        x = np.full(m.local_nodes, np.nan)
        for l in range(own.size):
            x[m.nodal_stride[l]:m.nodal_stride[l] + n3[l]] = _encode(int(m.elements[l]), 0, int(n3[l]))
        ex = P.TraceExchange(sched, P.DistTransport(), _np_copy_blocks, torch.device("cpu"))
        t = torch.from_numpy(x)
        ex.begin(t)
        ex.end(t)
        for l in range(m.n_elements):          # every ghost-layer element now holds its owner's data
            np.testing.assert_array_equal(x[m.nodal_stride[l]:m.nodal_stride[l] + n3[l]], _encode(int(m.elements[l]), 0, int(n3[l])))
        # backward: every ghost copy returns (global id, sender rank); the owner sees one block per peer that holds a copy
        back = P.TraceExchange(sched.reversed(), P.DistTransport(), _np_copy_blocks, torch.device("cpu"))
        y = np.full(m.local_nodes, np.nan)
        for l in range(own.size, m.n_elements):
            y[m.nodal_stride[l]:m.nodal_stride[l] + n3[l]] = _encode(int(m.elements[l]), rank, int(n3[l]))
        ty = torch.from_numpy(y)
        back.begin(ty)
        back.transport.finish(back._pending)
        hits = 0
        for p in back.s.peers:
            z = np.full(m.local_nodes, np.nan)
            _, _, _, ro, rp, rl = back.idx[p]
            _np_copy_blocks(len(rl), back.recv_buf[p], rp, torch.from_numpy(z), ro, rl)
            for g in needed_by.get(p, ()):
                l = int(m._g2l[g])
                np.testing.assert_array_equal(z[m.nodal_stride[l]:m.nodal_stride[l] + n3[l]], _encode(int(g), p, int(n3[l])))
                hits += 1
            assert np.isnan(z).sum() == m.local_nodes - sum(int(n3[m._g2l[g]]) for g in needed_by.get(p, ()))
        q.put((rank, "ok", int(ghosts.size), hits))
    except Exception as exc:  # pragma: no cover
        import traceback
        q.put((rank, "fail", traceback.format_exc(), str(exc)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,level,deg_spec", [(2, 2, [2]), (3, 2, [2, 3])])
def test_element_exchange_gloo(world, level, deg_spec):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_elements, args=(r, world, port, level, deg_spec, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] == "ok", r
    assert all(r[2] > 0 and r[3] > 0 for r in res)


@pytest.mark.parametrize("world,level,deg_spec", [(2, 2, [3]), (4, 2, [2, 3, 4]), (3, 1, [7])])
def test_schedules_of_all_ranks_agree_and_flatten(world, level, deg_spec):
    """The N > 1 preflight of bench.py (parallel.schedule_summary / check_schedules_match) and the flat arrays the RCCL transport takes
    (parallel.flatten_schedule -> d4est_hip_plan_set_rccl_exchange): every pair of ranks agrees on what travels between them, the flat
    form holds exactly the schedule's blocks, and a tampered summary is caught (it would hang the grouped send / recv round)."""
    from disco4est_amd import mesh as M, parallel as P
    n_global = 8 ** level
    deg_global = np.array([deg_spec[i % len(deg_spec)] for i in range(n_global)])
    parts = P.partition_by_dofs(deg_global, world)
    scheds = []
    for first, count in parts:
        m = M.BrickMesh(level, deg_global, first=first, count=count)
        sides = m.build_sides(None)
        toff, goff, blen = P.side_block_layout(sides)
        scheds.append(P.TraceSchedule(m, sides, parts, lambda s, b: toff[s], lambda s, b: goff[s], lambda s, b: blen[s]))
    summaries = [P.schedule_summary(s) for s in scheds]
    ok, why = P.check_schedules_match(summaries)
    assert ok, why
    total_sent = sum(v[0] for s in summaries for v in s.values())
    total_recv = sum(v[1] for s in summaries for v in s.values())
    assert total_sent == total_recv > 0
    for r, sched in enumerate(scheds):
        peers, sf, so, sl, rf, ro, rl = P.flatten_schedule(sched)
        assert list(peers) == list(sched.peers) and r not in peers
        assert sf[0] == 0 and rf[0] == 0 and sf[-1] == len(so) == len(sl) and rf[-1] == len(ro) == len(rl)
        for i, p in enumerate(peers):
            assert int(sl[sf[i]:sf[i + 1]].sum()) == summaries[r][int(p)][0]
            assert int(rl[rf[i]:rf[i + 1]].sum()) == summaries[r][int(p)][1]
            np.testing.assert_array_equal(so[sf[i]:sf[i + 1]], sched.send[p][:, 0])
            np.testing.assert_array_equal(ro[rf[i]:rf[i + 1]], sched.recv[p][:, 0])
        assert so.dtype == np.int64 and sl.dtype == np.int32 and peers.dtype == np.int32
    # a rank that expects one double more than its peer sends: refused
    bad = [dict(s) for s in summaries]
    a = 0
    b = next(iter(bad[a]))
    bad[a][b] = (bad[a][b][0], bad[a][b][1] + 1)
    ok, why = P.check_schedules_match(bad)
    assert not ok and "rank" in why
    # a rank missing from its peer's list: refused
    bad = [dict(s) for s in summaries]
    del bad[b][a]
    assert not P.check_schedules_match(bad)[0]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_weak_scaling_domain_of_sub_cubes(world):
    """bench.py's weak-scaling full-operator secondary: the domain is the box of the first `world` level-L sub-cubes of the level-(L+1)
    Morton sequence (BrickMesh(domain=...)), one sub-cube per rank; faces towards the rest of the cube are domain boundary.  Every pair
    of ranks agrees on the exchange, a rank has ghost sides on exactly the faces it shares with owned sub-cubes, and world = 8 is the
    whole cube."""
    from disco4est_amd import mesh as M, parallel as P
    L, deg = 1, 2                       # sub-cubes of 8 elements in the level-2 cube
    per = 8 ** L
    parts = [(r * per, per) for r in range(world)]
    scheds, n_ghost_sides, n_bnd_sides = [], [], []
    for r in range(world):
        m = M.BrickMesh(L + 1, deg, first=r * per, count=per, domain=world * per)
        sides = m.build_sides(None)
        nbr = np.asarray(sides["side_nbr"])
        assert not np.any(sides["ghost_global_ids"] >= world * per)      # nothing outside the domain is ever a neighbour
        n_ghost_sides.append(int((nbr <= -2).sum())); n_bnd_sides.append(int((nbr == -1).sum()))
        toff, goff, blen = P.side_block_layout(sides)
        scheds.append(P.TraceSchedule(m, sides, parts, lambda s, b: toff[s], lambda s, b: goff[s], lambda s, b: blen[s]))
    ok, why = P.check_schedules_match([P.schedule_summary(s) for s in scheds])
    assert ok, why
    # sub-cube r sits at the (x, y, z) bits of r; it shares a 2 x 2-element face with r ^ 1, r ^ 2, r ^ 4 where those are in the domain
    for r in range(world):
        shared = sum(1 for b in (1, 2, 4) if (r ^ b) < world)
        assert n_ghost_sides[r] == 4 * shared and n_bnd_sides[r] == 4 * (6 - shared)
        assert n_ghost_sides[r] + n_bnd_sides[r] == 24                  # the 6 outer faces of a 2 x 2 x 2 sub-cube: 4 sides each
        assert sorted(scheds[r].peers) == sorted((r ^ b) for b in (1, 2, 4) if (r ^ b) < world)
    if world == 8:
        full = M.BrickMesh(L + 1, deg).build_sides(None)
        assert sum(n_bnd_sides) == int((np.asarray(full["side_nbr"]) == -1).sum())
