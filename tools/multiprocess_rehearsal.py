"""Two real processes (torch.distributed, one rank per process) running the sharded Schwarz smoother and the sharded operator through
DistTransport; rank 0 compares the assembled result with the single-rank smoother.  Rehearsal on ONE GPU:
  D4EST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/multiprocess_rehearsal.py
(on a multi-GPU node the backend is nccl = RCCL and every rank takes its own device)."""
import os, sys
import numpy as np, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import mesh as M, parallel as P
from disco4est_amd.schwarz import Schwarz, SchwarzShard

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)) % max(torch.cuda.device_count(), 1))
torch.cuda.set_device(dev)
dist.init_process_group(backend=os.environ.get("D4EST_BACKEND", "nccl"))
level, rs, iters = int(os.environ.get("D4EST_LEVEL", 2)), 2, 5
pmin = int(os.environ.get("D4EST_DEG", 2))
refine = None
if os.environ.get("D4EST_HANGING"):          # locally refined brick: hanging faces cross the rank boundaries
    refine = np.zeros(8 ** level, dtype=bool)
    refine[[0, 8 ** level // 2 + 1, 8 ** level - 1]] = True
    n_el = M.HangingBrickMesh(level, refine, 2).n_elements
    make = lambda deg, **kw: M.HangingBrickMesh(level, refine, deg, **kw)
else:
    n_el = 8 ** level
    make = lambda deg, **kw: M.BrickMesh(level, deg, **kw)
# mixed degrees pmin, pmin + 1 (the two-phase face kernels) unless D4EST_UNIFORM is set (one degree: the direct face kernel, with
# ghost sides fed by the exchange)
deg_global = np.array([pmin + (0 if os.environ.get("D4EST_UNIFORM") else i % 2) for i in range(n_el)], dtype=np.int32)
mp = M.SineMap(0.04)
parts = P.partition_by_dofs(deg_global, world)
sh = SchwarzShard(level, deg_global, parts, rank, mp, rs, iters, 1e-15, 1e-15, P.DistTransport(), dev, refine=refine)
mg = make(deg_global)
u0 = M.splitmix64_uniform(81, mg.local_nodes) - 0.5
r = M.splitmix64_uniform(82, mg.local_nodes) - 0.5
first, count = parts[rank]
lo = int(mg.global_nodal_stride[first]); hi = lo + sh.own_nodes
u = torch.from_numpy(u0[lo:hi].copy()).to(dev)
sh.iterate(u, torch.from_numpy(r[lo:hi].copy()).to(dev))
torch.cuda.synchronize()
pieces = [None] * world
dist.all_gather_object(pieces, (lo, hi, u.cpu().numpy()))
if rank == 0:
    got = np.empty(mg.local_nodes)
    for a, b, x in pieces:
        got[a:b] = x
    Jg, rstg = mg.geometry(mp); sg = mg.build_sides(mp)
    single = Schwarz(mg, sg, Jg, rstg, rs, iters, 1e-15, 1e-15)
    ur = torch.from_numpy(u0).to(dev)
    single.iterate(ur, torch.from_numpy(r).to(dev))
    err = np.abs(got - ur.cpu().numpy()).max() / np.abs(ur.cpu().numpy() - u0).max()
    print("world %d: sharded Schwarz iterate vs single rank: rel err %.2e %s" % (world, err, "ok" if err <= 1e-10 else "MISMATCH"), flush=True)
# ---- the sharded operator through the C library's exchange hooks (d4est_hip_plan_set_comm -> TraceExchange -> DistTransport)
from disco4est_amd import Plan
ms = make(deg_global, first=first, count=count)
Js, rsts = ms.geometry(mp); ss = ms.build_sides(mp)
plan = Plan(ms.deg, ms.deg_quad, ms.nodal_stride, ms.quad_stride, 0)
plan.set_geometry(Js, rsts); plan.set_faces(ss)
ex = P.attach(plan, ms, ss, parts, P.DistTransport(), dev)
us = torch.from_numpy(u0[lo:hi].copy()).to(dev); Aus = torch.empty_like(us)
plan.apply_lhs(us, Aus)
torch.cuda.synchronize()
pieces = [None] * world
dist.all_gather_object(pieces, (lo, hi, Aus.cpu().numpy()))
if rank == 0:
    got = np.empty(mg.local_nodes)
    for a, b, x in pieces:
        got[a:b] = x
    pg = Plan(mg.deg, mg.deg_quad, mg.nodal_stride, mg.quad_stride, 0)
    pg.set_geometry(Jg, rstg); pg.set_faces(sg)
    ug = torch.from_numpy(u0).to(dev); Aug = torch.empty_like(ug)
    pg.apply_aij(ug, Aug)
    err = np.abs(got - Aug.cpu().numpy()).max() / np.abs(Aug.cpu().numpy()).max()
    print("world %d: sharded apply_lhs vs single rank: rel err %.2e %s" % (world, err, "ok" if err <= 1e-12 else "MISMATCH"), flush=True)
# ---- cg_eigs (three scalar reductions per iteration through the allreduce hook) and five Chebyshev iterations
rhs_s = torch.from_numpy(r[lo:hi].copy()).to(dev)
uz = torch.zeros_like(us)
bound, _ = plan.cg_eigs(uz, rhs_s, Aus, 8)
uc = torch.from_numpy(u0[lo:hi].copy()).to(dev); rc = torch.empty_like(uc)
plan.cheby_iterate(uc, rhs_s, Aus, rc, 5, bound / 30.0, bound, 1)
torch.cuda.synchronize()
pieces = [None] * world
dist.all_gather_object(pieces, (lo, hi, uc.cpu().numpy(), bound))
if rank == 0:
    got = np.empty(mg.local_nodes)
    for a, b, x, _ in pieces:
        got[a:b] = x
    rg = torch.from_numpy(r).to(dev)
    bg, _ = pg.cg_eigs(torch.zeros_like(ug), rg, Aug, 8)
    ucg = torch.from_numpy(u0).to(dev); rcg = torch.empty_like(ucg)
    pg.cheby_iterate(ucg, rg, Aug, rcg, 5, bg / 30.0, bg, 1)
    err = np.abs(got - ucg.cpu().numpy()).max() / np.abs(ucg.cpu().numpy()).max()
    eb = max(abs(p_[3] - bg) for p_ in pieces) / bg
    print("world %d: sharded cg_eigs bound %r (single rank %r) rel err %.2e, Chebyshev iterate rel err %.2e %s" % (
        world, pieces[-1][3], bg, eb, err, "ok" if max(eb, err) <= 1e-10 else "MISMATCH"), flush=True)
dist.barrier()
dist.destroy_process_group()
