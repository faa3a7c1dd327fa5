"""One rank of tests/test_rccl_gpu.py::test_two_ranks (started by torch.distributed.run, one process per GPU): the sharded operator,
Chebyshev iteration and cg_eigs through the library's RCCL hooks (csrc/d4est_hip_comm.hip) against the one-rank operator on the whole
mesh.  The unique id travels over a gloo group (host); every byte of the data path goes through ncclSend / ncclRecv / ncclAllReduce."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from disco4est_amd import Plan, mesh as M, parallel as P
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group(backend="gloo")
    comm = P.RcclComm(rank, world)
    assert comm.lib.d4est_hip_comm_nccl_count(comm.handle) == world
    fails = []
    for level, deg, hanging in ((2, 3, False), (1, 7, False), (1, 9, False), (1, 2, True)):
        if hanging:
            refine = np.zeros(8, dtype=bool); refine[[0, 7]] = True
            mk = lambda **kw: M.HangingBrickMesh(level, refine, deg, **kw)
        else:
            mk = lambda **kw: M.BrickMesh(level, deg, **kw)
        full = mk()
        parts = P.partition_by_dofs(full.deg_global, world)
        mp = M.SineMap(0.04)
        m = mk(first=parts[rank][0], count=parts[rank][1])
        J, rst = m.geometry(mp); sides = m.build_sides(mp)
        stream = torch.cuda.current_stream()
        plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=stream)
        plan.set_geometry(J, rst); plan.set_faces(sides, 10.0, 0)
        x = P.attach_rccl(plan, m, sides, parts, comm)
        ug = full.field(mp)
        rg = M.splitmix64_uniform(5, full.local_nodes) - 0.5
        s0 = m.global_nodal_offset
        u = torch.from_numpy(np.ascontiguousarray(ug[s0:s0 + m.local_nodes])).to(dev)
        rhs = torch.from_numpy(np.ascontiguousarray(rg[s0:s0 + m.local_nodes])).to(dev)
        Au, r = torch.empty_like(u), torch.empty_like(u)
        plan.apply_lhs(u, Au)
        uc = torch.zeros_like(u); Ac = torch.empty_like(u)
        plan.cheby_iterate(uc, rhs, Ac, r, 3, 1.0, 60.0, 1)
        ue = torch.zeros_like(u); Ae = torch.empty_like(u)
        bound, _ = plan.cg_eigs(ue, rhs, Ae, 5, 1)
        torch.cuda.synchronize()
        gathered = [None] * world if rank == 0 else None
        dist.gather_object((Au.cpu().numpy(), uc.cpu().numpy(), ue.cpu().numpy(), bound, x.count()), gathered, dst=0)
        if rank == 0:
            Jf, rstf = full.geometry(mp); sf = full.build_sides(mp)
            pf = Plan(full.deg, full.deg_quad, full.nodal_stride, full.quad_stride, 0, stream=stream)
            pf.set_geometry(Jf, rstf); pf.set_faces(sf, 10.0, 0)
            uf = torch.from_numpy(ug).to(dev); rf = torch.from_numpy(rg).to(dev)
            ref = torch.empty_like(uf); pf.apply_aij(uf, ref)
            ucf = torch.zeros_like(uf); Acf = torch.empty_like(uf); rr = torch.empty_like(uf)
            pf.cheby_iterate(ucf, rf, Acf, rr, 3, 1.0, 60.0, 1)
            uef = torch.zeros_like(uf); bref, _ = pf.cg_eigs(uef, rf, Acf, 5, 1)
            rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
            e_lhs = rel(np.concatenate([g[0] for g in gathered]), ref.cpu().numpy())
            e_chb = rel(np.concatenate([g[1] for g in gathered]), ucf.cpu().numpy())
            e_cg = rel(np.concatenate([g[2] for g in gathered]), uef.cpu().numpy())
            e_b = max(abs(g[3] - bref) / abs(bref) for g in gathered)
            line = "level %d p %d hanging %s: apply_lhs %.2e cheby %.2e cg_eigs u %.2e bound %.2e, exchanges %s" % (level, deg, hanging, e_lhs, e_chb, e_cg, e_b, [g[4] for g in gathered])
            print(line, flush=True)
            if not (e_lhs <= 1e-12 and e_chb <= 1e-11 and e_cg <= 1e-9 and e_b <= 1e-9 and (world == 1 or all(g[4] > 0 for g in gathered))):
                fails.append(line)
            pf.destroy()
        x.destroy(); plan.destroy()
    comm.destroy()
    ok = [not fails]
    dist.broadcast_object_list(ok, src=0)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("two-rank RCCL check:", "ok" if ok[0] else "MISMATCH", flush=True)
    sys.exit(0 if ok[0] else 1)


if __name__ == "__main__":
    main()
