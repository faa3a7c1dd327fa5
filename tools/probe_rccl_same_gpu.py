"""Probe: can 2 ranks that share ONE GPU form an RCCL communicator on this box? (decides how the RCCL exchange can be rehearsed)"""
import os, sys, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.full((4,), float(rank + 1), device="cuda")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    a = torch.full((8,), float(rank), device="cuda"); b = torch.empty(8, device="cuda")
    ops = [dist.P2POp(dist.irecv, b, 1 - rank), dist.P2POp(dist.isend, a, 1 - rank)]
    for r in dist.batch_isend_irecv(ops): r.wait()
    torch.cuda.synchronize()
    print("rank", rank, "allreduce", t.tolist(), "p2p", b.tolist(), flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank", rank, "FAILED:", repr(e)[:500], flush=True)
    sys.exit(3)
