"""Phase timeline of operator_mw_kernel from s_memtime stamps (diagnostic build: tools/build_variant.sh mw_stamps d4est_hip_direct_mw.hip
"-DD4EST_HIP_MW_ONLY=12 -DD4EST_HIP_MWD_STAMPS=1"; run with D4EST_HIP_LIBRARY=<that library>): tools/stamps_mw.py <level> <deg>"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
level, deg = int(sys.argv[1]), int(sys.argv[2])
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None); sides = m.build_sides(None); u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0); plan.set_tuning(11, 2); plan.set_faces(sides)
du = torch.from_numpy(u).to(dev); Au = torch.empty_like(du)
stamps = torch.zeros(m.n_elements * 40, dtype=torch.float64, device=dev)   # reinterpreted as uint64 by the diagnostic kernel
for _ in range(3): plan.apply_aij(du, Au, stamps)
torch.cuda.synchronize()
t = stamps.cpu().numpy().view(np.uint64).reshape(m.n_elements, 40).astype(np.int64)
names = {0: "start", 1: "volume term done"}
for d in range(3):
    for k, nm in ((2, "lines + nodal fields"), (3, "staged + rows read"), (4, "pass 1"), (5, "pass 2 products"), (6, "SIPG (2 faces)"), (7, "lift 1"), (8, "lift 2")):
        names[k + 10 * d] = "dir %d: %s" % (d, nm)
names[32] = "last line update"; names[33] = "A u stored"
order = sorted(names)
vol = {34: "volume: element image + S1", 35: "volume: S2 / S3 (forward s, t)", 36: "volume: quadrature stage (metric stream)", 37: "volume: S5 / S6 (backward t, s)", 38: "volume: S7 (backward r)"}
t0 = t[:, 0].min()
print("level %d p %d: %d workgroups; shader cycles (s_memtime), medians over workgroups" % (level, deg, m.n_elements))
life = np.median(t[:, 33] - t[:, 0])
prev = 0
for k in order[1:]:
    dt = np.median(t[:, k] - t[:, prev])
    print("  %-32s %8.0f cycles  (%4.1f %% of the workgroup's %d-cycle lifetime)" % (names[k], dt, 100 * dt / life, life))
    prev = k
prev = 0
for k in sorted(vol):
    dt = np.median(t[:, k] - t[:, prev])
    print("  %-44s %8.0f cycles" % (vol[k], dt))
    prev = k
print("(s_memtime counts shader cycles; the counters of different XCDs have different origins, so only differences inside a workgroup are used)")
