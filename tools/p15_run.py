"""p = 15 stiffness apply, 8192 elements (33.6 MDoF, 2.1 GB per apply), general path: the workload of tools/pmc_p15.sh.
argv[1] = value of tuning key 4 (1 vector-ALU kernel, 2 matrix-core kernel)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M
bigp = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_el = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dev = torch.device("cuda:0")
m = M.BrickMesh(5, 15, count=n_el)
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry_brick(np.ones(m.n_elements, dtype=np.int32), 32.0, [0, 1, 0, 1, 0, 1.0])
plan.set_tuning(7, 0); plan.set_tuning(4, bigp)
x = torch.rand(m.local_nodes, dtype=torch.float64, device=dev); y = torch.empty_like(x)
for _ in range(3): plan.apply_stiffness_matrix(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): plan.apply_stiffness_matrix(x, y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("%s: %d elements %.1f us %.1f GDoF/s" % (plan.last_kernel(), n_el, ms * 1e3, m.local_nodes / ms / 1e6))
