"""hp-multigrid transfer kernels: prolong / restrict / project time, algorithmic bytes and fraction of the HBM roof.
tools/time_transfer.py <coarse level> <degH> <degh> [h|p]   (h: eight children per coarse element; p: same elements, degh > degH)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Transfer
level, dH, dh = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "h"
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()
n = 8 ** level
if kind == "h":
    T = Transfer(np.ones(n, np.int32), np.full(n, dH, np.int32), np.full(8 * n, dh, np.int32), stream=st)
else:
    degh = np.zeros(8 * n, np.int32); degh[0::8] = dh
    T = Transfer(np.zeros(n, np.int32), np.full(n, dH, np.int32), degh, stream=st)
xc = torch.rand(T.coarse_nodes, dtype=torch.float64, device=dev)
xf = torch.rand(T.fine_nodes, dtype=torch.float64, device=dev)
oc, of = torch.empty_like(xc), torch.empty_like(xf)


def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


byts = 8.0 * (T.coarse_nodes + T.fine_nodes)      # one vector in, one out
print("%s-transfer, %d coarse elements, degH %d <-> degh %d: %.3f / %.3f MDoF, %.1f MB per transfer" %
      (kind, n, dH, dh, T.coarse_nodes * 1e-6, T.fine_nodes * 1e-6, byts * 1e-6))
for name, fn in (("prolong", lambda: T.prolong(xc, of)), ("restrict", lambda: T.restrict(xf, oc)), ("project", lambda: T.project(xf, oc))):
    us = t(fn)
    print("  %-8s %7.1f us  %6.0f GB/s  frac %.2f of 8 TB/s  (%.1f GDoF/s fine)" % (name, us, byts / us * 1e-3, byts / us * 1e-3 / 8000.0, T.fine_nodes / us * 1e-3))
