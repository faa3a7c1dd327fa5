"""Timing of the batched additive Schwarz smoother on uniform bricks (setup, one iterate, per-sweep cost)."""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from disco4est_amd import Plan, mesh as M  # noqa: E402
from disco4est_amd.schwarz import Schwarz  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--level", type=int, default=3)
ap.add_argument("--deg", type=int, default=7)
ap.add_argument("--overlap", type=int, default=3)
ap.add_argument("--iter", type=int, default=10)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda:0")
m = M.BrickMesh(a.level, a.deg)
t0 = time.time()
J, rst = m.geometry(None); sides = m.build_sides(None)
t1 = time.time()
sz = Schwarz(m, sides, J, rst, a.overlap, a.iter, 1e-300, 1e-300)
torch.cuda.synchronize()
t2 = time.time()
md = sz.metadata
print(f"level {a.level} p={a.deg} overlap {a.overlap}: {md.num_subdomains} subdomains, {md.num_elements} subdomain elements, "
      f"field over subdomains {sz.nodal_size * 8 / 1e6:.1f} MB, restricted {md.restricted_nodal_size * 8 / 1e6:.1f} MB, "
      f"zero ghost trace {sz.plan.ghost_trace_size * 8 / 1e6:.1f} MB; mesh setup {t1 - t0:.1f} s, schwarz setup {t2 - t1:.1f} s")
print("subdomain plan face path:", sz.plan.face_path())
u = torch.zeros(m.local_nodes, dtype=torch.float64, device=dev)
r = torch.from_numpy(M.splitmix64_uniform(1, m.local_nodes) - 0.5).to(dev)
sz.iterate(u, r)
torch.cuda.synchronize()
for _ in range(a.reps):
    u.zero_()
    t = time.time()
    sweeps = sz.iterate(u, r)
    torch.cuda.synchronize()
    dt = time.time() - t
    print(f"iterate: {sweeps} sweeps in {dt * 1e3:.2f} ms = {dt / sweeps * 1e3:.3f} ms/sweep; "
          f"{md.num_elements * sweeps * (a.deg + 1) ** 3 / dt / 1e9:.2f} G subdomain-DoF/s")
x = torch.empty(sz.nodal_size, dtype=torch.float64, device=dev); y = torch.empty_like(x)
sz.restrict_field(r, x)
torch.cuda.synchronize()
t = time.time()
for _ in range(10):
    sz.apply_over_subdomains(x, y)
torch.cuda.synchronize()
print(f"apply_over_subdomains: {(time.time() - t) / 10 * 1e3:.3f} ms")
