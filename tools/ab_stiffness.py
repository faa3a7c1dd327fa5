"""A/B timing of stiffness-kernel tuning variants in ONE process, interleaved rounds
(cdna_hip_programming.md section 5.4 rule 24).  Usage: python tools/ab_stiffness.py [level] [deg] [k=v,k=v ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, mesh as M  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 4
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 7
# each remaining argument is one configuration: "key=value,key=value"
cfgs = sys.argv[3:] if len(sys.argv) > 3 else ["0=0,1=0", "0=1,1=0", "0=0,1=1", "0=1,1=1"]
vals = cfgs
m = M.BrickMesh(level, deg)
J, rst = m.geometry(None)
u = m.field()
dev = torch.device("cuda:0")
plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
plan.set_geometry(J, rst); plan.set_tuning(7, 0)  # general path unless overridden
du = torch.from_numpy(u).to(dev)
out = torch.empty_like(du)
ref = None
res = {v: [] for v in vals}
steps = 50
for rnd in range(12):
    for v in vals:
        for kv in v.split(","):
            k_, v_ = kv.split("=")
            plan.set_tuning(int(k_), int(v_))
        for _ in range(5):
            plan.apply_stiffness_matrix(du, out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(steps):
            plan.apply_stiffness_matrix(du, out)
        e1.record()
        torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / steps * 1e3)
        if ref is None and not any(t in v for t in ("1=9", "1=7", "1=8")):
            ref = out.clone()
        else:
            assert any(t in v for t in ("1=9", "1=7", "1=8")) or (out - ref).abs().max().item() <= 1e-12 * ref.abs().max().item()
for v in vals:
    t = sorted(res[v])
    med, mn = t[len(t) // 2], t[0]
    print("level=%d p=%d tune{%s}: median %.2f us  min %.2f us  -> %.1f GDoF/s (median), %.1f GB/s algorithmic" %
          (level, deg, v, med, mn, m.local_nodes / med / 1e3, 64.0 * m.local_nodes / med / 1e3))
