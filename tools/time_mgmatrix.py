"""The multigrid matrix operator on a coarse level: apply_lhs with the zeroth-order term as dense element blocks and as the Galerkin
chain, against the Laplacian alone.  tools/time_mgmatrix.py <coarse level> <deg> [h|p]
  h: the fine level is the brick one level finer at the same degree (eight children per coarse element)
  p: the fine level is the same brick at degree deg + 2 (p-coarsening)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from disco4est_amd import Plan, Transfer, mesh as M
level, deg = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "h"
dev = torch.device("cuda:0")
st = torch.cuda.current_stream()


def make(m):
    J, rst = m.geometry(None); sides = m.build_sides(None)
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=st)
    p.set_geometry(J, rst); p.set_tuning(7, 0); p.set_faces(sides)
    return p


mc = M.BrickMesh(level, deg)
n_c = mc.n_elements
if kind == "h":
    mf = M.BrickMesh(level + 1, deg)
    T = Transfer(np.ones(n_c, np.int32), np.full(n_c, deg, np.int32), np.full(8 * n_c, deg, np.int32), stream=st)
else:
    mf = M.BrickMesh(level, deg + 2)
    degh = np.zeros(8 * n_c, np.int32); degh[0::8] = deg + 2
    T = Transfer(np.zeros(n_c, np.int32), np.full(n_c, deg, np.int32), degh, stream=st)
pc, pf = make(mc), make(mf)
coeff = 1.0 + torch.rand(mf.local_nodes_quad, dtype=torch.float64, device=dev)
pf.set_lhs_coefficient(coeff)
fine_blocks = torch.empty(pf.matrix_nodes(), dtype=torch.float64, device=dev)
blocks = torch.empty(pc.matrix_nodes(), dtype=torch.float64, device=dev)


def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


t_setup_f = t(lambda: pf.compute_weighted_mass_blocks(coeff, fine_blocks), 2)
t_setup_c = t(lambda: T.galerkin_blocks(fine_blocks, blocks), 2)
u = torch.from_numpy(mc.field()).to(dev); Au = torch.empty_like(u); A2 = torch.empty_like(u)
t_aij = t(lambda: pc.apply_aij(u, Au))
pc.set_lhs_element_blocks(blocks)
t_blk = t(lambda: pc.apply_lhs(u, Au))
pc.set_lhs_galerkin_chain([T], pf)
t_chn = t(lambda: pc.apply_lhs(u, A2))
err = float((Au - A2).abs().max() / Au.abs().max())
gb = blocks.numel() * 8e-9
print("coarse level %d p %d (%d elements, %.3f MDoF), fine = %s: blocks %.2f GB" % (level, deg, n_c, mc.local_nodes * 1e-6, "level+1" if kind == "h" else "p+2", gb))
print("  setup: fine blocks (QUAD_COMPUTE_MATRIX, %.2f GB) %.1f ms | Galerkin restriction P^T M P %.1f ms" % (fine_blocks.numel() * 8e-9, t_setup_f * 1e-3, t_setup_c * 1e-3))
print("  apply_aij %.1f us | apply_lhs blocks %.1f us (term %.1f us = %.0f GB/s of block stream) | apply_lhs chain %.1f us (term %.1f us) | blocks vs chain rel %.1e"
      % (t_aij, t_blk, t_blk - t_aij, gb / ((t_blk - t_aij) * 1e-6), t_chn, t_chn - t_aij, err))
