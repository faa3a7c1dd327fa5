"""GPU parity of the multigrid MATRIX OPERATOR (verdict row a14): the zeroth-order term of a linearised problem on the coarse levels of
an hp-multigrid hierarchy, as the reference's smoother applies it (dense Galerkin-restricted element blocks:
Solver/d4est_solver_multigrid_matrix_operator.c:6-48, :160-245; dGMath/d4est_operators.c:608-667;
Problems/ConstantDensityStar/constant_density_star_fcns.h:485-527, :806-850).

Checked against the oracle with dense P^T M P blocks, for BOTH device forms (element blocks; the matrix-free Galerkin chain), on every
coarse level of 2- and 3-level hierarchies: curved elements, mixed p with eight children of different degrees, a locally refined (hanging)
fine mesh; apply_lhs, 5 Chebyshev iterations, cg_eigs, and the Schwarz subdomain operator with the block term."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _plan(m, mp, prefactor=10.0):
    from disco4est_amd import Plan
    J, rst = m.geometry(mp)
    sides = m.build_sides(mp)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry(J, rst)
    plan.set_faces(sides, prefactor, 0)
    return plan, J, rst, sides


def _hierarchy(kind):
    """list of meshes, finest first, and the transfer item lists between consecutive levels (coarse <- fine)"""
    from disco4est_amd import mesh as M
    if kind == "hp3":
        # level 2 brick (64 elements, p = 2..4 scattered so that the eight children of a parent differ) -> h-coarsened level 1 brick
        # (degH = the smallest child degree, d4est_solver_multigrid_callbacks.h:52-75) -> p-coarsened (deg - 1, :9-20)
        deg2 = (2 + (np.arange(64) * 7 + (np.arange(64) // 8)) % 3).astype(np.int32)
        deg1 = deg2.reshape(8, 8).min(axis=1).astype(np.int32)
        deg0 = np.maximum(deg1 - 1, 1).astype(np.int32)
        meshes = [M.BrickMesh(2, deg2, deg_quad_inc=1), M.BrickMesh(1, deg1, deg_quad_inc=1), M.BrickMesh(1, deg0, deg_quad_inc=1)]
        items = [(np.ones(8, np.int32), deg1, deg2.copy()),
                 (np.zeros(8, np.int32), deg0, np.ascontiguousarray(np.stack([deg1] + [np.zeros(8, np.int32)] * 7, axis=1).reshape(-1)))]
        return meshes, items
    if kind == "hanging2":
        # fine: level-1 brick with octants 1 and 6 refined (22 elements, hanging faces), p = 2 / 3; coarse: the level-1 brick -- the refined
        # octants coarsen (eight children -> parent), the others are copied (the reference's third case) or lose one degree
        refine = np.zeros(8, dtype=bool)
        refine[[1, 6]] = True
        n_el = 8 - 2 + 16
        degf = (2 + (np.arange(n_el) * 5) % 2).astype(np.int32)
        mf = M.HangingBrickMesh(1, refine, degf, deg_quad_inc=0)
        hrefine, degH, degh = [], [], []
        k = 0
        for b in range(8):
            dh = np.zeros(8, np.int32)
            if refine[b]:
                dh[:] = degf[k:k + 8]
                hrefine.append(1); degH.append(int(dh.min())); k += 8
            else:
                dh[0] = degf[k]
                hrefine.append(0); degH.append(int(degf[k]) - (1 if b % 2 == 0 else 0)); k += 1   # p-coarsened or copied
            degh.append(dh)
        degH = np.array(degH, np.int32)
        mc = M.BrickMesh(1, degH, deg_quad_inc=0)
        return [mf, mc], [(np.array(hrefine, np.int32), degH, np.concatenate(degh))]
    raise ValueError(kind)


@pytest.mark.parametrize("unfused", [False, True])
@pytest.mark.parametrize("kind", ["hp3", "hanging2"])
def test_coarse_level_operator_blocks_and_chain(gpu, hiplib, oracle, kind, unfused, monkeypatch):
    """unfused = False: a chain of ONE transfer runs the fused kernel (galerkin_fast_kernel: composite prolong-and-interpolate operators,
    only the fine coefficient streamed); True (D4EST_HIP_CHAIN_UNFUSED): prolong, fine weighted mass, prolong-transpose as separate kernels
    -- what longer chains always do"""
    import torch
    from disco4est_amd import Transfer, mesh as M
    if unfused:
        monkeypatch.setenv("D4EST_HIP_CHAIN_UNFUSED", "1")
    else:
        monkeypatch.delenv("D4EST_HIP_CHAIN_UNFUSED", raising=False)
    mp = M.SineMap(0.04)
    meshes, items = _hierarchy(kind)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu)
    levels = [_plan(m, mp) for m in meshes]
    transfers = [Transfer(h, dH, dh) for (h, dH, dh) in items]
    for t, (mf, mc) in zip(transfers, zip(meshes[:-1], meshes[1:])):
        assert t.fine_nodes == mf.local_nodes and t.coarse_nodes == mc.local_nodes
    mf = meshes[0]
    pf, Jf = levels[0][0], levels[0][1]
    # f'(u0) at the fine quadrature nodes: positive (like the reference's 5 pi rho psi^4 terms), varying
    coeff = 0.5 + 2.0 * M.splitmix64_uniform(17, mf.local_nodes_quad)
    dcoeff = T(coeff)
    pf.set_lhs_coefficient(dcoeff)

    # ---- blocks on the finest level: QUAD_COMPUTE_MATRIX for every element
    fine_ref = oracle.mg_matrix_setup(mf, Jf, coeff)
    assert fine_ref.size == pf.matrix_nodes()
    dfine = torch.full((fine_ref.size,), float("nan"), dtype=torch.float64, device=gpu)
    pf.compute_weighted_mass_blocks(dcoeff, dfine)
    assert np.abs(dfine.cpu().numpy() - fine_ref).max() <= RTOL * np.abs(fine_ref).max()

    # ---- restriction of the blocks, level by level (exact Galerkin product and the reference's literal window)
    blocks_ref, blocks_dev = [fine_ref], [dfine]
    for lvl, (t, (h, dH, dh)) in enumerate(zip(transfers, items)):
        ref = oracle.mg_matrix_restriction(h, dH, dh, blocks_ref[-1])
        out = torch.full((ref.size,), float("nan"), dtype=torch.float64, device=gpu)
        t.galerkin_blocks(blocks_dev[-1], out)
        assert np.abs(out.cpu().numpy() - ref).max() <= RTOL * np.abs(ref).max(), lvl
        lit_ref = oracle.mg_matrix_restriction(h, dH, dh, blocks_ref[-1], literal_window=True)
        lit = torch.full((ref.size,), float("nan"), dtype=torch.float64, device=gpu)
        t.galerkin_blocks(blocks_dev[-1], lit, literal_window=True)
        assert np.abs(lit.cpu().numpy() - lit_ref).max() <= RTOL * np.abs(lit_ref).max(), lvl
        blocks_ref.append(ref)
        blocks_dev.append(out)

    # ---- every coarse level: apply_lhs, 5 Chebyshev iterations and cg_eigs with the term as blocks and as the chain
    for lvl in range(1, len(meshes)):
        m = meshes[lvl]
        plan, J, rst, sides = levels[lvl]
        oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=4)
        oracle.set_lhs_coefficient(None)
        oracle.set_lhs_element_blocks(blocks_ref[lvl])
        u = m.field(mp)
        rhs = M.splitmix64_uniform(23 + lvl, m.local_nodes) - 0.5
        ref_lhs = oracle.apply_lhs(u)
        lap = oracle.apply_aij(m, J, rst, sides, u)
        assert np.abs(ref_lhs - lap).max() > 1e-3 * np.abs(lap).max()      # the term is not negligible in this test
        lmax = 1.1 * oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 8)[0]
        lmin = lmax / 30.0
        ref_u, ref_r = oracle.cheby_iterate(np.zeros(m.local_nodes), rhs, 5, lmin, lmax, 1)
        ref_bound, ref_ucg = oracle.cg_eigs(np.zeros(m.local_nodes), rhs, 6)
        for form in ("blocks", "chain"):
            if form == "blocks":
                plan.set_lhs_element_blocks(blocks_dev[lvl])
            else:
                plan.set_lhs_galerkin_chain(list(reversed(transfers[:lvl])), pf)
            du = T(u)
            dAu = torch.full_like(du, float("nan"))
            plan.apply_lhs(du, dAu)
            err = np.abs(dAu.cpu().numpy() - ref_lhs).max() / np.abs(ref_lhs).max()
            assert err <= RTOL, (kind, lvl, form, err)
            x = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
            Au, r = torch.empty_like(x), torch.empty_like(x)
            plan.cheby_iterate(x, T(rhs), Au, r, 5, lmin, lmax, 1)
            assert np.abs(x.cpu().numpy() - ref_u).max() <= 1e-11 * np.abs(ref_u).max(), (kind, lvl, form)
            assert np.abs(r.cpu().numpy() - ref_r).max() <= 1e-11 * np.abs(rhs).max(), (kind, lvl, form)
            x.zero_()
            bound, _ = plan.cg_eigs(x, T(rhs), Au, 6, 1)
            assert abs(bound - ref_bound) <= 1e-9 * abs(ref_bound), (kind, lvl, form)
            assert np.abs(x.cpu().numpy() - ref_ucg).max() <= 1e-9 * np.abs(ref_ucg).max(), (kind, lvl, form)
            # apply_aij stays the Laplacian alone
            plan.apply_aij(du, dAu)
            assert np.abs(dAu.cpu().numpy() - lap).max() <= RTOL * np.abs(lap).max()
        # switching forms: the coefficient setter replaces the chain, NULL switches the term off
        plan.set_lhs_galerkin_chain([], None)
        plan.apply_lhs(du, dAu)
        assert np.abs(dAu.cpu().numpy() - lap).max() <= RTOL * np.abs(lap).max()
    oracle.set_lhs_element_blocks(None)
    for t in transfers:
        t.destroy()
    for lv in levels:
        lv[0].destroy()


def test_schwarz_subdomain_operator_with_blocks(gpu, hiplib, oracle):
    """the additive Schwarz smoother on a coarse level: the block term is part of the subdomain operator (copy k of mesh element e reads
    e's block), as d4est_solver_schwarz_laplacian_ext_apply_over_subdomain applies the problem's apply_lhs"""
    import torch
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import Schwarz
    mp = M.SineMap(0.03)
    m = M.BrickMesh(1, 3)
    plan, J, rst, sides = _plan(m, mp)
    coeff = 0.5 + 2.0 * M.splitmix64_uniform(29, m.local_nodes_quad)
    blocks = oracle.mg_matrix_setup(m, J, coeff)       # any SPD blocks serve; these are the level's own weighted mass matrices
    dblocks = torch.from_numpy(blocks).to(gpu)
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=2)
    oracle.set_lhs_coefficient(None)
    oracle.set_lhs_element_blocks(blocks)
    sz = Schwarz(m, sides, J, rst, 2, 6, 1e-15, 1e-15)
    sz.set_lhs_element_blocks(dblocks, m)
    r = M.splitmix64_uniform(31, m.local_nodes) - 0.5
    u_ref, it_ref, _ = oracle.schwarz_iterate(sz.metadata, np.zeros(m.local_nodes), r, 6, 1e-15, 1e-15)
    u = torch.zeros(m.local_nodes, dtype=torch.float64, device=gpu)
    sz.iterate(u, torch.from_numpy(r).to(gpu))
    assert np.abs(u.cpu().numpy() - u_ref).max() <= 1e-9 * np.abs(u_ref).max()
    # and it differs from the pure Laplacian's correction
    oracle.set_lhs_element_blocks(None)
    u_lap, _, _ = oracle.schwarz_iterate(sz.metadata, np.zeros(m.local_nodes), r, 6, 1e-15, 1e-15)
    assert np.abs(u_lap - u_ref).max() > 1e-4 * np.abs(u_ref).max()
    sz.destroy()
    plan.destroy()


def test_block_term_at_size(gpu, hiplib):
    """level 3, p = 7 (512 elements, 1.07 GB of blocks): the block matvec against a torch bmm of the same blocks, the chain against the
    blocks built from it -- size-independent identities, no oracle"""
    import torch
    from disco4est_amd import Plan, Transfer, mesh as M
    mc = M.BrickMesh(3, 7)
    mf = M.BrickMesh(3, 7)     # same mesh as "fine" level through an identity transfer: chain == coefficient form
    mp = M.SineMap(0.02)
    plan, J, rst, sides = _plan(mc, mp)
    coeff = torch.from_numpy(0.5 + M.splitmix64_uniform(37, mc.local_nodes_quad)).to(gpu)
    n_el, n3 = mc.n_elements, 512
    blocks = torch.empty(n_el * n3 * n3, dtype=torch.float64, device=gpu)
    plan.compute_weighted_mass_blocks(coeff, blocks)
    u = torch.from_numpy(mc.field(mp)).to(gpu)
    lap, a, b = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    plan.apply_aij(u, lap)
    plan.set_lhs_element_blocks(blocks)
    plan.apply_lhs(u, a)
    ref = lap + torch.bmm(blocks.view(n_el, n3, n3), u.view(n_el, n3, 1)).view(-1)
    assert float((a - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
    plan.set_lhs_coefficient(coeff)          # the matrix-free form of the same term (fused into the operator kernel)
    plan.apply_lhs(u, b)
    assert float((a - b).abs().max()) <= 1e-12 * float(ref.abs().max())
    # symmetric blocks (V^T W V), deterministic
    blk = blocks.view(n_el, n3, n3)
    assert float((blk - blk.transpose(1, 2)).abs().max()) <= 1e-13 * float(blk.abs().max())
    plan.set_lhs_element_blocks(blocks)
    plan.apply_lhs(u, b)
    assert torch.equal(a, b)
    plan.destroy()
