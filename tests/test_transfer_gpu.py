"""GPU parity tests of the hp-multigrid inter-grid transfer (d4est_hip_transfer_*) against the oracle's restatement of
d4est_operators_apply_p_prolong / _hp_prolong and their transposes, applied per coarse element in the traversal order of
d4est_solver_multigrid_callbacks.h."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _items(seed, n_items, pmax):
    if pmax == "p19":   # the top of the reference's degree range (d4est_operators.c:1205-1297): coarse p = 17, 18, fine up to p = 19
        hrefine = np.array([0, 1, 0], dtype=np.int32)
        degH = np.array([17, 17, 18], dtype=np.int32)
        degh = np.zeros(24, dtype=np.int32)
        degh[0] = 19
        degh[8:16] = [17, 18, 19, 19, 18, 17, 19, 18]
        degh[16] = 19
        return hrefine, degH, degh
    if pmax == "p15d3":   # the edge of the compile-time kernels: coarse p = 15 (16 nodes), fine sizes up to + 3; and one step beyond each
        hrefine = np.array([1, 0, 0, 1], dtype=np.int32)
        degH = np.array([15, 15, 16, 3], dtype=np.int32)
        degh = np.zeros(32, dtype=np.int32)
        degh[0:8] = [15, 16, 17, 18, 18, 17, 16, 15]
        degh[8] = 18
        degh[16] = 17
        degh[24:32] = [3, 4, 5, 6, 7, 3, 4, 7]     # + 4: beyond the fast kernels' range, the whole item takes the generic path
        return hrefine, degH, degh
    rng = np.random.RandomState(seed)
    hrefine = rng.randint(0, 2, size=n_items).astype(np.int32)
    degH = rng.randint(1, pmax, size=n_items).astype(np.int32)
    degh = np.zeros(8 * n_items, dtype=np.int32)
    for k in range(n_items):
        nc = 8 if hrefine[k] else 1
        degh[8 * k:8 * k + nc] = degH[k] + rng.randint(0, 3, size=nc)
    hrefine[0], degh[0] = 0, degH[0]   # a pure copy item
    return hrefine, degH, degh


def _oracle_transfer(oracle, hrefine, degH, degh, x, prolong):
    """item loop of the reference callbacks with the oracle's element functions"""
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    lib = oracle.lib
    lib.oracle_apply_p_prolong.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
    lib.oracle_apply_hp_prolong.argtypes = [dp, ctypes.c_int, ctypes.c_int, ip, dp]
    lib.oracle_apply_p_prolong_transpose.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
    lib.oracle_apply_hp_prolong_transpose.argtypes = [dp, ip, ctypes.c_int, ctypes.c_int, dp]
    nc_nodes = int(sum((int(d) + 1) ** 3 for d in degH))
    nf_nodes = int(sum((int(degh[8 * k + c]) + 1) ** 3 for k in range(len(hrefine)) for c in range(8 if hrefine[k] else 1)))
    out = np.zeros(nf_nodes if prolong else nc_nodes)
    co = fo = 0
    for k in range(len(hrefine)):
        dH = int(degH[k])
        nH = (dH + 1) ** 3
        dh = np.ascontiguousarray(degh[8 * k:8 * k + 8], dtype=np.int32)
        nc = 8 if hrefine[k] else 1
        nf = int(sum((int(d) + 1) ** 3 for d in dh[:nc]))
        if prolong:
            src = np.ascontiguousarray(x[co:co + nH]); dst = np.zeros(nf)
            if nc == 1:
                lib.oracle_apply_p_prolong(src.ctypes.data_as(dp), dH, 3, int(dh[0]), dst.ctypes.data_as(dp))
            else:
                lib.oracle_apply_hp_prolong(src.ctypes.data_as(dp), dH, 3, dh.ctypes.data_as(ip), dst.ctypes.data_as(dp))
            out[fo:fo + nf] = dst
        else:
            src = np.ascontiguousarray(x[fo:fo + nf]); dst = np.zeros(nH)
            if nc == 1:
                lib.oracle_apply_p_prolong_transpose(src.ctypes.data_as(dp), int(dh[0]), 3, dH, dst.ctypes.data_as(dp))
            else:
                lib.oracle_apply_hp_prolong_transpose(src.ctypes.data_as(dp), dh.ctypes.data_as(ip), 3, dH, dst.ctypes.data_as(dp))
            out[co:co + nH] = dst
        co += nH
        fo += nf
    return out


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("seed,n_items,pmax", [(1, 7, 4), (2, 40, 6), (3, 9, 9), (4, 3, 13), (7, 3, "p19"), (9, 4, "p15d3")])
def test_prolong_restrict_parity(gpu, hiplib, oracle, seed, n_items, pmax, generic, monkeypatch):
    """generic = True forces every item through the runtime-size kernels (D4EST_HIP_TRANSFER_GENERIC), False takes the compile-time
    kernels wherever they apply (coarse size <= 16 nodes, fine sizes within + 3)"""
    import torch
    from disco4est_amd import Transfer, mesh as M
    if generic:
        monkeypatch.setenv("D4EST_HIP_TRANSFER_GENERIC", "1")
    else:
        monkeypatch.delenv("D4EST_HIP_TRANSFER_GENERIC", raising=False)
    hrefine, degH, degh = _items(seed, n_items, pmax)
    t = Transfer(hrefine, degH, degh)
    xc = M.splitmix64_uniform(seed, t.coarse_nodes) - 0.5
    xf = M.splitmix64_uniform(seed + 100, t.fine_nodes) - 0.5
    dxc = torch.from_numpy(xc).to(gpu); dxf = torch.from_numpy(xf).to(gpu)
    out_f = torch.full((t.fine_nodes,), float("nan"), dtype=torch.float64, device=gpu)
    t.prolong(dxc, out_f)
    ref_f = _oracle_transfer(oracle, hrefine, degH, degh, xc, True)
    assert np.abs(out_f.cpu().numpy() - ref_f).max() <= RTOL * np.abs(ref_f).max()
    out_c = torch.full((t.coarse_nodes,), float("nan"), dtype=torch.float64, device=gpu)
    t.restrict(dxf, out_c)
    ref_c = _oracle_transfer(oracle, hrefine, degH, degh, xf, False)
    assert np.abs(out_c.cpu().numpy() - ref_c).max() <= RTOL * np.abs(ref_c).max()
    # adjointness  <P xc, xf> = <xc, P^T xf>  and exactness: prolongation reproduces a coarse polynomial (constants)
    a, b = torch.dot(out_f, dxf).item(), torch.dot(dxc, out_c).item()
    assert abs(a - b) <= 1e-12 * max(abs(a), abs(b))
    ones_f = torch.empty_like(out_f)
    t.prolong(torch.ones_like(dxc), ones_f)
    assert (ones_f - 1.0).abs().max().item() <= 1e-12
    t.destroy()


def test_empty_transfer(gpu, hiplib):
    import torch
    from disco4est_amd import Transfer
    t = Transfer(np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.int32))
    assert t.coarse_nodes == 0 and t.fine_nodes == 0
    z = torch.zeros(0, dtype=torch.float64, device=gpu)
    t.prolong(z, z.clone()); t.restrict(z, z.clone())
    torch.cuda.synchronize()
    t.destroy()


@pytest.mark.parametrize("seed,n_items,pmax", [(5, 12, 5), (6, 6, 9), (8, 3, "p19"), (10, 4, "p15d3")])
def test_projection_parity(gpu, hiplib, oracle, seed, n_items, pmax):
    """d4est_hip_transfer_project = d4est_operators_apply_p_restrict / _hp_restrict per item (the L2 projection of a field onto the
    coarse space): parity with the oracle, and project(prolong(x)) = x"""
    import torch
    from disco4est_amd import Transfer, mesh as M
    dp = ctypes.POINTER(ctypes.c_double)
    ip = ctypes.POINTER(ctypes.c_int)
    lib = oracle.lib
    lib.oracle_apply_p_restrict.argtypes = [dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp]
    lib.oracle_apply_hp_restrict.argtypes = [dp, ip, ctypes.c_int, ctypes.c_int, dp]
    hrefine, degH, degh = _items(seed, n_items, pmax)
    t = Transfer(hrefine, degH, degh)
    xf = M.splitmix64_uniform(seed + 200, t.fine_nodes) - 0.5
    ref = np.zeros(t.coarse_nodes)
    co = fo = 0
    for k in range(n_items):
        dH = int(degH[k]); nH = (dH + 1) ** 3
        dh = np.ascontiguousarray(degh[8 * k:8 * k + 8], dtype=np.int32)
        nc = 8 if hrefine[k] else 1
        nf = int(sum((int(d) + 1) ** 3 for d in dh[:nc]))
        src = np.ascontiguousarray(xf[fo:fo + nf]); dst = np.zeros(nH)
        if nc == 1:
            lib.oracle_apply_p_restrict(src.ctypes.data_as(dp), int(dh[0]), 3, dH, dst.ctypes.data_as(dp))
        else:
            lib.oracle_apply_hp_restrict(src.ctypes.data_as(dp), dh.ctypes.data_as(ip), 3, dH, dst.ctypes.data_as(dp))
        ref[co:co + nH] = dst
        co += nH; fo += nf
    out = torch.full((t.coarse_nodes,), float("nan"), dtype=torch.float64, device=gpu)
    t.project(torch.from_numpy(xf).to(gpu), out)
    assert np.abs(out.cpu().numpy() - ref).max() <= RTOL * np.abs(ref).max()
    xc = torch.from_numpy(M.splitmix64_uniform(seed + 300, t.coarse_nodes) - 0.5).to(gpu)
    fine = torch.empty(t.fine_nodes, dtype=torch.float64, device=gpu)
    back = torch.empty_like(xc)
    t.prolong(xc, fine)
    t.project(fine, back)
    assert float((back - xc).abs().max()) <= (1e-11 if pmax != "p19" else 1e-9)
