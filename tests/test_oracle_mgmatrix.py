"""CPU pins of the oracle's restatement of the multigrid MATRIX OPERATOR (oracle/d4est_oracle_mgmatrix.c) against an independent
numpy construction: dense Kronecker products of the 1-D tables, numpy matmul for P^T M P, and the reference's literal window
(d4est_operators.c:637, :651) restated as one reshape of the transposed stacked prolongation."""
import numpy as np
import pytest

from disco4est_amd import mesh as M


def _kron3(Az, Ay, Ax):
    return np.kron(Az, np.kron(Ay, Ax))


def _dense_prolong(oracle, degH, degh, children):
    """stacked (sum nh^3) x nH^3 prolongation from the 1-D tables: p_prolong for one child, the two hp halves for eight"""
    if children == 1:
        P1 = oracle.p_prolong(degH, degh[0])
        return _kron3(P1, P1, P1)
    rows = []
    for c in range(8):
        P2 = oracle.hp_prolong(degH, degh[c])
        rows.append(_kron3(P2[(c >> 2) & 1], P2[(c >> 1) & 1], P2[c & 1]))
    return np.vstack(rows)


@pytest.mark.parametrize("deg,inc", [(1, 0), (2, 1), (3, 0)])
def test_compute_matrix_is_dense_weighted_mass(oracle, deg, inc):
    m = M.BrickMesh(0, deg, deg_quad_inc=inc)
    mp = M.SineMap(0.05)
    J, _ = m.geometry(mp)
    coeff = 0.5 + M.splitmix64_uniform(3, m.local_nodes_quad)
    blocks = oracle.mg_matrix_setup(m, J, coeff)
    n3 = (deg + 1) ** 3
    assert blocks.size == n3 * n3
    B = oracle.lobatto_to_gauss(deg, deg + inc)
    _, w = oracle.gauss(deg + inc)
    V = _kron3(B, B, B)
    W = np.kron(w, np.kron(w, w)) * J * coeff
    ref = V.T @ (W[:, None] * V)
    got = blocks.reshape(n3, n3)
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    # column i of the block is the matrix-free apply of the unit vector (d4est_quadrature.c:1164-1182), and the block applied to a
    # vector is the matrix-free weighted mass apply (QUAD_APPLY_MATRIX and QUAD_COMPUTE_MATRIX are one operator)
    u = M.splitmix64_uniform(5, n3) - 0.5
    assert np.abs(got @ u - oracle.apply_weighted_mass(m, J, coeff, u)).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("degH,degh", [(2, [3]), (1, [1]), (1, [1, 2, 1, 2, 2, 1, 3, 1]), (2, [2, 3, 2, 2, 3, 3, 2, 4])])
def test_PT_mat_P_exact_and_literal(oracle, degH, degh):
    children = len(degh)
    hrefine = np.array([1 if children == 8 else 0], dtype=np.int32)
    dh = np.zeros(8, dtype=np.int32)
    dh[:children] = degh
    n3 = [(d + 1) ** 3 for d in degh]
    rng = np.random.RandomState(7)
    mats = [rng.rand(n, n) - 0.5 for n in n3]
    fine = np.concatenate([a.ravel() for a in mats])
    P = _dense_prolong(oracle, degH, degh, children)
    nH3 = (degH + 1) ** 3
    # exact: sum_c P_c^T M_c P_c
    ref = np.zeros((nH3, nH3))
    r0 = 0
    for c in range(children):
        Pc = P[r0:r0 + n3[c]]
        ref += Pc.T @ mats[c] @ Pc
        r0 += n3[c]
    got = oracle.mg_matrix_restriction(hrefine, [degH], dh, fine, literal_window=False).reshape(nH3, nH3)
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    # literal (:637, :651): PT = transpose of the STACKED P, child c's left factor = PT.ravel()[stride_P : stride_P + nH3 nh3] read as
    # an nH3 x nh3 matrix
    PT_flat = np.ascontiguousarray(P.T).ravel()
    lit = np.zeros((nH3, nH3))
    r0 = 0
    stride_P = 0
    for c in range(children):
        Wc = PT_flat[stride_P:stride_P + nH3 * n3[c]].reshape(nH3, n3[c])
        lit += Wc @ (mats[c] @ P[r0:r0 + n3[c]])
        r0 += n3[c]
        stride_P += n3[c] * nH3
    got_lit = oracle.mg_matrix_restriction(hrefine, [degH], dh, fine, literal_window=True).reshape(nH3, nH3)
    assert np.abs(got_lit - lit).max() <= 1e-13 * max(np.abs(lit).max(), 1e-300)
    if children == 1:
        assert np.array_equal(got, got_lit)          # one child: the window is P_0^T
    else:
        assert np.abs(got - got_lit).max() > 1e-3 * np.abs(ref).max()   # eight children: it is not (see DESIGN.md "MG matrix operator")


def test_blocks_term_is_the_galerkin_chain(oracle):
    """coarse blocks applied per element (constant_density_star_fcns.h:485-527) = P^T (V^T W J c V) P u through the element transfer
    functions and the matrix-free fine-level term: a two-level hp hierarchy, curved, children of different degrees"""
    from tests.test_transfer_gpu import _oracle_transfer
    mp = M.SineMap(0.05)
    deg_f = np.array([2, 3, 2, 2, 3, 3, 2, 4], dtype=np.int32)
    mf = M.BrickMesh(1, deg_f, deg_quad_inc=1)
    J, _ = mf.geometry(mp)
    coeff = 0.3 + M.splitmix64_uniform(11, mf.local_nodes_quad)
    fine = oracle.mg_matrix_setup(mf, J, coeff)
    hrefine, degH, degh = np.array([1], np.int32), np.array([2], np.int32), deg_f.copy()
    coarse = oracle.mg_matrix_restriction(hrefine, degH, degh, fine)
    u = M.splitmix64_uniform(13, 27) - 0.5
    Au = np.zeros(27)
    import ctypes
    from tests.oracle_lib import P as DP, I as IP, dp, ip
    f = oracle.lib.oracle_apply_element_blocks_add
    f.argtypes = [ctypes.c_int, ip, ip, ctypes.c_int, dp, dp, dp]
    f(1, IP(degH), IP(np.zeros(1, np.int32)), 27, DP(coarse), DP(u), DP(Au))
    uf = _oracle_transfer(oracle, hrefine, degH, degh, u, True)
    Mu = oracle.apply_weighted_mass(mf, J, coeff, uf)
    ref = _oracle_transfer(oracle, hrefine, degH, degh, Mu, False)
    assert np.abs(Au - ref).max() <= 1e-13 * np.abs(ref).max()
