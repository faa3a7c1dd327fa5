"""Independent DENSE evaluation of d4est's weak Laplacian (volume stiffness + SIPG face terms) in numpy -- the face twin of the
dense volume assembly in tests/test_oracle.py.  TEST INFRASTRUCTURE.

Nothing here shares code with oracle/*.c or with the library: the 1-D tables are rebuilt from the reference's tabulated nodes and
weights (tests/golden/reference_nodes_weights.json) by barycentric Lagrange formulas, every tensor apply is an explicit Kronecker
matrix, the slicer / lift / re-orientation are explicit selection matrices, and the terms follow the reference's formulas:
  volume      out = sum_{lp,l} D_lp^T V^T [W J (dr_lp/dx . dr_l/dx) V D_l u]                 src/Quadrature/d4est_quadrature.c:263-382
  interface   term1 = -sj n.(grad u_m + grad u_p)/2, term2_l = -(1/2) sum_d dr_l/dx_d sj n_d [u], term3 = sj sigma [u];
              V^T W, project onto the side, lift; D_l^T on term2; x 1/2 on the gradient and on term 2 of a side whose mortar is
              half-size                                              src/dGMath/d4est_laplacian_flux_sipg.c:494-833, d4est_laplacian_flux.c:232-1014
  Dirichlet   u_p -> g (on the Lobatto face nodes, interpolated), grad u_p -> grad u_m, factor 2 on term 2       d4est_laplacian_flux_sipg.c:15-336
  Robin       V^T W sj (coeff u_m - rhs), lifted                                                                 :339-489
  penalties                                                                                                     :945-1005
Inputs are exactly the arrays the oracle and the engine take (mesh.* / forest.* build_sides and geometry)."""
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_TAB = json.load(open(os.path.join(_HERE, "golden", "reference_nodes_weights.json")))


def lobatto(deg):
    d = _TAB["lobatto"][str(deg + 1)]
    return np.array(d["x"]), np.array(d["w"])


def gauss(deg):
    d = _TAB["gauss"][str(deg + 1)]
    return np.array(d["x"]), np.array(d["w"])


def quad_rule(quad_type, deg):
    return gauss(deg) if quad_type == 0 else lobatto(deg)


def lagrange_matrix(x_from, x_to):
    """L[a, i] = l_i(x_to[a]) for the Lagrange basis on x_from (barycentric form)"""
    n = x_from.size
    w = np.array([1.0 / np.prod(x_from[i] - np.delete(x_from, i)) for i in range(n)])
    L = np.zeros((x_to.size, n))
    for a, x in enumerate(x_to):
        d = x - x_from
        hit = np.nonzero(np.abs(d) < 1e-15)[0]
        if hit.size:
            L[a, hit[0]] = 1.0
        else:
            t = w / d
            L[a] = t / t.sum()
    return L


def diff_matrix(x):
    """D[a, i] = l_i'(x[a])"""
    n = x.size
    w = np.array([1.0 / np.prod(x[i] - np.delete(x, i)) for i in range(n)])
    D = np.zeros((n, n))
    for a in range(n):
        for i in range(n):
            if a != i:
                D[a, i] = (w[i] / w[a]) / (x[a] - x[i])
        D[a, a] = -D[a].sum()
    return D


def kron3(Az, Ay, Ax):
    return np.kron(Az, np.kron(Ay, Ax))      # x fastest (dGMath/d4est_operators.c:1318-1323)


def face_select(f, n):
    """S[(a + n b), vol] = 1: trace on face f, tangential axes in increasing order, the first fastest (d4est_operators.c:1521-1582)"""
    d, fix = f // 2, (n - 1 if f % 2 else 0)
    S = np.zeros((n * n, n ** 3))
    for b in range(n):
        for a in range(n):
            idx = [0, 0, 0]
            ax = [t for t in range(3) if t != d]
            idx[d], idx[ax[0]], idx[ax[1]] = fix, a, b
            S[a + n * b, idx[0] + n * (idx[1] + n * idx[2])] = 1.0
    return S


def reorient_matrix(code, n):
    """out = transpose?(flip1?(flip0?(in))) on an n x n face array, first index fastest (d4est_operators.c:2044-2081)"""
    R = np.zeros((n * n, n * n))
    for b in range(n):
        for a in range(n):
            a1, b1 = (b, a) if code & 4 else (a, b)
            if code & 2:
                b1 = n - 1 - b1
            if code & 1:
                a1 = n - 1 - a1
            R[a + n * b, a1 + n * b1] = 1.0
    return R


def p_prolong_1d(degH, degh):
    return lagrange_matrix(lobatto(degH)[0], lobatto(degh)[0])


def hp_prolong_1d(degH, degh, child):
    """parent degree degH -> child half (0: [-1,0], 1: [0,1]) at degree degh (d4est_operators.c:944-993, d4est_reference.c:38-49)"""
    xh = lobatto(degh)[0]
    return lagrange_matrix(lobatto(degH)[0], 0.5 * xh + (0.5 if child else -0.5))


def quad_interp_1d(quad_type, deg, deg_quad):
    return lagrange_matrix(lobatto(deg)[0], quad_rule(quad_type, deg_quad)[0])


def penalty(fcn, deg_m, h_m, deg_p, h_p, prefactor):
    if fcn == 0:
        return prefactor * max(deg_m, deg_p) ** 2 / np.minimum(h_m, h_p)
    if fcn == 1:
        return prefactor * (0.5 * (deg_m + deg_p)) ** 2 / (0.5 * (h_m + h_p))
    if fcn == 2:
        return prefactor * (max(deg_m, deg_p) + 1) ** 2 / np.minimum(h_m, h_p)
    return prefactor * 0.5 * (deg_m ** 2 / h_m + deg_p ** 2 / h_p)


class DenseLaplacian:
    def __init__(self, mesh, J, rst, sides, reorient_face_order, penalty_prefactor=10.0, penalty_fcn=0):
        self.m, self.J, self.rst = mesh, np.asarray(J), np.asarray(rst).reshape(9, -1)
        self.s, self.pref, self.fcn = sides, penalty_prefactor, penalty_fcn
        self.rfo = reorient_face_order
        self.qt = mesh.quad_type

    # ---- volume
    def stiffness(self, u):
        m = self.m
        out = np.zeros(m.local_nodes)
        for e in range(m.n_elements):
            p, pq = int(m.deg[e]), int(m.deg_quad[e])
            n, nq = p + 1, pq + 1
            D = diff_matrix(lobatto(p)[0])
            V1 = quad_interp_1d(self.qt, p, pq)
            w = quad_rule(self.qt, pq)[1]
            I = np.eye(n)
            Dl = [kron3(I, I, D), kron3(I, D, I), kron3(D, I, I)]
            V = kron3(V1, V1, V1)
            W = np.kron(w, np.kron(w, w))
            s0, q0 = int(m.nodal_stride[e]), int(m.quad_stride[e])
            ue = u[s0:s0 + n ** 3]
            Jq = self.J[q0:q0 + nq ** 3]
            r = self.rst[:, q0:q0 + nq ** 3]          # r[3 i + j] = d r_i / d x_j
            acc = np.zeros(n ** 3)
            for lp in range(3):
                for l in range(3):
                    g = sum(r[3 * lp + k] * r[3 * l + k] for k in range(3))
                    acc += Dl[lp].T @ (V.T @ (W * Jq * g * (V @ (Dl[l] @ ue))))
            out[s0:s0 + n ** 3] = acc
        return out

    # ---- helpers on faces
    def _deg(self, ref):
        return int(self.m.deg[ref]) if ref >= 0 else int(self.s["ghost_deg"][-(ref + 2)])

    def _degq(self, ref):
        return int(self.m.deg_quad[ref]) if ref >= 0 else int(self.s["ghost_deg_quad"][-(ref + 2)])

    def _vals(self, ref, u, u_ghost):
        if ref >= 0:
            s0 = int(self.m.nodal_stride[ref])
            return u[s0:s0 + (self._deg(ref) + 1) ** 3]
        g = -(ref + 2)
        s0 = int(self.s["ghost_nodal_stride"][g])
        return u_ghost[s0:s0 + (self._deg(ref) + 1) ** 3]

    def _side_to_mortar(self, deg_side, deg_mq, child=None):
        """(face nodes of degree deg_side) -> mortar quadrature nodes of degree deg_mq: p- or hp-prolongation to the Lobatto nodes of
        degree deg_mq, then interpolation to the quadrature nodes (d4est_laplacian_flux.c:635-694)"""
        if child is None:
            P = p_prolong_1d(deg_side, deg_mq)
            Pa = Pb = P
        else:
            Pa, Pb = hp_prolong_1d(deg_side, deg_mq, child & 1), hp_prolong_1d(deg_side, deg_mq, child >> 1)
        I = quad_interp_1d(self.qt, deg_mq, deg_mq)
        return np.kron(I @ Pb, I @ Pa)

    def _mortar_to_side(self, deg_side, deg_ml, deg_mq, child=None):
        """V^T W at the mortar (galerkin integral with deg_mortar_lobatto test functions), then the transposed prolongation onto the
        side (d4est_laplacian_flux_sipg.c:641-768, Mesh/d4est_mortars.c:510-547)"""
        I = quad_interp_1d(self.qt, deg_ml, deg_mq)
        w = quad_rule(self.qt, deg_mq)[1]
        if child is None:
            P = p_prolong_1d(deg_side, deg_ml)
            Pa = Pb = P
        else:
            Pa, Pb = hp_prolong_1d(deg_side, deg_ml, child & 1), hp_prolong_1d(deg_side, deg_ml, child >> 1)
        return np.kron(Pb.T @ I.T, Pa.T @ I.T) @ np.diag(np.kron(w, w))

    def _grad_ops(self, deg):
        n = deg + 1
        D = diff_matrix(lobatto(deg)[0])
        I = np.eye(n)
        return [kron3(I, I, D), kron3(I, D, I), kron3(D, I, I)]

    def _geom(self, S, T, Ttot, off):
        s = self.s
        sj = s["sj"][S + off:S + off + T]
        hm = s["hm"][S + off:S + off + T]
        hp = s["hp"][S + off:S + off + T]
        nrm = [s["n"][3 * S + d * Ttot + off:3 * S + d * Ttot + off + T] for d in range(3)]
        rm = [[s["drst_m"][9 * S + (i + 3 * j) * Ttot + off:9 * S + (i + 3 * j) * Ttot + off + T] for j in range(3)] for i in range(3)]
        return sj, hm, hp, nrm, rm

    def _rp(self, S, T, Ttot, off_p):
        s = self.s
        return [[s["drst_p"][9 * S + (i + 3 * j) * Ttot + off_p:9 * S + (i + 3 * j) * Ttot + off_p + T] for j in range(3)] for i in range(3)]

    # ---- one (-) element's contribution from one mortar (sub-)face
    def _mortar(self, e, f, ep, f_p, code, child_m, child_p, S, Ttot, off, off_p, u, u_ghost, half_m, half_p, out):
        deg_m, deg_p = self._deg(e), self._deg(ep)
        deg_mq = max(self._degq(e), self._degq(ep))
        deg_ml = max(deg_m, deg_p)
        nq = deg_mq + 1
        T = nq * nq
        sj, hm, hp, nrm, rm = self._geom(S, T, Ttot, off)
        rp = self._rp(S, T, Ttot, off_p)
        um, up = self._vals(e, u, u_ghost), self._vals(ep, u, u_ghost)
        Sm, Sp = face_select(f, deg_m + 1), face_select(f_p, deg_p + 1)
        Cm = self._side_to_mortar(deg_m, deg_mq, child_m)
        Gm, Gp = self._grad_ops(deg_m), self._grad_ops(deg_p)
        # (-) side
        u_m = Cm @ (Sm @ um)
        dudr_m = [Cm @ (Sm @ (Gm[i] @ um)) for i in range(3)]
        dudx_m = [sum(rm[i][j] * dudr_m[i] for i in range(3)) for j in range(3)]
        # (+) side: u is re-oriented on its Lobatto face nodes, then projected like the (-) side (d4est_laplacian_flux.c:575-657);
        # the gradient is projected and mapped to x in the (+) side's OWN order, then re-oriented at the mortar nodes (:733-900)
        Rl = reorient_matrix(code, deg_p + 1)
        Cp_m = self._side_to_mortar(deg_p, deg_mq, child_p[0])        # child index in (-) order, applied to the re-oriented face
        u_p = Cp_m @ (Rl @ (Sp @ up))
        Cp_own = self._side_to_mortar(deg_p, deg_mq, child_p[1])      # the (+) side's own child index
        dudr_p = [Cp_own @ (Sp @ (Gp[i] @ up)) for i in range(3)]
        Rq = reorient_matrix(code, nq)
        dudx_p = [Rq @ sum(rp[i][j] * dudr_p[i] for i in range(3)) for j in range(3)]
        if half_m:
            dudx_m = [0.5 * v for v in dudx_m]
        if half_p:
            dudx_p = [0.5 * v for v in dudx_p]
        sigma = penalty(self.fcn, deg_m, hm, deg_p, hp, self.pref)
        jump = u_m - u_p
        t1 = -sum(nrm[d] * sj * 0.5 * (dudx_p[d] + dudx_m[d]) for d in range(3))
        t2 = [-0.5 * sum(rm[l][d] * sj * nrm[d] for d in range(3)) * jump for l in range(3)]
        t3 = sj * sigma * jump
        E = self._mortar_to_side(deg_m, deg_ml, deg_mq, child_m)
        acc = Sm.T @ (E @ (t1 + t3))
        for l in range(3):
            acc = acc + (0.5 if half_m else 1.0) * (Gm[l].T @ (Sm.T @ (E @ t2[l])))
        s0 = int(self.m.nodal_stride[e])
        out[s0:s0 + acc.size] += acc

    def _boundary(self, e, f, u, g, robin, out):
        s = self.s
        sd = 6 * e + f
        deg, degq = self._deg(e), self._degq(e)
        nq = degq + 1
        T = nq * nq
        S = int(s["side_mortar_stride"][sd])
        sj, hm, _, nrm, rm = self._geom(S, T, T, 0)
        um = self._vals(e, u, None)
        Sm = face_select(f, deg + 1)
        C = np.kron(quad_interp_1d(self.qt, deg, degq), quad_interp_1d(self.qt, deg, degq))      # boundary: Lobatto(deg) -> quadrature(deg_quad)
        I = quad_interp_1d(self.qt, deg, degq)
        w = quad_rule(self.qt, degq)[1]
        E = np.kron(I.T, I.T) @ np.diag(np.kron(w, w))
        G = self._grad_ops(deg)
        u_m = C @ (Sm @ um)
        s0 = int(self.m.nodal_stride[e])
        if robin is not None:
            coeff, rhs = robin
            out[s0:s0 + um.size] += Sm.T @ (E @ (sj * (coeff[S:S + T] * u_m - rhs[S:S + T])))
            return
        B0 = int(s["side_bndry_stride"][sd])
        gq = C @ (g[B0:B0 + (deg + 1) ** 2] if g is not None else np.zeros((deg + 1) ** 2))
        dudr = [C @ (Sm @ (G[i] @ um)) for i in range(3)]
        dudx = [sum(rm[i][j] * dudr[i] for i in range(3)) for j in range(3)]
        sigma = penalty(self.fcn, deg, hm, deg, hm, self.pref)
        jump = u_m - gq
        t1 = -sum(nrm[d] * sj * dudx[d] for d in range(3))
        t2 = [-0.5 * sum(rm[l][d] * nrm[d] * sj for d in range(3)) * 2.0 * jump for l in range(3)]
        t3 = sj * sigma * jump
        acc = Sm.T @ (E @ (t1 + t3))
        for l in range(3):
            acc = acc + G[l].T @ (Sm.T @ (E @ t2[l]))
        out[s0:s0 + acc.size] += acc

    def apply(self, u, u_ghost=None, g=None, robin=None):
        m, s = self.m, self.s
        out = self.stiffness(u)
        hang = s.get("side_hang")
        nodes2 = lambda a, b: (max(self._degq(a), self._degq(b)) + 1) ** 2
        for e in range(m.n_elements):
            for f in range(6):
                sd = 6 * e + f
                nbr, f_p, code = int(s["side_nbr"][sd]), int(s["side_nbr_face"][sd]), int(s["side_reorder"][sd])
                S = int(s["side_mortar_stride"][sd])
                h = 0 if hang is None else int(hang[sd])
                if nbr == -1:
                    self._boundary(e, f, u, g, robin, out)
                elif h == 0:
                    T = nodes2(e, nbr)
                    self._mortar(e, f, nbr, f_p, code, None, (None, None), S, T, 0, 0, u, u_ghost, False, False, out)
                elif h == 1:      # big side: 4 sub-mortars in (-) order; the (+) elements in (-) order are side_nbr4
                    o = int(s["side_orientation"][sd])
                    n4 = [int(v) for v in s["side_nbr4"][4 * sd:4 * sd + 4]]
                    Tm = [nodes2(e, n4[i]) for i in range(4)]
                    Tp = [0] * 4
                    for i in range(4):
                        Tp[self.rfo(f, f_p, o, i)] = Tm[i]
                    for i in range(4):
                        j = self.rfo(f, f_p, o, i)
                        self._mortar(e, f, n4[i], f_p, code, i, (None, None), S, sum(Tm), sum(Tm[:i]), sum(Tp[:j]), u, u_ghost, True, False, out)
                else:             # small side: sub-mortar c of its group; the (+) side is the big element
                    o = int(s["side_orientation"][sd])
                    c = int(s["side_sub"][sd])
                    grp = [int(v) for v in s["side_nbr4"][4 * sd:4 * sd + 4]]
                    Tm = [nodes2(grp[i], nbr) for i in range(4)]
                    Tp = [0] * 4
                    for i in range(4):
                        Tp[self.rfo(f, f_p, o, i)] = Tm[i]
                    j = self.rfo(f, f_p, o, c)
                    self._mortar(e, f, nbr, f_p, code, None, (c, j), S, sum(Tm), sum(Tm[:c]), sum(Tp[:j]), u, u_ghost, False, True, out)
        return out
