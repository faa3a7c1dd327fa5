"""ctypes wrapper of the CPU oracle (oracle/libd4est_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "libd4est_oracle.so")

dp = ctypes.POINTER(ctypes.c_double)
ip = ctypes.POINTER(ctypes.c_int)


LIB_NATIVE = os.path.join(ODIR, "libd4est_oracle_native.so")


def build(native=False):
    """native=True: -march=native build made ON the machine that runs it (bench.py's cpu_baseline leg)."""
    srcs = [os.path.join(ODIR, f) for f in os.listdir(ODIR) if f.endswith((".c", ".h"))]
    newest = max(os.path.getmtime(s) for s in srcs)
    if native:
        if (not os.path.exists(LIB_NATIVE)) or os.path.getmtime(LIB_NATIVE) < newest:
            subprocess.check_call(["make", "-C", ODIR, "-s", "-B", "native"])
        return LIB_NATIVE
    if (not os.path.exists(LIB)) or os.path.getmtime(LIB) < newest:
        subprocess.check_call(["make", "-C", ODIR, "-s", "-B"])
    return LIB


def P(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(dp)


def I(a):
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ip)


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        lib.oracle_lgl_jacobi.restype = ctypes.c_double
        lib.oracle_lgl_jacobi.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        lib.oracle_linalg_matvec_plus_vec.argtypes = [ctypes.c_double, dp, dp, ctypes.c_double, dp, ctypes.c_int, ctypes.c_int]

    # ---- tables
    def lobatto(self, deg):
        x = np.zeros(deg + 1); w = np.zeros(deg + 1)
        self.lib.oracle_lobatto_nodes_and_weights(deg + 1, P(x), P(w))
        return x, w

    def gauss(self, deg):
        x = np.zeros(deg + 1); w = np.zeros(deg + 1)
        self.lib.oracle_gauss_nodes_and_weights(deg + 1, P(x), P(w))
        return x, w

    def _sq(self, fn, deg):
        n = deg + 1
        out = np.zeros((n, n))
        getattr(self.lib, fn)(P(out), deg)
        return out

    def Vij(self, deg): return self._sq("oracle_build_Vij_1d", deg)
    def mij(self, deg): return self._sq("oracle_build_mij_1d", deg)
    def invmij(self, deg): return self._sq("oracle_build_invmij_1d", deg)
    def dij(self, deg): return self._sq("oracle_build_dij_1d", deg)

    def lobatto_to_gauss(self, deg, deg_gauss):
        out = np.zeros((deg_gauss + 1, deg + 1))
        self.lib.oracle_build_lobatto_to_gauss_interp_1d(P(out), deg, deg_gauss)
        return out

    def p_prolong(self, degH, degh):
        out = np.zeros((degh + 1, degH + 1))
        self.lib.oracle_build_p_prolong_1d(P(out), degH, degh)
        return out

    def hp_prolong(self, degH, degh):
        out = np.zeros((2, degh + 1, degH + 1))
        self.lib.oracle_build_hp_prolong_1d(P(out), degH, degh)
        return out

    def p_restrict(self, degH, degh):
        out = np.zeros((degH + 1, degh + 1))
        self.lib.oracle_build_p_restrict_1d(P(out), degH, degh)
        return out

    def hp_restrict(self, degH, degh):
        out = np.zeros((2, degH + 1, degh + 1))
        self.lib.oracle_build_hp_restrict_1d(P(out), degH, degh)
        return out

    # ---- kron
    def kron_A1A2A3x(self, A1, A2, A3, x):
        out = np.zeros(A1.shape[0] * A2.shape[0] * A3.shape[0])
        self.lib.oracle_kron_A1A2A3x_nonsqr(P(out), P(A1), P(A2), P(A3), P(x), A1.shape[0], A1.shape[1], A2.shape[0],
                                            A2.shape[1], A3.shape[0], A3.shape[1])
        return out

    def kron_A1A2x(self, A1, A2, x):
        out = np.zeros(A1.shape[0] * A2.shape[0])
        self.lib.oracle_kron_A1A2x_nonsqr(P(out), P(A1), P(A2), P(x), A1.shape[0], A1.shape[1], A2.shape[0], A2.shape[1])
        return out

    def kron_AoBoC(self, A, B, C):
        out = np.zeros((A.shape[0] * B.shape[0] * C.shape[0], A.shape[1] * B.shape[1] * C.shape[1]))
        self.lib.oracle_kron_AoBoC(P(A), P(B), P(C), P(out), A.shape[0], A.shape[1], B.shape[0], B.shape[1], C.shape[0], C.shape[1])
        return out

    # ---- element-level
    def apply_dij(self, u, deg, d, transpose=False):
        out = np.zeros_like(u)
        (self.lib.oracle_apply_dij_transpose if transpose else self.lib.oracle_apply_dij)(P(u), deg, d, P(out))
        return out

    def apply_mij(self, u, deg):
        out = np.zeros_like(u); self.lib.oracle_apply_mij(P(u), deg, P(out)); return out

    def apply_invmij(self, u, deg):
        out = np.zeros_like(u); self.lib.oracle_apply_invmij(P(u), deg, P(out)); return out

    def apply_slicer(self, u, face, deg):
        out = np.zeros((deg + 1) ** 2); self.lib.oracle_apply_slicer(P(u), face, deg, P(out)); return out

    def apply_lift(self, f, deg, face):
        out = np.zeros((deg + 1) ** 3); self.lib.oracle_apply_lift(P(f), deg, face, P(out)); return out

    def stiffness_element(self, quad_type, u, deg, J, rst9, deg_quad):
        """rst9: list of 9 arrays (3*i+j) of quad-node values"""
        arr = (dp * 3 * 3)()
        keep = [np.ascontiguousarray(r, dtype=np.float64) for r in rst9]
        for i in range(3):
            for j in range(3):
                arr[i][j] = P(keep[3 * i + j])
        out = np.zeros_like(u)
        self.lib.oracle_quadrature_apply_stiffness_matrix(quad_type, P(u), deg, P(J), arr, deg_quad, P(out))
        return out

    # ---- mesh-level (flat element list)
    def apply_stiffness(self, mesh, J, rst, u, nthreads=1):
        Au = np.zeros(mesh.local_nodes)
        self.lib.oracle_laplacian_apply_stiffness_matrix(
            mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride), I(mesh.quad_stride),
            mesh.local_nodes_quad, P(J), P(rst), P(u), P(Au), nthreads)
        return Au

    def apply_mass(self, mesh, J, u, nthreads=1):
        Mu = np.zeros(mesh.local_nodes)
        self.lib.oracle_laplacian_apply_mass_matrix(
            mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride), I(mesh.quad_stride),
            P(J), P(u), P(Mu), nthreads)
        return Mu

    def apply_weighted_mass(self, mesh, J, coeff_q, u):
        out = np.zeros(mesh.local_nodes)
        self.lib.oracle_elements_apply_weighted_mass_matrix(
            mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride), I(mesh.quad_stride),
            P(J), P(coeff_q), P(u), P(out))
        return out

    def apply_inverse_mass(self, mesh, J, x):
        out = np.zeros(mesh.local_nodes)
        self.lib.oracle_elements_apply_inverse_mass_matrix(
            mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride), I(mesh.quad_stride), P(J), P(x), P(out))
        return out

    def apply_mij(self, mesh, x, inverse=False):
        out = np.zeros(mesh.local_nodes)
        self.lib.oracle_elements_apply_mij(mesh.n_elements, I(mesh.deg), I(mesh.nodal_stride), P(x), P(out), int(inverse))
        return out

    def apply_galerkin(self, mesh, J, fq):
        out = np.zeros(mesh.local_nodes)
        for e in range(mesh.n_elements):
            s, q = mesh.nodal_stride[e], mesh.quad_stride[e]
            n3, q3 = (mesh.deg[e] + 1) ** 3, (mesh.deg_quad[e] + 1) ** 3
            o = np.zeros(n3)
            self.lib.oracle_quadrature_apply_galerkin_integral(mesh.quad_type, P(np.ascontiguousarray(fq[q:q + q3])), int(mesh.deg[e]),
                                                               P(np.ascontiguousarray(J[q:q + q3])), int(mesh.deg_quad[e]), P(o))
            out[s:s + n3] = o
        return out

    def interpolate(self, mesh, u):
        out = np.zeros(mesh.local_nodes_quad)
        for e in range(mesh.n_elements):
            s, q = mesh.nodal_stride[e], mesh.quad_stride[e]
            n3, q3 = (mesh.deg[e] + 1) ** 3, (mesh.deg_quad[e] + 1) ** 3
            o = np.zeros(q3)
            self.lib.oracle_quadrature_interpolate(mesh.quad_type, P(np.ascontiguousarray(u[s:s + n3])), int(mesh.deg[e]), P(o), int(mesh.deg_quad[e]))
            out[q:q + q3] = o
        return out

    def apply_aij(self, mesh, J, rst, sides, u, u_ghost=None, bndry_lobatto=None, penalty_prefactor=10.0, penalty_fcn=0, nthreads=1,
                  robin=None):
        """d4est_laplacian_apply_aij on the flat side list of mesh.build_sides(); robin = (coeff_quad, rhs_quad) switches
        the boundary sides to BC_ROBIN"""
        self.lib.oracle_flux_set_robin.argtypes = [dp, dp]
        if robin is not None:
            rc, rr = (np.ascontiguousarray(a, dtype=np.float64) for a in robin)
            self.lib.oracle_flux_set_robin(P(rc), P(rr))
        self.lib.oracle_flux_set_hanging.argtypes = [ip, ip, ip, ip]
        if "side_hang" in sides:
            self._hang_keep = [np.ascontiguousarray(sides[k], dtype=np.int32) for k in ("side_hang", "side_sub", "side_nbr4", "side_orientation")]
            self.lib.oracle_flux_set_hanging(*[I(a) for a in self._hang_keep])
        try:
            return self._apply_aij(mesh, J, rst, sides, u, u_ghost, bndry_lobatto, penalty_prefactor, penalty_fcn, nthreads)
        finally:
            self.lib.oracle_flux_set_robin(None, None)
            self.lib.oracle_flux_set_hanging(None, None, None, None)

    def _apply_aij(self, mesh, J, rst, sides, u, u_ghost, bndry_lobatto, penalty_prefactor, penalty_fcn, nthreads):
        Au = np.zeros(mesh.local_nodes)
        ug = np.zeros(max(sides["ghost_nodes"], 1)) if u_ghost is None else np.ascontiguousarray(u_ghost)
        bl = None if bndry_lobatto is None else np.ascontiguousarray(bndry_lobatto, dtype=np.float64)
        f = self.lib.oracle_laplacian_apply_aij
        f.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, ip, ip, ctypes.c_int, ctypes.c_int, dp, dp,
                      ctypes.c_int, ip, ip, ip, ctypes.c_int, ip, ip, ip, ip, ip,
                      dp, dp, dp, dp, dp, dp, ctypes.c_double, ctypes.c_int, dp, dp, dp, dp, ctypes.c_int]
        gd, gq, gs = (np.ascontiguousarray(sides[k], dtype=np.int32) for k in ("ghost_deg", "ghost_deg_quad", "ghost_nodal_stride"))
        f(mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride), I(mesh.quad_stride),
          mesh.local_nodes, mesh.local_nodes_quad, P(J), P(rst), len(gd), I(gd), I(gq), I(gs), int(sides["ghost_nodes"]),
          I(sides["side_nbr"]), I(sides["side_nbr_face"]), I(sides["side_reorder"]), I(sides["side_mortar_stride"]),
          I(sides["side_bndry_stride"]), P(sides["sj"]), P(sides["n"]), P(sides["drst_m"]), P(sides["drst_p"]),
          P(sides["hm"]), P(sides["hp"]), float(penalty_prefactor), int(penalty_fcn), P(u), P(ug),
          (P(bl) if bl is not None else None), P(Au), nthreads)
        return Au

    def set_operator(self, mesh, J, rst, sides, penalty_prefactor=10.0, penalty_fcn=0, threads=8):
        f = self.lib.oracle_set_aij_operator
        f.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, ip, ip, ctypes.c_int, ctypes.c_int, dp, dp, ip, ip, ip, ip, ip,
                      dp, dp, dp, dp, dp, dp, ctypes.c_double, ctypes.c_int, ctypes.c_int]
        self._op_keep = (mesh, J, rst, sides)
        f(mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride), I(mesh.quad_stride),
          mesh.local_nodes, mesh.local_nodes_quad, P(J), P(rst), I(sides["side_nbr"]), I(sides["side_nbr_face"]),
          I(sides["side_reorder"]), I(sides["side_mortar_stride"]), I(sides["side_bndry_stride"]), P(sides["sj"]), P(sides["n"]),
          P(sides["drst_m"]), P(sides["drst_p"]), P(sides["hm"]), P(sides["hp"]), float(penalty_prefactor), int(penalty_fcn), threads)

    def set_lhs_coefficient(self, coeff_quad):
        """zeroth-order term of the registered operator (None switches it off); the array is kept alive here"""
        self.lib.oracle_set_lhs_coefficient.argtypes = [dp]
        self._lhs_coeff = None if coeff_quad is None else np.ascontiguousarray(coeff_quad, dtype=np.float64)
        self.lib.oracle_set_lhs_coefficient(None if self._lhs_coeff is None else P(self._lhs_coeff))

    # ---- multigrid matrix operator (oracle/d4est_oracle_mgmatrix.c)
    def mg_matrix_setup(self, mesh, J, coeff_q):
        """d4est_solver_multigrid_matrix_setup_fofufofvlilj_operator: one dense block per element (QUAD_COMPUTE_MATRIX), consecutive"""
        f = self.lib.oracle_mg_matrix_setup_fofufofvlilj_operator
        f.restype = ctypes.c_longlong
        f.argtypes = [ctypes.c_int, ctypes.c_int, ip, ip, ip, dp, dp, dp]
        cq = None if coeff_q is None else np.ascontiguousarray(coeff_q, dtype=np.float64)
        n = f(mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.quad_stride), P(J), None if cq is None else P(cq), None)
        out = np.zeros(n)
        f(mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.quad_stride), P(J), None if cq is None else P(cq), P(out))
        return out

    def mg_matrix_restriction(self, hrefine, degH, degh, fine_matrix, literal_window=False):
        """the restriction callback of the matrix operator over a transfer's item list: coarse blocks = sum_c P_c^T M_c P_c
        (literal_window: the reference's :651 window instead of P_c^T)"""
        hrefine, degH, degh = (np.ascontiguousarray(a, dtype=np.int32) for a in (hrefine, degH, degh))
        n = int(sum((int(d) + 1) ** 6 for d in degH))
        out = np.zeros(n)
        f = self.lib.oracle_mg_matrix_restriction
        f.argtypes = [ctypes.c_int, ip, ip, ip, ctypes.c_int, dp, dp]
        f(len(hrefine), I(hrefine), I(degH), I(degh), int(bool(literal_window)), P(np.ascontiguousarray(fine_matrix)), P(out))
        return out

    def set_lhs_element_blocks(self, matrix):
        """zeroth-order term of the registered operator as dense element blocks (None switches it off); kept alive here"""
        self.lib.oracle_set_lhs_element_blocks.argtypes = [dp]
        self._lhs_blocks = None if matrix is None else np.ascontiguousarray(matrix, dtype=np.float64)
        self.lib.oracle_set_lhs_element_blocks(None if self._lhs_blocks is None else P(self._lhs_blocks))

    def apply_lhs(self, u):
        out = np.zeros_like(u)
        self.lib.oracle_apply_lhs.argtypes = [dp, dp]
        self.lib.oracle_apply_lhs(P(np.ascontiguousarray(u)), P(out))
        return out

    def set_hanging(self, sides):
        """keep the hanging-face arrays of `sides` active for the following calls (None switches back to conforming)"""
        self.lib.oracle_flux_set_hanging.argtypes = [ip, ip, ip, ip]
        if sides is None or "side_hang" not in sides:
            self.lib.oracle_flux_set_hanging(None, None, None, None)
            return
        self._hang_keep2 = [np.ascontiguousarray(sides[k], dtype=np.int32) for k in ("side_hang", "side_sub", "side_nbr4", "side_orientation")]
        self.lib.oracle_flux_set_hanging(*[I(a) for a in self._hang_keep2])

    def cheby_iterate(self, u, rhs, iters, lmin, lmax, residual_at_end=1):
        """returns (u_new, r); operator from set_operator()"""
        u = u.copy(); Au = np.zeros_like(u); r = np.zeros_like(u)
        f = self.lib.oracle_cheby_iterate_aux
        f.argtypes = [dp, dp, dp, dp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]
        f(P(u), P(np.ascontiguousarray(rhs)), P(Au), P(r), iters, lmin, lmax, residual_at_end)
        return u, r

    def cg_eigs(self, u, rhs, imax, use_new=1):
        """returns (spectral_bound, u_after)"""
        u = u.copy(); Au = np.zeros_like(u)
        bound = ctypes.c_double(0.0)
        f = self.lib.oracle_cg_eigs
        f.argtypes = [dp, dp, dp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_double)]
        f(P(u), P(np.ascontiguousarray(rhs)), P(Au), imax, use_new, ctypes.byref(bound))
        return bound.value, u

    def geometry_numerical(self, mesh, xyz):
        """(J_quad, rst_xyz_quad) with DX_compute_method = GEOM_COMPUTE_NUMERICAL from xyz = (x, y, z) at the Lobatto nodes"""
        X = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in xyz]))
        J = np.zeros(mesh.local_nodes_quad); rst = np.zeros(9 * mesh.local_nodes_quad)
        self.lib.oracle_mesh_compute_geometry_numerical(mesh.quad_type, mesh.n_elements, I(mesh.deg), I(mesh.deg_quad), I(mesh.nodal_stride),
                                                        I(mesh.quad_stride), mesh.local_nodes, mesh.local_nodes_quad, P(X), P(J), P(rst))
        return J, rst

    # ---- additive Schwarz (oracle/d4est_oracle_schwarz.c); operator from set_operator()
    def schwarz_restrictor_1d(self, deg, rs):
        R = np.zeros((2, rs, deg + 1))
        self.lib.oracle_schwarz_build_restrictor_1d(P(R), deg, rs)
        return R

    def schwarz_weights_1d(self, deg, rs):
        w = np.zeros(2 * rs + deg + 1)
        self.lib.oracle_schwarz_build_weights_1d(P(w), deg, rs)
        return w

    def schwarz_apply_restrictor(self, x, faces, deg, rs, transpose=False):
        faces = np.ascontiguousarray(faces, dtype=np.int32)
        n_res = self.lib.oracle_schwarz_restricted_nodes(I(faces), deg, rs)
        out = np.zeros((deg + 1) ** 3 if transpose else n_res)
        self.lib.oracle_schwarz_apply_restrictor(P(np.ascontiguousarray(x, dtype=np.float64)), I(faces), deg, rs, int(transpose), P(out))
        return out

    def schwarz_apply_weights(self, x, core_faces, deg, rs):
        core_faces = np.ascontiguousarray(core_faces, dtype=np.int32)
        out = np.zeros_like(x)
        self.lib.oracle_schwarz_apply_weights(P(np.ascontiguousarray(x, dtype=np.float64)), I(core_faces), deg, rs, P(out))
        return out

    def schwarz_apply_over_subdomain(self, elem, faces, rs, u_res):
        elem = np.ascontiguousarray(elem, dtype=np.int32); faces = np.ascontiguousarray(faces, dtype=np.int32).reshape(-1)
        out = np.zeros_like(u_res)
        self.lib.oracle_schwarz_apply_over_subdomain(len(elem), I(elem), I(faces), rs, P(np.ascontiguousarray(u_res)), P(out))
        return out

    def schwarz_iterate(self, md, u, r, subdomain_iter, atol, rtol):
        """md: flat metadata (sub_first, sub_elem, sub_faces, sub_core_faces, num_nodes_overlap); returns (u_new, final_iter, final_res)"""
        u = u.copy()
        n = len(md.sub_first) - 1
        it = np.zeros(n, dtype=np.int32); res = np.zeros(n)
        f = self.lib.oracle_schwarz_iterate
        f.argtypes = [ctypes.c_int, ip, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, dp, dp, ip, dp]
        sf = np.ascontiguousarray(md.sub_faces, dtype=np.int32).reshape(-1)
        sc = np.ascontiguousarray(md.sub_core_faces, dtype=np.int32).reshape(-1)
        f(n, I(np.ascontiguousarray(md.sub_first, dtype=np.int32)), I(np.ascontiguousarray(md.sub_elem, dtype=np.int32)), I(sf), I(sc),
          int(md.num_nodes_overlap), int(subdomain_iter), float(atol), float(rtol), P(u), P(np.ascontiguousarray(r)), I(it), P(res))
        return u, it, res

    def schwarz_smoother(self, md, u, rhs, iterations, subdomain_iter, atol, rtol):
        """d4est_solver_multigrid_smoother_schwarz: returns (u_new, r = rhs - A u_new)"""
        u = u.copy(); r = np.zeros_like(u)
        n = len(md.sub_first) - 1
        f = self.lib.oracle_schwarz_smoother
        f.argtypes = [ctypes.c_int, ip, ip, ip, ip, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int, dp, dp, dp]
        sf = np.ascontiguousarray(md.sub_faces, dtype=np.int32).reshape(-1)
        sc = np.ascontiguousarray(md.sub_core_faces, dtype=np.int32).reshape(-1)
        f(n, I(np.ascontiguousarray(md.sub_first, dtype=np.int32)), I(np.ascontiguousarray(md.sub_elem, dtype=np.int32)), I(sf), I(sc),
          int(md.num_nodes_overlap), int(subdomain_iter), float(atol), float(rtol), int(iterations), P(u), P(np.ascontiguousarray(rhs)), P(r))
        return u, r

    def compute_dudr(self, mesh, u):
        d = [np.zeros(mesh.local_nodes) for _ in range(3)]
        self.lib.oracle_laplacian_compute_dudr(mesh.n_elements, I(mesh.deg), I(mesh.nodal_stride), P(u), P(d[0]), P(d[1]), P(d[2]))
        return d


_oracle = {}


def load(native=False):
    if native not in _oracle:
        _oracle[native] = Oracle(ctypes.CDLL(build(native)))
    return _oracle[native]
