"""ctypes binding of libd4est_hip.so (include/d4est_hip.h).

Only plumbing lives here: argument marshalling and a ``Plan`` class whose methods
take torch CUDA tensors and pass their ``data_ptr()`` through the C-ABI.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libd4est_hip.so")

_c_int_p = ctypes.POINTER(ctypes.c_int)
_c_double_p = ctypes.POINTER(ctypes.c_double)
_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/d4est_hip.h one to one
SIGNATURES = {
    "d4est_hip_version": (ctypes.c_char_p, []),
    "d4est_hip_device_count": (ctypes.c_int, []),
    "d4est_hip_table": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_double_p]),
    "d4est_hip_malloc": (_vp, [ctypes.c_size_t]),
    "d4est_hip_free": (None, [_vp]),
    "d4est_hip_memcpy_h2d": (None, [_vp, _vp, ctypes.c_size_t]),
    "d4est_hip_memcpy_d2h": (None, [_vp, _vp, ctypes.c_size_t]),
    "d4est_hip_memset": (None, [_vp, ctypes.c_int, ctypes.c_size_t]),
    "d4est_hip_device_synchronize": (None, []),
    "d4est_hip_plan_create": (_vp, [ctypes.c_int, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int]),
    "d4est_hip_plan_destroy": (None, [_vp]),
    "d4est_hip_plan_set_stream": (None, [_vp, _vp]),
    "d4est_hip_plan_set_tuning": (None, [_vp, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_plan_last_kernel": (ctypes.c_char_p, [_vp]),
    "d4est_hip_plan_face_path": (ctypes.c_char_p, [_vp]),
    "d4est_hip_plan_local_nodes": (ctypes.c_int, [_vp]),
    "d4est_hip_plan_local_nodes_quad": (ctypes.c_int, [_vp]),
    "d4est_hip_plan_stream_mode": (ctypes.c_int, [_vp]),
    "d4est_hip_plan_n_elements": (ctypes.c_int, [_vp]),
    "d4est_hip_plan_set_geometry": (None, [_vp, _vp, _vp, ctypes.c_int]),
    "d4est_hip_apply_stiffness_matrix": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_mass_matrix": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_galerkin_integral": (None, [_vp, _vp, _vp]),
    "d4est_hip_interpolate": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_weighted_mass_matrix": (None, [_vp, _vp, _vp, _vp]),
    "d4est_hip_apply_inverse_mass_matrix": (None, [_vp, _vp, _vp]),
    "d4est_hip_plan_face_nodes": (ctypes.c_int, [_vp]),
    "d4est_hip_apply_slicer": (None, [_vp, _vp, ctypes.c_int, _vp]),
    "d4est_hip_apply_lift": (None, [_vp, _vp, ctypes.c_int, _vp]),
    "d4est_hip_apply_dij": (None, [_vp, _vp, ctypes.c_int, _vp]),
    "d4est_hip_apply_dij_transpose": (None, [_vp, _vp, ctypes.c_int, _vp]),
    "d4est_hip_apply_mij": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_invmij": (None, [_vp, _vp, _vp]),
    "d4est_hip_compute_dudr": (None, [_vp, _vp, _vp, _vp, _vp]),
    "d4est_hip_build_sides": (ctypes.c_int, [ctypes.c_int, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, _vp, _vp, _vp,
                                             _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d4est_hip_topology_table": (ctypes.c_int, [ctypes.c_int, _vp]),
    "d4est_hip_plan_set_faces": (None, [_vp, _c_int_p, _c_int_p, _c_int_p, _c_int_p, _c_int_p, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _c_int_p, _c_int_p]),
    "d4est_hip_plan_set_hanging": (None, [_vp, _vp, _vp, _vp, _vp]),
    "d4est_hip_plan_set_geometry_numerical": (None, [_vp, _vp, ctypes.c_int]),
    "d4est_hip_plan_set_geometry_brick": (None, [_vp, _vp, ctypes.c_double, _vp]),
    "d4est_hip_plan_set_geometry_analytic": (None, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_double]),
    "d4est_hip_plan_set_mortar_geometry_analytic": (None, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_double]),
    "d4est_hip_plan_set_mortar_geometry_brick": (None, [_vp, _vp, ctypes.c_double, _vp]),
    "d4est_hip_transfer_create": (_vp, [ctypes.c_int, _vp, _vp, _vp]),
    "d4est_hip_transfer_destroy": (None, [_vp]),
    "d4est_hip_transfer_set_stream": (None, [_vp, _vp]),
    "d4est_hip_transfer_coarse_nodes": (ctypes.c_longlong, [_vp]),
    "d4est_hip_transfer_fine_nodes": (ctypes.c_longlong, [_vp]),
    "d4est_hip_transfer_prolong": (None, [_vp, _vp, _vp]),
    "d4est_hip_transfer_restrict": (None, [_vp, _vp, _vp]),
    "d4est_hip_transfer_project": (None, [_vp, _vp, _vp]),
    "d4est_hip_schwarz_create": (_vp, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    "d4est_hip_schwarz_destroy": (None, [_vp]),
    "d4est_hip_schwarz_nodal_size": (ctypes.c_longlong, [_vp]),
    "d4est_hip_schwarz_restricted_nodal_size": (ctypes.c_longlong, [_vp]),
    "d4est_hip_schwarz_condensed_copies": (ctypes.c_int, [_vp]),
    "d4est_hip_schwarz_restrict_field": (None, [_vp, _vp, _vp]),
    "d4est_hip_schwarz_apply_over_subdomains": (None, [_vp, _vp, _vp]),
    "d4est_hip_schwarz_add_correction": (None, [_vp, _vp, _vp]),
    "d4est_hip_schwarz_iterate": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_double]),
    "d4est_hip_schwarz_smooth": (None, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double]),
    "d4est_hip_schwarz_get_info": (None, [_vp, _vp, _vp]),
    "d4est_hip_plan_set_sipg": (None, [_vp, ctypes.c_double, ctypes.c_int]),
    "d4est_hip_plan_set_mortar_geometry": (None, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_int]),
    "d4est_hip_plan_set_dirichlet_values": (None, [_vp, _vp, ctypes.c_int]),
    "d4est_hip_plan_set_robin_values": (None, [_vp, _vp, _vp, ctypes.c_int]),
    "d4est_hip_plan_trace_size": (ctypes.c_longlong, [_vp]),
    "d4est_hip_plan_ghost_trace_size": (ctypes.c_longlong, [_vp]),
    "d4est_hip_compute_ghost_traces": (None, [_vp, _vp, _vp]),
    "d4est_hip_compute_face_traces": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_flux": (None, [_vp, _vp, _vp, _vp]),
    "d4est_hip_apply_aij": (None, [_vp, _vp, _vp, _vp]),
    "d4est_hip_build_rhs_with_strong_bc": (None, [_vp, _vp, ctypes.c_int, _vp]),
    "d4est_hip_build_rhs_with_strong_bc_host": (None, [_vp, _c_double_p, ctypes.c_int, _c_double_p]),
    "d4est_hip_plan_set_lhs_coefficient": (None, [_vp, _vp]),
    "d4est_hip_plan_matrix_nodes": (ctypes.c_longlong, [_vp]),
    "d4est_hip_compute_weighted_mass_blocks": (None, [_vp, _vp, _vp]),
    "d4est_hip_plan_set_lhs_element_blocks": (None, [_vp, _vp, _vp]),
    "d4est_hip_plan_set_lhs_galerkin_chain": (None, [_vp, ctypes.c_int, _vp, _vp]),
    "d4est_hip_transfer_fine_matrix_nodes": (ctypes.c_longlong, [_vp]),
    "d4est_hip_transfer_coarse_matrix_nodes": (ctypes.c_longlong, [_vp]),
    "d4est_hip_transfer_galerkin_blocks": (None, [_vp, _vp, _vp, ctypes.c_int]),
    "d4est_hip_plan_set_comm": (None, [_vp, _vp, _vp, _vp]),
    "d4est_hip_apply_lhs": (None, [_vp, _vp, _vp]),
    "d4est_hip_cheby_iterate": (None, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]),
    "d4est_hip_cheby_update": (None, [_vp, ctypes.c_int, _vp, _vp, ctypes.c_double, ctypes.c_double, _vp, _vp, _vp]),
    "d4est_hip_cg_eigs": (ctypes.c_double, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _c_double_p]),
    "d4est_hip_copy_blocks": (None, [_vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "d4est_hip_plan_trace_offset": (ctypes.c_longlong, [_vp, ctypes.c_int]),
    "d4est_hip_plan_side_blocks": (ctypes.c_int, [_vp, ctypes.c_int]),
    "d4est_hip_reorient_face_order": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_face_reorder_code": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_plan_trace_offset_sub": (ctypes.c_longlong, [_vp, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_plan_ghost_trace_offset_sub": (ctypes.c_longlong, [_vp, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_plan_trace_block_len_sub": (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_plan_ghost_trace_offset": (ctypes.c_longlong, [_vp, ctypes.c_int]),
    "d4est_hip_plan_trace_block_len": (ctypes.c_int, [_vp, ctypes.c_int]),
    "d4est_hip_vec_dot": (None, [_vp, ctypes.c_int, _vp, _vp, _vp]),
    "d4est_hip_comm_unique_id_bytes": (ctypes.c_int, []),
    "d4est_hip_comm_get_unique_id": (None, [_vp]),
    "d4est_hip_comm_create": (_vp, [_vp, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_comm_try_create": (_vp, [_vp, ctypes.c_int, ctypes.c_int]),
    "d4est_hip_comm_destroy": (None, [_vp]),
    "d4est_hip_comm_rank": (ctypes.c_int, [_vp]),
    "d4est_hip_comm_size": (ctypes.c_int, [_vp]),
    "d4est_hip_comm_nccl_count": (ctypes.c_int, [_vp]),
    "d4est_hip_plan_set_rccl_exchange": (_vp, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "d4est_hip_rccl_exchange_destroy": (None, [_vp]),
    "d4est_hip_rccl_exchange_count": (ctypes.c_longlong, [_vp]),
    "d4est_hip_rccl_exchange_send_doubles": (ctypes.c_longlong, [_vp]),
    "d4est_hip_rccl_exchange_recv_doubles": (ctypes.c_longlong, [_vp]),
    "d4est_hip_comm_sendrecv": (None, [_vp, _vp, ctypes.c_int, _vp, _vp, _vp, _vp, _vp]),
    "d4est_hip_comm_allreduce_sum": (None, [_vp, _vp, _vp, ctypes.c_int]),
    "d4est_hip_apply_stiffness_matrix_host": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_aij_host": (None, [_vp, _vp, _vp]),
    "d4est_hip_apply_lhs_host": (None, [_vp, _vp, _vp]),
    "d4est_hip_cheby_iterate_host": (None, [_vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_int]),
    "d4est_hip_cg_eigs_host": (ctypes.c_double, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, _c_double_p]),
    "d4est_hip_plan_set_jacobian": (None, [_vp, _vp, ctypes.c_int]),
    "d4est_hip_host_alloc": (_vp, [ctypes.c_size_t]),
    "d4est_hip_host_free": (None, [_vp]),
    "d4est_hip_memcpy_h2d_async": (None, [_vp, _vp, _vp, ctypes.c_size_t]),
    "d4est_hip_memcpy_d2h_async": (None, [_vp, _vp, _vp, ctypes.c_size_t]),
    "d4est_hip_plan_synchronize": (None, [_vp]),
}

TABLE = {
    "lobatto_nodes": 0, "lobatto_weights": 1, "gauss_nodes": 2, "gauss_weights": 3,
    "dij": 4, "mij": 5, "invmij": 6, "lobatto_to_gauss": 7,
    "p_prolong": 8, "hp_prolong": 9, "p_restrict": 10, "hp_restrict": 11,
}

_lib = None


def load_library(path=None):
    """Load libd4est_hip.so and attach the signatures.  Fails loudly when absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # D4EST_HIP_LIBRARY: another build of the same library (kernel experiments: tools/build_variant.sh)
    p = path or os.environ.get("D4EST_HIP_LIBRARY") or LIB_PATH
    if not os.path.exists(p):
        raise RuntimeError(
            "libd4est_hip.so not found at %s -- build it with `python -m disco4est_amd.build` "
            "(or __graft_entry__.build()); there is no CPU fallback" % p)
    # Load order matters in a process that also uses torch: torch ships its own HIP runtime under the same soname as /opt/rocm's.
    # Whichever is mapped first serves both; torch aborts on the system one, while this library runs on either.  So torch goes first.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def table(name, deg_a, deg_b=0):
    """1-D operator table as a numpy array (host-side, no GPU needed)."""
    lib = load_library()
    tid = TABLE[name]
    n = lib.d4est_hip_table(tid, int(deg_a), int(deg_b), None)
    out = np.empty(n, dtype=np.float64)
    lib.d4est_hip_table(tid, int(deg_a), int(deg_b), out.ctypes.data_as(_c_double_p))
    return out


def _ptr(t):
    """Device pointer of a contiguous float64 torch CUDA tensor."""
    import torch
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), \
        "expected a contiguous float64 CUDA tensor"
    return ctypes.c_void_p(t.data_ptr())


def _iarr(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_c_int_p)


class Plan:
    """Host mirror of a d4est mesh level: element arrays + geometric factors on the device.

    Argument meaning follows d4est_element_data_t (src/Mesh/d4est_element_data.h:13-48) and
    d4est_mesh_data_t (src/Mesh/d4est_mesh.h:123-169) of the reference.
    """

    def __init__(self, deg, deg_quad, nodal_stride, quad_stride, quad_type=0, stream=None):
        self.lib = load_library()
        self._keep = [_iarr(deg), _iarr(deg_quad), _iarr(nodal_stride), _iarr(quad_stride)]
        n = len(self._keep[0][0])
        self.handle = self.lib.d4est_hip_plan_create(n, self._keep[0][1], self._keep[1][1], self._keep[2][1],
                                                     self._keep[3][1], int(quad_type))
        self.n_elements = n
        self.torch_stream = None
        self.local_nodes = self.lib.d4est_hip_plan_local_nodes(self.handle)
        self.local_nodes_quad = self.lib.d4est_hip_plan_local_nodes_quad(self.handle)
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        """stream: a torch.cuda.Stream (its raw hipStream_t is passed through) or an int handle."""
        h = getattr(stream, "cuda_stream", stream)
        self.torch_stream = stream if hasattr(stream, "cuda_stream") else None   # parallel.attach makes it current around the exchange
        self.lib.d4est_hip_plan_set_stream(self.handle, ctypes.c_void_p(int(h)))

    def last_kernel(self):
        return self.lib.d4est_hip_plan_last_kernel(self.handle).decode()

    def face_path(self):
        """'direct' or 'two-phase': which face kernels the full operator runs on this plan."""
        return self.lib.d4est_hip_plan_face_path(self.handle).decode()

    def stream_mode(self):
        """1 when the large-plan kernels move once-touched data with the non-temporal hint (tuning key 12; automatic from 320 MB per apply)."""
        return self.lib.d4est_hip_plan_stream_mode(self.handle)

    def set_tuning(self, key, value):
        self.lib.d4est_hip_plan_set_tuning(self.handle, int(key), int(value))

    def set_geometry(self, J_quad, rst_xyz_quad):
        """J_quad[local_nodes_quad], rst_xyz_quad[9*local_nodes_quad] (reference SoA layout);
        numpy arrays (host) or torch CUDA tensors (device)."""
        if isinstance(J_quad, np.ndarray):
            J = np.ascontiguousarray(J_quad, dtype=np.float64)
            R = np.ascontiguousarray(rst_xyz_quad, dtype=np.float64).reshape(-1)
            assert J.size == self.local_nodes_quad and R.size == 9 * self.local_nodes_quad
            self.lib.d4est_hip_plan_set_geometry(self.handle, J.ctypes.data_as(_vp), R.ctypes.data_as(_vp), 0)
        else:
            assert J_quad.numel() == self.local_nodes_quad and rst_xyz_quad.numel() == 9 * self.local_nodes_quad
            self.lib.d4est_hip_plan_set_geometry(self.handle, _ptr(J_quad), _ptr(rst_xyz_quad), 1)

    def set_geometry_numerical(self, xyz_lobatto):
        """GEOM_COMPUTE_NUMERICAL volume factors from the nodal coordinates: xyz_lobatto = (x, y, z) arrays of local_nodes entries
        (numpy) or one torch CUDA tensor of 3*local_nodes entries"""
        if isinstance(xyz_lobatto, (list, tuple)) or isinstance(xyz_lobatto, np.ndarray):
            X = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float64).reshape(-1) for a in xyz_lobatto]))
            assert X.size == 3 * self.local_nodes
            self.lib.d4est_hip_plan_set_geometry_numerical(self.handle, X.ctypes.data_as(_vp), 0)
        else:
            assert xyz_lobatto.numel() == 3 * self.local_nodes
            self.lib.d4est_hip_plan_set_geometry_numerical(self.handle, _ptr(xyz_lobatto), 1)

    def apply_stiffness_matrix(self, u, Au):
        assert u.numel() == self.local_nodes and Au.numel() == self.local_nodes
        self.lib.d4est_hip_apply_stiffness_matrix(self.handle, _ptr(u), _ptr(Au))

    def apply_mass_matrix(self, u, Mu):
        assert u.numel() == self.local_nodes and Mu.numel() == self.local_nodes
        self.lib.d4est_hip_apply_mass_matrix(self.handle, _ptr(u), _ptr(Mu))

    def apply_galerkin_integral(self, f_quad, out):
        assert f_quad.numel() == self.local_nodes_quad and out.numel() == self.local_nodes
        self.lib.d4est_hip_apply_galerkin_integral(self.handle, _ptr(f_quad), _ptr(out))

    def interpolate(self, u, u_quad):
        assert u.numel() == self.local_nodes and u_quad.numel() == self.local_nodes_quad
        self.lib.d4est_hip_interpolate(self.handle, _ptr(u), _ptr(u_quad))

    def set_geometry_brick(self, elem_dq, root_len, extents, mortars=False):
        """device-generated factors of the reference's brick geometry (volume, or the mortars when mortars=True)"""
        dq = _iarr(elem_dq)
        ex = np.ascontiguousarray(extents, dtype=np.float64)
        fn = self.lib.d4est_hip_plan_set_mortar_geometry_brick if mortars else self.lib.d4est_hip_plan_set_geometry_brick
        fn(self.handle, dq[1], float(root_len), ex.ctypes.data_as(_vp))

    def set_geometry_analytic(self, geom_type, params, tree, q, dq, root_len, mortars=False, ghost=None):
        """factors of an analytic tree map generated on the device (geom_type 1 = cubed_sphere_7tree, params = (R0, R1,
        compactify_inner_shell)); tree / q[n,3] / dq: where every element sits in the forest; mortars=True: the mortar factors
        (ghost = (tree, q, dq) of the ghost elements)"""
        pr = np.ascontiguousarray(params, dtype=np.float64)
        t, qq, d = _iarr(tree), _iarr(np.asarray(q).reshape(-1)), _iarr(dq)
        if not mortars:
            self.lib.d4est_hip_plan_set_geometry_analytic(self.handle, int(geom_type), pr.ctypes.data_as(_vp), t[1], qq[1], d[1], float(root_len))
            return
        gt, gq, gd = (_iarr(ghost[0]), _iarr(np.asarray(ghost[1]).reshape(-1)), _iarr(ghost[2])) if ghost is not None else \
            (_iarr(np.zeros(0)), _iarr(np.zeros(0)), _iarr(np.zeros(0)))
        self.lib.d4est_hip_plan_set_mortar_geometry_analytic(self.handle, int(geom_type), pr.ctypes.data_as(_vp), t[1], qq[1], d[1],
                                                             gt[1], gq[1], gd[1], float(root_len))

    def apply_weighted_mass_matrix(self, u, coeff_quad, out):
        assert u.numel() == self.local_nodes and out.numel() == self.local_nodes
        assert coeff_quad.numel() == self.local_nodes_quad
        self.lib.d4est_hip_apply_weighted_mass_matrix(self.handle, _ptr(u), _ptr(coeff_quad), _ptr(out))

    def apply_inverse_mass_matrix(self, x, out):
        assert x.numel() == self.local_nodes and out.numel() == self.local_nodes
        self.lib.d4est_hip_apply_inverse_mass_matrix(self.handle, _ptr(x), _ptr(out))

    def apply_mij(self, x, out):
        assert x.numel() == self.local_nodes and out.numel() == self.local_nodes
        self.lib.d4est_hip_apply_mij(self.handle, _ptr(x), _ptr(out))

    def apply_invmij(self, x, out):
        assert x.numel() == self.local_nodes and out.numel() == self.local_nodes
        self.lib.d4est_hip_apply_invmij(self.handle, _ptr(x), _ptr(out))

    def apply_slicer(self, x, face, out_face):
        assert x.numel() == self.local_nodes and out_face.numel() == self.lib.d4est_hip_plan_face_nodes(self.handle)
        self.lib.d4est_hip_apply_slicer(self.handle, _ptr(x), int(face), _ptr(out_face))

    def apply_lift(self, x_face, face, out):
        assert out.numel() == self.local_nodes and x_face.numel() == self.lib.d4est_hip_plan_face_nodes(self.handle)
        self.lib.d4est_hip_apply_lift(self.handle, _ptr(x_face), int(face), _ptr(out))

    def apply_dij(self, x, direction, out, transpose=False):
        assert x.numel() == self.local_nodes and out.numel() == self.local_nodes
        fn = self.lib.d4est_hip_apply_dij_transpose if transpose else self.lib.d4est_hip_apply_dij
        fn(self.handle, _ptr(x), int(direction), _ptr(out))

    def compute_dudr(self, u, d0, d1, d2):
        for t in (u, d0, d1, d2):
            assert t.numel() == self.local_nodes
        self.lib.d4est_hip_compute_dudr(self.handle, _ptr(u), _ptr(d0), _ptr(d1), _ptr(d2))

    # ---- faces
    def set_faces(self, sides, penalty_prefactor=10.0, penalty_fcn=0, brick=None, analytic=None):
        """sides: the dict of mesh.BrickMesh.build_sides() (reference-layout side list + mortar factors);
        brick = (elem_dq, root_len, extents): generate the mortar factors of the brick geometry on the device instead"""
        keep = [_iarr(sides[k]) for k in ("side_nbr", "side_nbr_face", "side_reorder", "side_mortar_stride", "side_bndry_stride",
                                         "ghost_deg", "ghost_deg_quad")]
        self._keep_sides = keep
        if "side_hang" in sides and np.any(np.asarray(sides["side_hang"]) != 0):
            hk = [_iarr(sides[k]) for k in ("side_hang", "side_sub", "side_nbr4", "side_orientation")]
            self._keep_hang = hk
            self.lib.d4est_hip_plan_set_hanging(self.handle, hk[0][1], hk[1][1], hk[2][1], hk[3][1])
        self.lib.d4est_hip_plan_set_faces(self.handle, keep[0][1], keep[1][1], keep[2][1], keep[3][1], keep[4][1],
                                          int(sides["total_mortar_nodes"]), int(sides["total_bndry_nodes"]),
                                          len(keep[5][0]), keep[5][1], keep[6][1])
        self.lib.d4est_hip_plan_set_sipg(self.handle, float(penalty_prefactor), int(penalty_fcn))
        if brick is not None:
            self.set_geometry_brick(brick[0], brick[1], brick[2], mortars=True)
        elif analytic is not None:      # (geom_type, params, tree, q, dq, root_len, ghost)
            self.set_geometry_analytic(*analytic[:6], mortars=True, ghost=analytic[6])
        else:
            arrs = [np.ascontiguousarray(sides[k], dtype=np.float64) for k in ("sj", "n", "drst_m", "drst_p", "hm", "hp")]
            self.lib.d4est_hip_plan_set_mortar_geometry(self.handle, *[a.ctypes.data_as(_vp) for a in arrs], 0)
        self.trace_size = self.lib.d4est_hip_plan_trace_size(self.handle)
        self.ghost_trace_size = self.lib.d4est_hip_plan_ghost_trace_size(self.handle)

    def set_dirichlet_values(self, g):
        if g is None:
            self.lib.d4est_hip_plan_set_dirichlet_values(self.handle, None, 0)
        else:
            g = np.ascontiguousarray(g, dtype=np.float64)
            self.lib.d4est_hip_plan_set_dirichlet_values(self.handle, g.ctypes.data_as(_vp), 0)

    def set_robin_values(self, coeff_quad, rhs_quad):
        """Robin data at the boundary sides' mortar quadrature nodes (indexed like sj); None switches back to Dirichlet"""
        if coeff_quad is None:
            self.lib.d4est_hip_plan_set_robin_values(self.handle, None, None, 0)
            return
        c = np.ascontiguousarray(coeff_quad, dtype=np.float64)
        r = np.ascontiguousarray(rhs_quad, dtype=np.float64)
        self.lib.d4est_hip_plan_set_robin_values(self.handle, c.ctypes.data_as(_vp), r.ctypes.data_as(_vp), 0)

    def compute_ghost_traces(self, u_ghost, ghost_trace):
        assert ghost_trace.numel() == self.ghost_trace_size
        self.lib.d4est_hip_compute_ghost_traces(self.handle, _ptr(u_ghost), _ptr(ghost_trace))

    def compute_face_traces(self, u, trace):
        assert u.numel() == self.local_nodes and trace.numel() == self.trace_size
        self.lib.d4est_hip_compute_face_traces(self.handle, _ptr(u), _ptr(trace))

    def apply_flux(self, trace, ghost_trace, Au):
        self.lib.d4est_hip_apply_flux(self.handle, _ptr(trace), _ptr(ghost_trace) if ghost_trace is not None else None, _ptr(Au))

    def apply_aij(self, u, Au, ghost_trace=None):
        assert u.numel() == self.local_nodes and Au.numel() == self.local_nodes
        self.lib.d4est_hip_apply_aij(self.handle, _ptr(u), _ptr(ghost_trace) if ghost_trace is not None else None, _ptr(Au))

    # ---- smoother loops
    EXCHANGE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
    ALLREDUCE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int)

    def set_comm(self, exchange=None, allreduce=None):
        """exchange(phase, trace_ptr, ghost_trace_ptr), allreduce(scalars_ptr, n): python callables (pointers are ints)"""
        def guarded(fn):
            # ctypes prints and DROPS an exception raised inside a callback; the C caller would carry on with a stale ghost trace
            # or un-reduced scalars.  A failed exchange aborts the process, like every other failure of the library (D4EST_HIP_ABORT).
            def call(*args):
                try:
                    fn(*args)
                except BaseException:
                    import os
                    import sys
                    import traceback
                    sys.stderr.write("[D4EST_HIP_ABORT] exception in a communication callback:\n")
                    traceback.print_exc()
                    sys.stderr.flush()
                    os.abort()
            return call

        self._cb_ex = self.EXCHANGE_FN(guarded(lambda ctx, ph, a, b: exchange(ph, a, b))) if exchange else None
        self._cb_ar = self.ALLREDUCE_FN(guarded(lambda ctx, p, n: allreduce(p, n))) if allreduce else None
        self.lib.d4est_hip_plan_set_comm(self.handle, ctypes.cast(self._cb_ex, ctypes.c_void_p) if self._cb_ex else None,
                                         ctypes.cast(self._cb_ar, ctypes.c_void_p) if self._cb_ar else None, None)

    def build_rhs_with_strong_bc(self, f, rhs, f_on_quad=False):
        """rhs = M f - A(0) with the boundary data currently set on the plan (d4est_laplacian_build_rhs_with_strong_bc)"""
        self.lib.d4est_hip_build_rhs_with_strong_bc(self.handle, _ptr(f), int(bool(f_on_quad)), _ptr(rhs))

    def set_lhs_coefficient(self, coeff_quad):
        """zeroth-order term of apply_lhs (+ V^T W J c V u): a float64 CUDA tensor of local_nodes_quad entries, or None for the pure
        Laplacian.  The VALUES ARE CAPTURED by this call (a plan-owned copy; the tensor is not read afterwards): call it again after
        every change of u0.  The tensor must be complete on the plan's stream (or the device synchronised) when this is called."""
        self._lhs_coeff = coeff_quad
        if coeff_quad is not None:
            assert coeff_quad.numel() == self.local_nodes_quad
        self.lib.d4est_hip_plan_set_lhs_coefficient(self.handle, _ptr(coeff_quad) if coeff_quad is not None else None)

    # ---- the multigrid matrix operator: the zeroth-order term on coarse levels (d4est_solver_multigrid_matrix_operator.c)
    def matrix_nodes(self):
        """d4est_mesh_get_local_matrix_nodes: sum of (deg+1)^6"""
        return self.lib.d4est_hip_plan_matrix_nodes(self.handle)

    def compute_weighted_mass_blocks(self, coeff_quad, blocks):
        """QUAD_COMPUTE_MATRIX for every element: blocks = V^T (W J coeff) V, dense, consecutive (coeff_quad None: the mass matrix)"""
        assert blocks.numel() == self.matrix_nodes()
        self.lib.d4est_hip_compute_weighted_mass_blocks(self.handle, _ptr(coeff_quad) if coeff_quad is not None else None, _ptr(blocks))

    def set_lhs_element_blocks(self, blocks, block_offset=None):
        """zeroth-order term of apply_lhs as dense element blocks (read at every apply: the tensor is kept alive here); block_offset:
        per-element offsets in doubles (a Schwarz subdomain plan: the mesh element's block) or None = consecutive"""
        self._lhs_blocks = blocks
        off = None
        if block_offset is not None:
            off = np.ascontiguousarray(block_offset, dtype=np.int64)
            assert off.size == self.n_elements
        self.lib.d4est_hip_plan_set_lhs_element_blocks(self.handle, _ptr(blocks) if blocks is not None else None,
                                                       off.ctypes.data_as(_vp) if off is not None else None)

    def set_lhs_galerkin_chain(self, transfers, fine_plan):
        """zeroth-order term as the Galerkin chain T_0^T .. T_k^T (V^T W J c V)_fine T_k .. T_0 (transfers[0] starts at this plan's
        level; fine_plan carries the coefficient); transfers = [] switches it off"""
        self._lhs_chain = (list(transfers), fine_plan)
        arr = (ctypes.c_void_p * max(len(transfers), 1))(*[t.handle for t in transfers])
        self.lib.d4est_hip_plan_set_lhs_galerkin_chain(self.handle, len(transfers), arr, fine_plan.handle if fine_plan is not None else None)

    def apply_lhs(self, u, Au):
        self.lib.d4est_hip_apply_lhs(self.handle, _ptr(u), _ptr(Au))

    def cheby_iterate(self, u, rhs, Au, r, iters, lmin, lmax, residual_at_end=1):
        self.lib.d4est_hip_cheby_iterate(self.handle, _ptr(u), _ptr(rhs), _ptr(Au), _ptr(r), int(iters), float(lmin), float(lmax),
                                         int(residual_at_end))

    def cg_eigs(self, u, rhs, Au, imax, use_new=1):
        hist = np.zeros(2 * imax)
        b = self.lib.d4est_hip_cg_eigs(self.handle, _ptr(u), _ptr(rhs), _ptr(Au), int(imax), int(use_new), hist.ctypes.data_as(_c_double_p))
        return b, hist

    def copy_blocks(self, n_blocks, src, src_off, dst, dst_off, length):
        """src/dst: float64 CUDA tensors; src_off/dst_off: int64 CUDA tensors; length: int32 CUDA tensor"""
        self.lib.d4est_hip_copy_blocks(self.handle, int(n_blocks), _ptr(src), ctypes.c_void_p(src_off.data_ptr()), _ptr(dst),
                                       ctypes.c_void_p(dst_off.data_ptr()), ctypes.c_void_p(length.data_ptr()))

    def vec_dot(self, x, y, out):
        self.lib.d4est_hip_vec_dot(self.handle, int(x.numel()), _ptr(x), _ptr(y), _ptr(out))

    def apply_stiffness_matrix_host(self, u_host):
        u = np.ascontiguousarray(u_host, dtype=np.float64)
        assert u.size == self.local_nodes
        out = np.empty_like(u)
        self.lib.d4est_hip_apply_stiffness_matrix_host(self.handle, u.ctypes.data_as(_vp), out.ctypes.data_as(_vp))
        return out

    def destroy(self):
        if self.handle:
            self.lib.d4est_hip_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Transfer:
    """hp-multigrid inter-grid transfer (d4est_hip_transfer_*): items = coarse elements in traversal order,
    hrefine[k] 0 (1 <-> 1, p only) or 1 (8 children <-> parent), degH[k], degh[8k..8k+7]."""

    def __init__(self, hrefine, degH, degh, stream=None):
        self.lib = load_library()
        h, dH, dh = _iarr(hrefine), _iarr(degH), _iarr(degh)
        assert len(dh[0]) == 8 * len(h[0]) and len(dH[0]) == len(h[0])
        self.handle = self.lib.d4est_hip_transfer_create(len(h[0]), h[1], dH[1], dh[1])
        if stream is not None:
            self.lib.d4est_hip_transfer_set_stream(self.handle, ctypes.c_void_p(stream.cuda_stream))
        self.coarse_nodes = self.lib.d4est_hip_transfer_coarse_nodes(self.handle)
        self.fine_nodes = self.lib.d4est_hip_transfer_fine_nodes(self.handle)

    def prolong(self, x_coarse, x_fine):
        assert x_coarse.numel() == self.coarse_nodes and x_fine.numel() == self.fine_nodes
        self.lib.d4est_hip_transfer_prolong(self.handle, _ptr(x_coarse), _ptr(x_fine))

    def restrict(self, x_fine, x_coarse):
        assert x_coarse.numel() == self.coarse_nodes and x_fine.numel() == self.fine_nodes
        self.lib.d4est_hip_transfer_restrict(self.handle, _ptr(x_fine), _ptr(x_coarse))

    def galerkin_blocks(self, fine_blocks, coarse_blocks, literal_window=False):
        """coarse block k = sum over the item's children of P^T M P (the multigrid matrix operator's restriction callback);
        literal_window: the reference's arithmetic to the letter on items with eight children (d4est_operators.c:651)"""
        assert fine_blocks.numel() == self.lib.d4est_hip_transfer_fine_matrix_nodes(self.handle)
        assert coarse_blocks.numel() == self.lib.d4est_hip_transfer_coarse_matrix_nodes(self.handle)
        self.lib.d4est_hip_transfer_galerkin_blocks(self.handle, _ptr(fine_blocks), _ptr(coarse_blocks), int(bool(literal_window)))

    def project(self, x_fine, x_coarse):
        """L2 projection onto the coarse space (apply_p_restrict / apply_hp_restrict per item)"""
        assert x_fine.numel() == self.fine_nodes and x_coarse.numel() == self.coarse_nodes
        self.lib.d4est_hip_transfer_project(self.handle, _ptr(x_fine), _ptr(x_coarse))

    def destroy(self):
        if self.handle:
            self.lib.d4est_hip_transfer_destroy(self.handle)
            self.handle = None
