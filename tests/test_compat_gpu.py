"""The reference's own prototypes (include/d4est_hip_compat.h -> libd4est_hip_compat.so) driven from plain C99 with host pointers
(tests/c/compat_probe.c) and compared with the oracle: SURVEY.md section 8b's "same signatures" export list."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "disco4est_amd")


def _compile(tmp_path, oracle):
    exe = str(tmp_path / "compat_probe")
    odir = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + odir,
                           os.path.join(ROOT, "tests", "c", "compat_probe.c"), "-L" + LIBDIR, "-ld4est_hip_compat", "-ld4est_hip",
                           "-L" + odir, "-ld4est_oracle", "-lm", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath," + odir, "-o", exe])
    return exe


def test_compat_header_is_c99_and_every_declared_symbol_is_exported(hiplib, oracle, tmp_path):
    """CPU: the header compiles as plain C, the probe links, and libd4est_hip_compat.so exports every function the header declares"""
    assert os.path.exists(_compile(tmp_path, oracle))
    text = open(os.path.join(ROOT, "include", "d4est_hip_compat.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"^(?:void|double|int|d4est_hip_plan_t\s*\*)\s*\*?\s*(\w+)\s*\(", text, flags=re.M))
    assert {"d4est_quadrature_apply_stiffness_matrix", "d4est_operators_apply_hp_restrict", "d4est_laplacian_apply_aij", "cg_eigs",
            "d4est_laplacian_with_opt_apply_aij", "d4est_laplacian_with_opt_apply_stiffness_matrix",
            "d4est_solver_multigrid_smoother_cheby_iterate_aux", "d4est_hip_compat_bind_mesh", "d4est_quadrature_apply_fofufofvlilj",
            "d4est_quadrature_apply_fofufofvlj", "d4est_hip_compat_bind_operator", "d4est_hip_compat_build_rhs_with_strong_bc",
            "d4est_quadrature_compute_mass_matrix", "d4est_operators_compute_PT_mat_P", "d4est_operators_compute_prolong_matrix",
            "d4est_laplacian_build_rhs_with_strong_bc", "d4est_solver_schwarz_iterate", "d4est_operators_reorient_face_data",
            "d4est_operators_apply_flip", "d4est_mortars_project_side_onto_mortar_space", "d4est_mortars_project_mass_mortar_onto_side",
            "d4est_hip_compat_flatten_schwarz_metadata", "d4est_hip_compat_bind_flux"} <= names and len(names) >= 30
    lib = ctypes.CDLL(os.path.join(LIBDIR, "libd4est_hip_compat.so"))
    for n in names:
        getattr(lib, n)


def test_main_header_symbols_exported(hiplib):
    """every function include/d4est_hip.h declares resolves in libd4est_hip.so (no compute call)"""
    text = open(os.path.join(ROOT, "include", "d4est_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set(re.findall(r"\b(d4est_hip_\w+)\s*\(", text))
    names -= {"d4est_hip_exchange_fn", "d4est_hip_allreduce_fn"}
    assert len(names) > 90
    for n in names:
        getattr(hiplib, n)


@pytest.mark.gpu
def test_reference_prototypes_from_plain_c_match_oracle(gpu, hiplib, oracle, tmp_path):
    exe = _compile(tmp_path, oracle)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout[-6000:], out.stderr[-2000:])
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert out.stdout.strip().endswith("ok")
    assert out.stdout.count("rel-inf") >= 370
    for what in ("apply_fofufofvlilj", "apply_fofufofvlj", "MORTAR", "build_rhs_with_strong_bc", "registered apply_lhs accepted",
                 "fofufofvlilj COMPUTE_MATRIX", "compute_mass_matrix MORTAR", "fofufofvlilj MORTAR z", "fofufofvlj MORTAR z",
                 "compute_PT_mat_P children 8", "compute_prolong_matrix children 1", "reorient_face_data", "mortars_project mortar->side 4-4",
                 "d4est_laplacian_build_rhs_with_strong_bc", "apply_aij with matching flux data", "apply_hp_restrict dim 2"):
        assert what in out.stdout


@pytest.mark.gpu
def test_smoother_shims_refuse_another_operator(gpu, hiplib, oracle, tmp_path):
    """cg_eigs / cheby_iterate_aux never call fcns->apply_lhs (the bound plan applies the operator); with the plan's callback registered
    (d4est_hip_compat_bind_operator) a caller that passes a DIFFERENT apply_lhs is aborted instead of silently served the wrong operator"""
    exe = _compile(tmp_path, oracle)
    out = subprocess.run([exe, "mismatch"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "NOT ABORTED" not in out.stdout
    assert "fcns->apply_lhs is not the operator registered" in out.stderr


@pytest.mark.gpu
def test_apply_aij_refuses_other_flux_data(gpu, hiplib, oracle, tmp_path):
    """d4est_laplacian_apply_aij never reads the boundary / penalty callbacks of flux_fcn_data (the plan carries them); with the plan's
    parameters registered (d4est_hip_compat_bind_flux) a caller whose flux data says otherwise is aborted, not served the plan's operator"""
    exe = _compile(tmp_path, oracle)
    out = subprocess.run([exe, "fluxmismatch"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "NOT ABORTED" not in out.stdout
    assert "apply_aij with matching flux data" in out.stdout
    assert "sipg_penalty_prefactor" in out.stderr


@pytest.mark.gpu
def test_schwarz_iterate_and_metadata_flattening_through_the_reference_structs(gpu, hiplib, oracle):
    """d4est_solver_schwarz_iterate with the reference's argument list on a handle bound to the p4est, fed by
    d4est_hip_compat_flatten_schwarz_metadata from the reference's metadata structs (src/Solver/d4est_solver_schwarz_metadata.h:19-90,
    built here with ctypes mirrors of the header's mirrors): flat arrays identical to the Python glue's, correction equal to the oracle's"""
    import numpy as np
    from disco4est_amd import mesh as M
    from disco4est_amd.schwarz import Schwarz
    lib = ctypes.CDLL(os.path.join(LIBDIR, "libd4est_hip_compat.so"))

    class Elem(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int) for n in ("mpirank", "tree", "tree_quadid", "id", "deg")] + [("faces", ctypes.c_int * 3), ("core_faces", ctypes.c_int * 3)] + \
                   [(n, ctypes.c_int) for n in ("is_core", "nodal_size", "nodal_stride", "restricted_nodal_size", "restricted_nodal_stride")]

    class Sub(ctypes.Structure):
        _fields_ = [("mpirank", ctypes.c_int), ("subdomain_id", ctypes.c_int), ("core_id", ctypes.c_int), ("element_metadata", ctypes.POINTER(Elem))] + \
                   [(n, ctypes.c_int) for n in ("core_deg", "core_tree", "num_elements", "restricted_nodal_size", "restricted_nodal_stride", "nodal_size",
                                                "nodal_stride", "element_stride")]

    class Meta(ctypes.Structure):
        _fields_ = [(n, ctypes.c_int) for n in ("num_nodes_overlap", "restricted_nodal_size", "nodal_size", "num_subdomains", "num_elements")] + \
                   [("subdomain_metadata", ctypes.POINTER(Sub)), ("element_metadata", ctypes.POINTER(Elem)), ("subdomain_ghostdata", ctypes.c_void_p),
                    ("element_ghostdata", ctypes.c_void_p), ("d4est_ghost", ctypes.c_void_p), ("input_section", ctypes.c_char_p)]

    m = M.BrickMesh(1, 3)
    mp = M.SineMap(0.03)
    J, rst = m.geometry(mp); sides = m.build_sides(mp)
    sz = Schwarz(m, sides, J, rst, 2, 6, 1e-15, 1e-15)
    md = sz.metadata
    ne = int(md.sub_first[-1])
    elems = (Elem * ne)()
    subs = (Sub * md.num_subdomains)()
    for i in range(md.num_subdomains):
        k0, k1 = int(md.sub_first[i]), int(md.sub_first[i + 1])
        for k in range(k0, k1):
            elems[k].id = int(md.sub_elem[k]); elems[k].deg = int(m.deg[md.sub_elem[k]])
            for f in range(3):
                elems[k].faces[f] = int(md.sub_faces[k][f]); elems[k].core_faces[f] = int(md.sub_core_faces[k][f])
        subs[i].subdomain_id = i; subs[i].num_elements = k1 - k0; subs[i].element_stride = k0
        subs[i].element_metadata = ctypes.cast(ctypes.byref(elems, k0 * ctypes.sizeof(Elem)), ctypes.POINTER(Elem))
    meta = Meta(num_nodes_overlap=2, num_subdomains=md.num_subdomains, num_elements=ne, subdomain_metadata=subs, element_metadata=elems)
    sf = np.zeros(md.num_subdomains + 1, np.int32); se = np.zeros(ne, np.int32); f3 = np.zeros(3 * ne, np.int32); c3 = np.zeros(3 * ne, np.int32)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.d4est_hip_compat_flatten_schwarz_metadata(ctypes.byref(meta), vp(sf), vp(se), vp(f3), vp(c3))
    np.testing.assert_array_equal(sf, md.sub_first); np.testing.assert_array_equal(se, md.sub_elem)
    np.testing.assert_array_equal(f3, np.asarray(md.sub_faces, np.int32).reshape(-1)); np.testing.assert_array_equal(c3, np.asarray(md.sub_core_faces, np.int32).reshape(-1))
    # d4est_solver_schwarz_iterate(p4est, geom, quad, factors, ghost, schwarz, vecs, r) on host vectors
    class Elliptic(ctypes.Structure):   # src/EllipticSystem/d4est_elliptic_data.h:6-37 as include/d4est_hip_compat.h mirrors it
        _fields_ = [("mpirank", ctypes.c_int), ("local_nodes", ctypes.c_int), ("num_of_fields", ctypes.c_int), ("field_types", ctypes.c_void_p),
                    ("Au", ctypes.c_void_p), ("u", ctypes.c_void_p), ("u0", ctypes.c_void_p), ("rhs", ctypes.c_void_p), ("user", ctypes.c_void_p)]
    key = ctypes.c_int(0)
    lib.d4est_hip_compat_bind_schwarz.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_double]
    lib.d4est_hip_compat_bind_schwarz(ctypes.byref(key), ctypes.c_void_p(sz.handle), 6, 1e-15, 1e-15)
    r = M.splitmix64_uniform(31, m.local_nodes) - 0.5
    u = np.zeros(m.local_nodes)
    vecs = Elliptic(local_nodes=m.local_nodes, num_of_fields=1, u=u.ctypes.data)
    lib.d4est_solver_schwarz_iterate.argtypes = [ctypes.c_void_p] * 8
    lib.d4est_solver_schwarz_iterate(ctypes.byref(key), None, None, None, None, None, ctypes.byref(vecs), vp(r))
    oracle.set_operator(m, J, rst, sides, 10.0, 0, threads=2)
    oracle.set_lhs_coefficient(None); oracle.set_lhs_element_blocks(None)
    u_ref, _, _ = oracle.schwarz_iterate(md, np.zeros(m.local_nodes), r, 6, 1e-15, 1e-15)
    assert np.abs(u - u_ref).max() <= 1e-9 * np.abs(u_ref).max()
    lib.d4est_hip_compat_bind_schwarz(ctypes.byref(key), None, 0, 0.0, 0.0)
    sz.destroy()
