"""Integer work is bit-exact against reference-held data: the p8est face / corner tables and d4est's face re-orientation tables that the
library (csrc/d4est_hip_topology.h, read by d4est_hip_faces.hip and d4est_hip_sides.cpp), the host-side mesh code (forest.py) and the
oracle (d4est_oracle_flux.c) work with are compared ENTRY BY ENTRY with the originals, extracted as integers from the reference's own
files by tests/golden/make_reference_tables.py (third_party/p4est-2.8.tar.gz: src/p8est_connectivity.c:29-63, :145-152;
src/dGMath/d4est_reference.c:3-12) -- not with each other."""
import ctypes
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "p8est_tables.json")))
IDS = {0: "p8est_face_corners", 1: "p8est_face_dual", 2: "p8est_face_permutations", 3: "p8est_face_permutation_sets",
       4: "p8est_face_permutation_refs", 5: "p8est_corner_faces", 10: "d4est_reference_p8est_FToF_code",
       11: "d4est_reference_p8est_code_to_perm", 12: "d4est_reference_p8est_perm_to_order"}


def _gold(name):
    t = GOLD[name]
    return np.asarray(t["values"], dtype=np.int32).reshape(t["shape"])


def _get(fn, tid):
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int, ctypes.c_void_p]
    n = fn(tid, None)
    assert n > 0, tid
    out = np.full(n, -99, dtype=np.int32)
    assert fn(tid, out.ctypes.data_as(ctypes.c_void_p)) == n
    return out


def test_library_tables_equal_the_reference_data(hiplib):
    for tid, name in IDS.items():
        np.testing.assert_array_equal(_get(hiplib.d4est_hip_topology_table, tid), _gold(name).reshape(-1), err_msg=name)
    assert hiplib.d4est_hip_topology_table(7, None) == -1


def test_oracle_tables_equal_the_reference_data(oracle):
    for tid in (4, 10, 11, 12):
        np.testing.assert_array_equal(_get(oracle.lib.oracle_topology_table, tid), _gold(IDS[tid]).reshape(-1), err_msg=IDS[tid])


def test_host_mesh_tables_equal_the_reference_data():
    from disco4est_amd import forest as F
    np.testing.assert_array_equal(F.FACE_CORNERS, _gold("p8est_face_corners"))
    np.testing.assert_array_equal(F._PERM_REFS, _gold("p8est_face_permutation_refs"))


def test_reference_keeps_copies_of_the_p8est_tables():
    """d4est_reference.c:3-12 are p8est_face_permutation_refs / _sets / _permutations under other names (a fact about the reference data)"""
    np.testing.assert_array_equal(_gold("d4est_reference_p8est_FToF_code"), _gold("p8est_face_permutation_refs"))
    np.testing.assert_array_equal(_gold("d4est_reference_p8est_code_to_perm"), _gold("p8est_face_permutation_sets"))
    np.testing.assert_array_equal(_gold("d4est_reference_p8est_perm_to_order"), _gold("p8est_face_permutations"))


def test_reorient_face_order_is_the_table_composition(hiplib, oracle):
    """d4est_reference_reorient_face_order (dGMath/d4est_reference.c:84-110, face_dim = 2) = perm_to_order[code_to_perm[FToF[f_m][f_p]][o]][i],
    formed here from the reference data, for every input -- the library's and the oracle's functions, bit for bit"""
    ftof, c2p, p2o = _gold("d4est_reference_p8est_FToF_code"), _gold("d4est_reference_p8est_code_to_perm"), _gold("d4est_reference_p8est_perm_to_order")
    oracle.lib.oracle_reorient_face_order.restype = ctypes.c_int
    oracle.lib.oracle_reorient_face_order.argtypes = [ctypes.c_int] * 4
    for f_m in range(6):
        for f_p in range(6):
            for o in range(4):
                for i in range(4):
                    want = int(p2o[c2p[ftof[f_m][f_p]][o]][i])
                    assert hiplib.d4est_hip_reorient_face_order(f_m, f_p, o, i) == want
                    assert oracle.lib.oracle_reorient_face_order(f_m, f_p, o, i) == want
