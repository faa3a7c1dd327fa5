"""CPU tests of the product's host side: the C-ABI library loads, exports every symbol the
header declares, and its engine-owned 1-D tables agree with the oracle's (reference-style) tables."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "d4est_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(d4est_hip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(hiplib):
    from disco4est_amd import capi
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(hiplib, s), "libd4est_hip.so does not export %s" % s
    # the python binding covers the whole header, nothing more
    assert sorted(capi.SIGNATURES) == syms
    assert b"gfx950" in hiplib.d4est_hip_version()


def test_missing_library_fails_loudly(tmp_path):
    from disco4est_amd import capi
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load_library(str(tmp_path / "nope.so"))


@pytest.mark.parametrize("p", [1, 2, 3, 4, 7, 8, 11, 15, 19])
def test_tables_match_oracle(hiplib, oracle, p):
    from disco4est_amd import table
    n = p + 1
    x, w = oracle.lobatto(p)
    np.testing.assert_allclose(table("lobatto_nodes", p), x, atol=2e-15)
    np.testing.assert_allclose(table("lobatto_weights", p), w, rtol=1e-13)
    xg, wg = oracle.gauss(p)
    np.testing.assert_allclose(table("gauss_nodes", p), xg, atol=2e-15)
    np.testing.assert_allclose(table("gauss_weights", p), wg, rtol=1e-13)
    tol = 1e-11 * (1 + p) ** 2
    np.testing.assert_allclose(table("dij", p).reshape(n, n), oracle.dij(p), atol=tol)
    np.testing.assert_allclose(table("mij", p).reshape(n, n), oracle.mij(p), atol=1e-12)
    np.testing.assert_allclose(table("invmij", p).reshape(n, n), oracle.invmij(p), rtol=1e-9, atol=1e-9)
    for pq in (p, p + 1, p + 3):
        np.testing.assert_allclose(table("lobatto_to_gauss", p, pq).reshape(pq + 1, n), oracle.lobatto_to_gauss(p, pq), atol=1e-11)
    for ph in (p, p + 1, p + 2):
        np.testing.assert_allclose(table("p_prolong", p, ph).reshape(ph + 1, n), oracle.p_prolong(p, ph), atol=1e-11)
        np.testing.assert_allclose(table("hp_prolong", p, ph).reshape(2, ph + 1, n), oracle.hp_prolong(p, ph), atol=1e-11)
        np.testing.assert_allclose(table("p_restrict", p, ph).reshape(n, ph + 1), oracle.p_restrict(p, ph), atol=1e-9)
        np.testing.assert_allclose(table("hp_restrict", p, ph).reshape(2, n, ph + 1), oracle.hp_restrict(p, ph), atol=1e-9)


def test_mesh_layout():
    from disco4est_amd import mesh as M
    m = M.BrickMesh(2, 3)
    assert m.n_elements == 64 and m.local_nodes == 64 * 64
    # Morton order: first 8 elements are the 2x2x2 block at the origin, x fastest
    assert m.ijk[:8].tolist() == [[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0], [0, 0, 1], [1, 0, 1], [0, 1, 1], [1, 1, 1]]
    deg = np.arange(64) % 3 + 1
    mm = M.BrickMesh(2, deg, deg_quad_inc=1)
    assert mm.nodal_stride[1] == 8 and mm.quad_stride[1] == 27
    assert mm.local_nodes == int(((deg + 1) ** 3).sum())
    u1 = M.splitmix64_uniform(102321, 10)
    u2 = M.splitmix64_uniform(102321, 10)
    assert (u1 == u2).all() and (0 <= u1).all() and (u1 < 1).all() and len(set(u1)) == 10


def test_error_convention_aborts_like_d4est():
    """Invalid input aborts the process with a message (D4EST_ABORT = SC_ABORT semantics, src/Utilities/d4est_util.h:171): a NULL plan
    is caught before any HIP call, so this runs without a GPU."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from disco4est_amd import capi; lib = capi.load_library(); "
            "lib.d4est_hip_apply_stiffness_matrix(None, None, None)") % str(ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "[D4EST_HIP_ABORT]" in r.stderr and "apply_stiffness_matrix" in r.stderr


def test_comm_callback_exception_aborts():
    """An exception inside the exchange / allreduce hook must not be swallowed by ctypes (the C caller would continue with a stale
    ghost trace): the process aborts, the library's error convention (D4EST_ABORT, src/Utilities/d4est_util.h:171)."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent('''
        import sys; sys.path.insert(0, %r)
        from disco4est_amd import capi
        P = capi.Plan.__new__(capi.Plan)
        class L:
            def d4est_hip_plan_set_comm(self, *a): pass
        P.lib = L(); P.handle = None
        def boom(ph, a, b): raise RuntimeError("exchange failed")
        P.set_comm(boom, None)
        P._cb_ex(None, 0, None, None)
        print("survived")
    ''') % __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "survived" not in r.stdout
    assert "D4EST_HIP_ABORT" in r.stderr and "exchange failed" in r.stderr


def test_direct_kernarg_layout(hiplib, tmp_path):
    """The whole-operator kernel reads its late arguments from the kernel-argument segment through a struct that mirrors the
    parameter list (csrc/d4est_hip_direct.hip: DirectKernargs).  The compiler's own record of the argument offsets (.args in
    the code object's metadata) must agree with the struct for every instantiation."""
    import ctypes
    import subprocess
    from disco4est_amd import build
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-readelf")):
        pytest.skip("llvm-readelf not found")
    fn = hiplib.d4est_hipi_direct_kernarg_offsets
    fn.restype = ctypes.c_int
    buf = (ctypes.c_int * 32)()
    n = fn(buf, 32)
    want = [buf[i] for i in range(n)]
    assert n == 17 and want[0] == 0
    fb = str(tmp_path / "fatbin")
    subprocess.check_call([os.path.join(llvm, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fb, build.LIB])
    data = open(fb, "rb").read()
    starts = [m.start() for m in re.finditer(b"\x7fELF", data)]
    checked = 0
    for i, st in enumerate(starts):
        co = str(tmp_path / ("co%d" % i))
        open(co, "wb").write(data[st:(starts[i + 1] if i + 1 < len(starts) else len(data))])
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        if "faces_direct_kernel" not in notes:
            continue
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name or "faces_direct_kernel" not in name.group(1):
                continue
            args = blk.split(".args:")[1].split(".group_segment_fixed_size")[0]
            offs = [int(o) for o, kind in re.findall(r"\.offset:\s+(\d+)\s+\.size:\s+\d+\s+\.value_kind:\s+(\w+)", args)
                    if not kind.startswith("hidden")]
            assert offs == want, (name.group(1), offs, want)
            checked += 1
    assert checked >= 20
