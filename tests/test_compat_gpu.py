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
            "d4est_quadrature_compute_mass_matrix", "d4est_operators_compute_PT_mat_P", "d4est_operators_compute_prolong_matrix"} <= names and len(names) >= 30
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
                 "compute_PT_mat_P children 8", "compute_prolong_matrix children 1"):
        assert what in out.stdout


@pytest.mark.gpu
def test_smoother_shims_refuse_another_operator(gpu, hiplib, oracle, tmp_path):
    """cg_eigs / cheby_iterate_aux never call fcns->apply_lhs (the bound plan applies the operator); with the plan's callback registered
    (d4est_hip_compat_bind_operator) a caller that passes a DIFFERENT apply_lhs is aborted instead of silently served the wrong operator"""
    exe = _compile(tmp_path, oracle)
    out = subprocess.run([exe, "mismatch"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "NOT ABORTED" not in out.stdout
    assert "fcns->apply_lhs is not the operator registered" in out.stderr
