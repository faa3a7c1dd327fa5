"""A plain-C host (tests/c/abi_probe.c: gcc -std=c99, no Python / C++ / torch in the process) drives the C-ABI and reproduces the
outputs of the reference's d4est_quadrature_apply_stiffness_matrix recorded at survey time (tests/golden/survey_probe.json)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _compile(tmp_path):
    exe = str(tmp_path / "abi_probe")
    lib_dir = os.path.join(ROOT, "disco4est_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "abi_probe.c"), "-L" + lib_dir, "-ld4est_hip", "-lm",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_header_compiles_as_c99_and_links(hiplib, tmp_path):
    """CPU check: the header is plain C and every symbol the probe uses resolves against the built library"""
    assert os.path.exists(_compile(tmp_path))


@pytest.mark.gpu
def test_plain_c_host_reproduces_reference_probe(gpu, hiplib, tmp_path):
    exe = _compile(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.strip().endswith("ok")
