import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure): built on demand with oracle/Makefile."""
    from tests import oracle_lib
    return oracle_lib.load()


@pytest.fixture(scope="session")
def hiplib():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from disco4est_amd import build, capi
    if build.needs_build():
        build.build_library(verbose=False)
    return capi.load_library()


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible -- refusing to fall back to a CPU path")
    return torch.device("cuda:0")


# Modules whose GPU tests exercise the full operator through the smoothers, forests, shards or Schwarz: every test in them runs
# once per face path -- the direct kernel on conforming uniform plans up to deg_quad = 7 with the volume term in it at deg_quad = deg, the direct kernel for the faces only, and the two-phase kernels (D4EST_HIP_FACE_DIRECT=2 / 1 / 0 make tuning key 11
# default to that value for the plans the test creates; left alone, the library would pick the two-phase kernels on these small meshes).
_BOTH_FACE_PATHS = {"test_forest_gpu", "test_solver_gpu", "test_schwarz_gpu", "test_parallel_gpu", "test_multigrid_gpu"}


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.split(".")[-1] in _BOTH_FACE_PATHS:
        metafunc.fixturenames.append("_face_path")
        metafunc.parametrize("_face_path", ["direct+volume", "direct-faces-only", "two-phase"], indirect=True)


@pytest.fixture
def _face_path(request, monkeypatch):
    monkeypatch.setenv("D4EST_HIP_FACE_DIRECT", {"direct+volume": "2", "direct-faces-only": "1", "two-phase": "0"}[request.param])
    return request.param
