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
