"""disco4est_amd -- MI355X-native matrix-free DG operator-apply engine (d4est hot path).

The product is the C-ABI HIP library ``libd4est_hip.so`` (sources in ``csrc/``,
ABI in ``include/d4est_hip.h``).  This package is the thin host-side binding used
by the tests and ``bench.py``: ctypes signatures, a ``Plan`` wrapper that takes
torch CUDA tensors (torch is used for device memory and streams only) and the
synthetic-mesh helpers.  There is NO CPU fallback: if the library is missing or
fails to load, importing :mod:`disco4est_amd.capi` raises.
"""
from .capi import Plan, Transfer, load_library, table, TABLE  # noqa: F401
from . import mesh  # noqa: F401
from .schwarz import Schwarz, SchwarzMetadata  # noqa: F401

__all__ = ["Plan", "Transfer", "Schwarz", "SchwarzMetadata", "load_library", "table", "TABLE", "mesh"]
