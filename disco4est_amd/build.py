"""Builds libd4est_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

The built .so is git-ignored but travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libd4est_hip.so")
# the reference's own entry points (include/d4est_hip_compat.h): plain C++ on top of the C-ABI, kept in its own library so that
# loading the engine does not put d4est's symbol names into the process
COMPAT_LIB = os.path.join(HERE, "libd4est_hip_compat.so")
COMPAT_SRC = "d4est_hip_compat.cpp"
SOURCES = [
    "d4est_hip_tables.cpp",
    "d4est_hip_sides.cpp",
    "d4est_hip_capi.hip",
    "d4est_hip_volume.hip",
    "d4est_hip_faces.hip",
    "d4est_hip_direct.hip",
    "d4est_hip_direct_mw.hip",
    "d4est_hip_solver.hip",
    "d4est_hip_transfer.hip",
    "d4est_hip_schwarz.hip",
    "d4est_hip_comm.hip",
]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


def _newest_source_mtime():
    m = 0.0
    for root, _, files in os.walk(CSRC):
        for f in files:
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    m = max(m, os.path.getmtime(os.path.join(HERE, "..", "include", "d4est_hip.h")))
    return m


def needs_build():
    newest = _newest_source_mtime()
    return any((not os.path.exists(l)) or os.path.getmtime(l) < newest for l in (LIB, COMPAT_LIB))


def build_library(force=False, verbose=True, jobs=None):
    """Compile every HIP/C++ source for gfx950 and link libd4est_hip.so."""
    if not force and not needs_build():
        return LIB
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors="replace"))
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed on %s\n%s\n" % (src, out.decode(errors="replace")))
    if failed:
        raise RuntimeError("hipcc compilation failed")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-Wall", "-shared", os.path.join(CSRC, COMPAT_SRC), "-o", COMPAT_LIB,
           "-L" + HERE, "-ld4est_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
    print(LIB)
