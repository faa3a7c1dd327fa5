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
    "d4est_hip_direct_mw_hi.hip",
    "d4est_hip_solver.hip",
    "d4est_hip_transfer.hip",
    "d4est_hip_mgmatrix.hip",
    "d4est_hip_schwarz.hip",
    "d4est_hip_comm.hip",
]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-pass-failed"]


import re

INCLUDE_DIR = os.path.join(HERE, "..", "include")


def _deps(path, seen=None):
    """the file and every local header it includes, transitively (quoted #include lines, resolved next to the file or in include/)"""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        for base in (os.path.dirname(path), INCLUDE_DIR):
            cand = os.path.normpath(os.path.join(base, inc))
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(f) > t for f in sources)


def _obj(src):
    return os.path.join(CSRC, "build", src + ".o")


def stale_sources():
    """sources whose object is older than the source or any header it includes (a struct change in d4est_hip_internal.h makes every
    translation unit that sees the struct stale: objects of different layouts must never be linked together)"""
    return [src for src in SOURCES if _stale(_obj(src), _deps(os.path.join(CSRC, src)))]


def needs_build():
    if stale_sources():
        return True
    objs = [_obj(src) for src in SOURCES]
    return _stale(LIB, objs) or _stale(COMPAT_LIB, list(_deps(os.path.join(CSRC, COMPAT_SRC))) + [LIB])


def build_library(force=False, verbose=True, jobs=None, extra_flags=None):
    """Compile every stale HIP/C++ source for gfx950 (all of them with force) and link libd4est_hip.so."""
    todo = list(SOURCES) if force else stale_sources()
    if not todo and not needs_build():
        return LIB
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for src in todo:
        cmd = [HIPCC] + FLAGS + list(extra_flags or []) + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", _obj(src)]
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
            if os.path.exists(_obj(src)):
                os.remove(_obj(src))
            sys.stderr.write("hipcc failed on %s\n%s\n" % (src, out.decode(errors="replace")))
    if failed:
        raise RuntimeError("hipcc compilation failed")
    objs = [_obj(src) for src in SOURCES]
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
