#!/bin/bash
# Development: recompile the STALE sources of libd4est_hip.so (object older than the source or any header it includes) and relink.
# usage: tools/rebuild.sh   (all flags and the dependency scan live in disco4est_amd/build.py)
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
cd "$here" && python -c "from disco4est_amd import build as b; print('stale:', b.stale_sources()); b.build_library(verbose=False); print('rebuilt')"
