#!/bin/bash
# Development: recompile only the listed sources of libd4est_hip.so (objects of the others from the last full build) and relink.
# usage: tools/rebuild.sh d4est_hip_direct_mw.hip [more sources ...]   (extra hipcc flags: FLAGS="-D...")
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
B=$here/disco4est_amd/csrc/build
pids=""
for src in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed $FLAGS -x hip -c "$here/disco4est_amd/csrc/$src" -o "$B/$src.o" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/disco4est_amd/libd4est_hip.so" "$B"/*.o -ldl
g++ -O2 -std=c++17 -fPIC -Wall -shared "$here/disco4est_amd/csrc/d4est_hip_compat.cpp" -o "$here/disco4est_amd/libd4est_hip_compat.so" -L"$here/disco4est_amd" -ld4est_hip -Wl,-rpath,'$ORIGIN'
echo rebuilt
