#!/bin/bash
# Kernel experiments: build a variant of libd4est_hip.so in which ONE OR MORE sources (comma-separated) are recompiled with extra flags
# and the other objects are taken from the last full build (disco4est_amd/csrc/build).
# usage: tools/build_variant.sh NAME SOURCE[,SOURCE...] "-DFOO=1 ..."
# -> disco4est_amd/variants/libd4est_hip_NAME.so ; select it with D4EST_HIP_LIBRARY=<path>.
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
name=$1; srcs=${2//,/ }; flags=$3
mkdir -p "$here/disco4est_amd/variants"
for src in $srcs; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-pass-failed $flags -x hip -c "$here/disco4est_amd/csrc/$src" -o "$here/disco4est_amd/variants/$name.$src.o" &
done
wait
objs=""
for o in "$here"/disco4est_amd/csrc/build/*.o; do
  b=$(basename "$o" .o)
  if [ -f "$here/disco4est_amd/variants/$name.$b.o" ]; then objs="$objs $here/disco4est_amd/variants/$name.$b.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$here/disco4est_amd/variants/libd4est_hip_$name.so" $objs -ldl
rm -f "$here"/disco4est_amd/variants/$name.*.o
echo "$here/disco4est_amd/variants/libd4est_hip_$name.so"
