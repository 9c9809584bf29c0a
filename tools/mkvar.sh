#!/bin/bash
# A/B build: tools/mkvar.sh <name> <file.hip> [-DFLAG ...]  ->  build/var/<name>.so = the shipped objects with <file> recompiled
# with the extra flags (build/ is git-ignored but travels with gpurun; select with REPCONC_HIP_LIB=build/var/<name>.so)
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build/var
obj=$root/build/var/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -mllvm -amdgpu-mfma-vgpr-form \
  -Wall -Wno-unused-function "$@" -c $root/repconc_amd/csrc/$src -o $obj
objs=""
for o in $root/repconc_amd/lib/*.o; do
  if [ "$(basename $o)" == "${src%.hip}.o" ]; then objs="$objs $obj"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -ldl -o $root/build/var/$name.so
echo $root/build/var/$name.so
