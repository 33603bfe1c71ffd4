#!/bin/bash
# Build a variant of the WHOLE library with extra -D flags: tools/build_all_variant.sh <name> [-DX=..]...  -> build/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build/variants/$name
objs=""
for src in ccst_amd/csrc/*.hip ccst_amd/csrc/*.cpp; do
  obj=build/variants/$name/$(basename $src).o
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-function "$@" -c $src -o $obj &
  objs="$objs $obj"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_${name}.so $objs
echo build/variants/lib_${name}.so
