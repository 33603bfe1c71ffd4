#!/bin/bash
# Build an experimental variant of ONE kernel file with extra -D flags and link it against the other
# (already built) objects:  tools/build_variant.sh <name> <file.hip> [-DX=..]...
# Output: build/variants/lib_<name>.so (git-ignored, travels with gpurun).  On the GPU box, point
# CCST_HIP_LIB at it to A/B against the default build.
# The timing-experiment branches (ABL*, SPA_*, F23A_*, W4W_*, ABLW_*, tunable tile constants) are NOT in the shipped sources:
# tools/variants/<file>.patch (and common.h.patch) re-insert them into a scratch copy of ccst_amd/csrc before the variant is compiled.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build/variants
obj=build/variants/${name}.o
scratch=build/variants/src_${name}
rm -rf $scratch && mkdir -p $scratch/ccst_amd $scratch/include
cp -r ccst_amd/csrc $scratch/ccst_amd/csrc && cp include/*.h $scratch/include/
for pf in tools/variants/${src}.patch tools/variants/common.h.patch; do
    [ -f $pf ] && patch -s -p1 -d $scratch < $pf
done
/opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function "$@" -c $scratch/ccst_amd/csrc/$src -o $obj
others=$(ls ccst_amd/csrc/*.o | grep -v "/${src}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/lib_${name}.so $obj $others
echo build/variants/lib_${name}.so
