#!/bin/bash
# Timing ablations of the F(2,3) half-piece kernel (conv3x3_f23.hip) on layers of tools/f23_layers.py (F23_LAYER, default 3 = 128x128
# 256->256): tools/f23_ablate.sh build (here: one variant library per -D set), tools/f23_ablate.sh run [layers] (on the GPU box).
set -e
cd "$(dirname "$0")/.."
declare -A V=(
  [full]=""
  [vm2]="-DF23A_VM_EXTRA=2"
  [vm4]="-DF23A_VM_EXTRA=4"
  [nolgkm]="-DF23A_NO_LGKM"
  [nov]="-DF23A_NO_STOREV"
  [noread]="-DF23A_NO_READ"
  [mfmaonly]="-DF23A_NO_STOREV -DF23A_NO_LOADB -DF23A_NO_LOADV -DF23A_NO_READ"
)
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do tools/build_variant.sh f23_$k conv3x3_f23.hip ${V[$k]} >/dev/null & done
  wait
  ls build/variants/lib_f23_*.so
else
  L=${2:-3}
  for k in full vm2 vm4 nolgkm nov noread mfmaonly; do
    echo "== $k"
    CCST_HIP_LIB=$PWD/build/variants/lib_f23_$k.so F23_LAYER=$L python tools/f23_layers.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-24,80-160
  done
fi
