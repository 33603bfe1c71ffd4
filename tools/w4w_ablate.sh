#!/bin/bash
# Timing ablations of the wide F(4x4) kernel (conv3x3_wino4w.hip) on one layer of tools/wino_layers.py (WINO_LAYER, default 4 =
# 128x128 256->256): builds one variant library per -D set (CPU side, before gpurun), then on the GPU box runs the layer with each.
#   tools/w4w_ablate.sh build        (here)      tools/w4w_ablate.sh run [layers]   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
declare -A V=(
  [full]=""
  [nomfma]="-DABLW_NO_MFMA"
  [noxform]="-DABLW_NO_XFORM"
  [nob]="-DABLW_NO_B"
  [nohalo]="-DABLW_NO_HALO"
  [nohload]="-DABLW_NO_HALO_LOAD"
  [nohstore]="-DABLW_NO_HALO_STORE"
  [nolds]="-DABLW_NO_LDS_READ -DABLW_NO_XFORM"
  [nobar]="-DABLW_NO_BARRIER"
  [noepi]="-DABLW_NO_EPILOGUE"
  [mfmaonly]="-DABLW_NO_XFORM -DABLW_NO_B -DABLW_NO_HALO -DABLW_NO_LDS_READ -DABLW_NO_BARRIER"
  [loopmfma]="-DABLW_NO_XFORM -DABLW_NO_B -DABLW_NO_HALO -DABLW_NO_LDS_READ -DABLW_NO_BARRIER -DABLW_NO_EPILOGUE"
)
if [ "$1" = build ]; then
  for k in "${!V[@]}"; do tools/build_variant.sh w4w_$k conv3x3_wino4w.hip ${V[$k]} >/dev/null & done
  wait
  ls build/variants/lib_w4w_*.so
else
  L=${2:-4}
  for k in full nomfma noxform nob nohalo nohload nohstore nolds nobar noepi mfmaonly loopmfma; do
    echo "== $k"
    CCST_HIP_LIB=$PWD/build/variants/lib_w4w_$k.so WINO_LAYER=$L python tools/wino_layers.py 20 2>&1 | grep -v amdgpu.ids | cut -c1-28,58-100
  done
fi
