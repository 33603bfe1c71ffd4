#!/bin/bash
# Prices the 2-D form F(2,3) y x F(4,3) x on the hardware without building it: variants of the shipped F(4,3) kernel with the
# 2-D form's COST PROFILE (tools/variants/conv3x3_f43.hip.patch, F43A bits 256 / 512 / 1024: two of the three products = 36 instead of 54
# MFMAs per chunk and wave; every transform item computed twice = its 16 transformed rows per 10 raw ones, pessimistic; a fourth weight
# load per three k-steps = 24 instead of 18 slabs per chunk).  Results are wrong, the time is what is measured: the layers the form would
# replace (rows 1, 3, 4, 6, 7, 9 of tools/f43_layers.py) on each variant.
#   on the build box:  for v in 0 256 512 1024 1792; do tools/build_variant.sh f43a_$v conv3x3_f43.hip -DF43A=$v; done
#   on the GPU box:    tools/wino2d_price.sh > gpurun_out/wino2d_price.txt
cd "$(dirname "$0")/.."
for v in 0 256 512 1024 1792 0; do
    echo "== F43A=$v"
    CCST_HIP_LIB=build/variants/lib_f43a_$v.so F43_LAYER=1,3,4,6,7,9 python tools/f43_layers.py 20 2>&1 | grep -v amdgpu.ids | sed 's/split .* | f43/f43/'
done
