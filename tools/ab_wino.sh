#!/bin/bash
# A/B the AdaIN bench (Winograd path) over build/variants/lib_*.so given as arguments (default build first).
cd "$(dirname "$0")/.."
export CCST_CONV_WINO=${CCST_CONV_WINO:-4}
line() { python bench.py --steps 20 --warmup 3 --no-secondary --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k.replace('conv3x3_wino_kernel','wino').replace('conv_igemm_kernel','igemm'):(round(v['avg_us'],1), v['tflops']) for k,v in d['kernels'].items() if 'wino' in k})"; }
echo "default:"; line
for v in "$@"; do echo "$v:"; CCST_HIP_LIB=$PWD/build/variants/lib_$v.so line; done
