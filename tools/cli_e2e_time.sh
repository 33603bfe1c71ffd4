#!/bin/bash
# End-to-end rate of the stage-2 CLI on synthetic content (process start, model init, host-side image generation,
# H2D, the HIP path, D2H, JPEG encode and write), with and without saving:   tools/cli_e2e_time.sh [n_images]
# Measured on MI355X (16-core cgroup), N=1200 x 3 styles: 132 images/s with saving, ~140 without -- the single Python
# thread that produces the content batches bounds it, not the 549 images/s HIP path.  A thread pool for the JPEG encode
# and DataLoader worker processes were tried and measured SLOWER (123 and 72-119 images/s: GIL / IPC contention on this
# host), so the CLIs keep the reference's inline loader and writer.
cd "$(dirname "$0")/.."
N=${1:-1200}
W=$(mktemp -d)
export PYTHONPATH=$PWD
for dom in cartoon photo sketch; do
  python style_transfer/AdaIN/mean_std_computation_effcientMem.py --dataset pacs --target $dom --image_size 512 --batch 6 --synthetic 12 --random_weights --output $W/out >/dev/null 2>&1 || exit 1
done
mkdir -p $W/run && mv style_stats $W/run/
for extra in "" "--no_save"; do
  ( cd $W/run; rm -rf $W/out; s=$(date +%s.%N)
    python $OLDPWD/style_transfer/AdaIN/CCST_OverallStyleTransfer.py --dataset pacs --target art_painting --image_size 512 --batch 6 --synthetic $N --random_weights --output $W/out $extra >/dev/null 2>&1
    e=$(date +%s.%N); python3 -c "print('%s: %d images x 3 styles in %.1f s = %.0f images/s end to end' % ('$extra' or 'saving', $N, $e-$s, 3*$N/($e-$s)))" )
done
rm -rf $W
