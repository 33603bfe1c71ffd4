#!/bin/bash
# rocm-smi power / engine clock sampled beside a long bench.py run (run on the GPU box): tools/power_sample.sh [steps] -> gpurun_out/power_clock.txt
STEPS=${1:-14000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/power_clock.txt
cd $ROOT
python bench.py --steps $STEPS --warmup 5 --min-seconds 0 --no-secondary --no-cpu-baseline > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
sleep 25          # import + set-up
: > $OUT
for i in $(seq 1 14); do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' >> $OUT
    echo >> $OUT
    sleep 1.5
done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/power_bench.json')); print('bench:', d['value'], 'images/s', d['ms_per_step'], 'ms/step')" >> $OUT
cat $OUT
