#!/bin/bash
# tools/profile_cmd.sh <tag> <python args...>  -> kernel-trace stats for an arbitrary python command
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/run.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
head -45 $OUT/summary.txt
