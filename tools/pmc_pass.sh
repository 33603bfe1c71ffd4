#!/bin/bash
# tools/pmc_pass.sh <tag> "<counters>" <python args...>   one rocprofv3 PMC pass (kernel-trace only)
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$ROOT
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/pass -- python3 "$@" > $OUT/run.log 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
f = glob.glob(os.path.join(out, "pass", "**", "*counter_collection.csv"), recursive=True)
if not f:
    print("no counter file"); sys.exit(0)
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
with open(f[0]) as fh:
    for r in csv.DictReader(fh):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
print("%-60s" % "kernel", " ".join("%16s" % n[:16] for n in names))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].values()))[:12]:
    print("%-60s" % k, " ".join("%16.4g" % (v[n] / max(1, cnt[(k, n)])) for n in names))
PY
