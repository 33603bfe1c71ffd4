#!/bin/bash
# Run on the GPU box (via gpurun): the ResNet50 train step (bench_resnet.py) under rocprofv3 -- kernel-trace stats, then
# separate PMC passes for FETCH_SIZE and WRITE_SIZE -> gpurun_out/prof_<tag>/{summary.txt,traffic_resnet.json}
# Usage: tools/profile_resnet.sh <tag> [arch] [batch]
set -u
TAG=${1:-r2}; ARCH=${2:-resnet50}; BATCH=${3:-64}
STEPS=5; WARM=2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$ROOT/bench_resnet.py --steps $STEPS --warmup $WARM --arch $ARCH --batch $BATCH"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/bench_write.log 2>&1
cd $ROOT
python3 tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
python3 - "$OUT" "$ARCH" "$BATCH" "$((STEPS + WARM))" "$ROOT" <<'PY'
import csv, datetime, glob, json, os, sys
from collections import defaultdict
out, arch, batch, nsteps, root = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
# Per kernel: dispatches of the steady-state steps only.  A kernel that runs k times per step shows k * nsteps dispatches (+ extras in
# step 0: the one-off weight packs, first-touch fills); keep the LAST k * (nsteps - 1) of them, k = count // nsteps, and drop kernels
# that do not run every step (count < nsteps).  FETCH_SIZE x2 per the gfx950 correction of MI355X_MICROARCH.md.
tot = {}
for sub, ctr, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    fs = glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        continue
    per = defaultdict(list)
    with open(fs[0]) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") == ctr:
                per[r["Kernel_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
    s = 0.0
    for name, rows in per.items():
        k = len(rows) // nsteps
        if k == 0 or "pack_weight" in name:
            continue
        rows.sort()
        s += sum(v for _, v in rows[-k * (nsteps - 1):])
    tot[ctr] = s * mult * 1024.0 / (nsteps - 1)          # counters are in KB
if tot:
    try:
        stamp = open(os.path.join(root, "ccst_amd", "csrc", ".build_stamp")).read().strip()[:16]
    except OSError:
        stamp = None
    res = {"%s_b%d" % (arch, batch): round(sum(tot.values())), "detail_bytes_per_step": {k: round(v) for k, v in tot.items()},
           "steps_in_run": nsteps, "build_stamp": stamp, "date": datetime.date.today().isoformat(),
           "profile": os.path.basename(out.rstrip("/")) + "/summary.txt",
           "note": "FETCH_SIZE x2 + WRITE_SIZE over the kernels of the steady-state steps (step 0 and the one-off weight packs excluded) / (steps - 1)"}
    json.dump(res, open(os.path.join(out, "traffic_resnet.json"), "w"), indent=1)
    print(json.dumps(res))
PY
head -40 $OUT/summary.txt
