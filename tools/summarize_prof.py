#!/usr/bin/env python3
"""Condense rocprofv3 csv output (kernel trace + PMC passes) into a small text summary:
per kernel name: calls, total/avg duration; FETCH_SIZE / WRITE_SIZE per launch (FETCH doubled per the
gfx950 correction in MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pat):
    fs = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return fs[0] if fs else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return name.split("(")[0][:70]


trace = find("trace", "*kernel_trace.csv")
agg = defaultdict(lambda: [0, 0.0])
if trace:
    with open(trace) as f:
        for r in csv.DictReader(f):
            d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            k = agg[short(r["Kernel_Name"])]
            k[0] += 1
            k[1] += d
    tot = sum(v[1] for v in agg.values())
    print("== kernel trace (%s)" % os.path.relpath(trace, out))
    print("%-72s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %8d %12.1f %10.2f %6.2f" % (k, v[0], v[1], v[1] / v[0], 100 * v[1] / tot))
for sub, ctr, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != ctr:
                continue
            k = acc[short(r["Kernel_Name"])]
            k[0] += 1
            k[1] += float(r["Counter_Value"])
    print("== %s (KB per launch x%.0f correction => MB/launch)" % (ctr, mult))
    for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %8d launches  %12.3f MB/launch" % (k, v[0], v[1] / v[0] * mult * 1024 / 1e6))

# machine-readable HBM traffic per launch (bytes), FETCH_SIZE doubled per the gfx950 correction
import json
traffic = {}
launches = {}
for sub, ctr, mult in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    f = find(sub, "*counter_collection.csv")
    if not f:
        continue
    acc = defaultdict(lambda: [0, 0.0])
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != ctr:
                continue
            k = acc[short(r["Kernel_Name"])]
            k[0] += 1
            k[1] += float(r["Counter_Value"])
    for k, v in acc.items():
        traffic.setdefault(k, {})[ctr] = v[1] / v[0] * mult * 1024.0
        launches[k] = max(launches.get(k, 0), v[0])
if traffic:
    import datetime
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        with open(os.path.join(root, "ccst_amd", "csrc", ".build_stamp")) as fh:
            stamp = fh.read().strip()[:16]
    except OSError:
        stamp = None
    res = {k: dict(v, total_bytes_per_launch=sum(v.values()), launches=launches.get(k, 0)) for k, v in traffic.items()
           if "conv" in k or "adain" in k or "partials" in k or "chan_sums" in k}
    # provenance: bench.py reports these bytes only while the running build is the profiled one
    res["_source"] = {"build_stamp": stamp, "date": datetime.date.today().isoformat(), "profile": os.path.basename(out.rstrip("/")) + "/summary.txt",
                      "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950), per-launch average"}
    with open(os.path.join(out, "traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
