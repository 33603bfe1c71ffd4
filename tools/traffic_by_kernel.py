#!/usr/bin/env python3
"""Per-kernel HBM-side traffic of a profiled run: tools/traffic_by_kernel.py <prof dir> <steps in run>
Reads the FETCH_SIZE and WRITE_SIZE passes that tools/profile_resnet.sh / profile_bench.sh leave under <dir>/pmc_fetch and
<dir>/pmc_write plus the kernel trace under <dir>/trace, prints per kernel: launches per step, us per step, GB per step
(FETCH_SIZE x2 + WRITE_SIZE: the gfx950 correction of MI355X_MICROARCH.md) and the GB/s that implies."""
import csv
import glob
import os
import sys
from collections import defaultdict

out, nsteps = sys.argv[1], int(sys.argv[2])


def short(k):
    return k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:58]


byk = defaultdict(lambda: [0.0, 0.0, 0, 0.0])          # fetch bytes, write bytes, launches, us
for sub, ctr, mult, slot in (("pmc_fetch", "FETCH_SIZE", 2.0, 0), ("pmc_write", "WRITE_SIZE", 1.0, 1)):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True)[:1]:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") == ctr:
                    byk[short(r["Kernel_Name"])][slot] += float(r["Counter_Value"]) * mult * 1024.0
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)[:1]:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            k = short(r["Kernel_Name"])
            byk[k][2] += 1
            byk[k][3] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("%-58s %8s %10s %9s %9s %8s" % ("kernel", "n/step", "us/step", "rdGB/st", "wrGB/st", "GB/s"))
tot = [0.0, 0.0, 0.0]
for k, (fb, wb, n, us) in sorted(byk.items(), key=lambda kv: -(kv[1][0] + kv[1][1])):
    if fb + wb < 1e6 * nsteps:
        continue
    tot[0] += fb; tot[1] += wb; tot[2] += us
    print("%-58s %8.1f %10.1f %9.3f %9.3f %8.0f" % (k, n / nsteps, us / nsteps, fb / nsteps / 1e9, wb / nsteps / 1e9,
                                                 (fb + wb) / max(us, 1e-9) / 1e3))
print("%-58s %8s %10.1f %9.3f %9.3f" % ("total", "", tot[2] / nsteps, tot[0] / nsteps / 1e9, tot[1] / nsteps / 1e9))
