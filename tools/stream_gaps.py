#!/usr/bin/env python3
"""Busy time per HIP stream (queue) of a profiled run: tools/stream_gaps.py <kernel_trace.csv> <steps in run>
-> per queue: launches/step, busy ms/step (sum of kernel durations), span ms/step (first start .. last end), so idle = span - busy."""
import csv
import sys
from collections import defaultdict

f, nsteps = sys.argv[1], int(sys.argv[2])
q = defaultdict(lambda: [0, 0.0, None, None])
with open(f) as fh:
    for r in csv.DictReader(fh):
        k = r.get("Queue_Id", "?")
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        a = q[k]
        a[0] += 1
        a[1] += (e - s) / 1e6
        a[2] = s if a[2] is None else min(a[2], s)
        a[3] = e if a[3] is None else max(a[3], e)
for k, (n, busy, s, e) in sorted(q.items(), key=lambda kv: -kv[1][1]):
    print("queue %-6s launches/step %7.1f  busy %7.2f ms/step  span %8.2f ms/step" % (k, n / nsteps, busy / nsteps, (e - s) / 1e6 / nsteps))
