"""Aggregate `python -m ccst_amd.bench_resnet --layers` output by (role, shape): python tools/layer_agg.py <file> [top]"""
import collections
import re
import sys

rows = []
for l in open(sys.argv[1]):
    m = re.match(r'(\S+)\s+(n\d+ .*?)\s+([\d.]+) us\s+([\d.]+) TF', l)
    if m:
        rows.append((m.group(1), m.group(2).strip(), float(m.group(3)), float(m.group(4))))
agg = collections.OrderedDict()
for k, d, us, tf in rows:
    a = agg.setdefault((k.split(':')[0], d), [0, 0.0, tf, k])
    a[0] += 1
    a[1] += us
top = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for (k, d), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-46s %-58s x%2d %8.1f us %6.1f TF" % (a[3], d, a[0], a[1], a[2]))
print("total us", sum(a[1] for a in agg.values()))
