#!/usr/bin/env python3
"""Copy the rocprofv3 summaries of a profiling run from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names and
point the traffic files at them:   python tools/publish_profiles.py r03 prof_r3a prof_r3a_resnet50 [--r18=prof_r3a_resnet18] [r3a_pmc_sq1.txt r3a_pmc_sq2.txt]
  profiles/<round>_adain_bench_summary.txt, <round>_adain_kernel_stats.csv, <round>_traffic.json  + profiles/traffic.json
  profiles/<round>_resnet50_train_summary.txt, <round>_resnet50_kernel_stats.csv                  + profiles/traffic_resnet.json
  profiles/<round>_pmc_sq.txt (the SQ counter passes, if given)"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, adain, resnet = sys.argv[1], sys.argv[2], sys.argv[3]
r18 = [a.split("=", 1)[1] for a in sys.argv[4:] if a.startswith("--r18=")]
pmc = [a for a in sys.argv[4:] if not a.startswith("--r18=")]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def stats_csv(d):
    f = glob.glob(os.path.join(G, d, "trace", "**", "*kernel_stats.csv"), recursive=True)
    return f[0] if f else None


shutil.copy(os.path.join(G, adain, "summary.txt"), os.path.join(P, "%s_adain_bench_summary.txt" % rnd))
if stats_csv(adain):
    shutil.copy(stats_csv(adain), os.path.join(P, "%s_adain_kernel_stats.csv" % rnd))
tj = json.load(open(os.path.join(G, adain, "traffic.json")))
tj["_source"]["profile"] = "profiles/%s_adain_bench_summary.txt" % rnd
for name in ("%s_traffic.json" % rnd, "traffic.json"):
    json.dump(tj, open(os.path.join(P, name), "w"), indent=1)
shutil.copy(os.path.join(G, resnet, "summary.txt"), os.path.join(P, "%s_resnet50_train_summary.txt" % rnd))
if stats_csv(resnet):
    shutil.copy(stats_csv(resnet), os.path.join(P, "%s_resnet50_kernel_stats.csv" % rnd))
rj = json.load(open(os.path.join(G, resnet, "traffic_resnet.json")))
rj["profile"] = "profiles/%s_resnet50_train_summary.txt" % rnd
if r18:         # the ResNet18 step (config 5) of the same build: its bytes per step beside ResNet50's, its summary under its own name
    shutil.copy(os.path.join(G, r18[0], "summary.txt"), os.path.join(P, "%s_resnet18_train_summary.txt" % rnd))
    r18j = json.load(open(os.path.join(G, r18[0], "traffic_resnet.json")))
    assert r18j.get("build_stamp") == rj.get("build_stamp"), "ResNet18 and ResNet50 profiles are of different builds"
    for k, v in r18j.items():
        if k.startswith("resnet18"):
            rj[k] = v
    rj["detail_bytes_per_step_resnet18"] = r18j.get("detail_bytes_per_step")
    rj["profile_resnet18"] = "profiles/%s_resnet18_train_summary.txt" % rnd
json.dump(rj, open(os.path.join(P, "traffic_resnet.json"), "w"), indent=1)
if pmc:
    with open(os.path.join(P, "%s_pmc_sq.txt" % rnd), "w") as out:
        out.write("rocprofv3 --kernel-trace --pmc <counters> -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary   "
                  "(tools/pmc_pass.sh; per-launch averages; build %s)\n\n" % tj["_source"].get("build_stamp"))
        for name in pmc:                     # files under gpurun_out/ written by tools/pmc_pass.sh
            out.write(open(os.path.join(G, name)).read() + "\n")
print("published", rnd)
