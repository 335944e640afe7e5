#!/usr/bin/env python3
"""Condenses a gpurun_out/prof_<tag>/ directory (tools/profile_r01.sh) into the files committed under profiles/:
  <tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary, verbatim
  <tag>_pmc_hbm.json          per-kernel FETCH_SIZE / WRITE_SIZE per launch, with the gfx950 corrections of
                              /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB;
                              FETCH_SIZE x2 for wide (16 B/lane) coalesced streaming reads; WRITE_SIZE exact
  traffic_latest.json         the dominant kernel's corrected HBM bytes per launch (read by bench.py)"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
workload = sys.argv[2] if len(sys.argv) > 2 else "cornell_lambert"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
dom = max((r for r in rows if r["Name"].startswith(("k_", "void k_"))), key=lambda r: float(r["TotalDurationNs"]))
domname = dom["Name"].replace("void ", "").split("<")[0].split("(")[0]


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").split("<")[0].split("(")[0]
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


fetch = per_kernel(glob.glob(os.path.join(src, "pmc_fetch", "*", "*_counter_collection.csv"))[0], "FETCH_SIZE")
write = per_kernel(glob.glob(os.path.join(src, "pmc_write", "*", "*_counter_collection.csv"))[0], "WRITE_SIZE")
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu --spp 128",
       "units": "bytes per launch; FETCH_SIZE/WRITE_SIZE counters are KiB; FETCH_SIZE doubled (gfx950, 16 B/lane coalesced streams)", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if not k.startswith("k_"):
        continue
    f, nf = fetch.get(k, [0, 1]); w, nw = write.get(k, [0, 1])
    fb = f / max(1, nf) * 1024 * 2; wb = w / max(1, nw) * 1024
    out["kernels"][k] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_bytes_per_launch_corrected": int(fb), "fetch_counter_KiB_per_launch_raw": f / max(1, nf),
                         "write_bytes_per_launch": int(wb), "hbm_bytes_per_launch": int(fb + wb)}
json.dump(out, open(os.path.join(dst, tag + "_pmc_hbm.json"), "w"), indent=1)
json.dump({"kernel": domname, "workload": workload, "hbm_bytes_per_launch": out["kernels"].get(domname, {}).get("hbm_bytes_per_launch"),
           "source": "profiles/%s_pmc_hbm.json" % tag}, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
for f in ("bench_trace.json",):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, tag + "_" + f))
print(json.dumps(out, indent=1))
print("dominant:", domname, dom["AverageNs"])
