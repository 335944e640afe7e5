#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (tools/profile_r04.sh TAG CONFIG) into the files committed under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  <tag>_bench_trace.json   the bench line printed under the kernel-trace pass
  <tag>_pmc_hbm.json       per-kernel FETCH_SIZE / WRITE_SIZE per launch and per unit, with the gfx950 corrections of
                           /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; FETCH_SIZE x2 for wide
                           (16 B/lane) coalesced streaming reads; WRITE_SIZE exact
  <tag>_pmc_sq.txt         SQ counter table per kernel (instruction mix, stalls, lane utilisation)
  traffic_r04.json         per scene and kernel class: HBM bytes per unit, wave64 VALU instructions per unit, lane utilisation
                           (read by bench.py: roofline.traffic, roofline.valu)
usage: python tools/summarize_r04.py TAG CONFIG"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
cfg = int(sys.argv[2])
SCENE = {1: "cornell_lambert", 2: "cornell", 3: "bunny", 4: "bunny"}[cfg]
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(REPO, "gpurun_out", "prof_" + tag)
dst = os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)


def cls_of(kernel):
    k = kernel.replace("void ", "").split("<")[0].split("(")[0]
    for base in ("k_extend", "k_shadow", "k_shade"):            # k_extend_sort / k_extend_persist ... count as their class
        if k.startswith(base + "_") or k == base:
            return base
    return k


def bench_line(name):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        return None
    for line in open(p):
        line = line.strip()
        if line.startswith("{") and '"metric"' in line:
            return json.loads(line)
    return None


def units_of(b):
    """rays per run of a bench line: (closest-hit rays, shadow rays)"""
    wp = b["roofline"]["whole_path"]
    w = b["config"]["workload"]
    import re
    m = re.search(r"(\d+)x(\d+), (\d+) spp", w)
    samples = int(m.group(1)) * int(m.group(2)) * int(m.group(3)) * b["steps"]
    return wp["segments_per_sample"] * samples, wp["shadow_rays_per_sample"] * samples


stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
bt = bench_line("bench_trace.json")
if bt:
    json.dump(bt, open(os.path.join(dst, tag + "_bench_trace.json"), "w"), indent=1)


def per_kernel(pattern, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(src, pattern, "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = cls_of(r["Kernel_Name"])
            acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
    return acc


tpath = os.path.join(dst, "traffic_r04.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
entry = traffic.setdefault(SCENE, {})

fetch = per_kernel("pmc_fetch", "FETCH_SIZE"); write = per_kernel("pmc_write", "WRITE_SIZE")
bf = bench_line("bench_fetch.json")
if fetch and write and bf:
    seg, sh = units_of(bf)
    unit = {"k_extend": seg, "k_shade": seg, "k_shadow": sh}
    out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --config %d --configs= --no-cpu --no-exclusive --spp <reduced>" % cfg,
           "workload": bf["config"]["workload"],
           "units": "bytes; FETCH_SIZE/WRITE_SIZE counters are KiB; FETCH_SIZE doubled (gfx950, 16 B/lane coalesced streams); per unit = per closest-hit segment (k_extend, k_shade) / per shadow ray (k_shadow)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, nf = fetch.get(k, [0, 1]); w, nw = write.get(k, [0, 1])
        fb = f * 1024 * 2; wb = w * 1024
        rec = {"launches_fetch_pass": nf, "launches_write_pass": nw, "fetch_bytes_per_launch_corrected": int(fb / max(1, nf)), "fetch_counter_KiB_per_launch_raw": f / max(1, nf),
               "write_bytes_per_launch": int(wb / max(1, nw)), "hbm_bytes_per_launch": int(fb / max(1, nf) + wb / max(1, nw))}
        if k in unit and unit[k] > 0:
            rec["hbm_bytes_per_unit"] = (fb + wb) / unit[k]
            entry.setdefault(k, {})["hbm_bytes_per_unit"] = round((fb + wb) / unit[k], 2)
            entry["_hbm_source"] = "profiles/%s_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, FETCH_SIZE x2 per the gfx950 note of MI355X_MICROARCH.md; %s)" % (tag, bf["config"]["workload"])
        out["kernels"][k] = rec
    json.dump(out, open(os.path.join(dst, tag + "_pmc_hbm.json"), "w"), indent=1)

# SQ table
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int)); full = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(src, "sq*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        kfull = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not kfull.startswith("k_"):
            continue
        full[kfull][r["Counter_Name"]] += float(r["Counter_Value"]); n[kfull][r["Counter_Name"]] += 1
        acc[cls_of(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
bs = bench_line("bench_sq1.json")
lines = []
if full:
    if bs:
        lines.append("# " + bs["config"]["workload"])
    for k in sorted(full):
        a = full[k]
        lines.append("== %s launches %s" % (k, n[k].get("SQ_WAVES") or n[k].get("SQ_BUSY_CYCLES")))
        for c in sorted(a):
            lines.append("   %-26s %14.4g" % (c, a[c]))
        if "SQ_WAVE_CYCLES" in a:
            wc = a["SQ_WAVE_CYCLES"]
            lines.append("   -> wait_any %.2f  wait_inst %.2f  active %.2f of wave cycles" % (a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc))
            lines.append("   -> VALU insts/wave %.0f  SALU/wave %.0f  LDS/wave %.0f" % (a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_INSTS_SALU"] / a["SQ_WAVES"], a["SQ_INSTS_LDS"] / a["SQ_WAVES"]))
        if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
            lines.append("   -> lane utilisation %.2f" % (a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_ACTIVE_INST_VALU"] * 64)))
    if bs:
        seg, sh = units_of(bs)
        unit = {"k_extend": seg, "k_shade": seg, "k_shadow": sh}
        for k in ("k_extend", "k_shade", "k_shadow"):
            a = acc.get(k)
            if a and a.get("SQ_INSTS_VALU") and unit[k] > 0:
                e = entry.setdefault(k, {})
                e["valu_insts_per_unit"] = round(a["SQ_INSTS_VALU"] / unit[k], 3)        # wave64 instructions per ray (x64 = per 64-ray wave pass)
                if a.get("SQ_THREAD_CYCLES_VALU") and a.get("SQ_ACTIVE_INST_VALU"):
                    e["lane_utilisation"] = round(a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_ACTIVE_INST_VALU"] * 64), 3)
                entry["_valu_source"] = "profiles/%s_pmc_sq.txt (rocprofv3 --pmc SQ_INSTS_VALU ...; %s)" % (tag, bs["config"]["workload"])
                lines.append("   %s: %.1f wave64 VALU instructions per 64-unit wave pass" % (k, 64 * a["SQ_INSTS_VALU"] / unit[k]))
    open(os.path.join(dst, tag + "_pmc_sq.txt"), "w").write("\n".join(lines) + "\n")
# [round 4] VALUBusy per kernel: SQ_ACTIVE_INST_VALU (in units of 4 cycles of one SIMD) x 4 / (1024 SIMDs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter
# sums the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back").  rocprofv3 profiles every dispatch on its own, so this is the kernel with the GPU to itself.
vb = defaultdict(lambda: defaultdict(float)); vbfull = defaultdict(lambda: defaultdict(float))
for f in glob.glob(os.path.join(src, "vb", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        kfull = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not kfull.startswith("k_"):
            continue
        vb[cls_of(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"]); vbfull[kfull][r["Counter_Name"]] += float(r["Counter_Value"])
        vb[cls_of(r["Kernel_Name"])]["_ns_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
vlines = []
for k in sorted(vbfull):
    a = vbfull[k]
    if a.get("GRBM_GUI_ACTIVE") and a.get("SQ_ACTIVE_INST_VALU"):
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        vlines.append("%-46s VALUBusy %.3f   (SQ_ACTIVE_INST_VALU %.4g x 4 / (1024 SIMDs x %.4g kernel cycles); %.4g wave64 VALU instructions)" % (k, a["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), a["SQ_ACTIVE_INST_VALU"], cyc, a.get("SQ_INSTS_VALU", 0)))
for k in ("k_extend", "k_shade", "k_shadow"):
    a = vb.get(k)
    if a and a.get("GRBM_GUI_ACTIVE") and a.get("SQ_ACTIVE_INST_VALU"):
        cyc = a["GRBM_GUI_ACTIVE"] / 8.0
        e = entry.setdefault(k, {})
        e["valu_busy"] = round(a["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc), 3)
        ns = a.get("_ns_GRBM_GUI_ACTIVE", 0)
        if ns > 0:
            e["clock_ghz"] = round(cyc / ns, 3)
        entry["_valu_busy_source"] = "profiles/%s_pmc_valubusy.txt (rocprofv3 --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE; VALUBusy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs))" % tag
if vlines:
    bvb = bench_line("bench_vb.json")
    open(os.path.join(dst, tag + "_pmc_valubusy.txt"), "w").write(("# " + bvb["config"]["workload"] + "\n" if bvb else "") + "\n".join(vlines) + "\n")
json.dump(traffic, open(tpath, "w"), indent=1)
print("\n".join(vlines))
print("\n".join(lines[-40:]))
print(json.dumps(entry, indent=1))
