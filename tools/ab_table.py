"""SQ counter table of a tools/prof_ab.sh run:  python tools/ab_table.py TAG"""
import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("gpurun_out/ab_%s/sq*/*/*_counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    a = acc[k]
    if "SQ_WAVE_CYCLES" not in a: continue
    wc = a["SQ_WAVE_CYCLES"]
    print("%-34s launches %4d  VALU %.3e SALU %.3e LDS %.3e | wait_any %.2f wait_inst %.2f active %.2f | lane util %.2f | LDS active %.3e conflict %.3e busy %.3e" % (
        k, n[k]["SQ_WAVES"], a["SQ_INSTS_VALU"], a["SQ_INSTS_SALU"], a["SQ_INSTS_LDS"], a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc,
        a.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, a.get("SQ_ACTIVE_INST_VALU", 1) * 64), a.get("SQ_ACTIVE_INST_LDS", 0), a.get("SQ_LDS_BANK_CONFLICT", 0), a.get("SQ_BUSY_CYCLES", 0)))
