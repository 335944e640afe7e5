import csv, glob, sys
from collections import defaultdict
tag = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob("gpurun_out/pmc_%s/p*/*/*_counter_collection.csv" % tag):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
        acc[k]["_ns_" + r["Counter_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc):
    a = acc[k]
    print("==", k, "launches", n[k].get("SQ_WAVES"))
    for c in sorted(a):
        if not c.startswith("_"): print("   %-26s %14.4g" % (c, a[c]))
    if "SQ_WAVE_CYCLES" in a:
        wc = a["SQ_WAVE_CYCLES"]
        print("   -> wait_any %.2f  wait_inst %.2f  active %.2f of wave cycles" % (a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc))
        print("   -> VALU insts/wave %.0f  SALU/wave %.0f  LDS/wave %.0f" % (a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_INSTS_SALU"] / a["SQ_WAVES"], a["SQ_INSTS_LDS"] / a["SQ_WAVES"]))
    if "SQ_THREAD_CYCLES_VALU" in a and a.get("SQ_ACTIVE_INST_VALU"):
        print("   -> lane utilisation %.2f" % (a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_ACTIVE_INST_VALU"] * 64)))
