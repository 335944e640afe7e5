"""Per-kernel sums of whatever PMC counters the passes under DIR collected (DIR/p*/**/_counter_collection.csv)."""
import csv, glob, sys
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int)); ns = defaultdict(lambda: defaultdict(float))
for f in glob.glob(d + "/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if not k.startswith("k_"): continue
        c = r["Counter_Name"]
        acc[k][c] += float(r["Counter_Value"]); n[k][c] += 1
        ns[k][c] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k in sorted(acc):
    any_c = next(iter(acc[k]))
    print("==", k, "launches", n[k][any_c], "kernel ms %.2f" % (ns[k][any_c] / 1e6))
    for c in sorted(acc[k]):
        print("   %-40s %14.5g   per launch %12.5g" % (c, acc[k][c], acc[k][c] / max(1, n[k][c])))
