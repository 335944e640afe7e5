"""Sum the counter_collection.csv files of tools/prof_diag.sh per kernel class and counter."""
import csv, glob, os, re, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(float)); launches = defaultdict(set)
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k].add((f, r["Dispatch_Id"]))
for k in sorted(tot):
    if not k.startswith("k_"):
        continue
    print("== %s" % k)
    for c in sorted(tot[k]):
        print("   %-32s %12.4g" % (c, tot[k][c]))
