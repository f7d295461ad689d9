"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name (dev tool)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][:60]
        agg[name][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[(name, r["Counter_Name"])] += 1
for name in sorted(agg, key=lambda n: -sum(agg[n].values())):
    vals = agg[name]
    print(name)
    for k in sorted(vals):
        print("   %-28s %.4g  (%d dispatches)" % (k, vals[k], calls[(name, k)]))
