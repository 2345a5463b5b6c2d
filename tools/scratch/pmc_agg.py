#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches)."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void hsk::", "").replace("hsk::", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = (f, row["Dispatch_Id"])
        if key not in seen:
            seen.add(key)
    # calls: count dispatches once per file (first counter)
names = sorted(agg)
counters = sorted({c for k in agg for c in agg[k]})
for k in names:
    print(k)
    for c in counters:
        if c in agg[k]:
            print("   %-24s %18.0f" % (c, agg[k][c]))
