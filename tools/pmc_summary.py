#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter CSVs per kernel name.  usage: pmc_summary.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_mfma"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean {sum(v) / len(v):16.1f}  n={len(v)}")
