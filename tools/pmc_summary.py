"""Sums rocprofv3 counter_collection.csv files per kernel: python3 tools/pmc_summary.py DIR [kernel-substring]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
needle = sys.argv[2] if len(sys.argv) > 2 else "search_kernel"
agg = collections.defaultdict(float)
n = collections.defaultdict(int)
for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if needle in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
for k in sorted(agg):
    print(f"{k:32s} launches {n[k]:4d} sum {agg[k]:.6g}")
