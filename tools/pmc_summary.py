#!/usr/bin/env python3
"""Average per-dispatch PMC counter values per kernel from a rocprofv3 --pmc csv run.
usage: tools/pmc_summary.py <dir-with-*_counter_collection.csv> [kernel-substring]"""
import collections
import csv
import glob
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if sub not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x / cnt[(k, c)]:16.1f}   (n={cnt[(k, c)]})")
