#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, mean of each counter per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
acc = collections.defaultdict(lambda: collections.defaultdict(list))
import os
files = []
for d in glob.glob(root + "/*/*/"):      # newest file per pass directory (gpurun_out accumulates older runs)
    c = sorted(glob.glob(d + "*counter_collection.csv"), key=os.path.getmtime)
    if c:
        files.append(c[-1])
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()):
        print("   %-28s n=%-5d mean=%.4g" % (c, len(v), sum(v) / len(v)))
