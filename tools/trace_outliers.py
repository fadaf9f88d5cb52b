#!/usr/bin/env python3
"""rocprofv3 --kernel-trace directory -> for every hot kernel: average, the five slowest launches and where they sit in the
kernel's launch order (first launch of the process? first after another kernel family? random?).  Written because round
2's committed stats showed ONE k_policy_bwd_bf16 launch at 915 us against a 74 us average with nothing to explain it."""
import csv, glob, os, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], "*", "*kernel_trace.csv")), key=os.path.getmtime)
if not files:
    sys.exit("no kernel_trace.csv under " + sys.argv[1])
rows = list(csv.DictReader(open(files[-1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
by = {}
for i, r in enumerate(rows):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    prev = rows[i - 1]["Kernel_Name"].split("(")[0].replace("void ", "") if i else "-"
    gap = (int(r["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3 if i else 0.0
    by.setdefault(name, []).append((d, len(by.get(name, [])), i, prev, gap))
for name, v in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    if len(v) < 8 or "policy" not in name:
        continue
    avg = sum(x[0] for x in v) / len(v)
    med = sorted(x[0] for x in v)[len(v) // 2]
    print("%s: %d launches, avg %.1f us, median %.1f us" % (name, len(v), avg, med))
    for d, k, i, prev, gap in sorted(v, reverse=True)[:5]:
        print("    %.1f us = %.1fx median: launch #%d of this kernel (#%d of the process), previous kernel %s, idle gap before it %.1f us" % (d, d / med, k, i, prev, gap))
