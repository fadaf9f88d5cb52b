#!/usr/bin/env python3
"""Turn the FETCH_SIZE / WRITE_SIZE passes of tools/profile_pmc.sh into per-launch HBM bytes for the dominant kernels.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half of the bytes of a wide coalesced read
stream -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Both counters are in KiB."""
import csv, glob, json, sys, collections
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
        if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, v in acc.items():
    if "policy" not in k:
        continue
    fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
    wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    out[k] = {"fetch_bytes_corrected": 2 * fe * 1024, "write_bytes": wr * 1024, "hbm_bytes": 2 * fe * 1024 + wr * 1024,
              "launches_sampled": len(v["FETCH_SIZE"])}
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --t-steps 8 --epochs 1, 4096 states per launch",
           "kernels": out}, open(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/final/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
