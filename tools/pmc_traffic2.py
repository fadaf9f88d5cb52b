#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes (tools/final_profile_r2.sh) -> measured HBM bytes per launch of the backward kernel,
one entry per launch shape (bench.py looks its own shape up; nothing is scaled from another shape).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half of the bytes of a wide coalesced read
stream -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Both counters are in KiB.
usage: pmc_traffic2.py OUT.json  KEY=DIR [KEY=DIR ...]   (DIR holds the subdirectories fetch/ and write/)"""
import collections, csv, glob, json, os, sys


def passes(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub in ("fetch", "write"):
        files = sorted(glob.glob(os.path.join(d, sub, "*", "*counter_collection.csv")), key=os.path.getmtime)
        if not files:
            continue
        for r in csv.DictReader(open(files[-1])):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, v in acc.items():
        if "policy" not in k and "grad_reduce" not in k and "returns" not in k and "gae" not in k:
            continue
        fe = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"]))
        wr = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
        out[k] = {"fetch_bytes_corrected": 2 * fe * 1024, "write_bytes": wr * 1024, "hbm_bytes": 2 * fe * 1024 + wr * 1024,
                  "launches_sampled": len(v["FETCH_SIZE"])}
    return out


res = {"source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --t-steps 8 --epochs 1 at each launch shape; "
                 "FETCH_SIZE doubled (gfx950 wide-read correction)", "launch_shapes": {}}
for arg in sys.argv[2:]:
    key, d = arg.rsplit("=", 1)
    ks = passes(d)
    bwd = [v for k, v in ks.items() if k.startswith("k_policy_bwd<") or k.startswith("k_policy_bwd_x6") or k.startswith("k_policy_bwd_bf16")]
    if not bwd:      # three-product backward (deep policies): the pair
        pair = [v for k, v in ks.items() if k.startswith("k_policy_bwd_data") or k.startswith("k_policy_wgrad")]
        bwd = [{"hbm_bytes": sum(v["hbm_bytes"] for v in pair)}] if pair else []
    dw1 = [v for k, v in ks.items() if k.startswith("k_policy_dw1")]
    res["launch_shapes"][key] = {"k_policy_bwd_hbm_bytes": (bwd[0]["hbm_bytes"] if bwd else None),
                                 "k_policy_dw1_hbm_bytes": (dw1[0]["hbm_bytes"] if dw1 else None), "kernels": ks}
json.dump(res, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: v["k_policy_bwd_hbm_bytes"] for k, v in res["launch_shapes"].items()}, indent=1))
