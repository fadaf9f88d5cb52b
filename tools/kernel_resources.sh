#!/bin/bash
# per-kernel register / scratch / LDS usage of one HIP source (compile-only): tools/kernel_resources.sh ppo_policy_fwd.hip [extra flags]
cd "$(dirname "$0")/../proximalpolicyoptimization.jl_amd/csrc"
f=$1; shift
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 --offload-device-only -Rpass-analysis=kernel-resource-usage "$@" -c $f -o /dev/null 2>&1 |
  python3 -c '
import re,sys
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r"Function Name: (\S+)",ln)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    for k in ("VGPRs","AGPRs","ScratchSize \[bytes/lane\]","Occupancy \[waves/SIMD\]","LDS Size \[bytes/block\]","SGPRs"):
        m=re.search(k+r": (\d+)",ln)
        if m and cur is not None: cur[k.split(" ")[0]]=int(m.group(1))
import subprocess
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()[:90]
    print("%-90s vgpr %3d agpr %3d scratch %4d occ %d lds %6d sgpr %3d"%(name,r.get("VGPRs",-1),r.get("AGPRs",-1),r.get("ScratchSize",-1),r.get("Occupancy",-1),r.get("LDS",-1),r.get("SGPRs",-1)))
'
