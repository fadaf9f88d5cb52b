#!/bin/bash
# usage: tools_res.sh file.hip  -> per-kernel VGPR/AGPR/scratch/occupancy summary
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c "$1" -o /tmp/res_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
for line in sys.stdin:
    if "error" in line or "warning:" in line: print(line.rstrip())
    m=re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        cur=t.split(":",1)[1].strip(); print("\n"+cur, end=" | ")
    elif any(t.startswith(k) for k in ("VGPRs:","AGPRs:","ScratchSize","Occupancy","VGPRs Spill","SGPRs:","LDS Size")):
        print(t, end=" | ")
print()
'
rm -f /tmp/res_$$.o
