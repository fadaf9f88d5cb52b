#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2j; rm -rf $O; mkdir -p $O
for rep in 1 2; do for st in 0 1 2 3 5; do
    PPO_BWD_STAGGER=$st timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --hid 128 > $O/st_$st.json 2> $O/st_$st.err || { tail -5 $O/st_$st.err; exit 1; }
    python3 - $O/st_$st.json $st <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print("stagger", sys.argv[2], "value %.0f"%d["value"], "bwd %.4f ms frac %.4f"%(k["k_policy_bwd"]["avg_ms"], k["k_policy_bwd"]["frac"]), "fwd %.4f"%k["k_policy_fwd_train"]["avg_ms"])
PY
done; done
