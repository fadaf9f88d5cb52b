#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2l; rm -rf $O; mkdir -p $O
for rep in 1 2; do for v in default w3p4 w3p8 w2p8; do
  if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
  timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --hid 128 > $O/ab_$v.json 2> $O/ab_$v.err || { tail -5 $O/ab_$v.err; exit 1; }
  python3 - $O/ab_$v.json $v <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "value %.0f"%d["value"], {n:(k[n]["avg_ms"],k[n].get("frac")) for n in ("k_policy_fwd_train","k_rollout_persistent","k_policy_bwd") if n in k})
PY
done; done
