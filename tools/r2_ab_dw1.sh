#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2k; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -5 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for v in default dw1split; do
  if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
  for e in 4096 65536; do
    timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --hid 128 --dtype bf16 --envs $e > $O/ab_${v}_$e.json 2> $O/ab_${v}_$e.err || { tail -5 $O/ab_${v}_$e.err; exit 1; }
    python3 - $O/ab_${v}_$e.json $v $e <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "envs", sys.argv[3], "value %.0f"%d["value"], {n:k[n]["avg_ms"] for n in ("k_policy_bwd","k_policy_dw1","k_policy_fwd_train") if n in k})
PY
  done
done; done
