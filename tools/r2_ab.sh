#!/bin/bash
# generic A/B on the GPU box: tools/r2_ab.sh OUT "<variants>" "<hid list>" ["<pytest -k filter>"] ["<extra bench args>"] [variant to test too]
# variant "default" = the shipped library, anything else = proximalpolicyoptimization.jl_amd/libppo_hip_<variant>.so; two alternating rounds
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
K="${4:-gradient_vs_f64 or q32 or any_hidden or ragged}"
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "$K" > $O/gpu_tests.log 2>&1; rc=$?; tail -5 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
if [ -n "$6" ]; then   # the same tests on a variant library
  PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$6.so timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "$K" > $O/gpu_tests_$6.log 2>&1; rc=$?; tail -5 $O/gpu_tests_$6.log; echo "tests($6) rc=$rc"
  [ $rc -eq 0 ] || exit $rc
fi
for rep in 1 2; do for v in $2; do
  if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
  for h in $3; do
    timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --hid $h $5 > $O/ab_${v}_$h.json 2> $O/ab_${v}_$h.err || { tail -5 $O/ab_${v}_$h.err; exit 1; }
    python3 - $O/ab_${v}_$h.json $v $h <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "hid", sys.argv[3], "value %.0f"%d["value"], " ".join("%s %.4f"%(n.replace("k_policy_",""), k[n]["avg_ms"]) for n in ("k_policy_bwd","k_policy_fwd_train","k_rollout_persistent") if n in k))
PY
  done
done; done
