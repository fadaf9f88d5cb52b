#!/bin/bash
# A/B of the compact-storage paths on the GPU box: tests that use the compact form, then the streamed config-5 bench, the
# compact-forced headline and the plain headline, alternating.  tools/r2_ab_compact.sh OUT "<variants>"
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
timeout -k 10 800 python3 -m pytest tests -m gpu -x -q -k "compact or storage or stream or disk or bf16_gradient or rollout_bitexact" > $O/gpu_tests.log 2>&1; rc=$?; tail -3 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do for v in $2; do
  if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
  for cfg in "c5s:--dtype bf16 --envs 65536 --steps 1 --warmup 1 --stream /tmp/ppo_ab_stream" "f32c:--steps 3 --warmup 1" "f32:--steps 3 --warmup 1"; do
    name=${cfg%%:*}; extra=${cfg#*:}
    if [ $name = f32c ]; then export PPO_ROLLOUT_COMPACT=1; else unset PPO_ROLLOUT_COMPACT; fi
    timeout -k 10 300 python3 bench.py --no-cpu-baseline $extra > $O/ab_${v}_$name.json 2> $O/ab_${v}_$name.err || { tail -5 $O/ab_${v}_$name.err; exit 1; }
    python3 - $O/ab_${v}_$name.json $v $name <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], sys.argv[3], "value %.0f"%d["value"], " ".join("%s %.4f"%(n.replace("k_policy_",""), k[n]["avg_ms"]) for n in ("k_policy_bwd","k_policy_fwd_train","k_rollout_persistent") if n in k))
PY
  done
done; done
rm -rf /tmp/ppo_ab_stream
