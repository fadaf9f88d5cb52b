#!/bin/bash
# bf16 A/B on the GPU box: tools/r2_ab_bf16.sh OUT "<variants>"  (bf16 tests on the default library first, then alternating bench runs at 4096 and 65536 envs)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; rm -rf $O; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $O/gpu_tests.log 2>&1; rc=$?; tail -5 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for tv in $3; do   # the same tests on variant libraries
  PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$tv.so timeout -k 10 600 python3 -m pytest tests/test_gpu_bf16.py -m gpu -x -q > $O/gpu_tests_$tv.log 2>&1; rc=$?; tail -3 $O/gpu_tests_$tv.log; echo "tests($tv) rc=$rc"
  [ $rc -eq 0 ] || exit $rc
done
for rep in 1 2; do for v in $2; do
  if [ "$v" = default ]; then unset PPO_HIP_LIB; else export PPO_HIP_LIB=$PWD/proximalpolicyoptimization.jl_amd/libppo_hip_$v.so; fi
  for e in 4096 65536; do
    S=4; [ $e = 65536 ] && S=2
    timeout -k 10 300 python3 bench.py --steps $S --warmup 1 --no-cpu-baseline --dtype bf16 --envs $e > $O/ab_${v}_$e.json 2> $O/ab_${v}_$e.err || { tail -5 $O/ab_${v}_$e.err; exit 1; }
    python3 - $O/ab_${v}_$e.json $v $e <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d["kernels"]
print(sys.argv[2], "envs", sys.argv[3], "value %.0f"%d["value"], " ".join("%s %.4f"%(n.replace("k_policy_",""), k[n]["avg_ms"]) for n in ("k_policy_bwd","k_policy_dw1","k_policy_fwd_train","k_rollout_persistent","k_grad_reduce") if n in k))
PY
  done
done; done
