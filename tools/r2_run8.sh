#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2o; rm -rf $O; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "rollout or residue or smoke or episodes or average or two_ranks or generic" > $O/gpu_tests.log 2>&1; rc=$?; tail -8 $O/gpu_tests.log; echo "tests rc=$rc"
[ $rc -eq 0 ] || exit $rc
for e in 64 256 512; do for m in 0 512; do
  PPO_ROLLOUT_SPLIT_MAX_ENVS=$m PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --steps 5 --warmup 2 --envs $e --no-cpu-baseline > $O/shard_${e}_$m.json 2> $O/shard_${e}_$m.err || { tail -5 $O/shard_${e}_$m.err; exit 1; }
  python3 - $O/shard_${e}_$m.json "$m" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["envs_per_gpu"], "rollout_split", sys.argv[2], "ms/iter %.2f"%d["ms_per_step"], "rollout ms", d["kernels"]["k_rollout_persistent"]["avg_ms"])
PY
done; done
