#!/bin/bash
# train-tile path: parity tests, then A/B of the strong-scaling shards (PPO_TRAIN_TILE_MAX_TILES=0 vs on)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r3c}; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_deep_policy.py tests/test_gpu_parity.py -x -q -k "deep or gradient or step_batch or ppo_train or any_hidden or normalised or gae_advantage" > $O/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for e in 512 1024 256 2048; do
  for tt in 0 100000; do
    PPO_TRAIN_TILE_MAX_TILES=$tt PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/shard_${e}_tt$tt.json 2> $O/shard_${e}_tt$tt.err && python3 tools/show_bench.py $O/shard_${e}_tt$tt.json "envs=$e tt=$tt" || { tail -5 $O/shard_${e}_tt$tt.err; exit 1; }
  done
done
