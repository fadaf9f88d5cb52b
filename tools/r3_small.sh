#!/bin/bash
# small minibatches: three-product fp32 backward (<= 384 tiles by default) vs the fused split-fp32 backward
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-small1}; mkdir -p $O
for e in 128 256 384; do
  for m in 384 0; do
    PPO_BWD_SMALL_MAX_TILES=$m PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/shard_${e}_$m.json 2> $O/shard_${e}_$m.err && python3 tools/show_bench.py $O/shard_${e}_$m.json envs=$e small_max=$m | cut -c1-330
  done
done
