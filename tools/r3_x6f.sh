#!/bin/bash
# split-fp32 train forward: tests, then A/B by minibatch size (PPO_FWD_SPLIT_MAX_TILES = 0: fp32 forward; large: split forward)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6g}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_split_backward.py tests/test_gpu_parity.py -x -q -m gpu -k "split or gradient or adam or train or learning" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -8 $O/tests.log
[ $rc -eq 0 ] || exit 1
for e in 512 1024 2048 4096; do
  for m in 0 100000; do
    PPO_FWD_SPLIT_MAX_TILES=$m PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --envs $e > $O/shard_${e}_$m.json 2> $O/shard_${e}_$m.err && python3 tools/show_bench.py $O/shard_${e}_$m.json envs=$e fwdsplit=$m
  done
done
PPO_FWD_SPLIT_MAX_TILES=100000 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --hid 128 > $O/h128.json 2> $O/h128.err && python3 tools/show_bench.py $O/h128.json hid128 fwdsplit
