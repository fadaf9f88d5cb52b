#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-fin1}; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_split_backward.py tests/test_gpu_parity.py tests/test_gpu_deep_policy.py -x -q -m gpu -k "split or gradient or adam or train" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/tests.log
[ $rc -eq 0 ] || exit 1
for e in 512 1024; do
  PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 --envs $e > $O/shard_$e.json 2> $O/shard_$e.err && python3 tools/show_bench.py $O/shard_$e.json envs=$e
done
timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err && python3 tools/show_bench.py $O/bench.json headline
