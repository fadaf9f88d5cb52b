#!/bin/bash
# split train forward, two tiles per workgroup pass (HID = 256, >= 1536 tiles): tests, then A/B against the one-tile form
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-x6m}; mkdir -p $O
timeout -k 10 400 python3 -m pytest tests/test_gpu_split_backward.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -eq 0 ] || exit 1
tail -5 gpurun_out/split_backward_accuracy.jsonl
for r in 1 2; do
for v in 0 1536; do
  PPO_FWD_SPLIT_T2_MIN_TILES=$v timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 > $O/bench_${v}_$r.json 2> $O/bench_${v}_$r.err && python3 tools/show_bench.py $O/bench_${v}_$r.json t2min=$v
done
done
for v in 0 1536; do
  PPO_FWD_SPLIT_T2_MIN_TILES=$v PPO_BENCH_FORCE_DIST=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 4 --warmup 2 --envs 2048 > $O/shard2048_${v}.json 2> $O/shard2048_${v}.err && python3 tools/show_bench.py $O/shard2048_${v}.json envs=2048 t2min=$v
done
